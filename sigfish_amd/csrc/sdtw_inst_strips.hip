// sdtw_inst_strips.hip -- explicit instantiations of the row-strip kernels (queries beyond 2048 events, sdtw_strips.hpp)
#include "sdtw_strips.hpp"
namespace sfa {
template __global__ void sdtw_strip_kernel<false, false>(const StripArgs);
template __global__ void sdtw_strip_kernel<true, false>(const StripArgs);
template __global__ void sdtw_strip_kernel<false, true>(const StripArgs);
template __global__ void sdtw_strip_kernel<true, true>(const StripArgs);
template __global__ void sdtw_strip_chain_kernel<false>(const StripArgs);
template __global__ void sdtw_strip_chain_kernel<true>(const StripArgs);
template __global__ void sdtw_strip_pipe_kernel<false>(const StripArgs);
template __global__ void sdtw_strip_pipe_kernel<true>(const StripArgs);
}  // namespace sfa
