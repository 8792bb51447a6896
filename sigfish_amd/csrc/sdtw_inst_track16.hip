// sdtw_inst_track16.hip -- explicit instantiations (see sdtw_instances.hpp)
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_fill_kernel<16, true, false>(const DpArgs);
template __global__ void sdtw_fill_kernel<16, true, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, true, false>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, true, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, true, false>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, true, true>(const DpArgs);
}  // namespace sfa
