// sdtw_inst_strips_pipe.hip -- row strips, pipelined pass 1 (sdtw_strips.hpp)
#include "sdtw_strips.hpp"
namespace sfa {
template __global__ void sdtw_strip_pipe_kernel<false>(const StripArgs);
template __global__ void sdtw_strip_pipe_kernel<true>(const StripArgs);
}  // namespace sfa
