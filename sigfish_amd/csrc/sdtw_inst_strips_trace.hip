// sdtw_inst_strips_trace.hip -- row strips, unchained pass 2 (sdtw_strips.hpp)
#include "sdtw_strips.hpp"
namespace sfa {
template __global__ void sdtw_strip_kernel<false, true>(const StripArgs);
template __global__ void sdtw_strip_kernel<true, true>(const StripArgs);
}  // namespace sfa
