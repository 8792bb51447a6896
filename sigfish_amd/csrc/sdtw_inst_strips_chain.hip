// sdtw_inst_strips_chain.hip -- row strips, chained pass 2 (sdtw_strips.hpp)
#include "sdtw_strips.hpp"
namespace sfa {
template __global__ void sdtw_strip_chain_kernel<false>(const StripArgs);
template __global__ void sdtw_strip_chain_kernel<true>(const StripArgs);
}  // namespace sfa
