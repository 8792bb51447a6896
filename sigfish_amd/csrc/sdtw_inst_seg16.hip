// sdtw_inst_seg16.hip -- explicit instantiations (see sdtw_instances.hpp): column-segment fill, R <= 16
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_fill_kernel<16, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, false, true>(const DpArgs);
}  // namespace sfa
