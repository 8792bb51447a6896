// sdtw_inst_std16.hip -- explicit instantiations (see sdtw_instances.hpp)
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_fill_kernel<16, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, true>(const DpArgs);
}  // namespace sfa
