// sdtw_inst_fill16.hip -- explicit instantiations (see sdtw_instances.hpp)
#include "sdtw_kernels.hpp"

namespace sfa {
template __global__ void sdtw_fill_kernel<16, false>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, false>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, false>(const DpArgs);
}  // namespace sfa
