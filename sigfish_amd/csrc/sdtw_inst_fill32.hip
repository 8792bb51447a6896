// sdtw_inst_fill32.hip -- explicit instantiations (see sdtw_instances.hpp)
#include "sdtw_kernels.hpp"

namespace sfa {
template __global__ void sdtw_fill_kernel<32, false>(const DpArgs);
}  // namespace sfa
