// sdtw_inst_seg32.hip -- explicit instantiations (see sdtw_instances.hpp): column-segment fill, R = 32
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_fill_kernel<32, false, true>(const DpArgs);
}  // namespace sfa
