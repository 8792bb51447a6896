// sdtw_inst_fused32.hip -- explicit instantiation (see sdtw_instances.hpp): the 32-row fill (queries of 257 .. 2048 events, snapshots
// in HBM) with pass 2 by ticket in the same launch
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_fill_kernel<32, false, false, false, true>(const DpArgs);
}  // namespace sfa
