// sdtw_inst_lckstd16.hip -- explicit instantiations (see sdtw_instances.hpp): the std_dtw fill without per-interval HBM
// checkpoints (sparse store only) and with pass 2 by ticket in the same launch
#include "sdtw_kernels.hpp"

namespace sfa {
template __global__ void sdtw_fill_kernel<16, true, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, true, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, true, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<16, true, false, true, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, true, false, true, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, true, false, true, true>(const DpArgs);
template __global__ void sdtw_trace_kernel<16, true, true>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<8, true, true>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<4, true, true>(const DpArgs, int32_t *);
}  // namespace sfa
