// sdtw_inst_lck16.hip -- explicit instantiations (see sdtw_instances.hpp): fill with rolling checkpoints in LDS + its pass 2
#include "sdtw_kernels.hpp"

namespace sfa {
template __global__ void sdtw_fill_kernel<16, false, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, false, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, false, false, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<16, false, false, true, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<8, false, false, true, true>(const DpArgs);
template __global__ void sdtw_fill_kernel<4, false, false, true, true>(const DpArgs);
template __global__ void sdtw_trace_kernel<16, false, true>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<8, false, true>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<4, false, true>(const DpArgs, int32_t *);
}  // namespace sfa
