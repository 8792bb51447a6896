// sdtw_inst_trace16.hip -- explicit instantiations (see sdtw_instances.hpp)
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_trace_kernel<16, false>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<16, true>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<8, false>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<8, true>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<4, false>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<4, true>(const DpArgs, int32_t *);
}  // namespace sfa
