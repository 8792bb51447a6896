// sdtw_inst_long_trace.hip -- explicit instantiations (see sdtw_instances.hpp): queries of 513..2048 events
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_trace_kernel<128, false>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<128, true>(const DpArgs, int32_t *);
}  // namespace sfa
