// sdtw_inst_trace32.hip -- explicit instantiations (see sdtw_instances.hpp)
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_trace_kernel<32, false>(const DpArgs, int32_t *);
template __global__ void sdtw_trace_kernel<32, true>(const DpArgs, int32_t *);
}  // namespace sfa
