// sdtw_inst_long_track.hip -- explicit instantiations (see sdtw_instances.hpp): queries of 513..2048 events
#include "sdtw_kernels.hpp"
namespace sfa {
template __global__ void sdtw_fill_kernel<128, true, false>(const DpArgs);
template __global__ void sdtw_fill_kernel<128, true, true>(const DpArgs);
}  // namespace sfa
