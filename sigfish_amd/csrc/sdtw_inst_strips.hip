// sdtw_inst_strips.hip -- explicit instantiations of the row-strip kernels (queries beyond 2048 events, sdtw_strips.hpp):
// one wave per (read, job) / per read with all strips over the whole range (the alternatives to the pipelined pass 1 and the
// chained pass 2, which are instantiated in their own units so that `make -j` builds them side by side)
#include "sdtw_strips.hpp"
namespace sfa {
template __global__ void sdtw_strip_kernel<false, false>(const StripArgs);
template __global__ void sdtw_strip_kernel<true, false>(const StripArgs);
}  // namespace sfa
