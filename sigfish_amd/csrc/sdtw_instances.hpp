// sdtw_instances.hpp -- the DP kernels are instantiated in their own translation units (sdtw_inst_*.hip) so that
// the build can compile them in parallel; everybody else only sees these declarations.
#pragma once
#include "sdtw_kernels.hpp"

namespace sfa {
#define SFA_FILL_DECL(MR, SD) extern template __global__ void sdtw_fill_kernel<MR, SD>(const DpArgs);
#define SFA_TRACE_DECL(MR, SD) extern template __global__ void sdtw_trace_kernel<MR, SD>(const DpArgs, int32_t *);
#define SFA_FOR_MAXR(X, ...) X(4, __VA_ARGS__) X(8, __VA_ARGS__) X(16, __VA_ARGS__) X(32, __VA_ARGS__)
SFA_FOR_MAXR(SFA_FILL_DECL, false)  // snapshots in HBM, pass 2 as its own launch: sdtw_inst_fill16.hip / fill32.hip
SFA_FOR_MAXR(SFA_FILL_DECL, true)   // std_dtw: sdtw_inst_std16.hip / std32.hip
#define SFA_SEG_DECL(MR, ...) extern template __global__ void sdtw_fill_kernel<MR, false, true>(const DpArgs);
SFA_FOR_MAXR(SFA_SEG_DECL, 0)       // column segments (small batches): sdtw_inst_seg16.hip / seg32.hip
SFA_FOR_MAXR(SFA_TRACE_DECL, false)
SFA_FOR_MAXR(SFA_TRACE_DECL, true)
// LDS-checkpoint variants (subsequence DTW, R <= 16), alone and with pass 2 by ticket in the same launch: sdtw_inst_lck16.hip
#define SFA_LCK_DECL(MR)                                                                         \
    extern template __global__ void sdtw_fill_kernel<MR, false, false, true>(const DpArgs);       \
    extern template __global__ void sdtw_fill_kernel<MR, false, false, true, true>(const DpArgs); \
    extern template __global__ void sdtw_trace_kernel<MR, false, true>(const DpArgs, int32_t *);
SFA_LCK_DECL(4) SFA_LCK_DECL(8) SFA_LCK_DECL(16)
// ... and for std_dtw (no LDS snapshots, sparse HBM store; pass 2 by ticket): sdtw_inst_lckstd16.hip
#define SFA_LCKSTD_DECL(MR)                                                                     \
    extern template __global__ void sdtw_fill_kernel<MR, true, false, true>(const DpArgs);       \
    extern template __global__ void sdtw_fill_kernel<MR, true, false, true, true>(const DpArgs); \
    extern template __global__ void sdtw_trace_kernel<MR, true, true>(const DpArgs, int32_t *);
SFA_LCKSTD_DECL(4) SFA_LCKSTD_DECL(8) SFA_LCKSTD_DECL(16)
// the 32-row fill (snapshots in HBM) with pass 2 by ticket in the same launch: sdtw_inst_fused32.hip
extern template __global__ void sdtw_fill_kernel<32, false, false, false, true>(const DpArgs);
}  // namespace sfa
