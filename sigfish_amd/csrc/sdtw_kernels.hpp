// sdtw_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels for the subsequence-DTW alignment stage.
//
// What is computed (reference: hasindu2008/sigfish v0.2.0):
//   * accumulated-cost recurrence of subsequence() / std_dtw()      src/cdtw.c:171-189 / 69-94
//   * the alignment start column that subsequence_path() would       src/cdtw.c:98-167, 192-227
//     trace back to, propagated FORWARD with the same tie order
//     (diagonal, then left, then up), so no matrix is ever stored
//   * the per-window first-strict-minimum scan of the last row       src/sigfish.c:891-901, 938-948
//   * the top-2 of the reference's sorted top-5 candidate list       src/sigfish.c:575-626 (ties: later wins)
//   * strand flip, ref_st_offset, mapq                               src/sigfish.c:969-983
//
// Mapping to the machine (MI355X-first, not a translation of the CPU loops):
//   * one read occupies ONE DPP ROW (16 lanes) of a wave64; four reads ride in a wave.  Lane g of a row owns
//     R consecutive query rows (R = 4/8/16/32 -> queries up to 64/128/256/512 events), kept in VGPRs together
//     with their running cost and start column.  No LDS, no barriers, no cost matrix.
//   * the lanes of a row walk an anti-diagonal: at step t lane g is at reference column t-g, so the only
//     cross-lane traffic per step is ONE `v_mov_b32_dpp row_shr:1` of the bottom cost (and one of its start
//     column): the neighbour's value from the previous step is this lane's "up", the one before its "diagonal".
//   * +inf initial state makes not-yet-started columns (t-g < 0) and past-the-end columns harmless, so the inner
//     loop carries no per-lane predication; reference arrays are padded in HBM so the per-lane 16-byte loads of
//     four upcoming reference levels never leave the allocation.
//   * per cell: v_sub, v_min3, v_add(|d|) (+ 2 v_cmp_eq, 2 v_cndmask when the start column is tracked).
//   * all arithmetic is IEEE fp32 with denormals, no FMA contraction: every cell is bit-identical to the
//     reference's row-major evaluation because each cell is a pure function of its three neighbours.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfa {

constexpr int kLanesPerRead = 16;  // one DPP row
constexpr int kReadsPerWave = 4;
constexpr int kRefPad = 64;        // floats of padding on both sides of every (contig,strand) array in HBM
constexpr int kStepsPerLoad = 4;   // reference levels fetched per 16-byte load

struct __attribute__((packed, aligned(4))) float4u {
    float v[4];
};

// Arguments of one fill launch (one query-length class R).
struct FillArgs {
    const float *queries;        // HBM: concatenated z-normalised event means, event order
    const int64_t *q_off;        // [n_reads+1]
    const int32_t *order;        // [n_quads_total*4] read index per (quad, slot) or -1
    const int32_t *quad_qlen;    // [n_quads_total] query length shared by the quad's reads
    const float *ref;            // padded reference event arrays
    const int64_t *job_off;      // [n_jobs] offset of column 0 of job (contig,strand) in `ref`
    const int32_t *job_len;      // [n_jobs] rlen
    const int32_t *chunk_begin;  // [n_chunks+1] job ranges
    float *p_best;               // partial results, index (quad*n_chunks+chunk)*4+slot
    int32_t *p_end;
    int32_t *p_st;
    int32_t *p_job;
    float *p_second;
    int32_t quad_base;  // first quad of this class
    int32_t n_quads;    // quads in this class
    int32_t n_chunks;
    int32_t n_tasks;    // n_quads * n_chunks
    int32_t rev_query;  // 1: query rows are the events reversed (RNA without --invert)
};

__device__ __forceinline__ float dpp_row_shr1_zero(float v) {
    // lane g receives lane g-1 of its 16-lane row; lane 0 of every row receives 0.0f (bound_ctrl)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_shr1_old(float old, float v) {
    // same, but lane 0 of every row keeps `old`
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ int dpp_row_shr1_zero(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
}

// Wave-uniform register pick: c[idx] for a scalar idx without turning into R v_cndmasks.
template <int R, typename T>
__device__ __forceinline__ T pick_uniform(const T (&a)[R], int idx) {
    T out = a[R - 1];
#define SFA_PICK(r)                               \
    case r:                                       \
        if (r < R - 1) {                          \
            out = a[(r < R - 1) ? r : 0];         \
            asm volatile("" : "+v"(out));         \
        }                                         \
        break;
    switch (idx) {
        SFA_PICK(0) SFA_PICK(1) SFA_PICK(2) SFA_PICK(3) SFA_PICK(4) SFA_PICK(5) SFA_PICK(6) SFA_PICK(7)
        SFA_PICK(8) SFA_PICK(9) SFA_PICK(10) SFA_PICK(11) SFA_PICK(12) SFA_PICK(13) SFA_PICK(14) SFA_PICK(15)
        SFA_PICK(16) SFA_PICK(17) SFA_PICK(18) SFA_PICK(19) SFA_PICK(20) SFA_PICK(21) SFA_PICK(22) SFA_PICK(23)
        SFA_PICK(24) SFA_PICK(25) SFA_PICK(26) SFA_PICK(27) SFA_PICK(28) SFA_PICK(29) SFA_PICK(30)
        default: break;
    }
#undef SFA_PICK
    return out;
}

// Running top-2 of the reference's candidate list for one read (kept in the registers of the lane that owns
// the last query row).  Insertion rule of update_aln (src/sigfish.c:577-583): a candidate goes in front of
// everything that is not strictly better, so on equal scores the LATER candidate ranks higher.
struct Top2 {
    float best, second;
    int32_t end, st, job;
    __device__ __forceinline__ void init() {
        best = INFINITY;
        second = INFINITY;
        end = -1;
        st = -1;
        job = -1;
    }
    __device__ __forceinline__ void offer(float sc, int32_t pos, int32_t start, int32_t j) {
        const bool top = !(sc > best);
        const bool sec = !(sc > second);
        second = top ? best : (sec ? sc : second);
        best = top ? sc : best;
        end = top ? pos : end;
        st = top ? start : st;
        job = top ? j : job;
    }
};

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): each XCD gets a contiguous range of logical
// blocks, hence (tasks being chunk-major) mostly one reference chunk per XCD L2.  Speed only.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblk) {
    const int x = b & 7, i = b >> 3;
    const int q = nblk >> 3, rem = nblk & 7;
    return x * q + (x < rem ? x : rem) + i;
}

// ---------------------------------------------------------------------------------------------------------
// Fill kernel.  R rows per lane; TRACK: carry the alignment start column; STD: std_dtw instead of subsequence.
// grid: ceil(n_tasks/4) blocks of 256 threads (4 waves, 1 task each).
// ---------------------------------------------------------------------------------------------------------
template <int R, bool TRACK, bool STD>
__global__ void __launch_bounds__(256) sdtw_fill_kernel(const FillArgs a) {
    const int lblk = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int task = __builtin_amdgcn_readfirstlane(lblk * 4 + (threadIdx.x >> 6));
    if (task >= a.n_tasks) return;  // wave-uniform
    const int chunk = task / a.n_quads;
    const int quad = a.quad_base + (task - chunk * a.n_quads);
    const int lane = threadIdx.x & 63;
    const int g = lane & (kLanesPerRead - 1);
    const int slot = lane >> 4;
    const bool lane0 = (g == 0);

    const int qlen = a.quad_qlen[quad];
    const int read = a.order[quad * 4 + slot];
    const int lq = (qlen - 1) / R;  // lane / register holding the last query row (wave-uniform)
    const int rq = (qlen - 1) - lq * R;

    float x[R];
    {
        const float *q = a.queries + a.q_off[read >= 0 ? read : 0];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = g * R + r;
            const int src = a.rev_query ? (qlen - 1 - i) : i;
            x[r] = (read >= 0 && i < qlen) ? q[src] : 0.0f;
        }
    }

    Top2 top;
    top.init();

    const int jb = a.chunk_begin[chunk], je = a.chunk_begin[chunk + 1];
    for (int job = jb; job < je; ++job) {
        const int rlen = a.job_len[job];
        const float *yp = a.ref + a.job_off[job] - g;  // this lane's column at step t is t-g

        float c[R];
        int s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c[r] = INFINITY;
            s[r] = 0;
        }
        float dprev = INFINITY;  // "up" input of the previous step = diagonal input of this one
        int sdprev = 0;
        float wmin = INFINITY;  // running minimum of the current last-row window
        int wpos = -1, wst = -1;
        int wleft = qlen;

        const int nsteps = rlen + lq;  // lane lq sees column rlen-1 at step rlen-1+lq
        float4u ycur = *reinterpret_cast<const float4u *>(yp);
        for (int t0 = 0; t0 < nsteps; t0 += kStepsPerLoad) {
            const float4u ynext = *reinterpret_cast<const float4u *>(yp + t0 + kStepsPerLoad);
#pragma unroll
            for (int u = 0; u < kStepsPerLoad; ++u) {
                const int t = t0 + u;
                const float yv = ycur.v[u];
                // inputs from the lane above (row g*R-1): lane 0 owns query row 0 and gets the boundary instead
                float up;
                if (!STD) {
                    up = dpp_row_shr1_zero(c[R - 1]);  // row 0 of subsequence(): C[0][j] = d + 0
                } else {
                    up = dpp_row_shr1_old((t == 0) ? 0.0f : INFINITY, c[R - 1]);  // C[0][0]=d, C[0][j]=d+C[0][j-1]
                }
                int sup = 0;
                if (TRACK) sup = dpp_row_shr1_zero(s[R - 1]);
                float diag = dprev;
                int sdiag = sdprev;
                dprev = up;
                sdprev = sup;
                if (STD && t == 0) dprev = INFINITY;  // there is no column -1
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float left = c[r];
                    const int sleft = s[r];
                    const float m = fminf(fminf(up, diag), left);  // v_min3_f32; no NaNs on this path
                    const float cn = fabsf(x[r] - yv) + m;
                    int sn = 0;
                    if (TRACK) {
                        // traceback order of path(): diagonal first, then left, then up (src/cdtw.c:134-146)
                        sn = (diag == m) ? sdiag : ((left == m) ? sleft : sup);
                        if (r == 0) sn = lane0 ? t : sn;  // query row 0: the path starts in this column
                    }
                    diag = left;
                    sdiag = sleft;
                    up = cn;
                    sup = sn;
                    c[r] = cn;
                    s[r] = sn;
                }
                // last query row: windowed first-strict-minimum scan (wave-uniform control flow)
                const int jq = t - lq;
                if (jq >= 0 && jq < rlen) {
                    const float cl = pick_uniform<R>(c, rq);
                    const int sl = TRACK ? pick_uniform<R>(s, rq) : 0;
                    if (!STD) {
                        const bool lt = cl < wmin;
                        wmin = lt ? cl : wmin;
                        wpos = lt ? jq : wpos;
                        wst = lt ? sl : wst;
                        if (--wleft == 0 || jq == rlen - 1) {
                            top.offer(wmin, wpos, wst, job);
                            wmin = INFINITY;
                            wpos = -1;
                            wst = -1;
                            wleft = qlen;
                        }
                    } else if (jq == rlen - 1) {
                        top.offer(cl, jq, sl, job);  // std_dtw: the single candidate C[n-1][m-1]
                    }
                }
            }
            ycur = ynext;
        }
    }

    if (g == lq && read >= 0) {
        const int64_t o = (static_cast<int64_t>(quad) * a.n_chunks + chunk) * 4 + slot;
        a.p_best[o] = top.best;
        a.p_second[o] = top.second;
        a.p_end[o] = top.end;
        a.p_st[o] = top.st;
        a.p_job[o] = top.job;
    }
}

// One result row per read, POD mirror of sfa_result_t (include/sigfish_amd.h).
struct ResultRow {
    int32_t rid, pos_st, pos_end;
    float score, score2;
    int8_t strand;
    uint8_t mapq, valid, pad;
};

struct FinalizeArgs {
    const int32_t *slot_of_read;  // [n_reads] quad*4+slot, or -1 for skipped reads
    const float *p_best;
    const int32_t *p_end;
    const int32_t *p_st;
    const int32_t *p_job;
    const float *p_second;
    const int32_t *job_contig;  // [n_jobs]
    const int8_t *job_strand;   // [n_jobs] '+' / '-'
    const int32_t *ref_len;     // [num_ref]
    const int32_t *ref_st_offset;
    ResultRow *out;  // [n_reads]
    int32_t n_reads, n_chunks;
};

// src/sigfish.c:979-983: (int)round(500*(score2-score)/score) with x86 cvttsd2si saturation, cap 60, store u8
__device__ __forceinline__ uint8_t mapq_from_scores(float score, float score2) {
    const float v = 500.0f * (score2 - score) / score;
    const float r = roundf(v);  // == round((double)v) for float inputs
    int q;
    if (!(r >= -2147483648.0f && r < 2147483648.0f))
        q = INT32_MIN;
    else
        q = static_cast<int>(r);
    if (q > 60) q = 60;
    return static_cast<uint8_t>(q);
}

__global__ void __launch_bounds__(256) sdtw_finalize_kernel(const FinalizeArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_reads) return;
    ResultRow r;
    r.rid = -1;
    r.pos_st = -1;
    r.pos_end = -1;
    r.score = INFINITY;
    r.score2 = INFINITY;
    r.strand = 0;
    r.mapq = 0;
    r.valid = 0;
    r.pad = 0;
    const int sl = a.slot_of_read[i];
    if (sl >= 0) {
        const int64_t quad = sl >> 2, slot = sl & 3;
        float best = INFINITY, second = INFINITY;
        int end = -1, st = -1, job = -1;
        for (int ch = 0; ch < a.n_chunks; ++ch) {  // chunks in processing order: a later chunk wins ties
            const int64_t o = (quad * a.n_chunks + ch) * 4 + slot;
            const float b = a.p_best[o], s2 = a.p_second[o];
            const float hi = fmaxf(best, b);
            const float lo2 = fminf(second, s2);
            const bool take = !(b > best);
            second = fminf(hi, lo2);  // second smallest of {best, second, b, s2}
            if (take) {
                best = b;
                end = a.p_end[o];
                st = a.p_st[o];
                job = a.p_job[o];
            }
        }
        r.valid = 1;
        r.score = best;
        r.score2 = second;
        if (job >= 0) {
            const int rid = a.job_contig[job];
            const int8_t d = a.job_strand[job];
            const int rl = a.ref_len[rid];
            const int off = a.ref_st_offset[rid];
            r.rid = rid;
            r.strand = d;
            r.pos_st = ((d == '+') ? st : rl - end) + off;   // src/sigfish.c:971-975
            r.pos_end = ((d == '+') ? end : rl - st) + off;
            r.mapq = mapq_from_scores(best, second);
        }
    }
    a.out[i] = r;
}

}  // namespace sfa
