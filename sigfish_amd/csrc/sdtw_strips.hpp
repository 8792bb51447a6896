// sdtw_strips.hpp -- queries longer than 2048 events (`-q` beyond SFA_MAX_QUERY): ROW STRIPS.
//
// The reference has no limit on the query length: subsequence() / std_dtw() (src/cdtw.c:171-189, 69-94) fill a
// qlen x rlen matrix whatever qlen is.  The wave kernels of sdtw_kernels.hpp keep a whole query in the registers of one
// wave (64 lanes x 32 rows = 2048 events).  A longer query is cut into strips of 2048 consecutive query rows that one
// wave sweeps one after the other over the same (contig,strand):
//   * strip 0 is the ordinary sweep (row 0 has the free start of subsequence() or the cumulative first row of
//     std_dtw());
//   * the LAST row of every strip but the final one is written to HBM as it is produced -- cost and carried start
//     column of every reference column, 8 bytes per column -- and is the "row above" of the next strip: lane 0 of the
//     next sweep takes its `up` input (and, one step later, its diagonal input) from that row instead of the constant
//     boundary;
//   * the final strip holds the last query row: the window scan of src/sigfish.c:891-901 / 938-948 and the top-2 of
//     update_aln (src/sigfish.c:575-626) run there, with the start column tracked forward by the traceback rule of
//     path() (diagonal, then left, then up; src/cdtw.c:134-146) exactly as in the single-pass kernels.
// Every cell is still a pure function of its three neighbours in fp32, so the rows are bit-identical to the
// reference's; the boundary rows are stored and re-read as the same 32-bit patterns.
// One pass with tracking (7 VALU per cell instead of 3): a query of this length is 8x the default and rare; the strip
// path trades the checkpoint machinery for simplicity.  HBM traffic: 16 bytes per column and strip boundary against
// 2048 x 7 lane-operations -- nowhere near a bound.
#pragma once

#include "sdtw_kernels.hpp"

namespace sfa {

constexpr int kStripR = 32;                 // query rows per lane
constexpr int kStripRows = 64 * kStripR;    // query rows per strip
constexpr int kBndPad = 256;                // words behind every boundary row (block over-run of the sweep + prefetch)

struct StripArgs {
    const float *queries;     // HBM: concatenated z-normalised event means
    const int64_t *q_off;     // [n_reads+1]
    const int32_t *reads;     // [n_long] batch index of every long read
    const float *ref;         // padded reference event arrays (as DpArgs::ref)
    const int64_t *job_off;   // [n_jobs]
    const int32_t *job_len;   // [n_jobs]
    const int64_t *bnd_off;   // [n_jobs+1] word offset of job j's boundary row inside one buffer (multiples of 4)
    float *bnd_cost;          // [n_long][2 buffers][bnd_off[n_jobs]] last-row costs of the previous / current strip
    int32_t *bnd_start;       // same, carried start columns
    float *p_best, *p_second; // partial top-2 per (long read, job)
    int32_t *p_end, *p_st;
    int32_t n_long, n_jobs, rev_query;
};

// One anti-diagonal step of a strip: dp_step<32, TRACK = true> with the handling of query row 0 made conditional on
// FIRST (the strip that contains it).
template <bool STD, bool FIRST>
__device__ __forceinline__ void strip_step(typename Vec<float, kStripR>::type &c, typename Vec<int, kStripR>::type &s, float &dprev,
                                           int &sdprev, const float (&x)[kStripR], const float yv, const int t, const bool lane0,
                                           Exchange &xc) {
    float up = xc.shift(static_cast<float>(c[kStripR - 1]));
    int sup = xc.shift(static_cast<int>(s[kStripR - 1]));
    if (STD && FIRST) {
        if (t == 0) xc.template set_boundary<true>(lane0, INFINITY);  // std_dtw(): row 0 continues from its left neighbour only
    }
    float diag = dprev;
    int sdiag = sdprev;
    dprev = up;
    sdprev = sup;
    if (STD && FIRST) dprev = (t == 0) ? INFINITY : dprev;  // there is no column -1 next to the free corner
#pragma unroll
    for (int r = 0; r < kStripR; ++r) {
        const float left = c[r];
        const int sleft = s[r];
        float m;
        if (r == 0) {  // values out of LDS: unsigned min of the bit patterns (non-negative floats), as in dp_step
            const unsigned mu = min(min(__float_as_uint(up), __float_as_uint(diag)), __float_as_uint(left));
            m = __uint_as_float(mu);
        } else {
            m = fminf(fminf(up, diag), left);
        }
        const float cn = fabsf(x[r] - yv) + m;
        int sn = (diag == m) ? sdiag : ((left == m) ? sleft : sup);  // src/cdtw.c:134-146
        if (FIRST && r == 0) sn = lane0 ? t : sn;                    // query row 0: the path starts in this column
        diag = left;
        sdiag = sleft;
        up = cn;
        sup = sn;
        c[r] = cn;
        s[r] = sn;
    }
}

struct __attribute__((aligned(16))) int4a {
    int v[4];
};
struct __attribute__((aligned(16))) float4a {
    float v[4];
};

// One strip of one (contig,strand).  `last`: the strip holds the last query row (lane lq, register rq).
template <bool STD, bool FIRST>
__device__ __forceinline__ void strip_sweep(const float *yp, const int rlen, const int qlen, const bool last, const float (&x)[kStripR],
                                            const int lq, const int rq, const int lane, Exchange &xc, const float *bin_c,
                                            const int32_t *bin_s, float *bout_c, int32_t *bout_s, Top2<true> &top, const int job) {
    typename Vec<float, kStripR>::type c;
    typename Vec<int, kStripR>::type s;
#pragma unroll
    for (int r = 0; r < kStripR; ++r) {
        c[r] = INFINITY;
        s[r] = 0;
    }
    float dprev = INFINITY;
    int sdprev = 0;
    const bool lane0 = lane == 0;
    if (FIRST) xc.template set_boundary<true>(lane0, 0.0f);

    // window scan of the last row (final strip only), src/sigfish.c:891-901; std_dtw has the single candidate C[n-1][m-1]
    float wmin = INFINITY;
    int wpos = -1, wst = -1;
    int wend = min(qlen, rlen);

    const int n_steps = rlen + lq;  // lane lq meets the last column at step rlen - 1 + lq
    float4u ycur = *reinterpret_cast<const float4u *>(yp);
    float4a bc{};
    int4a bs{};
    if (!FIRST) {
        bc = *reinterpret_cast<const float4a *>(bin_c);
        bs = *reinterpret_cast<const int4a *>(bin_s);
    }
    for (int t0 = 0; t0 < n_steps; t0 += 4) {
        const float4u ynext = *reinterpret_cast<const float4u *>(yp + t0 + 4);
        float4a bcn{};
        int4a bsn{};
        if (!FIRST) {
            bcn = *reinterpret_cast<const float4a *>(bin_c + t0 + 4);
            bsn = *reinterpret_cast<const int4a *>(bin_s + t0 + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            if (!FIRST) {  // the row above query row 0 of this strip: column t of the previous strip's last row
                if (lane0) {
                    *((Exchange::lds_vf *)xc.rf) = bc.v[u];
                    *((Exchange::lds_vi *)xc.ri) = bs.v[u];
                }
            }
            strip_step<STD, FIRST>(c, s, dprev, sdprev, x, ycur.v[u], t, lane0, xc);
            if (!last) {  // (wave-uniform) lane 63 is at column t - 63 of the strip's last row
                const int col = t - 63;
                if (lane == 63 && col >= 0 && col < rlen) {
                    bout_c[col] = c[kStripR - 1];
                    bout_s[col] = s[kStripR - 1];
                }
            } else {
                const int col = t - lq;  // wave-uniform
                if (col >= 0 && col < rlen) {
                    const float cl = c[rq];
                    const int sl = s[rq];
                    if (!STD) {
                        const bool lt = cl < wmin;  // first strict minimum of the window
                        wmin = lt ? cl : wmin;
                        wpos = lt ? col : wpos;
                        wst = lt ? sl : wst;
                        if (col + 1 == wend) {
                            top.offer(wmin, wpos, wst, job);
                            wmin = INFINITY;
                            wpos = -1;
                            wst = -1;
                            wend = min(wend + qlen, rlen);
                        }
                    } else if (col == rlen - 1) {
                        top.offer(cl, col, sl, job);
                    }
                }
            }
        }
        ycur = ynext;
        bc = bcn;
        bs = bsn;
    }
}

// wave-task = (long read, job); 4 waves per block
template <bool STD>
__global__ void __launch_bounds__(256, 1) sdtw_strip_kernel(const StripArgs a) {
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (task >= a.n_long * a.n_jobs) return;  // wave-uniform
    const int job = task / a.n_long, li = task - job * a.n_long;  // job-major: neighbouring waves stream the same reference
    const int lane = threadIdx.x & 63;
    __shared__ float lds_f[4 * kXchWordsPerWave];
    __shared__ int lds_i[4 * kXchWordsPerWave];
    Exchange xc;
    xc.init(lds_f, lds_i, threadIdx.x >> 6, 0, lane, 64);

    const int read = a.reads[li];
    const int64_t qo = a.q_off[read];
    const int qlen = static_cast<int>(a.q_off[read + 1] - qo);
    const float *q = a.queries + qo;
    const int n_strips = (qlen + kStripRows - 1) / kStripRows;
    const int rlen = a.job_len[job];
    const float *yp = a.ref + a.job_off[job] - lane;  // this lane's column at step t is t - lane
    const int64_t per = a.bnd_off[a.n_jobs];
    float *bc = a.bnd_cost + static_cast<int64_t>(li) * 2 * per + a.bnd_off[job];
    int32_t *bs = a.bnd_start + static_cast<int64_t>(li) * 2 * per + a.bnd_off[job];

    Top2<true> top;
    top.init();
    int lq = 0;
    for (int sidx = 0; sidx < n_strips; ++sidx) {
        const int row0 = sidx * kStripRows;
        const int rows = min(kStripRows, qlen - row0);
        const bool last = sidx == n_strips - 1;
        lq = (rows - 1) / kStripR;
        const int rq = (rows - 1) - lq * kStripR;
        float x[kStripR];
#pragma unroll
        for (int r = 0; r < kStripR; ++r) {
            const int i = row0 + lane * kStripR + r;
            const int src = a.rev_query ? (qlen - 1 - i) : i;
            x[r] = (i < qlen) ? q[src] : 0.0f;
        }
        float *bout_c = bc + (sidx & 1) * per;
        int32_t *bout_s = bs + (sidx & 1) * per;
        const float *bin_c = bc + ((sidx & 1) ^ 1) * per;
        const int32_t *bin_s = bs + ((sidx & 1) ^ 1) * per;
        if (sidx == 0)
            strip_sweep<STD, true>(yp, rlen, qlen, last, x, lq, rq, lane, xc, bin_c, bin_s, bout_c, bout_s, top, job);
        else
            strip_sweep<STD, false>(yp, rlen, qlen, last, x, lq, rq, lane, xc, bin_c, bin_s, bout_c, bout_s, top, job);
        // the boundary row was stored by lane 63 and is loaded by every lane of the same wave in the next strip: complete
        // the stores and drop the lines the vector cache may still hold from two strips ago
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (lane == lq) {
        const int64_t o = static_cast<int64_t>(li) * a.n_jobs + job;
        a.p_best[o] = top.best;
        a.p_second[o] = top.second;
        a.p_end[o] = top.end;
        a.p_st[o] = top.st;
    }
}

// instantiated in sdtw_inst_strips.hip
extern template __global__ void sdtw_strip_kernel<false>(const StripArgs);
extern template __global__ void sdtw_strip_kernel<true>(const StripArgs);

struct StripFinalizeArgs {
    const int32_t *reads;  // [n_long]
    const float *p_best, *p_second;
    const int32_t *p_end, *p_st;
    const int32_t *job_contig;
    const int8_t *job_strand;
    const int32_t *ref_len, *ref_st_offset;
    ResultRow *out;  // rows of the whole batch
    int32_t n_long, n_jobs;
};

#ifdef SFA_DEFINE_FINALIZE_KERNEL
// merge the per-job top-2 of a long read in processing order (a later job wins ties, src/sigfish.c:577-583), then strand
// flip, offset and mapq (src/sigfish.c:969-983) -- the single-pass branch of sdtw_finalize_kernel for the strip path
__global__ void __launch_bounds__(64) sdtw_strip_finalize_kernel(const StripFinalizeArgs a) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= a.n_long) return;
    float best = INFINITY, second = INFINITY;
    int end = -1, st = -1, job = -1;
    for (int j = 0; j < a.n_jobs; ++j) {
        const int64_t o = static_cast<int64_t>(li) * a.n_jobs + j;
        const float b = a.p_best[o], s2 = a.p_second[o];
        const float hi = fmaxf(best, b);
        const float lo2 = fminf(second, s2);
        const bool take = !(b > best);
        second = fminf(hi, lo2);
        if (take) {
            best = b;
            end = a.p_end[o];
            st = a.p_st[o];
            job = j;
        }
    }
    ResultRow r;
    r.rid = -1;
    r.pos_st = -1;
    r.pos_end = -1;
    r.score = best;
    r.score2 = second;
    r.strand = 0;
    r.mapq = 0;
    r.valid = 1;
    r.pad = 0;
    if (job >= 0 && end >= 0) {
        const int rid = a.job_contig[job];
        const int8_t d = a.job_strand[job];
        const int rl = a.ref_len[rid], off = a.ref_st_offset[rid];
        r.rid = rid;
        r.strand = d;
        r.mapq = mapq_from_scores(best, second);
        r.pos_st = ((d == '+') ? st : rl - end) + off;
        r.pos_end = ((d == '+') ? end : rl - st) + off;
    }
    a.out[a.reads[li]] = r;
}
#endif

}  // namespace sfa
