// sdtw_strips.hpp -- queries longer than 2048 events (`-q` beyond SFA_MAX_QUERY): ROW STRIPS.
//
// The reference has no limit on the query length: subsequence() / std_dtw() (src/cdtw.c:171-189, 69-94) fill a
// qlen x rlen matrix whatever qlen is.  The wave kernels of sdtw_kernels.hpp keep a whole query in the registers of one
// wave (64 lanes x 32 rows = 2048 events).  A longer query is cut into strips of 2048 consecutive query rows that one
// wave sweeps one after the other over the same (contig,strand):
//   * strip 0 is the ordinary sweep (row 0 has the free start of subsequence() or the cumulative first row of
//     std_dtw());
//   * the LAST row of every strip but the final one is written to HBM as it is produced, one value per reference
//     column, and is the "row above" of the next strip: lane 0 of the next sweep takes its `up` input (and, one step
//     later, its diagonal input) from that row instead of the constant boundary;
//   * the final strip holds the last query row: the window scan of src/sigfish.c:891-901 / 938-948 and the top-2 of
//     update_aln (src/sigfish.c:575-626) run there.
// TWO PASSES, as in the wave kernels.  Pass 1 (sdtw_strip_pipe_kernel, one wave per (read, contig, strand, STRIP), strip s + 1
// following strip s through HBM a block of columns behind) evaluates costs only -- v_sub, v_min3, v_add per cell -- keeps every
// window's first strict minimum and its column, and drops a checkpoint of every strip's anti-diagonal state (up to 32 costs + the
// diagonal input per lane) every T steps; the strip finalize merges the per-job top-2 and names the winning (job, cell, score).
// Pass 2 (sdtw_strip_chain_kernel, one wave per read) traces the WINNING job strip by strip from the last one upwards, each
// strip from a checkpoint in front of its target cell, with the column through which the path enters the strip carried forward
// by the traceback rule of path() (diagonal, then left, then up; src/cdtw.c:134-146); the cell a strip has to reach is known
// by column and cost, compared bit for bit.  Restored cells carry -1; if -1 reaches the target cell the path entered before the
// checkpoint and the strip backs off 1, 2, 4 ... checkpoints, ultimately to column 0.  Costs are restored bit-exactly and both
// passes evaluate the same pure fp32 function of three neighbours per cell, so costs agree bit for bit with each other and with
// the reference.
// HBM traffic: 4 bytes per column and strip boundary + 8.4 KB per checkpoint and strip, against 2048 x 3..7 lane-operations per
// column.
#pragma once

#include "sdtw_kernels.hpp"

namespace sfa {

constexpr int kStripR = 32;                 // query rows per lane, at most
constexpr int kStripRows = 64 * kStripR;    // query rows per strip, at most: a query has ceil(qlen / 2048) strips
constexpr int kCkRec = (kStripR + 1) * 64;  // floats per checkpoint record: a plane of 64 lanes per row of a lane (up to 32) ...
constexpr int kCkDprev = kStripR * 64;      // ... and the plane of the diagonal inputs

// BALANCED STRIPS.  The strips of a query all have the same height, 64 lanes x R rows with the smallest R of {20, 24, 28, 32}
// that covers the query: 3 000 events are 2 strips of 64 x 24 rows (1 536 + 1 464) instead of 2 048 + 952 rows at R = 32 -- the
// sweep of a strip costs its R, whatever the number of lanes that hold rows, so a quarter of the time goes with the quarter of
// padding.  (At least 80 % of the rows of the strips of any query are query rows.)  R is a function of the query length alone:
// both passes, the boundary rows and the checkpoints of a read use the same one.
__host__ __device__ inline int strip_rows_per_lane(const int qlen) {
    const int n_strips = (qlen + kStripRows - 1) / kStripRows;
    const int per_lane = (qlen + 64 * n_strips - 1) / (64 * n_strips);
    const int R = per_lane <= 20 ? 20 : (per_lane <= 24 ? 24 : (per_lane <= 28 ? 28 : 32));
    return 64 * R * (n_strips - 1) < qlen ? R : kStripR;  // (always: the last strip holds at least one row)
}
constexpr int kBndPad = 256;                // words behind every boundary row (block over-run of the sweep + prefetch)

struct StripArgs {
    const float *queries;     // HBM: concatenated z-normalised event means
    const int64_t *q_off;     // [n_reads+1]
    const int32_t *reads;     // [n_long] batch index of every long read
    const float *ref;         // padded reference event arrays (as DpArgs::ref)
    const int64_t *job_off;   // [n_jobs]
    const int32_t *job_len;   // [n_jobs]
    const int64_t *bnd_off;   // [n_jobs+1] word offset of job j's boundary row inside one buffer (multiples of 4)
    float *bnd_cost;          // [n_long][max_strips - 1 rows][bnd_off[n_jobs]] last row of every strip but the final one (pass 1 writes, pass 2 reads)
    float *p_best, *p_second; // pass 1: partial top-2 per (long read, job); p_end = column of the best window's first strict minimum
    int32_t *p_end;
    const int32_t *w_job;     // pass 2: winner per long read (from the strip finalize): job, column of the winning cell, score
    const int32_t *w_ws;
    const float *w_score;
    int32_t *t_st, *t_end;    // pass 2 out: start / end column of the winning alignment per long read
    float *ck;                // checkpoints [n_long][ck_off[n_jobs]][33 planes][64 lanes]
    const int64_t *ck_off;    // [n_jobs+1] prefix sum over jobs of max_strips * nck(job), nck(job) = (rlen - 1) >> ck_shift
    int32_t ck_shift;         // checkpoint interval T = 1 << ck_shift steps (a multiple of the 4-step block)
    int32_t max_strips;       // strips of the longest query of the launch
    int32_t trace_margin;     // pass 2 resumes at least this many columns before the winning window; < 0: query length + 64
    int32_t n_long, n_jobs, rev_query;
    // pass 1 (sdtw_strip_pipe_kernel): one wave per (read, job, STRIP); strip s + 1 follows strip s through HBM
    int64_t bnd_stride;         // floats between two reads' boundary buffers (max_strips - 1 rows)
    const int32_t *strip_off;   // [n_long+1] prefix sum of the reads' strip counts
    int32_t *progress;          // [n_long][n_jobs][max_strips] columns of the strip's last row that are complete and visible
    unsigned *ticket;           // task counter (zeroed before the launch)
    // bounded waits (sdtw_kernels.hpp, bounded_wait_ge): error words, limit in 100 MHz ticks; debug_drop_strip >= 0: strip 0 of
    // that (long read of the group, job 0) publishes no progress (for the test of the bound)
    unsigned *err;
    long long spin_limit;
    int32_t debug_drop_strip;
#ifdef SFA_TASK_TIMES
    unsigned long long *task_times;  // measurement builds (tools/strip_task_times.py): [task][3] start, end (100 MHz ticks), SIMD position | strip << 32
#endif
};

#ifndef SFA_PIPE_BLOCK
#define SFA_PIPE_BLOCK 256
#endif
constexpr int kPipeBlock = SFA_PIPE_BLOCK;  // columns between two hand-overs of a boundary row (one release / acquire pair each)

// One anti-diagonal step of a strip: dp_step<R, TRACK> with the handling of query row 0 made conditional on FIRST (the
// strip that contains it).
template <bool STD, bool FIRST, bool TRACK, int R, typename XT>
__device__ __forceinline__ void strip_step(typename Vec<float, R>::type &c, typename Vec<int, R>::type &s, float &dprev,
                                           int &sdprev, const XT &x, const float yv, const int t, const bool lane0,
                                           Exchange &xc, const float bup = 0.0f, const int bsup = 0) {
    // bup / bsup (wave-uniform; strips below the first): the cell above lane 0's first row, from the boundary row in HBM
    float up = xc.shift(static_cast<float>(c[R - 1]));
    int sup = 0;
    if (TRACK) sup = xc.shift(static_cast<int>(s[R - 1]));
    if (!FIRST) {
        up = lane0 ? bup : up;
        if (TRACK) sup = lane0 ? bsup : sup;
    }
    if (STD && FIRST) {
        if (t == 0) xc.template set_boundary<TRACK>(lane0, INFINITY);  // std_dtw(): row 0 continues from its left neighbour only
    }
    float diag = dprev;
    int sdiag = sdprev;
    dprev = up;
    sdprev = sup;
    if (STD && FIRST) dprev = (t == 0) ? INFINITY : dprev;  // there is no column -1 next to the free corner
#pragma unroll
    for (int r = 0; r < R; ++r) {
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
        int sn = 0;
        if (TRACK) {
            sn = (diag == m) ? sdiag : ((left == m) ? sleft : sup);  // src/cdtw.c:134-146
            if (FIRST && r == 0) sn = lane0 ? t : sn;                // query row 0: the path starts in this column
        }
        diag = left;
        sdiag = sleft;
        up = cn;
        sup = sn;
        c[r] = cn;
        s[r] = sn;
    }
}

// Four columns of a boundary row.  The row is handed to another wave of the same launch, maybe on another XCD with its own L2:
// the store writes through to memory (sc1), so that the producer only has to wait for its own stores (publish_fence: an explicit
// s_waitcnt vmcnt(0)) instead of writing back the whole L2 (a release fence at agent scope is buffer_wbl2 -- with 17 GB of
// checkpoints passing through the same L2, once per kPipeBlock columns and wave).
__device__ __forceinline__ void store_row4(float *p, const typename Vec<float, 4>::type v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void publish_fence() { drain_stores(); }  // this wave's write-through stores have completed (sdtw_kernels.hpp)

// What the final strip reports.  Pass 1: the running top-2 over windows (by window).  Pass 2: the winning cell.
struct StripResult {
    Top2 top;          // pass 1
    int cap_end, cap_st;  // pass 2
};

// One strip over columns [0, ncols) of one (contig,strand).  LAST: the strip holds the last query row (lane lq,
// register rq) -- in pass 2 (TRACK) always: the row of the cell to reach.  Pass 2: column ws is the cell to reach, `best` its cost.
template <bool STD, bool FIRST, bool TRACK, bool LAST, int R, typename XT>
__device__ __forceinline__ void strip_sweep(const float *yp, const int rlen, const int ncols, const int qlen,
                                            const XT &x, const int lq, const int rq, const int lane, Exchange &xc,
                                            const float *bin_c, float *bout_c, StripResult &res,
                                            const int job, const int ws, const float best, const int t_begin, float *ckp,
                                            const int ck_shift, const int nck, const int32_t *prog_in = nullptr, int32_t *prog_out = nullptr,
                                            unsigned *err = nullptr, const long long spin_limit = 0) {
    static_assert(!TRACK || LAST, "pass 2 sweeps a strip up to a cell of its last row");
    constexpr bool PIPE = !TRACK;  // pass 1 is pipelined: strips of one (read, job) follow each other through HBM
    // PIPE, prog_in / prog_out: how far the strip above has got with the row this sweep reads / where to say
    // how far this sweep has got with the row it writes.  Both wave-uniform.  Pass 2: the rows are complete / nobody is waiting.
    // ckp: this strip's checkpoint records (+ lane).  Pass 1 stores record k - 1 before step k*T; pass 2 resumes from the
    // record of step t_begin (t_begin = 0: from the initial state).
    typename Vec<float, R>::type c;
    typename Vec<int, R>::type s;
    float dprev = INFINITY;
    int sdprev = 0;
    if (TRACK && t_begin > 0) {  // exact costs; where these cells came from is unknown (-1)
        const float *rec = ckp + static_cast<int64_t>((t_begin >> ck_shift) - 1) * kCkRec;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c[r] = rec[r * 64];
            s[r] = -1;
        }
        dprev = rec[kCkDprev];
        sdprev = (!FIRST && lane == 0) ? t_begin - 1 : -1;  // lane 0's diagonal input is the row above, column t_begin - 1
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c[r] = INFINITY;
            s[r] = 0;
        }
    }
    const bool lane0 = lane == 0;
    // std_dtw(): a sweep that resumes behind column 0 has the +inf boundary of its first row already
    if (FIRST) xc.template set_boundary<TRACK>(lane0, (STD && t_begin > 0) ? INFINITY : 0.0f);
    const int T = 1 << ck_shift;
    const int ck_last = nck << ck_shift;

    // pass 1, final strip: window scan of the last row (src/sigfish.c:891-901): the first strict minimum of every window AND
    // its column (two more operations per step, against the 100 a strip's step has anyway -- the wave kernels cannot afford
    // them); std_dtw has the single candidate C[n-1][m-1].  Knowing the winning CELL, pass 2 stops there and starts a query
    // length in front of it instead of in front of its window: half the steps.
    float wmin = INFINITY;
    int wpos = -1;
    int wend = min(qlen, rlen);

    // lane lq meets column ncols - 1 at step ncols - 1 + lq; pass 1 stores a boundary row four columns at a time, the last
    // group is complete up to three steps later (columns past ncols - 1 land in the pad behind the row)
    const int n_steps = ncols + lq + ((!TRACK && !LAST) ? 4 : 0);
    typename Vec<float, 4>::type ob = {0.0f, 0.0f, 0.0f, 0.0f};
    // hand-over of the boundary row in blocks of kPipeBlock columns (a multiple of the 64-column chunk).  The consumer loads a
    // chunk one chunk ahead -- inside steps [t0, t0 + kPipeBlock) it touches the columns below t0 + kPipeBlock + 64 --, the
    // producer's lane 63 has stored the columns below t0 - 64 at the top of its block t0.
    int wait_next = t_begin, pub_next = t_begin + kPipeBlock;
    auto wait_for_row = [&](int t0) {
        const int need = min(ncols, t0 + kPipeBlock + 72);
        int seen;
        wait_next = t0 + kPipeBlock;
        const bool arrived = bounded_wait_ge(prog_in, need, spin_limit, &seen, false);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!arrived) {
            // the row above never got there: say so and walk on without waiting any more (the batch is reported as failed; the
            // strips below still see this one's progress and end as well)
            if (lane == 0) report_device_error(err, kErrStripWait, need, seen);
            wait_next = 0x7fffffff;
        }
    };
    if (PIPE && !FIRST) wait_for_row(t_begin);
    float4u ycur = *reinterpret_cast<const float4u *>(yp + t_begin);
    // The row above, 64 columns at a time: lane l holds column (chunk base) + l, the next chunk is in flight while this one is
    // used (64 steps: a boundary row another wave has just written through comes from HBM, not from this XCD's L2), and
    // every step picks its column with v_readlane.
    float bcur = 0.0f, bnxt = 0.0f;
    if (!FIRST) bnxt = bin_c[t_begin + lane];
    for (int t0 = t_begin; t0 < n_steps; t0 += 4) {
        const int cpos = (t0 - t_begin) & 63;  // wave-uniform: position of step t0 inside the 64-column chunk
        // everything that is not a step happens at a chunk start or (checkpoint intervals below 64) at a multiple of T: ONE test
        // per block in the steady state
        const bool housekeeping = cpos == 0 || (T < 64 && (t0 & (T - 1)) == 0);
        if (!TRACK && __builtin_expect(housekeeping, 0)) {
            if (PIPE && !FIRST && t0 >= wait_next) wait_for_row(t0);
            if (PIPE && !LAST && t0 >= pub_next) {  // the stores of the columns below t0 - 64 have reached memory, then the counter says so
                publish_fence();
                if (lane == 0) __hip_atomic_store(prog_out, t0 - 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pub_next = t0 + kPipeBlock;
            }
            if (t0 > 0 && (t0 & (T - 1)) == 0 && t0 <= ck_last) {  // wave-uniform: snapshot of the state before step t0
                float *rec = ckp + static_cast<int64_t>((t0 >> ck_shift) - 1) * kCkRec;
#pragma unroll
                for (int r = 0; r < R; ++r) rec[r * 64] = c[r];
                rec[kCkDprev] = dprev;
            }
        }
        const float4u ynext = *reinterpret_cast<const float4u *>(yp + t0 + 4);
        if (!FIRST && cpos == 0) {
            // Past column ncols - 1 (up to kBndPad words) this reads words nobody wrote for this read and job: left-overs
            // of an earlier sweep, or the fill pattern of the allocation.  They only ever reach cells of columns >= ncols,
            // and every cell depends on cells of its own or a LOWER column alone (up, diagonal, left) -- nothing of a
            // column < ncols, the only ones that are read out, can see them.  (align_long() fills fresh allocations with
            // a large finite pattern so that tools inspecting the buffers see no NaNs; correctness does not rest on it.)
            bcur = bnxt;
            bnxt = bin_c[t0 + 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            float bup = 0.0f;
            int bsup = 0;
            if (!FIRST) {  // the row above query row 0 of this strip: column t of the previous strip's last row
                bup = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bcur), cpos + u));
                // what is carried is the COLUMN of the row above through which the path enters this strip
                if (TRACK) bsup = t;
            }
            strip_step<STD, FIRST, TRACK, R>(c, s, dprev, sdprev, x, ycur.v[u], t, lane0, xc, bup, bsup);
            // (everything below is branch-free but for the end of a window: a taken branch per step costs this loop a fifth of its
            // time -- the wave kernels' sweep has none either)
            if (!LAST) {
                // lane 63 is at column t - 63 of the strip's last row; t0 is a multiple of 4, so the column is u + 1 (mod 4):
                // four columns are collected and stored as one aligned 16-byte word when the fourth arrives
                ob[(u + 1) & 3] = c[R - 1];
                if (u == 2 && lane == 63 && t0 >= 64) store_row4(bout_c + (t0 - 64), ob);
            } else {
                const int col = t - lq;                        // wave-uniform
                const bool inside = col >= 0 && col < ncols;   // wave-uniform
                const float cl = c[rq];
                if (TRACK) {  // first cell of the winning window that attains the winning score
                    const bool hit = inside && res.cap_end < 0 && col >= ws && cl == best;
                    res.cap_end = hit ? col : res.cap_end;
                    res.cap_st = hit ? static_cast<int>(s[rq]) : res.cap_st;
                } else if (!STD) {
                    const float cv = inside ? cl : INFINITY;
                    const bool lt = cv < wmin;
                    wmin = lt ? cv : wmin;
                    wpos = lt ? col : wpos;
                    if (__builtin_expect(col + 1 == wend, 0)) {  // (wend <= ncols: never outside)
                        res.top.offer(wmin, wpos, job);
                        wmin = INFINITY;
                        wpos = -1;
                        wend = min(wend + qlen, rlen);
                    }
                } else if (__builtin_expect(col == rlen - 1, 0)) {
                    res.top.offer(cl, col, job);
                }
            }
        }
        ycur = ynext;
    }
    if (PIPE && !LAST) {  // the whole row is there
        publish_fence();
        if (lane == 0) __hip_atomic_store(prog_out, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// What every strip kernel knows about its read.
struct StripRead {
    const float *q;  // the query
    int qlen, n_strips;
};
__device__ __forceinline__ StripRead strip_read(const StripArgs &a, const int li) {
    const int read = a.reads[li];
    const int64_t qo = a.q_off[read];
    StripRead r;
    r.qlen = static_cast<int>(a.q_off[read + 1] - qo);
    r.q = a.queries + qo;
    r.n_strips = (r.qlen + kStripRows - 1) / kStripRows;
    return r;
}
// the R query rows of this lane in strip `sidx` (64 R rows per strip); rows past the query are zeros
template <int R>
__device__ __forceinline__ void strip_rows(const StripArgs &a, const StripRead &rd, const int sidx, const int lane, float (&x)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = sidx * 64 * R + lane * R + r;
        const int src = a.rev_query ? (rd.qlen - 1 - i) : i;
        x[r] = (i < rd.qlen) ? rd.q[src] : 0.0f;
    }
}
// (Pass 2 keeps its rows in registers: 199 VGPRs, two waves per SIMD.  With the rows in LDS -- 168 VGPRs, three waves -- it was 2.3x
// slower, 107.6 against 99.2 ms per step at 8 000 events: profiles/r04_logs/rejected_strip_pass2_rows_in_lds_three_waves.log.)
// the kernels pick the body for the read's rows per lane (wave-uniform)
#define SFA_STRIP_DISPATCH(BODY, qlen, ...)                    \
    switch (strip_rows_per_lane(qlen)) {                      \
        case 20: BODY<STD, 20>(__VA_ARGS__); break;           \
        case 24: BODY<STD, 24>(__VA_ARGS__); break;           \
        case 28: BODY<STD, 28>(__VA_ARGS__); break;           \
        default: BODY<STD, 32>(__VA_ARGS__); break;           \
    }

// Pass 1: wave-task = (job, long read, STRIP).  One wave walking all the strips of a (read, job) one after the other gives few,
// long tasks (6 250 for 3 125 reads of 8 000 events, each four sweeps long: 0.69 x 10^13 cells/s against 1.9 x 10^13 for the
// wave kernels).  Here every strip is its own wave, and strip s + 1 FOLLOWS strip s over the same columns, a block of kPipeBlock
// columns behind, reading the boundary row strip s writes (one row per strip boundary).  Waves claim tickets in the order job,
// read, strip: a strip's predecessor always holds a lower ticket, i.e. is running or done whenever the strip waits for it -- no
// deadlock whatever order the hardware starts blocks in.  Every strip drops its checkpoints, the final strip scans the windows
// and leaves the partial top-2 of its (read, job).
template <bool STD, int R>
__device__ __forceinline__ void strip_pipe_task(const StripArgs &a, const StripRead &rd, const int li, const int job, const int sidx, const int lane,
                                                Exchange &xc) {
    const int qlen = rd.qlen;
    const int rlen = a.job_len[job];
    const float *yp = a.ref + a.job_off[job] - lane;
    const int64_t per = a.bnd_off[a.n_jobs];
    float *rows = a.bnd_cost + static_cast<int64_t>(li) * a.bnd_stride + a.bnd_off[job];  // row s at rows + s * per
    int32_t *prog = a.progress + (static_cast<int64_t>(li) * a.n_jobs + job) * a.max_strips;
    const int nck = (rlen - 1) >> a.ck_shift;
    float *ckp = a.ck + (static_cast<int64_t>(li) * a.ck_off[a.n_jobs] + a.ck_off[job]) * kCkRec + lane + static_cast<int64_t>(sidx) * nck * kCkRec;
    const int nrows = min(64 * R, qlen - sidx * 64 * R);
    const bool last = sidx == rd.n_strips - 1;
    const int lq = (nrows - 1) / R;
    const int rq = (nrows - 1) - lq * R;
    float x[R];
    strip_rows<R>(a, rd, sidx, lane, x);
    StripResult res;
    res.top.init();
    res.cap_end = -1;
    res.cap_st = -1;
    float *bout_c = last ? nullptr : rows + static_cast<int64_t>(sidx) * per;
    const float *bin_c = sidx > 0 ? rows + static_cast<int64_t>(sidx - 1) * per : nullptr;
    if (sidx == 0) {  // (queries of this path have more than one strip: the first is never the last)
        // (test hook: the dropped strip publishes into a word nobody reads)
        int32_t *pub = (li == a.debug_drop_strip && job == 0) ? reinterpret_cast<int32_t *>(a.ticket + 8) : prog + sidx;
        strip_sweep<STD, true, false, false, R>(yp, rlen, rlen, qlen, x, lq, rq, lane, xc, bin_c, bout_c, res, job, 0, 0.0f, 0, ckp, a.ck_shift, nck,
                                                nullptr, pub);
    } else if (!last)
        strip_sweep<STD, false, false, false, R>(yp, rlen, rlen, qlen, x, lq, rq, lane, xc, bin_c, bout_c, res, job, 0, 0.0f, 0, ckp, a.ck_shift, nck,
                                                 prog + sidx - 1, prog + sidx, a.err, a.spin_limit);
    else
        strip_sweep<STD, false, false, true, R>(yp, rlen, rlen, qlen, x, lq, rq, lane, xc, bin_c, nullptr, res, job, 0, 0.0f, 0, ckp, a.ck_shift, nck,
                                                prog + sidx - 1, nullptr, a.err, a.spin_limit);
    if (last && lane == lq) {
        const int64_t o = static_cast<int64_t>(li) * a.n_jobs + job;
        a.p_best[o] = res.top.best;
        a.p_second[o] = res.top.second;
        a.p_end[o] = res.top.end;
    }
}

template <bool STD>
__global__ void __launch_bounds__(256, 4) sdtw_strip_pipe_kernel(const StripArgs a) {
    const int lane = threadIdx.x & 63;
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int task = __builtin_amdgcn_readfirstlane(t);
    const int total = a.strip_off[a.n_long];  // strips of all long reads of the group
    if (task >= total * a.n_jobs) return;
    const int job = task / total;
    const int r = task - job * total;
    int lo = 0, hi = a.n_long;  // the read whose strips hold index r: strip_off[li] <= r < strip_off[li + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.strip_off[mid] <= r)
            lo = mid;
        else
            hi = mid;
    }
    const int li = lo, sidx = r - a.strip_off[li];
    __shared__ float lds_f[4 * kXchWordsPerWave];
    __shared__ int lds_i[1];
    Exchange xc;
    xc.init(lds_f, lds_i, threadIdx.x >> 6, 0, lane, 64);
    const StripRead rd = strip_read(a, li);
#ifdef SFA_TASK_TIMES
    if (a.task_times && lane == 0) a.task_times[3 * static_cast<int64_t>(task)] = wall_clock64();
#endif
    SFA_STRIP_DISPATCH(strip_pipe_task, rd.qlen, a, rd, li, job, sidx, lane, xc)
#ifdef SFA_TASK_TIMES
    if (a.task_times && lane == 0) {
        a.task_times[3 * static_cast<int64_t>(task) + 1] = wall_clock64();
        a.task_times[3 * static_cast<int64_t>(task) + 2] = simd_position() | (static_cast<unsigned long long>(sidx) << 32);
    }
#endif
}

// Pass 2 (one wave per long read, its winning job): the strips are traced from the LAST one upwards, each over its own
// short range of columns.  Pass 1 kept the last row of every strip (one row per strip boundary), so a strip can be swept on
// its own: its `row above` is the row pass 1 stored.  What a strip carries is not the start of the whole path but the column b
// of the row above through which the path ENTERS the strip: a cell of the strip's first row that continues diagonally from
// column j - 1 or upwards from column j of the row above takes b = j - 1 or j (lane 0's `up` input carries its own column;
// its diagonal input is the previous step's); every other cell inherits b by the traceback rule of path()
// (src/cdtw.c:134-146), exactly as the start column is inherited in pass 2 of the wave kernels.  The rule is local, so the cells
// (last row of strip s - 1, b_s) are the cells of the reference's path, and strip 0 -- which holds query row 0 -- delivers the
// start column.  The cell a strip has to reach is known by column AND cost (the winning score for the last strip, the stored
// row's value for the others), compared bit for bit: a difference between the passes cannot go unnoticed.
// A strip of r rows is swept from the checkpoint r + 64 columns (or `trace_margin`) in front of its target cell and backs off
// 1, 2, 4 ... checkpoints while the path enters in front of the restored state: about (2048 + 64 + T/2) columns per strip.
template <bool STD, int R>
__device__ __forceinline__ void strip_chain_task(const StripArgs &a, const StripRead &rd, const int li, const int job, const int lane, Exchange &xc) {
    const int qlen = rd.qlen, n_strips = rd.n_strips;
    const int rlen = a.job_len[job];
    const float *yp = a.ref + a.job_off[job] - lane;
    const int64_t per = a.bnd_off[a.n_jobs];
    const float *rows = a.bnd_cost + static_cast<int64_t>(li) * a.bnd_stride + a.bnd_off[job];  // row s at rows + s * per
    const int nck = (rlen - 1) >> a.ck_shift;
    const int T = 1 << a.ck_shift;
    float *ck_job = a.ck + (static_cast<int64_t>(li) * a.ck_off[a.n_jobs] + a.ck_off[job]) * kCkRec + lane;

    int e = min(a.w_ws[li], rlen - 1);  // column of the cell to reach: the winning cell, then the entry columns
    float want = a.w_score[li];         // ... and its cost
    int t_end = -1, t_st = -1;
    int seen_rows = 0, seen_cols = 0;  // of the strips traced so far: rows and the columns the path took through them
    StripResult res;
    res.top.init();
    for (int sidx = n_strips - 1; sidx >= 0; --sidx) {
        const int nrows = min(64 * R, qlen - sidx * 64 * R);
        const int lq = (nrows - 1) / R;
        const int rq = (nrows - 1) - lq * R;
        float x[R];
        strip_rows<R>(a, rd, sidx, lane, x);
        const float *bin_c = sidx > 0 ? rows + static_cast<int64_t>(sidx - 1) * per : nullptr;
        float *ckp = ck_job + static_cast<int64_t>(sidx) * nck * kCkRec;
        // head start: as many columns as the strip has rows (+ 64) -- or, once the strips below have shown how many columns this
        // read's path takes per row (event detection over-segments: 1.5 events per reference position are typical, i.e. 2/3 of a
        // column per row), that many for this strip's rows + 25 % + 64.  Too short a head start only costs the back-off.
        int margin = nrows + 64;
        if (seen_rows >= 256) margin = min(margin, static_cast<int>((static_cast<int64_t>(seen_cols) * nrows * 5) / (static_cast<int64_t>(seen_rows) * 4)) + 64);
        const int from = e - (a.trace_margin >= 0 ? a.trace_margin : margin);
        int k = from > 0 ? min(from >> a.ck_shift, nck) : 0, back = 1, b = -1, hit = -1;
        for (int attempt = 0; attempt < 40; ++attempt) {  // until the entry is known (k reaches 0 after <= 32 halvings)
            res.cap_end = -1;
            res.cap_st = -1;
            if (sidx == 0)
                strip_sweep<STD, true, true, true, R>(yp, rlen, e + 1, qlen, x, lq, rq, lane, xc, nullptr, nullptr, res, job, e, want, k * T, ckp,
                                                      a.ck_shift, nck);
            else
                strip_sweep<STD, false, true, true, R>(yp, rlen, e + 1, qlen, x, lq, rq, lane, xc, bin_c, nullptr, res, job, e, want, k * T, ckp,
                                                       a.ck_shift, nck);
            b = __builtin_amdgcn_readlane(res.cap_st, lq);  // lq: wave-uniform
            hit = __builtin_amdgcn_readlane(res.cap_end, lq);
            if (b >= 0 || k == 0 || hit < 0) break;
            k = max(0, k - back);  // the path enters before this checkpoint
            back <<= 1;
        }
        if (sidx == n_strips - 1) t_end = hit;
        if (hit < 0 || b < 0) {  // the cell was not met with its cost (never seen; reported as an unaligned read rather than a wrong one)
            t_end = -1;
            break;
        }
        if (sidx == 0) {
            t_st = b;
        } else {
            seen_rows += nrows;
            seen_cols += e - b + 1;
            e = b;
            want = bin_c[b];
        }
    }
    if (lane == 0) {
        a.t_st[li] = t_st;
        a.t_end[li] = t_end;
    }
}

template <bool STD>
__global__ void __launch_bounds__(256, 1) sdtw_strip_chain_kernel(const StripArgs a) {
    const int li = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (li >= a.n_long) return;  // wave-uniform
    const int job = a.w_job[li];
    if (job < 0) return;  // nothing aligned
    const int lane = threadIdx.x & 63;
    __shared__ float lds_f[4 * kXchWordsPerWave];
    __shared__ int lds_i[4 * kXchWordsPerWave];
    Exchange xc;
    xc.init(lds_f, lds_i, threadIdx.x >> 6, 0, lane, 64);
    const StripRead rd = strip_read(a, li);
    SFA_STRIP_DISPATCH(strip_chain_task, rd.qlen, a, rd, li, job, lane, xc)
}

// instantiated in sdtw_inst_strips_pipe.hip / sdtw_inst_strips_chain.hip
extern template __global__ void sdtw_strip_chain_kernel<false>(const StripArgs);
extern template __global__ void sdtw_strip_chain_kernel<true>(const StripArgs);
extern template __global__ void sdtw_strip_pipe_kernel<false>(const StripArgs);
extern template __global__ void sdtw_strip_pipe_kernel<true>(const StripArgs);

struct StripFinalizeArgs {
    const int32_t *reads;  // [n_long]
    const float *p_best, *p_second;
    const int32_t *p_end;
    const int32_t *job_contig;
    const int8_t *job_strand;
    const int32_t *ref_len, *ref_st_offset;
    int32_t *w_job, *w_ws;  // mode 1 out: winners for pass 2
    float *w_score;
    const int32_t *t_st, *t_end;  // mode 2 in
    ResultRow *out;  // rows of the whole batch
    const uint8_t *bad;  // [reads of the batch] non-finite query (sdtw_screen_kernel): the read is skipped
    int32_t n_long, n_jobs;
    int32_t mode;  // 1: after pass 1 -> scores, contig, strand, mapq + winners; 2: after pass 2 -> positions
};

#ifdef SFA_DEFINE_FINALIZE_KERNEL
// merge the per-job top-2 of a long read in processing order (a later job wins ties, src/sigfish.c:577-583), then strand
// flip, offset and mapq (src/sigfish.c:969-983) -- sdtw_finalize_kernel for the strip path
__global__ void __launch_bounds__(64) sdtw_strip_finalize_kernel(const StripFinalizeArgs a) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= a.n_long) return;
    if (a.mode == 2) {
        ResultRow r = a.out[a.reads[li]];
        const int st = a.t_st[li], end = a.t_end[li];
        if (r.rid < 0 || end < 0) return;
        const int rl = a.ref_len[r.rid], off = a.ref_st_offset[r.rid];
        r.pos_st = ((r.strand == '+') ? st : rl - end) + off;  // src/sigfish.c:971-975
        r.pos_end = ((r.strand == '+') ? end : rl - st) + off;
        a.out[a.reads[li]] = r;
        return;
    }
    float best = INFINITY, second = INFINITY;
    int ws = -1, job = -1;
    for (int j = 0; j < a.n_jobs; ++j) {
        const int64_t o = static_cast<int64_t>(li) * a.n_jobs + j;
        const float b = a.p_best[o], s2 = a.p_second[o];
        const float hi = fmaxf(best, b);
        const float lo2 = fminf(second, s2);
        const bool take = !(b > best);
        second = fminf(hi, lo2);
        if (take) {
            best = b;
            ws = a.p_end[o];
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
    if (a.bad[a.reads[li]]) {  // the reference aborts on such a read (see sdtw_screen_kernel): skipped, like a read without events
        r.valid = 0;
        r.score = r.score2 = INFINITY;
        job = -1;
    }
    if (job >= 0 && ws >= 0) {
        r.rid = a.job_contig[job];
        r.strand = a.job_strand[job];
        r.mapq = mapq_from_scores(best, second);
    } else {
        job = -1;
    }
    a.w_job[li] = job;
    a.w_ws[li] = ws;
    a.w_score[li] = best;
    a.out[a.reads[li]] = r;
}
#endif

}  // namespace sfa
