// sfa_capi.hip -- C-ABI (include/sigfish_amd.h) over the gfx950 kernels in sdtw_kernels.hpp.
//
// Replaces the reference's accelerator hook align_db() (src/sigfish.c:1003-1015) and its init / teardown slots
// (src/sigfish.c:200-204, 221-225).  Host work done here: pack the reference event arrays into one padded HBM
// buffer, group reads into "quads" (four reads of one length class share a wavefront; sfa_plan.hpp), pick the
// rows-per-lane class, size the checkpoint interval, launch fill -> finalize -> trace -> finalize, and hand back
// one row per read in input order.
// There is NO CPU fallback: every failure is reported through the return code + sfa_last_error().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sigfish_amd.h"
#define SFA_DEFINE_FINALIZE_KERNEL
#include "sdtw_kernels.hpp"
#include "events_kernels.hpp"
#include "blow5_kernels.hpp"
#include "host/blow5.hpp"
#include "sdtw_instances.hpp"
#include "sdtw_strips.hpp"
#include "sfa_plan.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(SFA_ENODEV, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                  \
    } while (0)

// after a kernel launch: a rejected launch (bad configuration, missing code object) is SFA_EKERNEL
#define KERNEL_TRY()                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = hipGetLastError();                                                                      \
        if (e_ != hipSuccess) return fail(SFA_EKERNEL, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// A device buffer that only ever grows (batches reuse it; nothing is allocated inside a steady-state call).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SFA_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            return fail(SFA_ENOMEM, "hipMalloc(%zu bytes) failed", want);
        }
        cap = want;
        return SFA_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SFA_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            return fail(SFA_ENOMEM, "hipHostMalloc(%zu bytes) failed", want);
        }
        cap = want;
        return SFA_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

// One host thread per shard of a group context (sfa_init_devices), alive as long as the context: HIP's current device and
// sfa_last_error are per thread, so every shard's calls are made from its own thread -- but not from a fresh one per call
// (round 2: std::async per shard and call, i.e. a thread creation + the runtime's per-thread set-up inside every batch).
class ShardWorker {
  public:
    ShardWorker() : th_([this] { loop(); }) {}
    ~ShardWorker() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = std::move(job);
            state_ = 1;
        }
        cv_.notify_all();
    }
    std::pair<int, std::string> wait() {  // of the job posted last
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return state_ == 2; });
        state_ = 0;
        return {rc_, err_};
    }

  private:
    void loop() {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return quit_ || state_ == 1; });
                if (quit_) return;
                job = std::move(job_);
            }
            const int rc = job();
            {
                std::lock_guard<std::mutex> lk(m_);
                rc_ = rc;
                err_ = rc ? g_err : std::string();
                state_ = 2;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    int state_ = 0;  // 0 idle, 1 posted, 2 done
    bool quit_ = false;
    int rc_ = 0;
    std::string err_;
    std::thread th_;  // last: started when everything else exists
};

}  // namespace

struct sfa_ctx {
    // A GROUP context (sfa_init_devices) owns one ordinary context per listed device and nothing else: every batch is cut
    // into contiguous read ranges, one per shard, which run concurrently; rows land in the caller's array in input order.
    std::vector<sfa_ctx *> shards;
    std::vector<std::unique_ptr<ShardWorker>> workers;  // [shards - 1]: shard r > 0 is driven from workers[r - 1], shard 0 from the caller
    std::vector<int32_t> shard_lo;  // [shards+1] read ranges of the batch submitted with sfa_submit_batch

    int device = 0;
    uint32_t flag = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_long = nullptr;       // the row strips of long queries run beside the wave kernels of the same batch
    hipStream_t stream_pre = nullptr;        // SFA_RESERVED_CUS=N: record decoding + event detection on N CUs of their own (CU-masked stream), alignment on the rest
    hipEvent_t lev[2] = {nullptr, nullptr};  // inputs of the batch ready on `stream` / strips done on `stream_long`
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // fill start/end, finalize1 end, trace end, end, row strips start
    hipEvent_t eev[4] = {nullptr, nullptr, nullptr, nullptr};  // sfa_align_raw: event detection start/end, normalisation start/end
    bool eev_pending = false;
    int cu_count = 256;
    int pre_cus = 0;  // CUs of stream_pre (0: no partition, the pre-alignment stages see the whole device)

    // tunables (sfa_set_option)
    int64_t opt_single_pass = 0;             // 1: one fill with start tracking everywhere (first-round design)
    int64_t opt_ckpt_interval = 0;           // force the checkpoint interval (power of two >= 4); 0 = auto
    int64_t opt_ckpt_budget = 32ll << 30;    // bytes of HBM the checkpoints of one batch may take
    int64_t opt_ev_parallel_peaks = 1;       // sfa_align_raw: wave-per-read peak picker where its result is certified
    int64_t opt_ev_parallel_prefix = 1;      // sfa_align_raw: wave-per-read prefix sums where they are provably exact
    int64_t opt_min_slice_reads = 65536;     // a batch is only cut into slices of at least this many reads
    int64_t opt_segment_warm = 4;            // query lengths of warm-up in front of a segment
    int64_t opt_column_segments = 0;         // 0 = auto, 1 = off, N = segments per job for small batches (sweep_segment)
    int64_t opt_lane_widening = 0;           // 0 = by batch size; 1, 2, 4 = fixed (rows per lane / w, lanes per read * w)
    int64_t opt_widen_below = 5;             // auto: widen (x4) when the batch has fewer waves per SIMD than this
    int64_t opt_trace_margin = -1;           // steps of head start for pass 2; -1 = qlen_max + 16
    int64_t opt_waves_per_simd = 6;          // target occupancy used when chunking the job list
    int64_t opt_balanced_strips = 1;         // row strips of equal height, 64 x {20, 24, 28, 32} rows by query length (0: 64 x 32 and a short last one)
    int64_t opt_long_overlap = 1;            // row strips on their own stream, beside the wave kernels of the batch (0: behind them)
    int64_t opt_strip_chain = 1;             // row strips, pass 2: strip by strip from the last one upwards over short column ranges (1) or all strips over the whole range (0)
    int64_t opt_strip_pipeline = 1;          // row strips, pass 1: one wave per strip following the strip above (1) or one wave per (read, job) (0)
    int64_t opt_fused_trace = 1;             // 1: with LDS checkpoints, pass 2 rides in the fill launch as trailing tickets (fills the drain)
    int64_t opt_lds_ckpt = 1;                // 1: rolling checkpoints in LDS where the batch's shapes allow (R <= 16, sDTW); 0: all snapshots to HBM
    int64_t opt_prio_unit = 2048;            // longest-remaining-first issue priority of the fill: columns per level, 0 = off
    int64_t opt_adaptive_margin = 1;         // pass 2's head start follows the spans of the previous batch's alignments (HBM-snapshot route)
    int32_t span_sixteenths = 0;             // ... in sixteenths of the query length (0: not known yet -> a whole query length)
    int64_t opt_mixed_quads = 1;             // reads of different lengths (equal modulo the rows per lane) may share a wave
    int64_t opt_spin_limit_ms = 20000;       // bound of every in-launch wait (fused pass 2, pipelined strips); beyond it the batch fails with SFA_EKERNEL
    int64_t opt_debug_drop_quad = -1;        // test hook: the fill tasks of this quad never signal completion
    int64_t opt_debug_drop_strip = -1;       // test hook: strip 0 of this long read (job 0) never publishes its progress

    // reference model (immutable after init)
    int32_t num_ref = 0, n_jobs = 0;
    int64_t total_cols = 0;  // sum over jobs of rlen
    std::vector<int32_t> h_job_len;
    DevBuf d_ref, d_job_off, d_job_len, d_job_contig, d_job_strand, d_ref_len, d_ref_off;

    // per-batch scratch
    DevBuf d_verify, d_segfail;
    DevBuf d_lprog, d_lticket;  // pipelined strips: progress counters, ticket
    DevBuf d_bndc, d_bnds, d_long, d_lbest, d_lsecond, d_lend, d_lwin, d_lck;  // row strips (queries beyond SFA_MAX_QUERY, sdtw_strips.hpp)
    PinBuf h_long;
    bool long_pending = false;
    int64_t seg_reruns = 0;  // batches walked again because a segment hand-over did not verify
    DevBuf d_queries, d_stage, d_pbest, d_pend, d_pst, d_pjob, d_psecond, d_wjob, d_wend, d_wscore, d_tst, d_ck, d_out;
    PinBuf h_stage, h_out, h_small, h_flags;
    PinBuf h_queries;  // sfa_align_events: the gathered event means (page-locked: the upload from here is asynchronous)

    // raw-signal path (sfa_align_raw)
    DevBuf e_raw, e_rawoff, e_scale, e_sum, e_sumsq, e_t1, e_t2, e_evoff, e_evstart, e_evlen, e_evmean, e_evstdv, e_nev, e_qstart,
        e_qoff, e_b0, e_b1, e_b2, e_flag, e_qev, e_pflag;

    DevBuf b_in, b_inoff, b_out, b_outoff, b_len, b_head, b_bad;  // sfa_align_blow5: record bytes, inflated payloads, field rows
    PinBuf h_head;
    hipEvent_t bev[2] = {nullptr, nullptr};  // record decoding start / end
    bool bev_pending = false;
    int64_t blow5_fallbacks = 0;  // batches handed to the host reader because the device declined a record
    DevBuf d_args;  // fused launch: the kernel's argument block in device memory (DpArgs::self)
    DevBuf d_ticket, d_quaddone;  // fused launch: ticket counter, completed fill tasks per quad
    DevBuf d_bestrec, d_beste, d_gbest, d_wchunk;  // LDS-checkpoint fill: records of the best windows, their step, per-read best score, winning chunk
    DevBuf d_bad, d_badcount;  // sdtw_screen_kernel: per-read flag, number of flagged reads
    PinBuf h_badcount;
    DevBuf d_started;     // counter of the fill's tasks that have begun (IssuePriority)
    DevBuf d_times;       // -DSFA_TASK_TIMES builds: start / end / SIMD position of every wave-task of the last fill
    int64_t n_times = 0;
    DevBuf d_ltimes;      // ... and of the last pipelined pass 1 over row strips (tools/strip_task_times.py)
    int64_t n_ltimes = 0;
    sfa::BatchPlan plan;  // plan of the batch being submitted (scratch included)
    sfa_profile_t prof{};
    bool prof_pending = false;
    bool no_segments_once = false;  // re-run of a batch whose segment hand-overs did not verify
    bool in_slice = false;   // align_device is running one slice of a cut-up batch
    int32_t pending_n = -1;  // reads of the batch submitted with sfa_submit_batch and not yet collected
};

namespace {

using sfa::DpArgs;
using sfa::FinalizeArgs;
using sfa::ResultRow;

static_assert(sizeof(ResultRow) == sizeof(sfa_result_t), "result row layout");

template <bool TRACK>
void launch_fill(int maxr, bool std_dtw, const DpArgs &a, hipStream_t st) {
    const dim3 grid((a.n_tasks + 3) / 4), block(256);
    if (!TRACK && a.n_seg > 1) {  // column segments (cost-only sDTW, small batches): the SEG kernels
        if (maxr >= 32)
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<32, false, false, true>), grid, block, 0, st, a);
        else if (maxr >= 16)
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<16, false, false, true>), grid, block, 0, st, a);
        else if (maxr >= 8)
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<8, false, false, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<4, false, false, true>), grid, block, 0, st, a);
        return;
    }
#define SFA_FILL(MR)                                                                                   \
    if (std_dtw)                                                                                       \
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<MR, TRACK, true>), grid, block, 0, st, a);           \
    else                                                                                               \
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<MR, TRACK, false>), grid, block, 0, st, a)
    if (maxr >= 32) {
        SFA_FILL(32);
    } else if (maxr >= 16) {
        SFA_FILL(16);
    } else if (maxr >= 8) {
        SFA_FILL(8);
    } else {
        SFA_FILL(4);
    }
#undef SFA_FILL
}

void launch_trace(int maxr, bool std_dtw, const DpArgs &a, int32_t *out_st, hipStream_t st) {
    const dim3 grid((a.n_tasks + 3) / 4), block(256);
#define SFA_TRACE(MR)                                                                                  \
    if (std_dtw)                                                                                       \
        hipLaunchKernelGGL((sfa::sdtw_trace_kernel<MR, true>), grid, block, 0, st, a, out_st);         \
    else                                                                                               \
        hipLaunchKernelGGL((sfa::sdtw_trace_kernel<MR, false>), grid, block, 0, st, a, out_st)
    if (maxr >= 32) {
        SFA_TRACE(32);
    } else if (maxr >= 16) {
        SFA_TRACE(16);
    } else if (maxr >= 8) {
        SFA_TRACE(8);
    } else {
        SFA_TRACE(4);
    }
#undef SFA_TRACE
}

// the variants with rolling checkpoints in LDS (cost-only fill, R <= 16; std_dtw: the sparse HBM store alone)
#define SFA_LCK_LAUNCH(KERNEL, ...)                                                       \
    do {                                                                                  \
        if (maxr >= 16)                                                                   \
            hipLaunchKernelGGL((KERNEL(16)), grid, block, 0, st, __VA_ARGS__);            \
        else if (maxr >= 8)                                                               \
            hipLaunchKernelGGL((KERNEL(8)), grid, block, 0, st, __VA_ARGS__);             \
        else                                                                              \
            hipLaunchKernelGGL((KERNEL(4)), grid, block, 0, st, __VA_ARGS__);             \
    } while (0)
void launch_fill_lck(int maxr, bool std_dtw, const DpArgs &a, hipStream_t st) {
    const dim3 grid((a.n_tasks + 3) / 4), block(256);
#define K_(MR) sfa::sdtw_fill_kernel<MR, false, false, false, true>
#define KS_(MR) sfa::sdtw_fill_kernel<MR, false, true, false, true>
    if (std_dtw)
        SFA_LCK_LAUNCH(KS_, a);
    else
        SFA_LCK_LAUNCH(K_, a);
#undef K_
#undef KS_
}

void launch_fill_fused(int maxr, bool std_dtw, const DpArgs &a, hipStream_t st, int cu_count) {  // fill tasks + one pass-2 ticket per quad
    // waves claim tickets until they run out: no more blocks than the device holds at once (four per CU: the LDS buffers), so
    // that none of them starts only to find the counter exhausted
    unsigned blocks = static_cast<unsigned>((a.n_tasks + 3) / 4 + (a.n_quads_total + 3) / 4);
#if SFA_FUSED_PERSIST
    blocks = std::min<unsigned>(blocks, static_cast<unsigned>(cu_count) * SFA_LCK_WAVES);
#endif
    const dim3 grid(blocks), block(256);
#define K_(MR) sfa::sdtw_fill_kernel<MR, false, false, false, true, true>
#define KS_(MR) sfa::sdtw_fill_kernel<MR, false, true, false, true, true>
    if (std_dtw)
        SFA_LCK_LAUNCH(KS_, a);
    else
        SFA_LCK_LAUNCH(K_, a);
#undef K_
#undef KS_
}

void launch_trace_lck(int maxr, bool std_dtw, const DpArgs &a, int32_t *out_st, hipStream_t st) {
    const dim3 grid((a.n_tasks + 3) / 4), block(256);
#define K_(MR) sfa::sdtw_trace_kernel<MR, false, true>
#define KS_(MR) sfa::sdtw_trace_kernel<MR, true, true>
    if (std_dtw)
        SFA_LCK_LAUNCH(KS_, a, out_st);
    else
        SFA_LCK_LAUNCH(K_, a, out_st);
#undef K_
#undef KS_
}

int resolve_profile(sfa_ctx *c);
int align_device(sfa_ctx *c, const float *d_queries, const int64_t *q_off, int32_t n, ResultRow *d_out);

// Reads of more than SFA_MAX_QUERY events: row strips (sdtw_strips.hpp).  Pass 1, one wave per (read, job, strip) -- or per
// (read, job) with "strip_pipeline" 0 --, sweeps the query in strips of 64 x R rows, cost only, handing the last row of a strip
// to the next one through HBM; the strip finalize names each read's winning (job, cell, score); pass 2, one wave per read,
// traces the winning job strip by strip from the last one upwards (or, "strip_chain" 0, all strips again up to the winning
// cell) with start-column tracking.  Runs on `st`: the context's second stream, beside the wave kernels of the batch
// ("long_overlap"), or its main stream behind them; it writes the rows of these reads, which the wave kernels' finalize leaves
// alone.  Reads are taken in groups whose boundary rows and checkpoints fit the checkpoint budget.
int align_long(sfa_ctx *c, const float *d_queries, const int64_t *d_q_off, const int64_t *q_off_host, const std::vector<int32_t> &reads, int64_t max_qlen,
               ResultRow *d_out, hipStream_t st) {
    const int32_t n_long = static_cast<int32_t>(reads.size()), n_jobs = c->n_jobs;
    const size_t o_reads = 0, o_bnd = (sizeof(int32_t) * n_long + 7) & ~size_t(7);
    const size_t o_ck = o_bnd + sizeof(int64_t) * (n_jobs + 1);
    const size_t o_soff = o_ck + sizeof(int64_t) * (n_jobs + 1);  // per-group prefix sums of the reads' strip counts: group g0 at word g0 + (its index)
    const size_t stage_bytes = o_soff + sizeof(int32_t) * (2 * static_cast<size_t>(n_long) + 2);  // (at most n_long groups)
    std::vector<int32_t> n_strips_of(n_long);
    for (int32_t i = 0; i < n_long; ++i) n_strips_of[i] = static_cast<int32_t>((q_off_host[reads[i] + 1] - q_off_host[reads[i]] + sfa::kStripRows - 1) / sfa::kStripRows);
    int rc;
    if ((rc = c->h_long.reserve(stage_bytes)) || (rc = c->d_long.reserve(stage_bytes))) return rc;
    char *hs = c->h_long.as<char>();
    memcpy(hs + o_reads, reads.data(), sizeof(int32_t) * n_long);
    int64_t *bnd_off = reinterpret_cast<int64_t *>(hs + o_bnd);
    int64_t per = 0, row_max = 0;
    for (int32_t j = 0; j < n_jobs; ++j) {
        const int64_t row = (static_cast<int64_t>(c->h_job_len[j]) + sfa::kBndPad + 3) & ~int64_t(3);
        bnd_off[j] = per;
        per += row;
        row_max = std::max(row_max, row);
    }
    bnd_off[n_jobs] = per;
    // checkpoints: every strip of every job, every T steps, 33 planes of 64 lanes; T = 512 unless that takes more than the
    // budget for the whole set of long reads
    const int32_t max_strips = static_cast<int32_t>((max_qlen + sfa::kStripRows - 1) / sfa::kStripRows);
    int64_t *ck_off = reinterpret_cast<int64_t *>(hs + o_ck);
    const int64_t rec_floats = (sfa::kStripR + 1) * 64;
    int ck_shift = c->opt_ckpt_interval > 0 ? 2 : 9;
    if (c->opt_ckpt_interval > 0)
        while ((1ll << ck_shift) < c->opt_ckpt_interval) ++ck_shift;  // (the option's values are powers of two >= 4)
    for (;; ++ck_shift) {
        int64_t recs = 0;
        for (int32_t j = 0; j < n_jobs; ++j) {
            ck_off[j] = recs;
            recs += static_cast<int64_t>(max_strips) * ((c->h_job_len[j] - 1) >> ck_shift);
        }
        ck_off[n_jobs] = recs;
        if (c->opt_ckpt_interval > 0 || recs * rec_floats * 4 * n_long <= c->opt_ckpt_budget || ck_shift >= 14) break;
    }
    const int64_t ck_floats_per_read = ck_off[n_jobs] * rec_floats;
    // cost rows of every job: two in turn (classic pass 1, pass 2) or one per strip boundary (pipelined pass 1); start columns of
    // one job, two in turn (pass 2); + the checkpoints
    const bool pipe = c->opt_strip_pipeline != 0;
    const bool chain = c->opt_strip_chain != 0;  // pass 2 strip by strip from the last one upwards: needs every boundary row of pass 1
    const int64_t cost_rows = (pipe || chain) ? std::max<int64_t>(2, max_strips - 1) : 2;
    const int64_t bytes_per_read = (per * cost_rows + row_max * 2) * 4 + ck_floats_per_read * 4;
    const int32_t group = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(n_long, c->opt_ckpt_budget / std::max<int64_t>(bytes_per_read, 1))));
    const size_t n_part = static_cast<size_t>(n_long) * n_jobs;
    const size_t bndc_cap = c->d_bndc.cap;
    if (pipe && ((rc = c->d_lprog.reserve(sizeof(int32_t) * static_cast<size_t>(group) * n_jobs * max_strips)) || (rc = c->d_lticket.reserve(64)))) return rc;
    if ((rc = c->d_bndc.reserve(sizeof(float) * cost_rows * per * group)) || (rc = c->d_bnds.reserve(sizeof(int32_t) * 2 * row_max * group)) ||
        (rc = c->d_lbest.reserve(4 * n_part)) || (rc = c->d_lsecond.reserve(4 * n_part)) || (rc = c->d_lend.reserve(4 * n_part)) ||
        (rc = c->d_lwin.reserve(4 * 5 * static_cast<size_t>(n_long))) ||
        (rc = c->d_lck.reserve(sizeof(float) * std::max<int64_t>(ck_floats_per_read, 1) * group)))
        return rc;
    if (c->d_bndc.cap != bndc_cap) HIP_TRY(hipMemsetAsync(c->d_bndc.p, 0x7f, c->d_bndc.cap, st));  // fresh allocation: 3.4e38 everywhere (see the pad note in sdtw_strips.hpp)
    if (pipe) {  // the prefix sums of every group, each in its own words: one upload for all groups, no host wait between them
        int32_t *soff = reinterpret_cast<int32_t *>(hs + o_soff);
        for (int32_t g0 = 0, gi = 0; g0 < n_long; g0 += group, ++gi) {
            const int32_t gn = std::min(group, n_long - g0);
            int32_t *so = soff + g0 + gi;
            so[0] = 0;
            for (int32_t i = 0; i < gn; ++i) so[i + 1] = so[i] + n_strips_of[g0 + i];
        }
    }
    HIP_TRY(hipMemcpyAsync(c->d_long.p, hs, stage_bytes, hipMemcpyHostToDevice, st));
    const char *ds = c->d_long.as<char>();
    const bool std_dtw = (c->flag & SFA_DTW) != 0;
    int32_t *win = c->d_lwin.as<int32_t>();  // [5][n_long]: w_job, w_ws, w_score, t_st, t_end
    for (int32_t g0 = 0, gi = 0; g0 < n_long; g0 += group, ++gi) {
        const int32_t gn = std::min(group, n_long - g0);
        sfa::StripArgs sa{};
        sa.queries = d_queries;
        sa.q_off = d_q_off;
        sa.reads = reinterpret_cast<const int32_t *>(ds + o_reads) + g0;
        sa.ref = c->d_ref.as<float>();
        sa.job_off = c->d_job_off.as<int64_t>();
        sa.job_len = c->d_job_len.as<int32_t>();
        sa.bnd_off = reinterpret_cast<const int64_t *>(ds + o_bnd);
        sa.bnd_cost = c->d_bndc.as<float>();
        sa.bnd_start = c->d_bnds.as<int32_t>();
        sa.bnd_row_max = row_max;
        sa.bnd_stride = cost_rows * per;
        sa.keep_rows = chain ? 1 : 0;
        sa.balanced = c->opt_balanced_strips ? 1 : 0;
        sa.p_best = c->d_lbest.as<float>() + static_cast<size_t>(g0) * n_jobs;
        sa.p_second = c->d_lsecond.as<float>() + static_cast<size_t>(g0) * n_jobs;
        sa.p_end = c->d_lend.as<int32_t>() + static_cast<size_t>(g0) * n_jobs;
        sa.w_job = win + g0;
        sa.w_ws = win + n_long + g0;
        sa.w_score = reinterpret_cast<const float *>(win + 2 * static_cast<size_t>(n_long) + g0);
        sa.t_st = win + 3 * static_cast<size_t>(n_long) + g0;
        sa.t_end = win + 4 * static_cast<size_t>(n_long) + g0;
        sa.ck = c->d_lck.as<float>();
        sa.ck_off = reinterpret_cast<const int64_t *>(ds + o_ck);
        sa.ck_shift = ck_shift;
        sa.max_strips = max_strips;
        sa.trace_margin = static_cast<int32_t>(c->opt_trace_margin);
        sa.n_long = gn;
        sa.n_jobs = n_jobs;
        sa.rev_query = ((c->flag & SFA_RNA) && !(c->flag & SFA_INV)) ? 1 : 0;
        sa.err = c->d_badcount.as<unsigned>() + 4;
        sa.spin_limit = c->opt_spin_limit_ms * 100000;  // 100 MHz ticks
        sa.debug_drop_strip = (c->opt_debug_drop_strip >= g0 && c->opt_debug_drop_strip < g0 + gn) ? static_cast<int32_t>(c->opt_debug_drop_strip - g0) : -1;
        sfa::StripFinalizeArgs fa{};
        fa.reads = sa.reads;
        fa.p_best = sa.p_best;
        fa.p_second = sa.p_second;
        fa.p_end = sa.p_end;
        fa.job_contig = c->d_job_contig.as<int32_t>();
        fa.job_strand = c->d_job_strand.as<int8_t>();
        fa.ref_len = c->d_ref_len.as<int32_t>();
        fa.ref_st_offset = c->d_ref_off.as<int32_t>();
        fa.w_job = win + g0;
        fa.w_ws = win + n_long + g0;
        fa.w_score = reinterpret_cast<float *>(win + 2 * static_cast<size_t>(n_long) + g0);
        fa.t_st = sa.t_st;
        fa.t_end = sa.t_end;
        fa.out = d_out;
        fa.bad = c->d_bad.as<uint8_t>();
        fa.n_long = gn;
        fa.n_jobs = n_jobs;
        const dim3 block(256), fgrid((gn + 63) / 64), fblock(64);
        const dim3 grid1(static_cast<unsigned>((static_cast<int64_t>(gn) * n_jobs + 3) / 4)), grid2((gn + 3) / 4);
        if (pipe) {  // one wave per (job, read, strip), tickets in that order
            const int32_t *soff = reinterpret_cast<const int32_t *>(hs + o_soff) + g0 + gi;  // (uploaded with the staging area, before the loop)
            HIP_TRY(hipMemsetAsync(c->d_lprog.p, 0, sizeof(int32_t) * static_cast<size_t>(gn) * n_jobs * max_strips, st));
            HIP_TRY(hipMemsetAsync(c->d_lticket.p, 0, 4, st));
            sa.strip_off = reinterpret_cast<const int32_t *>(ds + o_soff) + g0 + gi;
            sa.progress = c->d_lprog.as<int32_t>();
            sa.ticket = c->d_lticket.as<unsigned>();
            const int64_t waves = static_cast<int64_t>(soff[gn]) * n_jobs;
            const dim3 gridp(static_cast<unsigned>((waves + 3) / 4));
#ifdef SFA_TASK_TIMES
            if ((rc = c->d_ltimes.reserve(24 * static_cast<size_t>(waves)))) return rc;
            sa.task_times = c->d_ltimes.as<unsigned long long>();
            c->n_ltimes = waves;
#endif
            if (std_dtw)
                hipLaunchKernelGGL((sfa::sdtw_strip_pipe_kernel<true>), gridp, block, 0, st, sa);
            else
                hipLaunchKernelGGL((sfa::sdtw_strip_pipe_kernel<false>), gridp, block, 0, st, sa);
        } else if (std_dtw) {
            hipLaunchKernelGGL((sfa::sdtw_strip_kernel<true, false>), grid1, block, 0, st, sa);
        } else {
            hipLaunchKernelGGL((sfa::sdtw_strip_kernel<false, false>), grid1, block, 0, st, sa);
        }
        KERNEL_TRY();
        fa.mode = 1;
        hipLaunchKernelGGL(sfa::sdtw_strip_finalize_kernel, fgrid, fblock, 0, st, fa);
        KERNEL_TRY();
        if (chain && std_dtw)
            hipLaunchKernelGGL((sfa::sdtw_strip_chain_kernel<true>), grid2, block, 0, st, sa);
        else if (chain)
            hipLaunchKernelGGL((sfa::sdtw_strip_chain_kernel<false>), grid2, block, 0, st, sa);
        else if (std_dtw)
            hipLaunchKernelGGL((sfa::sdtw_strip_kernel<true, true>), grid2, block, 0, st, sa);
        else
            hipLaunchKernelGGL((sfa::sdtw_strip_kernel<false, true>), grid2, block, 0, st, sa);
        KERNEL_TRY();
        fa.mode = 2;
        hipLaunchKernelGGL(sfa::sdtw_strip_finalize_kernel, fgrid, fblock, 0, st, fa);
        KERNEL_TRY();
        c->prof.fill_launches++;
    }
    return SFA_OK;
}

// A batch so large that its checkpoints only fit the budget at a long interval (a long pass 2) is cut into slices
// of contiguous reads that keep the interval short; slices of >= 64 Ki reads still fill the chip.  Slices run one
// after the other (each is planned and staged on its own), so such a call is synchronous.

int align_sliced(sfa_ctx *c, const float *d_queries, const int64_t *q_off, int32_t n, ResultRow *d_out, int32_t slices) {
    sfa_profile_t sum{};
    for (int32_t s = 0; s < slices; ++s) {
        const int32_t lo = static_cast<int32_t>(static_cast<int64_t>(n) * s / slices);
        const int32_t hi = static_cast<int32_t>(static_cast<int64_t>(n) * (s + 1) / slices);
        c->in_slice = true;
        int rc = align_device(c, d_queries, q_off + lo, hi - lo, d_out + lo);  // q_off holds absolute offsets into d_queries
        c->in_slice = false;
        if (rc) return rc;
        if ((rc = resolve_profile(c))) return rc;  // waits for the slice: the staging area is reused by the next one
        sum.fill_ms += c->prof.fill_ms;
        sum.trace_ms += c->prof.trace_ms;
        sum.finalize_ms += c->prof.finalize_ms;
        sum.total_ms += c->prof.total_ms;
        sum.cells += c->prof.cells;
        sum.fill_launches += c->prof.fill_launches;
        sum.ckpt_interval = std::max(sum.ckpt_interval, c->prof.ckpt_interval);
        sum.ckpt_bytes = std::max(sum.ckpt_bytes, c->prof.ckpt_bytes);
        sum.trace_margin = std::max(sum.trace_margin, c->prof.trace_margin);
        sum.lds_ckpt = std::max(sum.lds_ckpt, c->prof.lds_ckpt);
        sum.n_tasks += c->prof.n_tasks;
        sum.n_chunks = std::max(sum.n_chunks, c->prof.n_chunks);
        sum.non_finite_reads += c->prof.non_finite_reads;
    }
    c->prof = sum;
    return SFA_OK;
}

// Core of both align entry points: queries already in HBM, results left in HBM.
int align_device(sfa_ctx *c, const float *d_queries, const int64_t *q_off, int32_t n, ResultRow *d_out) {
    if (n == 0) return SFA_OK;
    // ---- host: plan the batch (quads, classes, chunks, checkpoint interval) -------------------------------
    sfa::PlanParams pp;
    pp.n_sims = static_cast<int64_t>(c->cu_count) * 4;
    pp.waves_per_simd = c->opt_waves_per_simd;
    pp.single_pass = c->opt_single_pass != 0;
    pp.ckpt_interval = c->opt_ckpt_interval;
    pp.ckpt_budget_bytes = c->opt_ckpt_budget;
    pp.trace_margin = c->opt_trace_margin;
    pp.lane_widening = c->opt_lane_widening;
    pp.widen_below = c->opt_widen_below;
    pp.column_segments = c->opt_column_segments;
    pp.segment_warm_windows = c->opt_segment_warm;
    pp.allow_segments = !(c->flag & SFA_DTW) && !c->no_segments_once;
    pp.lds_ckpt = static_cast<int>(c->opt_lds_ckpt);
    pp.std_dtw = (c->flag & SFA_DTW) != 0;
    pp.mixed_quads = c->opt_mixed_quads != 0;
    pp.span_sixteenths = c->opt_adaptive_margin ? c->span_sixteenths : 0;
    std::vector<int32_t> long_reads;  // queries beyond the wave kernels' 2048 events: row strips, after the rest of the batch
    int64_t long_events = 0, long_max = 0;
    for (int32_t i = 0; i < n; ++i)
        if (q_off[i + 1] - q_off[i] > sfa::kMaxQuery) {
            long_reads.push_back(i);
            long_events += q_off[i + 1] - q_off[i];
            long_max = std::max<int64_t>(long_max, q_off[i + 1] - q_off[i]);
        }
    pp.skip_long = !long_reads.empty();
    sfa::BatchPlan &plan = c->plan;  // kept with the context: its vectors are reused by every batch
    std::string perr;
    if (int rc = sfa::plan_batch(q_off, n, c->h_job_len, c->total_cols, pp, &plan, &perr)) return fail(rc, "%s", perr.c_str());
    if (!c->in_slice && !plan.single_pass && c->opt_ckpt_interval == 0 && plan.ck_shift > 9 && n >= 2 * c->opt_min_slice_reads) {
        // checkpoints at T = 512 would take about ck_bytes * T/512
        const int64_t want = (plan.ck_floats * 4 * (1ll << (plan.ck_shift - 9)) + pp.ckpt_budget_bytes - 1) / std::max<int64_t>(pp.ckpt_budget_bytes, 1);
        const int32_t slices = static_cast<int32_t>(std::min<int64_t>(want, n / c->opt_min_slice_reads));
        if (slices > 1) return align_sliced(c, d_queries, q_off, n, d_out, slices);
    }

    const int32_t n_quads = plan.n_quads, n_chunks = plan.n_chunks, n_jobs = c->n_jobs;
    // staging layout: q_off[n+1] | order[4*n_quads] | quad_qlen[n_quads] | slot[n] | chunk_begin[n_chunks+1] | job_ck_off[n_jobs+1]
    const size_t o_qoff = 0;
    const size_t o_order = o_qoff + sizeof(int64_t) * (n + 1);
    const size_t o_qq = o_order + sizeof(int32_t) * 4 * std::max(n_quads, 1);
    const size_t o_slot = o_qq + sizeof(int32_t) * std::max(n_quads, 1);
    const size_t o_chunk = o_slot + sizeof(int32_t) * n;
    const size_t o_ckoff = o_chunk + sizeof(int32_t) * (n_chunks + 1);
    const size_t stage_bytes = o_ckoff + sizeof(int32_t) * (n_jobs + 1);
    int rc;
    if ((rc = c->h_stage.reserve(stage_bytes)) || (rc = c->d_stage.reserve(stage_bytes))) return rc;
    char *hs = c->h_stage.as<char>();
    memcpy(hs + o_qoff, q_off, sizeof(int64_t) * (n + 1));
    memcpy(hs + o_order, plan.order.data(), sizeof(int32_t) * plan.order.size());
    memcpy(hs + o_qq, plan.quad_qlen.data(), sizeof(int32_t) * plan.quad_qlen.size());
    memcpy(hs + o_slot, plan.slot_of_read.data(), sizeof(int32_t) * n);
    memcpy(hs + o_chunk, plan.chunk_begin.data(), sizeof(int32_t) * (n_chunks + 1));
    memcpy(hs + o_ckoff, plan.job_ck_off.data(), sizeof(int32_t) * (n_jobs + 1));

    const size_t n_part = static_cast<size_t>(std::max(n_quads, 1)) * n_chunks * 4;
    if ((rc = c->d_pbest.reserve(4 * n_part)) || (rc = c->d_pend.reserve(4 * n_part)) || (rc = c->d_pjob.reserve(4 * n_part)) ||
        (rc = c->d_psecond.reserve(4 * n_part)) || (rc = c->d_wjob.reserve(4 * (size_t)n)) || (rc = c->d_wend.reserve(4 * (size_t)n)) ||
        (rc = c->d_tst.reserve(8 * (size_t)n)) || (rc = c->d_wscore.reserve(4 * (size_t)n)))
        return rc;
    if ((rc = c->d_wchunk.reserve(4 * static_cast<size_t>(n)))) return rc;
    // pass 2 inside the fill launch pays when the launch has more tasks than wave slots: its tickets then come up as the fill
    // drains.  With everything resident from the start the pass-2 waves would only sit next to the fill waves and poll
    // (measured: 2 048 reads 2.9 -> 3.3 ms per batch), so small launches keep the separate pass-2 launch.
    const bool fused = plan.lds_ckpt && c->opt_fused_trace &&
                       (c->opt_fused_trace > 1 || static_cast<int64_t>(n_quads) * n_chunks > static_cast<int64_t>(c->cu_count) * 4 * SFA_LCK_WAVES);
    if (fused && (rc = c->d_args.reserve(sizeof(DpArgs)))) return rc;
    if (fused && ((rc = c->d_ticket.reserve(64)) || (rc = c->d_quaddone.reserve(4 * static_cast<size_t>(std::max(n_quads, 1)))))) return rc;
    if (plan.lds_ckpt && ((rc = c->d_bestrec.reserve(sizeof(float) * sfa::kLdsCkPlanes * 64 * n_part / 4)) || (rc = c->d_beste.reserve(4 * n_part)) ||
                          (rc = c->d_gbest.reserve(4 * static_cast<size_t>(n)))))
        return rc;
    if ((rc = c->d_bad.reserve(static_cast<size_t>(n))) || (rc = c->d_badcount.reserve(256)) || (rc = c->h_badcount.reserve(256))) return rc;
    if (plan.single_pass && (rc = c->d_pst.reserve(4 * n_part))) return rc;
    if (!plan.single_pass && plan.ck_floats > 0 && (rc = c->d_ck.reserve(sizeof(float) * plan.ck_floats))) return rc;
    const int32_t verify_planes = plan.max_R + 1;
    if (plan.n_seg > 1) {
        const size_t vbytes = sizeof(float) * 64 * verify_planes * 2 * static_cast<size_t>(plan.n_seg) * n_jobs * std::max(n_quads, 1);
        if ((rc = c->d_verify.reserve(vbytes)) || (rc = c->d_segfail.reserve(4 * static_cast<size_t>(std::max(n_quads, 1)))) ||
            (rc = c->h_flags.reserve(4 * static_cast<size_t>(std::max(n_quads, 1)))))
            return rc;
    }

    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_stage.p, hs, stage_bytes, hipMemcpyHostToDevice, st));
    char *ds = c->d_stage.as<char>();

    const bool std_dtw = (c->flag & SFA_DTW) != 0;
    DpArgs da{};
    da.queries = d_queries;
    da.q_off = reinterpret_cast<const int64_t *>(ds + o_qoff);
    da.order = reinterpret_cast<const int32_t *>(ds + o_order);
    da.quad_qlen = reinterpret_cast<const int32_t *>(ds + o_qq);
    da.ref = c->d_ref.as<float>();
    da.job_off = c->d_job_off.as<int64_t>();
    da.job_len = c->d_job_len.as<int32_t>();
    da.chunk_begin = reinterpret_cast<const int32_t *>(ds + o_chunk);
    da.job_ck_off = reinterpret_cast<const int32_t *>(ds + o_ckoff);
    da.ck = c->d_ck.as<float>();
    da.p_best = c->d_pbest.as<float>();
    da.p_end = c->d_pend.as<int32_t>();
    da.p_st = c->d_pst.as<int32_t>();
    da.p_job = c->d_pjob.as<int32_t>();
    da.p_second = c->d_psecond.as<float>();
    da.w_job = c->d_wjob.as<int32_t>();
    da.w_end = c->d_wend.as<int32_t>();
    da.w_score = c->d_wscore.as<float>();
    da.n_reads_total = n;
    da.n_cls = static_cast<int32_t>(plan.classes.size());
    for (int i = 0; i < da.n_cls; ++i) {
        da.cls[i].R = plan.classes[i].R;
        da.cls[i].lanes = plan.classes[i].lanes;
        da.cls[i].quad_base = plan.classes[i].quad_base;
        da.cls[i].n_quads = plan.classes[i].n_quads;
        da.cls[i].task_base = plan.classes[i].quad_base * n_chunks;  // classes are contiguous in quad order
        da.cls[i].ck_base = plan.classes[i].ck_base;
    }
    da.n_chunks = n_chunks;
    da.n_tasks = n_quads * n_chunks;
    da.rev_query = ((c->flag & SFA_RNA) && !(c->flag & SFA_INV)) ? 1 : 0;
    da.ck_shift = plan.single_pass ? 0 : plan.ck_shift;
    da.trace_margin = plan.trace_margin;
    da.n_seg = plan.n_seg;
    da.warm_windows = plan.warm_windows;
    da.n_jobs = n_jobs;
    da.verify_planes = verify_planes;
    da.verify = c->d_verify.as<float>();
    da.seg_fail = c->d_segfail.as<int32_t>();
    da.best_rec = c->d_bestrec.as<float>();
    da.best_e = c->d_beste.as<int32_t>();
    da.g_best = c->d_gbest.as<unsigned>();
    da.w_chunk = c->d_wchunk.as<int32_t>();
    da.best_planes = sfa::kLdsCkPlanes;
    da.lck_shift = plan.lck_shift;
    da.coarse_every = (plan.lds_ckpt && plan.ck_shift >= plan.lck_shift) ? (1 << (plan.ck_shift - plan.lck_shift)) : 1;
    da.ticket = c->d_ticket.as<unsigned>();
    da.quad_done = c->d_quaddone.as<int32_t>();
    da.n_quads_total = n_quads;
    da.job_contig = c->d_job_contig.as<int32_t>();
    da.job_strand = c->d_job_strand.as<int8_t>();
    da.ref_len = c->d_ref_len.as<int32_t>();
    da.ref_st_offset = c->d_ref_off.as<int32_t>();
    da.bad = c->d_bad.as<uint8_t>();
    da.out = d_out;
    da.prio_unit = static_cast<int32_t>(c->opt_prio_unit);
    if ((rc = c->d_started.reserve(64))) return rc;
    da.started = c->d_started.as<unsigned>();
    da.err = c->d_badcount.as<unsigned>() + 4;
    {   // a pass-2 wave legitimately waits for as long as one fill task of its quad runs: never less than ~5x that (1 us per
        // column of the longest chunk, against 0.2 measured), however small the option -- a 250 Mb strand is minutes, not a hang
        int64_t longest = 0;
        for (int32_t ch = 0; ch < n_chunks; ++ch) {
            int64_t cols = 0;
            for (int32_t j = plan.chunk_begin[ch]; j < plan.chunk_begin[ch + 1] && j < n_jobs; ++j) cols += c->h_job_len[j];
            longest = std::max(longest, cols);
        }
        da.spin_limit = std::max<int64_t>(c->opt_spin_limit_ms, longest / 1000) * 100000;  // 100 MHz ticks
    }
    da.debug_drop_quad = static_cast<int32_t>(c->opt_debug_drop_quad);
#ifdef SFA_TASK_TIMES
    if ((rc = c->d_times.reserve(24 * static_cast<size_t>(std::max(da.n_tasks, 1))))) return rc;
    da.task_times = c->d_times.as<unsigned long long>();
    c->n_times = da.n_tasks;
#endif

    FinalizeArgs fz{};
    fz.slot_of_read = reinterpret_cast<const int32_t *>(ds + o_slot);
    fz.p_best = da.p_best;
    fz.p_end = da.p_end;
    fz.p_st = da.p_st;
    fz.p_job = da.p_job;
    fz.p_second = da.p_second;
    fz.job_contig = c->d_job_contig.as<int32_t>();
    fz.job_strand = c->d_job_strand.as<int8_t>();
    fz.ref_len = c->d_ref_len.as<int32_t>();
    fz.ref_st_offset = c->d_ref_off.as<int32_t>();
    fz.w_job = da.w_job;
    fz.w_end = da.w_end;
    fz.w_score = da.w_score;
    fz.w_chunk = da.w_chunk;
    fz.t_st = c->d_tst.as<int32_t>();
    fz.out = d_out;
    fz.bad = c->d_bad.as<uint8_t>();
    fz.q_off = da.q_off;
    fz.max_query = long_reads.empty() ? 0 : sfa::kMaxQuery;
    fz.n_reads = n;
    fz.n_chunks = n_chunks;
    fz.span_hist = c->d_badcount.as<unsigned>() + 8;
    const dim3 fgrid((n + 255) / 256), fblock(256);

    if (da.prio_unit > 0) HIP_TRY(hipMemsetAsync(c->d_started.p, 0, 4, st));
    HIP_TRY(hipEventRecord(c->ev[0], st));
    // reads with a NaN / inf query value are skipped (the reference aborts on them, see sdtw_screen_kernel)
    HIP_TRY(hipMemsetAsync(c->d_badcount.p, 0, 32 + 4 * sfa::kSpanBuckets, st));  // word 0: non-finite reads; words 4..6: error words of the in-launch waits; words 8..39: span histogram
    hipLaunchKernelGGL(sfa::sdtw_screen_kernel, dim3((n + 3) / 4), dim3(256), 0, st, d_queries, da.q_off, n, c->d_bad.as<uint8_t>(),
                       c->d_badcount.as<unsigned>());
    KERNEL_TRY();
    // Queries beyond 2048 events: row strips, on their own stream BESIDE the wave kernels of the shorter reads of the batch (a
    // handful of short reads is one sweep's latency on an empty chip: 6 + 2.5 ms in front of 110 ms of strips when run in a
    // row).  The two paths write disjoint rows (the finalize kernels here leave the long reads' rows alone).
    int32_t long_launches = 0;
    if (!long_reads.empty()) {
        hipStream_t ls = st;
        if (c->opt_long_overlap) {
            ls = c->stream_long;
            HIP_TRY(hipEventRecord(c->lev[0], st));  // queries, offsets and the non-finite screen are ready
            HIP_TRY(hipStreamWaitEvent(ls, c->lev[0], 0));
        }
        c->prof.fill_launches = 0;  // (counted per group of long reads inside)
        if ((rc = align_long(c, d_queries, da.q_off, q_off, long_reads, long_max, d_out, ls))) {
            (void)hipStreamSynchronize(ls);  // nothing of a failed call may still be running when the caller reuses its buffers
            return rc;
        }
        long_launches = c->prof.fill_launches;
        if (c->opt_long_overlap) HIP_TRY(hipEventRecord(c->lev[1], ls));
    }
    if (n_quads > 0) {
        if (plan.lds_ckpt) {
            HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_gbest.p), 0x7f800000, static_cast<size_t>(n), st));  // +inf: no score seen yet
            if (fused) {
                HIP_TRY(hipMemsetAsync(c->d_ticket.p, 0, 4, st));
                HIP_TRY(hipMemsetAsync(c->d_quaddone.p, 0, 4 * static_cast<size_t>(n_quads), st));
                fz.mode = 3;  // rows of the reads in no quad; every other row is written by the launch's pass-2 waves
                hipLaunchKernelGGL(sfa::sdtw_finalize_kernel, fgrid, fblock, 0, st, fz);
                KERNEL_TRY();
                da.self = c->d_args.as<DpArgs>();
                HIP_TRY(hipMemcpyAsync(c->d_args.p, &da, sizeof(DpArgs), hipMemcpyHostToDevice, st));  // (pageable source: staged before the call returns)
                launch_fill_fused(plan.max_R, std_dtw, da, st, c->cu_count);
            } else {
                launch_fill_lck(plan.max_R, std_dtw, da, st);
            }
        } else if (plan.single_pass) {
            launch_fill<true>(plan.max_R, std_dtw, da, st);
        } else {
            launch_fill<false>(plan.max_R, std_dtw, da, st);
        }
        KERNEL_TRY();
        if (plan.n_seg > 1) {  // every hand-over between consecutive segments: assumed state == reached state?
            HIP_TRY(hipMemsetAsync(c->d_segfail.p, 0, 4 * static_cast<size_t>(n_quads), st));
            const int64_t waves = static_cast<int64_t>(n_quads) * n_jobs * (plan.n_seg - 1);
            hipLaunchKernelGGL(sfa::sdtw_verify_kernel, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, st, da, n_quads);
            KERNEL_TRY();
        }
    }
    HIP_TRY(hipEventRecord(c->ev[1], st));
    if (fused && n_quads > 0) {  // nothing left to do: rows are complete
        HIP_TRY(hipEventRecord(c->ev[2], st));
        HIP_TRY(hipEventRecord(c->ev[3], st));
    } else {
    fz.mode = plan.single_pass ? 0 : 1;
    hipLaunchKernelGGL(sfa::sdtw_finalize_kernel, fgrid, fblock, 0, st, fz);
    KERNEL_TRY();
    HIP_TRY(hipEventRecord(c->ev[2], st));
    if (!plan.single_pass && n_quads > 0) {
        DpArgs ta = da;
        for (int i = 0; i < ta.n_cls; ++i) ta.cls[i].task_base = ta.cls[i].quad_base;  // one task per quad
        ta.n_tasks = n_quads;
        if (plan.lds_ckpt)
            launch_trace_lck(plan.max_R, std_dtw, ta, c->d_tst.as<int32_t>(), st);
        else
            launch_trace(plan.max_R, std_dtw, ta, c->d_tst.as<int32_t>(), st);
        KERNEL_TRY();
        HIP_TRY(hipEventRecord(c->ev[3], st));
        fz.mode = 2;
        hipLaunchKernelGGL(sfa::sdtw_finalize_kernel, fgrid, fblock, 0, st, fz);
        KERNEL_TRY();
    } else {
        HIP_TRY(hipEventRecord(c->ev[3], st));
    }
    }
    c->prof.fill_launches = (n_quads > 0 ? 1 : 0) + long_launches;
    c->long_pending = !long_reads.empty();
    if (c->long_pending) {  // join: what is left of the strips when the wave kernels are through counts as fill time
        HIP_TRY(hipEventRecord(c->ev[5], st));
        if (c->opt_long_overlap) HIP_TRY(hipStreamWaitEvent(st, c->lev[1], 0));
    }
    HIP_TRY(hipMemcpyAsync(c->h_badcount.p, c->d_badcount.p, 32 + 4 * sfa::kSpanBuckets, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(c->ev[4], st));

    c->prof.cells = (plan.query_events + long_events) * c->total_cols;
    c->prof.ckpt_interval = plan.single_pass ? 0 : (plan.ck_shift ? (1 << plan.ck_shift) : 0);
    c->prof.ckpt_bytes = plan.single_pass ? 0 : static_cast<int64_t>(sizeof(float)) * plan.ck_floats;
    c->prof.lds_ckpt = plan.lds_ckpt ? (fused ? 2 : 1) : 0;
    c->prof.trace_margin = plan.trace_margin;
    c->prof.n_tasks = da.n_tasks;
    c->prof.n_chunks = n_chunks;
    c->prof.n_segments = plan.n_seg;
    c->prof.segment_reruns = c->seg_reruns;
    c->prof_pending = true;
    if (plan.n_seg > 1 && n_quads > 0) {
        // the verdict of the hand-over checks has to be known before anybody uses the rows: wait here (these are
        // the small, latency-bound batches -- their caller is about to wait for them anyway)
        int32_t *flags = c->h_flags.as<int32_t>();
        HIP_TRY(hipMemcpyAsync(flags, c->d_segfail.p, 4 * static_cast<size_t>(n_quads), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        bool failed = false;
        for (int32_t k = 0; k < n_quads && !failed; ++k) failed = flags[k] != 0;
        if (failed) {  // a guessed state was not the true one somewhere: the batch is walked again, unsegmented
            c->no_segments_once = true;
            const int rc2 = align_device(c, d_queries, q_off, n, d_out);
            c->no_segments_once = false;
            c->seg_reruns++;
            c->prof.segment_reruns = c->seg_reruns;
            return rc2;
        }
    }
    return SFA_OK;
}

int resolve_profile(sfa_ctx *c) {
    if (!c->prof_pending) return SFA_OK;
    HIP_TRY(hipEventSynchronize(c->ev[4]));
    float a = 0, b = 0, d = 0, t = 0;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&d, c->ev[2], c->ev[3]));
    HIP_TRY(hipEventElapsedTime(&t, c->ev[0], c->ev[4]));
    if (c->long_pending) {  // the row-strip sweeps of long queries are fills
        float l = 0;
        HIP_TRY(hipEventElapsedTime(&l, c->ev[5], c->ev[4]));
        a += l;
        if (c->opt_long_overlap) {  // ... and ran BESIDE the wave kernels: the intervals on this stream say nothing about stages
            a = t;
            d = 0;
        }
        c->long_pending = false;
    }
    c->prof.events_ms = c->prof.normalise_ms = 0;
    c->prof.decode_ms = 0;
    c->prof.blow5_fallbacks = c->blow5_fallbacks;
    if (c->bev_pending) {
        float d1 = 0;
        HIP_TRY(hipEventElapsedTime(&d1, c->bev[0], c->bev[1]));
        c->prof.decode_ms = d1;
        c->bev_pending = false;
    }
    if (c->eev_pending) {
        float e1 = 0, e2 = 0;
        HIP_TRY(hipEventElapsedTime(&e1, c->eev[0], c->eev[1]));
        HIP_TRY(hipEventElapsedTime(&e2, c->eev[2], c->eev[3]));
        c->prof.events_ms = e1;
        c->prof.normalise_ms = e2;
        c->eev_pending = false;
    }
    c->prof.non_finite_reads = c->h_badcount.p ? *c->h_badcount.as<unsigned>() : 0;  // (copied before ev[4], which has been waited for)
    c->prof.fill_ms = a;
    c->prof.trace_ms = d;
    c->prof.finalize_ms = t - a - d;
    c->prof.total_ms = t;
    c->prof_pending = false;
    if (c->h_badcount.p) {  // spans of this batch's alignments -> head start of the next batch's pass 2 (sfa_plan.hpp)
        const unsigned *h = c->h_badcount.as<unsigned>() + 8;
        uint64_t total = 0;
        for (int b = 0; b < sfa::kSpanBuckets; ++b) total += h[b];
        if (total >= 64) {  // the bucket below which 99.9 % of the alignments lie, one more for safety, never above a whole query
            uint64_t acc = 0;
            int b = 0;
            for (; b < sfa::kSpanBuckets; ++b) {
                acc += h[b];
                if (acc * 1000 >= total * 999) break;
            }
            c->span_sixteenths = std::min(16, b + 2);
        }
    }
    if (c->h_badcount.p) {  // a wave of the batch gave up waiting for another one (bounded_wait_ge): the rows are not to be used
        const unsigned *e = c->h_badcount.as<unsigned>() + 4;
        if (e[0] == sfa::kErrQuadWait)
            return fail(SFA_EKERNEL, "fused launch: pass 2 of quad %u waited %lld ms for its fill tasks (%u of them had completed); rows of this batch are invalid",
                        e[1], (long long)c->opt_spin_limit_ms, e[2]);
        if (e[0] == sfa::kErrStripWait)
            return fail(SFA_EKERNEL, "row strips: a strip waited %lld ms for column %u of the row above (column %u was published); rows of this batch are invalid",
                        (long long)c->opt_spin_limit_ms, e[1], e[2]);
        if (e[0]) return fail(SFA_EKERNEL, "device error word %u (%u, %u)", e[0], e[1], e[2]);
    }
    return SFA_OK;
}

}  // namespace

extern "C" {

const char *sfa_last_error(void) { return g_err.c_str(); }
void sfa_set_error_(const char *msg) { g_err = msg ? msg : ""; }  // for the host-side units of this library
const char *sfa_version(void) { return SFA_VERSION; }
#ifndef SFA_BUILD_ID
#define SFA_BUILD_ID "unknown"
#endif
const char *sfa_build_id(void) { return SFA_BUILD_ID; }

}  // extern "C"

// The reference event model as the kernels want it, built once on the host: every (contig,strand) array in the
// reference's processing order (contig ascending, '+' before '-', src/sigfish.c:870-960) inside one buffer, +inf around
// each (cells of columns < 0 and past the end evaluate to +inf, see sweep_begin() in the kernels).
struct HostRef {
    int32_t num_ref = 0, n_jobs = 0;
    int64_t total_cols = 0;
    std::vector<float> packed;
    std::vector<int64_t> job_off;
    std::vector<int32_t> job_len, job_contig, ref_off, ref_len;
    std::vector<int8_t> job_strand;
};

static int pack_reference(const sfa_ref_t *ref, uint32_t flag, HostRef *h) {
    if (!ref || ref->num_ref <= 0 || !ref->ref_lengths || !ref->forward) return fail(SFA_EINVAL, "sfa_init: null or empty reference");
    const bool rna = (flag & SFA_RNA) != 0;
    if (!rna && !ref->reverse) return fail(SFA_EINVAL, "sfa_init: DNA needs reverse arrays");
    if ((flag & SFA_DTW) && !rna) return fail(SFA_EINVAL, "sfa_init: --dtw-std is RNA only (src/dtw_main.c:249-252)");
    const int strands = rna ? 1 : 2;
    h->num_ref = ref->num_ref;
    h->n_jobs = ref->num_ref * strands;
    h->job_off.resize(h->n_jobs);
    h->job_len.resize(h->n_jobs);
    h->job_contig.resize(h->n_jobs);
    h->job_strand.resize(h->n_jobs);
    h->ref_off.resize(ref->num_ref);
    h->ref_len.assign(ref->ref_lengths, ref->ref_lengths + ref->num_ref);
    int64_t total = sfa::kRefPad;
    for (int32_t r = 0; r < ref->num_ref; ++r) {
        const int32_t rl = ref->ref_lengths[r];
        if (rl <= 0) return fail(SFA_EINVAL, "contig %d has non-positive length %d", r, rl);
        h->ref_off[r] = ref->ref_st_offset ? ref->ref_st_offset[r] : 0;
        for (int s = 0; s < strands; ++s) {
            const int32_t j = r * strands + s;
            h->job_off[j] = total;
            h->job_len[j] = rl;
            h->job_contig[j] = r;
            h->job_strand[j] = s == 0 ? '+' : '-';
            total += rl + sfa::kRefPad;
            h->total_cols += rl;
        }
    }
    h->packed.assign(total, INFINITY);
    for (int32_t j = 0; j < h->n_jobs; ++j) {
        const float *src = (h->job_strand[j] == '+') ? ref->forward[h->job_contig[j]] : ref->reverse[h->job_contig[j]];
        if (!src) return fail(SFA_EINVAL, "missing reference array for contig %d", h->job_contig[j]);
        memcpy(&h->packed[h->job_off[j]], src, sizeof(float) * h->job_len[j]);
    }
    return SFA_OK;
}

// One context on one device.  `peer`: a context that already holds the packed arrays -- they then travel device to
// device (hipMemcpyPeer: xGMI between the GPUs of a node) instead of crossing PCIe once more (one broadcast
// of the reference event model, root = the first device).
static int create_context(sfa_ctx **out, const HostRef &h, uint32_t flag, int device, const sfa_ctx *peer) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SFA_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SFA_EINVAL, "device %d out of range (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SFA_ENODEV, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

    sfa_ctx *c = new sfa_ctx();
    c->device = device;
    c->flag = flag;
    c->cu_count = prop.multiProcessorCount;
    c->num_ref = h.num_ref;
    c->n_jobs = h.n_jobs;
    c->total_cols = h.total_cols;
    c->h_job_len = h.job_len;
    auto bail = [&](int rc) {
        sfa_destroy(c);
        return rc;
    };
    // CU partition (experiment, profiles/r03_logs/cu_partition_*.log): the inflate kernel (57 KB of LDS per wave) and the
    // LDS-checkpoint fill (152 of 160 KB per CU) cannot share a CU, so with two contexts in flight the decoder of one batch queues
    // behind the fill of the other.  SFA_RESERVED_CUS=N gives the pre-alignment stages (sfa_align_blow5 / sfa_align_raw up to the
    // event counts) a stream masked to the LAST N CUs and the alignment streams the others.
    int reserved = 0;
    if (const char *e = getenv("SFA_RESERVED_CUS")) reserved = atoi(e);
    if (reserved > 0 && reserved < c->cu_count) {
        const int words = (c->cu_count + 31) / 32;
        std::vector<uint32_t> main_mask(words, 0), pre_mask(words, 0);
        for (int i = 0; i < c->cu_count; ++i) (i < c->cu_count - reserved ? main_mask : pre_mask)[i / 32] |= 1u << (i % 32);
        if (hipExtStreamCreateWithCUMask(&c->stream, words, main_mask.data()) != hipSuccess ||
            hipExtStreamCreateWithCUMask(&c->stream_long, words, main_mask.data()) != hipSuccess ||
            hipExtStreamCreateWithCUMask(&c->stream_pre, words, pre_mask.data()) != hipSuccess)
            return bail(fail(SFA_ENODEV, "hipExtStreamCreateWithCUMask failed (SFA_RESERVED_CUS=%d of %d CUs)", reserved, c->cu_count));
        c->cu_count -= reserved;  // what the planner and the launch sizes see
        c->pre_cus = reserved;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(SFA_ENODEV, "hipStreamCreate failed"));
        if (hipStreamCreateWithFlags(&c->stream_long, hipStreamNonBlocking) != hipSuccess) return bail(fail(SFA_ENODEV, "hipStreamCreate failed"));
    }
    for (auto &e : c->lev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(fail(SFA_ENODEV, "hipEventCreate failed"));
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(SFA_ENODEV, "hipEventCreate failed"));
    for (auto &e : c->eev)
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(SFA_ENODEV, "hipEventCreate failed"));
    for (auto &e : c->bev)
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(SFA_ENODEV, "hipEventCreate failed"));
    int rc;
    const size_t ref_bytes = sizeof(float) * h.packed.size();
    if ((rc = c->d_ref.reserve(ref_bytes)) || (rc = c->d_job_off.reserve(sizeof(int64_t) * c->n_jobs)) ||
        (rc = c->d_job_len.reserve(sizeof(int32_t) * c->n_jobs)) || (rc = c->d_job_contig.reserve(sizeof(int32_t) * c->n_jobs)) ||
        (rc = c->d_job_strand.reserve(c->n_jobs)) || (rc = c->d_ref_len.reserve(sizeof(int32_t) * h.num_ref)) ||
        (rc = c->d_ref_off.reserve(sizeof(int32_t) * h.num_ref)))
        return bail(rc);
#define UP(dst, src, bytes) \
    if (hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(SFA_ENODEV, "upload of reference model failed"))
    if (peer) {
        if (hipMemcpyPeer(c->d_ref.p, device, peer->d_ref.p, peer->device, ref_bytes) != hipSuccess)
            return bail(fail(SFA_ENODEV, "device-to-device copy of the reference model (%d -> %d) failed", peer->device, device));
    } else {
        UP(c->d_ref.p, h.packed.data(), ref_bytes);
    }
    UP(c->d_job_off.p, h.job_off.data(), sizeof(int64_t) * c->n_jobs);
    UP(c->d_job_len.p, h.job_len.data(), sizeof(int32_t) * c->n_jobs);
    UP(c->d_job_contig.p, h.job_contig.data(), sizeof(int32_t) * c->n_jobs);
    UP(c->d_job_strand.p, h.job_strand.data(), c->n_jobs);
    UP(c->d_ref_len.p, h.ref_len.data(), sizeof(int32_t) * h.num_ref);
    UP(c->d_ref_off.p, h.ref_off.data(), sizeof(int32_t) * h.num_ref);
#undef UP
    *out = c;
    return SFA_OK;
}

// ---- group contexts: the same entry points over several devices ---------------------------------------------------
// contiguous read range of shard r: [r*n/G, (r+1)*n/G)
static void shard_ranges(int32_t n, size_t g, std::vector<int32_t> *lo) {
    lo->resize(g + 1);
    for (size_t r = 0; r <= g; ++r) (*lo)[r] = static_cast<int32_t>(static_cast<int64_t>(n) * static_cast<int64_t>(r) / static_cast<int64_t>(g));
}

// run fn(shard index) for every shard, each on the shard's own host thread (HIP's current device and sfa_last_error are
// per thread), shard 0 on the caller's; the first failure's code and message become the caller's
template <typename F>
static int for_each_shard(sfa_ctx *g, F fn) {
    const size_t G = g->shards.size();
    for (size_t r = 1; r < G; ++r) g->workers[r - 1]->post([&fn, r] { return fn(r); });
    int rc = fn(0);
    std::string msg = rc ? g_err : std::string();
    for (size_t r = 1; r < G; ++r) {  // (always all of them: fn and what it captures must outlive every worker's job)
        const auto res = g->workers[r - 1]->wait();
        if (res.first && !rc) {
            rc = res.first;
            msg = res.second;
        }
    }
    if (rc) g_err = msg;
    return rc;
}

extern "C" {

int sfa_init(sfa_ctx_t **out, const sfa_ref_t *ref, uint32_t flag, int device) {
    if (!out) return fail(SFA_EINVAL, "sfa_init: null context pointer");
    HostRef h;
    if (int rc = pack_reference(ref, flag, &h)) return rc;
    return create_context(out, h, flag, device, nullptr);
}

int sfa_init_devices(sfa_ctx_t **out, const sfa_ref_t *ref, uint32_t flag, const int *devices, int n_devices) {
    if (!out || !devices || n_devices <= 0) return fail(SFA_EINVAL, "sfa_init_devices: null or empty device list");
    HostRef h;
    if (int rc = pack_reference(ref, flag, &h)) return rc;
    sfa_ctx *g = new sfa_ctx();
    g->flag = flag;
    g->device = devices[0];
    for (int i = 0; i < n_devices; ++i) {
        sfa_ctx *c = nullptr;
        // one upload from the host, then device to device from the first shard
        if (int rc = create_context(&c, h, flag, devices[i], g->shards.empty() ? nullptr : g->shards[0])) {
            sfa_destroy(g);
            return rc;
        }
        g->shards.push_back(c);
    }
    for (int i = 1; i < n_devices; ++i) g->workers.emplace_back(new ShardWorker());
    *out = g;
    return SFA_OK;
}

void sfa_destroy(sfa_ctx_t *c) {
    if (!c) return;
    if (!c->shards.empty()) {
        c->workers.clear();  // joins the shard threads (none has a job: every entry point waits for its shards)
        for (sfa_ctx *sh : c->shards) sfa_destroy(sh);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->d_ref, &c->d_job_off, &c->d_job_len, &c->d_job_contig, &c->d_job_strand, &c->d_ref_len, &c->d_ref_off,
                      &c->d_queries, &c->d_stage, &c->d_pbest, &c->d_pend, &c->d_pst, &c->d_pjob, &c->d_psecond, &c->d_wjob,
                      &c->d_wend, &c->d_wscore, &c->d_tst, &c->d_ck, &c->d_out, &c->e_raw, &c->e_rawoff, &c->e_scale, &c->e_sum,
                      &c->e_sumsq, &c->e_t1, &c->e_t2, &c->e_evoff, &c->e_evstart, &c->e_evlen, &c->e_evmean, &c->e_evstdv, &c->e_nev,
                      &c->e_qstart, &c->e_qoff, &c->e_b0, &c->e_b1, &c->e_b2, &c->e_flag, &c->e_qev, &c->e_pflag, &c->d_verify, &c->d_segfail, &c->d_bndc, &c->d_bnds, &c->d_long, &c->d_lbest, &c->d_lsecond, &c->d_lend, &c->d_lwin, &c->d_lck, &c->d_lprog, &c->d_lticket, &c->d_times, &c->d_started, &c->d_bad, &c->d_badcount, &c->d_bestrec, &c->d_beste, &c->d_gbest, &c->d_wchunk, &c->d_ticket, &c->d_quaddone, &c->d_args, &c->b_in, &c->b_inoff, &c->b_out, &c->b_outoff, &c->b_len, &c->b_head, &c->b_bad})
        b->release();
    c->h_stage.release();
    c->h_out.release();
    c->h_small.release();
    c->h_flags.release();
    c->h_queries.release();
    c->h_long.release();
    c->h_badcount.release();
    c->h_head.release();
    for (auto &e : c->bev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->eev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->lev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream_pre) (void)hipStreamDestroy(c->stream_pre);
    if (c->stream_long) (void)hipStreamDestroy(c->stream_long);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sfa_set_option(sfa_ctx_t *c, const char *key, int64_t value) {
    if (!c || !key) return fail(SFA_EINVAL, "sfa_set_option: null argument");
    if (!c->shards.empty()) {
        for (sfa_ctx *sh : c->shards)
            if (int rc = sfa_set_option(sh, key, value)) return rc;
        return SFA_OK;
    }
    const std::string k(key);
    if (k == "single_pass") {
        c->opt_single_pass = value != 0;
    } else if (k == "ckpt_interval") {
        if (value != 0 && (value < 4 || (value & (value - 1)))) return fail(SFA_EINVAL, "ckpt_interval must be 0 or a power of two >= 4");
        c->opt_ckpt_interval = value;
    } else if (k == "ckpt_budget_bytes") {
        if (value < 0) return fail(SFA_EINVAL, "ckpt_budget_bytes must be >= 0");
        c->opt_ckpt_budget = value;
    } else if (k == "trace_margin") {
        c->opt_trace_margin = value;
    } else if (k == "lane_widening") {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(SFA_EINVAL, "lane_widening must be 0 (auto), 1, 2 or 4");
        c->opt_lane_widening = value;
    } else if (k == "ev_parallel_peaks") {
        c->opt_ev_parallel_peaks = value != 0;
    } else if (k == "ev_parallel_prefix") {
        c->opt_ev_parallel_prefix = value != 0;
    } else if (k == "min_slice_reads") {
        if (value < 1) return fail(SFA_EINVAL, "min_slice_reads must be >= 1");
        c->opt_min_slice_reads = value;
    } else if (k == "segment_warm_windows") {
        if (value < 0 || value > 64) return fail(SFA_EINVAL, "segment_warm_windows must be 0..64");
        c->opt_segment_warm = value;
    } else if (k == "column_segments") {
        if (value < 0 || value > 64) return fail(SFA_EINVAL, "column_segments must be 0 (auto), 1 (off) or 2..64");
        c->opt_column_segments = value;
    } else if (k == "widen_below") {
        if (value < 0) return fail(SFA_EINVAL, "widen_below must be >= 0");
        c->opt_widen_below = value;
    } else if (k == "strip_pipeline") {
        c->opt_strip_pipeline = value != 0;
    } else if (k == "balanced_strips") {
        c->opt_balanced_strips = value != 0;
    } else if (k == "long_overlap") {
        c->opt_long_overlap = value != 0;
    } else if (k == "strip_chain") {
        c->opt_strip_chain = value != 0;
    } else if (k == "fused_trace") {
        if (value < 0 || value > 2) return fail(SFA_EINVAL, "fused_trace must be 0 (off), 1 (launches with more tasks than wave slots) or 2 (always)");
        c->opt_fused_trace = value;
    } else if (k == "lds_ckpt") {
        if (value < 0 || value > 2) return fail(SFA_EINVAL, "lds_ckpt must be 0 (off), 1 (where shapes and batch size suit) or 2 (wherever the shapes allow)");
        c->opt_lds_ckpt = value;
    } else if (k == "prio_unit") {
        if (value < 0 || value > (1 << 28)) return fail(SFA_EINVAL, "prio_unit must be 0 (off) .. 2^28");
        c->opt_prio_unit = value;
    } else if (k == "adaptive_margin") {
        c->opt_adaptive_margin = value != 0;
        c->span_sixteenths = 0;
    } else if (k == "mixed_quads") {
        c->opt_mixed_quads = value != 0;
    } else if (k == "spin_limit_ms") {
        if (value < 1 || value > 3600000) return fail(SFA_EINVAL, "spin_limit_ms must be 1 .. 3 600 000");
        c->opt_spin_limit_ms = value;
    } else if (k == "debug_drop_quad") {
        c->opt_debug_drop_quad = value;
    } else if (k == "debug_drop_strip") {
        c->opt_debug_drop_strip = value;
    } else if (k == "waves_per_simd") {
        if (value < 1 || value > 8) return fail(SFA_EINVAL, "waves_per_simd must be 1..8");
        c->opt_waves_per_simd = value;
    } else {
        return fail(SFA_EINVAL, "unknown option '%s'", key);
    }
    return SFA_OK;
}

int sfa_align_batch_device(sfa_ctx_t *c, const float *d_queries, const int64_t *q_off, int32_t n, sfa_result_t *d_out, int sync) {
    if (!c || !q_off || n < 0 || (n > 0 && (!d_queries || !d_out))) return fail(SFA_EINVAL, "sfa_align_batch_device: bad argument");
    if (!c->shards.empty())
        return fail(SFA_EINVAL, "sfa_align_batch_device: device-resident buffers belong to one device; use a single-device context "
                                "(sfa_init) per GPU, or the host-buffer entry points on a group context");
    HIP_TRY(hipSetDevice(c->device));
    // the pinned staging area is reused by every call: the previous batch must have left it
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int rc = resolve_profile(c)) return rc;
    if (int rc = align_device(c, d_queries, q_off, n, reinterpret_cast<ResultRow *>(d_out))) return rc;
    if (sync) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return resolve_profile(c);
    }
    return SFA_OK;
}

int sfa_submit_batch(sfa_ctx_t *c, const float *queries, const int64_t *q_off, int32_t n) {
    if (!c || !q_off || n < 0 || (n > 0 && !queries)) return fail(SFA_EINVAL, "sfa_submit_batch: bad argument");
    if (!c->shards.empty()) {  // every shard queues its contiguous range of reads on its own device
        shard_ranges(n, c->shards.size(), &c->shard_lo);
        c->pending_n = -1;
        const int rc = for_each_shard(c, [&](size_t r) {
            const int32_t lo = c->shard_lo[r], hi = c->shard_lo[r + 1];
            return sfa_submit_batch(c->shards[r], queries, q_off + lo, hi - lo);  // (q_off holds absolute offsets into queries)
        });
        if (!rc) c->pending_n = n;
        return rc;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));  // one batch in flight per context
    c->pending_n = -1;
    if (n == 0) {
        c->pending_n = 0;
        return SFA_OK;
    }
    const int64_t nq = q_off[n] - q_off[0];
    if (nq < 0) return fail(SFA_EINVAL, "q_off not monotone");
    int rc;
    // the device path indexes queries by q_off directly, so upload the span [q_off[0], q_off[n]) re-based to 0
    std::vector<int64_t> rebased;
    const int64_t *qo = q_off;
    if (q_off[0] != 0) {
        rebased.resize(n + 1);
        for (int32_t i = 0; i <= n; ++i) rebased[i] = q_off[i] - q_off[0];
        qo = rebased.data();
    }
    if ((rc = c->d_queries.reserve(sizeof(float) * std::max<int64_t>(nq, 1))) || (rc = c->d_out.reserve(sizeof(sfa_result_t) * n)) ||
        (rc = c->h_out.reserve(sizeof(sfa_result_t) * n)))
        return rc;
    HIP_TRY(hipMemcpyAsync(c->d_queries.p, queries + q_off[0], sizeof(float) * nq, hipMemcpyHostToDevice, c->stream));
    if ((rc = align_device(c, c->d_queries.as<float>(), qo, n, c->d_out.as<ResultRow>()))) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_out.p, c->d_out.p, sizeof(sfa_result_t) * n, hipMemcpyDeviceToHost, c->stream));
    c->pending_n = n;
    return SFA_OK;
}

int sfa_wait_batch(sfa_ctx_t *c, sfa_result_t *out, int32_t n) {
    if (!c || n < 0 || (n > 0 && !out)) return fail(SFA_EINVAL, "sfa_wait_batch: bad argument");
    if (c->pending_n < 0) return fail(SFA_EINVAL, "sfa_wait_batch: no batch was submitted");
    if (c->pending_n != n) return fail(SFA_EINVAL, "sfa_wait_batch: %d reads were submitted, %d asked for", c->pending_n, n);
    c->pending_n = -1;
    if (!c->shards.empty())  // rows of shard r go to out[lo_r, hi_r): input order, no gather step in a single process
        return for_each_shard(c, [&](size_t r) {
            const int32_t lo = c->shard_lo[r], hi = c->shard_lo[r + 1];
            return sfa_wait_batch(c->shards[r], out ? out + lo : nullptr, hi - lo);
        });
    if (n == 0) return SFA_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_out.p, sizeof(sfa_result_t) * n);
    return resolve_profile(c);
}

int sfa_align_batch(sfa_ctx_t *c, const float *queries, const int64_t *q_off, int32_t n, sfa_result_t *out) {
    if (!c || !q_off || n < 0 || (n > 0 && (!queries || !out))) return fail(SFA_EINVAL, "sfa_align_batch: bad argument");
    if (int rc = sfa_submit_batch(c, queries, q_off, n)) return rc;
    return sfa_wait_batch(c, out, n);
}

int sfa_align_events(sfa_ctx_t *c, const sfa_event_t *const *events, const int64_t *n_events, const int64_t *qstart,
                     const int64_t *qend, int32_t n, sfa_result_t *out) {
    if (!c || n < 0 || (n > 0 && (!events || !n_events || !qstart || !qend || !out)))
        return fail(SFA_EINVAL, "sfa_align_events: bad argument");
    // gather db->et[i].event[qstart..qend).mean (AoS, stride 24 B) into the packed SoA the kernels read
    std::vector<int64_t> q_off(n + 1, 0);
    for (int32_t i = 0; i < n; ++i) {
        int64_t l = 0;
        if (n_events[i] > 0 && events[i]) {
            if (qstart[i] < 0 || qend[i] < qstart[i] || qend[i] > n_events[i])
                return fail(SFA_EINVAL, "sfa_align_events: read %d has query window [%lld,%lld) outside its %lld events", i,
                            (long long)qstart[i], (long long)qend[i], (long long)n_events[i]);
            l = qend[i] - qstart[i];
        }
        q_off[i + 1] = q_off[i] + l;
    }
    // the gather reads 24 bytes per event to keep 4: a 100 000-read batch is 600 MB through one core (70-90 ms, as long as
    // the whole alignment) unless it is spread over a few threads; a single-device context gathers straight into page-locked
    // memory, from where the upload is a true asynchronous copy
    const int64_t total = q_off[n];
    float *dst = nullptr;
    std::unique_ptr<float[]> heap;
    if (c->shards.empty()) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));  // the previous batch may still be uploading from the buffer
        if (int rc = c->h_queries.reserve(sizeof(float) * static_cast<size_t>(std::max<int64_t>(total, 1)))) return rc;
        dst = c->h_queries.as<float>();
    } else {
        heap.reset(new float[static_cast<size_t>(std::max<int64_t>(total, 1))]);
        dst = heap.get();
    }
    auto gather = [&](int32_t lo, int32_t hi) {
        for (int32_t i = lo; i < hi; ++i) {
            const int64_t l = q_off[i + 1] - q_off[i];
            const sfa_event_t *ev = l ? events[i] + qstart[i] : nullptr;
            float *d = dst + q_off[i];
            for (int64_t j = 0; j < l; ++j) d[j] = ev[j].mean;
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_thr = total < (int64_t(1) << 21) ? 1 : static_cast<int>(std::min<int64_t>(std::min(8u, hw), total >> 20));
    if (n_thr <= 1) {
        gather(0, n);
    } else {  // contiguous read ranges of about equal event counts
        std::vector<std::thread> th;
        int32_t lo = 0;
        for (int t = 0; t < n_thr; ++t) {
            const int64_t want = total * (t + 1) / n_thr;
            int32_t hi = (t + 1 == n_thr) ? n : static_cast<int32_t>(std::upper_bound(q_off.begin() + lo, q_off.begin() + n + 1, want) - q_off.begin() - 1);
            hi = std::max(hi, lo);
            if (t + 1 == n_thr)
                gather(lo, hi);
            else
                th.emplace_back(gather, lo, hi);
            lo = hi;
        }
        for (std::thread &x : th) x.join();
    }
    return sfa_align_batch(c, dst, q_off.data(), n, out);
}

int sfa_align_raw(sfa_ctx_t *c, const int16_t *raw, const int64_t *raw_off, const double *scaling, int32_t n, int32_t prefix_size,
                  int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info) {
    return sfa_align_raw_ex(c, raw, raw_off, scaling, n, prefix_size, query_size, rows, info, nullptr);
}

static int align_raw_impl(sfa_ctx_t *c, const int16_t *raw, const int64_t *raw_off, const double *scaling, int32_t n, int32_t prefix_size,
                          int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info, sfa_event_t *query_events);

int sfa_align_raw_ex(sfa_ctx_t *c, const int16_t *raw, const int64_t *raw_off, const double *scaling, int32_t n, int32_t prefix_size,
                     int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info, sfa_event_t *query_events) {
    if (!c || n < 0 || (n > 0 && (!raw || !raw_off || !scaling || !rows || !info))) return fail(SFA_EINVAL, "sfa_align_raw: bad argument");
    return align_raw_impl(c, raw, raw_off, scaling, n, prefix_size, query_size, rows, info, query_events);
}

// raw == nullptr: the samples are already in c->e_raw (decoded on the device, sfa_align_blow5), laid out by raw_off
static int align_raw_impl(sfa_ctx_t *c, const int16_t *raw, const int64_t *raw_off, const double *scaling, int32_t n, int32_t prefix_size,
                          int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info, sfa_event_t *query_events) {
    if (prefix_size < 0) return fail(SFA_EINVAL, "sfa_align_raw: automatic query start (-p -1) needs the host stages");
    if (query_size <= 0) return fail(SFA_EINVAL, "sfa_align_raw: query_size must be positive");
    if (n == 0) return SFA_OK;
    if (!c->shards.empty()) {
        std::vector<int32_t> lo;
        shard_ranges(n, c->shards.size(), &lo);
        return for_each_shard(c, [&](size_t r) {
            const int32_t a = lo[r], b = lo[r + 1];
            if (a == b) return static_cast<int>(SFA_OK);
            std::vector<int64_t> off(b - a + 1);  // the shard's sample offsets start at 0
            for (int32_t i = a; i <= b; ++i) off[i - a] = raw_off[i] - raw_off[a];
            if (!raw) return fail(SFA_EINVAL, "sfa_align_raw: device-resident samples need a single-device context");
            return sfa_align_raw_ex(c->shards[r], raw + raw_off[a], off.data(), scaling + 3 * static_cast<size_t>(a), b - a, prefix_size,
                                    query_size, rows + a, info + a,
                                    query_events ? query_events + static_cast<size_t>(a) * static_cast<size_t>(query_size) : nullptr);
        });
    }
    HIP_TRY(hipSetDevice(c->device));
    if (raw) HIP_TRY(hipStreamSynchronize(c->stream));  // (device-resident samples: their decoder is still in flight on this stream)
    const int64_t total = raw_off[n] - raw_off[0];
    if (total < 0 || raw_off[0] != 0) return fail(SFA_EINVAL, "sfa_align_raw: raw_off must start at 0 and be monotone");
    const bool rna = (c->flag & SFA_RNA) != 0;
    hipStream_t st = c->stream;
    // event capacity per read: every sample can close at most one event per detector, each detector at most every
    // second sample -> n samples bound the count
    std::vector<int64_t> ev_off(n + 1);
    std::vector<float> scale(2 * static_cast<size_t>(n));
    ev_off[0] = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int64_t len = raw_off[i + 1] - raw_off[i];
        if (len < 0) return fail(SFA_EINVAL, "sfa_align_raw: raw_off not monotone at read %d", i);
        ev_off[i + 1] = ev_off[i] + len + 2;
        const float range = static_cast<float>(scaling[3 * i + 2]), dig = static_cast<float>(scaling[3 * i]);
        scale[2 * i] = static_cast<float>(scaling[3 * i + 1]);
        scale[2 * i + 1] = range / dig;  // event_single(), src/sigfish.c:343
    }
    const int64_t ev_total = ev_off[n];
    int rc;
    if ((rc = c->e_raw.reserve(2 * (size_t)std::max<int64_t>(total, 1))) || (rc = c->e_rawoff.reserve(8 * (size_t)(n + 1))) ||
        (rc = c->e_scale.reserve(8 * (size_t)n)) || (rc = c->e_sum.reserve(8 * (size_t)(total + n))) ||
        (rc = c->e_sumsq.reserve(8 * (size_t)(total + n))) || (rc = c->e_t1.reserve(4 * (size_t)std::max<int64_t>(total, 1))) ||
        (rc = c->e_t2.reserve(4 * (size_t)std::max<int64_t>(total, 1))) || (rc = c->e_evoff.reserve(8 * (size_t)(n + 1))) ||
        (rc = c->e_evstart.reserve(4 * (size_t)ev_total)) || (rc = c->e_evlen.reserve(4 * (size_t)ev_total)) ||
        (rc = c->e_evmean.reserve(4 * (size_t)ev_total)) || (rc = c->e_evstdv.reserve(4 * (size_t)ev_total)) ||
        (rc = c->e_nev.reserve(4 * (size_t)n)) || (rc = c->e_qstart.reserve(8 * (size_t)n)) || (rc = c->e_qoff.reserve(8 * (size_t)(n + 1))) ||
        (rc = c->e_flag.reserve(4 * (size_t)n)) || (rc = c->e_pflag.reserve(4 * (size_t)n)) || (rc = c->e_b0.reserve(4 * (size_t)n)) || (rc = c->e_b1.reserve(4 * (size_t)n)) || (rc = c->e_b2.reserve(4 * (size_t)n)))
        return rc;
    hipStream_t sp = c->stream_pre ? c->stream_pre : st;  // (the section ends with a host wait on it: no cross-stream event needed)
    if (raw) HIP_TRY(hipMemcpyAsync(c->e_raw.p, raw, 2 * (size_t)total, hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemcpyAsync(c->e_rawoff.p, raw_off, 8 * (size_t)(n + 1), hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemcpyAsync(c->e_scale.p, scale.data(), 8 * (size_t)n, hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemcpyAsync(c->e_evoff.p, ev_off.data(), 8 * (size_t)(n + 1), hipMemcpyHostToDevice, sp));

    sfa::EvArgs ea{};
    ea.raw = c->e_raw.as<int16_t>();
    ea.raw_off = c->e_rawoff.as<int64_t>();
    ea.scale = c->e_scale.as<float>();
    ea.sum = c->e_sum.as<double>();
    ea.sumsq = c->e_sumsq.as<double>();
    ea.t1 = c->e_t1.as<float>();
    ea.t2 = c->e_t2.as<float>();
    ea.ev_off = c->e_evoff.as<int64_t>();
    ea.ev_start = c->e_evstart.as<int32_t>();
    ea.ev_length = c->e_evlen.as<float>();
    ea.ev_mean = c->e_evmean.as<float>();
    ea.ev_stdv = c->e_evstdv.as<float>();
    ea.n_events = c->e_nev.as<int32_t>();
    ea.n_reads = n;
    // detector parameters, src/events.c:47-58
    ea.w1 = rna ? 7 : 3;
    ea.w2 = rna ? 14 : 6;
    ea.thr1 = rna ? 2.5f : 1.4f;
    ea.thr2 = 9.0f;
    ea.peak_height = rna ? 1.0f : 0.2f;
    const dim3 lane_grid((n + 63) / 64), lane_block(64);
    ea.seq_flag = c->e_flag.as<int32_t>();
    ea.use_flags = c->opt_ev_parallel_prefix ? 1 : 0;
    ea.peak_flag = c->e_pflag.as<int32_t>();
    // measured: 76 us against 1.5 ms for a 512-read batch, 2.1 ms against 1.3 ms for 16 Ki reads (it does ~1.3x the work of
    // the sequential walk, in 64x more waves): used while the batch cannot fill the chip with one read per lane pair
    ea.use_peak_flags = (c->opt_ev_parallel_peaks && n <= 8192) ? 1 : 0;
    HIP_TRY(hipEventRecord(c->eev[0], sp));
    if (ea.use_flags) hipLaunchKernelGGL(sfa::ev_prefix_par_kernel, dim3(n), dim3(64), 0, sp, ea);  // flags what it cannot do exactly
    hipLaunchKernelGGL(sfa::ev_prefix_kernel, lane_grid, lane_block, 0, sp, ea);
    hipLaunchKernelGGL(sfa::ev_tstat_kernel, dim3(n), dim3(256), 0, sp, ea);
    if (ea.use_peak_flags) hipLaunchKernelGGL(sfa::ev_peaks_spec_kernel, dim3(n), dim3(64), 0, sp, ea);  // wave per read, flags what it cannot certify
    hipLaunchKernelGGL(sfa::ev_peaks_kernel, dim3((n + 31) / 32), dim3(64), 0, sp, ea);  // two lanes per read (all reads, or the flagged ones)
    hipLaunchKernelGGL(sfa::ev_stats_kernel, dim3(n), dim3(256), 0, sp, ea);
    KERNEL_TRY();
    HIP_TRY(hipEventRecord(c->eev[1], sp));
    if ((rc = c->h_small.reserve(16 * (size_t)n))) return rc;  // page-locked: event counts, then the three raw-coordinate columns
    int32_t *nev = c->h_small.as<int32_t>();
    HIP_TRY(hipMemcpyAsync(nev, c->e_nev.p, 4 * (size_t)n, hipMemcpyDeviceToHost, sp));
    HIP_TRY(hipStreamSynchronize(sp));

    // query windows on the host (normalise_single, src/sigfish.c:433-480); the arithmetic part runs on the device
    std::vector<int64_t> qstart(n), q_off(n + 1);
    q_off[0] = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int64_t ne = nev[i];
        int64_t s0 = 0, e0 = 0;
        int status = 0;
        bool keep = ne > 0 && (raw_off[i + 1] - raw_off[i]) > 0;
        if (keep) {
            if (!(c->flag & SFA_END)) {
                s0 = prefix_size;
                e0 = s0 + query_size;
                if (s0 + 25 > ne) {
                    s0 = e0 = 0;
                    keep = false;
                    status |= 2;
                } else if (e0 > ne) {
                    e0 = ne;
                    status |= 1;
                }
            } else {
                s0 = ne - prefix_size - query_size;
                e0 = ne - prefix_size;
                if (s0 < 0) {
                    s0 = 0;
                    status |= 1;
                }
                if (e0 < 0) {
                    e0 = 0;
                    keep = false;
                    status |= 2;
                }
            }
        }
        if (!keep) s0 = e0 = 0;
        qstart[i] = s0;
        q_off[i + 1] = q_off[i] + (e0 - s0);
        info[i].n_events = ne;
        info[i].qstart = s0;
        info[i].qend = e0;
        info[i].status = status;
        info[i].pad = 0;
    }
    const int64_t nq = q_off[n];
    if ((rc = c->d_queries.reserve(4 * (size_t)std::max<int64_t>(nq, 1))) || (rc = c->d_out.reserve(sizeof(sfa_result_t) * (size_t)n)) ||
        (rc = c->h_out.reserve(sizeof(sfa_result_t) * (size_t)n)))
        return rc;
    HIP_TRY(hipMemcpyAsync(c->e_qstart.p, qstart.data(), 8 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->e_qoff.p, q_off.data(), 8 * (size_t)(n + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(c->eev[2], st));
    sfa::QueryArgs qa{c->e_evmean.as<float>(), c->e_evoff.as<int64_t>(), c->e_qstart.as<int64_t>(), c->e_qoff.as<int64_t>(),
                      c->d_queries.as<float>(), n};
    hipLaunchKernelGGL(sfa::ev_query_kernel, dim3(n), dim3(64), 0, st, qa);  // one wave per read
    sfa::BoundsArgs ba{c->e_evstart.as<int32_t>(), c->e_evlen.as<float>(), c->e_evoff.as<int64_t>(), c->e_qstart.as<int64_t>(),
                       c->e_qoff.as<int64_t>(), c->e_b0.as<int32_t>(), c->e_b1.as<int32_t>(), c->e_b2.as<float>(), n};
    hipLaunchKernelGGL(sfa::ev_bounds_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ba);
    if (query_events) {  // the query windows' event tables, for SAM output on the host
        static_assert(sizeof(sfa_event_t) == 24, "event record layout");
        const size_t qe_bytes = sizeof(sfa_event_t) * static_cast<size_t>(n) * static_cast<size_t>(query_size);
        if ((rc = c->e_qev.reserve(qe_bytes))) return rc;
        sfa::PackArgs pa{c->e_evstart.as<int32_t>(), c->e_evlen.as<float>(), c->e_evstdv.as<float>(), c->e_evoff.as<int64_t>(),
                         c->e_qstart.as<int64_t>(), c->e_qoff.as<int64_t>(), c->d_queries.as<float>(), c->e_qev.as<uint64_t>(), query_size};
        hipLaunchKernelGGL(sfa::ev_pack_events_kernel, dim3(n), dim3(128), 0, st, pa);
        HIP_TRY(hipMemcpyAsync(query_events, c->e_qev.p, qe_bytes, hipMemcpyDeviceToHost, st));
    }
    KERNEL_TRY();
    HIP_TRY(hipEventRecord(c->eev[3], st));
    c->eev_pending = true;
    // the queries must be complete before align_device's uploads reuse the pinned staging area; same stream, in order
    if ((rc = align_device(c, c->d_queries.as<float>(), q_off.data(), n, c->d_out.as<ResultRow>()))) return rc;
    int32_t *b0 = c->h_small.as<int32_t>(), *b1 = b0 + n;
    float *b2 = reinterpret_cast<float *>(b1 + n);
    HIP_TRY(hipMemcpyAsync(c->h_out.p, c->d_out.p, sizeof(sfa_result_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b0, c->e_b0.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b1, c->e_b1.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b2, c->e_b2.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(rows, c->h_out.p, sizeof(sfa_result_t) * (size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        info[i].start_raw_idx = static_cast<uint64_t>(b0[i]);
        info[i].end_raw_idx = static_cast<uint64_t>(static_cast<float>(static_cast<uint64_t>(b1[i])) + b2[i]);  // u64 + float, as in C
    }
    return resolve_profile(c);
}

// records per wave of the device inflate: about one wave per SIMD (blow5_inflate_kernel)
static int inflate_lanes(int32_t n, int cu_count) {
    const int64_t simds = static_cast<int64_t>(cu_count) * 4;
    int lanes = 1;
    while (lanes < sfa::kInfMaxLanes && static_cast<int64_t>(n) > simds * lanes) lanes *= 2;
    return lanes;
}

// BLOW5 records in, result rows out: records are decompressed and parsed on the device (blow5_kernels.hpp), then the path of
// sfa_align_raw continues on the samples where they already are.
int sfa_align_blow5(sfa_ctx_t *c, const uint8_t *records, const int64_t *rec_off, int32_t n, int32_t record_zlib, int32_t signal_svb,
                    int32_t prefix_size, int32_t query_size, sfa_result_t *rows, sfa_query_info_t *info, sfa_read_head_t *heads,
                    sfa_event_t *query_events) {
    if (!c || n < 0 || (n > 0 && (!records || !rec_off || !rows || !info || !heads))) return fail(SFA_EINVAL, "sfa_align_blow5: bad argument");
    if (prefix_size < 0) return fail(SFA_EINVAL, "sfa_align_blow5: automatic query start (-p -1) needs the host stages");
    if (query_size <= 0) return fail(SFA_EINVAL, "sfa_align_blow5: query_size must be positive");
    if (n == 0) return SFA_OK;
    if (!c->shards.empty()) {
        std::vector<int32_t> lo;
        shard_ranges(n, c->shards.size(), &lo);
        return for_each_shard(c, [&](size_t r) {
            const int32_t a = lo[r], b = lo[r + 1];
            if (a == b) return static_cast<int>(SFA_OK);
            std::vector<int64_t> off(b - a + 1);
            for (int32_t i = a; i <= b; ++i) off[i - a] = rec_off[i] - rec_off[a];
            return sfa_align_blow5(c->shards[r], records + rec_off[a], off.data(), b - a, record_zlib, signal_svb, prefix_size, query_size,
                                   rows + a, info + a, heads + a,
                                   query_events ? query_events + static_cast<size_t>(a) * static_cast<size_t>(query_size) : nullptr);
        });
    }
    if (rec_off[0] != 0) return fail(SFA_EINVAL, "sfa_align_blow5: rec_off must start at 0");
    for (int32_t i = 0; i < n; ++i)
        if (rec_off[i + 1] < rec_off[i]) return fail(SFA_EINVAL, "sfa_align_blow5: rec_off not monotone at record %d", i);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipStream_t st = c->stream;
    const int64_t in_bytes = rec_off[n];
    int rc;
    // a compressed record inflates into a slot of 4x its size + 4 KB (svb-zd signals deflate by ~1.5x; a record that needs
    // more is handed to the host reader with the rest of the batch)
    std::vector<int64_t> slot(n + 1);
    slot[0] = 0;
    for (int32_t i = 0; i < n; ++i) slot[i + 1] = slot[i] + (record_zlib ? (((rec_off[i + 1] - rec_off[i]) * 4 + 4096 + 15) & ~int64_t(15)) : 0);
    if ((rc = c->b_in.reserve(static_cast<size_t>(in_bytes) + 128)) || (rc = c->b_inoff.reserve(8 * static_cast<size_t>(n + 1))) ||
        (rc = c->b_head.reserve(static_cast<size_t>(n) * sfa::kBlow5HeadBytes)) || (rc = c->h_head.reserve(static_cast<size_t>(n) * sfa::kBlow5HeadBytes + 8 * static_cast<size_t>(n))) ||
        (rc = c->b_len.reserve(4 * static_cast<size_t>(n))) || (rc = c->b_bad.reserve(4 * static_cast<size_t>(n))))
        return rc;
    if (record_zlib && ((rc = c->b_out.reserve(static_cast<size_t>(slot[n]) + 64)) || (rc = c->b_outoff.reserve(8 * static_cast<size_t>(n + 1))))) return rc;
    hipStream_t sp = c->stream_pre ? c->stream_pre : st;  // (every section below ends with a host wait on it)
    HIP_TRY(hipMemcpyAsync(c->b_in.p, records, static_cast<size_t>(in_bytes), hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemcpyAsync(c->b_inoff.p, rec_off, 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemsetAsync(c->b_bad.p, 0, 4 * static_cast<size_t>(n), sp));
    HIP_TRY(hipEventRecord(c->bev[0], sp));
    sfa::FieldsArgs fa{};
    if (record_zlib) {
        HIP_TRY(hipMemcpyAsync(c->b_outoff.p, slot.data(), 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, sp));
        sfa::InflateArgs ia{c->b_in.as<uint8_t>(), c->b_inoff.as<int64_t>(), c->b_out.as<uint8_t>(), c->b_outoff.as<int64_t>(), c->b_len.as<int32_t>(), n};
        const int lanes = inflate_lanes(n, c->pre_cus ? c->pre_cus : c->cu_count);
        hipLaunchKernelGGL(sfa::blow5_inflate_kernel, dim3((n + lanes - 1) / lanes), dim3(64), sizeof(sfa::InflateLds) * lanes, sp, ia, lanes);
        KERNEL_TRY();
        fa.payload = c->b_out.as<uint8_t>();
        fa.payload_off = c->b_outoff.as<int64_t>();
        fa.payload_len = c->b_len.as<int32_t>();
    } else {
        fa.payload = c->b_in.as<uint8_t>();
        fa.payload_off = c->b_inoff.as<int64_t>();
        fa.payload_len = nullptr;
    }
    fa.head = c->b_head.as<uint8_t>();
    fa.signal_svb = signal_svb ? 1 : 0;
    fa.n = n;
    hipLaunchKernelGGL(sfa::blow5_fields_kernel, dim3((n + 63) / 64), dim3(64), 0, sp, fa);
    KERNEL_TRY();
    uint8_t *hh = c->h_head.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(hh, c->b_head.p, static_cast<size_t>(n) * sfa::kBlow5HeadBytes, hipMemcpyDeviceToHost, sp));
    HIP_TRY(hipStreamSynchronize(sp));
    // the fields of every record; anything the device declined sends the whole batch to the host reader
    std::vector<int64_t> raw_off(n + 1);
    std::vector<double> scaling(3 * static_cast<size_t>(n));
    raw_off[0] = 0;
    bool fallback = false;
    for (int32_t i = 0; i < n && !fallback; ++i) {
        const uint8_t *h = hh + static_cast<size_t>(i) * sfa::kBlow5HeadBytes;
        int32_t status, id_len;
        int64_t ns;
        memcpy(&status, h, 4);
        memcpy(&id_len, h + 4, 4);
        memcpy(&ns, h + 8, 8);
        if (status != 0 || id_len < 0 || id_len > static_cast<int32_t>(sizeof(heads[i].read_id)) - 1 || ns < 0) {
            (void)fail(SFA_OK, "sfa_align_blow5: device declined record %d (status %d, id of %d bytes, %lld samples): host reader takes the batch", i,
                       status, id_len, static_cast<long long>(ns));  // kept in sfa_last_error() for whoever wants to know why
            fallback = true;
            break;
        }
        memcpy(heads[i].read_id, h + 56, id_len);
        heads[i].read_id[id_len] = 0;
        heads[i].id_len = id_len;
        heads[i].n_samples = ns;
        memcpy(&heads[i].digitisation, h + 16, 8);
        memcpy(&heads[i].offset, h + 24, 8);
        memcpy(&heads[i].range, h + 32, 8);
        heads[i].record_bytes = rec_off[i + 1] - rec_off[i];
        scaling[3 * i] = heads[i].digitisation;
        scaling[3 * i + 1] = heads[i].offset;
        scaling[3 * i + 2] = heads[i].range;
        raw_off[i + 1] = raw_off[i] + ns;
    }
    if (!fallback) {
        const int64_t total = raw_off[n];
        if ((rc = c->e_raw.reserve(2 * static_cast<size_t>(std::max<int64_t>(total, 1)))) || (rc = c->e_rawoff.reserve(8 * static_cast<size_t>(n + 1)))) return rc;
        HIP_TRY(hipMemcpyAsync(c->e_rawoff.p, raw_off.data(), 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, sp));
        sfa::SvbArgs sa{fa.payload, fa.payload_off, c->b_head.as<uint8_t>(), c->e_rawoff.as<int64_t>(), c->e_raw.as<int16_t>(), c->b_bad.as<int32_t>(),
                        signal_svb ? 1 : 0, n};
        hipLaunchKernelGGL(sfa::blow5_svb_kernel, dim3((n + 3) / 4), dim3(256), 0, sp, sa);
        KERNEL_TRY();
        int32_t *bad = reinterpret_cast<int32_t *>(hh + static_cast<size_t>(n) * sfa::kBlow5HeadBytes);
        HIP_TRY(hipMemcpyAsync(bad, c->b_bad.p, 4 * static_cast<size_t>(n), hipMemcpyDeviceToHost, sp));
        HIP_TRY(hipEventRecord(c->bev[1], sp));
        HIP_TRY(hipStreamSynchronize(sp));
        for (int32_t i = 0; i < n && !fallback; ++i)
            if (bad[i] != 0) {
                (void)fail(SFA_OK, "sfa_align_blow5: signal of record %d is shorter than its keys say: host reader takes the batch", i);
                fallback = true;
            }
        if (!fallback) {
            c->bev_pending = true;
            return align_raw_impl(c, nullptr, raw_off.data(), scaling.data(), n, prefix_size, query_size, rows, info, query_events);
        }
    }
    // host reader for the whole batch (own inflate / zlib, SSSE3 StreamVByte): malformed records are reported from there
    c->blow5_fallbacks++;
    std::vector<int16_t> raw;
    raw_off[0] = 0;
    for (int32_t i = 0; i < n; ++i) {
        sfa::Blow5Record rec;
        std::string err;
        if (!sfa::parse_blow5_record(records + rec_off[i], static_cast<size_t>(rec_off[i + 1] - rec_off[i]), record_zlib, signal_svb, &rec, &err))
            return fail(SFA_EINVAL, "sfa_align_blow5: record %d: %s", i, err.c_str());
        if (rec.read_id.size() > sizeof(heads[i].read_id) - 1) return fail(SFA_ERANGE, "sfa_align_blow5: record %d: read id of %zu bytes", i, rec.read_id.size());
        memcpy(heads[i].read_id, rec.read_id.c_str(), rec.read_id.size() + 1);
        heads[i].id_len = static_cast<int32_t>(rec.read_id.size());
        heads[i].n_samples = static_cast<int64_t>(rec.raw.size());
        heads[i].digitisation = rec.digitisation;
        heads[i].offset = rec.offset;
        heads[i].range = rec.range;
        heads[i].record_bytes = rec_off[i + 1] - rec_off[i];
        scaling[3 * i] = rec.digitisation;
        scaling[3 * i + 1] = rec.offset;
        scaling[3 * i + 2] = rec.range;
        raw.insert(raw.end(), rec.raw.begin(), rec.raw.end());
        raw_off[i + 1] = static_cast<int64_t>(raw.size());
    }
    if (raw.empty()) raw.push_back(0);
    return align_raw_impl(c, raw.data(), raw_off.data(), scaling.data(), n, prefix_size, query_size, rows, info, query_events);
}

// (testing hook of the device-side inflate alone: n zlib streams in, their bytes out; see include/sigfish_amd.h)
int sfa_inflate_zlib_device(sfa_ctx_t *c, const uint8_t *in, const int64_t *in_off, int32_t n, uint8_t *out, const int64_t *out_off, int32_t *out_len) {
    if (!c || !c->shards.empty() || n < 0 || (n > 0 && (!in || !in_off || !out || !out_off || !out_len)))
        return fail(SFA_EINVAL, "sfa_inflate_zlib_device: bad argument (single-device context needed)");
    if (n == 0) return SFA_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipStream_t st = c->stream;
    int rc;
    if ((rc = c->b_in.reserve(static_cast<size_t>(in_off[n]) + 128)) || (rc = c->b_inoff.reserve(8 * static_cast<size_t>(n + 1))) ||
        (rc = c->b_out.reserve(static_cast<size_t>(out_off[n]) + 64)) || (rc = c->b_outoff.reserve(8 * static_cast<size_t>(n + 1))) ||
        (rc = c->b_len.reserve(4 * static_cast<size_t>(n))))
        return rc;
    HIP_TRY(hipMemcpyAsync(c->b_in.p, in, static_cast<size_t>(in_off[n]), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->b_inoff.p, in_off, 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->b_outoff.p, out_off, 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, st));
    sfa::InflateArgs ia{c->b_in.as<uint8_t>(), c->b_inoff.as<int64_t>(), c->b_out.as<uint8_t>(), c->b_outoff.as<int64_t>(), c->b_len.as<int32_t>(), n};
    const int lanes = inflate_lanes(n, c->cu_count);
    hipLaunchKernelGGL(sfa::blow5_inflate_kernel, dim3((n + lanes - 1) / lanes), dim3(64), sizeof(sfa::InflateLds) * lanes, st, ia, lanes);
    KERNEL_TRY();
    HIP_TRY(hipMemcpyAsync(out, c->b_out.p, static_cast<size_t>(out_off[n]), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_len, c->b_len.p, 4 * static_cast<size_t>(n), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SFA_OK;
}

int sfa_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!free_bytes || !total_bytes) return fail(SFA_EINVAL, "sfa_device_memory: null argument");
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return SFA_OK;
}

void *sfa_pinned_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {  // usable from every device's context
        fail(SFA_ENOMEM, "hipHostMalloc(%zu bytes) failed", bytes);
        return nullptr;
    }
    return p;
}

void sfa_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int sfa_sync(sfa_ctx_t *c) {
    if (!c) return fail(SFA_EINVAL, "null context");
    if (!c->shards.empty()) {
        for (sfa_ctx *sh : c->shards)
            if (int rc = sfa_sync(sh)) return rc;
        return SFA_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return resolve_profile(c);
}

int sfa_get_profile(sfa_ctx_t *c, sfa_profile_t *p) {
    if (!c || !p) return fail(SFA_EINVAL, "null argument");
    if (!c->shards.empty()) {  // the shards run side by side: times are the slowest shard's, counts add up
        sfa_profile_t sum{};
        for (sfa_ctx *sh : c->shards) {
            sfa_profile_t q;
            if (int rc = sfa_get_profile(sh, &q)) return rc;
            sum.fill_ms = std::max(sum.fill_ms, q.fill_ms);
            sum.trace_ms = std::max(sum.trace_ms, q.trace_ms);
            sum.finalize_ms = std::max(sum.finalize_ms, q.finalize_ms);
            sum.total_ms = std::max(sum.total_ms, q.total_ms);
            sum.events_ms = std::max(sum.events_ms, q.events_ms);
            sum.decode_ms = std::max(sum.decode_ms, q.decode_ms);
            sum.blow5_fallbacks += q.blow5_fallbacks;
            sum.normalise_ms = std::max(sum.normalise_ms, q.normalise_ms);
            sum.cells += q.cells;
            sum.fill_launches += q.fill_launches;
            sum.n_tasks += q.n_tasks;
            sum.ckpt_bytes += q.ckpt_bytes;
            sum.segment_reruns += q.segment_reruns;
            sum.non_finite_reads += q.non_finite_reads;
            sum.ckpt_interval = std::max(sum.ckpt_interval, q.ckpt_interval);
            sum.n_chunks = std::max(sum.n_chunks, q.n_chunks);
            sum.n_segments = std::max(sum.n_segments, q.n_segments);
            sum.lds_ckpt = std::max(sum.lds_ckpt, q.lds_ckpt);
            sum.trace_margin = std::max(sum.trace_margin, q.trace_margin);
        }
        *p = sum;
        return SFA_OK;
    }
    if (int rc = resolve_profile(c)) return rc;
    *p = c->prof;
    return SFA_OK;
}

void *sfa_stream(sfa_ctx_t *c) { return (c && c->shards.empty()) ? static_cast<void *>(c->stream) : nullptr; }

int sfa_n_devices(sfa_ctx_t *c) { return !c ? 0 : (c->shards.empty() ? 1 : static_cast<int>(c->shards.size())); }

#ifdef SFA_TASK_TIMES
// measurement builds only (tools/task_times.py): [task][3] = start tick, end tick (100 MHz), SIMD position of the last fill
int64_t sfa_debug_task_times(sfa_ctx_t *c, unsigned long long *out, int64_t cap_tasks) {
    if (!c || !out) return -1;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    const int64_t n = std::min<int64_t>(cap_tasks, c->n_times);
    if (n > 0 && hipMemcpy(out, c->d_times.p, 24 * static_cast<size_t>(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
int64_t sfa_debug_task_times_long(sfa_ctx_t *c, unsigned long long *out, int64_t cap_tasks) {
    if (!c || !out) return -1;
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    const int64_t n = std::min<int64_t>(cap_tasks, c->n_ltimes);
    if (n > 0 && hipMemcpy(out, c->d_ltimes.p, 24 * static_cast<size_t>(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#endif

}  // extern "C"
