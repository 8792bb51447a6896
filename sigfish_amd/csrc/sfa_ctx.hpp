// sfa_ctx.hpp -- what the units of the C-ABI share (include/sigfish_amd.h): error reporting, grow-only device / page-locked
// buffers, the context, and the few internal functions that cross unit boundaries.
//   sfa_context.hip  contexts: init / destroy, devices, options, profile, small utilities
//   sfa_align.hip    the alignment stage: planner -> launches (wave kernels, row strips) -> rows; the batch entry points
//   sfa_pre.hip      the stages in front of it on the device: raw samples / BLOW5 records in (events_kernels.hpp, blow5_kernels.hpp)
// There is NO CPU fallback: every failure is reported through the return code + sfa_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sigfish_amd.h"
#include "sfa_plan.hpp"

namespace sfa {
std::string &last_error_slot();  // this thread's sfa_last_error() text (sfa_context.hip)
}

namespace {

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    sfa::last_error_slot() = buf;
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
                err_ = rc ? sfa::last_error_slot() : std::string();
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
    hipStream_t stream_long2 = nullptr;      // ... their groups alternating between two streams (pass 2 of one under pass 1 of the next)
    hipEvent_t lev[4] = {nullptr, nullptr, nullptr, nullptr};  // inputs of the batch ready on `stream` / strips done on `stream_long` / fork and join of `stream_long2`
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // fill start/end, finalize1 end, trace end, end, row strips start
    hipEvent_t eev[4] = {nullptr, nullptr, nullptr, nullptr};  // sfa_align_raw: event detection start/end, normalisation start/end
    bool eev_pending = false;
    int cu_count = 256;

    // tunables (sfa_set_option)
    int64_t opt_ckpt_interval = 0;           // force the checkpoint interval (power of two >= 4); 0 = auto
    int64_t opt_ckpt_budget = 32ll << 30;    // bytes of HBM the checkpoints of one batch may take
    int64_t opt_ev_parallel = 3;             // sfa_align_raw: bit 0 wave-per-read prefix sums where they are provably exact, bit 1 wave-per-read peak picker where its result is certified
    int64_t opt_min_slice_reads = 65536;     // a batch is only cut into slices of at least this many reads
    int64_t opt_segment_warm = 4;            // query lengths of warm-up in front of a segment
    int64_t opt_column_segments = 0;         // 0 = auto, 1 = off, N = segments per job for small batches (sweep_segment)
    int64_t opt_lane_widening = 0;           // 0 = by batch size; 1, 2, 4 = fixed (rows per lane / w, lanes per read * w)
    int64_t opt_widen_below = 5;             // auto: widen (x4) when the batch has fewer waves per SIMD than this
    int64_t opt_trace_margin = -1;           // steps of head start for pass 2; -1 = qlen_max + 16
    int64_t opt_waves_per_simd = 6;          // target occupancy used when chunking the job list
    int64_t opt_fused_trace = 1;             // 1: pass 2 rides in the fill launch as trailing tickets (fills the drain) where the launch has more tasks than wave slots; 2: always; 0: own launch
    int64_t opt_lds_ckpt = 1;                // 1: rolling checkpoints in LDS where the batch's shapes allow (R <= 16, sDTW); 0: all snapshots to HBM
    int64_t opt_prio_unit = 2048;            // longest-remaining-first issue priority of the fill: columns per level, 0 = off
    int32_t span_sixteenths = 0;             // pass 2's head start on the HBM-snapshot route follows the spans of the previous batch's alignments: sixteenths of the query length (0: not known yet -> a whole query length)
    int64_t quad_limit_ms = 0, strip_limit_ms = 0;  // the limits the last batch's waits actually ran with (floors applied), for the error messages
    int64_t opt_spin_limit_ms = 20000;       // bound of every in-launch wait (fused pass 2, pipelined strips); beyond it the batch fails with SFA_EKERNEL
    // test hooks (sfa_set_option refuses them unless SFA_TEST_HOOKS=1 is in the environment)
    int64_t opt_debug_drop_quad = -1;        // the fill tasks of this quad never signal completion
    int64_t opt_debug_drop_strip = -1;       // strip 0 of this long read (job 0) never publishes its progress

    // reference model (immutable after init)
    int32_t num_ref = 0, n_jobs = 0;
    int64_t total_cols = 0;  // sum over jobs of rlen
    std::vector<int32_t> h_job_len;
    DevBuf d_ref, d_job_off, d_job_len, d_job_contig, d_job_strand, d_ref_len, d_ref_off;

    // per-batch scratch
    DevBuf d_verify, d_segfail;
    DevBuf d_lprog, d_lticket;  // pipelined strips: progress counters, ticket
    DevBuf d_bndc, d_long, d_lbest, d_lsecond, d_lend, d_lwin, d_lck;  // row strips (queries beyond SFA_MAX_QUERY, sdtw_strips.hpp)
    PinBuf h_long;
    bool long_pending = false;
    int64_t seg_reruns = 0;  // batches walked again because a segment hand-over did not verify
    DevBuf d_queries, d_stage, d_pbest, d_pend, d_pjob, d_psecond, d_wjob, d_wend, d_wscore, d_tst, d_ck, d_out;
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

// ---- internal functions that cross unit boundaries ----
namespace sfa {
int resolve_profile(sfa_ctx *c);  // sfa_align.hip: timers + error words of the batch submitted last (waits for it)
// core of every align entry point: queries already in HBM, results left in HBM (sfa_align.hip)
int align_device(sfa_ctx *c, const float *d_queries, const int64_t *q_off, int32_t n, struct ResultRow *d_out);
// contiguous read range of shard r: [r*n/G, (r+1)*n/G)
inline void shard_ranges(int32_t n, size_t g, std::vector<int32_t> *lo) {
    lo->resize(g + 1);
    for (size_t r = 0; r <= g; ++r) (*lo)[r] = static_cast<int32_t>(static_cast<int64_t>(n) * static_cast<int64_t>(r) / static_cast<int64_t>(g));
}
// run fn(shard index) for every shard, each on the shard's own host thread (HIP's current device and sfa_last_error are
// per thread), shard 0 on the caller's; the first failure's code and message become the caller's
template <typename F>
int for_each_shard(sfa_ctx *g, F fn) {
    const size_t G = g->shards.size();
    for (size_t r = 1; r < G; ++r) g->workers[r - 1]->post([&fn, r] { return fn(r); });
    int rc = fn(0);
    std::string msg = rc ? last_error_slot() : std::string();
    for (size_t r = 1; r < G; ++r) {  // (always all of them: fn and what it captures must outlive every worker's job)
        const auto res = g->workers[r - 1]->wait();
        if (res.first && !rc) {
            rc = res.first;
            msg = res.second;
        }
    }
    if (rc) last_error_slot() = msg;
    return rc;
}
}  // namespace sfa
