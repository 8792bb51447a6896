// sfa_plan.hpp -- host-side batch planning for the sDTW kernels (pure C++, no HIP; unit-tested without a GPU).
//
// Replaces the reference's per-batch thread fan-out (work_db / pthread_db, src/thread.c:74-132: contiguous
// read ranges per thread + work stealing) with a static plan for the GPU:
//   * reads are grouped four at a time into "quads" (one wavefront each); lengths select a rows-per-lane class R
//     (4/8/16/32) -- long classes first so that short work fills the tail; inside a class reads whose lengths agree
//     modulo R share waves (lengths descending; MixedQuad in sdtw_kernels.hpp), std_dtw keeps ONE length per wave; queries of 513..1024 / 1025..2048 events take 32 / 64 lanes per read, i.e. two reads / one read
//     per "quad";
//   * SMALL BATCHES: when the waves of a batch would not fill the chip, every class trades rows per lane for lanes
//     per read (R/2 x 2L or R/4 x 4L, "lane widening" 2 or 4): 2-4x the waves, each with a 2-4x shorter step;
//   * the (contig,strand) job list is cut into contiguous chunks of similar size when there are too few quads
//     to fill the chip; a wave-task is (quad, chunk);
//   * the checkpoint interval T of pass 1 is the smallest power of two >= 512 whose checkpoints fit the budget.
#pragma once

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace sfa {

constexpr int kMaxQuery = 2048;

struct ClassShape {
    int R, lanes;  // query rows per lane, lanes per read: the class holds queries of up to R*lanes events
};
// classes in task order (long first)
constexpr ClassShape kClassShapes[6] = {{32, 64}, {32, 32}, {32, 16}, {16, 16}, {8, 16}, {4, 16}};

// shape of base class ci under lane widening w (1, 2, 4): rows per lane halve while lanes per read double
inline ClassShape widened(int ci, int w) {
    ClassShape s = kClassShapes[ci];
    while (w > 1 && s.lanes < 64 && s.R > 4) {
        s.R /= 2;
        s.lanes *= 2;
        w /= 2;
    }
    return s;
}

inline int class_for(int qlen) {  // index into kClassShapes, -1: too long
    if (qlen > kMaxQuery) return -1;
    int c = 0;
    while (c + 1 < 6 && qlen <= kClassShapes[c + 1].R * kClassShapes[c + 1].lanes) ++c;
    return c;
}

// checkpoints the fill stores for a job of rlen columns at interval 1<<shift: k*T <= rlen-4 (a block boundary at or
// after k*T then always exists before the sweep ends); must match sweep_job() / trace_body() in sdtw_kernels.hpp
inline int64_t ck_count(int32_t rlen, int shift) { return (rlen > 4 ? rlen - 4 : 0) >> shift; }

struct PlanParams {
    int64_t n_sims = 1024;          // SIMDs on the device (CUs * 4)
    int64_t waves_per_simd = 6;     // occupancy to aim for when chunking
    int64_t ckpt_interval = 0;      // 0 = auto
    int64_t ckpt_budget_bytes = 32ll << 30;
    int64_t trace_margin = -1;      // -1 = longest query + lanes per read (16 up to 512 events)
    int32_t span_sixteenths = 0;    // > 0 (with trace_margin < 0, snapshots in HBM): the head start of pass 2 is this many sixteenths of the
                                    // longest query (+ lanes + 16) instead of the whole query -- what the previous batch's alignments spanned
    int64_t lane_widening = 0;      // 0 = auto (by batch size), else 1, 2 or 4
    int64_t widen_below = 5;        // auto: widen (x4) when the batch has fewer than this many waves per SIMD
    int64_t column_segments = 0;    // 0 = auto (small batches of the 64-lane shapes), 1 = off, N = N segments per job
    int64_t segment_warm_windows = 4;  // windows (query lengths) a segment starts before its own first one
    bool allow_segments = true;     // false for std_dtw (its first row is cumulative: no finite memory)
    int lds_ckpt = 0;               // 1: rolling checkpoints in LDS + sparse HBM checkpoints (sdtw_kernels.hpp, LdsCkpt) where the shapes allow (queries up to
                                    // 256 events: the measured regime) and the batch size suits; 2: wherever the shapes allow (up to 1024 events at 16 rows per lane)
    bool std_dtw = false;           // --dtw-std: with lds_ckpt the fill keeps NO LDS snapshots, only the sparse HBM store (the margin is not capped)
    bool skip_long = false;         // true: reads of more than kMaxQuery events are left out (the caller runs them in row strips, sdtw_strips.hpp)
};

struct PlanClass {
    int R = 0, lanes = 16, quad_base = 0, n_quads = 0;
    int64_t ck_base = 0;  // float offset
};

struct BatchPlan {
    int32_t n_quads = 0, n_chunks = 1, max_R = 4, max_lanes = 16, widening = 1, ck_shift = 0, trace_margin = 0;
    int32_t lck_shift = 9;  // with lds_ckpt: log2 of the interval between two LDS snapshots (9, 10, 11 for queries up to 256, 512, 1024 events)
    int32_t n_seg = 1, warm_windows = 4;  // column segments per job (sdtw_kernels.hpp, sweep_segment)
    bool lds_ckpt = false;  // the fill keeps its snapshots in LDS; ck_shift is then the interval of the sparse HBM store
    int64_t ck_floats = 0, query_events = 0;
    std::vector<int32_t> order;         // [4*max(n_quads,1)] read per (quad,slot) or -1
    std::vector<int32_t> quad_qlen;     // [max(n_quads,1)]
    std::vector<int32_t> slot_of_read;  // [n] quad*4+slot or -1 (skipped read)
    std::vector<int32_t> chunk_begin;   // [n_chunks+1]
    std::vector<int32_t> job_ck_off;    // [n_jobs+1]
    std::vector<PlanClass> classes;
    // scratch of plan_batch(), kept with the plan: a caller that reuses one BatchPlan for every batch (the library does)
    // pays no allocation in the steady state -- the vectors of a 100 000-read plan are 400 KB each, i.e. one mmap / page-fault
    // round per vector and call otherwise (1.35 -> 0.75 ms per plan)
    std::vector<int32_t> s_qlen, s_count, s_fill_pos, s_slot_start, s_len_quad_base;
    std::vector<int8_t> s_per_shift;
    void reset() {  // scalars back to their defaults; vectors keep their capacity
        n_quads = 0; n_chunks = 1; max_R = 4; max_lanes = 16; widening = 1; ck_shift = 0; trace_margin = 0; lck_shift = 9;
        n_seg = 1; warm_windows = 4;
        lds_ckpt = false;
        ck_floats = 0; query_events = 0;
        classes.clear();
    }
};

// Split the job list into n_chunks contiguous, non-empty ranges of roughly equal reference columns.
inline void split_jobs(const std::vector<int32_t> &job_len, int64_t total_cols, int32_t n_chunks, int32_t *chunk_begin) {
    const int32_t n_jobs = static_cast<int32_t>(job_len.size());
    int64_t acc = 0;
    int32_t j = 0;
    chunk_begin[0] = 0;
    for (int32_t ch = 1; ch < n_chunks; ++ch) {
        const int64_t want = total_cols * ch / n_chunks;
        acc += job_len[j++];  // at least one job per chunk
        while (j < n_jobs - (n_chunks - ch) && acc + job_len[j] / 2 < want) acc += job_len[j++];
        chunk_begin[ch] = j;
    }
    chunk_begin[n_chunks] = n_jobs;
}

inline int plan_batch(const int64_t *q_off, int32_t n, const std::vector<int32_t> &job_len, int64_t total_cols,
                      const PlanParams &pp, BatchPlan *out, std::string *err) {
    BatchPlan &p = *out;
    p.reset();
    const int32_t n_jobs = static_cast<int32_t>(job_len.size());
    std::vector<int32_t> &qlen = p.s_qlen, &count = p.s_count, &fill_pos = p.s_fill_pos;
    qlen.assign(n, 0);
    count.assign(kMaxQuery + 2, 0);
    int maxq = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int64_t l = q_off[i + 1] - q_off[i];
        if (l < 0) {
            *err = "q_off is not monotone at read " + std::to_string(i);
            return -1;  // SFA_EINVAL
        }
        if (l > kMaxQuery && pp.skip_long) continue;  // qlen stays 0: planned like a read without events
        if (l > kMaxQuery) {
            *err = "read " + std::to_string(i) + " has " + std::to_string(l) + " events; the limit is " + std::to_string(kMaxQuery);
            return -4;  // SFA_ERANGE
        }
        qlen[i] = static_cast<int32_t>(l);
        count[l]++;
        maxq = std::max(maxq, qlen[i]);
        p.query_events += l;
    }
    // classes in task order (long first); inside a class by descending length.  layout(w) fills the plan for lane
    // widening w and returns the number of quads (waves' worth of reads).
    // (Queries of 257 .. 2048 events keep their 32-row base shapes in large batches: their snapshots -- 33 planes -- do not fit LDS
    // twice at four waves per SIMD, so they go to HBM and pass 2 rides in the fill launch by ticket.  Running them as 16 rows x 32 /
    // 64 lanes on the LDS route instead measured 1-2 % slower: LABNOTES.md.)
    // MIXED QUADS (sdtw_kernels.hpp, MixedQuad): reads of different lengths share a wave when their lengths agree modulo the
    // rows per lane R of their class -- every read's last query row then falls into the same lane and register.  A class is laid
    // out residue by residue, lengths descending inside a residue, a new wave only where a residue ends: a ragged batch with a few
    // hundred distinct lengths leaves at most R partly filled waves per class instead of one per length.  Not for std_dtw
    // (its kernels keep one length per wave) nor at lane widening 4 (the column segments align on ONE query length).
    auto mixing = [&](int w) { return !pp.std_dtw && w < 4; };
    std::vector<int32_t> &slot_start = p.s_slot_start, &len_quad_base = p.s_len_quad_base;
    slot_start.assign(maxq + 2, 0);
    len_quad_base.assign(maxq + 2, 0);
    auto layout = [&](int w) {
        p.classes.clear();
        int32_t n_quads = 0;
        const bool mix = mixing(w);
        for (int ci = 0; ci < 6; ++ci) {
            const ClassShape sh = widened(ci, w);
            PlanClass cl;
            cl.R = sh.R;
            cl.lanes = sh.lanes;
            cl.quad_base = n_quads;
            const int per = 64 / cl.lanes;  // reads per wave
            int32_t pos = 0;                // reads placed in this class so far (slot index inside the class)
            const int n_res = mix ? cl.R : 1;
            for (int res = 0; res < n_res; ++res) {
                for (int l = maxq; l >= 1; --l) {
                    if (count[l] == 0 || class_for(l) != ci || (mix && l % cl.R != res)) continue;
                    slot_start[l] = pos;
                    len_quad_base[l] = cl.quad_base;
                    pos += count[l];
                    if (!mix) pos = (pos + per - 1) / per * per;  // one length per wave
                }
                pos = (pos + per - 1) / per * per;  // a new residue starts a new wave
            }
            n_quads += pos / per;
            cl.n_quads = n_quads - cl.quad_base;
            if (cl.n_quads > 0) p.classes.push_back(cl);
        }
        return n_quads;
    };
    auto chunks_for = [&](int32_t n_quads) -> int32_t {  // chunk the job list only as far as needed to fill the machine
        if (n_quads <= 0 || n_jobs <= 1) return 1;
        // equal-length tasks finish in rounds; ask for >= 16 rounds so the last, partly filled one costs ~3 %
        const int64_t target = pp.n_sims * pp.waves_per_simd * 16;
        return static_cast<int32_t>(std::min<int64_t>(n_jobs, std::max<int64_t>(1, (target + n_quads - 1) / n_quads)));
    };
    int w = 1;
    int32_t n_quads = layout(1);
    if (pp.lane_widening > 0) {
        w = static_cast<int>(pp.lane_widening);
        if (w != 1) n_quads = layout(w);
    } else {
        // measured (nCoV, q = 250, profiles/r01_logs/small_batches_widening_crossover_head.log; t = wave-tasks per SIMD with the
        // 16-lane shapes = reads / 2048 here): the 4-rows-per-lane shapes win up to t ~ 0.9 (1 024 reads: 4.2 / 2.8 / 1.7 ms at
        // x1 / x2 / x4), the 8-rows shapes from there to t ~ 3.9 (4 096 reads: 6.5 / 5.1 / 5.5 ms; 6 144: 8.5 / 7.1 / 7.7), the
        // 16-lane shapes from t = 4 on (8 192 reads: 8.8 / 9.1 / 9.9 ms).  widen_below = 5 is the default; a caller with several
        // batches in flight per device lowers it and all three ranges shrink with it.
        const int64_t tasks1 = static_cast<int64_t>(n_quads) * chunks_for(n_quads);
        if (tasks1 * 5 < pp.widen_below * pp.n_sims * 4) {                         // t < 0.8 widen_below
            w = (tasks1 * 50 < pp.widen_below * pp.n_sims * 9) ? 4 : 2;            // t < 0.18 widen_below
            n_quads = layout(w);
        }
    }
    p.widening = w;
    p.n_quads = n_quads;
    p.max_R = 4;
    p.max_lanes = 16;
    for (const PlanClass &cl : p.classes) {
        p.max_R = std::max(p.max_R, cl.R);
        p.max_lanes = std::max(p.max_lanes, cl.lanes);
    }
    p.order.assign(4 * static_cast<size_t>(std::max(n_quads, 1)), -1);
    p.quad_qlen.assign(std::max(n_quads, 1), 0);
    p.slot_of_read.assign(n, -1);
    fill_pos.assign(maxq + 2, 0);
    std::vector<int8_t> &per_shift = p.s_per_shift;
    per_shift.assign(maxq + 2, 0);  // log2(reads per wave) of every query length that occurs (4, 2 or 1 reads)
    for (int l = 1; l <= maxq; ++l)
        if (count[l]) {
            const int per = 64 / widened(class_for(l), w).lanes;
            per_shift[l] = per == 4 ? 2 : (per == 2 ? 1 : 0);
        }
    for (int32_t i = 0; i < n; ++i) {
        const int l = qlen[i];
        if (l == 0) continue;
        const int32_t k = slot_start[l] + fill_pos[l]++;  // position inside the class
        const int sh = per_shift[l];
        const int32_t qd = len_quad_base[l] + (k >> sh);
        const int32_t sl = qd * 4 + (k & ((1 << sh) - 1));  // a wave always has four slots; wide shapes use 2 / 1
        p.order[sl] = i;
        p.slot_of_read[i] = sl;
        p.quad_qlen[qd] = std::max(p.quad_qlen[qd], l);  // the wave's longest read: where everybody's last query row sits
    }
    if (n_quads == 0) p.quad_qlen[0] = 1;

    // column segments: when even one wave per (read, job) leaves the chip idle, every job is cut into segments that
    // start from a guessed state a few windows early and are verified against their predecessor (sweep_segment)
    if (pp.allow_segments && p.widening == 4 && p.max_lanes == 64 && n_quads > 0) {
        int32_t min_len = n_jobs ? job_len[0] : 0;
        for (int32_t j = 1; j < n_jobs; ++j) min_len = std::min(min_len, job_len[j]);
        const int64_t tasks = static_cast<int64_t>(n_quads) * n_jobs;
        int64_t S = pp.column_segments > 0 ? pp.column_segments : ((pp.widen_below + 1) * pp.n_sims + tasks - 1) / tasks;  // a caller with several batches in flight lowers widen_below
        S = std::min<int64_t>(S, 64);
        S = std::min<int64_t>(S, min_len / (4 * std::max<int64_t>(pp.segment_warm_windows, 1) * std::max(maxq, 1)));  // a segment at least four warm-ups long
        // measured (nCoV, q = 250): 64 reads 1.94 -> 0.58 ms, 512 reads 2.0 -> 1.0 ms, 1 024 reads 2.2 -> 1.8 ms per batch; with
        // two segments the warm-up costs more than the shorter chain saves
        if (S >= 3 || (pp.column_segments > 1 && S >= 2)) p.n_seg = static_cast<int32_t>(S);
        p.warm_windows = static_cast<int32_t>(pp.segment_warm_windows);
    }
    if (p.n_seg > 1) {  // chunk = (job, segment)
        p.n_chunks = n_jobs * p.n_seg;
        p.chunk_begin.resize(p.n_chunks + 1);
        for (int32_t c = 0; c < p.n_chunks; ++c) p.chunk_begin[c] = c / p.n_seg;  // (the segmented kernels do not read it)
        p.chunk_begin[p.n_chunks] = n_jobs;
    } else {
        // chunk the job list only as far as needed to fill the machine
        p.n_chunks = chunks_for(n_quads);
        p.chunk_begin.resize(p.n_chunks + 1);
        split_jobs(job_len, total_cols, p.n_chunks, p.chunk_begin.data());
    }

    // checkpoint interval
    p.job_ck_off.assign(n_jobs + 1, 0);
    p.trace_margin = static_cast<int32_t>(pp.trace_margin >= 0 ? pp.trace_margin : maxq + p.max_lanes);
    const bool adapt = pp.trace_margin < 0 && pp.span_sixteenths > 0;  // (applied below, once the checkpoint route is known)
    // LDS checkpoints: every shape of the batch must hold its state in 17 planes (R <= 16), one sweep per (quad, job); the
    // margin is capped so that the snapshot pass 2 wants for a window is one of the last two (LdsCkpt::save):
    // interval (512 / 1024 / 2048 by the longest query) >= window length + margin + 3
    p.lds_ckpt = pp.lds_ckpt && pp.ckpt_interval == 0 && p.max_R <= 16 && p.n_seg == 1 && n_quads > 0 && maxq <= (pp.lds_ckpt >= 2 ? 1024 : 256);
    p.lck_shift = maxq <= 256 ? 9 : (maxq <= 512 ? 10 : 11);
    {   // the LDS buffers cap the fill at four waves per SIMD instead of six: a batch whose tasks are all resident at six
        // but not at four would need a second round (measured: 8 192 reads 7.35 -> 7.65 ms); everything else gains
        // (16 384 reads 13.7 -> 13.4 ms, 100 000 reads 74.6 -> 73.5 ms)
        const int64_t tasks = static_cast<int64_t>(n_quads) * p.n_chunks;
        if (pp.lds_ckpt < 2 && tasks > 4 * pp.n_sims && tasks <= 6 * pp.n_sims) p.lds_ckpt = false;
    }
    if (p.lds_ckpt && !pp.std_dtw) p.trace_margin = std::min<int32_t>(p.trace_margin, (1 << p.lck_shift) - maxq - 3);
    // Snapshots in HBM every 512 steps (the 32-row shapes): a head start that turns out too short costs one more attempt from the
    // snapshot before, so it can follow what alignments actually span (event detection over-segments: ~2/3 of a column per event)
    // instead of a whole query length.  Not on the LDS route: there a miss falls back to the sparse store or the strand's start.
    if (adapt && !p.lds_ckpt && !pp.std_dtw)
        p.trace_margin = std::min<int32_t>(p.trace_margin, static_cast<int32_t>((static_cast<int64_t>(maxq) * pp.span_sixteenths) / 16) + p.max_lanes + 16);
    if (n_quads > 0) {
        int shift = p.lds_ckpt ? 15 : 9;  // T = 512: measured optimum of fill (+checkpoint stores) against pass 2 (re-run length); sparse store: 32768
        if (pp.ckpt_interval > 0) {
            shift = 0;
            while ((1ll << shift) < pp.ckpt_interval) ++shift;
        }
        for (;; ++shift) {
            int64_t per_quad = 0;
            for (int32_t j = 0; j < n_jobs; ++j) per_quad += ck_count(job_len[j], shift);
            int64_t floats = 0;
            for (const PlanClass &cl : p.classes) floats += per_quad * cl.n_quads * (cl.R + 2) * 64;
            if (pp.ckpt_interval > 0 || floats * 4 <= pp.ckpt_budget_bytes || per_quad == 0 || shift >= 30) {
                p.ck_shift = shift;
                p.ck_floats = floats;
                break;
            }
        }
        int64_t base = 0, per_quad = 0;
        for (int32_t j = 0; j < n_jobs; ++j) {
            p.job_ck_off[j] = static_cast<int32_t>(per_quad);
            per_quad += ck_count(job_len[j], p.ck_shift);
        }
        p.job_ck_off[n_jobs] = static_cast<int32_t>(per_quad);
        for (PlanClass &cl : p.classes) {
            cl.ck_base = base;
            base += per_quad * cl.n_quads * (cl.R + 2) * 64;
        }
    }
    return 0;
}

}  // namespace sfa
