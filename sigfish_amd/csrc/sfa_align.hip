// sfa_align.hip -- the alignment stage behind the C-ABI (include/sigfish_amd.h): replaces the reference's accelerator hook
// align_db() (src/sigfish.c:1003-1015).  Host work: group reads into "quads" (reads of one length class share a wavefront;
// sfa_plan.hpp), pick the rows-per-lane class, size the checkpoint interval, launch fill (+ pass 2 by ticket, or fill ->
// finalize -> trace -> finalize), row strips for queries beyond 2048 events, and hand back one row per read in input order.
#include "sfa_ctx.hpp"
#define SFA_DEFINE_FINALIZE_KERNEL
#include "sdtw_kernels.hpp"
#include "sdtw_instances.hpp"
#include "sdtw_strips.hpp"

using sfa::DpArgs;
using sfa::FinalizeArgs;
using sfa::ResultRow;
using sfa::align_device;
using sfa::for_each_shard;
using sfa::resolve_profile;
using sfa::shard_ranges;

namespace {

static_assert(sizeof(ResultRow) == sizeof(sfa_result_t), "result row layout");

void launch_fill(int maxr, bool std_dtw, const DpArgs &a, hipStream_t st) {  // snapshots in HBM, pass 2 as its own launch
    const dim3 grid((a.n_tasks + 3) / 4), block(256);
    if (a.n_seg > 1) {  // column segments (subsequence DTW, small batches): the SEG kernels
        if (maxr >= 32)
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<32, false, true>), grid, block, 0, st, a);
        else if (maxr >= 16)
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<16, false, true>), grid, block, 0, st, a);
        else if (maxr >= 8)
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<8, false, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((sfa::sdtw_fill_kernel<4, false, true>), grid, block, 0, st, a);
        return;
    }
#define SFA_FILL(MR)                                                                     \
    if (std_dtw)                                                                         \
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<MR, true>), grid, block, 0, st, a);    \
    else                                                                                 \
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<MR, false>), grid, block, 0, st, a)
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
#define K_(MR) sfa::sdtw_fill_kernel<MR, false, false, true>
#define KS_(MR) sfa::sdtw_fill_kernel<MR, true, false, true>
    if (std_dtw)
        SFA_LCK_LAUNCH(KS_, a);
    else
        SFA_LCK_LAUNCH(K_, a);
#undef K_
#undef KS_
}

void launch_fill_fused(int maxr, bool std_dtw, const DpArgs &a, hipStream_t st) {  // fill tasks + one pass-2 ticket per quad: one wave per ticket
    const dim3 grid(static_cast<unsigned>((a.n_tasks + 3) / 4 + (a.n_quads_total + 3) / 4)), block(256);
    if (maxr >= 32) {  // the 32-row fill keeps its snapshots in HBM (write-through: its pass-2 waves read them in the same launch)
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<32, false, false, false, true>), grid, block, 0, st, a);
        return;
    }
#define K_(MR) sfa::sdtw_fill_kernel<MR, false, false, true, true>
#define KS_(MR) sfa::sdtw_fill_kernel<MR, true, false, true, true>
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

// Reads of more than SFA_MAX_QUERY events: row strips (sdtw_strips.hpp).  Pass 1, one wave per (read, job, strip), sweeps the
// query in strips of 64 x R rows, cost only, handing the last row of a strip to the next one through HBM; the strip finalize
// names each read's winning (job, cell, score); pass 2, one wave per read, traces the winning job strip by strip from the last
// one upwards with start-column tracking.  Runs on `st`: the context's second stream, beside the wave kernels of the batch; it
// writes the rows of these reads, which the wave kernels' finalize leaves alone.  Reads are taken in groups whose boundary rows
// and checkpoints fit the checkpoint budget -- and, when there are enough of them, in two groups on two streams with two sets of
// scratch: pass 2 of a group is one wave per read (0.8 waves per SIMD for 3 125 reads: bound by one wave's dependent chain, 12 ms
// at 8 000 events) and runs beside the other group's work instead of behind everything.  Measured on one box, ms per step at
// 3 000 / 8 000 events: one group 99.9 / 103.5, two 97.4 / 99.2, four 104.7 / 118.8 (more drains than they hide):
// profiles/r04_logs/strips_groups_ab*.log, rejected_strip_pass2_rows_in_lds_three_waves.log.
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
    int64_t per = 0;
    for (int32_t j = 0; j < n_jobs; ++j) {
        const int64_t row = (static_cast<int64_t>(c->h_job_len[j]) + sfa::kBndPad + 3) & ~int64_t(3);
        bnd_off[j] = per;
        per += row;
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
    // cost rows of every job, one per strip boundary (pass 1 writes them, pass 2 reads them) + the checkpoints
    const int64_t cost_rows = std::max<int64_t>(1, max_strips - 1);
    const int64_t bytes_per_read = per * cost_rows * 4 + ck_floats_per_read * 4;
    int64_t waves_total = 0;
    for (int32_t i = 0; i < n_long; ++i) waves_total += static_cast<int64_t>(n_strips_of[i]) * n_jobs;
    // two groups when each still has two rounds of the device's wave slots (16 per CU) to fill the chip with its pipeline of strips
    const int64_t want_groups = waves_total >= 4 * 16 * static_cast<int64_t>(c->cu_count) ? 2 : 1;
    const int64_t by_budget = c->opt_ckpt_budget / (2 * std::max<int64_t>(bytes_per_read, 1));  // (two sets of scratch)
    const int32_t group = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>((n_long + want_groups - 1) / want_groups, by_budget)));
    const bool two_sets = group < n_long;
    const size_t n_part = static_cast<size_t>(n_long) * n_jobs;
    const size_t prog_bytes = sizeof(int32_t) * static_cast<size_t>(group) * n_jobs * max_strips;
    const size_t bndc_floats = static_cast<size_t>(cost_rows) * per * group, lck_floats = static_cast<size_t>(std::max<int64_t>(ck_floats_per_read, 1)) * group;
    const int sets = two_sets ? 2 : 1;
    const size_t bndc_cap = c->d_bndc.cap;
    if ((rc = c->d_lprog.reserve(prog_bytes * sets)) || (rc = c->d_lticket.reserve(128))) return rc;
    if ((rc = c->d_bndc.reserve(sizeof(float) * bndc_floats * sets)) ||
        (rc = c->d_lbest.reserve(4 * n_part)) || (rc = c->d_lsecond.reserve(4 * n_part)) || (rc = c->d_lend.reserve(4 * n_part)) ||
        (rc = c->d_lwin.reserve(4 * 5 * static_cast<size_t>(n_long))) ||
        (rc = c->d_lck.reserve(sizeof(float) * lck_floats * sets)))
        return rc;
    if (c->d_bndc.cap != bndc_cap) HIP_TRY(hipMemsetAsync(c->d_bndc.p, 0x7f, c->d_bndc.cap, st));  // fresh allocation: 3.4e38 everywhere (see the pad note in sdtw_strips.hpp)
    {  // the prefix sums of every group, each in its own words: one upload for all groups, no host wait between them
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
    hipStream_t const first = st;
    if (two_sets) {  // the second stream starts behind the staging upload (and whatever `first` waited for)
        HIP_TRY(hipEventRecord(c->lev[2], first));
        HIP_TRY(hipStreamWaitEvent(c->stream_long2, c->lev[2], 0));
    }
    for (int32_t g0 = 0, gi = 0; g0 < n_long; g0 += group, ++gi) {
        const int32_t gn = std::min(group, n_long - g0);
        const int set = two_sets ? (gi & 1) : 0;  // groups of one set run on one stream: its scratch is free when the next one starts
        st = set ? c->stream_long2 : first;
        sfa::StripArgs sa{};
        sa.queries = d_queries;
        sa.q_off = d_q_off;
        sa.reads = reinterpret_cast<const int32_t *>(ds + o_reads) + g0;
        sa.ref = c->d_ref.as<float>();
        sa.job_off = c->d_job_off.as<int64_t>();
        sa.job_len = c->d_job_len.as<int32_t>();
        sa.bnd_off = reinterpret_cast<const int64_t *>(ds + o_bnd);
        sa.bnd_cost = c->d_bndc.as<float>() + set * bndc_floats;
        sa.bnd_stride = cost_rows * per;
        sa.p_best = c->d_lbest.as<float>() + static_cast<size_t>(g0) * n_jobs;
        sa.p_second = c->d_lsecond.as<float>() + static_cast<size_t>(g0) * n_jobs;
        sa.p_end = c->d_lend.as<int32_t>() + static_cast<size_t>(g0) * n_jobs;
        sa.w_job = win + g0;
        sa.w_ws = win + n_long + g0;
        sa.w_score = reinterpret_cast<const float *>(win + 2 * static_cast<size_t>(n_long) + g0);
        sa.t_st = win + 3 * static_cast<size_t>(n_long) + g0;
        sa.t_end = win + 4 * static_cast<size_t>(n_long) + g0;
        sa.ck = c->d_lck.as<float>() + set * lck_floats;
        sa.ck_off = reinterpret_cast<const int64_t *>(ds + o_ck);
        sa.ck_shift = ck_shift;
        sa.max_strips = max_strips;
        sa.trace_margin = static_cast<int32_t>(c->opt_trace_margin);
        sa.n_long = gn;
        sa.n_jobs = n_jobs;
        sa.rev_query = ((c->flag & SFA_RNA) && !(c->flag & SFA_INV)) ? 1 : 0;
        sa.err = c->d_badcount.as<unsigned>() + 4;
        // a strip legitimately waits for the strip above to get kPipeBlock + 72 columns ahead, behind every strip above that one: never
        // less than ~10 us per such column and strip (0.2 us measured), however small the option or busy the device
        sa.spin_limit = std::max<int64_t>(c->opt_spin_limit_ms, (static_cast<int64_t>(sfa::kPipeBlock + 72) * max_strips) / 100 + 1) * 100000;  // 100 MHz ticks
        c->strip_limit_ms = sa.spin_limit / 100000;
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
        const dim3 grid2((gn + 3) / 4);
        {  // pass 1: one wave per (job, read, strip), tickets in that order
            const int32_t *soff = reinterpret_cast<const int32_t *>(hs + o_soff) + g0 + gi;  // (uploaded with the staging area, before the loop)
            sa.progress = reinterpret_cast<int32_t *>(c->d_lprog.as<char>() + set * prog_bytes);
            sa.ticket = c->d_lticket.as<unsigned>() + set * 16;  // (words 8.. of a set's 64 bytes: where a dropped strip publishes, see strip_pipe_task)
            HIP_TRY(hipMemsetAsync(sa.progress, 0, sizeof(int32_t) * static_cast<size_t>(gn) * n_jobs * max_strips, st));
            HIP_TRY(hipMemsetAsync(sa.ticket, 0, 4, st));
            sa.strip_off = reinterpret_cast<const int32_t *>(ds + o_soff) + g0 + gi;
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
        }
        KERNEL_TRY();
        fa.mode = 1;
        hipLaunchKernelGGL(sfa::sdtw_strip_finalize_kernel, fgrid, fblock, 0, st, fa);
        KERNEL_TRY();
        if (std_dtw)  // pass 2: strip by strip from the last one upwards
            hipLaunchKernelGGL((sfa::sdtw_strip_chain_kernel<true>), grid2, block, 0, st, sa);
        else
            hipLaunchKernelGGL((sfa::sdtw_strip_chain_kernel<false>), grid2, block, 0, st, sa);
        KERNEL_TRY();
        fa.mode = 2;
        hipLaunchKernelGGL(sfa::sdtw_strip_finalize_kernel, fgrid, fblock, 0, st, fa);
        KERNEL_TRY();
        c->prof.fill_launches++;
    }
    if (two_sets) {  // whoever waits for `first` waits for both
        HIP_TRY(hipEventRecord(c->lev[3], c->stream_long2));
        HIP_TRY(hipStreamWaitEvent(first, c->lev[3], 0));
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
        sum.fused_trace = std::max(sum.fused_trace, c->prof.fused_trace);
        sum.n_tasks += c->prof.n_tasks;
        sum.n_chunks = std::max(sum.n_chunks, c->prof.n_chunks);
        sum.non_finite_reads += c->prof.non_finite_reads;
    }
    c->prof = sum;
    return SFA_OK;
}

}  // namespace

// Core of every align entry point: queries already in HBM, results left in HBM.
int sfa::align_device(sfa_ctx *c, const float *d_queries, const int64_t *q_off, int32_t n, ResultRow *d_out) {
    if (n == 0) return SFA_OK;
    // ---- host: plan the batch (quads, classes, chunks, checkpoint interval) -------------------------------
    sfa::PlanParams pp;
    pp.n_sims = static_cast<int64_t>(c->cu_count) * 4;
    pp.waves_per_simd = c->opt_waves_per_simd;
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
    pp.span_sixteenths = c->span_sixteenths;
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
    if (!c->in_slice && c->opt_ckpt_interval == 0 && plan.ck_shift > 9 && n >= 2 * c->opt_min_slice_reads) {
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
    // (measured: 2 048 reads 2.9 -> 3.3 ms per batch), so small launches keep the separate pass-2 launch.  Two fills can carry
    // tickets: the LDS-checkpoint fill (R <= 16; std_dtw: its sparse HBM store) and the 32-row subsequence fill, whose snapshots go
    // to HBM -- write-through in that launch, because its pass-2 waves read them from whatever XCD they land on.
    const bool std_dtw = (c->flag & SFA_DTW) != 0;
    const bool fusable = plan.lds_ckpt || (plan.max_R == 32 && !std_dtw && plan.n_seg == 1 && plan.ck_shift > 0);
    const bool fused = fusable && c->opt_fused_trace && n_quads > 0 &&
                       (c->opt_fused_trace > 1 || static_cast<int64_t>(n_quads) * n_chunks > static_cast<int64_t>(c->cu_count) * 4 * SFA_LCK_WAVES);
    if (fused && (rc = c->d_args.reserve(sizeof(DpArgs)))) return rc;
    if (fused && ((rc = c->d_ticket.reserve(64)) || (rc = c->d_quaddone.reserve(4 * static_cast<size_t>(std::max(n_quads, 1)))))) return rc;
    if (plan.lds_ckpt && ((rc = c->d_bestrec.reserve(sizeof(float) * sfa::kLdsCkPlanes * 64 * n_part / 4)) || (rc = c->d_beste.reserve(4 * n_part)) ||
                          (rc = c->d_gbest.reserve(4 * static_cast<size_t>(n)))))
        return rc;
    if ((rc = c->d_bad.reserve(static_cast<size_t>(n))) || (rc = c->d_badcount.reserve(256)) || (rc = c->h_badcount.reserve(256))) return rc;
    if (plan.ck_floats > 0 && (rc = c->d_ck.reserve(sizeof(float) * plan.ck_floats))) return rc;
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
    da.ck_shift = plan.ck_shift;
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
    da.span_hist = (fused && !plan.lds_ckpt) ? c->d_badcount.as<unsigned>() + 8 : nullptr;  // (the LDS route caps its head start instead)
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
        c->quad_limit_ms = da.spin_limit / 100000;
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
        ls = c->stream_long;
        HIP_TRY(hipEventRecord(c->lev[0], st));  // queries, offsets and the non-finite screen are ready
        HIP_TRY(hipStreamWaitEvent(ls, c->lev[0], 0));
        c->prof.fill_launches = 0;  // (counted per group of long reads inside)
        if ((rc = align_long(c, d_queries, da.q_off, q_off, long_reads, long_max, d_out, ls))) {
            (void)hipStreamSynchronize(ls);  // nothing of a failed call may still be running when the caller reuses its buffers
            (void)hipStreamSynchronize(c->stream_long2);
            return rc;
        }
        long_launches = c->prof.fill_launches;
        HIP_TRY(hipEventRecord(c->lev[1], ls));
    }
    if (n_quads > 0) {
        if (plan.lds_ckpt)
            HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_gbest.p), 0x7f800000, static_cast<size_t>(n), st));  // +inf: no score seen yet
        if (fused) {
            HIP_TRY(hipMemsetAsync(c->d_ticket.p, 0, 4, st));
            HIP_TRY(hipMemsetAsync(c->d_quaddone.p, 0, 4 * static_cast<size_t>(n_quads), st));
            fz.mode = 3;  // rows of the reads in no quad; every other row is written by the launch's pass-2 waves
            hipLaunchKernelGGL(sfa::sdtw_finalize_kernel, fgrid, fblock, 0, st, fz);
            KERNEL_TRY();
            da.self = c->d_args.as<DpArgs>();
            HIP_TRY(hipMemcpyAsync(c->d_args.p, &da, sizeof(DpArgs), hipMemcpyHostToDevice, st));  // (pageable source: staged before the call returns)
            launch_fill_fused(plan.max_R, std_dtw, da, st);
        } else if (plan.lds_ckpt) {
            launch_fill_lck(plan.max_R, std_dtw, da, st);
        } else {
            launch_fill(plan.max_R, std_dtw, da, st);
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
    fz.mode = 1;
    hipLaunchKernelGGL(sfa::sdtw_finalize_kernel, fgrid, fblock, 0, st, fz);
    KERNEL_TRY();
    HIP_TRY(hipEventRecord(c->ev[2], st));
    if (n_quads > 0) {
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
        HIP_TRY(hipStreamWaitEvent(st, c->lev[1], 0));
    }
    HIP_TRY(hipMemcpyAsync(c->h_badcount.p, c->d_badcount.p, 32 + 4 * sfa::kSpanBuckets, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(c->ev[4], st));

    c->prof.cells = (plan.query_events + long_events) * c->total_cols;
    c->prof.ckpt_interval = plan.ck_shift ? (1 << plan.ck_shift) : 0;
    c->prof.ckpt_bytes = static_cast<int64_t>(sizeof(float)) * plan.ck_floats;
    c->prof.lds_ckpt = plan.lds_ckpt ? (fused ? 2 : 1) : 0;
    c->prof.fused_trace = (fused && n_quads > 0) ? 1 : 0;
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

int sfa::resolve_profile(sfa_ctx *c) {
    if (!c->prof_pending) return SFA_OK;
    HIP_TRY(hipEventSynchronize(c->ev[4]));
    float a = 0, b = 0, d = 0, t = 0;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&d, c->ev[2], c->ev[3]));
    HIP_TRY(hipEventElapsedTime(&t, c->ev[0], c->ev[4]));
    if (c->prof.fused_trace) d = 0;  // pass 2 ran inside the fill launch: there was no trace launch to time
    if (c->long_pending) {  // the row-strip sweeps of long queries are fills
        float l = 0;
        HIP_TRY(hipEventElapsedTime(&l, c->ev[5], c->ev[4]));
        a += l;
        a = t;  // ... and ran BESIDE the wave kernels: the intervals on this stream say nothing about stages
        d = 0;
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
    if (c->h_badcount.p) {  // a wave of the batch gave up waiting for another one (bounded_wait_ge): the rows are not to be used
        const unsigned *e = c->h_badcount.as<unsigned>() + 4;
        if (e[0] == sfa::kErrQuadWait)
            return fail(SFA_EKERNEL, "fused launch: pass 2 of quad %u waited %lld ms for its fill tasks (%u of them had completed); rows of this batch are invalid",
                        e[1], (long long)c->quad_limit_ms, e[2]);
        if (e[0] == sfa::kErrStripWait)
            return fail(SFA_EKERNEL, "row strips: a strip waited %lld ms for column %u of the row above (column %u was published); rows of this batch are invalid",
                        (long long)c->strip_limit_ms, e[1], e[2]);
        if (e[0]) return fail(SFA_EKERNEL, "device error word %u (%u, %u)", e[0], e[1], e[2]);
    }
    if (c->h_badcount.p) {  // spans of this (valid) batch's alignments -> head start of the next batch's pass 2 (sfa_plan.hpp)
        const unsigned *h = c->h_badcount.as<unsigned>() + 8;
        uint64_t total = 0;
        for (int b = 0; b < sfa::kSpanBuckets; ++b) total += h[b];
        if (total >= 64) {  // the bucket below which 99.9 % of the alignments lie (its upper edge: b + 1 sixteenths) and one more, never above a whole query
            uint64_t acc = 0;
            int b = 0;
            for (; b < sfa::kSpanBuckets; ++b) {
                acc += h[b];
                if (acc * 1000 >= total * 999) break;
            }
            c->span_sixteenths = std::min(16, b + 2);
        }
    }
    return SFA_OK;
}

extern "C" {

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
    if (int rc = resolve_profile(c)) return rc;  // a batch submitted and never waited for: its error words are this call's
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

}  // extern "C"
