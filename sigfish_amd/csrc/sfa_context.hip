// sfa_context.hip -- contexts of the C-ABI (include/sigfish_amd.h): the reference's init / teardown slots (src/sigfish.c:200-204,
// 221-225).  Packs the reference event arrays into one padded HBM buffer per device, owns streams / events / scratch, options,
// profile, and the small device utilities.
#include "sfa_ctx.hpp"
#include "sdtw_kernels.hpp"  // kRefPad, the debug task-time layout

namespace sfa {
std::string &last_error_slot() {
    thread_local std::string err;
    return err;
}
}  // namespace sfa
using sfa::resolve_profile;

extern "C" {

const char *sfa_last_error(void) { return sfa::last_error_slot().c_str(); }
void sfa_set_error_(const char *msg) { sfa::last_error_slot() = msg ? msg : ""; }  // for the host-side units of this library
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
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(SFA_ENODEV, "hipStreamCreate failed"));
    if (hipStreamCreateWithFlags(&c->stream_long, hipStreamNonBlocking) != hipSuccess) return bail(fail(SFA_ENODEV, "hipStreamCreate failed"));
    if (hipStreamCreateWithFlags(&c->stream_long2, hipStreamNonBlocking) != hipSuccess) return bail(fail(SFA_ENODEV, "hipStreamCreate failed"));
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
                      &c->d_queries, &c->d_stage, &c->d_pbest, &c->d_pend, &c->d_pjob, &c->d_psecond, &c->d_wjob,
                      &c->d_wend, &c->d_wscore, &c->d_tst, &c->d_ck, &c->d_out, &c->e_raw, &c->e_rawoff, &c->e_scale, &c->e_sum,
                      &c->e_sumsq, &c->e_t1, &c->e_t2, &c->e_evoff, &c->e_evstart, &c->e_evlen, &c->e_evmean, &c->e_evstdv, &c->e_nev,
                      &c->e_qstart, &c->e_qoff, &c->e_b0, &c->e_b1, &c->e_b2, &c->e_flag, &c->e_qev, &c->e_pflag, &c->d_verify, &c->d_segfail, &c->d_bndc, &c->d_long, &c->d_lbest, &c->d_lsecond, &c->d_lend, &c->d_lwin, &c->d_lck, &c->d_lprog, &c->d_lticket, &c->d_times, &c->d_ltimes, &c->d_started, &c->d_bad, &c->d_badcount, &c->d_bestrec, &c->d_beste, &c->d_gbest, &c->d_wchunk, &c->d_ticket, &c->d_quaddone, &c->d_args, &c->b_in, &c->b_inoff, &c->b_out, &c->b_outoff, &c->b_len, &c->b_head, &c->b_bad})
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
    if (c->stream_long2) (void)hipStreamDestroy(c->stream_long2);
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
    if (k == "ckpt_interval") {
        if (value != 0 && (value < 4 || (value & (value - 1)))) return fail(SFA_EINVAL, "ckpt_interval must be 0 or a power of two >= 4");
        c->opt_ckpt_interval = value;
    } else if (k == "ckpt_budget_bytes") {
        if (value < 0) return fail(SFA_EINVAL, "ckpt_budget_bytes must be >= 0");
        c->opt_ckpt_budget = value;
    } else if (k == "trace_margin") {
        if (value < -1 || value > (1 << 28)) return fail(SFA_EINVAL, "trace_margin must be -1 (auto) .. 2^28");
        c->opt_trace_margin = value;
    } else if (k == "lane_widening") {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(SFA_EINVAL, "lane_widening must be 0 (auto), 1, 2 or 4");
        c->opt_lane_widening = value;
    } else if (k == "widen_below") {
        if (value < 0 || value > 64) return fail(SFA_EINVAL, "widen_below must be 0..64");
        c->opt_widen_below = value;
    } else if (k == "column_segments") {
        if (value < 0 || value > 64) return fail(SFA_EINVAL, "column_segments must be 0 (auto), 1 (off) or 2..64");
        c->opt_column_segments = value;
    } else if (k == "segment_warm_windows") {
        if (value < 0 || value > 64) return fail(SFA_EINVAL, "segment_warm_windows must be 0..64");
        c->opt_segment_warm = value;
    } else if (k == "min_slice_reads") {
        if (value < 1) return fail(SFA_EINVAL, "min_slice_reads must be >= 1");
        c->opt_min_slice_reads = value;
    } else if (k == "waves_per_simd") {
        if (value < 1 || value > 8) return fail(SFA_EINVAL, "waves_per_simd must be 1..8");
        c->opt_waves_per_simd = value;
    } else if (k == "fused_trace") {
        if (value < 0 || value > 2) return fail(SFA_EINVAL, "fused_trace must be 0 (off), 1 (launches with more tasks than wave slots) or 2 (always)");
        c->opt_fused_trace = value;
    } else if (k == "lds_ckpt") {
        if (value < 0 || value > 2) return fail(SFA_EINVAL, "lds_ckpt must be 0 (off), 1 (where shapes and batch size suit) or 2 (wherever the shapes allow)");
        c->opt_lds_ckpt = value;
    } else if (k == "prio_unit") {
        if (value < 0 || value > (1 << 28)) return fail(SFA_EINVAL, "prio_unit must be 0 (off) .. 2^28");
        c->opt_prio_unit = value;
    } else if (k == "spin_limit_ms") {
        if (value < 1 || value > 3600000) return fail(SFA_EINVAL, "spin_limit_ms must be 1 .. 3 600 000");
        c->opt_spin_limit_ms = value;
    } else if (k == "ev_parallel") {
        if (value < 0 || value > 3) return fail(SFA_EINVAL, "ev_parallel must be 0..3 (bit 0: prefix sums, bit 1: peak picker)");
        c->opt_ev_parallel = value;
    } else if (k == "debug_drop_quad" || k == "debug_drop_strip") {
        // test hooks (tests/test_bounded_waits_gpu.py): a hand-over that never comes.  Every batch then FAILS after the wait limit, so
        // they are not options: refused unless the process asks for them in its environment.
        const char *e = getenv("SFA_TEST_HOOKS");
        if (!e || strcmp(e, "1") != 0) return fail(SFA_EINVAL, "'%s' is a test hook: set SFA_TEST_HOOKS=1 in the environment to use it", key);
        if (value < -1 || value > INT32_MAX) return fail(SFA_EINVAL, "%s must be -1 (off) or an index", key);
        (k == "debug_drop_quad" ? c->opt_debug_drop_quad : c->opt_debug_drop_strip) = value;
    } else {
        return fail(SFA_EINVAL, "unknown option '%s'", key);
    }
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
            sum.fused_trace = std::max(sum.fused_trace, q.fused_trace);
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
