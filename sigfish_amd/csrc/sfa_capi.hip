// sfa_capi.hip -- C-ABI (include/sigfish_amd.h) over the gfx950 kernels in sdtw_kernels.hpp.
//
// Replaces the reference's accelerator hook align_db() (src/sigfish.c:1003-1015) and its init / teardown slots
// (src/sigfish.c:200-204, 221-225).  Host work done here: pack the reference event arrays into one padded HBM
// buffer, group reads into "quads" of equal query length (four reads share a wavefront), pick the
// rows-per-lane class, launch, and hand back one row per read in input order.
// There is NO CPU fallback: every failure is reported through the return code + sfa_last_error().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sigfish_amd.h"
#include "sdtw_kernels.hpp"

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

// A device buffer that only ever grows (batches reuse it; nothing is allocated inside a steady-state call).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SFA_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
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
        size_t want = bytes + bytes / 4 + 256;
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

int rows_per_lane_for(int qlen) {
    if (qlen <= 64) return 4;
    if (qlen <= 128) return 8;
    if (qlen <= 256) return 16;
    if (qlen <= 512) return 32;
    return 0;
}

// Split the job list into n_chunks contiguous, non-empty ranges of roughly equal reference columns.
void split_jobs(const std::vector<int32_t> &job_len, int64_t total_cols, int32_t n_chunks, int32_t *chunk_begin) {
    const int32_t n_jobs = static_cast<int32_t>(job_len.size());
    int64_t acc = 0;
    int32_t j = 0;
    chunk_begin[0] = 0;
    for (int32_t ch = 1; ch < n_chunks; ++ch) {
        const int64_t want = total_cols * ch / n_chunks;
        // take at least one job, then keep taking while below the target and enough jobs remain for the rest
        acc += job_len[j++];
        while (j < n_jobs - (n_chunks - ch) && acc + job_len[j] / 2 < want) acc += job_len[j++];
        chunk_begin[ch] = j;
    }
    chunk_begin[n_chunks] = n_jobs;
}

}  // namespace

struct sfa_ctx {
    int device = 0;
    uint32_t flag = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // fill start, fill end, finalize end, spare
    int cu_count = 256;

    // reference model (immutable after init)
    int32_t num_ref = 0, n_jobs = 0;
    int64_t total_cols = 0;  // sum over jobs of rlen
    std::vector<int32_t> h_job_len;
    DevBuf d_ref, d_job_off, d_job_len, d_job_contig, d_job_strand, d_ref_len, d_ref_off;

    // per-batch scratch
    DevBuf d_queries, d_qoff, d_order, d_quad_qlen, d_slot, d_chunk, d_pbest, d_pend, d_pst, d_pjob, d_psecond, d_out;
    PinBuf h_stage, h_out;

    sfa_profile_t prof{};
    bool prof_pending = false;
};

namespace {

using sfa::FillArgs;
using sfa::FinalizeArgs;
using sfa::ResultRow;

static_assert(sizeof(ResultRow) == sizeof(sfa_result_t), "result row layout");

template <int R>
void launch_fill(bool std_dtw, const FillArgs &a, hipStream_t st) {
    const int blocks = (a.n_tasks + 3) / 4;
    if (std_dtw)
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<R, true, true>), dim3(blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((sfa::sdtw_fill_kernel<R, true, false>), dim3(blocks), dim3(256), 0, st, a);
}

// Core of both align entry points: queries already in HBM, results left in HBM.
int align_device(sfa_ctx *c, const float *d_queries, const int64_t *q_off, int32_t n, ResultRow *d_out) {
    if (n == 0) return SFA_OK;
    // ---- host: classify reads by query length, form quads of identical length ---------------------------
    std::vector<int32_t> qlen(n);
    int maxq = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int64_t l = q_off[i + 1] - q_off[i];
        if (l < 0) return fail(SFA_EINVAL, "q_off not monotone at read %d", i);
        if (l > 512) return fail(SFA_ERANGE, "read %d has %lld events; this build supports up to 512", i, (long long)l);
        qlen[i] = static_cast<int32_t>(l);
        maxq = std::max(maxq, qlen[i]);
    }
    std::vector<int32_t> count(maxq + 2, 0);
    for (int32_t i = 0; i < n; ++i) count[qlen[i]]++;
    // classes in launch order: R = 32, 16, 8, 4 (long first); within a class by descending length
    struct Cls {
        int R, quad_base, n_quads;
    };
    std::vector<Cls> classes;
    std::vector<int32_t> quad_start(maxq + 2, -1);  // first quad of each length
    int32_t n_quads = 0;
    for (int R : {32, 16, 8, 4}) {
        Cls cl{R, n_quads, 0};
        for (int l = maxq; l >= 1; --l) {
            if (count[l] == 0 || rows_per_lane_for(l) != R) continue;
            quad_start[l] = n_quads;
            n_quads += (count[l] + 3) / 4;
        }
        cl.n_quads = n_quads - cl.quad_base;
        if (cl.n_quads > 0) classes.push_back(cl);
    }
    const int32_t n_valid = n - count[0];

    // chunking of the (contig,strand) job list: enough wave-tasks to fill the chip, otherwise one pass per read
    int32_t n_chunks = 1;
    if (n_quads > 0) {
        const int64_t target = static_cast<int64_t>(c->cu_count) * 4 * 6;  // ~6 waves per SIMD
        n_chunks = static_cast<int32_t>(std::min<int64_t>(c->n_jobs, std::max<int64_t>(1, target / n_quads)));
    }
    std::vector<int32_t> chunk_begin(n_chunks + 1);
    split_jobs(c->h_job_len, c->total_cols, n_chunks, chunk_begin.data());

    // staging layout (pinned): q_off[n+1] | order[n_quads*4] | quad_qlen[n_quads] | slot[n] | chunk_begin
    const size_t sz_qoff = sizeof(int64_t) * (n + 1);
    const size_t sz_order = sizeof(int32_t) * 4 * std::max(n_quads, 1);
    const size_t sz_qq = sizeof(int32_t) * std::max(n_quads, 1);
    const size_t sz_slot = sizeof(int32_t) * n;
    const size_t sz_chunk = sizeof(int32_t) * (n_chunks + 1);
    if (int rc = c->h_stage.reserve(sz_qoff + sz_order + sz_qq + sz_slot + sz_chunk)) return rc;
    char *hs = c->h_stage.as<char>();
    int64_t *h_qoff = reinterpret_cast<int64_t *>(hs);
    int32_t *h_order = reinterpret_cast<int32_t *>(hs + sz_qoff);
    int32_t *h_qq = reinterpret_cast<int32_t *>(hs + sz_qoff + sz_order);
    int32_t *h_slot = reinterpret_cast<int32_t *>(hs + sz_qoff + sz_order + sz_qq);
    int32_t *h_chunk = reinterpret_cast<int32_t *>(hs + sz_qoff + sz_order + sz_qq + sz_slot);
    memcpy(h_qoff, q_off, sz_qoff);
    std::fill(h_order, h_order + 4 * std::max(n_quads, 1), -1);
    std::vector<int32_t> fill_pos(maxq + 2, 0);
    for (int32_t i = 0; i < n; ++i) {
        const int l = qlen[i];
        if (l == 0) {
            h_slot[i] = -1;
            continue;
        }
        const int32_t k = fill_pos[l]++;
        const int32_t sl = (quad_start[l] + (k >> 2)) * 4 + (k & 3);
        h_order[sl] = i;
        h_slot[i] = sl;
    }
    for (int l = 1; l <= maxq; ++l)
        if (count[l])
            for (int32_t qd = quad_start[l]; qd < quad_start[l] + (count[l] + 3) / 4; ++qd) h_qq[qd] = l;
    memcpy(h_chunk, chunk_begin.data(), sz_chunk);

    const size_t n_part = static_cast<size_t>(std::max(n_quads, 1)) * n_chunks * 4;
    int rc;
    if ((rc = c->d_qoff.reserve(sz_qoff)) || (rc = c->d_order.reserve(sz_order)) || (rc = c->d_quad_qlen.reserve(sz_qq)) ||
        (rc = c->d_slot.reserve(sz_slot)) || (rc = c->d_chunk.reserve(sz_chunk)) || (rc = c->d_pbest.reserve(4 * n_part)) ||
        (rc = c->d_pend.reserve(4 * n_part)) || (rc = c->d_pst.reserve(4 * n_part)) || (rc = c->d_pjob.reserve(4 * n_part)) ||
        (rc = c->d_psecond.reserve(4 * n_part)))
        return rc;

    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_qoff.p, h_qoff, sz_qoff, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_order.p, h_order, sz_order, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_quad_qlen.p, h_qq, sz_qq, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_slot.p, h_slot, sz_slot, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_chunk.p, h_chunk, sz_chunk, hipMemcpyHostToDevice, st));

    // ---- device: fill (one launch per length class) + finalize ------------------------------------------
    const bool std_dtw = (c->flag & SFA_DTW) != 0;
    FillArgs fa{};
    fa.queries = d_queries;
    fa.q_off = c->d_qoff.as<int64_t>();
    fa.order = c->d_order.as<int32_t>();
    fa.quad_qlen = c->d_quad_qlen.as<int32_t>();
    fa.ref = c->d_ref.as<float>();
    fa.job_off = c->d_job_off.as<int64_t>();
    fa.job_len = c->d_job_len.as<int32_t>();
    fa.chunk_begin = c->d_chunk.as<int32_t>();
    fa.p_best = c->d_pbest.as<float>();
    fa.p_end = c->d_pend.as<int32_t>();
    fa.p_st = c->d_pst.as<int32_t>();
    fa.p_job = c->d_pjob.as<int32_t>();
    fa.p_second = c->d_psecond.as<float>();
    fa.n_chunks = n_chunks;
    fa.rev_query = ((c->flag & SFA_RNA) && !(c->flag & SFA_INV)) ? 1 : 0;

    HIP_TRY(hipEventRecord(c->ev[0], st));
    int64_t launches = 0;
    for (const Cls &cl : classes) {
        fa.quad_base = cl.quad_base;
        fa.n_quads = cl.n_quads;
        fa.n_tasks = cl.n_quads * n_chunks;
        switch (cl.R) {
            case 4: launch_fill<4>(std_dtw, fa, st); break;
            case 8: launch_fill<8>(std_dtw, fa, st); break;
            case 16: launch_fill<16>(std_dtw, fa, st); break;
            default: launch_fill<32>(std_dtw, fa, st); break;
        }
        HIP_TRY(hipGetLastError());
        ++launches;
    }
    HIP_TRY(hipEventRecord(c->ev[1], st));

    FinalizeArgs fz{};
    fz.slot_of_read = c->d_slot.as<int32_t>();
    fz.p_best = fa.p_best;
    fz.p_end = fa.p_end;
    fz.p_st = fa.p_st;
    fz.p_job = fa.p_job;
    fz.p_second = fa.p_second;
    fz.job_contig = c->d_job_contig.as<int32_t>();
    fz.job_strand = c->d_job_strand.as<int8_t>();
    fz.ref_len = c->d_ref_len.as<int32_t>();
    fz.ref_st_offset = c->d_ref_off.as<int32_t>();
    fz.out = d_out;
    fz.n_reads = n;
    fz.n_chunks = n_chunks;
    hipLaunchKernelGGL(sfa::sdtw_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, fz);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[2], st));

    int64_t qsum = 0;
    for (int32_t i = 0; i < n; ++i) qsum += qlen[i];
    c->prof.cells = qsum * c->total_cols;
    c->prof.fill_launches = launches;
    c->prof_pending = true;
    (void)n_valid;
    return SFA_OK;
}

int resolve_profile(sfa_ctx *c) {
    if (!c->prof_pending) return SFA_OK;
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    float a = 0, b = 0, t = 0;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&t, c->ev[0], c->ev[2]));
    c->prof.fill_ms = a;
    c->prof.finalize_ms = b;
    c->prof.total_ms = t;
    c->prof_pending = false;
    return SFA_OK;
}

}  // namespace

extern "C" {

const char *sfa_last_error(void) { return g_err.c_str(); }
const char *sfa_version(void) { return SFA_VERSION; }

int sfa_init(sfa_ctx_t **out, const sfa_ref_t *ref, uint32_t flag, int device) {
    if (!out || !ref || ref->num_ref <= 0 || !ref->ref_lengths || !ref->forward)
        return fail(SFA_EINVAL, "sfa_init: null or empty reference");
    const bool rna = (flag & SFA_RNA) != 0;
    if (!rna && !ref->reverse) return fail(SFA_EINVAL, "sfa_init: DNA needs reverse arrays");
    if ((flag & SFA_DTW) && !rna) return fail(SFA_EINVAL, "sfa_init: --dtw-std is RNA only (src/dtw_main.c:249-252)");
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
    c->num_ref = ref->num_ref;
    auto bail = [&](int rc) {
        sfa_destroy(c);
        return rc;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(SFA_ENODEV, "hipStreamCreate failed"));
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(SFA_ENODEV, "hipEventCreate failed"));

    // job list in the reference's processing order: contig ascending, '+' before '-' (src/sigfish.c:870-960)
    const int strands = rna ? 1 : 2;
    c->n_jobs = ref->num_ref * strands;
    std::vector<int64_t> job_off(c->n_jobs);
    std::vector<int32_t> job_contig(c->n_jobs), ref_off(ref->num_ref);
    std::vector<int8_t> job_strand(c->n_jobs);
    c->h_job_len.resize(c->n_jobs);
    int64_t total = sfa::kRefPad;
    for (int32_t r = 0; r < ref->num_ref; ++r) {
        const int32_t rl = ref->ref_lengths[r];
        if (rl <= 0) return bail(fail(SFA_EINVAL, "contig %d has non-positive length %d", r, rl));
        ref_off[r] = ref->ref_st_offset ? ref->ref_st_offset[r] : 0;
        for (int s = 0; s < strands; ++s) {
            const int32_t j = r * strands + s;
            job_off[j] = total;
            c->h_job_len[j] = rl;
            job_contig[j] = r;
            job_strand[j] = s == 0 ? '+' : '-';
            total += rl + sfa::kRefPad;
            c->total_cols += rl;
        }
    }
    std::vector<float> packed(total, 0.0f);
    for (int32_t j = 0; j < c->n_jobs; ++j) {
        const float *src = (job_strand[j] == '+') ? ref->forward[job_contig[j]] : ref->reverse[job_contig[j]];
        if (!src) return bail(fail(SFA_EINVAL, "missing reference array for contig %d", job_contig[j]));
        memcpy(&packed[job_off[j]], src, sizeof(float) * c->h_job_len[j]);
    }
    int rc;
    if ((rc = c->d_ref.reserve(sizeof(float) * total)) || (rc = c->d_job_off.reserve(sizeof(int64_t) * c->n_jobs)) ||
        (rc = c->d_job_len.reserve(sizeof(int32_t) * c->n_jobs)) || (rc = c->d_job_contig.reserve(sizeof(int32_t) * c->n_jobs)) ||
        (rc = c->d_job_strand.reserve(c->n_jobs)) || (rc = c->d_ref_len.reserve(sizeof(int32_t) * ref->num_ref)) ||
        (rc = c->d_ref_off.reserve(sizeof(int32_t) * ref->num_ref)))
        return bail(rc);
#define UP(dst, src, bytes)                                                                        \
    if (hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(SFA_ENODEV, "upload of reference model failed"))
    UP(c->d_ref.p, packed.data(), sizeof(float) * total);
    UP(c->d_job_off.p, job_off.data(), sizeof(int64_t) * c->n_jobs);
    UP(c->d_job_len.p, c->h_job_len.data(), sizeof(int32_t) * c->n_jobs);
    UP(c->d_job_contig.p, job_contig.data(), sizeof(int32_t) * c->n_jobs);
    UP(c->d_job_strand.p, job_strand.data(), c->n_jobs);
    UP(c->d_ref_len.p, ref->ref_lengths, sizeof(int32_t) * ref->num_ref);
    UP(c->d_ref_off.p, ref_off.data(), sizeof(int32_t) * ref->num_ref);
#undef UP
    *out = c;
    return SFA_OK;
}

void sfa_destroy(sfa_ctx_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->d_ref, &c->d_job_off, &c->d_job_len, &c->d_job_contig, &c->d_job_strand, &c->d_ref_len, &c->d_ref_off,
                      &c->d_queries, &c->d_qoff, &c->d_order, &c->d_quad_qlen, &c->d_slot, &c->d_chunk, &c->d_pbest, &c->d_pend,
                      &c->d_pst, &c->d_pjob, &c->d_psecond, &c->d_out})
        b->release();
    c->h_stage.release();
    c->h_out.release();
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sfa_align_batch_device(sfa_ctx_t *c, const float *d_queries, const int64_t *q_off, int32_t n, sfa_result_t *d_out, int sync) {
    if (!c || !q_off || n < 0 || (n > 0 && (!d_queries || !d_out))) return fail(SFA_EINVAL, "sfa_align_batch_device: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = align_device(c, d_queries, q_off, n, reinterpret_cast<ResultRow *>(d_out))) return rc;
    if (sync) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        return resolve_profile(c);
    }
    return SFA_OK;
}

int sfa_align_batch(sfa_ctx_t *c, const float *queries, const int64_t *q_off, int32_t n, sfa_result_t *out) {
    if (!c || !q_off || n < 0 || (n > 0 && (!queries || !out))) return fail(SFA_EINVAL, "sfa_align_batch: bad argument");
    if (n == 0) return SFA_OK;
    HIP_TRY(hipSetDevice(c->device));
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
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_out.p, sizeof(sfa_result_t) * n);
    return resolve_profile(c);
}

int sfa_align_events(sfa_ctx_t *c, const sfa_event_t *const *events, const int64_t *n_events, const int64_t *qstart,
                     const int64_t *qend, int32_t n, sfa_result_t *out) {
    if (!c || n < 0 || (n > 0 && (!events || !n_events || !qstart || !qend || !out)))
        return fail(SFA_EINVAL, "sfa_align_events: bad argument");
    // gather db->et[i].event[qstart..qend).mean (AoS, stride 24 B) into the packed SoA the kernels read
    std::vector<int64_t> q_off(n + 1, 0);
    for (int32_t i = 0; i < n; ++i) {
        int64_t l = (n_events[i] > 0 && events[i]) ? qend[i] - qstart[i] : 0;
        if (l < 0) l = 0;
        q_off[i + 1] = q_off[i] + l;
    }
    std::vector<float> q(std::max<int64_t>(q_off[n], 1));
    for (int32_t i = 0; i < n; ++i) {
        const int64_t l = q_off[i + 1] - q_off[i];
        for (int64_t j = 0; j < l; ++j) q[q_off[i] + j] = events[i][qstart[i] + j].mean;
    }
    return sfa_align_batch(c, q.data(), q_off.data(), n, out);
}

int sfa_sync(sfa_ctx_t *c) {
    if (!c) return fail(SFA_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return resolve_profile(c);
}

int sfa_get_profile(sfa_ctx_t *c, sfa_profile_t *p) {
    if (!c || !p) return fail(SFA_EINVAL, "null argument");
    if (int rc = resolve_profile(c)) return rc;
    *p = c->prof;
    return SFA_OK;
}

void *sfa_stream(sfa_ctx_t *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

}  // extern "C"
