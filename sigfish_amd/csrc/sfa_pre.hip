// sfa_pre.hip -- the stages in front of the alignment on the device: raw samples in (event detection,
// query window, normalisation: sfa_align_raw) and BLOW5 records in (inflate, field parsing, StreamVByte: sfa_align_blow5), each
// ending in the alignment stage of sfa_align.hip.
#include "sfa_ctx.hpp"
#include "sdtw_kernels.hpp"
#include "events_kernels.hpp"
#define SFA_DEFINE_BLOW5_KERNELS
#include "blow5_kernels.hpp"
#include "host/blow5.hpp"

using sfa::ResultRow;
using sfa::resolve_profile;
using sfa::align_device;
using sfa::for_each_shard;
using sfa::shard_ranges;

extern "C" {

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
    hipStream_t sp = st;
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
    ea.use_flags = (c->opt_ev_parallel & 1) ? 1 : 0;
    ea.peak_flag = c->e_pflag.as<int32_t>();
    // measured: 76 us against 1.5 ms for a 512-read batch, 2.1 ms against 1.3 ms for 16 Ki reads (it does ~1.3x the work of
    // the sequential walk, in 64x more waves): used while the batch cannot fill the chip with one read per lane pair
    ea.use_peak_flags = ((c->opt_ev_parallel & 2) && n <= 8192) ? 1 : 0;
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
    hipStream_t sp = st;
    HIP_TRY(hipMemcpyAsync(c->b_in.p, records, static_cast<size_t>(in_bytes), hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemcpyAsync(c->b_inoff.p, rec_off, 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, sp));
    HIP_TRY(hipMemsetAsync(c->b_bad.p, 0, 4 * static_cast<size_t>(n), sp));
    HIP_TRY(hipEventRecord(c->bev[0], sp));
    sfa::FieldsArgs fa{};
    if (record_zlib) {
        HIP_TRY(hipMemcpyAsync(c->b_outoff.p, slot.data(), 8 * static_cast<size_t>(n + 1), hipMemcpyHostToDevice, sp));
        sfa::InflateArgs ia{c->b_in.as<uint8_t>(), c->b_inoff.as<int64_t>(), c->b_out.as<uint8_t>(), c->b_outoff.as<int64_t>(), c->b_len.as<int32_t>(), n};
        const int lanes = inflate_lanes(n, c->cu_count);
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

}  // extern "C"
