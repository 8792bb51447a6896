// sfa_host.cpp -- host-side pieces of the alignment path that stay on the CPU so that the numbers fed to and
// read from the GPU are bit-identical to the reference's: reference-event-model synthesis, z-normalisation and
// PAF row formatting.  Plain C++ (no HIP); compiled with -ffp-contract=off.
//
// reference: hasindu2008/sigfish v0.2.0 -- src/genref.c:23-47,86-241, src/ref.h:13-77, src/sigfish.c:483-502,
// 628-660.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/sigfish_amd.h"
#include "sfa_plan.hpp"
#include "host/blow5.hpp"
#include "host/inflate.hpp"
#include "host/events.hpp"
#include "host/refio.hpp"
#include "host/sam.hpp"

extern "C" void sfa_set_error_(const char *msg);

namespace {

// 2-bit code of a base; anything outside ACGT/acgt ranks as 'A' (src/ref.h:13-26)
inline uint32_t code_of(unsigned char b) {
    static const struct Table {
        uint8_t t[256];
        Table() {
            memset(t, 0, sizeof t);
            t['C'] = t['c'] = 1;
            t['G'] = t['g'] = 2;
            t['T'] = t['t'] = 3;
        }
    } tab;
    return tab.t[b];
}

// code of the complement as the reference builds it: A<->T, C<->G, everything else -> 'T' (src/ref.h:43-68)
inline uint32_t comp_code_of(unsigned char b) {
    switch (b) {
        case 'A': case 'a': return 3;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 1;
        case 'T': case 't': return 0;
        default: return 3;
    }
}

}  // namespace

extern "C" {

void sfa_znormalise(float *v, uint64_t n) {
    // sequential fp32 accumulation, population variance, sqrt in double (src/genref.c:23-47)
    const float count = static_cast<float>(n);
    float mean = 0.0f;
    for (uint64_t i = 0; i < n; ++i) mean += v[i];
    mean /= count;
    float var = 0.0f;
    for (uint64_t i = 0; i < n; ++i) {
        const float d = v[i] - mean;
        var += d * d;
    }
    var /= count;
    const float sd = static_cast<float>(std::sqrt(static_cast<double>(var)));
    for (uint64_t i = 0; i < n; ++i) v[i] = (v[i] - mean) / sd;
}

int32_t sfa_gen_ref_record(const char *seq, int32_t len, const float *level_mean, uint32_t k, uint32_t flag,
                           int32_t query_size, float *fwd, float *rev, int32_t *st_offset) {
    if (!seq || !level_mean || !fwd || k == 0 || k > 9 || len < static_cast<int32_t>(k)) return -1;
    const bool rna = (flag & SFA_RNA) != 0;
    if (!rna && !rev) return -1;
    const int32_t full = len + 1 - static_cast<int32_t>(k);
    int32_t n = full;
    if (rna && !(flag & SFA_REF)) {  // only the 1.5*q events next to the 3' end (src/genref.c:132-135)
        const uint32_t heu = static_cast<uint32_t>(query_size * 1.5);
        n = heu > static_cast<uint32_t>(full) ? full : static_cast<int32_t>(heu);
    }
    const uint32_t mask = (k == 16) ? 0xffffffffu : ((1u << (2 * k)) - 1u);
    int32_t off = 0;
    // rolling k-mer rank over [begin, begin+n+k-1): rank = sum code(s[i]) * 4^(k-1-i)
    auto roll = [&](int32_t begin, auto code_at, auto store) {
        uint32_t r = 0;
        for (uint32_t i = 0; i + 1 < k; ++i) r = (r << 2) | code_at(begin + static_cast<int32_t>(i));
        for (int32_t j = 0; j < n; ++j) {
            r = ((r << 2) | code_at(begin + j + static_cast<int32_t>(k) - 1)) & mask;
            store(j, level_mean[r]);
        }
    };
    auto fcode = [&](int32_t i) { return code_of(static_cast<unsigned char>(seq[i])); };
    if (!rna) {
        roll(0, fcode, [&](int32_t j, float v) { fwd[j] = v; });
        // reverse complement read in place: rc[i] = comp(seq[len-1-i])
        auto rcode = [&](int32_t i) { return comp_code_of(static_cast<unsigned char>(seq[len - 1 - i])); };
        roll(0, rcode, [&](int32_t j, float v) { rev[j] = v; });
    } else if (flag & SFA_INV) {  // src/genref.c:166-177
        roll(len - n - (static_cast<int32_t>(k) - 1), fcode, [&](int32_t j, float v) { fwd[n - 1 - j] = v; });
    } else if (flag & SFA_END) {  // query end maps to the 5' start of the transcript (src/genref.c:186-187)
        roll(0, fcode, [&](int32_t j, float v) { fwd[j] = v; });
    } else {  // 3' slice (src/genref.c:189-192)
        off = len - n - (static_cast<int32_t>(k) - 1);
        roll(off, fcode, [&](int32_t j, float v) { fwd[j] = v; });
    }
    sfa_znormalise(fwd, static_cast<uint64_t>(n));
    if (!rna) sfa_znormalise(rev, static_cast<uint64_t>(n));
    if (st_offset) *st_offset = off;
    return n;
}

int sfa_plan_batch(const int64_t *q_off, int32_t n_reads, const int32_t *job_len, int32_t n_jobs, int64_t ckpt_interval,
                   int64_t ckpt_budget_bytes, int32_t lane_widening, int32_t *slot_of_read, sfa_plan_info_t *info) {
    if (!q_off || n_reads < 0 || !job_len || n_jobs <= 0 || !info) return SFA_EINVAL;
    std::vector<int32_t> jl(job_len, job_len + n_jobs);
    int64_t total = 0;
    for (int32_t v : jl) total += v;
    sfa::PlanParams pp;
    pp.ckpt_interval = ckpt_interval;
    if (ckpt_budget_bytes > 0) pp.ckpt_budget_bytes = ckpt_budget_bytes;
    if (lane_widening != 0 && lane_widening != 1 && lane_widening != 2 && lane_widening != 4) return SFA_EINVAL;
    pp.lane_widening = lane_widening;
    sfa::BatchPlan plan;
    std::string err;
    if (int rc = sfa::plan_batch(q_off, n_reads, jl, total, pp, &plan, &err)) return rc;
    if (slot_of_read) memcpy(slot_of_read, plan.slot_of_read.data(), sizeof(int32_t) * n_reads);
    info->n_quads = plan.n_quads;
    info->n_chunks = plan.n_chunks;
    info->n_classes = static_cast<int32_t>(plan.classes.size());
    info->max_rows_per_lane = plan.max_R;
    info->ckpt_interval = plan.ck_shift ? (1 << plan.ck_shift) : 0;
    info->trace_margin = plan.trace_margin;
    info->ckpt_bytes = plan.ck_floats * 4;
    info->n_tasks = static_cast<int64_t>(plan.n_quads) * plan.n_chunks;
    info->max_lanes_per_read = plan.max_lanes;
    info->lane_widening = plan.widening;
    return SFA_OK;
}

int sfa_paf_row(char *buf, size_t cap, const sfa_result_t *r, const char *read_id, const char *rname,
                uint64_t start_raw_idx, uint64_t end_raw_idx, uint64_t query_size, uint64_t len_raw_signal,
                uint64_t rlength) {
    // src/sigfish.c:634-635: both in fp32; query_size converts u64 -> float
    const float block_len = static_cast<float>(r->pos_end - r->pos_st);
    const float prod = r->score * block_len;
    const float residue = block_len - prod / static_cast<float>(query_size);
    const int n = snprintf(buf, cap, "%s\t%ld\t%ld\t%ld\t%c\t%s\t%d\t%d\t%d\t%d\t%d\t%d\ttp:A:P\td1:f:%.2f\td2:f:%.2f\n", read_id,
                           static_cast<long>(len_raw_signal), static_cast<long>(start_raw_idx), static_cast<long>(end_raw_idx),
                           static_cast<char>(r->strand), rname, static_cast<int>(rlength), r->pos_st, r->pos_end,
                           static_cast<int>(std::round(static_cast<double>(residue))),
                           static_cast<int>(std::round(static_cast<double>(block_len))), static_cast<int>(r->mapq),
                           static_cast<double>(r->score), static_cast<double>(r->score2));
    if (n < 0 || static_cast<size_t>(n) >= cap) return -1;
    return n;
}


int64_t sfa_detect_events(const int16_t *raw, int64_t n_raw, double digitisation, double offset, double range, int rna,
                          sfa_event_t *out, int64_t cap) {
    if (!raw || n_raw < 0 || (cap > 0 && !out)) return SFA_EINVAL;
    std::vector<float> pa(static_cast<size_t>(n_raw));
    sfa::raw_to_picoamps(raw, n_raw, digitisation, offset, range, pa.data());
    const std::vector<sfa_event_t> ev = sfa::detect_events(pa.data(), n_raw, rna != 0);
    const int64_t n = static_cast<int64_t>(ev.size());
    if (n > 0 && cap > 0) memcpy(out, ev.data(), sizeof(sfa_event_t) * static_cast<size_t>(n < cap ? n : cap));
    return n;
}

int sfa_select_query(sfa_event_t *events, int64_t n_events, const int16_t *raw, int64_t n_raw, double digitisation,
                     double offset, double range, int32_t prefix_size, int32_t query_size, uint32_t flag, int pore,
                     int64_t *qstart, int64_t *qend) {
    if (!events || n_events <= 0 || !qstart || !qend) return 0;
    std::vector<sfa_event_t> ev(events, events + n_events);
    std::vector<float> pa;
    if (prefix_size < 0 && raw && n_raw > 0) {
        pa.resize(static_cast<size_t>(n_raw));
        sfa::raw_to_picoamps(raw, n_raw, digitisation, offset, range, pa.data());
    }
    int status = 0;
    const bool keep = sfa::select_and_normalise(ev, raw, n_raw, pa.data(), prefix_size, query_size, flag, pore, qstart, qend, &status);
    memcpy(events, ev.data(), sizeof(sfa_event_t) * static_cast<size_t>(n_events));
    return keep ? 1 : 0;
}

}  // extern "C"
namespace {
// the winner's warp path from its result row: query in DP order, columns of the strand's own array
sfa::WarpPath path_of_row(const sfa_result_t *r, const sfa_event_t *events, int64_t qstart, int64_t qend, const float *ref_array,
                          int32_t ref_len, int32_t ref_st_offset, uint32_t flag) {
    const bool rna = (flag & SFA_RNA) != 0;
    const int32_t qlen = static_cast<int32_t>(qend - qstart);
    std::vector<float> q(static_cast<size_t>(qlen));
    const bool reversed = rna && !(flag & SFA_INV);  // src/sigfish.c:860-866
    for (int32_t j = 0; j < qlen; ++j) q[reversed ? qlen - 1 - j : j] = events[qstart + j].mean;
    // undo the strand flip and offset of src/sigfish.c:971-975 to get columns of the strand's own array
    const bool plus = r->strand == '+';
    const int32_t st = plus ? r->pos_st - ref_st_offset : ref_len - (r->pos_end - ref_st_offset);
    const int32_t en = plus ? r->pos_end - ref_st_offset : ref_len - (r->pos_st - ref_st_offset);
    return sfa::band_traceback(q.data(), qlen, ref_array, ref_len, st, en, (flag & SFA_DTW) != 0);
}
}  // namespace
extern "C" {

int32_t sfa_r2qevent_map(const sfa_result_t *r, const sfa_event_t *events, int64_t qstart, int64_t qend, const float *ref_array,
                         int32_t ref_len, int32_t ref_st_offset, uint32_t flag, int32_t *pairs, int32_t cap_pairs) {
    if (!r || !events || !ref_array || qend <= qstart || !r->valid || r->rid < 0) return SFA_EINVAL;
    const int32_t need = r->pos_end - r->pos_st + 1;  // r2qevent_size, src/sigfish.c:610-612
    if (need <= 0) return SFA_EINVAL;
    if (!pairs || cap_pairs < need) return pairs ? SFA_ERANGE : need;
    const sfa::WarpPath path = path_of_row(r, events, qstart, qend, ref_array, ref_len, ref_st_offset, flag);
    if (path.px.empty()) return SFA_EINVAL;
    const std::vector<int32_t> p = sfa::path_to_pairs(path);
    if (static_cast<int32_t>(p.size() / 2) != need) return SFA_EINVAL;
    memcpy(pairs, p.data(), sizeof(int32_t) * p.size());
    return need;
}

int sfa_sam_row(char *buf, size_t cap, const sfa_result_t *r, const char *read_id, const char *rname, const sfa_event_t *events,
                int64_t qstart, int64_t qend, const float *ref_array, int32_t ref_len, int32_t ref_st_offset, uint32_t flag) {
    if (!buf || !r || !read_id || !rname || !events || !ref_array || qend <= qstart || !r->valid || r->rid < 0) return SFA_EINVAL;
    const bool rna = (flag & SFA_RNA) != 0;
    const sfa::WarpPath path = path_of_row(r, events, qstart, qend, ref_array, ref_len, ref_st_offset, flag);
    if (path.px.empty()) return SFA_EINVAL;
    const std::string line = sfa::sam_record(*r, path, read_id, rname, events, qstart, qend, rna);
    if (line.size() + 1 > cap) return SFA_ERANGE;
    memcpy(buf, line.c_str(), line.size() + 1);
    return static_cast<int>(line.size());
}

int sfa_read_kmer_model(const char *path, float *levels, uint32_t *k) {
    if (!path || !levels || !k) return SFA_EINVAL;
    std::vector<float> lv;
    std::string err, warn;
    if (!sfa::read_kmer_model(path, &lv, k, &err, &warn)) {
        sfa_set_error_(err.c_str());
        return SFA_EINVAL;
    }
    memcpy(levels, lv.data(), sizeof(float) * lv.size());
    sfa_set_error_(warn.c_str());  // rows the reference would only have logged (src/model.c:98-100); empty when there were none
    return SFA_OK;
}

struct sfa_blow5 {
    sfa::Blow5Reader reader;
    sfa::Blow5Record rec;
};

sfa_blow5_t *sfa_blow5_open(const char *path) {
    sfa_blow5 *f = new sfa_blow5();
    bool ok = false;
    try {
        ok = path && f->reader.open(path);
        if (!ok) sfa_set_error_(path ? f->reader.error().c_str() : "null path");
    } catch (const std::exception &e) {  // e.g. bad_alloc on a corrupt size field: no exception crosses the C boundary
        sfa_set_error_((std::string("malformed BLOW5: ") + e.what()).c_str());
    }
    if (!ok) {
        delete f;
        return nullptr;
    }
    return f;
}

const char *sfa_blow5_attr(sfa_blow5_t *f, const char *key) { return (f && key) ? f->reader.attr(key) : nullptr; }

int sfa_blow5_next(sfa_blow5_t *f, const char **read_id, double meta[4], const int16_t **raw, int64_t *n_raw) {
    if (!f) return SFA_EINVAL;
    int rc;
    try {
        rc = f->reader.next(&f->rec);
        if (rc < 0) sfa_set_error_(f->reader.error().c_str());
    } catch (const std::exception &e) {
        sfa_set_error_((std::string("malformed BLOW5: ") + e.what()).c_str());
        rc = -1;
    }
    if (rc <= 0) return rc;
    if (read_id) *read_id = f->rec.read_id.c_str();
    if (meta) {
        meta[0] = f->rec.digitisation;
        meta[1] = f->rec.offset;
        meta[2] = f->rec.range;
        meta[3] = f->rec.sampling_rate;
    }
    if (raw) *raw = f->rec.raw.data();
    if (n_raw) *n_raw = static_cast<int64_t>(f->rec.raw.size());
    return 1;
}

void sfa_blow5_close(sfa_blow5_t *f) { delete f; }

int sfa_blow5_select_shard(sfa_blow5_t *f, int32_t r, int32_t G) {
    if (!f || G < 1 || r < 0 || r >= G) return SFA_EINVAL;
    if (f->reader.select_shard(static_cast<uint32_t>(r), static_cast<uint32_t>(G))) return SFA_OK;
    sfa_set_error_(f->reader.error().c_str());
    return SFA_EINVAL;
}

int sfa_blow5_select_records(sfa_blow5_t *f, int64_t first, int64_t count) {
    if (!f || first < 0) return SFA_EINVAL;
    if (f->reader.select_records(static_cast<uint64_t>(first), count < 0 ? UINT64_MAX : static_cast<uint64_t>(count))) return SFA_OK;
    sfa_set_error_(f->reader.error().c_str());
    return SFA_EINVAL;
}

int64_t sfa_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap) {
    if (!in || (!out && cap)) return SFA_EINVAL;
    thread_local std::vector<uint8_t> buf;
    size_t len = 0;
    if (!sfa::fast_inflate_zlib(in, n, &buf, &len)) return SFA_EINVAL;
    if (len > cap) return SFA_ERANGE;
    if (len) memcpy(out, buf.data(), len);
    return static_cast<int64_t>(len);
}

int sfa_inflate_zlib_pair(const uint8_t *in0, size_t n0, const uint8_t *in1, size_t n1, uint8_t *out0, size_t cap0, uint8_t *out1,
                          size_t cap1, int64_t len[2]) {
    if (!in0 || !in1 || !len || (!out0 && cap0) || (!out1 && cap1)) return SFA_EINVAL;
    thread_local std::vector<uint8_t> b0, b1;
    const uint8_t *const in[2] = {in0, in1};
    const size_t n[2] = {n0, n1};
    std::vector<uint8_t> *const bufs[2] = {&b0, &b1};
    uint8_t *const outs[2] = {out0, out1};
    const size_t caps[2] = {cap0, cap1};
    size_t got[2] = {0, 0};
    bool ok[2];
    sfa::fast_inflate_zlib_pair(in, n, bufs, got, ok);
    for (int k = 0; k < 2; ++k) {
        if (!ok[k]) {
            len[k] = SFA_EINVAL;
        } else if (got[k] > caps[k]) {
            len[k] = SFA_ERANGE;
        } else {
            if (got[k]) memcpy(outs[k], bufs[k]->data(), got[k]);
            len[k] = static_cast<int64_t>(got[k]);
        }
    }
    return SFA_OK;
}

}  // extern "C"
