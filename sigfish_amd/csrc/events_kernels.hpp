// events_kernels.hpp -- the pre-DP stages of the `dtw` path on the GPU: raw ADC samples -> pA ->
// events (scrappie's t-statistic peak picker) -> query window -> z-normalised query, device resident end to end.
//
// reference: hasindu2008/sigfish v0.2.0 -- event_single src/sigfish.c:330-378, getevents/detect_events
// src/events.c:297-577, normalise_single src/sigfish.c:424-505.  The CPU twin of this file is host/events.cpp; both
// follow the reference operation by operation (types and evaluation order), and both are tested against the
// events the compiled reference produced.
//
// Bit-exactness dictates the parallelisation: the prefix sums are SEQUENTIAL double additions (a parallel scan
// would round differently), and the peak picker is a data-dependent state machine.  So those two run one READ PER
// LANE (64 reads per wave, thousands of reads in flight); only the t-statistic is data parallel (one block per
// read, one sample per thread).  The work is tiny next to the DTW (a few ms per 100 k reads), so the strided
// access of the lane-per-read kernels is acceptable.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfa {

struct EvArgs {
    const int16_t *raw;     // concatenated samples
    const int64_t *raw_off; // [n+1]
    const float *scale;     // [n][2]: offset, raw_unit = range/digitisation (already fp32, as event_single computes)
    double *sum, *sumsq;    // [total + n]: read i owns [raw_off[i] + i, raw_off[i+1] + i + 1)
    float *t1, *t2;         // [total] t-statistics, short and long window
    // events, capacity per read cap_off[i+1]-cap_off[i] (= n/2 + 2)
    const int64_t *ev_off;  // [n+1]
    int32_t *ev_start;
    float *ev_length, *ev_mean, *ev_stdv;
    int32_t *n_events;      // [n]
    int32_t n_reads;
    int32_t w1, w2;         // detector windows (3,6 DNA / 7,14 RNA)
    float thr1, thr2, peak_height;
};

// event_single(): pA = ((float)raw + offset) * raw_unit; compute_sum_sumsq(): sequential double prefix sums, the
// square being a FLOAT product that is promoted afterwards (src/events.c:297-307).  One read per lane.
__global__ void __launch_bounds__(64) ev_prefix_kernel(const EvArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_reads) return;
    const int64_t b = a.raw_off[i], n = a.raw_off[i + 1] - b;
    const float off = a.scale[2 * i], unit = a.scale[2 * i + 1];
    double *s = a.sum + b + i, *q = a.sumsq + b + i;
    double acc = 0.0, acc2 = 0.0;
    s[0] = 0.0;
    q[0] = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        const float pa = (static_cast<float>(a.raw[b + j]) + off) * unit;
        const float sq = pa * pa;
        acc = acc + static_cast<double>(pa);
        acc2 = acc2 + static_cast<double>(sq);
        s[j + 1] = acc;
        q[j + 1] = acc2;
    }
}

// compute_tstat(), src/events.c:319-368, for one sample
__device__ __forceinline__ float tstat_at(const double *sum, const double *sumsq, int64_t n, int64_t w, int64_t i) {
    if (n < 2 * w || w < 2 || i < w || i > n - w) return 0.0f;
    const float wf = static_cast<float>(w);
    double s1 = sum[i], q1 = sumsq[i];
    if (i > w) {
        s1 -= sum[i - w];
        q1 -= sumsq[i - w];
    }
    const float s2 = static_cast<float>(sum[i + w] - sum[i]);
    const float q2 = static_cast<float>(sumsq[i + w] - sumsq[i]);
    const float mean1 = static_cast<float>(s1 / static_cast<double>(wf));
    const float mean2 = s2 / wf;
    double cv = q1 / static_cast<double>(wf);
    cv -= static_cast<double>(mean1 * mean1);
    cv += static_cast<double>(q2 / wf);
    cv -= static_cast<double>(mean2 * mean2);
    float combined = static_cast<float>(cv);
    combined = fmaxf(combined, 1.17549435e-38f);  // FLT_MIN
    const float delta = mean2 - mean1;
    return static_cast<float>(fabs(static_cast<double>(delta)) / sqrt(static_cast<double>(combined / wf)));
}

// one block per read, one sample per thread (grid-stride over the read)
__global__ void __launch_bounds__(256) ev_tstat_kernel(const EvArgs a) {
    const int i = blockIdx.x;
    const int64_t b = a.raw_off[i], n = a.raw_off[i + 1] - b;
    const double *s = a.sum + b + i, *q = a.sumsq + b + i;
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        a.t1[b + j] = tstat_at(s, q, n, a.w1, j);
        a.t2[b + j] = tstat_at(s, q, n, a.w2, j);
    }
}

// short_long_peak_detector() + create_events(), src/events.c:375-508.  One read per lane; an event is written as soon
// as its closing peak fires.
struct PeakDet {
    float threshold;
    int window;
    int64_t masked_to;
    int64_t peak_pos;  // -1: none
    float peak_value;
    bool valid;
};

__device__ __forceinline__ void emit_event(const EvArgs &a, const double *s, const double *q, int64_t slot, int64_t start, int64_t end) {
    const float length = static_cast<float>(end - start);
    const float mean = static_cast<float>(s[end] - s[start]) / length;  // create_event(), src/events.c:461-477
    const float dsq = static_cast<float>(q[end] - q[start]);
    const float var = dsq / length - mean * mean;
    a.ev_start[slot] = static_cast<int32_t>(start);
    a.ev_length[slot] = length;
    a.ev_mean[slot] = mean;
    a.ev_stdv[slot] = sqrtf(fmaxf(var, 0.0f));
}

__global__ void __launch_bounds__(64) ev_peaks_kernel(const EvArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_reads) return;
    const int64_t b = a.raw_off[i], n = a.raw_off[i + 1] - b;
    const double *s = a.sum + b + i, *q = a.sumsq + b + i;
    const float *sig[2] = {a.t1 + b, a.t2 + b};
    const int64_t eo = a.ev_off[i], ecap = a.ev_off[i + 1] - eo;
    PeakDet d[2];
    d[0] = PeakDet{a.thr1, a.w1, 0, -1, 3.402823466e+38f, false};
    d[1] = PeakDet{a.thr2, a.w2, 0, -1, 3.402823466e+38f, false};
    int64_t nev = 0, last = 0;  // events written, start of the open event
    for (int64_t j = 0; j < n; ++j) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            PeakDet &p = d[k];
            if (p.masked_to >= j) continue;
            const float cur = sig[k][j];
            if (p.peak_pos == -1) {
                if (cur < p.peak_value) {
                    p.peak_value = cur;
                } else if (cur - p.peak_value > a.peak_height) {
                    p.peak_value = cur;
                    p.peak_pos = j;
                }
            } else {
                if (cur > p.peak_value) {
                    p.peak_value = cur;
                    p.peak_pos = j;
                }
                if (k == 0 && p.peak_value > p.threshold) {  // the short detector masks the long one
                    d[1].masked_to = p.peak_pos + p.window;
                    d[1].peak_pos = -1;
                    d[1].peak_value = 3.402823466e+38f;
                    d[1].valid = false;
                }
                if (p.peak_value - cur > a.peak_height && p.peak_value > p.threshold) p.valid = true;
                if (p.valid && (j - p.peak_pos) > p.window / 2) {
                    const int64_t pk = p.peak_pos;
                    if (pk > 0 && pk < n && nev < ecap) {  // create_events() skips peaks at 0 / >= n
                        emit_event(a, s, q, eo + nev, last, pk);
                        ++nev;
                        last = pk;
                    }
                    p.peak_pos = -1;
                    p.peak_value = cur;
                    p.valid = false;
                }
            }
        }
    }
    if (nev > 0 && nev < ecap) {  // the last event runs to the end of the signal; no peak at all -> no events
        emit_event(a, s, q, eo + nev, last, n);
        ++nev;
    } else if (nev >= ecap) {
        nev = 0;
    }
    a.n_events[i] = static_cast<int32_t>(nev);
}

// normalise_single(), src/sigfish.c:483-502 + the query extraction of dtw_single (857-867, reversal is done by the
// fill kernels): z-normalise event means [qstart,qend) with sequential fp32 sums, pack them at q_off.  One read per lane.
struct QueryArgs {
    const float *ev_mean;
    const int64_t *ev_off;
    const int64_t *qstart;  // [n] (host-chosen window; qend-qstart = q_off[i+1]-q_off[i])
    const int64_t *q_off;   // [n+1]
    float *queries;
    int32_t n_reads;
};

__global__ void __launch_bounds__(64) ev_query_kernel(const QueryArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_reads) return;
    const int64_t o = a.q_off[i], len = a.q_off[i + 1] - o;
    if (len <= 0) return;
    const float *m = a.ev_mean + a.ev_off[i] + a.qstart[i];
    const float cnt = static_cast<float>(len);
    float mean = 0.0f, var = 0.0f;
    for (int64_t j = 0; j < len; ++j) mean += m[j];
    mean /= cnt;
    for (int64_t j = 0; j < len; ++j) {
        const float dv = m[j] - mean;
        var += dv * dv;
    }
    var /= cnt;
    const float sd = static_cast<float>(sqrt(static_cast<double>(var)));
    for (int64_t j = 0; j < len; ++j) a.queries[o + j] = (m[j] - mean) / sd;
}

// raw-signal coordinates of the query for the PAF columns 3-4 (aln_to_str, src/sigfish.c:800-805)
struct BoundsArgs {
    const int32_t *ev_start;
    const float *ev_length;
    const int64_t *ev_off;
    const int64_t *qstart;
    const int64_t *q_off;
    int32_t *first_start;  // [n] event[qstart].start
    int32_t *last_start;   // [n] event[qend-1].start
    float *last_length;    // [n] event[qend-1].length
    int32_t n_reads;
};

__global__ void __launch_bounds__(256) ev_bounds_kernel(const BoundsArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_reads) return;
    const int64_t len = a.q_off[i + 1] - a.q_off[i];
    if (len <= 0) {
        a.first_start[i] = 0;
        a.last_start[i] = 0;
        a.last_length[i] = 0.0f;
        return;
    }
    const int64_t e0 = a.ev_off[i] + a.qstart[i], e1 = e0 + len - 1;
    a.first_start[i] = a.ev_start[e0];
    a.last_start[i] = a.ev_start[e1];
    a.last_length[i] = a.ev_length[e1];
}

}  // namespace sfa
