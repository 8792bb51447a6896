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
// LANE (64 reads per wave).  What a lane-per-read loop must not do is touch HBM itself (64 lanes, 64 different
// reads: every access its own cache line, every iteration a memory round trip).  Both kernels therefore stage
// TILES of 64 reads x 32 samples through LDS: the wave loads a tile with coalesced accesses (half a wave per read),
// every lane then runs its sequential recurrence over its row of the tile out of LDS, and results leave the same
// way.  The t-statistic and the event statistics are data parallel (one block per read).
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
    int32_t *seq_flag;      // [n] written by ev_prefix_par_kernel: 1 = this read needs the sequential prefix sums
    int32_t use_flags;      // 0: ev_prefix_kernel sums every read; 1: only the flagged ones
    int32_t *peak_flag;     // [n] written by ev_peaks_spec_kernel: 1 = this read needs the sequential peak picker
    int32_t use_peak_flags; // 0: ev_peaks_kernel walks every read; 1: only the flagged ones
    float thr1, thr2, peak_height;
};

constexpr int kEvTile = 32;  // samples per read staged per round (one half-wave covers one read's slice)

// The 64 reads of a wave, as every lane needs them in the cooperative phases: iteration `it` of a half-wave touches
// read r = 2*it + half, and the lane keeps that read's offset (relative to the wave's first read, 32 bits) and length
// in registers -- the inner loops then cost one add, one compare and the access itself.
struct TileReads {
    uint32_t rel[32];
    int32_t len[32];
    int64_t b0;    // sample offset of the wave's first read (wave-uniform)
    int32_t maxn;  // longest read of the wave (wave-uniform)
    __device__ __forceinline__ void load(const EvArgs &a, int lane, int64_t *lds_b, int32_t *lds_n, bool *live, int64_t *b_out,
                                         int32_t *n_out, bool flagged_only = false) {
        const int i = blockIdx.x * 64 + lane;
        *live = i < a.n_reads && !(flagged_only && a.seq_flag[i] == 0);
        const int last = a.n_reads - 1;
        *b_out = a.raw_off[i < a.n_reads ? i : last];
        *n_out = *live ? static_cast<int32_t>(a.raw_off[i + 1] - *b_out) : 0;
        lds_b[lane] = *b_out;
        lds_n[lane] = *n_out;
        int m = *n_out;
        for (int o = 32; o; o >>= 1) m = max(m, __shfl_xor(m, o));
        maxn = m;
        __syncthreads();
        b0 = lds_b[0];
        const int half = lane >> 5;
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            rel[it] = static_cast<uint32_t>(lds_b[2 * it + half] - b0);
            len[it] = lds_n[2 * it + half];
        }
    }
};

// event_single(): pA = ((float)raw + offset) * raw_unit; compute_sum_sumsq(): sequential double prefix sums, the
// square being a FLOAT product that is promoted afterwards (src/events.c:297-307).  One read per lane, LDS tiles.
__global__ void __launch_bounds__(64) ev_prefix_kernel(const EvArgs a) {
    __shared__ float raw_t[64][kEvTile + 1];
    __shared__ double s_t[64][kEvTile + 1];
    __shared__ double q_t[64][kEvTile + 1];
    __shared__ int64_t lds_b[64];
    __shared__ int32_t lds_n[64];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    bool live;
    int64_t b;
    int32_t n;
    TileReads tr;
    tr.load(a, lane, lds_b, lds_n, &live, &b, &n, a.use_flags != 0);
    const float off = live ? a.scale[2 * i] : 0.0f, unit = live ? a.scale[2 * i + 1] : 0.0f;
    if (live) {
        a.sum[b + i] = 0.0;
        a.sumsq[b + i] = 0.0;
    }
    const int16_t *rawp = a.raw + tr.b0;
    double *sump = a.sum + tr.b0 + static_cast<int64_t>(blockIdx.x) * 64 + 1;  // read i owns [raw_off[i] + i, ...)
    double *sqp = a.sumsq + tr.b0 + static_cast<int64_t>(blockIdx.x) * 64 + 1;
    double acc = 0.0, acc2 = 0.0;
    const int half = lane >> 5, j = lane & 31;
    // coalesced: 32 consecutive samples of read r per half-wave; all loads of a tile are in flight together, and the
    // next tile is fetched while this one is summed.  A position past the end of its read is fetched from sample 0 and
    // yields garbage sums that are never stored (a read's tail is its last tile).
    float v[32];
    auto fetch = [&](int base) {
        const int bj = base + j;
#pragma unroll
        for (int it = 0; it < 32; ++it) v[it] = static_cast<float>(rawp[bj < tr.len[it] ? tr.rel[it] + static_cast<uint32_t>(bj) : 0u]);
    };
    fetch(0);
    for (int base = 0; base < tr.maxn; base += kEvTile) {
#pragma unroll
        for (int it = 0; it < 32; ++it) raw_t[2 * it + half][j] = v[it];
        if (base + kEvTile < tr.maxn) fetch(base + kEvTile);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < kEvTile; ++jj) {
            const float pa = (raw_t[lane][jj] + off) * unit;
            const float sq = pa * pa;
            acc = acc + static_cast<double>(pa);
            acc2 = acc2 + static_cast<double>(sq);
            s_t[lane][jj] = acc;
            q_t[lane][jj] = acc2;
        }
        __syncthreads();
        const int bj = base + j;
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            if (bj < tr.len[it]) {
                const uint32_t o = tr.rel[it] + static_cast<uint32_t>(bj + 2 * it + half);
                sump[o] = s_t[2 * it + half][j];
                sqp[o] = q_t[2 * it + half][j];
            }
        }
        __syncthreads();
    }
}

// The same prefix sums with the 64 lanes of a wave on ONE read -- allowed whenever the additions cannot round.  All
// addends are integer multiples of the ulp q of the smallest one; if the sum of their magnitudes stays below 2^52 q,
// every partial sum, in ANY order, is an integer multiple of q below 2^53 q, i.e. exactly representable: no addition
// rounds, the parallel scan and the reference's sequential loop both return the true sums, bit for bit.  Real signal
// passes (pA values span a few binades, 10^4..10^6 samples); a read that does not is flagged and left to
// ev_prefix_kernel.  One wave per read: pass 1 = the certificate, pass 2 = the scan, 64 samples per step.
__device__ __forceinline__ double wave_inclusive_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o);
        v = lane >= o ? v + u : v;
    }
    return v;
}

__global__ void __launch_bounds__(64) ev_prefix_par_kernel(const EvArgs a) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int64_t b = a.raw_off[i];
    const int32_t n = static_cast<int32_t>(a.raw_off[i + 1] - b);
    const float off = a.scale[2 * i], unit = a.scale[2 * i + 1];
    const int16_t *raw = a.raw + b;
    // ---- pass 1: smallest exponent among the non-zero addends, sum of magnitudes ----
    int e_pa = 255, e_sq = 255;  // biased exponents; 255 = "none seen"
    bool finite = true;
    double m_pa = 0.0, m_sq = 0.0;
    for (int32_t j = lane; j < n; j += 64) {
        const float pa = (static_cast<float>(raw[j]) + off) * unit;
        const float sq = pa * pa;
        const int xp = static_cast<int>((__float_as_uint(pa) >> 23) & 0xffu), xs = static_cast<int>((__float_as_uint(sq) >> 23) & 0xffu);
        finite = finite && xp != 255 && xs != 255;
        if (pa != 0.0f) e_pa = min(e_pa, max(xp, 1));  // denormals: exponent field 0, ulp 2^-149 like field 1
        if (sq != 0.0f) e_sq = min(e_sq, max(xs, 1));
        m_pa += fabs(static_cast<double>(pa));
        m_sq += static_cast<double>(sq);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        e_pa = min(e_pa, __shfl_xor(e_pa, o));
        e_sq = min(e_sq, __shfl_xor(e_sq, o));
        m_pa += __shfl_xor(m_pa, o);
        m_sq += __shfl_xor(m_sq, o);
    }
    finite = __all(finite);
    // ulp of a float with biased exponent e is 2^(e-150); bound 2^52 ulp (one bit of slack covers the rounding of m_*)
    const bool exact = finite && m_pa < ldexp(1.0, e_pa - 98) && m_sq < ldexp(1.0, e_sq - 98);
    if (lane == 0) a.seq_flag[i] = exact ? 0 : 1;
    if (!exact) return;
    // ---- pass 2: scan ----
    double *s = a.sum + b + i, *q = a.sumsq + b + i;
    if (lane == 0) {
        s[0] = 0.0;
        q[0] = 0.0;
    }
    double carry = 0.0, carry2 = 0.0;
    for (int32_t base = 0; base < n; base += 64) {
        const int32_t j = base + lane;
        float pa = 0.0f;
        if (j < n) pa = (static_cast<float>(raw[j]) + off) * unit;
        const float sq = pa * pa;
        const double v = wave_inclusive_scan(static_cast<double>(pa), lane) + carry;
        const double v2 = wave_inclusive_scan(static_cast<double>(sq), lane) + carry2;
        if (j < n) {
            s[j + 1] = v;
            q[j + 1] = v2;
        }
        carry = __shfl(v, 63);
        carry2 = __shfl(v2, 63);
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

// short_long_peak_detector() + the event boundaries of create_events(), src/events.c:375-508.  One read per lane over
// LDS tiles of the two t-statistics; an event START is written as soon as its closing peak fires, the statistics of
// the events follow in ev_stats_kernel.
struct PeakDet32 {  // PeakDet with 32-bit positions (a read has < 2^31 samples)
    float threshold;
    int window;
    int masked_to;
    int peak_pos;  // -1: none
    float peak_value;
    bool valid;
};

// TWO LANES PER READ: the even lane runs the short detector, the odd lane the long one, so the wave executes the state
// machine once per sample instead of twice (its branches diverge anyway) and covers 32 reads.  The detectors are
// coupled one way -- at sample j the short one may mask the long one before the long one looks at j -- so the odd lane
// runs ONE SAMPLE BEHIND: in iteration m the short detector handles sample m, the long one sample m-1, and the short
// lane's mask for sample m reaches the long lane (one DPP move) after that.  The reference's order of side effects,
// S(0) L(0) S(1) L(1) ..., becomes L(m-1) S(m) inside an iteration; both lanes keep identical copies of the event
// counter and of the open event's start, exchanging their peaks with a second DPP move.
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true); }

__global__ void __launch_bounds__(64) ev_peaks_kernel(const EvArgs a) {
    __shared__ float t_t[2][32][kEvTile + 1];
    __shared__ int64_t lds_b[32];
    __shared__ int32_t lds_n[32];
    const int lane = threadIdx.x;
    const int r = lane >> 1, k = lane & 1;  // read of the block, detector
    const int i = blockIdx.x * 32 + r;
    const bool live = i < a.n_reads && !(a.use_peak_flags && a.peak_flag[i] == 0);
    if (lane < 32) {
        const int ii = blockIdx.x * 32 + lane;
        const bool lv = ii < a.n_reads && !(a.use_peak_flags && a.peak_flag[ii] == 0);  // done by ev_peaks_spec_kernel
        const int64_t bb = a.raw_off[ii < a.n_reads ? ii : a.n_reads - 1];
        lds_b[lane] = bb;
        lds_n[lane] = lv ? static_cast<int32_t>(a.raw_off[ii + 1] - bb) : 0;
    }
    __syncthreads();
    const int32_t n = lds_n[r];
    int maxn = n;
    for (int o = 32; o; o >>= 1) maxn = max(maxn, __shfl_xor(maxn, o));
    const int64_t b0 = lds_b[0];
    const int half = lane >> 5, jl = lane & 31;
    uint32_t rel[16];
    int32_t len[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {  // cooperative phase: iteration `it` of a half-wave covers read 2*it + half
        rel[it] = static_cast<uint32_t>(lds_b[2 * it + half] - b0);
        len[it] = lds_n[2 * it + half];
    }
    const int64_t eo = live ? a.ev_off[i] : 0;
    const int ecap = live ? static_cast<int>(a.ev_off[i + 1] - eo) : 0;
    int32_t *evs = a.ev_start + eo;
    const float *t1p = a.t1 + b0, *t2p = a.t2 + b0;
    PeakDet32 p = k ? PeakDet32{a.thr2, a.w2, 0, -1, 3.402823466e+38f, false} : PeakDet32{a.thr1, a.w1, 0, -1, 3.402823466e+38f, false};
    int nev = 0, last = 0;  // events opened so far, start of the open event (same in both lanes of a pair)
    float carry = 0.0f;     // the long detector's sample from the previous tile
    // all loads of a tile in flight together; the next tile is fetched while the detectors walk this one
    float v0[16], v1[16];
    auto fetch = [&](int base) {
        const int bj = base + jl;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const uint32_t o = bj < len[it] ? rel[it] + static_cast<uint32_t>(bj) : 0u;
            v0[it] = t1p[o];
            v1[it] = t2p[o];
        }
    };
    fetch(0);
    for (int base = 0; base < maxn + 1; base += kEvTile) {  // + 1: the long detector's last sample
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            t_t[0][2 * it + half][jl] = v0[it];
            t_t[1][2 * it + half][jl] = v1[it];
        }
        if (base + kEvTile < maxn + 1) fetch(base + kEvTile);
        __syncthreads();
        float c[kEvTile];  // this lane's statistic over the tile
#pragma unroll
        for (int jj = 0; jj < kEvTile; ++jj) c[jj] = t_t[k][r][jj];
#pragma unroll
        for (int jj = 0; jj < kEvTile; ++jj) {
            const int j = base + jj - k;  // the sample this lane looks at
            const float cur = k ? (jj ? c[jj ? jj - 1 : 0] : carry) : c[jj];
            int peak = -1, mask_to = -1;  // what this lane's detector did: a peak to emit, a mask for the long detector
            // the reference's nested ifs (src/events.c:375-446) as selects: with one detector per lane this is 18 % faster
            // than branches (the two-detectors-per-lane version was the other way round)
            {
                const bool active = j >= 0 && j < n && p.masked_to < j;
                const bool searching = p.peak_pos == -1;
                const bool lower = cur < p.peak_value;
                const bool rise = !lower && (cur - p.peak_value > a.peak_height);
                const bool higher = cur > p.peak_value;
                const float b_value = higher ? cur : p.peak_value;
                const int b_pos = higher ? j : p.peak_pos;
                const bool tracking = active && !searching;
                mask_to = (tracking && k == 0 && b_value > p.threshold) ? b_pos + p.window : -1;
                const bool v2 = p.valid || (b_value - cur > a.peak_height && b_value > p.threshold);
                const bool fire = tracking && v2 && (j - b_pos) > p.window / 2;
                peak = (fire && b_pos > 0 && b_pos < n) ? b_pos : -1;
                const float s_value = (lower || rise) ? cur : p.peak_value;
                const int s_pos = rise ? j : -1;
                p.peak_value = active ? (searching ? s_value : (fire ? cur : b_value)) : p.peak_value;
                p.peak_pos = active ? (searching ? s_pos : (fire ? -1 : b_pos)) : p.peak_pos;
                p.valid = tracking ? (fire ? false : v2) : p.valid;
            }
            const int other_peak = pair_swap(peak), other_mask = pair_swap(mask_to);
            const int peak_long = k ? peak : other_peak, peak_short = k ? other_peak : peak;
            if (peak_long >= 0 && nev < ecap) {  // L(m-1) comes before S(m)
                if (k == 1) evs[nev] = last;
                ++nev;
                last = peak_long;
            }
            if (peak_short >= 0 && nev < ecap) {
                if (k == 0) evs[nev] = last;
                ++nev;
                last = peak_short;
            }
            if (k == 1 && other_mask >= 0) {  // the short detector's mask for its sample m = j + 1, before the long one gets there
                p.masked_to = other_mask;
                p.peak_pos = -1;
                p.peak_value = 3.402823466e+38f;
                p.valid = false;
            }
        }
        carry = c[kEvTile - 1];
        __syncthreads();
    }
    if (!live || k != 0) return;
    if (nev > 0 && nev < ecap) {  // the last event runs to the end of the signal; no peak at all -> no events
        evs[nev] = last;
        ++nev;
    } else if (nev >= ecap) {
        nev = 0;
    }
    a.n_events[i] = nev;
}

// ---------------------------------------------------------------------------------------------------------
// The peak picker with the 64 lanes of a wave on ONE read: lane l walks samples [l*C, (l+1)*C) from a GUESSED state (the
// initial one), and the result is accepted only where it is provably the sequential one:
//   * whenever the short detector fires at sample j for a peak at pk, the complete state after that sample is a
//     function of (j, pk, t1[j]) alone: the short detector restarts from t1[j], and the long one was masked and reset
//     by the short one in that very sample (a firing short detector stands above its threshold: src/events.c:411-425);
//   * so lane l, once its own state is known to be true, keeps walking into lane l+1's samples until it fires the short
//     detector at a (j, pk) that lane l+1's speculative walk fired too: from there on lane l+1's walk IS the sequential
//     one (induction from lane 0, whose start is the true start).  What lane l+1 emitted before that point is replaced
//     by what lane l emitted while catching up.
// A read where some lane finds no such point inside the next chunk (or overflows its list) is flagged and left to the
// sequential kernel; so are very short and very long reads.
constexpr int kSpecCap = 96;        // emissions a lane can hold (own chunk + catching up)
constexpr int kSpecMinChunk = 24;   // samples per lane below which speculation is not worth it
constexpr int kSpecBias = 1024;     // list entries hold (position - chunk start + kSpecBias): peaks up to 1024 samples in front of a chunk

struct Det2 {
    PeakDet32 d[2];
};

// one sample through both detectors, in the reference's order (short, then long); a fired peak is returned if
// create_events() would keep it (0 < pk < n), else -1.  stop_after_short: leave the long detector alone.
__device__ __forceinline__ void det2_step(Det2 &D, const EvArgs &a, int n, int j, float c0, float c1, int &pk_short, int &pk_long,
                                          bool &short_fired, int &short_pk) {
    pk_short = -1;
    pk_long = -1;
    short_fired = false;
    short_pk = -1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        PeakDet32 &p = D.d[k];
        if (p.masked_to >= j) continue;
        const float cur = k ? c1 : c0;
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
                D.d[1].masked_to = p.peak_pos + p.window;
                D.d[1].peak_pos = -1;
                D.d[1].peak_value = 3.402823466e+38f;
                D.d[1].valid = false;
            }
            if (p.peak_value - cur > a.peak_height && p.peak_value > p.threshold) p.valid = true;
            if (p.valid && (j - p.peak_pos) > p.window / 2) {
                const int pk = p.peak_pos;
                if (k == 0) {
                    short_fired = true;
                    short_pk = pk;
                }
                if (pk > 0 && pk < n) (k ? pk_long : pk_short) = pk;
                p.peak_pos = -1;
                p.peak_value = cur;
                p.valid = false;
            }
        }
    }
}

__global__ void __launch_bounds__(64) ev_peaks_spec_kernel(const EvArgs a) {
    // per lane: the kept peaks in emission order, and for every firing of the short detector its sample, its peak and the
    // number of list entries before it.  Positions are kept RELATIVE to the lane's chunk (+ kSpecBias: a peak may lie a window in
    // front of it) in 16 bits -- a lane never looks beyond two chunks of at most 3 * kSpecCap samples -- which makes the lists
    // 30 KB per wave instead of 61: five waves per CU instead of two for a kernel whose lanes chase their own strided loads.
    __shared__ unsigned short e_pk[64][kSpecCap];
    __shared__ unsigned short f_j[64][kSpecCap / 2], f_pk[64][kSpecCap / 2], f_at[64][kSpecCap / 2];
    __shared__ int n_own[64], n_fire[64], sync_from[64];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int64_t b = a.raw_off[i];
    const int n = static_cast<int>(a.raw_off[i + 1] - b);
    const int C = (n + 63) / 64;
    if (C < kSpecMinChunk || C > 3 * kSpecCap) {  // too short to split, or too long for the lists: sequential kernel
        if (lane == 0) a.peak_flag[i] = 1;
        return;
    }
    const float *t1 = a.t1 + b, *t2 = a.t2 + b;
    const int c0 = min(n, lane * C), c1 = min(n, c0 + C);
    const int rel0 = lane * C - kSpecBias;  // what this lane's list entries are relative to (lane * C, not c0: lane l - 1 derives it too)
    Det2 D;
    D.d[0] = PeakDet32{a.thr1, a.w1, 0, -1, 3.402823466e+38f, false};
    D.d[1] = PeakDet32{a.thr2, a.w2, 0, -1, 3.402823466e+38f, false};
    bool fail = false;
    int cnt = 0, fires = 0;
    // ---- phase 1: own chunk from the guessed state ----
    for (int j = c0; j < c1; ++j) {
        int ps, pl, spk;
        bool sf;
        det2_step(D, a, n, j, t1[j], t2[j], ps, pl, sf, spk);
        if (sf) {
            if (fires < kSpecCap / 2 && spk >= rel0) {
                f_j[lane][fires] = static_cast<unsigned short>(j - rel0);
                f_pk[lane][fires] = static_cast<unsigned short>(spk - rel0);
                f_at[lane][fires] = static_cast<unsigned short>(cnt);
                ++fires;
            } else {
                fail = true;  // (list full, or a peak further in front of the chunk than the bias: the sequential kernel takes the read)
            }
        }
        if (ps >= 0) {
            if (ps < rel0) fail = true;
            if (cnt < kSpecCap) e_pk[lane][cnt] = static_cast<unsigned short>(ps - rel0);
            ++cnt;
        }
        if (pl >= 0) {
            if (pl < rel0) fail = true;
            if (cnt < kSpecCap) e_pk[lane][cnt] = static_cast<unsigned short>(pl - rel0);
            ++cnt;
        }
    }
    fail = fail || cnt > kSpecCap;
    n_own[lane] = cnt;
    n_fire[lane] = fires;
    sync_from[lane] = 0;  // lane 0 starts from the true state: everything it emitted counts
    __syncthreads();
    // ---- phase 2: catch up into the next lane's chunk until the two walks agree on a firing of the short detector ----
    const bool has_next = lane < 63 && c1 < n;  // (a chunk that starts at n is empty)
    if (has_next && !fail) {
        const int nl = lane + 1;
        const int nf = n_fire[nl];
        const int end = min(n, c1 + C);
        const int reln = nl * C - kSpecBias;  // lane l + 1's origin
        int m = 0;  // next candidate firing of lane l+1
        bool synced = false;
        for (int j = c1; j < end && !synced; ++j) {
            int ps, pl, spk;
            bool sf;
            det2_step(D, a, n, j, t1[j], t2[j], ps, pl, sf, spk);
            while (m < nf && static_cast<int>(f_j[nl][m]) + reln < j) ++m;
            if (sf && m < nf && static_cast<int>(f_j[nl][m]) + reln == j && static_cast<int>(f_pk[nl][m]) + reln == spk) {
                sync_from[nl] = f_at[nl][m];  // lane l+1's list counts from the entry of this firing on (or the next one kept)
                synced = true;
                break;                        // the long detector's turn at this sample belongs to lane l+1's walk
            }
            if (ps >= 0) {
                if (ps < rel0) fail = true;
                if (cnt < kSpecCap) e_pk[lane][cnt] = static_cast<unsigned short>(ps - rel0);
                ++cnt;
            }
            if (pl >= 0) {
                if (pl < rel0) fail = true;
                if (cnt < kSpecCap) e_pk[lane][cnt] = static_cast<unsigned short>(pl - rel0);
                ++cnt;
            }
        }
        fail = fail || !synced || cnt > kSpecCap;
    }
    if (__any(fail)) {
        if (lane == 0) a.peak_flag[i] = 1;
        return;
    }
    __syncthreads();
    // ---- phase 3: lane l contributes its own entries from its agreed point on, then what it emitted while catching up ----
    const int from = sync_from[lane], own = n_own[lane];
    const int mine = (c0 < n ? own - from : 0) + (cnt - own);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    const int total = __shfl(incl, 63);
    const int64_t eo = a.ev_off[i];
    int32_t *evs = a.ev_start + eo;  // capacity n + 2 > total + 1
    int w = 1 + incl - mine;         // event e starts at the (e-1)-th peak, event 0 at sample 0
    if (c0 < n)
        for (int k = from; k < own; ++k) evs[w++] = static_cast<int>(e_pk[lane][k]) + rel0;
    for (int k = own; k < cnt; ++k) evs[w++] = static_cast<int>(e_pk[lane][k]) + rel0;
    if (lane == 0) {
        evs[0] = 0;
        a.n_events[i] = total ? total + 1 : 0;  // the last event runs to the end of the signal; no peak at all -> no events
        a.peak_flag[i] = 0;
    }
}

// create_event(), src/events.c:461-477: event e of a read spans [start_e, start_{e+1}) (the last one ends with the
// signal).  One block per read, one event per thread.
__global__ void __launch_bounds__(256) ev_stats_kernel(const EvArgs a) {
    const int i = blockIdx.x;
    const int64_t b = a.raw_off[i], n = a.raw_off[i + 1] - b;
    const double *s = a.sum + b + i, *q = a.sumsq + b + i;
    const int64_t eo = a.ev_off[i];
    const int nev = a.n_events[i];
    for (int e = threadIdx.x; e < nev; e += 256) {
        const int64_t start = a.ev_start[eo + e];
        const int64_t end = (e + 1 < nev) ? static_cast<int64_t>(a.ev_start[eo + e + 1]) : n;
        const float length = static_cast<float>(end - start);
        const float mean = static_cast<float>(s[end] - s[start]) / length;
        const float dsq = static_cast<float>(q[end] - q[start]);
        const float var = dsq / length - mean * mean;
        a.ev_length[eo + e] = length;
        a.ev_mean[eo + e] = mean;
        a.ev_stdv[eo + e] = sqrtf(fmaxf(var, 0.0f));
    }
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

// One wave per read: the window's means are staged in LDS with coalesced loads, lane 0 accumulates the two sequential
// fp32 sums out of LDS (their order is what makes the result the reference's), all lanes normalise and store.
constexpr int kQueryStage = 2048;  // = SFA_MAX_QUERY events; longer windows (row strips) are read from HBM directly

__global__ void __launch_bounds__(64) ev_query_kernel(const QueryArgs a) {
    __shared__ float m[kQueryStage];
    __shared__ float stat[2];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int64_t o = a.q_off[i];
    const int len = static_cast<int>(a.q_off[i + 1] - o);
    if (len <= 0) return;
    const float *src = a.ev_mean + a.ev_off[i] + a.qstart[i];
    const bool staged = len <= kQueryStage;  // block-uniform
    if (staged)
        for (int j = lane; j < len; j += 64) m[j] = src[j];
    __syncthreads();
    if (lane == 0) {
        const float cnt = static_cast<float>(len);
        float mean = 0.0f, var = 0.0f;
        if (staged) {
            for (int j = 0; j < len; ++j) mean += m[j];
        } else {
            for (int j = 0; j < len; ++j) mean += src[j];
        }
        mean /= cnt;
        for (int j = 0; j < len; ++j) {
            const float dv = (staged ? m[j] : src[j]) - mean;
            var += dv * dv;
        }
        var /= cnt;
        stat[0] = mean;
        stat[1] = static_cast<float>(sqrt(static_cast<double>(var)));
    }
    __syncthreads();
    const float mean = stat[0], sd = stat[1];
    for (int j = lane; j < len; j += 64) a.queries[o + j] = ((staged ? m[j] : src[j]) - mean) / sd;
}

// The query window's event table in the layout of event_t / sfa_event_t (src/sigfish.h:57-64), means z-normalised, for
// hosts that need it (SAM output rebuilds the warp path and the signal segments from it).  One block per read.
struct PackArgs {
    const int32_t *ev_start;
    const float *ev_length, *ev_stdv;
    const int64_t *ev_off;
    const int64_t *qstart;
    const int64_t *q_off;
    const float *queries;  // normalised means, packed at q_off
    uint64_t *out;         // [n][query_size] records of 24 bytes: u64 start, f32 length, mean, stdv, pad
    int32_t query_size;
};

__global__ void __launch_bounds__(128) ev_pack_events_kernel(const PackArgs a) {
    const int i = blockIdx.x;
    const int64_t o = a.q_off[i], len = a.q_off[i + 1] - o;
    const int64_t e0 = a.ev_off[i] + a.qstart[i];
    char *dst = reinterpret_cast<char *>(a.out) + static_cast<int64_t>(i) * a.query_size * 24;
    for (int64_t e = threadIdx.x; e < len; e += 128) {
        char *rec = dst + e * 24;
        *reinterpret_cast<uint64_t *>(rec) = static_cast<uint64_t>(a.ev_start[e0 + e]);
        float *f = reinterpret_cast<float *>(rec + 8);
        f[0] = a.ev_length[e0 + e];
        f[1] = a.queries[o + e];
        f[2] = a.ev_stdv[e0 + e];
        f[3] = 0.0f;
    }
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
