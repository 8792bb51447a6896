// events.cpp -- see events.hpp.  Compiled with -ffp-contract=off: the reference is built by gcc -O2 -std=c99
// for x86-64 (no FMA contraction, float expressions evaluated in float, FLT_EVAL_METHOD 0).
#include "events.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace sfa {

void raw_to_picoamps(const int16_t *raw, int64_t n, double digitisation, double offset, double range, float *out) {
    const float rangef = static_cast<float>(range), digf = static_cast<float>(digitisation), offf = static_cast<float>(offset);
    const float unit = rangef / digf;
    for (int64_t i = 0; i < n; ++i) out[i] = (static_cast<float>(raw[i]) + offf) * unit;
}

namespace {

struct DetectorParam {
    size_t w1, w2;
    float thr1, thr2, peak_height;
};
// src/events.c:47-58
const DetectorParam kDna = {3, 6, 1.4f, 9.0f, 0.2f};
const DetectorParam kRna = {7, 14, 2.5f, 9.0f, 1.0f};

// windowed two-sample t statistic from prefix sums, src/events.c:319-368
std::vector<float> tstat(const std::vector<double> &sum, const std::vector<double> &sumsq, size_t n, size_t w) {
    std::vector<float> t(n, 0.0f);
    if (n < 2 * w || w < 2) return t;
    const float wf = static_cast<float>(w);
    for (size_t i = w; i <= n - w; ++i) {
        double s1 = sum[i], q1 = sumsq[i];
        if (i > w) {
            s1 -= sum[i - w];
            q1 -= sumsq[i - w];
        }
        const float s2 = static_cast<float>(sum[i + w] - sum[i]);
        const float q2 = static_cast<float>(sumsq[i + w] - sumsq[i]);
        const float mean1 = static_cast<float>(s1 / static_cast<double>(wf));
        const float mean2 = s2 / wf;
        // double accumulation, left to right, exactly as the mixed-type C expression promotes
        double cv = q1 / static_cast<double>(wf);
        cv -= static_cast<double>(mean1 * mean1);
        cv += static_cast<double>(q2 / wf);
        cv -= static_cast<double>(mean2 * mean2);
        float combined = static_cast<float>(cv);
        combined = std::fmax(combined, FLT_MIN);
        const float delta = mean2 - mean1;
        t[i] = static_cast<float>(std::fabs(static_cast<double>(delta)) / std::sqrt(static_cast<double>(combined / wf)));
    }
    return t;
}

struct Detector {  // src/events.c:281-293
    const float *signal;
    float threshold;
    size_t window;
    size_t masked_to = 0;
    int peak_pos = -1;
    float peak_value = FLT_MAX;
    bool valid_peak = false;
};

// short/long detector interplay, src/events.c:375-458
std::vector<size_t> pick_peaks(Detector &sd, Detector &ld, size_t n, float peak_height) {
    std::vector<size_t> peaks;
    Detector *det[2] = {&sd, &ld};
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 2; ++k) {
            Detector &d = *det[k];
            if (d.masked_to >= i) continue;
            const float cur = d.signal[i];
            if (d.peak_pos == -1) {
                if (cur < d.peak_value) {
                    d.peak_value = cur;  // deeper minimum
                } else if (cur - d.peak_value > peak_height) {
                    d.peak_value = cur;  // a qualifying rise: start tracking a peak
                    d.peak_pos = static_cast<int>(i);
                }
            } else {
                if (cur > d.peak_value) {
                    d.peak_value = cur;
                    d.peak_pos = static_cast<int>(i);
                }
                if (k == 0 && d.peak_value > d.threshold) {  // the short detector masks the long one
                    ld.masked_to = d.peak_pos + d.window;
                    ld.peak_pos = -1;
                    ld.peak_value = FLT_MAX;
                    ld.valid_peak = false;
                }
                if (d.peak_value - cur > peak_height && d.peak_value > d.threshold) d.valid_peak = true;
                if (d.valid_peak && (i - static_cast<size_t>(d.peak_pos)) > d.window / 2) {
                    peaks.push_back(static_cast<size_t>(d.peak_pos));
                    d.peak_pos = -1;
                    d.peak_value = cur;
                    d.valid_peak = false;
                }
            }
        }
    }
    return peaks;
}

sfa_event_t make_event(size_t start, size_t end, const std::vector<double> &sum, const std::vector<double> &sumsq) {
    sfa_event_t e;  // src/events.c:461-477
    e.start = start;
    e.length = static_cast<float>(end - start);
    e.mean = static_cast<float>(sum[end] - sum[start]) / e.length;
    const float dsq = static_cast<float>(sumsq[end] - sumsq[start]);
    const float var = dsq / e.length - e.mean * e.mean;
    e.stdv = std::sqrt(std::fmax(var, 0.0f));
    return e;
}

}  // namespace

std::vector<sfa_event_t> detect_events(const float *pa, int64_t n_, bool rna) {
    std::vector<sfa_event_t> out;
    if (n_ <= 0) return out;
    const size_t n = static_cast<size_t>(n_);
    const DetectorParam &p = rna ? kRna : kDna;
    // prefix sums in double; the square is a FLOAT product promoted afterwards (src/events.c:297-307)
    std::vector<double> sum(n + 1), sumsq(n + 1);
    sum[0] = 0.0;
    sumsq[0] = 0.0;
    for (size_t i = 0; i < n; ++i) {
        sum[i + 1] = sum[i] + static_cast<double>(pa[i]);
        sumsq[i + 1] = sumsq[i] + static_cast<double>(pa[i] * pa[i]);
    }
    const std::vector<float> t1 = tstat(sum, sumsq, n, p.w1), t2 = tstat(sum, sumsq, n, p.w2);
    Detector sd{t1.data(), p.thr1, p.w1}, ld{t2.data(), p.thr2, p.w2};
    std::vector<size_t> peaks = pick_peaks(sd, ld, n, p.peak_height);
    // create_events(), src/events.c:479-508: peaks equal to 0 or >= n do not open an event
    std::vector<size_t> cuts;
    for (size_t pk : peaks)
        if (pk > 0 && pk < n) cuts.push_back(pk);
    // the reference indexes the raw peak list positionally; a peak at 0 cannot occur (tstat[0..w) is zero)
    if (cuts.empty()) return out;  // the reference reads out of bounds here; treat as "no events"
    out.reserve(cuts.size() + 1);
    out.push_back(make_event(0, cuts[0], sum, sumsq));
    for (size_t e = 1; e < cuts.size(); ++e) out.push_back(make_event(cuts[e - 1], cuts[e], sum, sumsq));
    out.push_back(make_event(cuts.back(), n, sum, sumsq));
    return out;
}

// ---------------------------------------------------------------------------------------------------------
// adaptor / poly-A segmenters (src/jnn.c), only reached with RNA and -p -1
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Seg {
    int64_t x, y;
};

inline float clamp_outlier(float v) { return v > 1200.0f ? 1200.0f : (v < 0.0f ? 0.0f : v); }  // jnn.c:19-20,49-83

float mean_f(const float *x, int n) {  // stat.h:17-24: float accumulator, divide by int
    float s = 0;
    for (int i = 0; i < n; ++i) s += x[i];
    return s / n;
}
float stdv_f(const float *x, int n) {  // stat.h:36-44
    const float m = mean_f(x, n);
    float s = 0;
    for (int i = 0; i < n; ++i) s += (x[i] - m) * (x[i] - m);
    return std::sqrt(s / n);
}

// jnnv2(), src/jnn.c:100-180: first low-mean stretch of plausible length in a rolling mean = the adaptor
Seg find_adaptor(const int16_t *raw, int64_t n, int pore) {
    const float std_scale = pore == 2 ? 0.7f : 0.5f;     // JNNV2_RNA_RNA004_ADAPTOR / JNNV2_RNA_R9_ADAPTOR
    const int seg_dist = 1500, window = 2000, hi = 200000, lo = pore == 2 ? 500 : 2000;
    if (n <= window) return Seg{-1, -1};
    std::vector<float> cur(n);
    for (int64_t i = 0; i < n; ++i) cur[i] = clamp_outlier(static_cast<float>(raw[i]));
    const int m = static_cast<int>(n) - window;
    std::vector<float> t(m);
    float run = 0.0f;  // rolling_window(), jnn.c:22-46
    for (int i = 0; i < window; ++i) run += cur[i];
    t[0] = run / window;
    for (int i = 1; i < m; ++i) {
        run -= cur[i - 1];
        run += cur[i + window - 1];
        t[i] = run / window;
    }
    const float mn = mean_f(t.data(), m), sd = stdv_f(t.data(), m);
    const float bot = mn - (sd * std_scale);
    std::vector<Seg> segs;
    bool begin = false;
    int start = 0, end = 0;
    for (int j = 0; j < m; ++j) {
        const float v = t[j];
        if (v < bot && !begin) {
            start = j;
            begin = true;
        } else if (v < bot) {
            end = j;
        } else if (v > bot && begin) {
            if (!segs.empty() && start - segs.back().y < seg_dist)
                segs.back().y = end;
            else
                segs.push_back(Seg{start, end});
            start = end = 0;
            begin = false;
        }
    }
    Seg p{0, 0};
    for (const Seg &s : segs) {
        const int a = static_cast<int>(s.x), b = static_cast<int>(s.y);
        if (b - a > hi || b - a < lo) continue;
        p.x = a + window / 2 - 1;
        p.y = b + window / 2 - 1;
        break;
    }
    return p;
}

// jnn_core() with JNNV1_R9_POLYA / JNNV1_RNA004_POLYA (identical), src/jnn.c:191-279: first stretch that stays
// inside (bot, top) for >= window samples, tolerating `error` excursions
Seg find_polya(const float *pa, int64_t n, float top, float bot) {
    const int corrector = 50, seg_dist = 200, window = 250, error = 30;
    const float stall_len = 1.0f;
    std::vector<Seg> segs;
    bool prev = false;
    int err = 0, prev_err = 0, c = 0, w = corrector, start = 0, end = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float a = clamp_outlier(pa[i]);
        if (a < top && a > bot) {
            if (!prev) {
                start = static_cast<int>(i);
                prev = true;
            }
            c++;
            w++;
            if (prev_err) prev_err = 0;
            if (c >= window && c >= w && !(c % w)) err--;
        } else {
            if (prev && err < error) {
                c++;
                err++;
                prev_err++;
                if (c >= window && c >= w && !(c % w)) err--;
            } else if (prev && (c >= window || (segs.empty() && c >= window * stall_len))) {
                end = static_cast<int>(i) - prev_err;
                prev = false;
                if (!segs.empty() && start - segs.back().y < seg_dist)
                    segs.back().y = end;
                else
                    segs.push_back(Seg{start, end});
                c = err = prev_err = 0;
            } else if (prev) {
                prev = false;
                c = err = prev_err = 0;
            }
        }
    }
    return segs.empty() ? Seg{-1, -1} : segs.front();
}

}  // namespace

int64_t detect_query_start(const int16_t *raw, int64_t n, const float *pa, const std::vector<sfa_event_t> &ev, int pore) {
    const Seg ad = find_adaptor(raw, n, pore);
    if (ad.y <= 0) return -1;
    const float m_a = mean_f(pa + ad.x, static_cast<int>(ad.y - ad.x));
    Seg polya = find_polya(pa + ad.y, n - ad.y, m_a + 30 + 20, m_a + 30 - 20);
    if (polya.y <= 0) return -1;
    polya.y += ad.y;
    uint64_t i = 0;
    while (i < ev.size() && ev[i].start < static_cast<uint64_t>(polya.y)) i++;
    return i >= ev.size() ? -1 : static_cast<int64_t>(i);
}

bool select_and_normalise(std::vector<sfa_event_t> &ev, const int16_t *raw, int64_t nraw, const float *pa, int32_t prefix_size,
                          int32_t query_size, uint32_t flag, int pore, int64_t *qstart, int64_t *qend, int *status) {
    const int64_t n = static_cast<int64_t>(ev.size());
    int64_t st, en;
    *status = 0;
    bool keep = true;
    if (!(flag & SFA_END)) {  // src/sigfish.c:435-463
        st = prefix_size;
        if (prefix_size < 0) {
            st = detect_query_start(raw, nraw, pa, ev, pore);
            if (st < 0) {
                *status |= 4;
                st = 50;
            }
        }
        en = st + query_size;
        if (st + 25 > n) {
            st = en = 0;
            keep = false;
            *status |= 2;
        } else if (en > n) {
            en = n;
            *status |= 1;
        }
    } else {  // src/sigfish.c:464-478
        st = n - prefix_size - query_size;
        en = n - prefix_size;
        if (st < 0) {
            st = 0;
            *status |= 1;
        }
        if (en < 0) {
            en = 0;
            keep = false;
            *status |= 2;
        }
    }
    *qstart = st;
    *qend = en;
    if (!keep) return false;
    // z-normalise event[st..en).mean in place (src/sigfish.c:483-502), sequential fp32
    const float cnt = static_cast<float>(en - st);
    float mean = 0.0f, var = 0.0f;
    for (int64_t j = st; j < en; ++j) mean += ev[j].mean;
    mean /= cnt;
    for (int64_t j = st; j < en; ++j) {
        const float d = ev[j].mean - mean;
        var += d * d;
    }
    var /= cnt;
    const float sd = static_cast<float>(std::sqrt(static_cast<double>(var)));
    for (int64_t j = st; j < en; ++j) ev[j].mean = (ev[j].mean - mean) / sd;
    return true;
}

}  // namespace sfa
