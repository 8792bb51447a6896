#include "sam.hpp"

#include <cassert>
#include <cmath>
#include <cstdio>
#include <limits>

namespace sfa {

namespace {
inline float low3(float a, float b, float c) {  // src/cdtw.c:25-36
    float m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return m;
}
}  // namespace

WarpPath band_traceback(const float *x, int32_t n, const float *y, int32_t rlen, int32_t col_st, int32_t col_end, bool std_dtw) {
    WarpPath p;
    if (n <= 0 || col_st < 0 || col_end < col_st || col_end >= rlen) return p;
    const int32_t m = col_end - col_st + 1;
    const float inf = std::numeric_limits<float>::infinity();
    std::vector<float> cost(static_cast<size_t>(n) * m);
    auto C = [&](int32_t i, int32_t j) -> float & { return cost[static_cast<size_t>(i) * m + j]; };
    // row 0: free start (subsequence) or the cumulative sum from column 0 (std_dtw, src/cdtw.c:85-86)
    if (!std_dtw) {
        for (int32_t j = 0; j < m; ++j) C(0, j) = std::fabs(x[0] - y[col_st + j]);
    } else {
        float acc = std::fabs(x[0] - y[0]);
        for (int32_t j = 1; j <= col_end; ++j) {
            if (j - 1 >= col_st) C(0, j - 1 - col_st) = acc;
            acc = std::fabs(x[0] - y[j]) + acc;
        }
        C(0, m - 1) = acc;
    }
    for (int32_t i = 1; i < n; ++i) {
        // first band column: the neighbours at column col_st-1 are outside the band (+inf), except that column 0 of
        // the matrix really has no left neighbours (same arithmetic either way: d + up)
        C(i, 0) = std::fabs(x[i] - y[col_st]) + low3(C(i - 1, 0), inf, inf);
        for (int32_t j = 1; j < m; ++j) C(i, j) = std::fabs(x[i] - y[col_st + j]) + low3(C(i - 1, j), C(i - 1, j - 1), C(i, j - 1));
    }
    // path(), src/cdtw.c:98-167, from (n-1, col_end); stops when it reaches row 0 (the trimmed start of
    // subsequence_path, src/cdtw.c:204-221, is the LAST row-0 cell, i.e. the first one the walk meets)
    int32_t i = n - 1, j = m - 1;
    std::vector<int32_t> bx{i}, by{j};
    while (i > 0) {
        if (j == 0) {
            i--;
        } else {
            const float up = C(i - 1, j), dg = C(i - 1, j - 1), lf = C(i, j - 1);
            const float best = low3(up, dg, lf);
            if (dg == best) {
                i--;
                j--;
            } else if (lf == best) {
                j--;
            } else {
                i--;
            }
        }
        bx.push_back(i);
        by.push_back(j);
    }
    p.px.assign(bx.rbegin(), bx.rend());
    p.py.resize(by.size());
    for (size_t k = 0; k < by.size(); ++k) p.py[k] = by[by.size() - 1 - k] + col_st;
    return p;
}

std::vector<int32_t> path_to_pairs(const WarpPath &path) {
    // path_to_map(), src/sigfish.c:530-571: per reference column the first/last query row; a column entered
    // without advancing in the query (horizontal move) is blanked
    const int32_t ref_st = path.py.front();
    const int32_t len = path.py.back() - ref_st + 1;
    std::vector<int32_t> pairs(2 * static_cast<size_t>(len), -1);
    int32_t prev_q = -1;
    for (size_t k = 0; k < path.px.size(); ++k) {
        const int32_t ri = path.py[k] - ref_st, qi = path.px[k];
        if (pairs[2 * ri] == -1) pairs[2 * ri] = qi;
        pairs[2 * ri + 1] = qi;
        if (prev_q == qi) pairs[2 * ri] = pairs[2 * ri + 1] = -1;
        prev_q = qi;
    }
    return pairs;
}

std::string sam_record(const sfa_result_t &row, const WarpPath &path, const char *read_id, const char *rname, const sfa_event_t *ev,
                       int64_t qstart, int64_t qend, bool rna) {
    struct Pair {
        int32_t start, stop;
    };
    const std::vector<int32_t> pairs = path_to_pairs(path);
    const int32_t len = static_cast<int32_t>(pairs.size() / 2);
    std::vector<Pair> map(len);
    for (int32_t i = 0; i < len; ++i) map[i] = Pair{pairs[2 * i], pairs[2 * i + 1]};
    // r2qevent_map_to_ss(), src/sigfish.c:663-768
    if (rna) {
        const int32_t end = map[len - 1].stop;
        for (Pair &m : map)
            if (m.start != -1) {
                m.start = end - m.start;
                m.stop = end - m.stop;
            }
    }
    for (Pair &m : map)
        if (m.start != -1) {
            m.start += static_cast<int32_t>(qstart);
            m.stop += static_cast<int32_t>(qstart);
        }
    if (rna) {
        for (int32_t a = 0; a < len / 2; ++a) std::swap(map[a], map[len - 1 - a]);
        for (Pair &m : map) std::swap(m.start, m.stop);
    }
    std::string ss;
    char tmp[64];
    int64_t ci = 0, mi = 0, d = 0;
    bool first = true;
    for (int32_t jx = 0; jx < len; ++jx) {
        if (map[jx].start == -1) {
            if (!first) d++;
            continue;
        }
        const int64_t sig_st = static_cast<int64_t>(ev[map[jx].start].start);
        first = false;
        const int64_t sig_en = static_cast<int64_t>(ev[map[jx].stop].start) + static_cast<int>(ev[map[jx].stop].length);
        if (d > 0) {
            snprintf(tmp, sizeof tmp, "%dD", static_cast<int>(d));
            ss += tmp;
            d = 0;
        }
        if (jx == 0) ci = sig_st;
        ci += (mi = sig_st - ci);
        if (mi) {
            snprintf(tmp, sizeof tmp, "%dI", static_cast<int>(mi));
            ss += tmp;
        }
        ci += (mi = sig_en - sig_st);
        if (mi) {
            snprintf(tmp, sizeof tmp, "%d,", static_cast<int>(mi));
            ss += tmp;
        }
    }
    // sam_str(), src/sigfish.c:770-794
    const sfa_event_t &e0 = ev[qstart], &e1 = ev[qend - 1];
    const uint64_t start_raw = e0.start;
    const uint64_t end_raw = static_cast<uint64_t>(static_cast<float>(e1.start) + e1.length);
    const uint64_t post_st = rna ? row.pos_end : row.pos_st, post_end = rna ? row.pos_st : row.pos_end;
    char head[1024];
    snprintf(head, sizeof head, "%s\t%d\t%s\t%ld\t%d\t%ldM\t*\t0\t0\t*\t*\tsi:Z:%ld,%ld,%ld,%ld\tss:Z:", read_id, row.strand == '+' ? 0 : 16, rname,
             static_cast<long>(row.pos_st) + 1, static_cast<int>(row.mapq), static_cast<long>((qend - 1) - qstart), static_cast<long>(start_raw),
             static_cast<long>(end_raw), static_cast<long>(post_st), static_cast<long>(post_end));
    return std::string(head) + ss + "\n";
}

}  // namespace sfa
