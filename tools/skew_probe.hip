// skew_probe.hip -- what would the fill gain from a ROW-skewed anti-diagonal?  ("one wave alone runs at
// half rate".)
//
// The shipped sweep skews by LANE: lane g of a read is one column behind lane g - 1, and the R rows a lane holds are evaluated
// one after the other at the SAME column -- cell r waits for cell r - 1 (v_min3 -> v_add -> v_min3 ...), and the neighbour's
// bottom cell makes an LDS round trip between two steps.  A lone wave therefore issues 0.25 VALU instructions per cycle where
// six to eight waves together reach 0.45.
// Skewed by ROW, global row i works on column t - i: all three neighbours of a cell were computed in EARLIER steps (up: row
// i - 1 one step ago, diagonal: row i - 1 two steps ago, left: row i one step ago), so the R cells of a step are independent,
// the bottom row can be evaluated first and sent at once, the top row last -- the exchange has a whole step to arrive.
// Price: two generations of state (2 R registers), a ring of R reference levels per lane (row r is r columns behind row 0 of
// its lane) -- free of moves when R steps are unrolled --, R x lanes - lanes more steps per sweep (+0.8 % at 29 903 columns),
// and snapshots that hold a staircase instead of a straight column front.
//
// Both formulations here: 4 reads of 256 events per wave (16 lanes x 16 rows, the throughput shape), subsequence DTW costs
// only, the last row's minimum and a checksum of its bit patterns per read as the result -- which must agree bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o skew_probe skew_probe.hip ; run: ./skew_probe [columns]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int R = 16, L = 16, Q = R * L, PAD = 512;

struct Out {
    float best;
    unsigned sum;
};

typedef __attribute__((address_space(3))) volatile float lds_f;
struct __attribute__((aligned(4))) float4u {  // four consecutive levels from any 4-byte boundary (one global_load_dwordx4)
    float v[4];
};

// ---- shipped formulation: rows of a lane in sequence at one column ----
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) sweep_lane_skew(const float *q, const float *y, int ncols, Out *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane & 15, slot = lane >> 4;
    const int read = (blockIdx.x * 4 + wv) * 4 + slot;
    __shared__ float lds[4 * 80];
    float *wr = lds + wv * 80 + slot * 17 + g + 1, *rd = lds + wv * 80 + slot * 17 + g;
    if (g == 0) *((lds_f *)rd) = 0.0f;  // free start above query row 0
    float x[R], c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        x[r] = q[read * Q + g * R + r];
        c[r] = INFINITY;
    }
    float dprev = (g == 0) ? 0.0f : INFINITY;
    float best = INFINITY;
    unsigned sum = 0;
    const float *yp = y + PAD - g;  // lane g is at column t - g
    const int steps = ncols + L - 1;
    for (int t0 = 0; t0 < steps; t0 += 4) {
        const float4u yv = *reinterpret_cast<const float4u *>(yp + t0);
        const float ys[4] = {yv.v[0], yv.v[1], yv.v[2], yv.v[3]};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *((lds_f *)wr) = c[R - 1];
            float up = *((lds_f *)rd);
            float diag = dprev;
            dprev = up;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float left = c[r];
                const float m = fminf(fminf(up, diag), left);
                const float cn = fabsf(x[r] - ys[u]) + m;
                diag = left;
                up = cn;
                c[r] = cn;
            }
            const int col = t0 + u - (L - 1);
            if (g == L - 1 && col >= 0 && col < ncols) {
                best = fminf(best, c[R - 1]);
                sum += __float_as_uint(c[R - 1]) * 2654435761u + static_cast<unsigned>(col);
            }
        }
    }
    if (g == L - 1) {
        out[read].best = best;
        out[read].sum = sum;
    }
}

// ---- row skew: global row i = g R + r at column t - i ----
// One step, K = t mod R (compile time: the ring of levels is indexed statically).  GEN: which of the two state arrays holds the
// step before (0: a, 1: b); the new values go into the other one, bottom row first.
template <int K>
__device__ __forceinline__ void row_skew_step(float (&a)[R], float (&b)[R], const float (&x)[R], const float (&yr)[R], float &u1, float &u2,
                                              float *wr, float *rd) {
    float(&p1)[R] = (K & 1) ? b : a;  // one step ago
    float(&p2)[R] = (K & 1) ? a : b;  // two steps ago; receives the new values
#pragma unroll
    for (int r = R - 1; r >= 1; --r) {
        const float m = fminf(fminf(p1[r - 1], p2[r - 1]), p1[r]);
        p2[r] = fabsf(x[r] - yr[(K - r) & (R - 1)]) + m;
        if (r == R - 1) *((lds_f *)wr) = p2[r];  // sent as soon as it exists
    }
    const float m0 = fminf(fminf(u1, u2), p1[0]);
    p2[0] = fabsf(x[0] - yr[K]) + m0;
    u2 = u1;
    u1 = *((lds_f *)rd);  // the neighbour's bottom cell of THIS step: first used at the end of the next one
}

template <int K0>
__device__ __forceinline__ void row_skew_steps4(float (&a)[R], float (&b)[R], const float (&x)[R], float (&yr)[R], float &u1, float &u2, float *wr,
                                                float *rd, const float *yp, int t0, bool lastlane, int ncols, float &best, unsigned &sum, bool g0) {
    const float4u yv = *reinterpret_cast<const float4u *>(yp + t0);
    // (a level enters the ring at ITS step: slot K still holds the level of 16 steps ago, which row 15 needs until then)
#define SKEW_ONE(KK)                                                                        \
    yr[KK] = yv.v[(KK) - K0];                                                               \
    row_skew_step<KK>(a, b, x, yr, u1, u2, wr, rd);                                         \
    {                                                                                       \
        const int col = t0 + (KK - K0) - (Q - 1);                                           \
        const float v = ((KK) & 1) ? a[R - 1] : b[R - 1];                                   \
        if (lastlane && col >= 0 && col < ncols) {                                          \
            best = fminf(best, v);                                                          \
            sum += __float_as_uint(v) * 2654435761u + static_cast<unsigned>(col);           \
        }                                                                                   \
    }
    SKEW_ONE(K0)
    SKEW_ONE(K0 + 1)
    SKEW_ONE(K0 + 2)
    SKEW_ONE(K0 + 3)
#undef SKEW_ONE
}

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) sweep_row_skew(const float *q, const float *y, int ncols, Out *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane & 15, slot = lane >> 4;
    const int read = (blockIdx.x * 4 + wv) * 4 + slot;
    __shared__ float lds[4 * 80];
    float *wr = lds + wv * 80 + slot * 17 + g + 1, *rd = lds + wv * 80 + slot * 17 + g;
    if (g == 0) *((lds_f *)rd) = 0.0f;  // free start above query row 0: lane 0 reads this word every step
    float x[R], a[R], b[R], yr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        x[r] = q[read * Q + g * R + r];
        a[r] = INFINITY;
        b[r] = INFINITY;
        yr[r] = INFINITY;
    }
    const bool g0 = g == 0;
    float u1 = g0 ? 0.0f : INFINITY, u2 = u1;  // lane 0: the free start above query row 0
    float best = INFINITY;
    unsigned sum = 0;
    const float *yp = y + PAD - g * R;  // row 0 of lane g is at column t - g R
    const int steps = ncols + Q - 1;
    const bool lastlane = g == L - 1;
    for (int t0 = 0; t0 < steps; t0 += R) {
        row_skew_steps4<0>(a, b, x, yr, u1, u2, wr, rd, yp, t0, lastlane, ncols, best, sum, g0);
        row_skew_steps4<4>(a, b, x, yr, u1, u2, wr, rd, yp, t0 + 4, lastlane, ncols, best, sum, g0);
        row_skew_steps4<8>(a, b, x, yr, u1, u2, wr, rd, yp, t0 + 8, lastlane, ncols, best, sum, g0);
        row_skew_steps4<12>(a, b, x, yr, u1, u2, wr, rd, yp, t0 + 12, lastlane, ncols, best, sum, g0);
    }
    if (lastlane) {
        out[read].best = best;
        out[read].sum = sum;
    }
}

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <int WAVES, int WAVES_ROW = WAVES>
int run(const char *what, int n_reads, const float *dq, const float *dy, int ncols, Out *dout, std::vector<Out> &h0, std::vector<Out> &h1) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid(n_reads / 16), block(256);
    float ms[2] = {0, 0};
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<Out> &h = variant ? h1 : h0;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (variant == 0)
                hipLaunchKernelGGL(sweep_lane_skew<WAVES>, grid, block, 0, 0, dq, dy, ncols, dout);
            else
                hipLaunchKernelGGL(sweep_row_skew<WAVES_ROW>, grid, block, 0, 0, dq, dy, ncols, dout);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms[variant], e0, e1));
        }
        CHECK(hipMemcpy(h.data(), dout, sizeof(Out) * n_reads, hipMemcpyDeviceToHost));
    }
    int bad = 0;
    for (int i = 0; i < n_reads; ++i) bad += h0[i].sum != h1[i].sum || memcmp(&h0[i].best, &h1[i].best, 4) != 0;
    const double cells = static_cast<double>(n_reads) * Q * ncols;
    printf("%-28s %6d reads: lane skew %7.3f ms (%.3e cells/s)   row skew %7.3f ms (%.3e cells/s)   x%.2f   reads that differ: %d\n", what, n_reads,
           ms[0], cells / ms[0] * 1e3, ms[1], cells / ms[1] * 1e3, ms[0] / ms[1], bad);
    return bad;
}

int main(int argc, char **argv) {
    const int ncols = argc > 1 ? atoi(argv[1]) : 29903;
    const int max_reads = 16 * 1024 * 8;
    std::vector<float> hq(static_cast<size_t>(max_reads) * Q), hy(ncols + 2 * PAD, INFINITY);
    srand(7);
    for (float &v : hq) v = (rand() % 2001 - 1000) / 400.0f;
    for (int j = 0; j < ncols; ++j) hy[PAD + j] = (rand() % 2001 - 1000) / 400.0f;
    float *dq, *dy;
    Out *dout;
    CHECK(hipMalloc(&dq, hq.size() * 4));
    CHECK(hipMalloc(&dy, hy.size() * 4));
    CHECK(hipMalloc(&dout, sizeof(Out) * max_reads));
    CHECK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    std::vector<Out> h0(max_reads), h1(max_reads);
    int bad = 0;
    // 1024 SIMDs: 256 blocks of four waves = one wave per SIMD = 4 096 reads
    bad += run<1>("1 wave per SIMD", 4096, dq, dy, ncols, dout, h0, h1);
    bad += run<2>("2 waves per SIMD", 8192, dq, dy, ncols, dout, h0, h1);
    bad += run<4>("4 waves per SIMD", 16384, dq, dy, ncols, dout, h0, h1);
    // (the row-skewed kernel needs 106 VGPRs: four waves per SIMD; the lane-skewed one runs six, as the shipped fill does)
    bad += run<6, 4>("2 rounds of 6 / 3 of 4", 49152, dq, dy, ncols, dout, h0, h1);
    bad += run<6, 4>("4 rounds of 6 / 6 of 4", 98304, dq, dy, ncols, dout, h0, h1);
    printf(bad ? "MISMATCH\n" : "both formulations agree bit for bit\n");
    return bad != 0;
}
