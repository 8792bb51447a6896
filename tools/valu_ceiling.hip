// valu_ceiling.hip -- what VALU issue rate does MI355X sustain on the fill kernel's instruction mix?
// Variant A: 16 x (v_sub, v_min3, v_add|d|) per step with the same serial dependency (min3 -> add -> min3 ...)
// Variant B: A + one DPP per step (cross-lane chain)
// Variant C: independent chains (no serial dependency) as the upper bound
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ceiling valu_ceiling.hip ; run: ./valu_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k(const float *in, float *out, int steps) {
    float x[16], c[16];
    for (int r = 0; r < 16; ++r) {
        x[r] = in[threadIdx.x + r * 256];
        c[r] = in[threadIdx.x + r * 256 + 4096];
    }
    __shared__ float lds[4 * 80 + 16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane & 15, row = lane >> 4;
    const bool lane0 = g == 0;
    const int bp_addr = ((lane - 1) & 63) * 4;
    // per wave 4 rows x 17 words: word 0 of a row is a constant zero, lane g writes word g+1 and reads word g
    const int wslot = wv * 80 + row * 17 + g + 1, rslot = wv * 80 + row * 17 + g;
    if (MODE == 5) { lds[wv * 80 + row * 17] = 0.0f; }
    float dprev = in[threadIdx.x];
    float y = in[threadIdx.x + 17];
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float up;
            if (MODE == 1)
                up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[15]), 0x111, 0xF, 0xF, true));
            else if (MODE == 3)  // whole-wave shift
                up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c[15]), 0x138, 0xF, 0xF, true));
            else if (MODE == 4) {  // LDS crossbar gather + select for lane 0 of each row
                const int v = __builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(c[15]));
                up = lane0 ? 0.0f : __int_as_float(v);
            } else if (MODE == 5) {  // LDS round trip: own slot written, neighbour's slot read (slot 0 of a row's window holds 0)
                lds[wslot] = c[15];
                up = lds[rslot];
            } else
                up = c[15];
            float diag = dprev;
            dprev = up;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float left = c[r];
                float m;
                if (MODE == 2)
                    m = fminf(fminf(dprev, diag), left);  // no dependency on the previous row
                else
                    m = fminf(fminf(up, diag), left);
                const float cn = fabsf(x[r] - y) + m;
                diag = left;
                up = cn;
                c[r] = cn;
            }
            y += 1.0f;
        }
    }
    float s = dprev;
    for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int WAVES>
void run(const char *name, const float *din, float *dout) {
    const int steps = 20000, blocks = 256 * WAVES * 4;  // 4 rounds of resident blocks (1 block = 4 waves = 1/SIMD)
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(256), 0, 0, din, dout, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(256), 0, 0, din, dout, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr = double(blocks) * 4 * steps * 4 * (48.0 + (MODE == 1 ? 1 : 0));  // wave-instructions
    printf("%-28s waves/SIMD=%d  %.2f ms  %.3e VALU wave-instr/s  = %.3f per SIMD per cycle @2.4GHz (%.3f @2.2GHz)\n", name, WAVES, ms,
           instr / (ms * 1e-3), instr / (ms * 1e-3) / 1024 / 2.4e9, instr / (ms * 1e-3) / 1024 / 2.2e9);
}

int main() {
    float *din, *dout;
    std::vector<float> h(16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) % 1000) / 500.0f - 1.0f;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 256 * 8 * 4 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 1>("serial chain", din, dout);
    run<0, 2>("serial chain", din, dout);
    run<0, 4>("serial chain", din, dout);
    run<0, 6>("serial chain", din, dout);
    run<0, 8>("serial chain", din, dout);
    run<1, 4>("serial chain + dpp", din, dout);
    run<1, 6>("serial chain + dpp", din, dout);
    run<1, 8>("serial chain + dpp", din, dout);
    run<3, 6>("serial chain + wave_shr dpp", din, dout);
    run<4, 4>("serial chain + bpermute", din, dout);
    run<4, 6>("serial chain + bpermute", din, dout);
    run<4, 8>("serial chain + bpermute", din, dout);
    run<5, 4>("serial chain + lds rt", din, dout);
    run<5, 6>("serial chain + lds rt", din, dout);
    run<5, 8>("serial chain + lds rt", din, dout);
    return 0;
}
