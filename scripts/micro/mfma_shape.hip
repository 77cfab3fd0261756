// Which MFMA shape should the fp16x3 tap loop use at the power wall?  (MI355X_MICROARCH.md, DVFS give-back item 7: bare bf16 loops on random data,
// 16x16x32 delivers ~1.15 x the FLOP/s of 32x32x16 at equal cycles per FLOP.)  This probe has the structure of a conv3_wino_sres tap, not a bare
// loop: 8 waves per CU (2 per SIMD), 128 accumulator registers per wave, A fragments re-read from LDS every tap, weight fragments re-loaded from an
// L2-resident panel every tap, split-fp16 data (high terms O(1000), low terms = the fp16 residual), three products per (a, b) pair:
//   V0  v_mfma_f32_32x32x16_f16: 4 m-tiles x 2 n-tiles; per tap a0.b0, a0.b1, a1.b0 = 24 MFMAs, 8 ds_read_b128, 4 global_load_dwordx4
//   V1  v_mfma_f32_16x16x32_f16: 8 m-tiles x 4 n-tiles; per tap [a0|a0].[b0|b1] = 32 MFMAs (K = 32 carries two of the three products) and, per PAIR of
//       taps, [a1(t)|a1(t+1)].[b0(t)|b0(t+1)] = 32 MFMAs (b0 halves gathered from the two main fragments by v_permlane32_swap);
//       per tap 12 ds_read_b128, 4 global_load_dwordx4, 8 swaps: the same 48 K-16-equivalent MFMA steps per tap (+ the odd ninth tap in the real kernel)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool LOADS, bool AGPR = false>
__global__ void __launch_bounds__(512, 1) tap_loop(const float4* __restrict__ lds_init, const float4* __restrict__ panel, int panel_frags, float* out, int iters, unsigned long long* stamps) {
    __shared__ float4 T[4096];                                        // 64 KB of records (like the 60-KB T image)
    for (int i = threadIdx.x; i < 4096; i += 512) T[i] = lds_init[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (V == 0) {
        f32x16 acc[4][2];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        int frag = (blockIdx.x * 8 + wave) * 97;
        float4 bnext[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n) bnext[k][n] = panel[((frag + k * 2 + n) % panel_frags) * 64 + lane];
        for (int it = 0; it < iters; ++it) {
            float4 b[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int n = 0; n < 2; ++n) b[k][n] = bnext[k][n];
            frag += 4;
            if (LOADS) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int n = 0; n < 2; ++n) bnext[k][n] = panel[((frag + k * 2 + n) % panel_frags) * 64 + lane];
            }
            float4 a[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 4; ++m) a[k][m] = T[((it * 7 + m * 5 + k * 3 + wave) & 63) * 64 + lane];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(__builtin_bit_cast(f32x4, a[0][m])), "v"(__builtin_bit_cast(f32x4, b[p][n])));
                        else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0][m]), __builtin_bit_cast(f16x8, b[p][n]), acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(__builtin_bit_cast(f32x4, a[1][m])), "v"(__builtin_bit_cast(f32x4, b[0][n])));
                    else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1][m]), __builtin_bit_cast(f16x8, b[0][n]), acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    } else if constexpr (V == 2) {
        // tap-pair scheme: K = 32 is (tap t, tap t + 1) x 16 channels for all three passes; operands are natural [x(t) | x(t+1)] records:
        // per pair 16 ds_read_b128 (a0, a1 for 8 m-tiles), 8 fragment loads (asm, one pair ahead, counted wait), 96 MFMAs
        f32x4 acc[8][4];
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
        int frag = (blockIdx.x * 8 + wave) * 97;
        f32x4 bx[2][2][4];                                            // [buffer][X' / Y'][n]
        auto issue = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float4* src = panel + ((frag + k * 4 + n) % panel_frags) * 64 + lane;
                    if (LOADS) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bx[buf][k][n]) : "v"(src) : "memory");
                }
            frag += 8;
        };
        for (int k = 0; k < 2; ++k) for (int n = 0; n < 4; ++n) bx[0][k][n] = bx[1][k][n] = __builtin_bit_cast(f32x4, panel[(k * 4 + n) * 64 + lane]);
        issue(0);
        for (int it = 0; it < iters; it += 4) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                             // pair u uses buffer u, requests buffer u ^ 1
                if (LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bx[u][0][0]), "+v"(bx[u][0][1]), "+v"(bx[u][0][2]), "+v"(bx[u][0][3]), "+v"(bx[u][1][0]), "+v"(bx[u][1][1]), "+v"(bx[u][1][2]), "+v"(bx[u][1][3]));
                issue(u ^ 1);
                float4 a0[8], a1[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) a0[m] = T[(((it + 2 * u) * 7 + m * 5 + wave) & 63) * 64 + lane];
#pragma unroll
                for (int m = 0; m < 8; ++m) a1[m] = T[(((it + 2 * u) * 11 + m * 3 + wave + 32) & 63) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0[m]), __builtin_bit_cast(f16x8, bx[u][k][n]), acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[m]), __builtin_bit_cast(f16x8, bx[u][0][n]), acc[m][n], 0, 0, 0);
            }
        }
        if (LOADS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    } else if constexpr (V == 3) {
        // V0 with hand-placed fragment loads (asm, one tap ahead, counted wait): what the shipped kernel does
        f32x16 acc[4][2];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        int frag = (blockIdx.x * 8 + wave) * 97;
        f32x4 bx[2][2][2];
        auto issue = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float4* src = panel + ((frag + k * 2 + n) % panel_frags) * 64 + lane;
                    if (LOADS) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bx[buf][k][n]) : "v"(src) : "memory");
                }
            frag += 4;
        };
        for (int k = 0; k < 2; ++k) for (int n = 0; n < 2; ++n) bx[0][k][n] = bx[1][k][n] = __builtin_bit_cast(f32x4, panel[(k * 2 + n) * 64 + lane]);
        issue(0);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bx[u][0][0]), "+v"(bx[u][0][1]), "+v"(bx[u][1][0]), "+v"(bx[u][1][1]));
                issue(u ^ 1);
                float4 a[2][4];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[k][m] = T[(((it + u) * 7 + m * 5 + k * 3 + wave) & 63) * 64 + lane];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0][m]), __builtin_bit_cast(f16x8, bx[u][p][n]), acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1][m]), __builtin_bit_cast(f16x8, bx[u][0][n]), acc[m][n], 0, 0, 0);
            }
        }
        if (LOADS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    } else {
        f32x4 acc[8][4];
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
        int frag = (blockIdx.x * 8 + wave) * 97;
        float4 bnext[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bnext[n] = panel[((frag + n) % panel_frags) * 64 + lane];
        for (int it = 0; it < iters; it += 2) {                       // a pair of taps
            float4 bb[2][4];                                          // [tap][n]: lanes 0..31 b0 halves, lanes 32..63 b1 halves
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int n = 0; n < 4; ++n) bb[t][n] = bnext[n];
                frag += 4;
                if (LOADS) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bnext[n] = panel[((frag + n) % panel_frags) * 64 + lane];
                }
                float4 a[8];                                          // [a0 | a0]: lanes 32..63 read what lanes 0..31 read
#pragma unroll
                for (int m = 0; m < 8; ++m) a[m] = T[(((it + t) * 7 + m * 5 + wave) & 63) * 64 + (lane & 31)];
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m]), __builtin_bit_cast(f16x8, bb[t][n]), acc[m][n], 0, 0, 0);
            }
            // [b0(t) | b0(t+1)]: swap the upper half of tap t's register with the lower half of tap t+1's
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float* x = reinterpret_cast<float*>(&bb[0][n]);
                float* y = reinterpret_cast<float*>(&bb[1][n]);
#pragma unroll
                for (int c = 0; c < 4; ++c) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[c]), "+v"(y[c]));
            }
            float4 a1[8];                                             // [a1(t) | a1(t+1)]: one read per lane
#pragma unroll
            for (int m = 0; m < 8; ++m) a1[m] = T[((it * 11 + m * 3 + wave + 32) & 63) * 64 + lane];
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[m]), __builtin_bit_cast(f16x8, bb[0][n]), acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 100 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }      // (read after the kernel; nothing is computed from them)
}

int main() {
    // records: 16-byte units of 8 fp16; even units hold high terms (|x| up to ~2000), odd units the residual low terms
    std::vector<_Float16> h(4096 * 8 + 65536 * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); i += 16)
        for (int c = 0; c < 8; ++c) {
            const float x = ((float)rand() / RAND_MAX - 0.5f) * 4000.0f;
            const _Float16 hi = (_Float16)x;
            h[i + c] = hi;
            h[i + 8 + c] = (_Float16)(x - (float)hi);
        }
    float4 *init, *panel; float* out; unsigned long long* stamps; hipMalloc(&stamps, 16);
    hipMalloc(&init, 4096 * 16); hipMalloc(&panel, 65536 * 16); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(init, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipMemcpy(panel, h.data() + 4096 * 8, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400000;                                         // taps per wave: 24 x 32x32x16 each; ~0.4 s per launch
    struct Var { const char* name; void (*k)(const float4*, const float4*, int, float*, int, unsigned long long*); int pf; };
    const Var vars[] = {
        {"V0 32x32x16, compiler-scheduled fragment loads (sunk to their use)", tap_loop<0, true>, 1024},
        {"V0 32x32x16, no global loads", tap_loop<0, false>, 1024},
        {"V3 32x32x16, asm fragment loads one tap ahead (the shipped form), L2-resident panel", tap_loop<3, true>, 1024},
        {"V3 32x32x16, asm fragment loads one tap ahead, L1-resident panel", tap_loop<3, true>, 8},
        {"V3 32x32x16, no global loads", tap_loop<3, false>, 1024},
        {"V2 16x16x32 tap pairs, asm fragment loads one pair ahead, L2-resident panel", tap_loop<2, true>, 1024},
        {"V2 16x16x32 tap pairs, asm fragment loads one pair ahead, L1-resident panel", tap_loop<2, true>, 8},
        {"V2 16x16x32 tap pairs, no global loads", tap_loop<2, false>, 1024},
        {"V1 16x16x32 [a0|a0].[b0|b1] + swapped pairs, no global loads", tap_loop<1, false>, 1024},
    };
    for (int round = 0; round < 2; ++round)
        for (const Var& v : vars) {
            float ms = 0;
            for (int rep = 0; rep < 6; ++rep) {                       // ~2.5 s per variant: the last repetition is at the settled clock
                hipEventRecord(e0);
                v.k<<<256, 512>>>(init, panel, v.pf, out, iters, stamps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double flops = 256.0 * 8 * iters * 24 * 32.0 * 32 * 16 * 2;     // the same per tap for every variant
            const double cyc = 2.0 * iters * 24 * 32;                             // MFMA pipe cycles per SIMD (two waves)
            unsigned long long st[2]; hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost);
            const double clk = (double)st[0] / (double)st[1] * 0.1;               // GHz: shader cycles per 100-MHz tick
            printf("%-88s %6.1f ms %5.0f TFLOP/s executed, clock %.3f GHz, MFMA-busy %.2f\n", v.name, ms, flops / ms / 1e9, clk, cyc / (ms * 1e-3 * clk * 1e9));
        }
    return 0;
}
