// Would the y axis in Winograd F(2,3) form as well pay on this chip?  The tap stream of that kernel in
// isolation -- no halo DMA, no input transform, no barriers, no epilogue: the UPPER bound of what the form can deliver -- next to the shipped
// two-group x-only stream in the same harness, both with their REAL weight-fragment streams out of L2 (panel footprints of dc2 / dc5) and their
// real LDS fragment reads.
//
// Resource arithmetic per CU and 16-channel chunk (block = 4 z x 8 y x 8 x = 256 voxels; fp16x3 = three MFMA passes per product):
//                                   x-only, two cout groups (shipped)         x + y, ONE cout group (all the accumulators allow)
//   couts per workgroup             128                                       64      (16 frequencies x 64 tiles x 64 couts = 128 acc registers x 8 waves)
//   MFMA pipe cycles per SIMD       2 waves x 448 x 16 = 14 336               2 waves x (96 x 16 + 96 x 8) = 4 608   (per 64 couts: 7 168 vs 4 608, -36 %)
//   weight bytes from L2            2 x 4 f x 40 KiB = 320 KiB (22 B/clk)     16 f x 12 KiB = 192 KiB (42 B/clk: 1.9 x the rate per pipe cycle)
//   T image (LDS)                   60 KB                                     96 KB  (+ 38 KB raw box: no room for a second T, i.e. no overlap of
//                                                                                     the transform with the taps)
//   input transform per 64 couts    1/2 x (480 units x ~190 VALU)             ~155 k lane-ops (2-D: joins, 16 + 16 adds per tile, 24.6 k splits)
//                                                                             = ~300 VALU per thread = 2 400 issue cycles per SIMD, serial with the taps
// V0 = the shipped stream (five steps of tap pairs on v_mfma_f32_16x16x32_f16, 8 x 4 tiles per wave), V1 = x + y with the lone third dz tap as a padded
// K = 32 step ([a0 | a1].[b0 | b0], [a0 | a1].[b1 | 0]: 5 K-32 passes where 4.5 are exact), V2 = x + y with the lone tap on v_mfma_f32_16x16x16_f16
// (exact 4.5, if that legacy shape runs at half the K-32 shape's cycles).  LOADS = 0: the same streams without their global loads.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/wino_xy_taps.hip -o /tmp/wino_xy_taps && /tmp/wino_xy_taps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 gl16(const unsigned char* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x2 gl8(const unsigned char* p) {
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// panel: `panel_bytes` of fragments, walked sequentially by every wave from its own offset (wave w of block b starts at the slab of its frequency /
// cout group, like the real kernels: all blocks of a launch stream the SAME panel, so it is L2- / Infinity-Cache-resident)
template <int V, bool LOADS>
__global__ void __launch_bounds__(512, 1) taps(const float4* __restrict__ lds_init, const unsigned char* __restrict__ panel, size_t panel_bytes, float* out,
                                               int chunks, unsigned long long* stamps) {
    constexpr int TREC = V == 0 ? 3840 : 6144;                        // 16-byte units x 4 per record: 60 KB (x only) / 96 KB (x + y) of T
    __shared__ float4 T[TREC * 4 / 4 * 1];
    for (int i = threadIdx.x; i < TREC; i += 512) T[i] = lds_init[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[8][4];                                                  // 128 accumulator registers either way
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    // this wave's slab of the panel: per chunk V0 40 KiB (5 steps x 8 fragments), V1 32 KiB in instructions (2 f x 2 steps x 8 fragments), V2 24 KiB
    constexpr size_t kPerChunk = V == 0 ? 5 * 8 * 1024 : V == 1 ? 2 * 2 * 8 * 1024 : 2 * (8 * 1024 + 8 * 512);
    constexpr size_t kSlab = kPerChunk * 12;                         // one wave's fragments for a block of 12 chunks (dc2), then from the head again
    const size_t nslabs = panel_bytes / kSlab > 0 ? panel_bytes / kSlab : 1;
    const unsigned char* wp0 = panel + ((size_t)(blockIdx.x & 1) * 8 + wave) % nslabs * kSlab;      // (the prefetch runs <= 8 KiB past a slab: the allocation has slack)
    const unsigned char* wp = wp0;
    auto lda = [&](int rec) __attribute__((always_inline)) { return T[(rec % (TREC / 64)) * 64 + lane]; };
    auto mma32 = [&](const float4& a, const f32x4& b, f32x4& c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
    if constexpr (V == 0) {
        f32x4 bx[2][8];                                               // [set][X' 0..3 | Y' 0..3]
        auto req = [&](f32x4 (&d)[8]) __attribute__((always_inline)) {
            if (LOADS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) d[q] = gl16(wp + q * 1024 + lane * 16);
            }
            wp += 8 * 1024;
        };
        for (int q = 0; q < 8; ++q) bx[0][q] = bx[1][q] = __builtin_bit_cast(f32x4, lds_init[q * 64 + lane]);
        req(bx[0]);
        // five steps per chunk is odd: the set a chunk starts with alternates, so the chunk loop is unrolled by two (an asm load whose result is never
        // read -- a request into the "wrong" set -- would have its registers re-used while it is in flight)
        auto chunk = [&](int ch, auto ptag) __attribute__((always_inline)) {
            constexpr int P = decltype(ptag)::value;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                f32x4 (&cur)[8] = bx[(P + j) & 1];
                if (LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                __builtin_amdgcn_sched_barrier(0);
                req(bx[(P + j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                float4 a0[8];                                                             // (a1 takes a0's registers behind pass A, as in the shipped kernel)
#pragma unroll
                for (int m = 0; m < 8; ++m) a0[m] = lda(ch * 7 + j * 13 + m * 5 + wave);
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma32(a0[m], cur[4 + n], acc[m][n]);      // B: a0 . Y'
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma32(a0[m], cur[n], acc[m][n]);          // A: a0 . X'
                __builtin_amdgcn_sched_barrier(0);
                if (j < 4) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) a0[m] = lda(ch * 11 + j * 3 + m * 7 + wave + 17);
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma32(a0[m], cur[n], acc[m][n]);      // C: a1 . X'
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int ch = 0; ch < chunks; ch += 2) {
            if (ch % 12 == 0) wp = wp0 + 8 * 1024;                    // (a block's 12 chunks, then the next block starts at the panel's head again; one set is in flight)
            chunk(ch, std::integral_constant<int, 0>{});
            chunk(ch + 1, std::integral_constant<int, 1>{});
        }
    } else {
        // x + y: acc[f * 4 + slice][n]; per frequency a PAIR step (taps dz 0, 1: B, A, C passes, 48 MFMAs) and the LONE third tap
        f32x4 bx[2][8];
        f32x2 by[2][8];                                               // V2: the lone tap's b0 / b1 fragments (8 bytes per lane)
        auto req = [&](f32x4 (&d)[8]) __attribute__((always_inline)) {
            if (LOADS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) d[q] = gl16(wp + q * 1024 + lane * 16);
            }
            wp += 8 * 1024;
        };
        auto req8 = [&](f32x2 (&d)[8]) __attribute__((always_inline)) {
            if (LOADS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) d[q] = gl8(wp + q * 512 + lane * 8);
            }
            wp += 8 * 512;
        };
        for (int q = 0; q < 8; ++q) { bx[0][q] = bx[1][q] = __builtin_bit_cast(f32x4, lds_init[q * 64 + lane]); by[0][q] = by[1][q] = f32x2{bx[0][q][0], bx[0][q][1]}; }
        req(bx[0]);
        for (int ch = 0; ch < chunks; ++ch) {
            if (ch % 12 == 0) wp = wp0 + 8 * 1024;                    // (the set in flight was requested from the slab's end or, the first time, from its head)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                // ---- pair step (set 0)
                {
                    f32x4 (&cur)[8] = bx[0];
                    if (LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (V == 1) req(bx[1]); else req8(by[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    float4 a0[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) a0[m] = lda(ch * 7 + f * 29 + m * 5 + wave);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma32(a0[m], cur[4 + n], acc[f * 4 + m][n]);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma32(a0[m], cur[n], acc[f * 4 + m][n]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) a0[m] = lda(ch * 11 + f * 31 + m * 7 + wave + 17);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma32(a0[m], cur[n], acc[f * 4 + m][n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- lone step
                if constexpr (V == 1) {
                    f32x4 (&cur)[8] = bx[1];
                    if (LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                    __builtin_amdgcn_sched_barrier(0);
                    req(bx[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    float4 a[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[m] = lda(ch * 13 + f * 37 + m * 3 + wave + 5);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma32(a[m], cur[n], acc[f * 4 + m][n]);       // [a0 | a1] . [b0 | b0]
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma32(a[m], cur[4 + n], acc[f * 4 + m][n]);   // [a0 | a1] . [b1 | 0]
                } else {
                    f32x2 (&cur)[8] = by[0];
                    if (LOADS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
                    __builtin_amdgcn_sched_barrier(0);
                    req(bx[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    float2 a0[4], a1[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float4 t = lda(ch * 13 + f * 37 + m * 3 + wave + 5);                // one 16-byte read: [a0 (8 B) | a1 (8 B)] of this lane's K-4 slice
                        a0[m] = float2{t.x, t.y}; a1[m] = float2{t.z, t.w};
                    }
                    auto mma16 = [&](const float2& a, const f32x2& b, f32x4& c) __attribute__((always_inline)) {
                        c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
                    };
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma16(a0[m], cur[4 + n], acc[f * 4 + m][n]);  // a0 . b1
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma16(a0[m], cur[n], acc[f * 4 + m][n]);      // a0 . b0
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma16(a1[m], cur[n], acc[f * 4 + m][n]);      // a1 . b0
                }
            }
        }
    }
    if (LOADS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 100 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

int main() {
    const size_t kPanel = 16u << 20;                                  // bytes of fragments available
    std::vector<_Float16> h(4096 * 8 + kPanel / 2);
    srand(1);
    for (size_t i = 0; i < h.size(); i += 16)
        for (int c = 0; c < 8; ++c) {
            const float x = ((float)rand() / RAND_MAX - 0.5f) * 4000.0f;
            const _Float16 hi = (_Float16)x;
            h[i + c] = hi;
            h[i + 8 + c] = (_Float16)(x - (float)hi);
        }
    float4* init; unsigned char* panel; float* out; unsigned long long* stamps;
    hipMalloc(&stamps, 16); hipMalloc(&init, 4096 * 16); hipMalloc(&panel, kPanel + (1u << 20)); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(init, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipMemcpy(panel, h.data() + 4096 * 8, kPanel, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chunks = 12 * 2000;                                     // 4000 "blocks" of dc2's 12 chunks per workgroup
    struct Var { const char* name; void (*k)(const float4*, const unsigned char*, size_t, float*, int, unsigned long long*); size_t panel; double couts, pipe_cyc; };
    // panel footprints: dc2 = 64 couts x 12 chunks: x-only 4 f x 40 KiB x 12 = 1.9 MB, x + y 16 f x (16 | 12) KiB x 12 = 3.1 | 2.4 MB (one L2 holds it);
    // dc5 = 128 couts x 24 chunks: 7.7 / 12.6 / 9.4 MB (Infinity Cache)
    const Var vars[] = {
        {"V0 x-only two-group stream (hipcc spills 0.5 KB here; the shipped kernel does not), dc2 panel", taps<0, true>, 4u << 20, 128, 2.0 * 448 * 16},
        {"V0 x-only, dc5-size panel", taps<0, true>, 8u << 20, 128, 2.0 * 448 * 16},
        {"V0 x-only, no global loads", taps<0, false>, 4u << 20, 128, 2.0 * 448 * 16},
        {"V1 x+y, lone tap padded to K 32, dc2-size panel", taps<1, true>, 3u << 20, 64, 2.0 * 160 * 16},
        {"V1 x+y, lone tap padded to K 32, dc5-size panel", taps<1, true>, 12u << 20, 64, 2.0 * 160 * 16},
        {"V1 x+y, lone tap padded to K 32, no global loads", taps<1, false>, 3u << 20, 64, 2.0 * 160 * 16},
        {"V2 x+y, lone tap on 16x16x16, dc2-size panel", taps<2, true>, 5u << 19, 64, 2.0 * (96 * 16 + 96 * 8)},
        {"V2 x+y, lone tap on 16x16x16, dc5-size panel", taps<2, true>, 9u << 20, 64, 2.0 * (96 * 16 + 96 * 8)},
        {"V2 x+y, lone tap on 16x16x16, no global loads", taps<2, false>, 5u << 19, 64, 2.0 * (96 * 16 + 96 * 8)},
    };
    for (int round = 0; round < 2; ++round)
        for (const Var& v : vars) {
            float ms = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                v.k<<<256, 512>>>(init, panel, v.panel, out, chunks, stamps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned long long st[2]; hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost);
            const double clk = (double)st[0] / (double)st[1] * 0.1;   // GHz
            const double us_chunk = ms * 1e3 / chunks;
            // ns per chunk and 64 couts of a 256-voxel block: the comparable figure (V0 does 128 couts per chunk)
            printf("%-60s %7.1f ms  %6.3f us/chunk  %6.3f us per chunk and 64 couts  clock %.3f GHz  MFMA-busy %.2f (if 16x16x16 = 8 cyc)\n", v.name, ms, us_chunk,
                   us_chunk * 64.0 / v.couts, clk, v.pipe_cyc / (us_chunk * 1e-6 * clk * 1e9));
        }
    return 0;
}
