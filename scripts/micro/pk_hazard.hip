// Standalone reproducer of profiles/r02_packed_fp32_hazard.md (MI355X, ROCm 7.2): a packed-fp32 VALU op that reads an operand across
// halves (op_sel) returns a ZERO multiplicand in lanes 48..63 of its wave while ANOTHER wave on the chip runs MFMAs whose B operand comes
// straight from a global_load_dwordx4.  No memory in the victim: every lane evaluates the packed op and the same arithmetic with scalar
// ops and counts disagreements (a packed op is two IEEE ops: there is no rounding to disagree about).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/pk_hazard.hip -o /tmp/pk_hazard && /tmp/pk_hazard [rounds]      (hit rates: profiles/r03_packed_fp32_hazard.md)
//   -shared -fPIC -DPK_SO: exports pk_victim() for scripts/micro/pk_hazard_real.py (the victim beside liboai_hip.so's conv kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: MFMA on register operands.  MODE 1: the B operand is reloaded from global memory every step (what a GEMM / conv kernel does)
template <int MODE>
__global__ void __launch_bounds__(256, 2) aggressor(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    f16x8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        if (MODE) b = in[(threadIdx.x + 31 * it) & 511];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// counts[f]: lanes x iterations where form f disagrees with scalar ops; counts[8 + lane / 16]: by 16-lane group; counts[12]: wrong
// results that equal the addend alone (product term zero); counts[13]: scalar-vs-scalar control
__global__ void __launch_bounds__(256) victim(unsigned* __restrict__ counts, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x2 x = {seed + 0.001f * threadIdx.x, 0.5f - 0.002f * threadIdx.x}, y = {0.75f - 0.0005f * threadIdx.x, 0.125f + 0.003f * threadIdx.x};
    for (int it = 0; it < iters; ++it) {
        f32x2 f, g0, g1, g2, g3;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(f) : "v"(x), "v"(y));                                                    // plain packed op (never wrong)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(g0) : "v"(x), "v"(f), "v"(y));                                       // {x0 f0 + y0, x1 f1 + y1}
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(g1) : "v"(x), "v"(f), "v"(y));   // {x0 f1 + y0, x1 f1 + y0}  <- hipcc's SLP form in chain_kernel
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(g2) : "v"(x), "v"(f), "v"(y));   // {x1 f0 + y0, x1 f1 + y1}
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(g3) : "v"(x), "v"(f));                      // {x0 f1, x1 f1}
        float r[8], c;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[0]) : "v"(x[0]), "v"(f[0]), "v"(y[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[1]) : "v"(x[1]), "v"(f[1]), "v"(y[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[2]) : "v"(x[0]), "v"(f[1]), "v"(y[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[3]) : "v"(x[1]), "v"(f[1]), "v"(y[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[4]) : "v"(x[1]), "v"(f[0]), "v"(y[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[6]) : "v"(x[0]), "v"(f[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[7]) : "v"(x[1]), "v"(f[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c) : "v"(x[0]), "v"(f[0]), "v"(y[0]));
        const bool b0 = g0[0] != r[0] || g0[1] != r[1], b1 = g1[0] != r[2] || g1[1] != r[3], b2 = g2[0] != r[4] || g2[1] != r[1], b3 = g3[0] != r[6] || g3[1] != r[7];
        if (b0) atomicAdd(&counts[0], 1u);
        if (b1) { atomicAdd(&counts[1], 1u); if (g1[0] == y[0] || g1[1] == y[0]) atomicAdd(&counts[12], 1u); }
        if (b2) atomicAdd(&counts[2], 1u);
        if (b3) atomicAdd(&counts[3], 1u);
        if (b0 || b1 || b2 || b3) atomicAdd(&counts[8 + lane / 16], 1u);
        if (c != r[0]) atomicAdd(&counts[13], 1u);
        x[0] = __builtin_amdgcn_fractf(r[2] * 0.37f + 0.11f); x[1] = __builtin_amdgcn_fractf(r[3] * 0.53f + 0.07f);      // operands stay finite in [0, 1)
        y[0] = __builtin_amdgcn_fractf(y[0] * 1.7f + 0.3f); y[1] = __builtin_amdgcn_fractf(y[1] * 1.3f + 0.6f);
    }
}

extern "C" int pk_victim(void* stream, int n_blocks, int iters, unsigned* counts_dev) {
    victim<<<n_blocks, 256, 0, (hipStream_t)stream>>>(counts_dev, iters, 0.25f);
    return (int)hipGetLastError();
}

#ifndef PK_SO
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 30;
    hipStream_t sa, sv;
    (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sv, hipStreamNonBlocking);
    f16x8* in; float* out; unsigned* counts;
    (void)hipMalloc(&in, 512 * sizeof(f16x8)); (void)hipMalloc(&out, 2048 * 256 * sizeof(float)); (void)hipMalloc(&counts, 16 * sizeof(unsigned));
    _Float16 h[512 * 8];
    srand(1);
    for (int i = 0; i < 512 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[3] = {"victim alone", "beside MFMA on register operands", "beside MFMA + global_load_dwordx4 of its B operand"};
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(counts, 0, 16 * sizeof(unsigned));
        for (int r = 0; r < rounds; ++r) {                              // one ~10 ms aggressor launch on every SIMD, 8 victim launches beside it
            if (mode == 1) aggressor<0><<<2048, 256, 0, sa>>>(in, out, 20000);
            if (mode == 2) aggressor<1><<<2048, 256, 0, sa>>>(in, out, 20000);
            for (int k = 0; k < 8; ++k) victim<<<512, 256, 0, sv>>>(counts, 1 << 12, 0.25f + r);
            (void)hipStreamSynchronize(sv); (void)hipStreamSynchronize(sa);
        }
        unsigned c[16];
        (void)hipMemcpy(c, counts, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-52s of %.3g evaluations per form: plain fma %u | fma op_sel:[0,1,0] %u (product term zero: %u) | fma op_sel:[1,0,0] %u | mul op_sel:[0,1] %u | "
               "by 16-lane group %u %u %u %u | scalar control %u\n", names[mode], (double)rounds * 8 * 512 * 256 * 4096, c[0], c[1], c[12], c[2], c[3], c[8], c[9], c[10], c[11], c[13]);
    }
    return 0;
}
#endif
