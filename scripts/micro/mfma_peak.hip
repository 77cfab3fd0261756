// Sustained MFMA issue rate on this chip: bf16 32x32x16 and f32 32x32x2, operands in registers (random data),
// 1 or 2 waves per SIMD, all CUs.  Gives the clock-limited ceiling the conv kernels are measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BF>
__global__ void __launch_bounds__(256, 2) k(const float* in, float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 a4 = *(const float4*)(in + threadIdx.x * 4), b4 = *(const float4*)(in + 1024 + threadIdx.x * 4);
    bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.y, acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 24);
    float* h = (float*)malloc(1 << 20);
    for (int i = 0; i < (1 << 18); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bf = 1; bf >= 0; --bf)
        for (int wgs = 256; wgs <= 512; wgs *= 2) {
            const int iters = bf ? 20000 : 4000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (bf) k<1><<<wgs, 256>>>(in, out, iters); else k<0><<<wgs, 256>>>(in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double flops = (double)wgs * 4 * iters * 24 * (bf ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2);
                if (rep == 2) printf("%s waves/SIMD=%d: %.1f ms  %.1f TFLOP/s\n", bf ? "bf16 32x32x16" : "f32  32x32x2 ", wgs / 256, ms, flops / ms / 1e9);
            }
        }
    return 0;
}
