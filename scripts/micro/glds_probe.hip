// Probe: __builtin_amdgcn_global_load_lds, 16 bytes per lane, per-lane source address, lane-linear LDS destination.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* src, const int* perm, float4* out) {
    __shared__ __attribute__((aligned(16))) float4 lds[512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave-uniform LDS base (16 B x 64 lanes = 1 KiB per instruction); per-lane global source
    const float4* g = src + perm[threadIdx.x];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
    (void)lane;
}
int main() {
    float4 *src, *out; int* perm;
    hipMalloc(&src, 256 * 16); hipMalloc(&out, 256 * 16); hipMalloc(&perm, 256 * 4);
    float4 h[256]; int p[256];
    for (int i = 0; i < 256; ++i) { h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f); p[i] = (i * 37 + 11) % 256; }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(perm, p, sizeof(p), hipMemcpyHostToDevice);
    k<<<1, 256>>>(src, perm, out);
    float4 o[256]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (o[i].x != (float)p[i] || o[i].w != p[i] + 0.75f) ++bad;
    printf("glds probe: %d mismatches (o[5] = %g %g %g %g, expected %d)\n", bad, o[5].x, o[5].y, o[5].z, o[5].w, p[5]);
    return bad != 0;
}
