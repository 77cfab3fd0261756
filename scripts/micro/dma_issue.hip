// How does the issue cost of an LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave) depend on the number of waves of a CU that
// issue them?  Each wave issues PIECES pieces back to back from an L2-resident (or HBM-sized) source into its own LDS slice and
// stamps s_memtime around the burst (issue only) and after vmcnt(0) (landed).  Same for plain global_load_dwordx4 into registers.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int PIECES = 16;
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
template <int WAVES, bool DMA>
__global__ void __launch_bounds__(WAVES * 64) probe(const unsigned char* src, size_t span, int reps, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[WAVES * 4096];      // a wave re-uses four 1-KiB slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)buf + wave * 4096);
    size_t off = ((size_t)(blockIdx.x * WAVES + wave) * PIECES * 1024 * 37) & (span - 1);      // span: a power of two
    unsigned long long issue = 0, land = 0;
    float4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        const unsigned char* p = src + off + lane * 16;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) lds_dma16(p + i * 1024, base + (i & 3) * 1024);
        } else {                                                  // (register variant: not used by main(); it faulted with 4+ waves and was not debugged)
            float4 v[PIECES];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(p + i * 1024) : "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < PIECES; ++i) acc.x += v[i].x;
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            issue += t1 - t0; land += t2 - t0;
        }
        if constexpr (DMA) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            issue += t1 - t0; land += t2 - t0;
            acc.x += *reinterpret_cast<const float*>(buf + wave * 4096 + lane * 16);
        }
        off = (off + (size_t)gridDim.x * WAVES * PIECES * 1024) & (span - 1);
    }
    if (lane == 0) { atomicAdd(out, issue); atomicAdd(out + 1, land); atomicAdd(out + 2, 1ull); }
    if (acc.x == 12345.678f) sink[0] = acc.x;
}
template <int WAVES, bool DMA>
static void run(const unsigned char* src, size_t span, const char* what, unsigned long long* out, float* sink, int blocks = 256 * 2) {
    const int reps = 200;
    hipMemset(out, 0, 32);
    printf("[%s %d %d] ", what, WAVES, (int)DMA);
    probe<WAVES, DMA><<<blocks, WAVES * 64>>>(src, span, reps, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("%-18s %s %2d waves per block, %3d blocks: issue %6.0f ticks per piece, landed after %7.0f ticks per burst of %d\n", what, DMA ? "LDS-DMA " : "registers",
           WAVES, blocks, (double)h[0] / h[2] / reps / PIECES, (double)h[1] / h[2] / reps, PIECES);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned char* src; unsigned long long* out; float* sink;
    const size_t big = (size_t)8 << 30;
    if (hipMalloc(&src, big + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, big + (1 << 20)); hipMalloc(&out, 32); hipMalloc(&sink, 4);
    for (int pass = 0; pass < 2; ++pass) {
        const size_t span = pass == 0 ? ((size_t)2 << 20) : big;     // 2 MiB: L2-resident; 8 GiB: HBM
        const char* what = pass == 0 ? "source in L2 (2 MiB)" : "source in HBM (8 GiB)";
        run<1, true>(src, span, what, out, sink); run<2, true>(src, span, what, out, sink); run<4, true>(src, span, what, out, sink); run<8, true>(src, span, what, out, sink);
        run<16, true>(src, span, what, out, sink);
        // a few workgroups only: the chip's HBM is idle, what remains is what ONE CU can keep in flight
        run<4, true>(src, span, what, out, sink, 2); run<4, true>(src, span, what, out, sink, 16); run<4, true>(src, span, what, out, sink, 64);
    }
    return 0;
}
