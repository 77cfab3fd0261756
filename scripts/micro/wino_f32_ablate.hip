// Where does conv3_wino_f32 (csrc/unet_wino_f32.h) lose its MFMA slots?  One layer shape (32 tiles of 32 x 128 x 128, Cin 64 -> Cout 64: 8 chunks), the
// shipped kernel next to the EXP probes of its copy unet_wino_f32_probe.h (each removes one ingredient and computes garbage), the direct kernel conv3_igemm_f32 on the same shape, and a
// register-only loop of the same MFMA (the sustained fp32 MFMA rate at this occupancy).  Build + run: scripts/micro/run_wino_f32_ablate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../oai_analysis_2_amd/csrc/unet_wino_f32.h"     // the shipped kernel
#include "unet_wino_f32_probe.h"                            // its copy with the EXP switches

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace oai;

__global__ void __launch_bounds__(256, 2) mfma_only(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const float av = threadIdx.x * 1e-3f, bv = blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 36; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main() {
    const int tiles = 32, D = 32, H = 128, W = 128, C = 64, Cout = 64, nch = C / 8;
    const size_t vox = (size_t)tiles * D * H * W;
    float *src, *out, *scale, *shift, *zero;
    float4 *panel_w, *panel_d;
    CK(hipMalloc(&src, vox * C * 4)); CK(hipMalloc(&out, vox * Cout * 4));
    CK(hipMalloc(&scale, 256)); CK(hipMalloc(&shift, 256)); CK(hipMalloc(&zero, 64));
    const size_t pw = (size_t)4 * nch * 9 * 2 * 64 + 6 * 64, pd = (size_t)nch * 27 * 2 * 64 + 2 * 64;
    CK(hipMalloc(&panel_w, pw * 16)); CK(hipMalloc(&panel_d, pd * 16));
    CK(hipMemset(src, 0, vox * C * 4)); CK(hipMemset(zero, 0, 64));
    std::vector<float> hw(pw * 4), hd(pd * 4), hs(64, 1.0f);
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& v : hd) v = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(panel_w, hw.data(), pw * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(panel_d, hd.data(), pd * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(scale, hs.data(), 256, hipMemcpyHostToDevice)); CK(hipMemset(shift, 0, 256));
    {   // inputs: a few MB of random values tiled over the tensor
        std::vector<float> h(1 << 22);
        for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
        for (size_t o = 0; o < vox * C; o += h.size()) CK(hipMemcpy(src + o, h.data(), std::min(h.size(), vox * C - o) * 4, hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    a.src0 = src; a.src1 = nullptr; a.C0 = C; a.C1 = 0; a.out = out; a.Cout = Cout; a.scale = scale; a.shift = shift;
    a.D = D; a.H = H; a.W = W; a.relu = 1; a.boxes = nullptr; a.pool_out = nullptr;
    a.lo[0] = a.lo[1] = a.lo[2] = 0; a.hi[0] = D; a.hi[1] = H; a.hi[2] = W; a.ncb = 1;
    const double clk = 2.4e9, simds = 1024.0;
    auto report = [&](const char* name, float ms, double mfmas) {
        printf("%-44s %8.3f ms   MFMA slots used %.3f (at 2.4 GHz)   %7.1f TFLOP/s executed\n", name, ms, mfmas * 64.0 / (simds * ms * 1e-3 * clk),
               mfmas * 64.0 * 4096.0 / 64.0 / (ms * 1e-3) / 1e12);
        fflush(stdout);
    };
    // Winograd: blocks of 2 x 8 x 8, four waves x (8 chunks x 144 MFMAs)
    a.wpanel = panel_w; a.nbz = D / 2; a.nby = H / 8; a.nbx = W / 8;
    const unsigned gw = (unsigned)tiles * a.nbz * a.nby * a.nbx;
    const double mw = (double)gw * 4 * nch * 144;
    const ConvArgs aw = a;
#define RUNW(E, label) report(label, time_ms([&] { conv3_wino_f32_probe<8, 4, E><<<gw, 256>>>(aw, zero); }, 5), mw)
    report("conv3_wino_f32 (shipped)", time_ms([&] { conv3_wino_f32<8, 4><<<gw, 256>>>(aw, zero); }, 5), mw);
    RUNW(0, "  probe copy, EXP 0");
    RUNW(1, "  EXP 1: no fold per chunk");
    RUNW(2, "  EXP 2: no weight loads in the tap loop");
    RUNW(4, "  EXP 4: no input loads, no transform");
    RUNW(8, "  EXP 8: no barrier per chunk");
    RUNW(16, "  EXP 16: no A reads per tap");
    RUNW(32, "  EXP 32: weight loads from one address (L1)");
    RUNW(64, "  EXP 64: input loads from one address (L1)");
    RUNW(96, "  EXP 32+64");
    RUNW(128, "  EXP 128: full-line input fetches, once per 4 chunks");
    RUNW(160, "  EXP 128+32");
    RUNW(256, "  EXP 256: first generation staggered (random)");
    RUNW(512, "  EXP 512: first generation staggered (slot 1 waits)");
    RUNW(256 + 31, "  EXP 256 + 1+2+4+8+16");
    RUNW(1024, "  EXP 1024: chunk loop at wave priority 3");
    RUNW(6, "  EXP 2+4");
    RUNW(22, "  EXP 2+4+16");
    RUNW(30, "  EXP 2+4+8+16");
    RUNW(31, "  EXP 1+2+4+8+16 (MFMAs + epilogue only)");
    RUNW(31 + 2048, "  ... without the epilogue");
    RUNW(31 + 4096, "  ... without the first chunk's loads and transform");
    RUNW(31 + 2048 + 4096, "  ... without both");
    RUNW(2048, "  EXP 2048: the full kernel without its epilogue");
    // direct: blocks of 2 x 8 x 16, (8 chunks x 27 taps x 4 k-steps x 4 accumulators) per wave
    a.wpanel = panel_d; a.nbz = D / 2; a.nby = H / 8; a.nbx = W / 16;
    const unsigned gd = (unsigned)tiles * a.nbz * a.nby * a.nbx;
    const ConvArgs ad = a;
    report("conv3_igemm_f32<2, 8, 16, 2, 4, 1> (direct)", time_ms([&] { conv3_igemm_f32<2, 8, 16, 2, 4, 1><<<gd, 256>>>(ad); }, 5), (double)gd * 4 * nch * 27 * 4 * 4);
    // register-only MFMAs, the Winograd kernel's count per block
    report("register-only v_mfma_f32_32x32x2_f32", time_ms([&] { mfma_only<<<gw, 256>>>(out, nch); }, 5), mw);
    return 0;
}
