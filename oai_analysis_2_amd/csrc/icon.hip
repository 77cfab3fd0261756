// ICON gradICON registration network on gfx950: N tallUNet2s + the warp/compose chains of whatever tree of
// TwoStepRegistration / DownsampleRegistration / FunctionFromVectorField wrappers the checkpoint holds, e.g.
// TwoStep(Downsample(TwoStep(FFVF(u1),FFVF(u2))), FFVF(u3)) or the four-step form with one more
// full-resolution FFVF around it; the tree is compiled into a linear plan at oai_icon_create.  Restates icon_registration 1.1.2
// (see oracle/icon.py; reference call sites oai_analysis/registration.py:20,25).
//
// The nets are 0.1 TFLOP per direction against 80-156 TFLOP for the segmentation U-Net, with
// channel counts (2,16,18,48,3) that do not fill an MFMA tile, so they run as direct fp32 VALU
// convolutions: one thread = one output voxel x COUT_T couts held in registers, lanes along x
// (coalesced NCDHW reads), weights repacked [ci][tap][co] so the COUT_T weights of a step are one
// wave-uniform scalar load (s_load_dwordxN feeding v_fma with an SGPR operand: no LDS, no barrier).
// torch.cat is free: every tensor is produced directly into its channel slice of the level's
// concat buffer  cat_d = [ up_out[d] | down[d] ]  (UNet2.forward: x = cat([x, skips[d]], 1)).
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

constexpr int kDown[6] = {2, 16, 32, 64, 256, 512};
constexpr int kUpOut[5] = {16, 32, 64, 128, 256};
constexpr int kUpIn[5] = {48, 96, 192, 512, 512};
constexpr float kLeaky = 0.01f;
constexpr float kBnEps = 1e-5f;
constexpr long long kSplitKBelow = 16384;     // output voxels (per launch / per parity class) below which K is split over threads
constexpr long long kMfmaUpFrom = 2048;       // output voxels per parity class from which the up-conv runs on MFMA (icon_up_mfma_kernel)
#ifndef OAI_LAST_XT
#define OAI_LAST_XT 2
#endif
constexpr long long kFewBlocks = 1024;        // split-K launches with fewer workgroups than this use 4 couts per block (4x the blocks)

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : v * kLeaky; }

struct __attribute__((packed, aligned(4))) pair_f32s { float a, b; };       // 8-byte store at 4-byte alignment (dwordx2)
__device__ __forceinline__ void pair_store(float* p, float a, float b) { *reinterpret_cast<pair_f32s*>(p) = pair_f32s{a, b}; }

// Conv3d k3 p1, stride S, on leaky_relu(x) (PRE) + bias, optional residual
//   RES: + avg_pool3d(x,2,ceil_mode=True) zero-padded to Cout channels, in front or behind (res_ofs; UNet2 down path)
// out = (acc + bias [+ res]) / div
// KS > 1 (deep levels, a few hundred voxels or fewer but K = 27*Cin up to 6912): the block's 256 threads are
// 256/KS voxels x KS slices of the input channels; partial sums meet in LDS and slice 0 runs the epilogue.
template <int S, int COUT_T, bool PRE, bool RES, int KS>
__global__ void __launch_bounds__(256)
icon_conv3_kernel(const float* __restrict__ x, int Cin, int D, int H, int W,
                  const float* __restrict__ wk /*[Cin][27][Cout]*/, const float* __restrict__ bias,
                  float* __restrict__ out, int Cout, int Do, int Ho, int Wo, float div, int res_ofs) {
    constexpr int VT = 256 / KS;
    __shared__ float red[KS > 1 ? (KS - 1) * VT * COUT_T : 1];
    const int cg = blockIdx.y;
    const int ks = KS == 1 ? 0 : threadIdx.x / VT;      // literal 0 keeps the channel loop wave-uniform when K is not split
    const long long nvox = (long long)Do * Ho * Wo;
    const long long v = (long long)blockIdx.x * VT + threadIdx.x % VT;
    const bool live = v < nvox;
    const long long vv = live ? v : 0;
    const int ox = (int)(vv % Wo), oy = (int)((vv / Wo) % Ho), oz = (int)(vv / ((long long)Wo * Ho));
    const long long plane = (long long)D * H * W;
    float acc[COUT_T];
#pragma unroll
    for (int j = 0; j < COUT_T; ++j) acc[j] = 0.0f;
    const int iz0 = oz * S - 1, iy0 = oy * S - 1, ix0 = ox * S - 1;
    if (KS == 1 && !live) return;                 // (with KS > 1 every thread must reach the barrier)
    if (KS == 1 || live)
        for (int ci = ks; ci < Cin; ci += KS) {
            const float* xp = x + ci * plane;
            const float* wp = wk + ((long long)ci * 27) * Cout + cg * COUT_T;
#pragma unroll 9
            for (int t = 0; t < 27; ++t) {
                const int iz = iz0 + t / 9, iy = iy0 + (t / 3) % 3, ix = ix0 + t % 3;
                float in = 0.0f;
                if ((unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    in = xp[((long long)iz * H + iy) * W + ix];
                if (PRE) in = leaky(in);
#pragma unroll
                for (int j = 0; j < COUT_T; ++j) acc[j] = fmaf(in, wp[t * Cout + j], acc[j]);
            }
        }
    if (KS > 1) {
        if (ks > 0) {
#pragma unroll
            for (int j = 0; j < COUT_T; ++j) red[((ks - 1) * COUT_T + j) * VT + threadIdx.x % VT] = acc[j];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k)              // fixed order: deterministic
#pragma unroll
            for (int j = 0; j < COUT_T; ++j) acc[j] += red[((k - 1) * COUT_T + j) * VT + threadIdx.x];
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < COUT_T; ++j) {
        const int co = cg * COUT_T + j;
        float r = acc[j] + bias[co];
        if (RES) {
            const int cs = co - res_ofs;               // pad_or_crop: zero channels in front (res_ofs = Cout - Cin) or behind (0)
            if (cs >= 0 && cs < Cin) {
                const int z1 = min(2 * oz + 2, D), y1 = min(2 * oy + 2, H), x1 = min(2 * ox + 2, W);
                float s = 0.0f;
                for (int z = 2 * oz; z < z1; ++z)
                    for (int y = 2 * oy; y < y1; ++y)
                        for (int xx = 2 * ox; xx < x1; ++xx) s += x[cs * plane + ((long long)z * H + y) * W + xx];
                r += s / (float)((z1 - 2 * oz) * (y1 - 2 * oy) * (x1 - 2 * ox));
            }
        }
        out[co * nvox + v] = div == 1.0f ? r : r / div;
    }
}

__device__ __forceinline__ void up_src(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    // F.interpolate(scale_factor=2, mode='trilinear', align_corners=False): src = (dst + 0.5)/2 - 0.5, clamped at 0
    float src = 0.5f * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

// ConvTranspose3d k4 s2 p1 on leaky_relu(x) + bias + trilinear-x2 upsample of x[:Cout] -> BatchNorm (eval)
// -> cropped to (Dc,Hc,Wc).  One block = one output parity class (uniform weights).
template <int COUT_T, int KS>
__global__ void __launch_bounds__(256)
icon_up_kernel(const float* __restrict__ x, int Cin, int D, int H, int W,
               const float* __restrict__ wk /*[Cin][64][Cout]*/, const float* __restrict__ bias,
               const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
               float* __restrict__ out, int Cout, int Dc, int Hc, int Wc) {
    constexpr int VT = 256 / KS;
    __shared__ float red[KS > 1 ? (KS - 1) * VT * COUT_T : 1];
    const int cg = blockIdx.y;
    const int ks = KS == 1 ? 0 : threadIdx.x / VT;      // literal 0 keeps the channel loop wave-uniform when K is not split
    const int par = blockIdx.z, pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
    const int nz = (Dc - pz + 1) / 2, ny = (Hc - py + 1) / 2, nx = (Wc - px + 1) / 2;
    const long long nv = (long long)nz * ny * nx;
    const long long v = (long long)blockIdx.x * VT + threadIdx.x % VT;
    const bool live = v < nv;
    const long long vv = live ? v : 0;
    const int tx = (int)(vv % nx), ty = (int)((vv / nx) % ny), tz = (int)(vv / ((long long)nx * ny));
    const int oz = 2 * tz + pz, oy = 2 * ty + py, ox = 2 * tx + px;
    const long long plane = (long long)D * H * W;
    const long long oplane = (long long)Dc * Hc * Wc;
    // o = 2 i - 1 + k  ->  k in {q, q+2}, q = (p+1)&1;  i = (o + 1 - k)/2
    const int qz = (pz + 1) & 1, qy = (py + 1) & 1, qx = (px + 1) & 1;
    float acc[COUT_T];
#pragma unroll
    for (int j = 0; j < COUT_T; ++j) acc[j] = 0.0f;
    if (KS == 1 && !live) return;
    if (KS == 1 || live)
        for (int ci = ks; ci < Cin; ci += KS) {
            const float* xp = x + ci * plane;
            const float* wp = wk + ((long long)ci * 64) * Cout + cg * COUT_T;
#pragma unroll 8
            for (int t = 0; t < 8; ++t) {
                const int kz = qz + 2 * (t >> 2), ky = qy + 2 * ((t >> 1) & 1), kx = qx + 2 * (t & 1);
                const int iz = (oz + 1 - kz) >> 1, iy = (oy + 1 - ky) >> 1, ix = (ox + 1 - kx) >> 1;
                float in = 0.0f;
                if ((unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    in = leaky(xp[((long long)iz * H + iy) * W + ix]);
                const float* w = wp + ((kz * 4 + ky) * 4 + kx) * Cout;
#pragma unroll
                for (int j = 0; j < COUT_T; ++j) acc[j] = fmaf(in, w[j], acc[j]);
            }
        }
    if (KS > 1) {
        if (ks > 0) {
#pragma unroll
            for (int j = 0; j < COUT_T; ++j) red[((ks - 1) * COUT_T + j) * VT + threadIdx.x % VT] = acc[j];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k)
#pragma unroll
            for (int j = 0; j < COUT_T; ++j) acc[j] += red[((k - 1) * COUT_T + j) * VT + threadIdx.x];
    }
    if (!live) return;
    int z0, z1, y0, y1, x0, x1;
    float a0, a1, b0, b1, c0, c1;
    up_src(oz, D, z0, z1, a0, a1);
    up_src(oy, H, y0, y1, b0, b1);
    up_src(ox, W, x0, x1, c0, c1);
#pragma unroll
    for (int j = 0; j < COUT_T; ++j) {
        const int co = cg * COUT_T + j;
        const float* p = x + co * plane;          // pad_or_crop(x, Cout): the first Cout channels (Cin >= Cout)
        auto at = [&](int zz, int yy, int xx) { return p[((long long)zz * H + yy) * W + xx]; };
        const float res = a0 * (b0 * (c0 * at(z0, y0, x0) + c1 * at(z0, y0, x1)) + b1 * (c0 * at(z0, y1, x0) + c1 * at(z0, y1, x1))) +
                          a1 * (b0 * (c0 * at(z1, y0, x0) + c1 * at(z1, y0, x1)) + b1 * (c0 * at(z1, y1, x0) + c1 * at(z1, y1, x1)));
        const float r = (acc[j] + bias[co]) + res;
        out[co * oplane + ((long long)oz * Hc + oy) * Wc + ox] = r * bn_scale[co] + bn_shift[co];
    }
}

// ---- MFMA up-conv for the big levels (full and half resolution: 0.75 of the nets' FLOPs) ----------------------------------------
// ConvTranspose3d k4 s2 p1 as a GEMM per output row: an output voxel (oz, oy, ox = 2 tx + px) takes 2 x 2 x 2 taps; along x the
// two parities share their inputs:   px = 0: (kx 1, ix = tx) + (kx 3, ix = tx - 1)      px = 1: (kx 0, ix = tx + 1) + (kx 2, ix = tx)
// so one wave owns 16 MB consecutive tx of one (oz, oy) row, BOTH x parities, 16 couts, on v_mfma_f32_16x16x4_f32 (exact fp32
// products): A operand = weights (rows = couts), B operand = three x-shifted 16-voxel segments of leaky(x) (columns = tx), k = 4
// input channels per instruction.  Per (kz, ky) tap pair and 4 channels: 4 weight fragments + 3 MB voxel fragments feed 4 MB
// MFMAs -- one dword load per MFMA, straight from L1/L2 (the VALU kernel it replaces issued one load per 16 FMAs but 8 x fewer
// FLOPs per instruction).  The C/D layout gives a lane 4 couts x 1 tx: with the two parities paired, every store instruction
// writes four full 128-byte lines (the VALU kernel wrote every other float of a line).  Epilogue as icon_up_kernel: + bias +
// trilinear x2 residual of x[:Cout] (same expression, same order) -> BatchNorm -> crop.
typedef float f32x4m __attribute__((ext_vector_type(4)));

// SPLIT (the small, K-deep levels: a few hundred output rows, K = 8 x 512): the four waves of a block are the four (kz, ky) tap
// pairs of ONE unit -- four times the waves, a quarter of the K loop each --, partial sums meet in LDS in a fixed order and wave 0
// runs the epilogue.  The scalar-weight VALU kernels these levels used were bound by the latency of their scalar loads.
template <int MB, bool SPLIT>
__global__ void __launch_bounds__(256)
icon_up_mfma_kernel(const float* __restrict__ x, int Cin, int D, int H, int W,
                    const float* __restrict__ wk /*[Cin][64][Cout]*/, const float* __restrict__ bias,
                    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                    float* __restrict__ out, int Cout, int Dc, int Hc, int Wc, int ntxb, long long nunits) {
    __shared__ float red[SPLIT ? 3 * MB * 2 * 4 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long unit = SPLIT ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + wave;   // one wave (SPLIT: one block) = one (oz, oy, tx block)
    if (unit >= nunits) return;                                          // (SPLIT: block-uniform, so every wave of a live block reaches the barrier)
    const int txb = (int)(unit % ntxb);
    const int rowi = (int)(unit / ntxb);
    const int oy = rowi % Hc, oz = rowi / Hc;
    const int co0 = blockIdx.y * 16;
    const int pz = oz & 1, py = oy & 1, tz = oz >> 1, ty = oy >> 1;
    const int col = lane & 15, kq = lane >> 4;
    const long long plane = (long long)D * H * W;
    f32x4m acc[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[m][p] = f32x4m{0.0f, 0.0f, 0.0f, 0.0f};
    const int tx0 = txb * MB * 16 + col;
    for (int a = SPLIT ? (wave >> 1) : 0; a < (SPLIT ? (wave >> 1) + 1 : 2); ++a) {
        const int kz = pz ? 2 * a : 1 + 2 * a, iz = pz ? tz + 1 - a : tz - a;          // o = 2 i - 1 + k
        if ((unsigned)iz >= (unsigned)D) continue;
        for (int b = SPLIT ? (wave & 1) : 0; b < (SPLIT ? (wave & 1) + 1 : 2); ++b) {
            const int ky = py ? 2 * b : 1 + 2 * b, iy = py ? ty + 1 - b : ty - b;
            if ((unsigned)iy >= (unsigned)H) continue;
            const float* xrow = x + ((long long)iz * H + iy) * W;
            const float* wt = wk + (long long)((kz * 4 + ky) * 4) * Cout + co0 + col;
            for (int ci0 = 0; ci0 < Cin; ci0 += 4) {
                const int ci = ci0 + kq;
                const float* wp = wt + (long long)ci * 64 * Cout;
                const float w0 = wp[0], w1 = wp[Cout], w2 = wp[2 * Cout], w3 = wp[3 * Cout];
                const float* xp = xrow + ci * plane;
                float xm[MB], xc[MB], xq[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int tx = tx0 + m * 16;
                    xm[m] = (unsigned)(tx - 1) < (unsigned)W ? leaky(xp[tx - 1]) : 0.0f;
                    xc[m] = tx < W ? leaky(xp[tx]) : 0.0f;
                    xq[m] = tx + 1 < W ? leaky(xp[tx + 1]) : 0.0f;
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, xc[m], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, xq[m], acc[m][1], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3, xm[m], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2, xc[m], acc[m][1], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (SPLIT) {
        if (wave > 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((((wave - 1) * MB + m) * 2 + p) * 4 + r) * 64 + lane] = acc[m][p][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)                                      // fixed order: deterministic
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][p][r] += red[(((w * MB + m) * 2 + p) * 4 + r) * 64 + lane];
    }
    // epilogue: lane = couts co0 + 4 kq + r (r = 0..3) of tx; the two parities are x-adjacent outputs
    int z0, z1, y0, y1;
    float a0, a1, b0, b1;
    up_src(oz, D, z0, z1, a0, a1);
    up_src(oy, H, y0, y1, b0, b1);
    const long long oplane = (long long)Dc * Hc * Wc;
    const int nxt = (Wc + 1) / 2;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int tx = tx0 + m * 16;
        if (tx >= nxt) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 4 * kq + r;
            const float* p = x + co * plane;          // pad_or_crop(x, Cout): the first Cout channels (Cin >= Cout)
            auto at = [&](int zz, int yy, int xx) { return p[((long long)zz * H + yy) * W + xx]; };
            float v[2];
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                int x0, x1;
                float c0, c1;
                up_src(2 * tx + px, W, x0, x1, c0, c1);
                const float res = a0 * (b0 * (c0 * at(z0, y0, x0) + c1 * at(z0, y0, x1)) + b1 * (c0 * at(z0, y1, x0) + c1 * at(z0, y1, x1))) +
                                  a1 * (b0 * (c0 * at(z1, y0, x0) + c1 * at(z1, y0, x1)) + b1 * (c0 * at(z1, y1, x0) + c1 * at(z1, y1, x1)));
                const float t = (acc[m][px][r] + bias[co]) + res;
                v[px] = t * bn_scale[co] + bn_shift[co];
            }
            float* o = out + co * oplane + ((long long)oz * Hc + oy) * Wc + 2 * tx;
            if (2 * tx + 1 < Wc) { pair_store(o, v[0], v[1]); }
            else o[0] = v[0];
        }
    }
}

// lastConv: Conv3d 18 -> 3, k3 p1, / 10, at full resolution.  N = 3 fills no MFMA tile; as a direct VALU convolution the first
// version did one global load per 3 FMAs.  Here a thread owns XT x-consecutive voxels x 3 couts: per (ci, dz, dy) it loads the XT + 2
// inputs once and does 9 XT FMAs (weights are wave-uniform scalar loads).  Measured per ICON direction: XT = 8 5.70 ms, 4 5.14 ms,
// 2 4.85 ms (fewer loads per FMA lose against coalescing and thread count), XT = 1 (round 1) 5.4 ms-equivalent.
template <int XT>
__global__ void __launch_bounds__(256)
icon_last_conv_kernel(const float* __restrict__ x, int Cin, int D, int H, int W,
                      const float* __restrict__ wk /*[Cin][27][3]*/, const float* __restrict__ bias, float* __restrict__ out, float div) {
    const int nxq = (W + XT - 1) / XT;
    const long long total = (long long)D * H * nxq;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int xq = (int)(id % nxq);
    const int y = (int)((id / nxq) % H), z = (int)(id / ((long long)nxq * H));
    const int xs = XT * xq;
    const long long plane = (long long)D * H * W;
    float acc[XT][3];
#pragma unroll
    for (int v = 0; v < XT; ++v)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[v][j] = 0.0f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xp = x + ci * plane;
        const float* wp = wk + (long long)ci * 27 * 3;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int iz = z + dz - 1, iy = y + dy - 1;
                const bool rowok = (unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H;
                const float* rp = xp + ((long long)(rowok ? iz : 0) * H + (rowok ? iy : 0)) * W;
                float in[XT + 2];
#pragma unroll
                for (int e = 0; e < XT + 2; ++e) {
                    const int ix = xs - 1 + e;
                    in[e] = rowok && (unsigned)ix < (unsigned)W ? rp[ix] : 0.0f;
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* w = wp + ((dz * 3 + dy) * 3 + dx) * 3;
#pragma unroll
                    for (int v = 0; v < XT; ++v)
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc[v][j] = fmaf(in[v + dx], w[j], acc[v][j]);
                }
            }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int v = 0; v < XT; ++v)
            if (xs + v < W) {
                const float r = acc[v][j] + bias[j];
                out[j * plane + ((long long)z * H + y) * W + xs + v] = div == 1.0f ? r : r / div;
            }
}

// Device-to-device copies inside the registration path are KERNELS, not hipMemcpyAsync: on a non-blocking side stream the runtime's
// async D2D copy was observed not to hold back the kernels queued behind it on the same stream (a later launch overwrote the
// source while the copy was still reading it: scripts/dbg_graph_race.py) -- a kernel is stream-ordered by construction, and
// captures into the graph as an ordinary kernel node.
__global__ void __launch_bounds__(256) copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * 256;
    if (((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    }
}

int copy_f32(const float* src, float* dst, long long n, hipStream_t st) {
    long long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    copy_f32_kernel<<<(unsigned)blocks, 256, 0, st>>>(src, dst, n);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

struct NetWeights {
    float* down_w[5]; float* down_b[5];
    float* up_w[5]; float* up_b[5]; float* bn_s[5]; float* bn_t[5];
    float* last_w; float* last_b;
};

}  // namespace

// One launch group of a direction, compiled from the step tree (plan_tree below).  Buffers are float offsets into the workspace.
struct IconStep {
    enum Kind { POOL, UNET, CHAIN } kind;
    int lvl = 0;                          // grid = the network shape halved lvl times (ceil)
    // POOL: src (lvl) -> dst (lvl + 1).  UNET: net(a, b) -> out [3][grid].  CHAIN: out = image ? image(c) : c with
    // c = id(lvl) [+ start]; c += sample(field_i, c)
    int net = -1;
    size_t src = 0, src2 = 0, dst = 0;
    bool has_start = false, has_image = false;
    size_t start = 0;
    int nf = 0;
    size_t field[OAI_WARP_CHAIN_MAX_FIELDS];
    int field_lvl[OAI_WARP_CHAIN_MAX_FIELDS];
};

struct oai_icon {
    std::vector<NetWeights> net;
    std::vector<oai_icon_node> nodes;
    int root = 0;
    int D, H, W;
    std::vector<void*> allocs;
    // the compiled direction: offsets of A, B, phi (fixed homes: what the captured graph reads and writes), of the U-Net scratch,
    // and the steps in launch order
    std::vector<IconStep> steps;
    size_t off_A = 0, off_B = 0, off_phi = 0, off_unet = 0, ws_floats = 0;
    int nets_at_level[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int chain_len = 0;
    // hipGraph replay of the dependent launches of one direction (oai_icon_forward; ~70 for three U-Nets): captured once per workspace on an
    // internal stream (the caller's stream may be the legacy null stream, which cannot be captured), replayed on the caller's.
    bool use_graph = true, graph_broken = false;
    bool pad_front = true;            // pad_or_crop's zero channels in front or behind (option "pad_front")
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    void* g_ws = nullptr;
    long long replays = 0, direct_runs = 0;
};

namespace {

int upload(oai_icon* h, const std::vector<float>& v, float** dst) {
    void* d = nullptr;
    OAI_CHECK_HIP(hipMalloc(&d, v.size() * sizeof(float)));
    h->allocs.push_back(d);
    OAI_CHECK_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *dst = (float*)d;
    return OAI_OK;
}

// Conv3d [Co][Ci][27] -> [Ci][27][Co]
std::vector<float> repack_conv(const float* w, int co_n, int ci_n, int taps) {
    std::vector<float> o((size_t)co_n * ci_n * taps);
    for (int co = 0; co < co_n; ++co)
        for (int ci = 0; ci < ci_n; ++ci)
            for (int t = 0; t < taps; ++t) o[((size_t)ci * taps + t) * co_n + co] = w[((size_t)co * ci_n + ci) * taps + t];
    return o;
}
// ConvTranspose3d [Ci][Co][64] -> [Ci][64][Co]
std::vector<float> repack_convT(const float* w, int ci_n, int co_n, int taps) {
    std::vector<float> o((size_t)co_n * ci_n * taps);
    for (int ci = 0; ci < ci_n; ++ci)
        for (int co = 0; co < co_n; ++co)
            for (int t = 0; t < taps; ++t) o[((size_t)ci * taps + t) * co_n + co] = w[((size_t)ci * co_n + co) * taps + t];
    return o;
}

struct Dims { int d[6][3]; long long vox[6]; };

Dims level_dims(int D, int H, int W) {
    Dims r;
    r.d[0][0] = D; r.d[0][1] = H; r.d[0][2] = W;
    for (int l = 1; l < 6; ++l) for (int i = 0; i < 3; ++i) r.d[l][i] = (r.d[l - 1][i] + 1) / 2;
    for (int l = 0; l < 6; ++l) r.vox[l] = (long long)r.d[l][0] * r.d[l][1] * r.d[l][2];
    return r;
}

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

// floats needed by one tallUNet2 forward at (D,H,W): cat_0..cat_4 and the bottom tensor
size_t unet_ws_floats(int D, int H, int W) {
    const Dims dm = level_dims(D, H, W);
    size_t n = 0;
    for (int l = 0; l < 5; ++l) n += align256((size_t)(kUpOut[l] + kDown[l]) * dm.vox[l] * 4) / 4;
    n += align256((size_t)kDown[5] * dm.vox[5] * 4) / 4;
    return n;
}

int unet_forward(const NetWeights& nw, bool pad_front, const float* a, const float* b, int D, int H, int W, float* out,
                 float* ws, hipStream_t st) {
    const Dims dm = level_dims(D, H, W);
    // every axis must survive five halvings with a >= 2 input to each pooling (avg_pool3d needs size >= kernel)
    float* cat[5];
    size_t o = 0;
    for (int l = 0; l < 5; ++l) { cat[l] = ws + o; o += align256((size_t)(kUpOut[l] + kDown[l]) * dm.vox[l] * 4) / 4; }
    float* bottom = ws + o;
    // x = cat([a, b], 1) lives in the skip slice of cat_0
    if (int rc = copy_f32(a, cat[0] + (size_t)kUpOut[0] * dm.vox[0], dm.vox[0], st)) return rc;
    if (int rc = copy_f32(b, cat[0] + (size_t)(kUpOut[0] + 1) * dm.vox[0], dm.vox[0], st)) return rc;
    for (int l = 0; l < 5; ++l) {
        const float* src = cat[l] + (size_t)kUpOut[l] * dm.vox[l];
        float* dst = l < 4 ? cat[l + 1] + (size_t)kUpOut[l + 1] * dm.vox[l + 1] : bottom;
        const int res_ofs = pad_front ? kDown[l + 1] - kDown[l] : 0;
        if (dm.vox[l + 1] >= kSplitKBelow || kDown[l] < 8) {
            dim3 grid(oai::cdiv(dm.vox[l + 1], 256), kDown[l + 1] / 16);
            icon_conv3_kernel<2, 16, true, true, 1><<<grid, 256, 0, st>>>(src, kDown[l], dm.d[l][0], dm.d[l][1], dm.d[l][2],
                                                                           nw.down_w[l], nw.down_b[l], dst, kDown[l + 1],
                                                                           dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2], 1.0f, res_ofs);
        } else if ((long long)oai::cdiv(dm.vox[l + 1], 32) * (kDown[l + 1] / 16) >= kFewBlocks) {
            dim3 grid(oai::cdiv(dm.vox[l + 1], 32), kDown[l + 1] / 16);
            icon_conv3_kernel<2, 16, true, true, 8><<<grid, 256, 0, st>>>(src, kDown[l], dm.d[l][0], dm.d[l][1], dm.d[l][2],
                                                                           nw.down_w[l], nw.down_b[l], dst, kDown[l + 1],
                                                                           dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2], 1.0f, res_ofs);
        } else if ((long long)oai::cdiv(dm.vox[l + 1], 32) * (kDown[l + 1] / 16) < kFewBlocks / 16) {
            dim3 grid(oai::cdiv(dm.vox[l + 1], 32), kDown[l + 1]);          // one cout per block
            icon_conv3_kernel<2, 1, true, true, 8><<<grid, 256, 0, st>>>(src, kDown[l], dm.d[l][0], dm.d[l][1], dm.d[l][2],
                                                                          nw.down_w[l], nw.down_b[l], dst, kDown[l + 1],
                                                                          dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2], 1.0f, res_ofs);
        } else {        // the deepest levels (a few dozen voxels): 4 couts per block instead of 16, four times the blocks
            dim3 grid(oai::cdiv(dm.vox[l + 1], 32), kDown[l + 1] / 4);
            icon_conv3_kernel<2, 4, true, true, 8><<<grid, 256, 0, st>>>(src, kDown[l], dm.d[l][0], dm.d[l][1], dm.d[l][2],
                                                                          nw.down_w[l], nw.down_b[l], dst, kDown[l + 1],
                                                                          dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2], 1.0f, res_ofs);
        }
        OAI_CHECK_LAUNCH();
    }
    for (int l = 4; l >= 0; --l) {
        const float* src = l == 4 ? bottom : cat[l + 1];
        const long long per_par = (long long)((dm.d[l][0] + 1) / 2) * ((dm.d[l][1] + 1) / 2) * ((dm.d[l][2] + 1) / 2);
        if (per_par >= kMfmaUpFrom) {                   // the big levels: MFMA (fp32 products), one wave per 32 tx of an output row
            constexpr int MB = 2;
            const int ntxb = (int)oai::cdiv((dm.d[l][2] + 1) / 2, 16 * MB);
            const long long nunits = (long long)dm.d[l][0] * dm.d[l][1] * ntxb;
            dim3 grid(oai::cdiv(nunits, 4), kUpOut[l] / 16);
            icon_up_mfma_kernel<MB, false><<<grid, 256, 0, st>>>(src, kUpIn[l], dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2],
                                                                 nw.up_w[l], nw.up_b[l], nw.bn_s[l], nw.bn_t[l], cat[l], kUpOut[l],
                                                                 dm.d[l][0], dm.d[l][1], dm.d[l][2], ntxb, nunits);
        } else if (kUpOut[l] % 16 == 0 && kUpIn[l] % 4 == 0) {      // the small, K-deep levels: one block per unit, its waves split the (kz, ky) taps
            constexpr int MB = 1;
            const int ntxb = (int)oai::cdiv((dm.d[l][2] + 1) / 2, 16 * MB);
            const long long nunits = (long long)dm.d[l][0] * dm.d[l][1] * ntxb;
            dim3 grid((unsigned)nunits, kUpOut[l] / 16);
            icon_up_mfma_kernel<MB, true><<<grid, 256, 0, st>>>(src, kUpIn[l], dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2],
                                                                nw.up_w[l], nw.up_b[l], nw.bn_s[l], nw.bn_t[l], cat[l], kUpOut[l],
                                                                dm.d[l][0], dm.d[l][1], dm.d[l][2], ntxb, nunits);
        } else if ((long long)oai::cdiv(per_par, 32) * (kUpOut[l] / 16) * 8 >= kFewBlocks) {
            dim3 grid(oai::cdiv(per_par, 32), kUpOut[l] / 16, 8);
            icon_up_kernel<16, 8><<<grid, 256, 0, st>>>(src, kUpIn[l], dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2],
                                                         nw.up_w[l], nw.up_b[l], nw.bn_s[l], nw.bn_t[l], cat[l], kUpOut[l],
                                                         dm.d[l][0], dm.d[l][1], dm.d[l][2]);
        } else if ((long long)oai::cdiv(per_par, 32) * (kUpOut[l] / 16) * 8 < kFewBlocks / 16) {
            dim3 grid(oai::cdiv(per_par, 32), kUpOut[l], 8);
            icon_up_kernel<1, 8><<<grid, 256, 0, st>>>(src, kUpIn[l], dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2],
                                                        nw.up_w[l], nw.up_b[l], nw.bn_s[l], nw.bn_t[l], cat[l], kUpOut[l],
                                                        dm.d[l][0], dm.d[l][1], dm.d[l][2]);
        } else {
            dim3 grid(oai::cdiv(per_par, 32), kUpOut[l] / 4, 8);
            icon_up_kernel<4, 8><<<grid, 256, 0, st>>>(src, kUpIn[l], dm.d[l + 1][0], dm.d[l + 1][1], dm.d[l + 1][2],
                                                        nw.up_w[l], nw.up_b[l], nw.bn_s[l], nw.bn_t[l], cat[l], kUpOut[l],
                                                        dm.d[l][0], dm.d[l][1], dm.d[l][2]);
        }
        OAI_CHECK_LAUNCH();
    }
    constexpr int XT = OAI_LAST_XT;
    icon_last_conv_kernel<XT><<<oai::cdiv((long long)D * H * ((W + XT - 1) / XT), 256), 256, 0, st>>>(cat[0], kUpOut[0] + kDown[0], D, H, W,
                                                                                                     nw.last_w, nw.last_b, out, 10.0f);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

bool dims_ok(int D, int H, int W) {
    const Dims dm = level_dims(D, H, W);
    for (int i = 0; i < 3; ++i)
        if (dm.d[4][i] < 2) return false;     // the 5th avg_pool3d(2) needs an input of at least 2 per axis
    return true;
}

// ---- the step tree -> a linear plan ---------------------------------------------------------------------------------------
// network_wrappers of the package, restated (oracle/icon.py:forward_tree is the same recursion on torch CPU ops):
//   FFVF(net k).forward(A, B):       d_k = net_k(A, B) on A's grid; closure = [d_k]
//   Downsample(n).forward(A, B):     n.forward(avg_pool(A), avg_pool(B)); the closure is the child's
//   TwoStep(phi, psi).forward(A, B): F_phi = phi.forward(A, B); A_w = A sampled at F_phi(identity of A's grid);
//                                    F_psi = psi.forward(A_w, B); closure x -> phi(psi(x)) = links F_psi then F_phi
// A closure is the list of its displacement fields in APPLICATION order; applied to the tagged identity map of a grid, the first
// link is `id + d` when d lives on that same grid (isIdentity shortcut) and `id + sample(d, id)` otherwise, every later link is
// `c + sample(d, c)`: exactly one oai_warp_chain launch, with the warped image as its last gather when there is one.
struct Link { size_t off; int lvl; };

struct TreePlanner {
    oai_icon* h;
    long long vox[8];
    int gd[8][3];
    size_t cur = 0;               // floats
    std::vector<std::pair<std::pair<size_t, int>, size_t>> pooled;      // (source offset, source level) -> pooled buffer
    std::vector<char> node_used, net_used;
    int err = OAI_OK;

    size_t take(size_t floats) { size_t o = cur; cur += align256(floats * 4) / 4; return o; }

    int fail(const char* msg, int v) { if (!err) err = oai::set_error(OAI_ERR_ARG, msg, v); return err; }

    size_t pool(size_t src, int lvl) {
        for (auto& e : pooled) if (e.first.first == src && e.first.second == lvl) return e.second;
        IconStep st; st.kind = IconStep::POOL; st.lvl = lvl; st.src = src; st.dst = take(vox[lvl + 1]);
        h->steps.push_back(st);
        pooled.push_back({{src, lvl}, st.dst});
        return st.dst;
    }

    void chain(const std::vector<Link>& F, int lvl, bool has_image, size_t image, size_t out) {
        IconStep st; st.kind = IconStep::CHAIN; st.lvl = lvl; st.dst = out; st.has_image = has_image; st.src = image;
        size_t i = 0;
        if (!F.empty() && F[0].lvl == lvl) { st.has_start = true; st.start = F[0].off; i = 1; }
        for (; i < F.size(); ++i) { st.field[st.nf] = F[i].off; st.field_lvl[st.nf] = F[i].lvl; ++st.nf; }
        h->steps.push_back(st);
    }

    std::vector<Link> eval(int node, size_t A, size_t B, int lvl, int depth) {
        std::vector<Link> none;
        if (err) return none;
        if (node < 0 || node >= (int)h->nodes.size()) { fail("oai_icon_create: node index %d out of range", node); return none; }
        if (depth > 32 || node_used[node]) { fail("oai_icon_create: node %d is used twice (the step tree must be a tree)", node); return none; }
        node_used[node] = 1;
        const oai_icon_node nd = h->nodes[node];
        if (nd.kind == OAI_ICON_FFVF) {
            if (nd.a < 0 || nd.a >= (int)h->net.size()) { fail("oai_icon_create: FFVF names net %d, which does not exist", nd.a); return none; }
            if (net_used[nd.a]) { fail("oai_icon_create: net %d is used by two FFVF nodes", nd.a); return none; }
            net_used[nd.a] = 1;
            if (!dims_ok(gd[lvl][0], gd[lvl][1], gd[lvl][2])) {
                fail("oai_icon_create: the grid of net %d is too small for five 2x poolings (each axis must be >= 17)", nd.a);
                return none;
            }
            IconStep st; st.kind = IconStep::UNET; st.lvl = lvl; st.net = nd.a; st.src = A; st.src2 = B; st.dst = take(3 * vox[lvl]);
            h->steps.push_back(st);
            ++h->nets_at_level[lvl];
            return {Link{st.dst, lvl}};
        }
        if (nd.kind == OAI_ICON_DOWN) {
            if (lvl + 1 >= 8 || gd[lvl][0] < 2 || gd[lvl][1] < 2 || gd[lvl][2] < 2) { fail("oai_icon_create: too many Downsample levels at node %d", node); return none; }
            const size_t a = pool(A, lvl), b = pool(B, lvl);
            return eval(nd.a, a, b, lvl + 1, depth + 1);
        }
        if (nd.kind == OAI_ICON_TWO) {
            std::vector<Link> Fphi = eval(nd.a, A, B, lvl, depth + 1);
            if (err) return none;
            const size_t Aw = take(vox[lvl]);
            if ((int)Fphi.size() > OAI_WARP_CHAIN_MAX_FIELDS) { fail("oai_icon_create: more than %d steps in one chain", OAI_WARP_CHAIN_MAX_FIELDS); return none; }
            chain(Fphi, lvl, true, A, Aw);
            std::vector<Link> F = eval(nd.b, Aw, B, lvl, depth + 1);
            if (err) return none;
            F.insert(F.end(), Fphi.begin(), Fphi.end());
            return F;
        }
        fail("oai_icon_create: node %d has an unknown kind", node);
        return none;
    }
};

int plan_tree(oai_icon* h) {
    TreePlanner P;
    P.h = h;
    int d = h->D, hh = h->H, w = h->W;
    for (int l = 0; l < 8; ++l) {
        P.gd[l][0] = d; P.gd[l][1] = hh; P.gd[l][2] = w;
        P.vox[l] = (long long)d * hh * w;
        d = (d + 1) / 2; hh = (hh + 1) / 2; w = (w + 1) / 2;
    }
    P.node_used.assign(h->nodes.size(), 0);
    P.net_used.assign(h->net.size(), 0);
    h->steps.clear();
    h->off_A = P.take(P.vox[0]); h->off_B = P.take(P.vox[0]);
    std::vector<Link> F = P.eval(h->root, h->off_A, h->off_B, 0, 0);
    if (P.err) return P.err;
    if ((int)F.size() > OAI_WARP_CHAIN_MAX_FIELDS)
        return oai::set_error(OAI_ERR_ARG, "oai_icon_create: %d steps in the final chain (limit %d)", (int)F.size(), OAI_WARP_CHAIN_MAX_FIELDS);
    for (size_t n = 0; n < P.net_used.size(); ++n)
        if (!P.net_used[n]) return oai::set_error(OAI_ERR_ARG, "oai_icon_create: net %d is not used by the step tree", (int)n);
    for (size_t n = 0; n < P.node_used.size(); ++n)
        if (!P.node_used[n]) return oai::set_error(OAI_ERR_ARG, "oai_icon_create: node %d is not reachable from the root", (int)n);
    h->off_phi = P.take(3 * P.vox[0]);
    P.chain(F, 0, false, 0, h->off_phi);
    h->chain_len = (int)F.size();
    int top = 0;
    while (top < 8 && !h->nets_at_level[top]) ++top;          // the largest grid a U-Net runs on sizes the shared U-Net scratch
    h->off_unet = P.take(unet_ws_floats(P.gd[top][0], P.gd[top][1], P.gd[top][2]));
    h->ws_floats = P.cur;
    return OAI_OK;
}

}  // namespace

extern "C" {

int oai_icon_create(const oai_icon_unet_params* nets, int n_nets, const oai_icon_node* nodes, int n_nodes, int root,
                    int D, int H, int W, oai_icon** out) {
    OAI_CHECK_ARG(nets && nodes && out, "oai_icon_create: null pointer");
    OAI_CHECK_ARG(n_nets >= 1 && n_nets <= OAI_WARP_CHAIN_MAX_FIELDS, "oai_icon_create: 1..%d U-Nets", OAI_WARP_CHAIN_MAX_FIELDS);
    OAI_CHECK_ARG(n_nodes >= 1 && n_nodes <= 64 && root >= 0 && root < n_nodes, "oai_icon_create: bad node list");
    OAI_CHECK_ARG(D > 1 && H > 1 && W > 1, "oai_icon_create: bad network shape");
    oai_icon* h = new oai_icon();
    h->D = D; h->H = H; h->W = W;
    h->net.resize(n_nets);
    h->nodes.assign(nodes, nodes + n_nodes);
    h->root = root;
    int rc = plan_tree(h);
    for (int n = 0; n < n_nets && rc == OAI_OK; ++n) {
        const oai_icon_unet_params& p = nets[n];
        NetWeights& nw = h->net[n];
        for (int l = 0; l < 5 && rc == OAI_OK; ++l) {
            if (!p.down_w[l] || !p.down_b[l] || !p.up_w[l] || !p.up_b[l]) {
                rc = oai::set_error(OAI_ERR_ARG, "oai_icon_create: net %d level %d has a null parameter", n, l);
                break;
            }
            // BatchNorm3d after the up-conv: all four arrays, or none of them = no normalisation (identity).  Whether the package's
            // UNet2.forward applies batchNorms[depth] cannot be checked here (icon_registration 1.1.2 is not vendored): the caller decides.
            const int nbn = (p.bn_gamma[l] != nullptr) + (p.bn_beta[l] != nullptr) + (p.bn_mean[l] != nullptr) + (p.bn_var[l] != nullptr);
            if (nbn != 0 && nbn != 4) {
                rc = oai::set_error(OAI_ERR_ARG, "oai_icon_create: net %d level %d: BatchNorm needs gamma, beta, mean and var (or none of them)", n, l);
                break;
            }
            if ((rc = upload(h, repack_conv(p.down_w[l], kDown[l + 1], kDown[l], 27), &nw.down_w[l]))) break;
            if ((rc = upload(h, std::vector<float>(p.down_b[l], p.down_b[l] + kDown[l + 1]), &nw.down_b[l]))) break;
            if ((rc = upload(h, repack_convT(p.up_w[l], kUpIn[l], kUpOut[l], 64), &nw.up_w[l]))) break;
            if ((rc = upload(h, std::vector<float>(p.up_b[l], p.up_b[l] + kUpOut[l]), &nw.up_b[l]))) break;
            std::vector<float> s(kUpOut[l]), t(kUpOut[l]);
            for (int c = 0; c < kUpOut[l]; ++c) {
                s[c] = nbn ? p.bn_gamma[l][c] / sqrtf(p.bn_var[l][c] + kBnEps) : 1.0f;
                t[c] = nbn ? p.bn_beta[l][c] - p.bn_mean[l][c] * s[c] : 0.0f;
            }
            if ((rc = upload(h, s, &nw.bn_s[l]))) break;
            if ((rc = upload(h, t, &nw.bn_t[l]))) break;
        }
        if (rc) break;
        if (!p.last_w || !p.last_b) { rc = oai::set_error(OAI_ERR_ARG, "oai_icon_create: net %d lastConv is null", n); break; }
        if ((rc = upload(h, repack_conv(p.last_w, 3, 18, 27), &nw.last_w))) break;
        if ((rc = upload(h, std::vector<float>(p.last_b, p.last_b + 3), &nw.last_b))) break;
    }
    if (rc) { oai_icon_destroy(h); return rc; }
    *out = h;
    return OAI_OK;
}

void oai_icon_destroy(oai_icon* h) {
    if (!h) return;
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

size_t oai_icon_workspace_bytes(const oai_icon* h) {
    if (!h) return 0;
    return h->ws_floats * sizeof(float);
}

int oai_icon_describe(const oai_icon* h, int* n_nets, int* levels, int* chain_len) {
    OAI_CHECK_ARG(h, "oai_icon_describe: null handle");
    if (n_nets) *n_nets = (int)h->net.size();
    if (levels) for (int l = 0; l < 8; ++l) levels[l] = h->nets_at_level[l];
    if (chain_len) *chain_len = h->chain_len;
    return OAI_OK;
}

int oai_icon_unet_forward(oai_icon* h, int which, const float* a, const float* b, int D, int H, int W, float* out,
                          void* ws, size_t ws_bytes, void* stream) {
    OAI_CHECK_ARG(h && a && b && out && ws, "oai_icon_unet_forward: null pointer");
    OAI_CHECK_ARG(which >= 0 && which < (int)h->net.size(), "oai_icon_unet_forward: net index must be 0..%d", (int)h->net.size() - 1);
    OAI_CHECK_ARG(dims_ok(D, H, W), "oai_icon_unet_forward: %dx%dx%d too small for five 2x poolings (each axis >= 17)", D, H, W);
    if (unet_ws_floats(D, H, W) * 4 > ws_bytes)
        return oai::set_error(OAI_ERR_WORKSPACE, "oai_icon_unet_forward: workspace %zu B < %zu B", ws_bytes, unet_ws_floats(D, H, W) * 4);
    return unet_forward(h->net[which], h->pad_front, a, b, D, H, W, out, (float*)ws, (hipStream_t)stream);
}

// the launches of one direction, on `st`, reading A / B and writing phi at their fixed homes inside the workspace
static int icon_forward_body(oai_icon* h, float* ws, hipStream_t st) {
    int gd[8][3];
    int d = h->D, hh = h->H, w = h->W;
    for (int l = 0; l < 8; ++l) { gd[l][0] = d; gd[l][1] = hh; gd[l][2] = w; d = (d + 1) / 2; hh = (hh + 1) / 2; w = (w + 1) / 2; }
    for (const IconStep& s : h->steps) {
        const int* g = gd[s.lvl];
        int rc = OAI_OK;
        if (s.kind == IconStep::POOL) {                                                   // DownsampleRegistration.forward
            rc = oai_avgpool2_3d(ws + s.src, 1, g[0], g[1], g[2], ws + s.dst, st);
        } else if (s.kind == IconStep::UNET) {                                            // FunctionFromVectorField: d = net(A, B)
            rc = unet_forward(h->net[s.net], h->pad_front, ws + s.src, ws + s.src2, g[0], g[1], g[2], ws + s.dst, ws + h->off_unet, st);
        } else {
            // the warp / compose closures run as fused chains (oai_warp_chain, warp.hip): bit-identical to the op-by-op sequence of
            // oai_compose / oai_grid_sample3d calls (tests/test_warp_gpu.py), no intermediate map is materialised
            const float* f[OAI_WARP_CHAIN_MAX_FIELDS];
            int fdims[3 * OAI_WARP_CHAIN_MAX_FIELDS];
            for (int i = 0; i < s.nf; ++i) {
                f[i] = ws + s.field[i];
                for (int k = 0; k < 3; ++k) fdims[3 * i + k] = gd[s.field_lvl[i]][k];
            }
            rc = oai_warp_chain(s.has_start ? ws + s.start : nullptr, g[0], g[1], g[2], s.nf, f, fdims,
                                s.has_image ? ws + s.src : nullptr, g[0], g[1], g[2], ws + s.dst, st);
        }
        if (rc) return rc;
    }
    return OAI_OK;
}

int oai_icon_forward(oai_icon* h, const float* A, const float* B, float* phi, void* ws, size_t ws_bytes, void* stream) {
    OAI_CHECK_ARG(h && A && B && phi && ws, "oai_icon_forward: null pointer");
    const long long vh = (long long)h->D * h->H * h->W;
    if (h->ws_floats * sizeof(float) > ws_bytes)
        return oai::set_error(OAI_ERR_WORKSPACE, "oai_icon_forward: workspace %zu B < %zu B", ws_bytes, h->ws_floats * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    if (int rc = copy_f32(A, wsf + h->off_A, vh, st)) return rc;
    if (int rc = copy_f32(B, wsf + h->off_B, vh, st)) return rc;
    bool replayed = false;
    if (h->use_graph && !h->graph_broken) {
        if (!h->gexec || h->g_ws != ws) {                       // first call (or another workspace): capture, do not execute
            if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
            if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) h->graph_broken = true;
            hipGraph_t g = nullptr;
            if (!h->graph_broken && hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = icon_forward_body(h, wsf, h->cap_stream);
                const hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
                if (rc != OAI_OK || e != hipSuccess || !g || hipGraphInstantiate(&h->gexec, g, nullptr, nullptr, 0) != hipSuccess) {
                    h->gexec = nullptr;
                    h->graph_broken = true;                     // not an error: the same launches run directly below (oai_icon_graph_info tells)
                }
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
            } else h->graph_broken = true;
            h->g_ws = ws;
        }
        if (h->gexec) {
            OAI_CHECK_HIP(hipGraphLaunch(h->gexec, st));
            ++h->replays;
            replayed = true;
        }
    }
    if (!replayed) {
        if (int rc = icon_forward_body(h, wsf, st)) return rc;
        ++h->direct_runs;
    }
    return copy_f32(wsf + h->off_phi, phi, 3 * vh, st);
}

int oai_icon_set_graph(oai_icon* h, int enable) {
    OAI_CHECK_ARG(h, "oai_icon_set_graph: null handle");
    h->use_graph = enable != 0;
    return OAI_OK;
}

int oai_icon_set_option(oai_icon* h, const char* name, int value) {
    OAI_CHECK_ARG(h && name, "oai_icon_set_option: null pointer");
    if (!strcmp(name, "pad_front")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_icon_set_option: pad_front must be 0 or 1");
        if (h->pad_front != (value != 0)) {
            h->pad_front = value != 0;
            if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }     // the captured launches carry the old offset
            h->g_ws = nullptr;
        }
        return OAI_OK;
    }
    return oai::set_error(OAI_ERR_ARG, "oai_icon_set_option: unknown option '%s'", name);
}

int oai_icon_graph_info(const oai_icon* h, int* captured, long long* replays, long long* direct_runs) {
    OAI_CHECK_ARG(h, "oai_icon_graph_info: null handle");
    if (captured) *captured = h->gexec ? 1 : (h->graph_broken ? -1 : 0);
    if (replays) *replays = h->replays;
    if (direct_runs) *direct_runs = h->direct_runs;
    return OAI_OK;
}

}  // extern "C"
