// Host side of the segmentation path: weight ingest (reference layouts -> MFMA panels),
// workspace planning, the layer schedule of UNet.forward (networks.py:109-149) with the
// bit-identical dead-output trim, and the C ABI.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "unet_kernels.h"
#include "unet_sres.h"
#include "unet_sres2.h"
#include "unet_wino.h"
#include "unet_wino_f32.h"

namespace oai {

enum LayerId { EC0, EC1, EC2, EC3, EC4, EC5, EC6, EC7, DC9, DC8, DC7, DC6, DC5, DC4, DC3, DC2, DC1, DC0 };
__host__ __device__ static inline int layer_level(int k) { const int v[18] = {0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0, 0}; return v[k]; }   // OUTPUT level
__host__ __device__ static inline int layer_kind(int k) { const int v[18] = {0, 0, 0, 0, 0, 0, 0, 0, 2, 1, 1, 2, 1, 1, 2, 1, 1, 3}; return v[k]; }
static const int kKind[18] = {0, 0, 0, 0, 0, 0, 0, 0, 2, 1, 1, 2, 1, 1, 2, 1, 1, 3};

struct Layer {
    int kind = 0, cin = 0, cout = 0;
    int c0 = 0, c1 = 0;                 // concat split of cin (c1 = skip channels)
    float4* panel = nullptr;            // MFMA weight panel (kinds 0,1,2 except ec0)
    float4* panel_bf[3] = {nullptr, nullptr, nullptr};   // split panels of the k3 layers: bf16 x2 terms, bf16 x3 terms, fp16 x2 terms
    // fp16x3 epilogue affine (refresh_fp16_affine): scale / ws * 2^(e_out - e_in0), shift * 2^e_out with ws = the fp16 panel's per-cout
    // power-of-two weight scale and e = the activation exponents of the layer's output / first input tensor (oai_unet::act_exp)
    float* scale_f16 = nullptr;
    float* shift_f16 = nullptr;
    float* plain_f16 = nullptr;         // dc0 in fp16x3: head weights * 2^-e(dc1)
    std::vector<float> ws;              // per-cout weight scale of the fp16 panel (exact powers of two)
    int rel1 = 0;                       // exponent folded into the source-1 (skip) weights of the current fp16 panel: e(src0) - e(src1)
    size_t panel_f16_floats = 0;
    float4* panel_wino_f32 = nullptr;   // fp32 panel of conv3_wino_f32 (pack_wino_f32_panel): the x axis in Winograd F(2,3) form, exact-fp32 path
    float4* panel_wino = nullptr;       // fp16 panel of conv3_wino_sres (pack_wino_panel): the x axis in Winograd F(2,3) form; packed with the same ws / rel1
    size_t panel_wino_floats = 0;
    float4* panel_wino16 = nullptr;     // the same weights laid out for the 16x16x32 taps of conv3_wino_sres<..., M16> (pack_wino16_panel; Cout % 128 == 0 only)
    size_t panel_wino16_floats = 0;
    float4* panel_m16 = nullptr;        // the fp16 panel laid out for the 16x16x32 tap pairs of conv3_igemm_sres<..., M16> (pack_conv3_m16_panel): same values as panel_bf[2]
    size_t panel_m16_floats = 0;
    std::vector<float> wk_host;         // canonical [27][cin][cout] weights of the k3 layers (for re-packing)
    std::vector<float> scale_host, shift_host, plain_host;
    float* plain = nullptr;             // ec0: [27][cout]; dc0: [ncls][cin]
    float* scale = nullptr;
    float* shift = nullptr;
};

}  // namespace oai

struct oai_unet {
    oai::Layer L[18];
    int variant = 0;                    // 0: MREP4/KC8, 1: MREP2/KC16
    int precision = OAI_PREC_F32;
    int* range_flag = nullptr;          // device word set by the split-fp16 kernels when an activation exceeds fp16's range
    unsigned* census = nullptr;         // [18][16] float bits: max |stored activation| per layer since the last reset (census_note)
    int* eval_out = nullptr;            // device scratch of oai_unet_range_flag
    // Per-tensor activation exponents of the fp16x3 path: layer k stores x * 2^act_exp[k].  fp16's low split term needs |x| >= 2^-3 to be
    // normal; below that the pair (h0, h1) has an ABSOLUTE error floor of 2^-25, so a tensor whose values sit near 2^-10 would carry
    // 1e-5 relative error.  Calibration (oai_unet_calibrate_step) puts every layer's maximum in [2^10, 2^11): 5 bits of
    // headroom below 65504 and a floor of 2^-35 of the maximum.  Exact: powers of two fold into the epilogue affine and the panels.
    int act_exp[18] = {0};
    bool calibrated = false;
    int opt_wino_f32 = 1;               // option "winograd_f32": the plain k3 layers of the exact-fp32 path run conv3_wino_f32 (x axis in Winograd F(2,3) form: 2/3 of the fp32 MFMAs)
    int opt_first_blocks = 24;          // option "first_blocks": workgroups per tile of the ec0 kernel conv3_first_sres_kernel (each walks the tile's voxel pairs with a grid stride, the next pair's gathers under the current pair's FMAs)
    int opt_up_nbw = 0;                 // option "up_nbw": column blocks per workgroup of the k2s2 up-conv kernel: 0 = as many as keep >= 16 workgroups per slot, 1 = one (rounds 1-5), n = at most n
    int opt_shared = 1;                 // option "shared_enc": ec0 -> ec1 computed ONCE over the reflect-padded volume + a 2-voxel shell per tile (oai_segment_tiles)
    int opt_wide = 1;                   // option "wide": layers with Cout % 128 == 0 run conv3_igemm_sres2 (one 8-wave workgroup per CU, double-buffered halo)
    int opt_wino = 19;                  // option "winograd": plain k3 layers with Cout % 64 == 0 run conv3_wino_sres (x axis in Winograd F(2,3) form: 2/3 of the MFMAs);
                                        // bit 4 (round 4): the taps of the two-group form on v_mfma_f32_16x16x32_f16 (same cycles per FLOP, +14 % clock at the power wall);
                                        // bit 5: the same for the 64-cout layer (dc2) -- ALL its launch shapes (specialised main blocks and x strip, slice-split y strip:
                                        // wino_m16_64), so a voxel's bits do not depend on the shape that covers it; measured within noise of bit 4 alone (-0.2 %), not the default
    int opt_wino_layers = 0x3FFFF;      // option "winograd_layers": bit k = layer k may take the Winograd kernel (A/B of single layers)
    int opt_m16 = 1;                    // option "m16": the direct split-resident kernel (conv3_igemm_sres: ec1 + fused ec0, ec2, dc1 + fused head, ...) on
                                        // v_mfma_f32_16x16x32_f16 tap pairs (round 5; another summation order) for every layer that can never take the bit-identical
                                        // 128-cout form conv3_igemm_sres2 -- i.e. Cout % 128 != 0 -- so that a layer runs ONE order whatever shapes cover it and
                                        // option "wide" stays bit-preserving
    int opt_m16_layers = 0x3FFFF;       // option "m16_layers": bit k = layer k may take the tap-pair form (A/B of single layers)
    int opt_dead_stores = 1;            // option "dead_stores": 1 = the encoder does not write the part of a skip tensor that the decoder never reads
    int opt_census = 1;                 // option "census": 0 = the kernels do not record the per-layer maxima (A/B timing of the bookkeeping; no LOW flag)
    unsigned char* zero_rec = nullptr;  // 64 zero bytes: source of halo voxels outside the tile for the LDS-DMA staging
    int opt_persist = 0;                // option "persistent": bit 0 = the specialised 64-cout Winograd form (dc2) with persistent workgroups, the staging waves one block
                                        // ahead (conv3_wino_sres<..., PS>; bit-identical maps)
    int* ps_plan = nullptr;             // the block plan of a persistent launch (wino_plan_kernel): 288 + 8 x 256 ints
    int n_cus = 256;                    // compute units of the device (persistent launches: one workgroup per CU)
    int xcd_group = 32;                 // logical blocks per XCD deal (option "xcd_group"; 0 = launch order)
    bool sres_ring = false;             // MREP 2 with the six-slot z-plane ring (option "sres_ring")
    int b_lds = 0;                      // weight fragments through a three-slot LDS ring shared by the workgroup (option "b_lds")
    int fuse_first = 1;                 // ec0 computed inside ec1's halo staging when ec1 is one main-shape launch (option "fuse_first")
    int sres_mrep = 4;                  // z slices per block of the split-resident conv kernel (option "sres_mrep" 2|4; 2 runs three workgroups per CU: -2 % on 32 border tiles, +0.5 % on the whole volume)
    bool opt_sres = true;               // option "sres": fp16x3 uses the split-resident kernels
    bool sres = false;                  // fp16x3 runs split-resident (activations stored as fp16 term pairs, unet_sres.h)
    int n_classes = 0;
    std::vector<void*> allocs;
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;      // reused start/stop pairs
    size_t ev_used = 0;
};

#ifdef OAI_DIAG
static unsigned long long* g_diag_stamps = nullptr;
static unsigned long long* diag_stamps() {
    if (!g_diag_stamps && oai::diag_env("OAI_STAMPS", 0)) {
        if (hipMalloc((void**)&g_diag_stamps, 32 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(g_diag_stamps, 0, 32 * sizeof(unsigned long long));
    }
    return g_diag_stamps;
}
extern "C" int oai_diag_stamps(unsigned long long out[32], int reset) {      // diagnostic builds only (not in include/oai_hip.h)
    if (!g_diag_stamps) return 1;
    if (hipMemcpy(out, g_diag_stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (reset) (void)hipMemset(g_diag_stamps, 0, 32 * sizeof(unsigned long long));
    return 0;
}
#endif

namespace oai {

static int conv_kc(int variant) { return variant == 1 ? 16 : 8; }

// Wk[tap][cin][cout] of the equivalent correlation (Conv3d as is; ConvTranspose3d k3 s1 p1 = conv with
// the kernel flipped and in/out swapped).
static std::vector<float> canonical_k3(const oai_layer_params& p) {
    std::vector<float> w((size_t)27 * p.cin * p.cout);
    for (int t = 0; t < 27; ++t)
        for (int ci = 0; ci < p.cin; ++ci)
            for (int co = 0; co < p.cout; ++co) {
                float v;
                if (p.kind == 0) v = p.weight_host[((size_t)co * p.cin + ci) * 27 + t];
                else v = p.weight_host[((size_t)ci * p.cout + co) * 27 + (26 - t)];   // flip all three axes
                w[((size_t)t * p.cin + ci) * p.cout + co] = v;
            }
    return w;
}

// Panel order = consumption order of conv3_igemm_f32: [cb][chunk][tap][kg][nr][lane] x float4, where the
// float4 of lane (half h, column j) holds channels 4h..4h+3 of the k-group for cout cb*64+nr*32+j.
static std::vector<float> pack_conv3_panel(const std::vector<float>& wk, int C0, int C1, int Cout, int KC) {
    const int Cin = C0 + C1, KG = KC / 8;
    const int ncb = (Cout + 63) / 64, nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
    std::vector<float> out(((size_t)ncb * (nch0 + nch1) * 27 * KG * 2 + 2) * 64 * 4, 0.0f);   // +1 step of prefetch slack
    size_t o = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nch0 + nch1; ++ch) {
            const bool first = ch < nch0;
            const int Csrc = first ? C0 : C1, cofs = first ? 0 : C0, cl0 = (first ? ch : ch - nch0) * KC;
            for (int t = 0; t < 27; ++t)
                for (int kg = 0; kg < KG; ++kg)
                    for (int nr = 0; nr < 2; ++nr)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int s = 0; s < 4; ++s, ++o) {
                                const int cl = cl0 + kg * 8 + 4 * (lane >> 5) + s;
                                const int co = cb * 64 + nr * 32 + (lane & 31);
                                if (cl < Csrc && co < Cout) out[o] = wk[((size_t)t * Cin + cofs + cl) * Cout + co];
                            }
        }
    return out;
}

// conv3_wino_f32 (unet_wino_f32.h): the x-transformed weights u0 = g0, u1 = (g0 + g1 + g2) / 2, u2 = (g0 - g1 + g2) / 2, u3 = g2 per (dz, dy, cin, cout), formed in double
// and rounded to fp32 once; layout [cout block 64][frequency 4][chunk of 8 channels][tap (dz, dy) 9][cout half 2][lane 64][4]: lane (column = lane & 31, k half =
// lane >> 5) holds channels 4 (lane >> 5) .. + 3 of cout 32 nr + column -- the B operands of four v_mfma_f32_32x32x2_f32 k-steps, like pack_conv3_panel.
static std::vector<float> pack_wino_f32_panel(const std::vector<float>& wk, int C0, int C1, int Cout) {
    constexpr int KC = 8;
    const int Cin = C0 + C1;
    const int ncb = (Cout + 63) / 64, nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
    std::vector<float> out(((size_t)ncb * 4 * (nch0 + nch1) * 9 * 2 + 6) * 64 * 4, 0.0f);     // + three taps of prefetch slack
    size_t o = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int f = 0; f < 4; ++f)
            for (int ch = 0; ch < nch0 + nch1; ++ch) {
                const bool first = ch < nch0;
                const int Csrc = first ? C0 : C1, cofs = first ? 0 : C0, cl0 = (first ? ch : ch - nch0) * KC;
                for (int q = 0; q < 9; ++q)
                    for (int nr = 0; nr < 2; ++nr)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int s = 0; s < 4; ++s, ++o) {
                                const int cl = cl0 + 4 * (lane >> 5) + s;
                                const int co = cb * 64 + nr * 32 + (lane & 31);
                                if (cl >= Csrc || co >= Cout) continue;
                                const double g0 = wk[((size_t)(q * 3 + 0) * Cin + cofs + cl) * Cout + co], g1 = wk[((size_t)(q * 3 + 1) * Cin + cofs + cl) * Cout + co],
                                             g2 = wk[((size_t)(q * 3 + 2) * Cin + cofs + cl) * Cout + co];
                                out[o] = (float)(f == 0 ? g0 : f == 1 ? (g0 + g1 + g2) * 0.5 : f == 2 ? (g0 - g1 + g2) * 0.5 : g2);
                            }
            }
    return out;
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline uint16_t f32_to_f16_rne(float f) {            // IEEE binary16, round to nearest even, subnormals kept
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                    // rounds to >= 65520 -> inf
    if (u < 0x33000001u) return (uint16_t)sign;                                 // < 2^-25 -> 0
    int e = (int)(u >> 23) - 127;
    uint32_t m = (u & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? (-14 - e) + 13 : 13;                                  // subnormal: extra shift
    uint32_t half = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) ++half;
    uint32_t out = e < -14 ? half : (((uint32_t)(e + 15) << 10) + (half - 0x400u));   // carry propagates into the exponent
    return (uint16_t)(sign | out);
}
static inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } u = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13); }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Split-bf16 panel of conv3_igemm_bf16s: [cb][chunk of 16][tap][term][nr][lane] x 8 bf16, where lane (half h, column j)
// holds channels 8h..8h+7 of the chunk for cout cb*64+nr*32+j.  Terms: w = t0 + t1 (+ t2), each the RNE bf16 of the rest.
// fp16: `wscale[co]` (an exact power of two) multiplies every weight of output channel co before the split.
// `src1_factor` (an exact power of two) multiplies the weights of the source-1 (skip) channels: the two sources of a concat layer are stored
// with different activation exponents and the panel carries their ratio.
static std::vector<float> pack_conv3_panel_bf(const std::vector<float>& wk, int C0, int C1, int Cout, int NS,
                                              bool fp16 = false, const std::vector<float>* wscale = nullptr, float src1_factor = 1.0f) {
    const int Cin = C0 + C1, KC = 16;
    const int ncb = (Cout + 63) / 64, nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
    const size_t units = ((size_t)ncb * (nch0 + nch1) * 27 + 1) * NS * 2 * 64;      // 16-byte units, +1 tap of prefetch slack
    std::vector<float> out(units * 4, 0.0f);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
    size_t u = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nch0 + nch1; ++ch) {
            const bool first = ch < nch0;
            const int Csrc = first ? C0 : C1, cofs = first ? 0 : C0, cl0 = (first ? ch : ch - nch0) * KC;
            for (int t = 0; t < 27; ++t)
                for (int k = 0; k < NS; ++k)
                    for (int nr = 0; nr < 2; ++nr)
                        for (int lane = 0; lane < 64; ++lane, ++u)
                            for (int j = 0; j < 8; ++j) {
                                const int cl = cl0 + 8 * (lane >> 5) + j;
                                const int co = cb * 64 + nr * 32 + (lane & 31);
                                if (cl < Csrc && co < Cout) {
                                    float r = wk[((size_t)t * Cin + cofs + cl) * Cout + co] * (wscale ? (*wscale)[co] : 1.0f) * (first ? 1.0f : src1_factor);
                                    uint16_t b = 0;
                                    for (int kk = 0; kk <= k; ++kk) {
                                        if (fp16) { b = f32_to_f16_rne(r); r -= f16_to_f32(b); }
                                        else { b = f32_to_bf16_rne(r); r -= bf16_to_f32(b); }
                                    }
                                    o16[u * 8 + j] = b;
                                }
                            }
        }
    return out;
}

// Tap of step j (0..13), lane half tsel, of conv3_igemm_sres<..., M16>: 27 taps = 13 pairs + 1 (see the kernel's header comment).  t = (dz * 3 + dy) * 3 + dx.
static inline int m16_step_tap(int j, int tsel) {
    if (j < 9) return 3 * j + tsel;                          // (q = j, dx 0) | (q, dx 1)
    if (j < 12) return 3 * (3 * (j - 9) + tsel) + 2;         // (q, dx 2) | (q + 1, dx 2), q = 0, 3, 6
    if (j == 12) return 3 * (2 + 3 * tsel) + 2;              // (2, dx 2) | (5, dx 2)
    return 26;                                               // (8, dx 2) alone
}

// Panel of conv3_igemm_sres<..., M16> (v_mfma_f32_16x16x32_f16, K = a PAIR of taps x 16 channels): [cb][chunk of 16][step 14][X' | Y'][n2 4][lane] x 8 fp16.
// Lane (column c = lane & 15, group g = lane >> 4) of tile n2 holds cout cb * 64 + n2 * 16 + c, channels 8 (g & 1) .. + 7 of tap m16_step_tap(j, g >> 1):
// X' = the HIGH terms b0 of both taps, Y' = the LOW terms b1.  Step 13 (the 27th tap alone, A = [a0 | a1]): X' = [b0 | b0], Y' = [b1 | 0].  The values are
// pack_conv3_panel_bf's fp16 terms (same scales, same split).  + 1 step of prefetch slack (the kernel requests one step past its last).
static std::vector<float> pack_conv3_m16_panel(const std::vector<float>& wk, int C0, int C1, int Cout, const std::vector<float>& wscale, float src1_factor) {
    const int Cin = C0 + C1, KC = 16;
    const int ncb = (Cout + 63) / 64, nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
    const size_t units = ((size_t)ncb * (nch0 + nch1) * 14 + 1) * 2 * 4 * 64;          // 16-byte units
    std::vector<float> out(units * 4, 0.0f);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
    size_t u = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nch0 + nch1; ++ch) {
            const bool first = ch < nch0;
            const int Csrc = first ? C0 : C1, cofs = first ? 0 : C0, cl0 = (first ? ch : ch - nch0) * KC;
            for (int j = 0; j < 14; ++j)
                for (int kind = 0; kind < 2; ++kind)             // 0: X', 1: Y'
                    for (int n2 = 0; n2 < 4; ++n2)
                        for (int lane = 0; lane < 64; ++lane, ++u) {
                            const int g = lane >> 4, tsel = g >> 1;
                            if (j == 13 && kind == 1 && tsel == 1) continue;           // step 13: Y' = [b1 | 0]
                            const int t = m16_step_tap(j, tsel), term = kind;
                            const int co = cb * 64 + n2 * 16 + (lane & 15);
                            if (co >= Cout) continue;
                            for (int e = 0; e < 8; ++e) {
                                const int cl = cl0 + 8 * (g & 1) + e;
                                if (cl >= Csrc) continue;
                                float r = wk[((size_t)t * Cin + cofs + cl) * Cout + co] * wscale[co] * (first ? 1.0f : src1_factor);
                                uint16_t b = 0;
                                for (int kk = 0; kk <= term; ++kk) { b = f32_to_f16_rne(r); r -= f16_to_f32(b); }
                                o16[u * 8 + e] = b;
                            }
                        }
        }
    return out;
}

// Panel of conv3_wino_sres (unet_wino.h): [cb][frequency f][chunk of 16][tap (dz, dy)][term][nr][lane] x 8 fp16, lane (half h, column j) =
// channels 8h..8h+7 of the chunk for cout cb*64+nr*32+j, of the x-transformed weights
//   u0 = g0, u1 = (g0 + g1 + g2) / 2, u2 = (g0 - g1 + g2) / 2, u3 = g2     (g = the three x taps of (dz, dy, cin, cout)),
// formed in double from the scaled weights (wscale, src1_factor: see pack_conv3_panel_bf), rounded to fp32 once, then split.
static std::vector<float> pack_wino_panel(const std::vector<float>& wk, int C0, int C1, int Cout, const std::vector<float>& wscale, float src1_factor) {
    const int Cin = C0 + C1, KC = 16;
    const int ncb = Cout / 64, nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
    const size_t units = ((size_t)ncb * 4 * (nch0 + nch1) * 9 + 2) * 2 * 2 * 64;      // 16-byte units, +2 taps of prefetch slack (one is read)
    std::vector<float> out(units * 4, 0.0f);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
    size_t u = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int f = 0; f < 4; ++f)
            for (int ch = 0; ch < nch0 + nch1; ++ch) {
                const bool first = ch < nch0;
                const int Csrc = first ? C0 : C1, cofs = first ? 0 : C0, cl0 = (first ? ch : ch - nch0) * KC;
                for (int t = 0; t < 9; ++t)                          // t = dz * 3 + dy
                    for (int k = 0; k < 2; ++k)
                        for (int nr = 0; nr < 2; ++nr)
                            for (int lane = 0; lane < 64; ++lane, ++u)
                                for (int j = 0; j < 8; ++j) {
                                    const int cl = cl0 + 8 * (lane >> 5) + j;
                                    const int co = cb * 64 + nr * 32 + (lane & 31);
                                    if (cl >= Csrc) continue;
                                    double g[3];
                                    for (int dx = 0; dx < 3; ++dx)
                                        g[dx] = (double)wk[((size_t)(t * 3 + dx) * Cin + cofs + cl) * Cout + co] * (double)wscale[co] * (first ? 1.0 : (double)src1_factor);
                                    const double uf = f == 0 ? g[0] : f == 1 ? 0.5 * (g[0] + g[1] + g[2]) : f == 2 ? 0.5 * (g[0] - g[1] + g[2]) : g[2];
                                    float r = (float)uf;
                                    uint16_t b = 0;
                                    for (int kk = 0; kk <= k; ++kk) { b = f32_to_f16_rne(r); r -= f16_to_f32(b); }
                                    o16[u * 8 + j] = b;
                                }
            }
    return out;
}

// Panel of conv3_wino_sres<.., M16> (v_mfma_f32_16x16x32_f16, K = a PAIR of taps x 16 channels): [cb][frequency f][chunk of 16][step 5][X' | Y'][n2 4][lane]
// x 8 fp16.  Lane (column c = lane & 15, group g = lane >> 4) of tile n2 holds cout cb * 64 + n2 * 16 + c, channels 8 (g & 1) .. + 7 of tap
// 2 j + (g >> 1) (steps j < 4): X' = the HIGH terms b0 of both taps, Y' = the LOW terms b1.  Step 4 (the ninth tap alone, A = [a0 | a1]):
// X' = [b0(8) | b0(8)], Y' = [b1(8) | 0].  The values are pack_wino_panel's (same transformed weights, same scales, same split).
static std::vector<float> pack_wino16_panel(const std::vector<float>& wk, int C0, int C1, int Cout, const std::vector<float>& wscale, float src1_factor) {
    const int Cin = C0 + C1, KC = 16;
    const int ncb = Cout / 64, nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
    const size_t units = ((size_t)ncb * 4 * (nch0 + nch1) * 5 + 2) * 2 * 4 * 64;       // 16-byte units, + 2 steps of prefetch slack
    std::vector<float> out(units * 4, 0.0f);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
    size_t u = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int f = 0; f < 4; ++f)
            for (int ch = 0; ch < nch0 + nch1; ++ch) {
                const bool first = ch < nch0;
                const int Csrc = first ? C0 : C1, cofs = first ? 0 : C0, cl0 = (first ? ch : ch - nch0) * KC;
                for (int j = 0; j < 5; ++j)
                    for (int kind = 0; kind < 2; ++kind)             // 0: X', 1: Y'
                        for (int n2 = 0; n2 < 4; ++n2)
                            for (int lane = 0; lane < 64; ++lane, ++u) {
                                const int g = lane >> 4, tsel = g >> 1;
                                int t, term;
                                if (j < 4) { t = 2 * j + tsel; term = kind; }
                                else { t = 8; term = kind; if (kind == 1 && tsel == 1) continue; }     // step 4: X' = [b0 | b0], Y' = [b1 | 0]
                                const int co = cb * 64 + n2 * 16 + (lane & 15);
                                for (int e = 0; e < 8; ++e) {
                                    const int cl = cl0 + 8 * (g & 1) + e;
                                    if (cl >= Csrc) continue;
                                    double gx[3];
                                    for (int dx = 0; dx < 3; ++dx)
                                        gx[dx] = (double)wk[((size_t)(t * 3 + dx) * Cin + cofs + cl) * Cout + co] * (double)wscale[co] * (first ? 1.0 : (double)src1_factor);
                                    const double uf = f == 0 ? gx[0] : f == 1 ? 0.5 * (gx[0] + gx[1] + gx[2]) : f == 2 ? 0.5 * (gx[0] - gx[1] + gx[2]) : gx[2];
                                    float r = (float)uf;
                                    uint16_t b = 0;
                                    for (int kk = 0; kk <= term; ++kk) { b = f32_to_f16_rne(r); r -= f16_to_f32(b); }
                                    o16[u * 8 + e] = b;
                                }
                            }
            }
    return out;
}

// [N/64][Cin/8][2][lane] x float4 for upconv2_igemm_f32; column n = parity*Cout + co
static std::vector<float> pack_up_panel(const oai_layer_params& p) {
    const int N = 8 * p.cout, nnb = (N + 63) / 64, nkg = (p.cin + 7) / 8;
    std::vector<float> out((size_t)nnb * nkg * 2 * 64 * 4, 0.0f);
    size_t o = 0;
    for (int nb = 0; nb < nnb; ++nb)
        for (int kg = 0; kg < nkg; ++kg)
            for (int nr = 0; nr < 2; ++nr)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s, ++o) {
                        const int ci = kg * 8 + 4 * (lane >> 5) + s;
                        const int col = nb * 64 + nr * 32 + (lane & 31);
                        if (ci < p.cin && col < N) {
                            const int par = col / p.cout, co = col % p.cout;
                            out[o] = p.weight_host[((size_t)ci * p.cout + co) * 8 + par];   // [ci][co][a][b][c]
                        }
                    }
    return out;
}

// split-fp16 panel of upconv2_igemm<true>: [N/64][Cin/16][term 2][nr 2][lane] x 8 fp16; lane (half h, column j) holds
// channels 16*ks + 8h .. +7 for column nb*64 + nr*32 + j (column n = parity*Cout + co); weights of output channel co are
// first multiplied by wscale[co] (exact power of two)
static std::vector<float> pack_up_panel_f16(const float* w /*[ci][co][8]*/, int cin, int cout, const std::vector<float>& wscale) {
    const int N = 8 * cout, nnb = (N + 63) / 64, nks = (cin + 15) / 16;
    std::vector<float> out((size_t)nnb * nks * 4 * 64 * 4, 0.0f);
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out.data());
    size_t u = 0;
    for (int nb = 0; nb < nnb; ++nb)
        for (int ks = 0; ks < nks; ++ks)
            for (int k = 0; k < 2; ++k)
                for (int nr = 0; nr < 2; ++nr)
                    for (int lane = 0; lane < 64; ++lane, ++u)
                        for (int j = 0; j < 8; ++j) {
                            const int ci = ks * 16 + 8 * (lane >> 5) + j;
                            const int col = nb * 64 + nr * 32 + (lane & 31);
                            if (ci < cin && col < N) {
                                const int par = col / cout, co = col % cout;
                                float r = w[((size_t)ci * cout + co) * 8 + par] * wscale[co];
                                uint16_t b = 0;
                                for (int kk = 0; kk <= k; ++kk) { b = f32_to_f16_rne(r); r -= f16_to_f32(b); }
                                o16[u * 8 + j] = b;
                            }
                        }
    return out;
}

template <typename T>
static int upload(oai_unet* h, const std::vector<float>& v, T** dst) {
    void* d = nullptr;
    OAI_CHECK_HIP(hipMalloc(&d, v.size() * sizeof(float)));
    h->allocs.push_back(d);
    OAI_CHECK_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *dst = reinterpret_cast<T*>(d);
    return OAI_OK;
}

// (re)fill a device array that already exists (same size) or create it
template <typename T>
static int upload_into(oai_unet* h, const std::vector<float>& v, T** dst) {
    if (!*dst) return upload(h, v, dst);
    OAI_CHECK_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return OAI_OK;
}

// tensors a layer reads: its first source is the previous layer of the schedule (a pooled tensor keeps its producer's exponent);
// dc8 / dc5 / dc2 also read the skips ec5 / ec3 / ec1
static inline int layer_src0(int k) { return k - 1; }
static inline int layer_src1(int k) { return k == DC8 ? EC5 : k == DC5 ? EC3 : k == DC2 ? EC1 : -1; }

// The Winograd panel of layer k (k3 layers with whole blocks of 64 couts) for the weight scales / skip exponent of its current fp16 panel
static int pack_wino_layer(oai_unet* h, int k) {
    Layer& L = h->L[k];
    if (L.kind == 2 || L.wk_host.empty() || L.cout % 64 != 0 || L.ws.empty()) return OAI_OK;
    const std::vector<float> panel = pack_wino_panel(L.wk_host, L.c0, L.c1, L.cout, L.ws, ldexpf(1.0f, L.rel1));
    if (L.panel_wino && panel.size() != L.panel_wino_floats) return set_error(OAI_ERR_ARG, "Winograd panel of layer %d changed size", k);
    L.panel_wino_floats = panel.size();
    if (int rc = upload_into(h, panel, &L.panel_wino)) return rc;
    const std::vector<float> p16 = pack_wino16_panel(L.wk_host, L.c0, L.c1, L.cout, L.ws, ldexpf(1.0f, L.rel1));
    if (L.panel_wino16 && p16.size() != L.panel_wino16_floats) return set_error(OAI_ERR_ARG, "16x16x32 Winograd panel of layer %d changed size", k);
    L.panel_wino16_floats = p16.size();
    return upload_into(h, p16, &L.panel_wino16);
}

// The fp16 panel of layer k (k3 conv or k2s2 up-conv) for the current activation exponents.  Every output channel's weights are scaled
// by the power of two that puts max|w| in [2^7, 2^8) -- exact, undone by the epilogue scale -- so that the low split term of all
// weights down to 2^-11 of the largest stays in fp16's normal range.
static int pack_fp16_layer(oai_unet* h, int k) {
    Layer& L = h->L[k];
    const int s1 = layer_src1(k);
    const int rel1 = s1 >= 0 ? h->act_exp[layer_src0(k)] - h->act_exp[s1] : 0;
    const float f1 = ldexpf(1.0f, rel1);
    L.ws.assign(L.cout, 1.0f);
    std::vector<float> panel;
    if (L.kind == 2) {
        for (int co = 0; co < L.cout; ++co) {
            float amax = 0.0f;
            for (int ci = 0; ci < L.cin; ++ci)
                for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf(L.wk_host[((size_t)ci * L.cout + co) * 8 + q]));
            if (amax > 0.0f && std::isfinite(amax)) { int e; frexpf(amax, &e); L.ws[co] = ldexpf(1.0f, 8 - e); }   // amax = m 2^e, m in [0.5,1)
        }
        panel = pack_up_panel_f16(L.wk_host.data(), L.cin, L.cout, L.ws);
    } else {
        const int Cin = L.c0 + L.c1;
        const size_t per = L.wk_host.size() / L.cout;                 // index i = tap * Cin + ci
        for (int co = 0; co < L.cout; ++co) {
            float amax = 0.0f;
            for (size_t i = 0; i < per; ++i)
                amax = fmaxf(amax, fabsf(L.wk_host[i * L.cout + co]) * ((int)(i % Cin) >= L.c0 ? f1 : 1.0f));
            if (amax > 0.0f && std::isfinite(amax)) { int e; frexpf(amax, &e); L.ws[co] = ldexpf(1.0f, 8 - e); }
        }
        panel = pack_conv3_panel_bf(L.wk_host, L.c0, L.c1, L.cout, 2, true, &L.ws, f1);
    }
    if (L.panel_bf[2] && panel.size() != L.panel_f16_floats) return set_error(OAI_ERR_ARG, "fp16 panel of layer %d changed size", k);
    L.panel_f16_floats = panel.size();
    L.rel1 = rel1;
    if (int rc = upload_into(h, panel, &L.panel_bf[2])) return rc;
    if (L.kind != 2) {                                               // the same terms in the order of the 16x16x32 tap pairs (conv3_igemm_sres<..., M16>)
        const std::vector<float> p16 = pack_conv3_m16_panel(L.wk_host, L.c0, L.c1, L.cout, L.ws, f1);
        if (L.panel_m16 && p16.size() != L.panel_m16_floats) return set_error(OAI_ERR_ARG, "16x16x32 panel of layer %d changed size", k);
        L.panel_m16_floats = p16.size();
        if (int rc = upload_into(h, p16, &L.panel_m16)) return rc;
    }
    return (h->opt_wino || L.panel_wino) ? pack_wino_layer(h, k) : OAI_OK;
}

// The fp16x3 epilogue arrays of every layer for the current activation exponents (see Layer::scale_f16).  ec0 reads the raw volume
// (exponent 0) and has no panel; dc0 reads dc1's records and produces logits (exponent 0): its weights carry 2^-e(dc1).
static int refresh_fp16_affine(oai_unet* h) {
    for (int k = 0; k < 17; ++k) {
        Layer& L = h->L[k];
        const int e_out = h->act_exp[k], e_in = k == EC0 ? 0 : h->act_exp[layer_src0(k)];
        std::vector<float> sc(L.cout), sh(L.cout);
        for (int co = 0; co < L.cout; ++co) {
            const float ws = L.ws.empty() ? 1.0f : L.ws[co];
            sc[co] = ldexpf(L.scale_host[co] / ws, e_out - e_in);
            sh[co] = ldexpf(L.shift_host[co], e_out);
        }
        if (int rc = upload_into(h, sc, &L.scale_f16)) return rc;
        if (int rc = upload_into(h, sh, &L.shift_f16)) return rc;
    }
    Layer& H = h->L[DC0];
    std::vector<float> w(H.plain_host.size());
    for (size_t i = 0; i < w.size(); ++i) w[i] = ldexpf(H.plain_host[i], -h->act_exp[DC1]);
    return upload_into(h, w, &H.plain_f16);
}

struct Box { int lo[3], hi[3]; };

__host__ __device__ static void full_box(Box& b, const int dims[3]) { for (int i = 0; i < 3; ++i) { b.lo[i] = 0; b.hi[i] = dims[i]; } }

// Output box each layer must produce so that the kept centre is unchanged (oracle/seg.py:trim_regions)
__host__ __device__ static void plan_regions(const int tile[3], const int keep_lo[3], const int keep_hi[3], bool trimmed, Box need[18]) {
    int dims[4][3];
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) dims[l][i] = tile[i] >> l;
    for (int k = 0; k < 18; ++k) full_box(need[k], dims[layer_level(k)]);
    for (int i = 0; i < 3; ++i) { need[DC0].lo[i] = keep_lo[i]; need[DC0].hi[i] = keep_hi[i]; }   // the head always crops
    if (!trimmed) return;
    auto grow = [&](const Box& b, int lvl) { Box r; for (int i = 0; i < 3; ++i) { r.lo[i] = b.lo[i] > 0 ? b.lo[i] - 1 : 0; r.hi[i] = b.hi[i] + 1 < dims[lvl][i] ? b.hi[i] + 1 : dims[lvl][i]; } return r; };
    auto halve = [&](const Box& b) { Box r; for (int i = 0; i < 3; ++i) { r.lo[i] = b.lo[i] / 2; r.hi[i] = (b.hi[i] + 1) / 2; } return r; };
    Box keep; for (int i = 0; i < 3; ++i) { keep.lo[i] = keep_lo[i]; keep.hi[i] = keep_hi[i]; }
    need[DC0] = keep; need[DC1] = keep;
    need[DC2] = grow(need[DC1], 0);
    need[DC3] = grow(need[DC2], 0);
    need[DC4] = halve(need[DC3]);
    need[DC5] = grow(need[DC4], 1);
    need[DC6] = grow(need[DC5], 1);
    need[DC7] = halve(need[DC6]);
    need[DC8] = grow(need[DC7], 2);
    // dc9 and the encoder stay full: the bottleneck sees the whole tile
}

struct Plan {
    size_t off[24];
    size_t total;
};
constexpr int kMaxTilesPerTable = 4096;
constexpr size_t kBoxTableBytes = ((size_t)18 * kMaxTilesPerTable * 6 * sizeof(int) + 255) / 256 * 256;
enum Buf { B_E0, B_SYN0, B_P0, B_E2, B_SYN1, B_P1, B_E4, B_SYN2, B_P2, B_E6, B_E7, B_U9, B_D8, B_D7, B_U6, B_D5, B_D4, B_U3, B_D2, B_D1, B_COUNT };

static Plan plan_workspace(const oai_unet* h, int td, int th, int tw, int batch) {
    const size_t v0 = (size_t)td * th * tw, v1 = v0 / 8, v2 = v1 / 8, v3 = v2 / 8;
    const oai::Layer* L = h->L;
    auto pc = [&](int k) { return (size_t)((L[k].cout + 15) / 16 * 16); };   // channels padded to whole 16-chunks (format S)
    const size_t sizes[B_COUNT] = {
        v0 * pc(EC0), v0 * pc(EC1), v1 * pc(EC1), v1 * pc(EC2), v1 * pc(EC3), v2 * pc(EC3),
        v2 * pc(EC4), v2 * pc(EC5), v3 * pc(EC5), v3 * pc(EC6), v3 * pc(EC7), v2 * pc(DC9),
        v2 * pc(DC8), v2 * pc(DC7), v1 * pc(DC6), v1 * pc(DC5), v1 * pc(DC4), v0 * pc(DC3),
        v0 * pc(DC2), v0 * pc(DC1)};
    Plan p;
    size_t o = kBoxTableBytes;          // per-tile box table of the current oai_segment_tiles call lives at the start
    for (int i = 0; i < B_COUNT; ++i) {
        p.off[i] = o;
        o += ((sizes[i] * batch * sizeof(float) + 255) / 256) * 256;
    }
    p.total = o;
    return p;
}

template <int MREP, int KC, int RX, int RY, int WY, int WX>
static int launch_conv3_shape(const oai_unet* h, ConvArgs a, const Box& box, int ntiles, hipStream_t st, int mrep_override = 0) {   // a.boxes set by the caller
    for (int i = 0; i < 3; ++i) { a.lo[i] = box.lo[i]; a.hi[i] = box.hi[i]; }
    if (box.hi[0] <= box.lo[0] || box.hi[1] <= box.lo[1] || box.hi[2] <= box.lo[2]) return OAI_OK;
    // conv3_igemm_sres2 (unet_sres2.h): main shape of the default split-resident configuration, 128 couts per workgroup
    // ... when the launch is at least four rounds of its one-workgroup-per-CU blocks (ec6 of the reference network: 2.5 rounds, 13 % slower
    // than as twice as many 4-wave workgroups)
    const int sres_mrep = mrep_override ? mrep_override : h->sres_mrep;       // z slices per block of the split-resident kernel for THIS launch
    constexpr bool kMainShape = RX == 16 && RY == 2 && WY == 4 && WX == 1;
    // + the 4-row y strip (dc5: 2.27 -> 1.89 ms); the 4-column x strip <4,8,4,1> measured slower there (2.05 -> 2.30 ms) and stays on the 4-wave kernel
    constexpr bool kWideShape = kMainShape || (RX == 16 && RY == 2 && WY == 2 && WX == 2);
    bool wide = KC == 8 && kWideShape && h->sres && h->opt_wide && sres_mrep == 4 && !h->sres_ring &&
                !h->b_lds && !a.first_w && !a.head_w && a.Cout % 128 == 0;
    if (wide && h->opt_wide == 1) {
        const size_t nwg = (size_t)ntiles * cdiv(box.hi[0] - box.lo[0], 4) * cdiv(box.hi[1] - box.lo[1], WY * RY) * cdiv(box.hi[2] - box.lo[2], WX * RX) * (a.Cout / 128);
        wide = nwg >= (kMainShape ? 1024 : 512);
    }
    if (wide) a.ncb = a.Cout / 128;
    const bool bf = KC == 8 && h->precision != OAI_PREC_F32;       // the split kernels use 2 z slices per block (4: split-resident)
    // exact fp32: two z slices per block -- the registers of the other two hold the per-chunk partial sums of its two-level accumulation (conv3_igemm_f32)
    constexpr int kF32Mrep = MREP > 2 ? 2 : MREP;
    a.nbz = cdiv(box.hi[0] - box.lo[0], bf ? (h->sres ? sres_mrep : 2) : kF32Mrep);
    a.nby = cdiv(box.hi[1] - box.lo[1], WY * RY);
    a.nbx = cdiv(box.hi[2] - box.lo[2], WX * RX);
    unsigned grid = (unsigned)((size_t)ntiles * a.nbz * a.nby * a.nbx * a.ncb);
    if (KC == 8 && h->sres && h->xcd_group > 0) {          // XCD-aware dealing of the logical block list (xcd_block_id)
        a.nblocks = (int)grid; a.xcd_group = h->xcd_group;
        const unsigned q = 8u * (unsigned)h->xcd_group;
        grid = (grid + q - 1) / q * q;
    }
    oai_unet* hm = const_cast<oai_unet*>(h);
    if (h->profile) {
        if (hm->ev_used + 2 > hm->ev_pool.size()) {
            hipEvent_t e0, e1;
            OAI_CHECK_HIP(hipEventCreate(&e0));
            OAI_CHECK_HIP(hipEventCreate(&e1));
            hm->ev_pool.push_back(e0);
            hm->ev_pool.push_back(e1);
        }
        OAI_CHECK_HIP(hipEventRecord(hm->ev_pool[hm->ev_used], st));
    }
    if (KC == 8 && h->sres) {
        // (diagnostic builds: OAI_ONE_WG=1 asks for 24 KB of unused dynamic LDS, which leaves room for ONE workgroup per CU -- the tap
        // stream of a wave that has the SIMD to itself, scripts/stamp_phases.py)
        static const int one_wg = diag_env("OAI_ONE_WG", 0);
        bool done = false;
        // the 16x16x32 tap pairs (option "m16"): a per-LAYER decision -- never for a layer that the bit-identical 128-cout form conv3_igemm_sres2 may
        // take for some of its launches (that choice depends on the launch size) -- and every shape of the kernel has the variant
        const bool m16 = h->opt_m16 && a.wpanel16 && !h->sres_ring && !h->b_lds && a.Cout % 128 != 0 && !one_wg;
        if (m16) a.wpanel = a.wpanel16;
        if constexpr (RX == 16 && RY == 2 && WY == 4 && WX == 1) {
            if (a.first_w) {                                         // ec0 fused into ec1's halo staging (first_fusable guarantees mrep 4, no strips)
                if (m16) conv3_igemm_sres<4, RX, RY, WY, WX, false, true, false, true><<<grid, 256, 0, st>>>(a, h->zero_rec);
                else conv3_igemm_sres<4, RX, RY, WY, WX, false, true><<<grid, 256, 0, st>>>(a, h->zero_rec);
                done = true;
            }
        }
        if constexpr (kWideShape) {
            if (wide) {
                static const int var = diag_env("OAI_WIDE_VAR", 0);         // -DOAI_DIAG builds only; constant 0 otherwise
#ifdef OAI_DIAG
                if (var == 1) conv3_igemm_sres2<RX, RY, WY, WX, 1><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 2) conv3_igemm_sres2<RX, RY, WY, WX, 2><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 3) conv3_igemm_sres2<RX, RY, WY, WX, 3><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 4) conv3_igemm_sres2<RX, RY, WY, WX, 4><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 5) conv3_igemm_sres2<RX, RY, WY, WX, 5><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 6) conv3_igemm_sres2<RX, RY, WY, WX, 6><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 7) conv3_igemm_sres2<RX, RY, WY, WX, 7><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 8) conv3_igemm_sres2<RX, RY, WY, WX, 8><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 9) conv3_igemm_sres2<RX, RY, WY, WX, 9><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else if (var == 10) conv3_igemm_sres2<RX, RY, WY, WX, 10><<<grid, 512, 0, st>>>(a, h->zero_rec);
                else
#endif
                conv3_igemm_sres2<RX, RY, WY, WX, 0><<<grid, 512, 0, st>>>(a, h->zero_rec);
                (void)var;
                done = true;
            }
        }
        if (done) { }
        else if (a.first_w) return set_error(OAI_ERR_ARG, "fused ec0 asked of a tile shape that has no such kernel");
        else if (m16 && sres_mrep == 4) conv3_igemm_sres<4, RX, RY, WY, WX, false, false, false, true><<<grid, 256, 0, st>>>(a, h->zero_rec);
        else if (m16) conv3_igemm_sres<2, RX, RY, WY, WX, false, false, false, true><<<grid, 256, 0, st>>>(a, h->zero_rec);
        else if (sres_mrep == 4 && h->b_lds) conv3_igemm_sres<4, RX, RY, WY, WX, false, false, true><<<grid, 256, 0, st>>>(a, h->zero_rec);
        else if (sres_mrep == 4) conv3_igemm_sres<4, RX, RY, WY, WX><<<grid, 256, one_wg ? 24 * 1024 : 0, st>>>(a, h->zero_rec);
        else if (h->sres_ring && !mrep_override) conv3_igemm_sres<2, RX, RY, WY, WX, true><<<grid, 256, 0, st>>>(a, h->zero_rec);
        else conv3_igemm_sres<2, RX, RY, WY, WX><<<grid, 256, 0, st>>>(a, h->zero_rec);
    }
    else if (KC == 8 && h->precision == OAI_PREC_BF16X3) conv3_igemm_bf16s<2, false, 2, RX, RY, WY, WX><<<grid, 256, 0, st>>>(a);
    else if (KC == 8 && h->precision == OAI_PREC_BF16X6) conv3_igemm_bf16s<3, false, 2, RX, RY, WY, WX><<<grid, 256, 0, st>>>(a);
    else if (KC == 8 && h->precision == OAI_PREC_FP16X3) conv3_igemm_bf16s<2, true, 2, RX, RY, WY, WX><<<grid, 256, 0, st>>>(a);
    else conv3_igemm_f32<kF32Mrep, KC, RX, RY, WY, WX><<<grid, 256, 0, st>>>(a);
    OAI_CHECK_LAUNCH();
    if (h->profile) {
        OAI_CHECK_HIP(hipEventRecord(hm->ev_pool[hm->ev_used + 1], st));
        hm->ev_used += 2;
    }
    return OAI_OK;
}

// One conv layer = the main launch over the part of the output box that 8 x 16 (y, x) tiles cover exactly, plus up
// to two thin remainder strips computed with tile shapes that fit them (trimmed boxes are e.g. 18 x 98 x 98).
// how launch_conv3 covers a box: ny x nx main (8 x 16) tiles, plus an x strip of wr columns and a y strip of hr rows
static void strip_plan(const oai_unet* h, const Box& box, int& ny, int& nx, int& hr, int& wr) {
    const int ry = box.hi[1] - box.lo[1], rx = box.hi[2] - box.lo[2];
    ny = ry / 8; nx = rx / 16;
    hr = ry - 8 * ny; wr = rx - 16 * nx;
    if (h->variant == 2 || ny == 0 || nx == 0) { ny = cdiv(ry, 8); nx = cdiv(rx, 16); hr = wr = 0; }   // no strips
    if (hr > 4) { ++ny; hr = 0; }            // a tall remainder is cheaper as one more row of main tiles
    if (wr > 8) { ++nx; wr = 0; }
}

// ec0 can be computed inside ec1's halo staging (conv3_igemm_sres<..., FIRST>) when ec1 runs as ONE launch of the main shape of the
// default split-resident kernel and the channel counts are the reference's (1 -> 32 -> ...)
static bool first_fusable(const oai_unet* h, const Box& ec1_box) {
    if (!h->sres || !h->fuse_first || h->variant != 0 || h->sres_mrep != 4 || h->sres_ring) return false;
    if (h->L[EC0].cout != 32 || h->L[EC1].c0 != 32) return false;
    int ny, nx, hr, wr;
    strip_plan(h, ec1_box, ny, nx, hr, wr);
    return hr == 0 && wr == 0;
}

// shared encoder pass: where the copy-out of the volume-wide ec1 scatters its voxels (ConvArgs::sc_*)
struct Scatter { const int* boxes; int ntiles, tile0; int g[3], e[3], t[3]; };

// everything of ConvArgs that does not depend on the tile shape
static int fill_conv_args(const oai_unet* h, const Layer& L, const float* s0, const float* s1, float* out, const int dims[3],
                          const int* boxes, float* pool_out, const ConvArgs* head, const TileSource* first, const int* store_boxes,
                          const Scatter* sc, ConvArgs& a) {
    if (head) a = *head;                   // the fused dc0 fields (see ConvArgs); everything else is set below
    if (store_boxes && h->sres && h->opt_dead_stores) { a.store_boxes = store_boxes; a.store_grow = 1; }
    if (first) {                           // ec0 fused into this layer's staging (first_fusable)
        a.first_w = h->L[EC0].plain; a.first_scale = h->L[EC0].scale_f16; a.first_shift = h->L[EC0].shift_f16; a.first_src = *first;   // (sres => fp16x3)
        a.first_census = h->opt_census ? h->census + 16 * EC0 : nullptr;
    }
    if (sc) {
        a.sc_boxes = sc->boxes; a.sc_ntiles = sc->ntiles; a.sc_tile0 = sc->tile0;
        for (int i = 0; i < 3; ++i) { a.sc_g[i] = sc->g[i]; a.sc_e[i] = sc->e[i]; a.sc_t[i] = sc->t[i]; }
    }
    a.boxes = boxes;
    a.pool_out = pool_out;
    a.range_flag = h->range_flag;
#ifdef OAI_DIAG
    a.stamps = diag_stamps();
    { static const int only = diag_env("OAI_STAMP_LAYER", -1); if (only >= 0 && &L != &h->L[only]) a.stamps = nullptr; }   // one layer's budget (EC1 = 1 ...)
#endif
    a.src0 = s0; a.src1 = s1; a.C0 = L.c0; a.C1 = s1 ? L.c1 : 0;
    const bool f16 = h->precision == OAI_PREC_FP16X3;
    a.out = out; a.Cout = L.cout; a.scale = f16 ? L.scale_f16 : L.scale; a.shift = f16 ? L.shift_f16 : L.shift;
    a.census = h->sres && h->opt_census ? h->census + 16 * (int)(&L - h->L) : nullptr;
    a.wpanel = h->precision == OAI_PREC_F32 ? L.panel : L.panel_bf[h->precision == OAI_PREC_BF16X3 ? 0 : h->precision == OAI_PREC_BF16X6 ? 1 : 2];
    a.wpanel16 = h->precision == OAI_PREC_FP16X3 && ((h->opt_m16_layers >> (int)(&L - h->L)) & 1) ? L.panel_m16 : nullptr;
    a.D = dims[0]; a.H = dims[1]; a.W = dims[2];
    a.ncb = (L.cout + 63) / 64;
    a.relu = 1;
    if (h->sres && (size_t)dims[0] * dims[1] * dims[2] * 128 >= (1ull << 32))      // staging plan and copy-out hold 32-bit byte offsets inside one / two chunk planes
        return set_error(OAI_ERR_ARG, "tile level %dx%dx%d too large for the split-resident kernel (>= 2^25 voxels)", dims[0], dims[1], dims[2]);
    return OAI_OK;
}

// one launch of ONE given tile shape over `box` (the faces of the per-tile shell pass, where launch_conv3's main + strips cover does not fit)
// `pool_only` (main shape): the max-pooled tensor is all this launch leaves behind -- fused pool into pool_only, 2 z slices per block (both
// slices of a block are then inside the 2-voxel face), the copy-out of the layer's own output suppressed through an empty store box
template <int RX, int RY, int WY, int WX>
static int launch_conv3_one(const oai_unet* h, const Layer& L, const float* s0, float* out, const int dims[3], const Box& box, int ntiles,
                            hipStream_t st, const int* boxes, float* pool_only = nullptr) {
    ConvArgs a;
    if (int rc = fill_conv_args(h, L, s0, nullptr, out, dims, boxes, pool_only, nullptr, nullptr, nullptr, nullptr, a)) return rc;
    if (pool_only) { a.store_boxes = boxes; a.store_grow = -(1 << 20); }          // consumer box "grown" by -2^20: nothing of the output tensor is written
    return launch_conv3_shape<4, 8, RX, RY, WY, WX>(h, a, box, ntiles, st, pool_only ? 2 : 0);
}

// The specialised 64-cout form takes ALL of a CU's LDS (160 KB): asked once whether a workgroup of it (both block shapes, both tap forms) fits this
// device / driver at all
static bool wino_ws_fits() {
    static const bool fits = [] {
        auto one = [](auto kern) { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 512, 0) == hipSuccess && n >= 1; };
        return one(conv3_wino_sres<1, 8, 4, 1, true>) && one(conv3_wino_sres<1, 16, 2, 1, true>) &&
               one(conv3_wino_sres<1, 8, 4, 1, true, true>) && one(conv3_wino_sres<1, 16, 2, 1, true, true>);
    }();
    return fits;
}
// a 64-cout layer on the 16x16x32 taps (option "winograd" bit 5): its main blocks and x strip run the specialised form's tap-pair variant, its y strip
// the slice-split form's -- ALL its launch shapes or none: a voxel's bits must not depend on which shape covers it
static bool wino_m16_64(const oai_unet* h, const Layer& L, int cout) {
    return cout % 128 != 0 && (h->opt_wino & 32) && !(h->opt_wino & (4 | 8)) && L.panel_wino16 && wino_ws_fits();
}

// The block plan of one persistent launch of conv3_wino_sres<..., PS> (one workgroup): per tile the sub-box of the block grid that meets the tile's own
// box, the number of blocks in front of every tile, the total -- and the XCDs' counters back at zero.  Layout: ConvArgs::ps_plan.
constexpr int kPsMaxTiles = 256;
__global__ void __launch_bounds__(kPsMaxTiles) wino_plan_kernel(int* __restrict__ plan, const int* __restrict__ boxes, int ntiles, int lo0, int lo1, int lo2, int hi0, int hi1, int hi2,
                                                                 int tz, int ty, int tx, int nbz, int nby, int nbx, int ncb) {
    __shared__ int cnt[kPsMaxTiles];
    const int t = threadIdx.x;
    int c = 0;
    if (t < ntiles) {
        const int llo[3] = {lo0, lo1, lo2}, lhi[3] = {hi0, hi1, hi2}, bs[3] = {tz, ty, tx}, nbk[3] = {nbz, nby, nbx};
        int b0[3], nn[3];
        bool any = true;
        for (int i = 0; i < 3; ++i) {
            int l = llo[i], h = lhi[i];
            if (boxes) { l = max(l, boxes[6 * t + i]); h = min(h, boxes[6 * t + 3 + i]); }
            any = any && l < h;
            b0[i] = any ? (l - llo[i]) / bs[i] : 0;                                  // first block whose [o, o + bs) reaches l
            const int b1 = any ? min(nbk[i], (h - llo[i] + bs[i] - 1) / bs[i]) : 0;     // one past the last block that starts below h
            nn[i] = max(0, b1 - b0[i]);
        }
        c = any ? nn[0] * nn[1] * nn[2] * ncb : 0;
        int* sb = plan + 288 + 8 * t;
        sb[0] = b0[0]; sb[1] = nn[0]; sb[2] = b0[1]; sb[3] = nn[1]; sb[4] = b0[2]; sb[5] = nn[2];
    }
    cnt[t] = c;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < ntiles; ++i) { plan[16 + i] = run; run += cnt[i]; }
        plan[16 + ntiles] = run;
        plan[8] = run; plan[9] = ntiles;
        for (int i = 0; i < 8; ++i) plan[i] = 0;
    }
}

// One launch of conv3_wino_sres (unet_wino.h) with blocks of 4 x TY x 2 NP over `box` (box.lo[2] even)
template <int TY, int NP>
static int launch_wino_shape(const oai_unet* h, const Layer& L, ConvArgs a, const Box& box, int ntiles, hipStream_t st) {
    for (int i = 0; i < 3; ++i) { a.lo[i] = box.lo[i]; a.hi[i] = box.hi[i]; }
    if (box.hi[0] <= box.lo[0] || box.hi[1] <= box.lo[1] || box.hi[2] <= box.lo[2]) return OAI_OK;
    const int ng = (a.Cout % 128 == 0 && !((h->opt_wino & 8) && TY != 4)) ? 2 : 1;      // (bit 3, A/B: the specialised 64-cout form for every layer)
    a.wpanel = L.panel_wino;
    a.ncb = a.Cout / (64 * ng);
    a.nbz = cdiv(box.hi[0] - box.lo[0], 4); a.nby = cdiv(box.hi[1] - box.lo[1], TY); a.nbx = cdiv(box.hi[2] - box.lo[2], 2 * NP);
    unsigned grid = (unsigned)((size_t)ntiles * a.nbz * a.nby * a.nbx * a.ncb);
    if (h->xcd_group > 0) {
        a.nblocks = (int)grid; a.xcd_group = h->xcd_group;
        const unsigned q = 8u * (unsigned)h->xcd_group;
        grid = (grid + q - 1) / q * q;
    }
    oai_unet* hm = const_cast<oai_unet*>(h);
    if (h->profile) {
        if (hm->ev_used + 2 > hm->ev_pool.size()) {
            hipEvent_t e0, e1;
            OAI_CHECK_HIP(hipEventCreate(&e0));
            OAI_CHECK_HIP(hipEventCreate(&e1));
            hm->ev_pool.push_back(e0);
            hm->ev_pool.push_back(e1);
        }
        OAI_CHECK_HIP(hipEventRecord(hm->ev_pool[hm->ev_used], st));
    }
    if (ng == 2 && (h->opt_wino & 16) && L.panel_wino16) {              // the taps on v_mfma_f32_16x16x32_f16 (higher clock at the power wall)
        a.wpanel = L.panel_wino16;
        conv3_wino_sres<2, TY, NP, 1, false, true><<<grid, 512, 0, st>>>(a, h->zero_rec);
    } else if (ng == 2) conv3_wino_sres<2, TY, NP><<<grid, 512, 0, st>>>(a, h->zero_rec);
    else if constexpr (TY != 4) {
        if ((h->opt_wino & 4) || !wino_ws_fits()) conv3_wino_sres<1, TY, NP, 2><<<grid, 512, 0, st>>>(a, h->zero_rec);        // (A/B, or no room: the eight waves split the z slices)
        else if (wino_m16_64(h, L, a.Cout)) {                                                          // the multipliers' taps on v_mfma_f32_16x16x32_f16
            a.wpanel = L.panel_wino16;
            conv3_wino_sres<1, TY, NP, 1, true, true><<<grid, 512, 0, st>>>(a, h->zero_rec);
        } else if (h->opt_persist && h->ps_plan && ntiles <= kPsMaxTiles && (a.C0 + 15) / 16 + (a.C1 + 15) / 16 >= 6) {
            // persistent workgroups, one per CU, the staging waves one block ahead (unet_wino.h, PS): the plan first, same stream
            wino_plan_kernel<<<1, kPsMaxTiles, 0, st>>>(h->ps_plan, a.boxes, ntiles, a.lo[0], a.lo[1], a.lo[2], a.hi[0], a.hi[1], a.hi[2], 4, TY, 2 * NP, a.nbz, a.nby, a.nbx, a.ncb);
            OAI_CHECK_LAUNCH();
            a.ps_plan = h->ps_plan;
            const unsigned total = (unsigned)((size_t)ntiles * a.nbz * a.nby * a.nbx * a.ncb);
            const unsigned pgrid = total < (unsigned)h->n_cus ? total : (unsigned)h->n_cus;
            conv3_wino_sres<1, TY, NP, 1, true, false, true><<<pgrid, 512, 0, st>>>(a, h->zero_rec);
        } else conv3_wino_sres<1, TY, NP, 1, true><<<grid, 512, 0, st>>>(a, h->zero_rec);              // one block of 64 couts: four waves multiply, four stage
    } else if (wino_m16_64(h, L, a.Cout)) {                                          // (the y strip's two T buffers would not fit: the eight waves split the z slices --
        a.wpanel = L.panel_wino16;                                                     //  on the same tap pairs as the layer's other shapes: one summation order per layer)
        conv3_wino_sres<1, TY, NP, 2, false, true><<<grid, 512, 0, st>>>(a, h->zero_rec);
    } else conv3_wino_sres<1, TY, NP, 2><<<grid, 512, 0, st>>>(a, h->zero_rec);
    OAI_CHECK_LAUNCH();
    if (h->profile) {
        OAI_CHECK_HIP(hipEventRecord(hm->ev_pool[hm->ev_used + 1], st));
        hm->ev_used += 2;
    }
    return OAI_OK;
}

// A plain layer (no fused ec0 / pool / head / scatter) of the default split-resident configuration through conv3_wino_sres: main blocks of
// 4 x 8 x 8, a y strip of 4 x 4 x 16 blocks for a remainder of <= 4 rows, an x strip of 4 x 16 x 4 blocks for a remainder of <= 4 columns
// .  The box starts at an even x: an output's arithmetic depends on the
// parity of its x only -- not on which launch or block computes it.
static int launch_conv3_wino(const oai_unet* h, const Layer& L, const ConvArgs& a, Box box, int ntiles, hipStream_t st) {
    box.lo[2] &= ~1;
    const int ry = box.hi[1] - box.lo[1], rx = box.hi[2] - box.lo[2];
    int ny = ry / 8, nx = rx / 8, hr = ry - 8 * ny, wr = rx - 8 * nx;
    if (ny == 0 || nx == 0) { ny = cdiv(ry, 8); nx = cdiv(rx, 8); hr = wr = 0; }
    if (hr > 4) { ++ny; hr = 0; }
    if (wr > 4) { ++nx; wr = 0; }
    Box main = box, xs = box, ys = box;
    main.hi[1] = hr ? box.lo[1] + 8 * ny : box.hi[1];
    main.hi[2] = wr ? box.lo[2] + 8 * nx : box.hi[2];
    if (int rc = launch_wino_shape<8, 4>(h, L, a, main, ntiles, st)) return rc;
    if (wr) {                                 // x strip: all y rows, the last wr columns
        xs.lo[2] = main.hi[2];
        if (int rc = launch_wino_shape<16, 2>(h, L, a, xs, ntiles, st)) return rc;
    }
    if (hr) {                                 // y strip: the last hr rows, main columns only
        ys.lo[1] = main.hi[1];
        ys.hi[2] = main.hi[2];
        if (int rc = launch_wino_shape<4, 8>(h, L, a, ys, ntiles, st)) return rc;
    }
    return OAI_OK;
}

// the fused max-pool of conv3_wino_sres pools whole 2 x 2 x 2 windows of a block's own image: the box must be the whole (even-sized) tile
// level, so that every block (any shape) starts at even coordinates and lies inside it
static bool wino_pool_box(const Box& box, const int dims[3]) {
    for (int i = 0; i < 3; ++i)
        if (box.lo[i] != 0 || box.hi[i] != dims[i] || (dims[i] & 1)) return false;
    return dims[0] % 4 == 0 && dims[1] % 8 == 0 && dims[2] % 8 == 0;      // main blocks only: no strips
}

// One launch of conv3_wino_f32 (unet_wino_f32.h) with blocks of 2 x TY x 2 NP over `box` (box.lo[2] even)
template <int TY, int NP>
static int launch_wino_f32_shape(const oai_unet* h, const Layer& L, ConvArgs a, const Box& box, int ntiles, hipStream_t st) {
    for (int i = 0; i < 3; ++i) { a.lo[i] = box.lo[i]; a.hi[i] = box.hi[i]; }
    if (box.hi[0] <= box.lo[0] || box.hi[1] <= box.lo[1] || box.hi[2] <= box.lo[2]) return OAI_OK;
    a.wpanel = L.panel_wino_f32;
    a.ncb = (a.Cout + 63) / 64;
    a.nbz = cdiv(box.hi[0] - box.lo[0], 2); a.nby = cdiv(box.hi[1] - box.lo[1], TY); a.nbx = cdiv(box.hi[2] - box.lo[2], 2 * NP);
    unsigned grid = (unsigned)((size_t)ntiles * a.nbz * a.nby * a.nbx * a.ncb);
    if (h->xcd_group > 0) {                                // XCD-aware dealing of the logical block list (xcd_block_id), as the split-resident kernels
        a.nblocks = (int)grid; a.xcd_group = h->xcd_group;
        const unsigned q = 8u * (unsigned)h->xcd_group;
        grid = (grid + q - 1) / q * q;
    }
    oai_unet* hm = const_cast<oai_unet*>(h);
    if (h->profile) {
        if (hm->ev_used + 2 > hm->ev_pool.size()) {
            hipEvent_t e0, e1;
            OAI_CHECK_HIP(hipEventCreate(&e0));
            OAI_CHECK_HIP(hipEventCreate(&e1));
            hm->ev_pool.push_back(e0);
            hm->ev_pool.push_back(e1);
        }
        OAI_CHECK_HIP(hipEventRecord(hm->ev_pool[hm->ev_used], st));
    }
    conv3_wino_f32<TY, NP><<<grid, 256, 0, st>>>(a, reinterpret_cast<const float*>(h->zero_rec));
    OAI_CHECK_LAUNCH();
    if (h->profile) {
        OAI_CHECK_HIP(hipEventRecord(hm->ev_pool[hm->ev_used + 1], st));
        hm->ev_used += 2;
    }
    return OAI_OK;
}

// The exact-fp32 path's k3 layers through conv3_wino_f32: main blocks of 2 x 8 x 8, a y strip of 2 x 4 x 16 blocks for a remainder of <= 4 rows, an x strip
// of 2 x 16 x 4 blocks for a remainder of <= 4 columns (the cover of launch_conv3_wino).  The box starts at an even x: an output's arithmetic depends on the
// parity of its x only -- not on which launch or block computes it.  A fused MaxPool3d(2) rides in the main blocks (the host passes it for whole-tile boxes only).
static int launch_conv3_wino_f32(const oai_unet* h, const Layer& L, const ConvArgs& a, Box box, int ntiles, hipStream_t st) {
    box.lo[2] &= ~1;
    const int ry = box.hi[1] - box.lo[1], rx = box.hi[2] - box.lo[2];
    int ny = ry / 8, nx = rx / 8, hr = ry - 8 * ny, wr = rx - 8 * nx;
    if (ny == 0 || nx == 0 || a.pool_out) { ny = cdiv(ry, 8); nx = cdiv(rx, 8); hr = wr = 0; }
    if (hr > 4) { ++ny; hr = 0; }
    if (wr > 4) { ++nx; wr = 0; }
    Box main = box, xs = box, ys = box;
    main.hi[1] = hr ? box.lo[1] + 8 * ny : box.hi[1];
    main.hi[2] = wr ? box.lo[2] + 8 * nx : box.hi[2];
    if (int rc = launch_wino_f32_shape<8, 4>(h, L, a, main, ntiles, st)) return rc;
    if (wr) {                                 // x strip: all y rows, the last wr columns
        xs.lo[2] = main.hi[2];
        if (int rc = launch_wino_f32_shape<16, 2>(h, L, a, xs, ntiles, st)) return rc;
    }
    if (hr) {                                 // y strip: the last hr rows, main columns only
        ys.lo[1] = main.hi[1];
        ys.hi[2] = main.hi[2];
        if (int rc = launch_wino_f32_shape<4, 8>(h, L, a, ys, ntiles, st)) return rc;
    }
    return OAI_OK;
}

static int launch_conv3(const oai_unet* h, const Layer& L, const float* s0, const float* s1, float* out,
                        const int dims[3], const Box& box, int ntiles, hipStream_t st, const int* boxes = nullptr,
                        float* pool_out = nullptr, const ConvArgs* head = nullptr, const TileSource* first = nullptr,
                        const int* store_boxes = nullptr, const Scatter* sc = nullptr) {
    ConvArgs a;
    if (int rc = fill_conv_args(h, L, s0, s1, out, dims, boxes, pool_out, head, first, store_boxes, sc, a)) return rc;
    if (h->variant == 1) return launch_conv3_shape<2, 16, 16, 2, 4, 1>(h, a, box, ntiles, st);
    if (h->precision == OAI_PREC_F32 && h->variant == 0 && h->opt_wino_f32 && L.panel_wino_f32 && a.Cout % 4 == 0 && !a.head_w && !a.first_w && !a.sc_boxes &&
        (!a.pool_out || (wino_pool_box(box, dims) && dims[0] % 2 == 0)) && ((h->opt_wino_layers >> (int)(&L - h->L)) & 1))
        return launch_conv3_wino_f32(h, L, a, box, ntiles, st);
    if (h->sres && h->opt_wino && L.panel_wino && h->sres_mrep == 4 && !h->sres_ring && !h->b_lds && !a.first_w && !a.head_w && !a.sc_boxes &&
        (!a.pool_out || (a.Cout % 128 == 0 && a.relu && wino_pool_box(box, dims))) && a.Cout % 64 == 0 && (h->opt_wino & (a.Cout % 128 == 0 ? 1 : 2)) && (a.Cout % 128 == 0 || (a.C0 + 15) / 16 + (a.C1 + 15) / 16 >= 8) && ((h->opt_wino_layers >> (int)(&L - h->L)) & 1) && (size_t)dims[0] * dims[1] * dims[2] < (1u << 24))
        return launch_conv3_wino(h, L, a, box, ntiles, st);
    int ny, nx, hr, wr;
    strip_plan(h, box, ny, nx, hr, wr);
    Box main = box, xs = box, ys = box;
    main.hi[1] = hr ? box.lo[1] + 8 * ny : box.hi[1];
    main.hi[2] = wr ? box.lo[2] + 16 * nx : box.hi[2];
    int rc = launch_conv3_shape<4, 8, 16, 2, 4, 1>(h, a, main, ntiles, st);
    if (rc) return rc;
    if (wr) {                                 // x strip: all y rows, the last wr columns
        xs.lo[2] = main.hi[2];
        if (wr <= 2) rc = launch_conv3_shape<4, 8, 2, 16, 4, 1>(h, a, xs, ntiles, st);
        else if (wr <= 4) rc = launch_conv3_shape<4, 8, 4, 8, 4, 1>(h, a, xs, ntiles, st);
        else rc = launch_conv3_shape<4, 8, 8, 4, 4, 1>(h, a, xs, ntiles, st);
        if (rc) return rc;
    }
    if (hr) {                                 // y strip: the last hr rows, main columns only
        ys.lo[1] = main.hi[1];
        ys.hi[2] = main.hi[2];
        if (hr <= 2) rc = launch_conv3_shape<4, 8, 16, 2, 1, 4>(h, a, ys, ntiles, st);
        else rc = launch_conv3_shape<4, 8, 16, 2, 2, 2>(h, a, ys, ntiles, st);
        if (rc) return rc;
    }
    return OAI_OK;
}

static int launch_up(const oai_unet* h, const Layer& L, const float* src, float* out, const int in_dims[3], const Box& out_need,
                     int ntiles, hipStream_t st, const int* in_boxes = nullptr) {
    UpArgs a;
    a.boxes = in_boxes;
    a.range_flag = h->range_flag;
    a.zero = h->zero_rec;
#ifdef OAI_DIAG
    a.stamps = diag_stamps();
    { static const int only = diag_env("OAI_STAMP_LAYER", -1); if (only >= 0 && &L != &h->L[only]) a.stamps = nullptr; }   // DC9 = 8, DC6 = 11, DC3 = 14
#endif
    const bool split = h->precision == OAI_PREC_FP16X3 && L.panel_bf[2];
    a.src = src; a.Cin = L.cin; a.out = out; a.Cout = L.cout;
    a.wpanel = split ? L.panel_bf[2] : L.panel;
    a.scale = split ? L.scale_f16 : L.scale;
    a.shift = split ? L.shift_f16 : L.shift;
    a.census = h->sres && h->opt_census ? h->census + 16 * (int)(&L - h->L) : nullptr;
    a.D = in_dims[0]; a.H = in_dims[1]; a.W = in_dims[2];
    for (int i = 0; i < 3; ++i) { a.lo[i] = out_need.lo[i] / 2; a.hi[i] = (out_need.hi[i] + 1) / 2; }
    const int nvox = (a.hi[0] - a.lo[0]) * (a.hi[1] - a.lo[1]) * (a.hi[2] - a.lo[2]);
    a.nmb = cdiv(nvox, split && h->sres ? 128 : 64);       // voxels per workgroup: 128 in the split-resident kernel
    a.nnb = cdiv(8 * L.cout, 256);
    a.relu = 1;
    if (split && h->sres) {
        // column blocks per workgroup (round 6): as many as leave >= 16 workgroups per workgroup slot of the chip (256 CUs x 2); narrow test networks
        // (the dword-store path of the kernel) keep one; option "up_nbw": 0 = this rule, n = at most n
        a.nbw = 1;
        if (L.cout % 16 == 0 && h->opt_up_nbw > 1) a.nbw = std::min(h->opt_up_nbw, a.nnb);
        else if (L.cout % 16 == 0 && h->opt_up_nbw == 0) {
            a.nbw = a.nnb;
            while (a.nbw > 1 && (size_t)ntiles * a.nmb * cdiv(a.nnb, a.nbw) < 8192) a.nbw = (a.nbw + 1) / 2;
        }
        unsigned grid = (unsigned)((size_t)ntiles * a.nmb * cdiv(a.nnb, a.nbw));
        if (h->xcd_group > 0) {                               // same dealing as the conv kernel's (launch_conv3_shape)
            a.nblocks = (int)grid; a.xcd_group = h->xcd_group;
            const unsigned q = 8u * (unsigned)h->xcd_group;
            grid = (grid + q - 1) / q * q;
        }
        upconv2_igemm_sres<<<grid, 256, 0, st>>>(a);
    }
    else if (split) upconv2_igemm<true><<<(unsigned)((size_t)ntiles * a.nmb * a.nnb), 256, 0, st>>>(a);
    else upconv2_igemm<false><<<(unsigned)((size_t)ntiles * a.nmb * a.nnb), 256, 0, st>>>(a);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

// MaxPool3d(2) can ride in the conv epilogue when the (full) box is tiled exactly by the main 4 x 8 x 16 shape
static bool pool_fusable(const oai_unet* h, const int dims[3], const Box& box) {
    if (h->variant != 0) return false;
    for (int i = 0; i < 3; ++i) if (box.lo[i] != 0 || box.hi[i] != dims[i]) return false;
    return dims[0] % 4 == 0 && dims[1] % 8 == 0 && dims[2] % 16 == 0;
}

static int launch_pool(const oai_unet* h, const float* in, float* out, const int dims[3], int C, int ntiles, hipStream_t st, int shell_only = 0) {
    if (h->sres) {
        const int nch = (C + 15) / 16;
        const size_t Do = dims[0] / 2, Ho = dims[1] / 2, Wo = dims[2] / 2;
        const size_t faces = Do * Ho * Wo - (Do - 2) * (Ho - 2) * (Wo - 2) - (shell_only == 2 ? 2 * Ho * Wo : 0);
        const size_t total = (size_t)ntiles * (shell_only ? faces : Do * Ho * Wo) * nch * 4;
        size_t blocks = (total + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        maxpool2_sres_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const unsigned char*>(in), reinterpret_cast<unsigned char*>(out),
                                                                dims[0], dims[1], dims[2], nch, total, shell_only);
        OAI_CHECK_LAUNCH();
        return OAI_OK;
    }
    const size_t total4 = (size_t)ntiles * (dims[0] / 2) * (dims[1] / 2) * (dims[2] / 2) * (C / 4);
    size_t blocks = (total4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    maxpool2_kernel<<<(unsigned)blocks, 256, 0, st>>>(in, out, dims[0], dims[1], dims[2], C, total4);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

// Shared encoder pass.  Tiles overlap 2 x in z and 1.33 x in y and x, so ec0 / ec1 evaluate every input voxel ~3.6 x.  An ec1
// output at least 2 voxels from its tile's faces sees no zero padding (ec0 differs from the volume-wide convolution at distance 0, ec1 at
// distances 0 and 1) and -- every launch accumulates in the same k order -- is BIT-IDENTICAL in every tile that contains it.  So ec0 -> ec1
// (+ the fused max-pool) run ONCE per batch over the part of the reflect-padded volume its tiles cover, as one big "tile": the pooled tensor
// goes to SP0 (volume strides), and the copy-out of ec1 itself scatters every voxel straight into the per-tile syn0 tensors of the tiles
// whose dc2 reads it (the skip read [need - 1, need + 1) lies inside the interior).  Per tile only the 2-voxel shell of ec1 is recomputed
// (ec0 on a 3-voxel shell, six thin face launches, the pooled faces) and the interior of the tile's pooled tensor is copied out of SP0.
// ec1 is 22 ms of a 169 ms volume at 0.167 of the MFMA peak; the shared pass takes 8.7 ms.
struct SharedEnc {
    float* SP0;                         // [cout / 16 chunks][PD / 2][PH / 2][PW / 2]: the max-pooled ec1 of the padded volume
    int PD, PH, PW;                     // reflect-padded volume = eff * grid + 2 * overlap
    const float* vol; int D, H, W;      // the volume itself (the pass gathers it with Partition's reflect padding)
    int grid[3], eff[3], overlap[3];
};

// One batch of `n` tiles through the whole network; kept-centre blocks go to blocks_out.
// `table` (device, may be null) = per-tile boxes [layer][table_tiles][6] of the whole call; this batch starts at tile
// `t0` of it.  For up-convs the table row holds the INPUT box (the halved output box).
static int run_batch(oai_unet* h, const TileSource& src, int n, const Box need[18], int out_mode,
                     float* blocks_out, char* ws, const Plan& plan, hipStream_t st,
                     const int* table = nullptr, int table_tiles = 0, int t0 = 0, const SharedEnc* se = nullptr) {
    auto tb = [&](int layer) -> const int* { return table ? table + ((size_t)layer * table_tiles + t0) * 6 : nullptr; };
    const Layer* L = h->L;
    float* buf[B_COUNT];
    for (int i = 0; i < B_COUNT; ++i) buf[i] = reinterpret_cast<float*>(ws + plan.off[i]);
    int d[4][3];
    for (int l = 0; l < 4; ++l) { d[l][0] = src.td >> l; d[l][1] = src.th >> l; d[l][2] = src.tw >> l; }
    const size_t v0 = (size_t)src.td * src.th * src.tw;

    int rc;
#define RUN(x) do { rc = (x); if (rc) return rc; } while (0)
    if (se) {
        // ---- ec0 -> ec1 once over the z range of the padded volume that this batch's tiles cover
        {
            TileSource vs{};                                        // the padded volume as ONE tile: reflect_index(c - overlap) is Partition's padding
            vs.vol = se->vol; vs.D = se->D; vs.H = se->H; vs.W = se->W; vs.td = se->PD; vs.th = se->PH; vs.tw = se->PW;
            vs.ez = se->PD; vs.ey = se->PH; vs.ex = se->PW; vs.oz = se->overlap[0]; vs.oy = se->overlap[1]; vs.ox = se->overlap[2]; vs.gy = 1; vs.gx = 1;
            const int per_row = se->grid[1] * se->grid[2], P[3] = {se->PD, se->PH, se->PW};
            Scatter sc;
            sc.boxes = tb(DC2); sc.ntiles = n; sc.tile0 = src.tile_begin;
            for (int i = 0; i < 3; ++i) { sc.g[i] = se->grid[i]; sc.e[i] = se->eff[i]; }
            sc.t[0] = src.td; sc.t[1] = src.th; sc.t[2] = src.tw;
            // One launch per z slab of `eff` slices, over the y / x BOUNDING BOX of the batch's tiles that touch the slab (a tile of z row i
            // covers the slabs [i, i + td / eff)): a whole volume is one box per slab = the full y / x extent as before, but a rank's share of
            // a tile-sharded volume (19-23 tiles = parts of two or three z rows) no longer pays for the full extent of every row it touches
            // (round 4: profiles/r04_tileshard_projection.md).  Values do not depend on the box: the halo is read from the padded volume.
            const int t_first = src.tile_begin, t_last = src.tile_begin + n - 1;
            const int zspan = (src.td + se->eff[0] - 1) / se->eff[0];           // slabs per tile (2 for 32 / 16)
            const int s_first = t_first / per_row, s_last = t_last / per_row + zspan - 1;
            Box prev{}; bool have_prev = false;
            for (int s = s_first; s <= s_last; ++s) {
                int jlo = 1 << 30, jhi = -1, klo = 1 << 30, khi = -1;
                for (int i = s - zspan + 1; i <= s; ++i) {                      // z rows whose tiles cover slab s
                    if (i < 0 || i >= se->grid[0]) continue;
                    const int a0 = std::max(t_first, i * per_row), a1 = std::min(t_last, (i + 1) * per_row - 1);
                    if (a0 > a1) continue;
                    const int j0 = (a0 - i * per_row) / se->grid[2], j1 = (a1 - i * per_row) / se->grid[2];
                    jlo = std::min(jlo, j0); jhi = std::max(jhi, j1);
                    if (j0 == j1) { klo = std::min(klo, (a0 - i * per_row) % se->grid[2]); khi = std::max(khi, (a1 - i * per_row) % se->grid[2]); }
                    else { klo = 0; khi = se->grid[2] - 1; }
                }
                if (jhi < 0) continue;
                Box bx;
                bx.lo[0] = se->eff[0] * s; bx.hi[0] = std::min(P[0], se->eff[0] * (s + 1));
                if (s == s_last) bx.hi[0] = std::min(P[0], se->eff[0] * (t_last / per_row) + src.td);       // (the last slab takes the remainder of the tile height)
                bx.lo[1] = se->eff[1] * jlo; bx.hi[1] = std::min(P[1], se->eff[1] * jhi + src.th);
                bx.lo[2] = se->eff[2] * klo; bx.hi[2] = std::min(P[2], se->eff[2] * khi + src.tw);
                if (have_prev && prev.lo[1] == bx.lo[1] && prev.hi[1] == bx.hi[1] && prev.lo[2] == bx.lo[2] && prev.hi[2] == bx.hi[2] && prev.hi[0] == bx.lo[0]) {
                    prev.hi[0] = bx.hi[0];                                      // same footprint as the slab below: one launch for both
                    continue;
                }
                if (have_prev) RUN(launch_conv3(h, L[EC1], nullptr, nullptr, buf[B_SYN0], P, prev, 1, st, nullptr, se->SP0, nullptr, &vs, nullptr, &sc));
                prev = bx; have_prev = true;
            }
            if (have_prev) RUN(launch_conv3(h, L[EC1], nullptr, nullptr, buf[B_SYN0], P, prev, 1, st, nullptr, se->SP0, nullptr, &vs, nullptr, &sc));
        }
        // ---- the interior of this batch's pooled tensors is a copy, their faces are computed
        const int nch1 = (L[EC1].cout + 15) / 16;
        {
            const size_t total = (size_t)n * (d[1][0] * d[1][1] * d[1][2]) * nch1 * 4;
            size_t blocks = (total + 255) / 256;
            if (blocks > 256 * 64) blocks = 256 * 64;
            pooled_gather_kernel<<<(unsigned)blocks, 256, 0, st>>>(reinterpret_cast<const unsigned char*>(se->SP0), reinterpret_cast<unsigned char*>(buf[B_P0]), src,
                                                                   d[1][0], d[1][1], d[1][2], se->PD / 2, se->PH / 2, se->PW / 2, nch1, total);
            OAI_CHECK_LAUNCH();
        }
        {   // ec0 where ec1's shell reads it: closer than 3 voxels to a face (voxel pairs along x, enumerated slab by slab)
            const size_t pairs = (size_t)6 * d[0][1] * (d[0][2] / 2) + (size_t)(d[0][0] - 6) * 6 * (d[0][2] / 2) + (size_t)(d[0][0] - 6) * (d[0][1] - 6) * 4;
            dim3 grid(std::min<unsigned>(cdiv(pairs, 256), (unsigned)h->opt_first_blocks), n);
            conv3_first_sres_kernel<32><<<grid, 256, 0, st>>>(src, L[EC0].plain, L[EC0].scale_f16, L[EC0].shift_f16, reinterpret_cast<unsigned char*>(buf[B_E0]), 1,
                                                              h->range_flag, h->opt_census ? h->census + 16 * EC0 : nullptr, 3);
            OAI_CHECK_LAUNCH();
        }
        const int td = d[0][0], th = d[0][1], tw = d[0][2];
        const Box zlo = {{0, 0, 0}, {2, th, tw}}, zhi = {{td - 2, 0, 0}, {td, th, tw}};                      // the six faces, 2 voxels thick, disjoint
        const Box ylo = {{2, 0, 0}, {td - 2, 2, tw}}, yhi = {{2, th - 2, 0}, {td - 2, th, tw}};
        const Box xlo = {{2, 2, 0}, {td - 2, th - 2, 2}}, xhi = {{2, 2, tw - 2}, {td - 2, th - 2, tw}};
        // z faces: main shape with 2 slices per block; nobody reads their part of the per-tile syn0, only its max-pool (fused into the launch)
        const bool zpool = tb(EC1) != nullptr && h->opt_dead_stores;
        RUN((launch_conv3_one<16, 2, 4, 1>(h, L[EC1], buf[B_E0], buf[B_SYN0], d[0], zlo, n, st, tb(EC1), zpool ? buf[B_P0] : nullptr)));
        RUN((launch_conv3_one<16, 2, 4, 1>(h, L[EC1], buf[B_E0], buf[B_SYN0], d[0], zhi, n, st, tb(EC1), zpool ? buf[B_P0] : nullptr)));
        RUN((launch_conv3_one<16, 2, 1, 4>(h, L[EC1], buf[B_E0], buf[B_SYN0], d[0], ylo, n, st, tb(EC1))));   // y faces: blocks of 4 x 2 x 64
        RUN((launch_conv3_one<16, 2, 1, 4>(h, L[EC1], buf[B_E0], buf[B_SYN0], d[0], yhi, n, st, tb(EC1))));
        RUN((launch_conv3_one<2, 16, 4, 1>(h, L[EC1], buf[B_E0], buf[B_SYN0], d[0], xlo, n, st, tb(EC1))));   // x faces: blocks of 4 x 64 x 2
        RUN((launch_conv3_one<2, 16, 4, 1>(h, L[EC1], buf[B_E0], buf[B_SYN0], d[0], xhi, n, st, tb(EC1))));
        RUN(launch_pool(h, buf[B_SYN0], buf[B_P0], d[0], L[EC1].cout, n, st, zpool ? 2 : 1));               // the (other) pooled faces: windows inside the shell
    } else {
    const bool fuse_first = first_fusable(h, need[EC1]);
    const TileSource* fsrc = fuse_first ? &src : nullptr;
    if (!fuse_first) {   // ec0 (+ gather)
        const dim3 grid_all(cdiv(v0 / 2, 256), n);                                          // one workgroup per 256 voxel pairs (the fp32 kernel)
        const dim3 grid(std::min<unsigned>(cdiv(v0 / 2, 256), (unsigned)h->opt_first_blocks), n);     // the split-resident kernel walks the pairs with a grid stride
        const int c = L[EC0].cout;
        unsigned char* e0s = reinterpret_cast<unsigned char*>(buf[B_E0]);
        const bool f16 = h->precision == OAI_PREC_FP16X3;
        const float* sc0 = f16 ? L[EC0].scale_f16 : L[EC0].scale;
        const float* sh0 = f16 ? L[EC0].shift_f16 : L[EC0].shift;
        unsigned* cen0 = h->opt_census ? h->census + 16 * EC0 : nullptr;
        if (h->sres && c == 32) conv3_first_sres_kernel<32><<<grid, 256, 0, st>>>(src, L[EC0].plain, sc0, sh0, e0s, 1, h->range_flag, cen0, 0);
        else if (h->sres && c == 16) conv3_first_sres_kernel<16><<<grid, 256, 0, st>>>(src, L[EC0].plain, sc0, sh0, e0s, 1, h->range_flag, cen0, 0);
        else if (h->sres && c == 8) conv3_first_sres_kernel<8><<<grid, 256, 0, st>>>(src, L[EC0].plain, sc0, sh0, e0s, 1, h->range_flag, cen0, 0);
        else if (c == 32) conv3_first_kernel<32><<<grid_all, 256, 0, st>>>(src, L[EC0].plain, sc0, sh0, buf[B_E0], 1);
        else if (c == 16) conv3_first_kernel<16><<<grid_all, 256, 0, st>>>(src, L[EC0].plain, sc0, sh0, buf[B_E0], 1);
        else if (c == 8) conv3_first_kernel<8><<<grid_all, 256, 0, st>>>(src, L[EC0].plain, sc0, sh0, buf[B_E0], 1);
        else return set_error(OAI_ERR_ARG, "ec0 cout %d unsupported (8, 16 or 32)", c);
        OAI_CHECK_LAUNCH();
    }
    if (pool_fusable(h, d[0], need[EC1])) {
        RUN(launch_conv3(h, L[EC1], buf[B_E0], nullptr, buf[B_SYN0], d[0], need[EC1], n, st, tb(EC1), buf[B_P0], nullptr, fsrc, tb(DC2)));
    } else {
        RUN(launch_conv3(h, L[EC1], buf[B_E0], nullptr, buf[B_SYN0], d[0], need[EC1], n, st, tb(EC1), nullptr, nullptr, fsrc));
        RUN(launch_pool(h, buf[B_SYN0], buf[B_P0], d[0], L[EC1].cout, n, st));
    }
    }
    RUN(launch_conv3(h, L[EC2], buf[B_P0], nullptr, buf[B_E2], d[1], need[EC2], n, st, tb(EC2)));
    if (pool_fusable(h, d[1], need[EC3])) {
        RUN(launch_conv3(h, L[EC3], buf[B_E2], nullptr, buf[B_SYN1], d[1], need[EC3], n, st, tb(EC3), buf[B_P1], nullptr, nullptr, tb(DC5)));
    } else {
        RUN(launch_conv3(h, L[EC3], buf[B_E2], nullptr, buf[B_SYN1], d[1], need[EC3], n, st, tb(EC3)));
        RUN(launch_pool(h, buf[B_SYN1], buf[B_P1], d[1], L[EC3].cout, n, st));
    }
    RUN(launch_conv3(h, L[EC4], buf[B_P1], nullptr, buf[B_E4], d[2], need[EC4], n, st, tb(EC4)));
    if (pool_fusable(h, d[2], need[EC5])) {
        RUN(launch_conv3(h, L[EC5], buf[B_E4], nullptr, buf[B_SYN2], d[2], need[EC5], n, st, tb(EC5), buf[B_P2], nullptr, nullptr, tb(DC8)));
    } else {
        RUN(launch_conv3(h, L[EC5], buf[B_E4], nullptr, buf[B_SYN2], d[2], need[EC5], n, st, tb(EC5)));
        RUN(launch_pool(h, buf[B_SYN2], buf[B_P2], d[2], L[EC5].cout, n, st));
    }
    RUN(launch_conv3(h, L[EC6], buf[B_P2], nullptr, buf[B_E6], d[3], need[EC6], n, st, tb(EC6)));
    RUN(launch_conv3(h, L[EC7], buf[B_E6], nullptr, buf[B_E7], d[3], need[EC7], n, st, tb(EC7)));
    RUN(launch_up(h, L[DC9], buf[B_E7], buf[B_U9], d[3], need[DC9], n, st, tb(DC9)));
    RUN(launch_conv3(h, L[DC8], buf[B_U9], buf[B_SYN2], buf[B_D8], d[2], need[DC8], n, st, tb(DC8)));   // cat(up, skip) :127
    RUN(launch_conv3(h, L[DC7], buf[B_D8], nullptr, buf[B_D7], d[2], need[DC7], n, st, tb(DC7)));
    RUN(launch_up(h, L[DC6], buf[B_D7], buf[B_U6], d[2], need[DC6], n, st, tb(DC6)));
    RUN(launch_conv3(h, L[DC5], buf[B_U6], buf[B_SYN1], buf[B_D5], d[1], need[DC5], n, st, tb(DC5)));   // :134
    RUN(launch_conv3(h, L[DC4], buf[B_D5], nullptr, buf[B_D4], d[1], need[DC4], n, st, tb(DC4)));
    RUN(launch_up(h, L[DC3], buf[B_D4], buf[B_U3], d[1], need[DC3], n, st, tb(DC3)));
    RUN(launch_conv3(h, L[DC2], buf[B_U3], buf[B_SYN0], buf[B_D2], d[0], need[DC2], n, st, tb(DC2)));   // :141
    // dc0 + sigmoid/threshold + centre crop: blocks are laid out over the full kept centre.  In the split-resident path it rides
    // in dc1's epilogue (dc1's 64 output channels sit in one workgroup): no dc1 output round trip through HBM, no head launch.
    const int kz = src.vol ? src.oz : 0, ky = src.vol ? src.oy : 0, kx = src.vol ? src.ox : 0;
    const int ez = src.vol ? src.ez : src.td, ey = src.vol ? src.ey : src.th, ex = src.vol ? src.ex : src.tw;
    static const bool no_fuse = diag_env("OAI_NO_HEAD_FUSE", 0) != 0;
    const bool fuse_head = h->sres && !no_fuse && L[DC1].cout <= 64 && L[DC0].cin == L[DC1].cout && h->n_classes <= 4 &&
                           need[DC1].lo[0] <= need[DC0].lo[0] && need[DC1].hi[0] >= need[DC0].hi[0];
    if (fuse_head) {
        ConvArgs ha;
        ha.head_w = L[DC0].plain_f16; ha.head_b = L[DC0].shift;       // (fuse_head => sres => fp16x3: head weights carry 2^-e(dc1))
        ha.head_boxes = tb(DC0); ha.head_out = blocks_out;
        ha.head_ncls = h->n_classes; ha.head_mode = out_mode;
        ha.head_k[0] = kz; ha.head_k[1] = ky; ha.head_k[2] = kx; ha.head_e[0] = ez; ha.head_e[1] = ey; ha.head_e[2] = ex;
        RUN(launch_conv3(h, L[DC1], buf[B_D2], nullptr, buf[B_D1], d[0], need[DC0], n, st, tb(DC0), nullptr, &ha));
        return OAI_OK;
    }
    RUN(launch_conv3(h, L[DC1], buf[B_D2], nullptr, buf[B_D1], d[0], need[DC1], n, st, tb(DC1)));
#undef RUN
    {
        const Box& k = need[DC0];
        const float* hw = h->precision == OAI_PREC_FP16X3 ? L[DC0].plain_f16 : L[DC0].plain;
        const int bz = k.hi[0] - k.lo[0], by = k.hi[1] - k.lo[1], bx = k.hi[2] - k.lo[2];
        dim3 grid(cdiv((size_t)bz * by * bx, 256), n);
        if (h->sres)
            head_sres_kernel<<<grid, 256, 0, st>>>(reinterpret_cast<const unsigned char*>(buf[B_D1]), L[DC0].cin, d[0][0], d[0][1], d[0][2],
                                                   k.lo[0], k.lo[1], k.lo[2], bz, by, bx, kz, ky, kx, ez, ey, ex, hw,
                                                   L[DC0].shift, h->n_classes, out_mode, blocks_out, tb(DC0));
        else
        head_kernel<<<grid, 256, 0, st>>>(buf[B_D1], L[DC0].cin, d[0][0], d[0][1], d[0][2], k.lo[0], k.lo[1], k.lo[2],
                                          bz, by, bx, kz, ky, kx, ez, ey, ex, hw, L[DC0].shift, h->n_classes,
                                          out_mode, blocks_out, tb(DC0));
        OAI_CHECK_LAUNCH();
    }
    return OAI_OK;
}

static double layer_flops(const oai_unet* h, int k, const Box& b) {
    const double vox = (double)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
    const int taps = (kKind[k] == 0 || kKind[k] == 1) ? 27 : 1;
    return 2.0 * vox * taps * h->L[k].cin * h->L[k].cout;
}

// Lower edge of the calibrated range: a layer whose largest stored activation is below kLowRange (and not zero) is at least 128 x smaller
// than at calibration time (or was never calibrated): its low split terms are subnormal and the result is no longer fp32 grade.
constexpr float kLowRange = 8.0f;
constexpr int kTargetExp = 10;            // calibration puts a layer's maximum in [2^10, 2^11)

// dst[0] = range flag of the work queued so far: bit 0 = an activation beyond fp16's range (set by the kernels), bit 1 = a layer's
// maximum below kLowRange (from the census).  One wave; `reset` clears the flag and the census for the next volume.
__global__ void range_eval_kernel(int* __restrict__ flag, unsigned* __restrict__ census, int* __restrict__ dst, int reset, float low) {
    const int t = threadIdx.x;
    unsigned m = 0;
    if (t < 18)
        for (int i = 0; i < 16; ++i) m = max(m, census[t * 16 + i]);
    const bool lowhit = t < 17 && m != 0 && __uint_as_float(m) < low;
    const unsigned long long any = __ballot(lowhit);
    if (t == 0) {
        dst[0] = flag[0] | (any ? 2 : 0);
        if (reset) flag[0] = 0;
    }
    if (reset && t < 18)
        for (int i = 0; i < 16; ++i) census[t * 16 + i] = 0;
}

// the raw form of range_eval_kernel: [0] = overflow bit, [1 + k] = layer k's maximum (float bits); clears flag and census
__global__ void range_state_kernel(int* __restrict__ flag, unsigned* __restrict__ census, int* __restrict__ dst) {
    const int t = threadIdx.x;
    if (t < 18) {
        unsigned m = 0;
        for (int i = 0; i < 16; ++i) { m = max(m, census[t * 16 + i]); census[t * 16 + i] = 0; }
        dst[1 + t] = (int)m;
    }
    if (t == 0) { dst[0] = flag[0] & 1; flag[0] = 0; }
}

__global__ void range_flag_from_state_kernel(const int* __restrict__ state, int* __restrict__ dst, float low) {
    const int t = threadIdx.x;
    const unsigned m = t < 17 ? (unsigned)state[1 + t] : 0u;
    const bool lowhit = t < 17 && m != 0 && __uint_as_float(m) < low;
    const unsigned long long any = __ballot(lowhit);
    if (t == 0) dst[0] = (state[0] & 1) | (any ? 2 : 0);
}

}  // namespace oai

using namespace oai;

static int stitch_impl(const float* blocks, int ncls, int D, int H, int W, const int tile[3], const int overlap[3],
                       const int crop[3], float* maps, void* stream, const StitchRanges& rg);

extern "C" {

int oai_unet_create(const oai_layer_params layers[OAI_UNET_NUM_LAYERS], float bn_eps, oai_unet** out) {
    OAI_CHECK_ARG(layers && out, "oai_unet_create: null pointer");
    for (int k = 0; k < 18; ++k) {
        OAI_CHECK_ARG(layers[k].weight_host, "oai_unet_create: layer %d has no weight (strict load)", k);
        OAI_CHECK_ARG(layers[k].kind == kKind[k], "oai_unet_create: layer %d kind %d, expected %d", k, layers[k].kind, kKind[k]);
        OAI_CHECK_ARG(layers[k].cin > 0 && layers[k].cout > 0, "oai_unet_create: layer %d bad channels", k);
    }
    // channel wiring of UNet.__init__ (networks.py:43-66)
    const int chain[][2] = {{EC0, EC1}, {EC1, EC2}, {EC2, EC3}, {EC3, EC4}, {EC4, EC5}, {EC5, EC6}, {EC6, EC7}, {EC7, DC9},
                            {DC8, DC7}, {DC7, DC6}, {DC5, DC4}, {DC4, DC3}, {DC2, DC1}, {DC1, DC0}};
    for (auto& c : chain)
        OAI_CHECK_ARG(layers[c[0]].cout == layers[c[1]].cin, "oai_unet_create: layer %d cout != layer %d cin", c[0], c[1]);
    OAI_CHECK_ARG(layers[DC8].cin == layers[DC9].cout + layers[EC5].cout, "oai_unet_create: dc8 cin != dc9 + ec5");
    OAI_CHECK_ARG(layers[DC5].cin == layers[DC6].cout + layers[EC3].cout, "oai_unet_create: dc5 cin != dc6 + ec3");
    OAI_CHECK_ARG(layers[DC2].cin == layers[DC3].cout + layers[EC1].cout, "oai_unet_create: dc2 cin != dc3 + ec1");
    OAI_CHECK_ARG(layers[EC0].cin == 1, "oai_unet_create: only in_channels == 1 is supported (the reference's config)");
    OAI_CHECK_ARG(layers[DC0].cout <= 4, "oai_unet_create: n_classes must be <= 4");
    for (int k = 0; k < 18; ++k) {
        OAI_CHECK_ARG(k == EC0 || layers[k].cin % 8 == 0, "oai_unet_create: layer %d cin must be a multiple of 8", k);
        OAI_CHECK_ARG(k == DC0 || layers[k].cout % 8 == 0, "oai_unet_create: layer %d cout must be a multiple of 8", k);
    }

    oai_unet* h = new oai_unet();
    {
        void* f = nullptr;
        if (hipMalloc(&f, 256) != hipSuccess) { delete h; return set_error(OAI_ERR_HIP, "oai_unet_create: hipMalloc failed"); }
        h->allocs.push_back(f);
        h->range_flag = reinterpret_cast<int*>(f);
        (void)hipMemset(f, 0, 256);
        void* z = nullptr;
        if (hipMalloc(&z, 256) != hipSuccess) { delete h; return set_error(OAI_ERR_HIP, "oai_unet_create: hipMalloc failed"); }
        h->allocs.push_back(z);
        h->zero_rec = reinterpret_cast<unsigned char*>(z);
        (void)hipMemset(z, 0, 256);
        void* pp = nullptr;
        if (hipMalloc(&pp, (288 + 8 * kPsMaxTiles) * sizeof(int)) != hipSuccess) { delete h; return set_error(OAI_ERR_HIP, "oai_unet_create: hipMalloc failed"); }
        h->allocs.push_back(pp);
        h->ps_plan = reinterpret_cast<int*>(pp);
        (void)hipMemset(pp, 0, (288 + 8 * kPsMaxTiles) * sizeof(int));
        {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) h->n_cus = prop.multiProcessorCount;
        }
        void* c = nullptr;
        if (hipMalloc(&c, 18 * 16 * sizeof(unsigned) + 256) != hipSuccess) { delete h; return set_error(OAI_ERR_HIP, "oai_unet_create: hipMalloc failed"); }
        h->allocs.push_back(c);
        h->census = reinterpret_cast<unsigned*>(c);
        h->eval_out = reinterpret_cast<int*>(reinterpret_cast<char*>(c) + 18 * 16 * sizeof(unsigned));
        (void)hipMemset(c, 0, 18 * 16 * sizeof(unsigned) + 256);
    }
    h->variant = diag_env("OAI_CONV_VARIANT", 0);
    if (h->variant < 0 || h->variant > 2) h->variant = 0;
    h->n_classes = layers[DC0].cout;
    const int KC = conv_kc(h->variant);
    int rc = OAI_OK;
    for (int k = 0; k < 18 && rc == OAI_OK; ++k) {
        const oai_layer_params& p = layers[k];
        Layer& L = h->L[k];
        L.kind = p.kind; L.cin = p.cin; L.cout = p.cout; L.c0 = p.cin; L.c1 = 0;
        if (k == DC8) { L.c0 = layers[DC9].cout; L.c1 = layers[EC5].cout; }
        if (k == DC5) { L.c0 = layers[DC6].cout; L.c1 = layers[EC3].cout; }
        if (k == DC2) { L.c0 = layers[DC3].cout; L.c1 = layers[EC1].cout; }
        // epilogue affine: eval-mode BatchNorm3d folded with the conv bias (per channel, weights untouched)
        std::vector<float> sc(p.cout, 1.0f), sh(p.cout, 0.0f);
        for (int c = 0; c < p.cout; ++c) {
            const float b = p.bias_host ? p.bias_host[c] : 0.0f;
            if (p.bn_gamma_host) {
                const float s = p.bn_gamma_host[c] / sqrtf(p.bn_var_host[c] + bn_eps);
                sc[c] = s;
                sh[c] = (b - p.bn_mean_host[c]) * s + p.bn_beta_host[c];
            } else sh[c] = b;
        }
        L.scale_host = sc; L.shift_host = sh;
        if ((rc = upload(h, sc, &L.scale))) break;
        if ((rc = upload(h, sh, &L.shift))) break;
        if (k == EC0) {
            std::vector<float> wk = canonical_k3(p);          // [27][1][cout]
            rc = upload(h, wk, &L.plain);
        } else if (p.kind == 0 || p.kind == 1) {
            L.wk_host = canonical_k3(p);
            rc = upload(h, pack_conv3_panel(L.wk_host, L.c0, L.c1, p.cout, KC), &L.panel);
            if (!rc && KC == 8) rc = upload(h, pack_wino_f32_panel(L.wk_host, L.c0, L.c1, p.cout), &L.panel_wino_f32);
        } else if (p.kind == 2) {
            L.wk_host.assign(p.weight_host, p.weight_host + (size_t)p.cin * p.cout * 8);     // [ci][co][2][2][2], for re-packing
            rc = upload(h, pack_up_panel(p), &L.panel);
        } else {
            std::vector<float> w(p.weight_host, p.weight_host + (size_t)p.cout * p.cin);
            L.plain_host = w;
            rc = upload(h, w, &L.plain);
        }
    }
    if (rc != OAI_OK) { oai_unet_destroy(h); return rc; }
    *out = h;
    return OAI_OK;
}

int oai_unet_set_precision(oai_unet* h, int mode) {
    OAI_CHECK_ARG(h, "oai_unet_set_precision: null handle");
    OAI_CHECK_ARG(mode >= OAI_PREC_F32 && mode <= OAI_PREC_FP16X3, "oai_unet_set_precision: unknown mode %d", mode);
    OAI_CHECK_ARG(mode == OAI_PREC_F32 || h->variant == 0, "oai_unet_set_precision: split modes need OAI_CONV_VARIANT=0");
    if (mode != OAI_PREC_F32) {
        const int slot = mode == OAI_PREC_BF16X3 ? 0 : mode == OAI_PREC_BF16X6 ? 1 : 2, NS = slot == 1 ? 3 : 2;
        for (int k = 1; k < 17; ++k) {
            Layer& L = h->L[k];
            if (slot == 2) {                                           // fp16x3: k3 convs and k2s2 up-convs (pack_fp16_layer)
                if (!L.panel_bf[2])
                    if (int rc = pack_fp16_layer(h, k)) return rc;
                continue;
            }
            if ((L.kind != 0 && L.kind != 1) || L.panel_bf[slot]) continue;
            if (int rc = upload(h, pack_conv3_panel_bf(L.wk_host, L.c0, L.c1, L.cout, NS), &L.panel_bf[slot])) return rc;
        }
        if (slot == 2 && !h->L[EC0].scale_f16)
            if (int rc = refresh_fp16_affine(h)) return rc;
    }
    h->precision = mode;
    h->sres = mode == OAI_PREC_FP16X3 && h->opt_sres;
    return OAI_OK;
}

// Result-preserving tuning options of the fp16x3 path (every combination is parity-tested: tests/test_unet_gpu.py).  An explicit
// call on the handle -- the production library does not read the environment.
int oai_unet_set_option(oai_unet* h, const char* name, int value) {
    OAI_CHECK_ARG(h && name, "oai_unet_set_option: null pointer");
    if (!strcmp(name, "sres")) {                       // 1: activations resident as fp16 term pairs (default); 0: fp32-resident split kernels
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: sres must be 0 or 1");
        h->opt_sres = value != 0;
        h->sres = h->precision == OAI_PREC_FP16X3 && h->opt_sres;
    } else if (!strcmp(name, "sres_mrep")) {           // z slices per block of the split-resident conv kernel
        OAI_CHECK_ARG(value == 2 || value == 4, "oai_unet_set_option: sres_mrep must be 2 or 4");
        OAI_CHECK_ARG(!(h->sres_ring && value != 2), "oai_unet_set_option: the plane ring needs sres_mrep 2");
        h->sres_mrep = value;
    } else if (!strcmp(name, "sres_ring")) {           // six-slot z-plane ring (implies sres_mrep 2)
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: sres_ring must be 0 or 1");
        h->sres_ring = value != 0;
        if (h->sres_ring) h->sres_mrep = 2;
    } else if (!strcmp(name, "b_lds")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: b_lds must be 0 or 1");
        h->b_lds = value;
    } else if (!strcmp(name, "shared_enc")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: shared_enc must be 0 or 1");
        h->opt_shared = value;
    } else if (!strcmp(name, "wide")) {
        OAI_CHECK_ARG(value >= 0 && value <= 2, "oai_unet_set_option: wide must be 0, 1 or 2 (2 = also for launches of fewer than 1024 workgroups)");
        h->opt_wide = value;
    } else if (!strcmp(name, "winograd")) {            // bit 0: layers with Cout % 128 == 0 (two cout groups per workgroup), bit 1: one block of 64 couts and >= 8 chunks
                                                       // (specialised waves); A/B only: bit 2 = the slice-split form instead, bit 3 = the specialised form for every layer;
                                                       // bit 4 / bit 5: the two-group / the specialised form's taps on v_mfma_f32_16x16x32_f16 (tap pairs; another summation order)
        OAI_CHECK_ARG(value >= 0 && value <= 63, "oai_unet_set_option: winograd must be in [0, 63]");
        if (value && !h->opt_wino && h->L[EC0].scale_f16) {
            OAI_CHECK_HIP(hipDeviceSynchronize());
            for (int k = 1; k < 17; ++k)
                if (h->L[k].panel_bf[2] && !h->L[k].panel_wino)
                    if (int rc = pack_wino_layer(h, k)) return rc;
        }
        h->opt_wino = value;
    } else if (!strcmp(name, "winograd_layers")) {
        OAI_CHECK_ARG(value >= 0 && value <= 0x3FFFF, "oai_unet_set_option: winograd_layers is a mask over the 18 layers");
        h->opt_wino_layers = value;
    } else if (!strcmp(name, "m16")) {                 // the direct kernel's taps on v_mfma_f32_16x16x32_f16 tap pairs (another summation order; default 1)
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: m16 must be 0 or 1");
        h->opt_m16 = value;
    } else if (!strcmp(name, "persistent")) {          // persistent workgroups with the staging one block ahead (bit 0: the specialised 64-cout Winograd form); bit-identical maps
        OAI_CHECK_ARG(value >= 0 && value <= 1, "oai_unet_set_option: persistent is a mask (bit 0)");
        h->opt_persist = value;
    } else if (!strcmp(name, "m16_layers")) {
        OAI_CHECK_ARG(value >= 0 && value <= 0x3FFFF, "oai_unet_set_option: m16_layers is a mask over the 18 layers");
        h->opt_m16_layers = value;
    } else if (!strcmp(name, "dead_stores")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: dead_stores must be 0 or 1");
        h->opt_dead_stores = value;
    } else if (!strcmp(name, "census")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: census must be 0 or 1");
        // without the census there is no LOW bit in the range flag and nothing to calibrate from: allowed on a calibrated handle only
        // (A/B timing of the bookkeeping), so that the subnormal low-term loss cannot come back silently
        OAI_CHECK_ARG(value == 1 || h->calibrated, "oai_unet_set_option: census 0 needs a calibrated handle (activation exponents set)");
        h->opt_census = value;
    } else if (!strcmp(name, "winograd_f32")) {        // NOT bit-preserving: the exact-fp32 path's plain k3 layers in Winograd F(2,3) form along x
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: winograd_f32 must be 0 or 1");
        h->opt_wino_f32 = value;
    } else if (!strcmp(name, "first_blocks")) {        // bit-preserving: workgroups per tile of the ec0 kernel (grid stride over the tile's voxel pairs)
        OAI_CHECK_ARG(value >= 1 && value <= 4096, "oai_unet_set_option: first_blocks must be in [1, 4096]");
        h->opt_first_blocks = value;
    } else if (!strcmp(name, "up_nbw")) {              // bit-preserving: column blocks a workgroup of the k2s2 up-conv walks (0 = automatic)
        OAI_CHECK_ARG(value >= 0 && value <= 64, "oai_unet_set_option: up_nbw must be in [0, 64]");
        h->opt_up_nbw = value;
    } else if (!strcmp(name, "calibrated")) {          // only clearing: setting goes through oai_unet_set_act_exponents / a settled oai_unet_calibrate_step
        OAI_CHECK_ARG(value == 0, "oai_unet_set_option: calibrated can only be cleared (0)");
        OAI_CHECK_ARG(h->opt_census == 1, "oai_unet_set_option: an uncalibrated handle needs the census (option census is 0)");
        h->calibrated = false;
    } else if (!strcmp(name, "fuse_first")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_unet_set_option: fuse_first must be 0 or 1");
        h->fuse_first = value;
    } else if (!strcmp(name, "xcd_group")) {           // logical blocks per XCD deal; 0 = plain launch order
        OAI_CHECK_ARG(value >= 0 && value <= 4096, "oai_unet_set_option: xcd_group must be in [0, 4096]");
        h->xcd_group = value;
    } else {
        return set_error(OAI_ERR_ARG, "oai_unet_set_option: unknown option '%s'", name);
    }
    return OAI_OK;
}

int oai_unet_range_flag(oai_unet* h, int reset, int* out, void* stream) {
    OAI_CHECK_ARG(h && out, "oai_unet_range_flag: null pointer");
    // ordered on the caller's stream (the one the segment calls were queued on): torch's side streams are non-blocking, so
    // the null stream would not wait for them and a flag could be read -- and reset -- before the kernels that set it ran
    hipStream_t st = (hipStream_t)stream;
    range_eval_kernel<<<1, 64, 0, st>>>(h->range_flag, h->census, h->eval_out, reset, kLowRange);
    OAI_CHECK_LAUNCH();
    OAI_CHECK_HIP(hipMemcpyAsync(out, h->eval_out, sizeof(int), hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipStreamSynchronize(st));
    return OAI_OK;
}

int oai_unet_range_flag_snapshot(oai_unet* h, int* dst_dev, void* stream) {
    OAI_CHECK_ARG(h && dst_dev, "oai_unet_range_flag_snapshot: null pointer");
    hipStream_t st = (hipStream_t)stream;
    range_eval_kernel<<<1, 64, 0, st>>>(h->range_flag, h->census, dst_dev, 1, kLowRange);   // a kernel: stream-ordered with the conv launches around it
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_unet_range_state_snapshot(oai_unet* h, int* state_dev, void* stream) {
    OAI_CHECK_ARG(h && state_dev, "oai_unet_range_state_snapshot: null pointer");
    range_state_kernel<<<1, 64, 0, (hipStream_t)stream>>>(h->range_flag, h->census, state_dev);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_unet_range_flag_from_state(const int* state_dev, int* flag_dev, void* stream) {
    OAI_CHECK_ARG(state_dev && flag_dev, "oai_unet_range_flag_from_state: null pointer");
    range_flag_from_state_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state_dev, flag_dev, kLowRange);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_unet_census(oai_unet* h, float max_out[OAI_UNET_NUM_LAYERS], int reset, void* stream) {
    OAI_CHECK_ARG(h && max_out, "oai_unet_census: null pointer");
    hipStream_t st = (hipStream_t)stream;
    unsigned host[18 * 16];
    OAI_CHECK_HIP(hipMemcpyAsync(host, h->census, sizeof(host), hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 18; ++k) {
        unsigned m = 0;
        for (int i = 0; i < 16; ++i) m = host[k * 16 + i] > m ? host[k * 16 + i] : m;
        memcpy(&max_out[k], &m, 4);
    }
    if (reset) {
        OAI_CHECK_HIP(hipMemsetAsync(h->census, 0, sizeof(host), st));
        OAI_CHECK_HIP(hipMemsetAsync(h->range_flag, 0, sizeof(int), st));
    }
    return OAI_OK;
}

int oai_unet_get_act_exponents(const oai_unet* h, int e_out[OAI_UNET_NUM_LAYERS], int* calibrated) {
    OAI_CHECK_ARG(h && e_out, "oai_unet_get_act_exponents: null pointer");
    for (int k = 0; k < 18; ++k) e_out[k] = h->act_exp[k];
    if (calibrated) *calibrated = h->calibrated ? 1 : 0;
    return OAI_OK;
}

int oai_unet_set_act_exponents(oai_unet* h, const int e[OAI_UNET_NUM_LAYERS]) {
    OAI_CHECK_ARG(h && e, "oai_unet_set_act_exponents: null pointer");
    for (int k = 0; k < 17; ++k) OAI_CHECK_ARG(e[k] >= -100 && e[k] <= 100, "oai_unet_set_act_exponents: exponent %d of layer %d outside [-100, 100]", e[k], k);
    OAI_CHECK_ARG(e[DC0] == 0, "oai_unet_set_act_exponents: dc0 produces logits, its exponent must be 0");
    OAI_CHECK_HIP(hipDeviceSynchronize());               // the arrays rewritten below may be in use by queued launches (rare call: once per network)
    for (int k = 0; k < 18; ++k) h->act_exp[k] = e[k];
    h->calibrated = true;
    if (!h->L[EC0].scale_f16) return OAI_OK;             // fp16x3 not set up yet: oai_unet_set_precision packs with these exponents
    for (int k : {DC8, DC5, DC2}) {                      // concat layers: the panel carries 2^(e(src0) - e(src1)) on the skip channels
        const int rel1 = h->act_exp[layer_src0(k)] - h->act_exp[layer_src1(k)];
        if (rel1 != h->L[k].rel1)
            if (int rc = pack_fp16_layer(h, k)) return rc;
    }
    return refresh_fp16_affine(h);
}

int oai_unet_calibrate_step(oai_unet* h, void* stream, int* more) {
    OAI_CHECK_ARG(h && more, "oai_unet_calibrate_step: null pointer");
    OAI_CHECK_ARG(h->precision == OAI_PREC_FP16X3, "oai_unet_calibrate_step: the activation exponents belong to OAI_PREC_FP16X3");
    float mx[18];
    if (int rc = oai_unet_census(h, mx, 1, stream)) return rc;
    int e[18], changed = 0, reported = 0;
    for (int k = 0; k < 18; ++k) e[k] = h->act_exp[k];
    for (int k = 0; k < 17; ++k) {
        if (!(mx[k] > 0.0f)) continue;                   // nothing stored (layer not run by the split-resident kernels, or all zero): keep
        ++reported;
        if (!std::isfinite(mx[k])) { e[k] = e[k] - 32 < -100 ? -100 : e[k] - 32; changed = 1; continue; }
        int b;
        frexpf(mx[k], &b);                               // mx = m 2^b, m in [0.5, 1): mx in [2^(b-1), 2^b)
        if (b - 1 >= kTargetExp - 1 && b - 1 <= kTargetExp + 1) continue;       // inside [2^9, 2^12): leave it (a verify pass ends here)
        e[k] += kTargetExp - (b - 1);
        if (e[k] > 100) e[k] = 100;
        if (e[k] < -100) e[k] = -100;
        changed = 1;
    }
    *more = changed;
    // a census that holds nothing (option census 0 / sres 0 before the first run, or no pass queued since the last step) says
    // nothing about the exponents: refuse instead of reporting "calibrated" with whatever they were
    if (!reported)
        return oai::set_error(OAI_ERR_ARG, "oai_unet_calibrate_step: the range census is empty (no fp16x3 pass since the last step, or option census / sres is 0)");
    if (!changed) { h->calibrated = true; return OAI_OK; }
    const bool was = h->calibrated;
    const int rc = oai_unet_set_act_exponents(h, e);
    h->calibrated = was;                                 // exponents moved: only a later pass that finds every layer inside the window calibrates
    return rc;
}

int oai_unet_profile(oai_unet* h, int enable) {
    OAI_CHECK_ARG(h, "oai_unet_profile: null handle");
    h->profile = enable != 0;
    return OAI_OK;
}

int oai_unet_profile_read(oai_unet* h, double* conv3_ms, long long* conv3_launches) {
    OAI_CHECK_ARG(h && conv3_ms && conv3_launches, "oai_unet_profile_read: null pointer");
    double ms = 0.0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        OAI_CHECK_HIP(hipEventSynchronize(h->ev_pool[i + 1]));
        float t = 0.0f;
        OAI_CHECK_HIP(hipEventElapsedTime(&t, h->ev_pool[i], h->ev_pool[i + 1]));
        ms += t;
    }
    *conv3_ms = ms;
    *conv3_launches = (long long)(h->ev_used / 2);
    h->ev_used = 0;
    return OAI_OK;
}

double oai_unet_tile_flops_conv3(const oai_unet* h, int td, int th, int tw, const int overlap[3], int trimmed) {
    if (!h) return 0.0;
    const int tile[3] = {td, th, tw};
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) { lo[i] = overlap ? overlap[i] : 0; hi[i] = tile[i] - lo[i]; }
    Box need[18];
    plan_regions(tile, lo, hi, trimmed != 0, need);
    double f = 0;
    for (int k = 1; k < 17; ++k)
        if (kKind[k] == 0 || kKind[k] == 1) f += layer_flops(h, k, need[k]);
    return f;
}

void oai_unet_destroy(oai_unet* h) {
    if (!h) return;
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

size_t oai_unet_workspace_bytes(const oai_unet* h, int td, int th, int tw, int batch) {
    if (!h || td <= 0 || th <= 0 || tw <= 0 || batch <= 0) return 0;
    return plan_workspace(h, td, th, tw, batch).total;
}

double oai_unet_tile_flops(const oai_unet* h, int td, int th, int tw, const int overlap[3], int trimmed) {
    if (!h) return 0.0;
    const int tile[3] = {td, th, tw};
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) { lo[i] = overlap ? overlap[i] : 0; hi[i] = tile[i] - lo[i]; }
    Box need[18];
    plan_regions(tile, lo, hi, trimmed != 0, need);
    double f = 0;
    for (int k = 0; k < 18; ++k) f += layer_flops(h, k, need[k]);
    return f;
}

static int check_tile(int td, int th, int tw) {
    OAI_CHECK_ARG(td >= 8 && th >= 8 && tw >= 8 && td % 8 == 0 && th % 8 == 0 && tw % 8 == 0,
                  "tile size %dx%dx%d must be a positive multiple of 8 per axis (three 2x poolings)", td, th, tw);
    return OAI_OK;
}

int oai_unet_forward_tiles(oai_unet* h, const float* tiles, float* logits, int B, int td, int th, int tw,
                           void* ws, size_t ws_bytes, void* stream) {
    OAI_CHECK_ARG(h && tiles && logits && ws, "oai_unet_forward_tiles: null pointer");
    OAI_CHECK_ARG(B > 0, "oai_unet_forward_tiles: B must be > 0");
    if (int rc = check_tile(td, th, tw)) return rc;
    // as many tiles per pass as the workspace holds
    int batch = B;
    while (batch > 1 && plan_workspace(h, td, th, tw, batch).total > ws_bytes) --batch;
    const Plan plan = plan_workspace(h, td, th, tw, batch);
    if (plan.total > ws_bytes)
        return set_error(OAI_ERR_WORKSPACE, "oai_unet_forward_tiles: workspace %zu B < %zu B needed for one tile", ws_bytes, plan.total);
    const int tile[3] = {td, th, tw};
    const int lo[3] = {0, 0, 0};
    Box need[18];
    plan_regions(tile, lo, tile, false, need);
    TileSource src{};
    src.vol = nullptr; src.td = td; src.th = th; src.tw = tw;
    const size_t v0 = (size_t)td * th * tw;
    for (int s = 0; s < B; s += batch) {
        const int n = B - s < batch ? B - s : batch;
        src.tiles = tiles + (size_t)s * v0;
        src.tile_begin = 0;
        if (int rc = run_batch(h, src, n, need, 2, logits + (size_t)s * h->n_classes * v0, (char*)ws, plan, (hipStream_t)stream)) return rc;
    }
    return OAI_OK;
}

// What a tile must produce along one axis: its kept centre [ovl, tile-ovl) clipped to the part of the volume that
// Partition.assemble keeps (image_transforms.py:504 trims to the image, :509-513 zeroes a frame of `crop` voxels).
__host__ __device__ static void keep_interval(int t, int eff, int ovl, int size, int crop, int& lo, int& hi) {
    const int v0 = t * eff;                                   // volume coordinate of the kept centre's first voxel
    const int a = crop - v0 > 0 ? crop - v0 : 0;
    const int b = size - crop - v0 < eff ? size - crop - v0 : eff;
    lo = ovl + a;
    hi = ovl + b;                                             // hi <= lo: this tile contributes nothing
}

struct SegGeom {
    int eff[3], grid[3], ntiles;
};

static int seg_geometry(int D, int H, int W, const int tile[3], const int overlap[3], SegGeom& g) {
    const int size[3] = {D, H, W};
    for (int i = 0; i < 3; ++i) {
        OAI_CHECK_ARG(overlap[i] >= 0 && tile[i] - 2 * overlap[i] > 0, "overlap too large for the tile");
        g.eff[i] = tile[i] - 2 * overlap[i];
        g.grid[i] = (size[i] + g.eff[i] - 1) / g.eff[i];
    }
    g.ntiles = g.grid[0] * g.grid[1] * g.grid[2];
    return OAI_OK;
}

struct SegParams {          // everything tile_regions needs, by value (kernel argument)
    int size[3], tile[3], overlap[3], crop[3], eff[3], grid[3];
    int trimmed;
};

// need[layer] of tile t (z-major index), or all-empty boxes for a tile that contributes nothing
__host__ __device__ static void tile_regions(int t, const SegParams& p, Box need[18]) {
    const int idx[3] = {t / (p.grid[1] * p.grid[2]), (t / p.grid[2]) % p.grid[1], t % p.grid[2]};
    int lo[3], hi[3];
    bool dead = false;
    for (int i = 0; i < 3; ++i) {
        keep_interval(idx[i], p.eff[i], p.overlap[i], p.size[i], p.crop[i], lo[i], hi[i]);
        if (hi[i] <= lo[i]) dead = true;
        if (!p.trimmed) { lo[i] = p.overlap[i]; hi[i] = p.tile[i] - p.overlap[i]; }
    }
    plan_regions(p.tile, lo, hi, p.trimmed != 0, need);
    if (dead && p.trimmed)
        for (int k = 0; k < 18; ++k) for (int i = 0; i < 3; ++i) need[k].lo[i] = need[k].hi[i] = 0;
}

// device table [layer][n][6] of tiles [t0, t0+n): computed on the device (no host copy: graph-capturable);
// up-conv rows hold the INPUT box (the halved output box)
__global__ void box_table_kernel(SegParams p, int t0, int n, int* __restrict__ table) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Box need[18];
    tile_regions(t0 + t, p, need);
    for (int k = 0; k < 18; ++k) {
        int* row = table + ((size_t)k * n + t) * 6;
        const bool up = layer_kind(k) == 2;
        for (int i = 0; i < 3; ++i) {
            row[i] = up ? need[k].lo[i] / 2 : need[k].lo[i];
            row[3 + i] = up ? (need[k].hi[i] + 1) / 2 : need[k].hi[i];
        }
    }
}

static SegParams seg_params(int D, int H, int W, const int tile[3], const int overlap[3], const int* crop, const SegGeom& g, bool trimmed) {
    SegParams p;
    const int size[3] = {D, H, W};
    for (int i = 0; i < 3; ++i) {
        p.size[i] = size[i]; p.tile[i] = tile[i]; p.overlap[i] = overlap[i]; p.crop[i] = crop ? crop[i] : 0;
        p.eff[i] = g.eff[i]; p.grid[i] = g.grid[i];
    }
    p.trimmed = trimmed ? 1 : 0;
    return p;
}

// bytes of the two volume-wide tensors of the shared encoder pass, or 0 when its conditions do not hold for this geometry
static size_t shared_enc_bytes(const oai_unet* h, const int tile[3], const int overlap[3], const SegGeom& g, int P[3]) {
    if (!h->sres || !h->opt_shared || !h->fuse_first || h->variant != 0 || h->sres_mrep != 4 || h->sres_ring || h->b_lds) return 0;
    if (h->L[EC0].cout != 32 || h->L[EC1].c0 != 32 || h->L[EC1].cout % 16 != 0) return 0;
    const int blk[3] = {4, 8, 16};                                         // the main block shape must tile the padded volume and the tile exactly
    for (int i = 0; i < 3; ++i) {
        P[i] = g.eff[i] * g.grid[i] + 2 * overlap[i];
        // tile origins (multiples of eff) on block boundaries: a block of the pass lies inside or outside a tile, never across its start (scatter
        // copy-out), and pooling windows are aligned; overlap >= 4: dc2's skip read stays >= 2 voxels inside the tile
        if (tile[i] % blk[i] || P[i] % blk[i] || g.eff[i] % blk[i] || overlap[i] < 4 || tile[i] < 8) return 0;
    }
    const size_t pv = (size_t)P[0] * P[1] * P[2];
    if (pv * 128 >= (1ull << 32)) return 0;                                // 32-bit piece offsets inside the pass's own planes
    const size_t nch = (size_t)h->L[EC1].cout / 16;
    return (nch * (pv / 8) * 64 + 255) / 256 * 256;
}

size_t oai_segment_workspace_bytes(const oai_unet* h, int D, int H, int W, const int tile[3], const int overlap[3], int batch) {
    if (!h || !tile || !overlap || batch <= 0) return 0;
    SegGeom g;
    if (seg_geometry(D, H, W, tile, overlap, g)) return 0;
    int P[3];
    return plan_workspace(h, tile[0], tile[1], tile[2], batch).total + shared_enc_bytes(h, tile, overlap, g, P);
}

int oai_segment_tiles(oai_unet* h, const float* vol, int D, int H, int W, const int tile[3], const int overlap[3],
                      const int crop[3], int tile_begin, int tile_end, int out_mode, float* blocks, int batch,
                      void* ws, size_t ws_bytes, void* stream) {
    OAI_CHECK_ARG(h && vol && tile && overlap && blocks && ws, "oai_segment_tiles: null pointer");
    OAI_CHECK_ARG(D > 1 && H > 1 && W > 1, "oai_segment_tiles: volume axes must be > 1 (reflect padding)");
    OAI_CHECK_ARG(out_mode >= 0 && out_mode <= 2, "oai_segment_tiles: out_mode must be 0, 1 or 2");
    if (int rc = check_tile(tile[0], tile[1], tile[2])) return rc;
    SegGeom g;
    if (int rc = seg_geometry(D, H, W, tile, overlap, g)) return rc;
    OAI_CHECK_ARG(0 <= tile_begin && tile_begin <= tile_end && tile_end <= g.ntiles,
                  "oai_segment_tiles: tile range [%d,%d) outside [0,%d)", tile_begin, tile_end, g.ntiles);
    OAI_CHECK_ARG(batch > 0, "oai_segment_tiles: batch must be > 0");
    if (crop) for (int i = 0; i < 3; ++i) OAI_CHECK_ARG(crop[i] >= 0, "oai_segment_tiles: negative crop");
    const Plan plan = plan_workspace(h, tile[0], tile[1], tile[2], batch);
    if (plan.total > ws_bytes)
        return set_error(OAI_ERR_WORKSPACE, "oai_segment_tiles: workspace %zu B < %zu B needed for batch %d", ws_bytes, plan.total, batch);
    const bool trimmed = diag_env("OAI_NO_TRIM", 0) == 0;
    hipStream_t st = (hipStream_t)stream;
    TileSource src{};
    src.vol = vol; src.tiles = nullptr; src.D = D; src.H = H; src.W = W;
    src.td = tile[0]; src.th = tile[1]; src.tw = tile[2];
    src.ez = g.eff[0]; src.ey = g.eff[1]; src.ex = g.eff[2];
    src.oz = overlap[0]; src.oy = overlap[1]; src.ox = overlap[2];
    src.gy = g.grid[1]; src.gx = g.grid[2];
    const size_t bvox = (size_t)g.eff[0] * g.eff[1] * g.eff[2];
    int* table_dev = reinterpret_cast<int*>(ws);
    const SegParams sp = seg_params(D, H, W, tile, overlap, crop, g, trimmed);
    Box need[18];
    // ---- shared encoder pass (run_batch: once per batch over the z range its tiles cover), when the geometry and the workspace allow it
    SharedEnc se{};
    bool shared = false;
    if (trimmed && tile_end > tile_begin) {
        int P[3];
        const size_t sb = shared_enc_bytes(h, tile, overlap, g, P);
        if (sb && plan.total + sb <= ws_bytes) {
            shared = true;
            se.SP0 = reinterpret_cast<float*>((char*)ws + plan.total);
            se.PD = P[0]; se.PH = P[1]; se.PW = P[2];
            se.vol = vol; se.D = D; se.H = H; se.W = W;
            for (int i = 0; i < 3; ++i) { se.grid[i] = g.grid[i]; se.eff[i] = g.eff[i]; se.overlap[i] = overlap[i]; }
        }
    }
    for (int c0 = tile_begin; c0 < tile_end; c0 += kMaxTilesPerTable) {
        const int nc = tile_end - c0 < kMaxTilesPerTable ? tile_end - c0 : kMaxTilesPerTable;
        box_table_kernel<<<cdiv(nc, 64), 64, 0, st>>>(sp, c0, nc, table_dev);
        OAI_CHECK_LAUNCH();
        for (int t = c0; t < c0 + nc; t += batch) {
            const int n = c0 + nc - t < batch ? c0 + nc - t : batch;
            // launch boxes of this batch = union over its tiles (host mirror of the same planning code)
            Box uni[18];
            for (int k = 0; k < 18; ++k) for (int i = 0; i < 3; ++i) { uni[k].lo[i] = 1 << 30; uni[k].hi[i] = 0; }
            for (int j = 0; j < n; ++j) {
                tile_regions(t + j, sp, need);
                for (int k = 0; k < 18; ++k) {
                    if (need[k].hi[0] <= need[k].lo[0]) continue;
                    for (int i = 0; i < 3; ++i) {
                        if (need[k].lo[i] < uni[k].lo[i]) uni[k].lo[i] = need[k].lo[i];
                        if (need[k].hi[i] > uni[k].hi[i]) uni[k].hi[i] = need[k].hi[i];
                    }
                }
            }
            if (uni[0].hi[0] == 0) continue;                 // every tile of the batch is dead
            src.tile_begin = t;
            if (int rc = run_batch(h, src, n, uni, out_mode, blocks + (size_t)(t - tile_begin) * h->n_classes * bvox,
                                   (char*)ws, plan, st, table_dev, nc, t - c0, shared ? &se : nullptr)) return rc;
        }
    }
    return OAI_OK;
}

double oai_unet_volume_flops(const oai_unet* h, int D, int H, int W, const int tile[3], const int overlap[3],
                             const int crop[3], int trimmed, int conv3_only) {
    if (!h || !tile || !overlap) return 0.0;
    SegGeom g;
    if (seg_geometry(D, H, W, tile, overlap, g)) return 0.0;
    double f = 0.0;
    Box need[18];
    const SegParams sp = seg_params(D, H, W, tile, overlap, crop, g, trimmed != 0);
    for (int t = 0; t < g.ntiles; ++t) {
        tile_regions(t, sp, need);
        for (int k = 0; k < 18; ++k) {
            if (conv3_only && !(k >= 1 && k < 17 && (kKind[k] == 0 || kKind[k] == 1))) continue;
            f += layer_flops(h, k, need[k]);
        }
    }
    return f;
}

int oai_unet_tile_costs(const oai_unet* h, int D, int H, int W, const int tile[3], const int overlap[3], const int crop[3],
                        double* costs_host, int n_tiles) {
    OAI_CHECK_ARG(h && tile && overlap && costs_host, "oai_unet_tile_costs: null pointer");
    SegGeom g;
    if (int rc = seg_geometry(D, H, W, tile, overlap, g)) return rc;
    OAI_CHECK_ARG(n_tiles == g.ntiles, "oai_unet_tile_costs: the volume has %d tiles, not %d", g.ntiles, n_tiles);
    Box need[18];
    const SegParams sp = seg_params(D, H, W, tile, overlap, crop, g, true);
    for (int t = 0; t < g.ntiles; ++t) {
        tile_regions(t, sp, need);
        double f = 0.0;
        for (int k = 0; k < 18; ++k) f += layer_flops(h, k, need[k]);
        costs_host[t] = f;
    }
    return OAI_OK;
}

int oai_stitch_blocks(const float* blocks, int ncls, int D, int H, int W, const int tile[3], const int overlap[3],
                      const int crop[3], float* maps, void* stream) {
    StitchRanges rg{};
    return stitch_impl(blocks, ncls, D, H, W, tile, overlap, crop, maps, stream, rg);
}

int oai_stitch_blocks_ranged(const float* blocks, int ncls, int D, int H, int W, const int tile[3], const int overlap[3],
                             const int crop[3], const int* bounds_host, int n_ranges, int slot_stride, float* maps, void* stream) {
    OAI_CHECK_ARG(bounds_host, "oai_stitch_blocks_ranged: null pointer");
    OAI_CHECK_ARG(n_ranges >= 1 && n_ranges <= 64, "oai_stitch_blocks_ranged: 1..64 ranges, got %d", n_ranges);
    StitchRanges rg{};
    rg.n = n_ranges; rg.stride = slot_stride;
    OAI_CHECK_ARG(bounds_host[0] == 0, "oai_stitch_blocks_ranged: the ranges must start at tile 0");
    for (int r = 0; r <= n_ranges; ++r) {
        rg.bound[r] = bounds_host[r];
        if (r) OAI_CHECK_ARG(bounds_host[r] >= bounds_host[r - 1] && bounds_host[r] - bounds_host[r - 1] <= slot_stride,
                             "oai_stitch_blocks_ranged: range %d is not ascending or longer than the slot stride %d", r - 1, slot_stride);
    }
    if (tile && overlap) {
        long long nt = 1;
        const int size[3] = {D, H, W};
        for (int i = 0; i < 3; ++i) { const int e = tile[i] - 2 * overlap[i]; if (e > 0) nt *= (size[i] + e - 1) / e; }
        OAI_CHECK_ARG(bounds_host[n_ranges] == nt, "oai_stitch_blocks_ranged: the ranges cover %d tiles, the volume has %lld", bounds_host[n_ranges], nt);
    }
    return stitch_impl(blocks, ncls, D, H, W, tile, overlap, crop, maps, stream, rg);
}

}  // extern "C"

static int stitch_impl(const float* blocks, int ncls, int D, int H, int W, const int tile[3], const int overlap[3],
                       const int crop[3], float* maps, void* stream, const StitchRanges& rg) {
    OAI_CHECK_ARG(blocks && maps && tile && overlap, "oai_stitch_blocks: null pointer");
    int eff[3], grid[3];
    const int size[3] = {D, H, W};
    for (int i = 0; i < 3; ++i) {
        eff[i] = tile[i] - 2 * overlap[i];
        OAI_CHECK_ARG(eff[i] > 0, "oai_stitch_blocks: overlap too large for the tile");
        grid[i] = (size[i] + eff[i] - 1) / eff[i];
    }
    const size_t total = (size_t)ncls * D * H * W;
    if (crop && (crop[0] == 0 || crop[1] == 0 || crop[2] == 0)) {
        // image_transforms.py:509-513 copies [c:-c] per axis into a zero array; with c == 0 the numpy slice 0:-0 is EMPTY, so a
        // crop_size with a zero component (an overlap of 0 on some axis) yields an all-zero map in the reference.  Reproduced.
        OAI_CHECK_HIP(hipMemsetAsync(maps, 0, total * sizeof(float), (hipStream_t)stream));
        return OAI_OK;
    }
    size_t nblk = (total + 255) / 256;
    if (nblk > 256 * 32) nblk = 256 * 32;
    stitch_kernel<<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(blocks, ncls, D, H, W, eff[0], eff[1], eff[2], grid[1], grid[2],
                                                                crop ? crop[0] : 0, crop ? crop[1] : 0, crop ? crop[2] : 0, maps, rg);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}
