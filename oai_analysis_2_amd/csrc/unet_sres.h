// Split-resident fp16x3 path: activations live in HBM already split into two fp16 terms.
//
// Format S of a C-channel tensor:  [tile][C/16 chunks][z][y][x][2 terms][16 channels] fp16  = 4 bytes per element, the same
// footprint as fp32, so every buffer of the fp32 path is reused as is.  x = t0 + t1 with t0 = fp16(x), t1 = fp16(x - t0)
// (22 mantissa bits).  A (voxel, chunk) "record" is 64 contiguous bytes = four 16-byte "slots" (term, channel half) = exactly
// the four MFMA A-operand fragments v_mfma_f32_32x32x16_f16 wants for that voxel and k-chunk.
//
// Why: the first fp16x3 kernel (conv3_igemm_bf16s) keeps fp32 in memory and splits every halo voxel while staging it -- once
// per consumer workgroup and cout block, ~2.8 x ncb times per element -- and that VALU work competes with the MFMAs for
// issue slots (profiles/r01_ablation.md).  Here each element is split ONCE, by the epilogue that produces it; staging is a
// pure copy done by the LDS-DMA engine (global_load_lds, 16 B per lane, no VGPRs, no VALU); two workgroups share a CU so that
// one multiplies while the other's DMA lands.
//
// LDS image of a halo box: record r = halo voxel (hz*HY + hy)*HX + hx at byte 64*r, its four slots XOR-swizzled by
// key(hx) = (hx >> 2) & 3 so that the 16 lanes of a ds_read_b128 group (16 consecutive x, or 2 rows of 8) hit 16 different
// 16-byte bank groups.  global_load_lds writes lane-linearly, so the swizzle is applied to the SOURCE address (rule 21 of the
// CDNA guide): LDS slot p of record r receives logical slot p ^ key.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "unet_kernels.h"

namespace oai {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned split2_f16(float x, unsigned& lo_bits) {     // returns hi bits
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    lo_bits = __builtin_bit_cast(unsigned short, l);
    return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ float join2_f16(unsigned short hi, unsigned short lo) {
    return (float)__builtin_bit_cast(_Float16, hi) + (float)__builtin_bit_cast(_Float16, lo);
}

// Range bookkeeping of a workgroup's stored activations (vmax = this lane's max |value|, >= 0): one wave reduction, then lane 0 sets the
// overflow flag (fp16 cannot hold |x| > 65504: report, never silently inf) and folds the wave's maximum into the layer's census -- 16 words
// per layer, picked by block id so that the ~10^5 waves of a launch do not queue on one L2 atomic unit.  The census is what
// oai_unet_calibrate_step turns into per-layer power-of-two activation exponents, and what the range flag's LOW bit is derived from.
// `seen` (census_peek at the start of the block): the word's value as this workgroup saw it -- the atomic is issued only for a larger maximum.
// After the first blocks of a launch nearly every wave's maximum is below what the word already holds; unconditionally, the ~10^5-10^6 waves
// of a launch sent one atomic each to the SAME 64-byte line (16 words per layer): 2 % of a pass (option census 0 vs 1, same box).  A stale peek
// (the scalar cache is invalidated per dispatch; other workgroups raise the word meanwhile) only costs a redundant atomic, never loses a maximum.
__device__ __forceinline__ unsigned census_peek(const unsigned* census) {
    typedef const __attribute__((address_space(4))) unsigned* cuptr;
    return census ? *((cuptr)census + (blockIdx.x & 15)) : 0u;      // wave-uniform: a scalar load
}
// maximum of a NON-NEGATIVE (or NaN) value over the wave, wave-uniform: six DPP steps on the VALU (quad swaps, row mirrors, row broadcasts: lane 63
// ends up with the total) and one v_readlane -- __shfl_xor is six serial ds_bpermute round trips through the LDS crossbar at the very end of a block.
// (Lanes a masked step does not write keep their value -- update_dpp's `old` operand is the value itself.  fmaxf drops a NaN, so a NaN is folded in
// as +inf first: the overflow test below sees it either way.)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = v == v ? v : __builtin_inff();
    auto step = [](float x, auto ctrl, auto rmask) __attribute__((always_inline)) {
        return fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xF, false)));
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});       // quad_perm 1,0,3,2
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});       // quad_perm 2,3,0,1
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});      // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});      // row_mirror: every lane of a row holds the row's maximum
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});      // row_bcast15 into rows 1, 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});      // row_bcast31 into rows 2, 3: lane 63 = the wave's maximum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ void census_note(unsigned* census, int* range_flag, float vmax, unsigned seen = 0u) {
    vmax = wave_max_nonneg(vmax);
    if ((threadIdx.x & 63) == 0) {
        if (!(vmax <= 65504.0f)) atomicOr(range_flag, 1);
        if (census && vmax > 0.0f && __float_as_uint(vmax) > seen) atomicMax(census + (blockIdx.x & 15), __float_as_uint(vmax));
    }
}

// byte offset of the 64-byte record (tile, chunk, voxel) of a format-S tensor with `nch` chunks and `plane` voxels per tile.
// Chunk-planar: the records of x-consecutive voxels of one chunk are contiguous, so a halo row is one dense run for the LDS-DMA
// (with the chunks interleaved per voxel every DMA instruction touched 16 lines instead of 4-5: 7 % of the segmentation).
__device__ __forceinline__ size_t srec(size_t tile, int nch, size_t plane, int chunk, size_t voxel) {
    return (((size_t)tile * nch + chunk) * plane + voxel) * 64;
}

// Store one fp32 value per lane as its two fp16 terms into a format-S row, lanes = consecutive channels `co` of one voxel.
// Lane pairs (co even, co+1) swap one term with a single DPP move so that every lane issues ONE dword store:
// even lane -> (hi[co], hi[co+1]) into term 0, odd lane -> (lo[co-1], lo[co]) into term 1.
__device__ __forceinline__ void store_split_pair(unsigned char* voxel_base /*record of chunk 0 of the voxel*/, size_t chunk_stride /*plane * 64*/,
                                                 int co, float v, bool pred, int* range_flag) {
    if (pred && !(fabsf(v) <= 65504.0f)) atomicOr(range_flag, 1);            // fp16 cannot hold it: report, never silently inf
    unsigned lo;
    const unsigned hi = split2_f16(v, lo);
    const bool odd = co & 1;
    const unsigned send = odd ? hi : lo;                                     // what the partner lane needs
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1 /*quad_perm 1,0,3,2*/, 0xF, 0xF, true);
    const unsigned word = odd ? (recv | (lo << 16)) : (hi | (recv << 16));
    if (pred)
        *reinterpret_cast<unsigned*>(voxel_base + (co >> 4) * chunk_stride + (odd ? 32 : 0) + ((co & 15) >> 1) * 4) = word;
}

// Epilogue form of the split: a lane holds ONE channel of two voxels A, B (two C/D rows); its partner lane (lane ^ 1) holds the
// neighbouring channel of the same voxels.  Both values are split with packed conversions, the lane pair swaps halves (one DPP
// move + one byte permute per term), and the EVEN lane ends up with voxel A's two record words -- (hi[c], hi[c+1]) and
// (lo[c], lo[c+1]) -- the ODD lane with voxel B's.  ~9 VALU per element instead of ~18 for two store_split_pair-style splits.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_two_voxels(float vA, float vB, unsigned sel /*odd lane 0x03020706 : even 0x05040100*/,
                                                 unsigned& w_hi, unsigned& w_lo) {
    const f32x2 v = {vA, vB};
    const f16x2 hi = __builtin_convertvector(v, f16x2);
    const f32x2 res = v - __builtin_convertvector(hi, f32x2);
    const f16x2 lo = __builtin_convertvector(res, f16x2);
    const unsigned H = __builtin_bit_cast(unsigned, hi), L = __builtin_bit_cast(unsigned, lo);
    const unsigned PH = (unsigned)__builtin_amdgcn_mov_dpp((int)H, 0xB1 /*quad_perm 1,0,3,2*/, 0xF, 0xF, true);
    const unsigned PL = (unsigned)__builtin_amdgcn_mov_dpp((int)L, 0xB1, 0xF, 0xF, true);
    w_hi = __builtin_amdgcn_perm(PH, H, sel);
    w_lo = __builtin_amdgcn_perm(PL, L, sel);
}

// One LDS-DMA piece (global_load_lds_dwordx4: lane l moves 16 bytes from its global address to LDS byte lds_addr + 16 l; lds_addr is
// wave-uniform) as inline assembly.  Why not the builtin: the compiler's wait-count pass treats every LDS-DMA as a store to ALL of LDS and
// puts `s_waitcnt vmcnt(0)` in front of the next LDS read -- also when that read is from another ring slot -- and __syncthreads() drains
// vmcnt as well: a "three-stage ring" built from the builtin waits for the piece it has just requested (the up-conv's k loop spent
// ~10 k cycles per step that way).  Through the asm the pass sees no LDS-DMA; the counted waits are written by hand next to the barrier.
// M0 is saved and restored so that the compiler's own view of it (it sets M0 for the builtin form elsewhere) stays true.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr /* wave-uniform; an SGPR */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {      // LDS byte address of a __shared__ object
    return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

// ---- converters (bring-up / B3 seam only): fp32 channels-last <-> format S ------------------------------------------------
__global__ void __launch_bounds__(256) f32_to_sres_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, size_t nvox, int C) {
    const int nch = (C + 15) / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvox * nch * 4; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i & 3);                         // 4 channels of the chunk
        const size_t rec = i >> 2;
        const size_t vox = rec / nch;
        const int ch = (int)(rec - vox * nch);
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = ch * 16 + 4 * q + j; x[j] = c < C ? in[vox * C + c] : 0.0f; }
        u16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { unsigned l; hi[j] = (unsigned short)split2_f16(x[j], l); lo[j] = (unsigned short)l; }
        unsigned char* o = out + srec(0, nch, nvox, ch, vox);
        *reinterpret_cast<u16x4*>(o + q * 8) = hi;
        *reinterpret_cast<u16x4*>(o + 32 + q * 8) = lo;
    }
}

typedef float f32x4m __attribute__((ext_vector_type(4)));         // a native vector: inline asm can tie it to a 128-bit VGPR tuple (= f32x4 of unet_sres2.h)
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
// 16 bytes per lane from wave-uniform base (SGPR pair) + per-lane 32-bit offset + immediate, from inline assembly (= gload16_asm of unet_sres2.h)
template <int IMM>
__device__ __forceinline__ f32x4m gload16m(const void* sbase, unsigned voff) {
    f32x4m v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
    return v;
}

// ---- conv3, split-resident ---------------------------------------------------------------------------------------------
// Same decomposition as conv3_igemm_bf16s (4 waves, 2 z slices per block, two workgroups per CU so that one computes
// while the other stages), but the halo arrives by LDS-DMA: no staging registers, no staging VALU.
// MREP = z slices per block = accumulator row blocks per wave: 4 halves the weight-fragment bytes per MFMA (the L1 path
// moving 4 KiB of B per wave per tap is what bounds the 3-pass kernel), at 128 accumulator VGPRs.
// RING (MREP = 2 only): the halo lives in a ring of six z-plane slots instead of one 4-plane box.  A chunk's taps run in dz
// order and touch planes (dz, dz+1), so the planes of chunk c+1 can be DMA'd into the two spare slots and into the slots chunk c
// frees as it goes -- every plane is requested three dz-phases (~10 us) before its first use, nothing is ever waited for, and
// the block synchronises once per phase.  Same footprint as the MREP = 4 box (6 x 11.5 KB), two workgroups per CU.
// FIRST (ec1 of the network, Cin = 32 = two chunks): the input of this layer is ec0 = relu(Conv3d(1 -> 32, k3 p1)(tile)), which costs 27
// FMAs per value.  Instead of a launch that writes it to memory (10.7 GB per 160-tile pass) and a halo box that reads it back by
// LDS-DMA (2.1 x that, HBM-latency misses that occupy the CU's L1 in front of the epilogue's stores), the block stages the raw
// (HZ+2) x (HY+2) x (HX+2) patch of the volume once (7.7 KB, gathered with the reflect padding of Partition.__call__) and every
// thread computes its halo records on the VALU -- the same fmaf chain, scale/shift, ReLU and split as conv3_first_sres_kernel, so the
// records are bit-identical to the ones that kernel writes -- while the partner workgroup of the CU owns the matrix pipe.
// BLDS (MREP 4, not FIRST / RING): the weight fragments of a tap -- the same 4 KiB for the four waves of the workgroup, which today each
// fetch all of it through the CU's L1, in order behind the halo misses and the partner's stores -- go through a three-slot LDS ring:
// every wave LDS-DMAs one KiB of the slab two taps ahead, one s_barrier per tap publishes it, and the fragments are read from LDS one tap
// ahead.  12 KB on top of the 68 KB halo box: two workgroups fill the CU's 160 KB exactly.
// M16 (round 5): the taps on v_mfma_f32_16x16x32_f16 instead of 32x32x16 -- what conv3_wino_sres<..., M16> (unet_wino.h) did for the two-group
// Winograd form, for the DIRECT kernel (ec1 with the fused ec0, ec2, dc1 with the fused head, and every layer with option winograd 0).  This
// loop runs at the power wall (1.72-1.74 GHz, profiles/r04_sq_summary.md); at equal cycles per FLOP the 16x16x32 shape holds a ~12-14 % higher
// clock on split-fp16 data.  K = 32 = a PAIR OF TAPS x 16 channels, lanes 0-31 carrying the first tap of the pair, lanes 32-63 the second, so
// that every A operand is a natural record read and every B operand a natural panel load (pack_conv3_m16_panel).  27 taps = 13 pairs + 1:
//     steps 0..8   taps (q, dx 0) | (q, dx 1)                 q = dz * 3 + dy
//     steps 9..11  taps (q, dx 2) | (q + 1, dx 2)             q = 0, 3, 6
//     step  12     taps (2, dx 2) | (5, dx 2)
//     step  13     tap (8, dx 2) alone: lanes 32-63 carry its LOW terms -- [a0 | a1] . [b0 | b0] = a0.b0 + a1.b0 and [a0 | a1] . [b1 | 0] = a0.b1
// (pairs along dx first: the second tap of a pair is then ONE record further in x, or one row / one slice further for the dx = 2 taps -- seven
// per-lane offsets cover all 14 steps).  Per step: pass B [a0 | a0'] . Y' (Y' = the low terms b1 of both taps), pass A [a0 | a0'] . X' (X' = the
// high terms b0), pass C [a1 | a1'] . X'; Y' is dead after pass B and re-requested there, X' in halves behind pass C (32 fragment registers,
// asm loads with counted waits -- the scheme of conv3_wino_sres<1, .., WS, M16>).  = 14 K-32 steps per chunk where 13.5 would be exact.  A wave's
// tile is the same MREP x 32 rows x 64 couts as MREP x 2 tiles of 32 x 32 = MREP x 8 tiles of 16 x 16 (the same accumulator registers): element
// (p, q, i) of the old (m, n) tile at row 16 p + 4 (lane >> 4) + i, cout column 16 q + (lane & 15).  Another summation order than the 32x32x16
// form's (two taps at once), same arithmetic class; EVERY shape of this kernel has the variant, so a layer runs one order whatever shapes cover it.
template <int MREP, int RX, int RY, int WY, int WX, bool RING = false, bool FIRST = false, bool BLDS = false, bool M16 = false>
__global__ void __launch_bounds__(256, (MREP == 2 && !RING) ? 3 : 2) conv3_igemm_sres(const ConvArgs a, const unsigned char* __restrict__ zero_rec) {
    static_assert(RX * RY == 32 && WY * WX == 4 && (MREP == 2 || MREP == 4) && (!RING || MREP == 2) && (!FIRST || !RING), "bad tile shape");
    static_assert(!BLDS || (MREP == 4 && !RING && !FIRST), "the weight ring is for the default kernel");
    static_assert(!M16 || (!RING && !BLDS), "the 16x16x32 taps: the plain and the ec0-fused form");
    constexpr int NREP = 2, TZ = MREP;
    constexpr int kTY = WY * RY, kTX = WX * RX, HY = kTY + 2, HX = kTX + 2, HZ = TZ + 2;
    constexpr int HVOX = HZ * HY * HX;
    constexpr int PIECES = HVOX * 4;                            // 16-byte slots of the halo buffer
    constexpr int NIT = (PIECES + 255) / 256;                    // LDS-DMA instructions per thread per chunk
    constexpr int NITP = (HY * HX * 4 + 255) / 256;              // RING: ... per thread per z plane
    constexpr int PLB = NITP * 256 * 16;                         // RING: bytes of one plane slot (whole 1-KiB wave writes)
    constexpr int BUF = RING ? 6 * PLB : NIT * 256 * 16;         // bytes
    __shared__ __attribute__((aligned(16))) unsigned char lds[BUF];
    __shared__ __attribute__((aligned(16))) unsigned char blds[BLDS ? 3 * 4096 : 16];     // BLDS: weight slabs of taps g, g+1, g+2 (slot = tap % 3)
    constexpr int PZ = HZ + 2, PY = HY + 2, PX = HX + 2;         // FIRST: raw patch = halo box + ec0's own halo
    __shared__ float raw[FIRST ? PZ * PY * PX : 1];
    __shared__ int rawidx[FIRST ? PZ + PY + PX : 1];             // per-axis source offsets (reflect-padded volume index) or -1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int id = a.xcd_group ? xcd_block_id(a.nblocks, a.xcd_group) : (int)blockIdx.x;
    if (id < 0) return;
    const int cb = id % a.ncb; id /= a.ncb;
    const int bx = id % a.nbx; id /= a.nbx;          // (x, y, z, tile order: a z-fastest order measured 1.5 % slower)
    const int by = id % a.nby; id /= a.nby;
    const int bz = id % a.nbz; id /= a.nbz;
    const int tile = id;
    const int oz0 = a.lo[0] + bz * TZ, oy0 = a.lo[1] + by * kTY, ox0 = a.lo[2] + bx * kTX;
    int blo[3], bhi[3];
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    if (oz0 >= bhi[0] || oz0 + TZ <= blo[0] || oy0 >= bhi[1] || oy0 + kTY <= blo[1] || ox0 >= bhi[2] || ox0 + kTX <= blo[2]) return;

    const int wy = wave / WX, wx = wave % WX;
    const int row = lane & 31, half = lane >> 5;
    const int lx = wx * RX + row % RX, ly = wy * RY + row / RX;
    const int m_lo = max(0, blo[0] - oz0), m_hi = min(MREP, bhi[0] - oz0);

    // -DOAI_DIAG: per-wave cycle sums of the phases of the chunk loop (s_memtime at the phase boundaries), atomically added to
    // a.stamps[phase] by lane 0 at the end: [0] taps end -> barrier 1 entered (loop overhead), [1] barrier 1, [2] DMA issue,
    // [3] DMA wait, [4] barrier 2, [5] 27 taps, [6] epilogue, [7] prologue, [8] waves, [9] chunks.  OAI_STAMP is empty in production.
    // -DOAI_STAMP_SET=1 times the epilogue instead: [0] last tap -> first epilogue barrier passed (wave skew), [1] split + LDS image,
    // [2] barriers, [3] copy-out stores, [4] fused dc0 dot products, [5] fused pool stores, [6] head write + tail, [7] whole chunk loop.
#ifdef OAI_DIAG
#ifndef OAI_STAMP_SET
#define OAI_STAMP_SET 0
#endif
    unsigned st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_chunks = 0;
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define OAI_STAMP_(i) do { if (a.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_sum[i] += (unsigned)(now_ - st_last); st_last = now_; } } while (0)
#if OAI_STAMP_SET == 0
#define OAI_STAMP(i) do { OAI_STAMP_(i); if ((i) == 5) ++st_chunks; } while (0)
#define OAI_STAMPB(i) do { } while (0)
#else
#define OAI_STAMP(i) do { if ((i) == 5) ++st_chunks; } while (0)
#define OAI_STAMPB(i) OAI_STAMP_(i)
#endif
#else
#define OAI_STAMP(i) do { } while (0)
#define OAI_STAMPB(i) do { } while (0)
#endif
    f32x16 acc[M16 ? 1 : MREP][M16 ? 1 : NREP];
    f32x4m acc4[M16 ? MREP : 1][M16 ? NREP : 1][4];                   // M16: [m][n][p * 2 + q], element i at row 16 p + 4 (lane >> 4) + i, column 16 q + (lane & 15)
#pragma unroll
    for (int m = 0; m < (M16 ? 1 : MREP); ++m)
#pragma unroll
        for (int n = 0; n < (M16 ? 1 : NREP); ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
#pragma unroll
    for (int m = 0; m < (M16 ? MREP : 1); ++m)
#pragma unroll
        for (int n = 0; n < (M16 ? NREP : 1); ++n)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc4[m][n][t] = f32x4m{0.0f, 0.0f, 0.0f, 0.0f};
    // element r (0..15) of the (m, n) tile, whichever shape holds it, and its C/D row (voxel index inside the wave's 32-row tile)
    auto acc_el = [&](int m, int n, int r) __attribute__((always_inline)) -> float {
        if constexpr (M16) return acc4[m][n][r >> 2][r & 3];
        else return acc[m][n][r];
    };
    const int col16 = lane & 15, rq16 = lane >> 4;                    // M16: this lane's cout column inside a 16-column tile, its row quarter
    auto row_of = [&](int r) __attribute__((always_inline)) -> int {
        return M16 ? 16 * (r >> 3) + 4 * rq16 + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    };

    const int nch0 = (a.C0 + 15) / 16, nch1 = (a.C1 + 15) / 16, nchunks = nch0 + nch1;
    const size_t plane = (size_t)a.D * a.H * a.W;
    const unsigned char* s0 = reinterpret_cast<const unsigned char*>(a.src0) + srec(tile, nch0, plane, 0, 0);
    const unsigned char* s1 = reinterpret_cast<const unsigned char*>(a.src1) + srec(tile, nch1, plane, 0, 0);

    // ---- staging plan, once per workgroup: for each of this thread's NIT slots, the 32-bit BYTE OFFSET inside one chunk plane of
    // the 16-byte slot it fetches (record of the voxel + swizzled slot), or kNoPiece = outside the tile (Conv3d's zero padding) or
    // beyond the halo box -> the zero record.  One VGPR per piece and four VALU per piece and chunk; the first version kept
    // (voxel << 2 | key) and rebuilt a 64-bit address with a 64-bit multiply-add per piece, which the register allocator answered
    // with ten 8-byte scratch spills RELOADED INSIDE the chunk loop, each behind an `s_waitcnt vmcnt(0)` that also drained the
    // DMA pieces already in flight (the staging of a chunk was serialised, one L2 round trip per piece).
    constexpr unsigned kNoPiece = 0xFFFFFFFFu;
    unsigned poff[NIT];
    const int pslot = tid & 3;       // LDS slot position of this thread's pieces (P = it*256 + tid, so P & 3 = tid & 3)
#pragma unroll
    for (int it = 0; it < (FIRST ? 0 : NIT); ++it) {
        const int r = (it * 256 + tid) >> 2;
        const int hx = r % HX, t2 = r / HX, hy = t2 % HY, hz = t2 / HY;
        const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
        const bool ok = r < HVOX && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        poff[it] = ok ? ((unsigned)((gz * a.H + gy) * a.W + gx) << 6) | (unsigned)((pslot ^ ((hx >> 2) & 3)) << 4) : kNoPiece;
    }
    auto stage = [&](int ch) __attribute__((always_inline)) {
        const bool first = ch < nch0;
        const unsigned char* cb = (first ? s0 : s1) + (size_t)(first ? ch : ch - nch0) * plane * 64;     // wave-uniform chunk plane
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned char* g = poff[it] != kNoPiece ? cb + poff[it] : zero_rec;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds + (it * 256 + wave * 64) * 16), 16, 0, 0);     // cache-policy bits sc0 / sc1 / nt on the DMA: measured, no effect (profiles/r02_conv_phases.md)
        }
    };

    // ---- FIRST: the raw patch, then ec0 on the VALU straight into the halo box (chunk ch = ec0 channels [16 ch, 16 ch + 16))
    float first_max = 0.0f;                       // FIRST: max of the ec0 values this thread produced (>= 0 behind the ReLU)
    if constexpr (FIRST) {
        const TileSource& s = a.first_src;
        if (tid < PZ + PY + PX) {
            const int ax = tid < PZ ? 0 : tid < PZ + PY ? 1 : 2;
            const int p = tid - (ax == 0 ? 0 : ax == 1 ? PZ : PZ + PY);
            const int c = (ax == 0 ? oz0 : ax == 1 ? oy0 : ox0) - 2 + p;                   // tile coordinate
            const int tdim = ax == 0 ? s.td : ax == 1 ? s.th : s.tw;
            int v = -1;
            if ((unsigned)c < (unsigned)tdim) {                                               // outside the tile: Conv3d's zero padding
                if (s.vol) {
                    const int t = s.tile_begin + tile;
                    const int tk = t % s.gx, tj = (t / s.gx) % s.gy, ti = t / (s.gx * s.gy);
                    v = ax == 0 ? reflect_index(ti * s.ez + c - s.oz, s.D) * s.H * s.W
                      : ax == 1 ? reflect_index(tj * s.ey + c - s.oy, s.H) * s.W : reflect_index(tk * s.ex + c - s.ox, s.W);
                } else v = ax == 0 ? c * s.th * s.tw : ax == 1 ? c * s.tw : c;
            }
            rawidx[tid] = v;
        }
        __syncthreads();
        const float* base = s.vol ? s.vol : s.tiles + (size_t)tile * s.td * s.th * s.tw;
        constexpr int RIT = (PZ * PY * PX + 255) / 256;
        float rv[RIT];                                                                        // all gathers in flight before the first LDS write
#pragma unroll
        for (int k = 0; k < RIT; ++k) {
            const int i = min(k * 256 + tid, PZ * PY * PX - 1);
            const int px = i % PX, t2 = i / PX, py = t2 % PY, pz = t2 / PY;
            const int iz = rawidx[pz], iy = rawidx[PZ + py], ix = rawidx[PZ + PY + px];
            rv[k] = (iz | iy | ix) >= 0 ? base[(size_t)iz + iy + ix] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < RIT; ++k)
            if (k * 256 + tid < PZ * PY * PX) raw[k * 256 + tid] = rv[k];                     // (visible behind the chunk loop's first barrier)
    }
    // ec0 for V halo voxels of this thread at a time (records r = (j0 + v) * 256 + tid): every 16 weights of a tap are fetched once
    // (one s_load_dwordx16) for V x 16 FMAs, so the scalar-cache latency hides behind them; V = 2 is what the registers next to the
    // 128 accumulators allow.
    auto first_group = [&](auto vtag, int j0, int ch) __attribute__((always_inline)) {
        constexpr int V = decltype(vtag)::value;
        // [27][32] weights, scale, shift: wave-uniform addresses.  Read through the constant address space so that they become scalar
        // loads (SGPR operands of v_fmac): behind a barrier hipcc otherwise picks per-lane global loads.
        typedef const __attribute__((address_space(4))) float* cptr;
        const float* wg = a.first_w + ch * 16;
        asm volatile("" : "+s"(wg));          // an opaque copy per group: hipcc otherwise keeps the first group's 432 weights for the later groups, in VGPR lanes (v_readlane per FMA)
        const cptr w = (cptr)wg, fsc = (cptr)(a.first_scale + ch * 16), fsh = (cptr)(a.first_shift + ch * 16);
        int r[V], hx[V];
        bool ok[V], inside[V];
        const float* rp[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            r[v] = (j0 + v) * 256 + tid;
            ok[v] = r[v] < HVOX;
            const int rr = ok[v] ? r[v] : 0;
            hx[v] = rr % HX;
            const int t2 = rr / HX, hy = t2 % HY, hz = t2 / HY;
            rp[v] = raw + (hz * PY + hy) * PX + hx[v];
            const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx[v];
            inside[v] = (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        }
        float e[V][16];
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int c = 0; c < 16; ++c) e[v][c] = 0.0f;
        // Tap order of conv3_first_sres_kernel: (dz, dy) outer, dx inner.  Hand-placed prefetch: the 16 weights (one s_load_dwordx16) and
        // the V inputs (LDS) of tap t+1 are requested BEHIND the first FMAs of tap t -- i.e. behind the wait for tap t's operands, scalar
        // loads return out of order so every wait is lgkmcnt(0) -- and have the other 15 V FMAs of tap t to arrive.  (Left to itself the
        // scheduler hoists all 27 s_loads to the top, runs out of SGPRs and parks the weights in VGPR lanes: a v_readlane per FMA.)
        float wc[16], inc[V];
#pragma unroll
        for (int c = 0; c < 16; ++c) wc[c] = w[c];
#pragma unroll
        for (int v = 0; v < V; ++v) inc[v] = rp[v][0];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            float wn[16], inn[V];
#pragma unroll
            for (int v = 0; v < V; ++v) e[v][0] = fmaf(inc[v], wc[0], e[v][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 27) {
#pragma unroll
                for (int c = 0; c < 16; ++c) wn[c] = w[(t + 1) * 32 + c];
#pragma unroll
                for (int v = 0; v < V; ++v) inn[v] = rp[v][(((t + 1) / 9) * PY + ((t + 1) / 3) % 3) * PX + (t + 1) % 3];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 1; c < 16; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) e[v][c] = fmaf(inc[v], wc[c], e[v][c]);
            __builtin_amdgcn_sched_barrier(0);
            // pin the tap: without a chained use the instruction selector's linearisation moves a whole voxel's FMA chain behind the other's
            // (sched_barrier orders only what has a chain), keeping 27 inputs in VGPRs and all 432 weights in VGPR lanes for it
#pragma unroll
            for (int v = 0; v < V; ++v)
                asm volatile("" : "+v"(e[v][0]), "+v"(e[v][1]), "+v"(e[v][2]), "+v"(e[v][3]), "+v"(e[v][4]), "+v"(e[v][5]), "+v"(e[v][6]), "+v"(e[v][7]),
                                  "+v"(e[v][8]), "+v"(e[v][9]), "+v"(e[v][10]), "+v"(e[v][11]), "+v"(e[v][12]), "+v"(e[v][13]), "+v"(e[v][14]), "+v"(e[v][15]));
            if (t + 1 < 27) {
#pragma unroll
                for (int c = 0; c < 16; ++c) wc[c] = wn[c];
#pragma unroll
                for (int v = 0; v < V; ++v) inc[v] = inn[v];
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            u16x8 hi[2], lo[2];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float x = e[v][c] * fsc[c] + fsh[c];
                x = fmaxf(x, 0.0f);
                if (!inside[v]) x = 0.0f;                                                     // ec1's own zero padding at the tile border
                first_max = fmaxf(first_max, x);
                unsigned l;
                hi[c >> 3][c & 7] = (unsigned short)split2_f16(x, l);
                lo[c >> 3][c & 7] = (unsigned short)l;
            }
            if (ok[v]) {
                unsigned char* rec = lds + r[v] * 64;
                const int key = (hx[v] >> 2) & 3;                                             // logical slot s lives at position s ^ key
                *reinterpret_cast<u16x8*>(rec + ((0 ^ key) << 4)) = hi[0];
                *reinterpret_cast<u16x8*>(rec + ((1 ^ key) << 4)) = hi[1];
                *reinterpret_cast<u16x8*>(rec + ((2 ^ key) << 4)) = lo[0];
                *reinterpret_cast<u16x8*>(rec + ((3 ^ key) << 4)) = lo[1];
            }
        }
    };
    auto stage_first = [&](int ch) __attribute__((always_inline)) {
        static_assert(!FIRST || (HVOX > 4 * 256 && HVOX <= 4 * 256 + 64), "the voxel grouping below is for the 6 x 10 x 18 halo box");
        // 2 + 2 (+ 1 in wave 0: records 1024 .. 1079).  A 3 + 2 grouping was measured slower (176.1 vs 173.9 ms per volume): its record
        // addresses spill to scratch around the groups.
#pragma unroll 1
        for (int j0 = 0; j0 < 4; j0 += 2) first_group(std::integral_constant<int, 2>{}, j0, ch);
        if (wave == 0) first_group(std::integral_constant<int, 1>{}, 4, ch);
        // (an if / else over a <2> and a <1> group made hipcc hoist all 432 weight loads above the branch and spill them to VGPR lanes)
    };

    // ---- RING staging plan: this thread's NITP pieces of each of the chunk's four z planes
    int pvr[RING ? 4 : 1][RING ? NITP : 1];
    if constexpr (RING) {
#pragma unroll
        for (int lp = 0; lp < 4; ++lp)
#pragma unroll
            for (int it = 0; it < NITP; ++it) {
                const int r = (it * 256 + tid) >> 2;
                const int hx = r % HX, hy = r / HX;
                const int gz = oz0 - 1 + lp, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
                const bool ok = r < HY * HX && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                pvr[lp][it] = ok ? ((((gz * a.H + gy) * a.W + gx) << 2) | ((hx >> 2) & 3)) : -1;
            }
    }
    auto issue_plane = [&](int ch, int lp) {              // logical plane lp of chunk ch -> slot (4 ch + lp) mod 6
        const bool first = ch < nch0;
        const unsigned char* sb = first ? s0 : s1;
        const int c = first ? ch : ch - nch0;
        unsigned char* dst = lds + ((4 * ch + lp) % 6) * PLB + wave * 1024;
#pragma unroll
        for (int it = 0; it < (RING ? NITP : 0); ++it) {
            const int v = pvr[RING ? lp : 0][RING ? it : 0];
            const unsigned char* g = v >= 0 ? sb + ((size_t)c * plane + (size_t)(v >> 2)) * 64 + (((tid & 3) ^ (v & 3)) << 4) : zero_rec;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(dst + it * 4096), 16, 0, 0);
        }
    };
    // ---- A fragments: byte offset of this lane's voxel for tap (0,0,0), and its swizzled slot per dx and term
    const unsigned char* abase = lds + (ly * HX + lx) * 64;
    int sl[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int t = 0; t < 2; ++t) sl[dx][t] = ((t * 2 + half) ^ (((lx + dx) >> 2) & 3)) * 16;

    constexpr int STEP = 2 * NREP * 64;                             // 16-byte units of weights per tap: [term][nr][lane]
    const float4* wp = a.wpanel + (size_t)cb * nchunks * 27 * STEP + lane;
    float4 bcur[2][NREP], bnext[2][NREP];
    // BLDS: this lane's 16 bytes of the slab this wave copies (wave w moves bytes [1024 w, 1024 w + 1024) of every slab); `gleft` = slabs
    // not yet requested (the pointer sticks to the last slab when they run out, so that every tap issues exactly one piece)
    const unsigned char* bsrc = reinterpret_cast<const unsigned char*>(a.wpanel + (size_t)cb * nchunks * 27 * STEP) + wave * 1024 + lane * 16;
    int gleft = nchunks * 27;
    const unsigned bl_addr = __builtin_amdgcn_readfirstlane(lds_addr_of(blds) + wave * 1024);
    auto issue_b = [&](int slot) __attribute__((always_inline)) {
        lds_dma16(bsrc, bl_addr + slot * 4096);               // (asm form: no compiler-made vmcnt(0) in front of the next LDS read, see lds_dma16)
        if (--gleft > 0) bsrc += 4096;
    };
    auto read_b = [&](float4 (&dst)[2][NREP], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int n = 0; n < NREP; ++n) dst[k][n] = *reinterpret_cast<const float4*>(blds + slot * 4096 + ((k * NREP + n) * 64 + lane) * 16);
    };
    // M16: the panel of pack_conv3_m16_panel, [cb][chunk][step 14][X' | Y'][n2 4][lane] x 16 B = 8 KiB per step; a wave-uniform base forced into an
    // SGPR pair + this lane's 16 bytes, loaded by gload16_asm (invisible to the compiler's wait counting: every use goes through m16_wait first)
    constexpr int STEP16 = 2 * 4 * 64 * 16;                         // bytes per step
    const size_t wp16_v = (size_t)a.wpanel + (size_t)cb * nchunks * 14 * STEP16;
    const unsigned char* wp16 = reinterpret_cast<const unsigned char*>(
        ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(wp16_v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp16_v));
    const unsigned wlane = lane * 16;
    f32x4m bXlo[2], bXhi[2], bY[4];                                 // X' (the high terms b0), couts n2 0, 1 and n2 2, 3, and Y' (the low terms b1) of the running step
    auto m16_req_y = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n) bY[n] = n == 0 ? gload16m<0>(wp16 + 4096, wlane) : n == 1 ? gload16m<1024>(wp16 + 4096, wlane) : n == 2 ? gload16m<2048>(wp16 + 4096, wlane) : gload16m<3072>(wp16 + 4096, wlane);      // (the immediate is 13 bits, signed)
    };
    auto m16_req_lo = [&]() __attribute__((always_inline)) { bXlo[0] = gload16m<0>(wp16, wlane); bXlo[1] = gload16m<1024>(wp16, wlane); };
    auto m16_req_hi = [&]() __attribute__((always_inline)) { bXhi[0] = gload16m<2048>(wp16, wlane); bXhi[1] = gload16m<3072>(wp16, wlane); };
    if constexpr (M16) {
        // (no request here: the fragments of a chunk's step 0 are requested at the chunk's top, see run_chunks -- an asm load must never be in
        // flight across a control-flow merge: the compiler believes its result register already holds the value and may COPY it on an edge)
        asm volatile("s_nop 4" : "+s"(wp16) :: "memory");          // wp16 has just been made uniform by v_readfirstlane: five wait states before a VMEM reads it
    } else if constexpr (BLDS) {
        issue_b(0); issue_b(1); issue_b(2);
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int n = 0; n < NREP; ++n) bcur[k][n] = wp[(k * NREP + n) * 64];
        wp += STEP;
    }

    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    float4 acur[2][MREP];           // (a register prefetch of the next tap's A fragments was measured: no gain, and it costs MREP 2 its third workgroup per CU)
    OAI_STAMP(7);
    if constexpr (RING) {
        const int arel = (ly * HX + lx) * 64;                          // this lane's voxel inside a plane, tap (dy, dx) = (0, 0)
        issue_plane(0, 0); issue_plane(0, 1); issue_plane(0, 2); issue_plane(0, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            int sb[4];
#pragma unroll
            for (int lp = 0; lp < 4; ++lp) sb[lp] = ((4 * ch + lp) % 6) * PLB + arel;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
                if (t % 9 == 0) {                                        // a new dz phase
                    if (t > 0 || ch > 0) {
                        // every DMA requested in earlier phases is older than the 2*NREP weight loads in flight: it has landed
                        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        __syncthreads();                                 // ... for everybody; and the slots refilled below are read out
                    }
                    if (ch + 1 < nchunks) {
                        if (dz == 0) { issue_plane(ch + 1, 0); issue_plane(ch + 1, 1); }
                        else issue_plane(ch + 1, dz + 1);
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int m = 0; m < MREP; ++m)
                        acur[k][m] = *reinterpret_cast<const float4*>(lds + sb[m + dz] + (dy * HX + dx) * 64 + sl[dx][k]);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) bnext[k][n] = wp[(k * NREP + n) * 64];
                wp += STEP;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int m = 0; m < MREP; ++m) {
                        if (m >= m_lo && m < m_hi) {
#pragma unroll
                            for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[PA[p]][m], bcur[PB[p]][n], acc[m][n]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) bcur[k][n] = bnext[k][n];
            }
        }
    } else {
        // One chunk = 27 taps x 24 MFMAs.  Pass order per tap: a0.b0, a0.b1, a1.b0 -- so the registers of the a0 fragments are dead
        // after the second pass and those of a1 before the third: the a1 fragments of tap t are read from LDS at the top of the tap
        // (needed 16 MFMAs = 512 cycles later) and the a0 fragments of tap t+1 behind the second pass (needed 8 MFMAs later): the
        // LDS latency never meets an MFMA that waits for it, at no extra register (a second set of A registers does not fit).
        // Blocks whose four z slices are all inside the tile's box (the large majority) run a branch-free stream; the first
        // version tested `m >= m_lo && m < m_hi` around every pair of MFMAs, which hipcc turned into twelve taken branches per tap
        // with the MFMA pairs in out-of-line blocks behind `s_waitcnt lgkmcnt(0)`.
        // ML = number of live z slices of the block (slices [0, ML) are inside the tile's box): a compile-time count, so that the
        // tap stream is branch-free and the dead slices cost neither LDS reads nor MFMAs.  Blocks whose live slices do not start at
        // 0 (rare: a border tile's box starting inside a block) run ML = MREP; their dead rows accumulate values that the
        // epilogue never stores (rows of an MFMA are independent).
        auto run_chunks = [&](auto ml_tag) __attribute__((always_inline)) {
            constexpr int ML = decltype(ml_tag)::value;
            auto load_a = [&](float4 (&dst)[MREP], int t, int k) __attribute__((always_inline)) {
                const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
#pragma unroll
                for (int m = 0; m < ML; ++m)
                    dst[m] = *reinterpret_cast<const float4*>(abase + (((m + dz) * HY + dy) * HX + dx) * 64 + sl[dx][k]);
            };
            for (int ch = 0; ch < nchunks; ++ch) {
                OAI_STAMP(0);
                // every wave is done reading the previous chunk.  A bare barrier behind lgkmcnt(0): __syncthreads() would also wait (vmcnt(0)) for
                // the weight fragments of the next tap, requested a moment ago
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                OAI_STAMP(1);
                if constexpr (M16) {
                    // the weight fragments of this chunk's step 0, requested in front of the halo staging: they land under the same wait.  They
                    // are NOT requested during the previous chunk's last step: an inline-asm load is invisible to the compiler -- it believes
                    // the result register holds the value from the asm statement on -- so a load in flight across the loop's back edge, the
                    // dispatch over the live-slice variants or the loop exit can be COPIED (register re-assignment on an edge) or its register
                    // re-used before the data has arrived.  Found with blocks of 1 and 3 live slices: their loop pre-headers copied the
                    // prologue's in-flight fragments (v_mov of stale registers), intermittently wrong results at small tile levels.  Every asm
                    // load of this kernel is now requested AND waited for inside one basic block.
                    if constexpr (!FIRST) { m16_req_y(); m16_req_lo(); m16_req_hi(); }
                }
                if constexpr (FIRST) stage_first(ch);
                else stage(ch);
                if constexpr (M16 && FIRST) { m16_req_y(); m16_req_lo(); m16_req_hi(); }      // (behind ec0's arithmetic, which has branches: same basic block as the wait below)
                OAI_STAMP(2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's DMA pieces have landed ...
                OAI_STAMP(3);
                __syncthreads();                                             // ... and everybody else's
                OAI_STAMP(4);
                if constexpr (M16) {
                    // ---- 14 steps of tap pairs on v_mfma_f32_16x16x32_f16 (see the kernel's header comment).  Vector-memory order per step:
                    // Y'(j+1) behind pass B | X' lo(j+1) behind pass C lo | X' hi(j+1) behind pass C hi; the counted waits leave exactly the
                    // younger requests in flight.  (The fragments of this chunk's step 0 were requested at the chunk's top, above, and have
                    // landed behind the staging wait: vmcnt(0) is global.  Invariant: every asm load is requested AND waited for inside one basic block.)
                    constexpr int SLB = HY * HX * 64;                           // bytes between the z slices of the halo box
                    constexpr int POFF = (RX >= 32 ? 16 : (16 / RX) * HX) * 64;  // ... between rows 0-15 and rows 16-31 of the wave's tile
                    auto tapc = [](int q) constexpr { return ((q / 3) * HY + q % 3) * HX * 64; };      // tap (dz, dy) = q, dx 0
                    // this lane's record slot per step: lane group g = lane >> 4 reads channel half g & 1 of the pair's tap g >> 1
                    int lq = lane;
                    asm volatile("" : "+v"(lq));                                 // (recomputed every chunk: not held -- or spilled -- across the staging)
                    const int r16 = lq & 15, hsel = (lq >> 4) & 1, tsel = lq >> 5;
                    const int lx0 = wx * RX + r16 % RX, ly0 = wy * RY + r16 / RX;
                    const unsigned vbase = (unsigned)((ly0 * HX + lx0) * 64);
                    unsigned oA[2], oB[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        oA[k] = vbase + (unsigned)(tsel * 64 + (((k * 2 + hsel) ^ (((lx0 + tsel) >> 2) & 3)) << 4));       // steps 0..8: dx = tsel
                        oB[k] = vbase + (unsigned)(128 + (((k * 2 + hsel) ^ (((lx0 + 2) >> 2) & 3)) << 4));                // steps 9..12: dx = 2
                    }
                    const unsigned oC = vbase + (unsigned)(128 + (((tsel * 2 + hsel) ^ (((lx0 + 2) >> 2) & 3)) << 4));     // step 13: term = tsel
                    const unsigned tHX = (unsigned)(tsel * HX * 64);             // the second tap of a dx = 2 pair: one row (steps 9..11) / one slice (step 12) further
                    auto a_off = [&](int j, int k) __attribute__((always_inline)) -> unsigned {
                        if (j < 9) return oA[k] + (unsigned)tapc(j);
                        if (j < 12) return oB[k] + tHX + (unsigned)tapc(3 * (j - 9));
                        if (j == 12) return oB[k] + tHX * (unsigned)HY + (unsigned)tapc(2);
                        return oC + (unsigned)tapc(8);
                    };
                    auto lda = [&](unsigned off, int m, int p) __attribute__((always_inline)) {
                        return *reinterpret_cast<const float4*>(lds + off + m * SLB + p * POFF);
                    };
                    auto mma = [&](const float4& av, const f32x4m& bv, f32x4m& c) __attribute__((always_inline)) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m, av), __builtin_bit_cast(f16x8m, bv), c, 0, 0, 0);
                    };
                    // the staging wait was vmcnt(0): the fragments of step 0 are in their registers (tie them behind it)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bY[0]), "+v"(bY[1]), "+v"(bY[2]), "+v"(bY[3]) :: "memory");
                    asm volatile("" : "+v"(bXlo[0]), "+v"(bXlo[1]), "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                    float4 af[MREP][2];                                          // [m][p]: the A fragments of the running pass
#pragma unroll
                    for (int m = 0; m < ML; ++m)
#pragma unroll
                        for (int p = 0; p < 2; ++p) af[m][p] = lda(a_off(0, 0), m, p);
                    // one step, J a compile-time constant: the 14 steps are 14 explicit calls (a `#pragma unroll` loop is a request -- hipcc left the
                    // ML = 4 body rolled, with run-time tests of j, i.e. branches while fragment loads are in flight: tests/test_abi_cpu.py (g))
                    auto m16_step = [&](auto jtag) __attribute__((always_inline)) {
                        constexpr int j = decltype(jtag)::value;
                        if (j > 0) asm volatile("s_waitcnt vmcnt(4)" : "+v"(bY[0]), "+v"(bY[1]), "+v"(bY[2]), "+v"(bY[3]) :: "memory");      // Y'(j) has landed; younger: lo(j), hi(j)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < ML; ++m)                              // pass B: a0 . Y'   (step 13: [a0 | a1] . [b1 | 0])
#pragma unroll
                            for (int p = 0; p < 2; ++p)
#pragma unroll
                                for (int n = 0; n < 4; ++n) mma(af[m][p], bY[n], acc4[m][n >> 1][p * 2 + (n & 1)]);
                        __builtin_amdgcn_sched_barrier(0);
                        wp16 += STEP16;
                        if (j < 13) m16_req_y();                                  // Y' of the next step (step 13 requests nothing: see the chunk's top)
                        if (j > 0 && j < 13) asm volatile("s_waitcnt vmcnt(6)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");      // lo(j); younger: hi(j), Y'(j + 1)
                        if (j == 13) asm volatile("s_waitcnt vmcnt(2)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");              // (step 13: only hi(13) is younger)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < ML; ++m)                              // pass A, low couts: a0 . X'[0, 1] over all slices
#pragma unroll
                            for (int p = 0; p < 2; ++p)
#pragma unroll
                                for (int n = 0; n < 2; ++n) mma(af[m][p], bXlo[n], acc4[m][0][p * 2 + n]);
                        __builtin_amdgcn_sched_barrier(0);                        // (the wait below must not rise above these MFMAs: they are its lead)
                        if (j > 0 && j < 13) asm volatile("s_waitcnt vmcnt(4)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");      // hi(j); younger: Y'(j + 1)
                        if (j == 13) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");              // (step 13: nothing is younger)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < ML; ++m) {                            // pass A, high couts; behind each slice the a1 fragments of pass C take its registers
#pragma unroll
                            for (int p = 0; p < 2; ++p)
#pragma unroll
                                for (int n = 0; n < 2; ++n) mma(af[m][p], bXhi[n], acc4[m][1][p * 2 + n]);
                            __builtin_amdgcn_sched_barrier(0);
                            if (j < 13) {
#pragma unroll
                                for (int p = 0; p < 2; ++p) af[m][p] = lda(a_off(j, 1), m, p);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (j < 13) {
#pragma unroll
                            for (int m = 0; m < ML; ++m)                          // pass C, low couts: a1 . X'[0, 1]
#pragma unroll
                                for (int p = 0; p < 2; ++p)
#pragma unroll
                                    for (int n = 0; n < 2; ++n) mma(af[m][p], bXlo[n], acc4[m][0][p * 2 + n]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (j < 13) m16_req_lo();                                 // the low couts of the next step's X'
                        __builtin_amdgcn_sched_barrier(0);
                        if (j < 13) {
#pragma unroll
                            for (int m = 0; m < ML; ++m) {                        // pass C, high couts; behind each slice the a0 fragments of the next step
#pragma unroll
                                for (int p = 0; p < 2; ++p)
#pragma unroll
                                    for (int n = 0; n < 2; ++n) mma(af[m][p], bXhi[n], acc4[m][1][p * 2 + n]);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int p = 0; p < 2; ++p) af[m][p] = lda(a_off(j + 1, 0), m, p);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if (j < 13) m16_req_hi();                                 // the high couts of the next step's X'
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    m16_step(std::integral_constant<int, 0>{}); m16_step(std::integral_constant<int, 1>{}); m16_step(std::integral_constant<int, 2>{});
                    m16_step(std::integral_constant<int, 3>{}); m16_step(std::integral_constant<int, 4>{}); m16_step(std::integral_constant<int, 5>{});
                    m16_step(std::integral_constant<int, 6>{}); m16_step(std::integral_constant<int, 7>{}); m16_step(std::integral_constant<int, 8>{});
                    m16_step(std::integral_constant<int, 9>{}); m16_step(std::integral_constant<int, 10>{}); m16_step(std::integral_constant<int, 11>{});
                    m16_step(std::integral_constant<int, 12>{}); m16_step(std::integral_constant<int, 13>{});
                    OAI_STAMP(5);
                } else {                                                     // (discarded for M16: its accumulators have another type)
                if constexpr (BLDS) { if (ch == 0) read_b(bcur, 0); }           // slab 0 (landed with the halo, behind the barrier above)
                load_a(acur[0], 0, 0);
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    if constexpr (BLDS) {
                        // in flight: the slabs of taps t+1 (requested two taps ago) and t+2: the older one has landed -- for everybody behind
                        // the barrier, which also says that everybody has read slab t (slot t % 3) into registers: it is refilled with t+3
                        if (t > 0 || ch == 0) {          // (tap 0 stands right behind the chunk's own barrier -- except in chunk 0, where slab 0 has just been read from the slot refilled below)
                            asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                        issue_b(t % 3);
                        read_b(bnext, (t + 1) % 3);
                    }
                    load_a(acur[1], t, 1);
                    if constexpr (!BLDS) {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int n = 0; n < NREP; ++n) bnext[k][n] = wp[(k * NREP + n) * 64];
                        wp += STEP;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 2; ++p)                              // a0.b0, a0.b1
#pragma unroll
                        for (int m = 0; m < ML; ++m)
#pragma unroll
                            for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[0][m], bcur[p][n], acc[m][n]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < 27) load_a(acur[0], t + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < ML; ++m)                             // a1.b0
#pragma unroll
                        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[1][m], bcur[0][n], acc[m][n]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int n = 0; n < NREP; ++n) bcur[k][n] = bnext[k][n];
                }
                OAI_STAMP(5);
                }
            }
        };
        const int ml = m_lo == 0 ? m_hi : MREP;                         // workgroup-uniform
        if constexpr (FIRST) run_chunks(std::integral_constant<int, MREP>{});       // (one copy of the inlined ec0 code; dead slices are never stored)
        else if constexpr (MREP == 4) {
            if (ml == 4) run_chunks(std::integral_constant<int, 4>{});
            else if (ml == 3) run_chunks(std::integral_constant<int, 3>{});
            else if (ml == 2) run_chunks(std::integral_constant<int, 2>{});
            else run_chunks(std::integral_constant<int, 1>{});
        } else {
            if (ml == 2) run_chunks(std::integral_constant<int, 2>{});
            else run_chunks(std::integral_constant<int, 1>{});
        }
    }

    // (M16: nothing is in flight here -- step 13 requests nothing and waits for everything it uses)
    // ---- epilogue: relu(acc*scale + shift), split once, stored as format S.  The C/D layout gives a lane ONE channel of 16
    // voxels, i.e. 4-byte pieces of 64-byte records; stored directly that is 128 dword stores per wave with 64-bit address
    // arithmetic each (13% of the whole segmentation, profiles/r01_ablation.md).  Instead the block's output image is built in
    // the (now idle) halo buffer, 128 B per voxel and cout half, and copied out 16 B per lane: 32 full-width stores per wave.
    OAI_STAMPB(7);
    constexpr int TV = TZ * kTY * kTX;                                // voxels of the block
    const unsigned seen = census_peek(a.census), seen_first = FIRST ? census_peek(a.first_census) : 0u;      // (here: waited for under the epilogue, short live range)
    constexpr int EIT = TV * 8 / 256;                                 // 16-byte pieces per thread and cout half
    static_assert(TV * 128 <= BUF && (TV * 8) % 256 == 0, "output image must fit the halo buffer");
    static_assert((kTX & (kTX - 1)) == 0 && (kTY & (kTY - 1)) == 0, "power-of-two block");
    unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
    const int nco = (a.Cout + 15) / 16;                               // chunks of the output tensor
    float vmax = 0.0f;
    const bool head = !FIRST && a.head_w != nullptr;                  // uniform; host guarantees ncb == 1 and head_ncls <= 4
    typedef const __attribute__((address_space(4))) float* cfptr;     // wave-uniform reads -> scalar loads (see stage_first)
    const cfptr head_w = (cfptr)a.head_w, head_b = (cfptr)a.head_b;     // (head_b as a per-lane load sat, with its vmcnt(0), in front of every head store)
    constexpr int HV = TV / 256;                                      // block voxels per thread in the fused dc0
    float hacc[HV][4];
#pragma unroll
    for (int j = 0; j < HV; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) hacc[j][k] = 0.0f;
    // this lane's column scales / shifts of both cout halves, loaded and WAITED FOR once, here: loaded inside the loop, the second half's
    // pair is waited for with vmcnt(0) behind the first half's copy-out stores, i.e. until those have reached memory
    float scv[NREP][M16 ? 2 : 1], shv[NREP][M16 ? 2 : 1];            // M16: this lane's columns 16 q + (lane & 15), q = 0, 1, of the 32-cout half n
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int q = 0; q < (M16 ? 2 : 1); ++q) {
            const int co = cb * 64 + n * 32 + (M16 ? q * 16 + col16 : row);
            scv[n][q] = co < a.Cout ? a.scale[co] : 0.0f; shv[n][q] = co < a.Cout ? a.shift[co] : 0.0f;
            asm volatile("" : "+v"(scv[n][q]), "+v"(shv[n][q]));
        }
    // what the copy-out writes: the tile's box, cut down to what the consumer of this tensor reads (ConvArgs::store_boxes)
    int clo[3], chi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { clo[i] = blo[i]; chi[i] = bhi[i]; }
    if (a.store_boxes) {
        const int* sb = a.store_boxes + 6 * tile;
#pragma unroll
        for (int i = 0; i < 3; ++i) { clo[i] = max(clo[i], sb[i] - a.store_grow); chi[i] = min(chi[i], sb[3 + i] + a.store_grow); }
    }
    // Which of this lane's 16 C/D rows (voxels of one z slice) lie inside the tile's box in y and x: ONE mask register, made once per block; the z
    // and cout tests are uniform / per half and select the whole mask.  (As compares inside the loops below the same tests were SGPR-pair mask
    // operations per value, enough of them live at once that SGPRs were spilled through v_writelane / v_readlane; conv3_wino_sres, round 4.)
    unsigned okmask = 0;
    {
        const unsigned ylen = (unsigned)(bhi[1] - blo[1]), xlen = (unsigned)(bhi[2] - blo[2]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = row_of(r);
            const int tx = wx * RX + rr % RX, ty = wy * RY + rr / RX;
            okmask |= ((unsigned)(oy0 + ty - blo[1]) < ylen && (unsigned)(ox0 + tx - blo[2]) < xlen ? 1u : 0u) << r;
        }
    }
    const float relu_floor = a.relu ? 0.0f : -__builtin_inff();
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int co = cb * 64 + n * 32 + row;                        // (32x32 form: this lane's cout)
        // padded channels of the last chunk are written as 0.  M16: a lane holds the columns 16 q + col16 of both records q of the half; nco * 16 is a
        // multiple of 16, so record q is valid or not as a whole -- the record-1 test selects per element below
        const bool cvalid = M16 ? cb * 64 + n * 32 < nco * 16 : co < nco * 16;
        const bool cvalid1 = cb * 64 + n * 32 + 16 < nco * 16;        // M16: the half's second record
        const bool odd = (M16 ? col16 : row) & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;
        unsigned char* lrow = lds + (M16 ? 0 : (row >> 4) * 64) + (((M16 ? col16 : row) & 15) >> 1) * 4;
        unsigned omn = cvalid ? okmask : 0u;
        if constexpr (M16) { if (!cvalid1) omn &= 0x0F0Fu; }          // elements r with (r >> 2) & 1 == 1 belong to record 1
        asm volatile("" : "+v"(omn));                                 // (per half: no compare masks carried from the first half to the second)
        __syncthreads();                                              // halo reads / the previous half's copy-out are done
        OAI_STAMPB(n == 0 ? 0 : 2);
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int oz = oz0 + m;
            const bool zok = oz >= blo[0] && oz < bhi[0];
            const int om = zok ? (int)omn : 0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                          // C/D rows r, r+1 = x-adjacent voxels
                float v[2];
                int vox[2];
                const float sc = scv[n][M16 ? (r >> 2) & 1 : 0], sh = shv[n][M16 ? (r >> 2) & 1 : 0];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rr = row_of(r + e);
                    const int tx = wx * RX + rr % RX, ty = wy * RY + rr / RX;
                    const float x = fmaxf(acc_el(m, n, r + e) * sc + sh, relu_floor);
                    // voxels outside the box are never copied out: 0 (bitwise AND with the sign-extended mask bit: v_bfe_i32, no compare, no SGPR mask)
                    v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & (unsigned)__builtin_amdgcn_sbfe(om, r + e, 1));
                    vox[e] = (m * kTY + ty) * kTX + tx;
                }
                vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1])));  // fp16 range guard (checked once, below)
                unsigned w_hi, w_lo;
                split_two_voxels(v[0], v[1], sel, w_hi, w_lo);
                unsigned char* dst = lrow + (odd ? vox[1] : vox[0]) * 128 + (M16 ? ((r >> 2) & 1) * 64 : 0);      // (M16: record q = (r >> 2) & 1 of the half)
                *reinterpret_cast<unsigned*>(dst) = w_hi;
                *reinterpret_cast<unsigned*>(dst + 32) = w_lo;
            }
        }
        OAI_STAMPB(1);
        __syncthreads();
        OAI_STAMPB(2);
        if (head) {
            // dc0 on this half's 32 channels, read back from the image exactly as head_sres_kernel reads format S from
            // memory (same join, same channel order: bit-identical logits), accumulated across the two halves
#pragma unroll
            for (int j = 0; j < HV; ++j) {
                const unsigned char* rec = lds + (j * 256 + tid) * 128;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u16x4 hi = *reinterpret_cast<const u16x4*>(rec + ch * 64 + q * 8), lo = *reinterpret_cast<const u16x4*>(rec + ch * 64 + 32 + q * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = n * 32 + ch * 16 + 4 * q + e;
                            if (c < a.Cout) {
                                const float xv = join2_f16(hi[e], lo[e]);
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (k < a.head_ncls) hacc[j][k] = fmaf(xv, head_w[k * a.Cout + c], hacc[j][k]);
                            }
                        }
                    }
            }
        } else if (FIRST && a.sc_boxes) {
            // Shared encoder pass: this block of the padded volume lies in the read region of up to 2 x 2 x 2 overlapping tiles; its image is
            // copied into each of their per-tile tensors (ConvArgs::sc_boxes), so that the consumer addresses an ordinary per-tile tensor.
            static_assert(!FIRST || kTX == 16, "lane split of the scatter copy-out");
            const int t5 = tid >> 3, q = tid & 7;
            const int x_lane = t5 % kTX, y_lane = t5 / kTX;
            const bool cok = cb * 4 + n * 2 + (q >> 2) < nco;
            const size_t tplane = (size_t)a.sc_t[0] * a.sc_t[1] * a.sc_t[2];
            for (int iz = oz0 / a.sc_e[0]; iz >= 0 && oz0 - iz * a.sc_e[0] < a.sc_t[0]; --iz) {
                if (iz >= a.sc_g[0]) continue;
                for (int iy = oy0 / a.sc_e[1]; iy >= 0 && oy0 - iy * a.sc_e[1] < a.sc_t[1]; --iy) {
                    if (iy >= a.sc_g[1]) continue;
                    for (int ix = ox0 / a.sc_e[2]; ix >= 0 && ox0 - ix * a.sc_e[2] < a.sc_t[2]; --ix) {
                        if (ix >= a.sc_g[2]) continue;
                        const int t = (iz * a.sc_g[1] + iy) * a.sc_g[2] + ix - a.sc_tile0;
                        if (t < 0 || t >= a.sc_ntiles) continue;
                        const int lz0 = oz0 - iz * a.sc_e[0], ly0 = oy0 - iy * a.sc_e[1], lx0 = ox0 - ix * a.sc_e[2];      // block origin in tile coordinates
                        // (constant address space: the box is read by SCALAR loads -- lgkmcnt.  As vector loads each candidate tile's box was waited for
                        // with vmcnt(0), i.e. behind the round trip of every copy-out store issued so far: stamps had 16 % of ec1 in this loop)
                        typedef const __attribute__((address_space(4))) int* ciptr;
                        const ciptr sb = (ciptr)(a.sc_boxes + 6 * t);
                        const int s0z = sb[0] - 1, s0y = sb[1] - 1, s0x = sb[2] - 1, s1z = sb[3] + 1, s1y = sb[4] + 1, s1x = sb[5] + 1;
                        if (sb[3] <= sb[0] || lz0 >= s1z || lz0 + TZ <= s0z || ly0 >= s1y || ly0 + kTY <= s0y || lx0 >= s1x || lx0 + kTX <= s0x) continue;
                        unsigned char* ob = outb + ((size_t)t * nco + cb * 4 + n * 2) * tplane * 64;                         // wave-uniform
                        const long long lane_off = (long long)(((size_t)(q >> 2) * tplane + (size_t)y_lane * a.sc_t[2] + x_lane) * 64 + (q & 3) * 16);
#pragma unroll
                        for (int it = 0; it < EIT; ++it) {
                            const int c = (it & 3) * 32, zc = it >> 2, yc = c / kTX;
                            const int lz = lz0 + zc, ly = ly0 + y_lane + yc, lx = lx0 + x_lane;
                            if (cok && lz >= s0z && lz < s1z && ly >= s0y && ly < s1y && lx >= s0x && lx < s1x &&
                                (unsigned)lz < (unsigned)a.sc_t[0] && (unsigned)ly < (unsigned)a.sc_t[1] && (unsigned)lx < (unsigned)a.sc_t[2]) {
                                const long long uni = ((long long)(lz * a.sc_t[1] + ly0 + yc) * a.sc_t[2] + lx0) * 64;           // wave-uniform
                                float4* dstp = reinterpret_cast<float4*>(ob + (lane_off + uni));
                                const float4 val = *reinterpret_cast<const float4*>(lds + (it * 256 + tid) * 16);
                                __builtin_nontemporal_store(val.x, &dstp->x); __builtin_nontemporal_store(val.y, &dstp->y);
                                __builtin_nontemporal_store(val.z, &dstp->z); __builtin_nontemporal_store(val.w, &dstp->w);
                            }
                        }
                    }
                }
            }
        } else {
            // Piece sidx = it*256 + tid of the image = 16 bytes (q & 3) of the record (chunk q >> 2) of block voxel w = it*32 + (tid >> 3).
            // kTX * kTY = 128, so z = it >> 2 and the (y, x) of w split into a per-LANE part (from tid >> 3 < 32) and a per-ITERATION part (from
            // (it & 3) * 32): the destination is one per-lane 32-bit offset (computed once) + a wave-uniform offset per iteration from a
            // wave-uniform base.  The first version computed a 64-bit address per iteration; hipcc hoisted all sixteen above the image build,
            // spilled some, and reloaded them inside this loop with `scratch_load ; s_waitcnt vmcnt(0)` -- i.e. behind ALL earlier copy-out
            // stores of the block: the stores went to memory one reload at a time.
            static_assert(kTX * kTY == 128, "the index split below");
            constexpr bool kWide = kTX >= 32;
            const int t5 = tid >> 3, q = tid & 7;
            const int x_lane = kWide ? t5 : t5 % kTX, y_lane = kWide ? 0 : t5 / kTX;
            const bool cok = cb * 4 + n * 2 + (q >> 2) < nco;
            unsigned char* ob = outb + ((size_t)tile * nco + cb * 4 + n * 2) * plane * 64;                        // wave-uniform
            const unsigned lane_off = (unsigned)(((size_t)(q >> 2) * plane + (size_t)y_lane * a.W + x_lane) * 64 + (q & 3) * 16);   // 2 * plane * 64 < 2^32 (host check)
            const int oyl = oy0 + y_lane, oxl = ox0 + x_lane;
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                constexpr int kDummy = 0; (void)kDummy;
                const int c = (it & 3) * 32, zc = it >> 2;
                const int xc = kWide ? c % kTX : 0, yc = c / kTX;
                const int oz = oz0 + zc, oy = oyl + yc, ox = oxl + xc;
                if (cok && oz >= clo[0] && oz < chi[0] && oy >= clo[1] && oy < chi[1] && ox >= clo[2] && ox < chi[2]) {
                    const unsigned uni = (unsigned)(((oz * a.H + oy0 + yc) * a.W + ox0 + xc) * 64);                // wave-uniform
                    float4* dstp = reinterpret_cast<float4*>(ob + (size_t)(lane_off + uni));
                    const float4 val = *reinterpret_cast<const float4*>(lds + (it * 256 + tid) * 16);
                    // one global_store_dwordx4 ... nt: the tensor (GBs per layer) is next read by another launch, from HBM either way (-0.9 %)
                    __builtin_nontemporal_store(val.x, &dstp->x); __builtin_nontemporal_store(val.y, &dstp->y);
                    __builtin_nontemporal_store(val.z, &dstp->z); __builtin_nontemporal_store(val.w, &dstp->w);
                }
            }
        }
        OAI_STAMPB(head ? 4 : 3);
        if constexpr (RX == 16 && RY == 2 && WY == 4 && WX == 1) {
            if (a.pool_out) {            // MaxPool3d(2) fused (see pooled_store in unet_kernels.h), result also in format S
                unsigned char* pb = reinterpret_cast<unsigned char*>(a.pool_out);
                const int Dp = a.D / 2, Hp = a.H / 2, Wp = a.W / 2;
                const bool relu = a.relu != 0;
                // a 2 x 2 x 2 window = elements r0, r0 + 1 (x pair), r0 + 8, r0 + 9 (the next y row) of slices m, m + 1 -- in both accumulator layouts.
                // 32x32: g = the x group (rows 0..3 / 8..11), one cout per lane; M16: g = the 16-cout record q of the half, x from the row quarter
                auto val = [&](int m, int r) {
                    const float v = acc_el(m, n, r) * scv[n][M16 ? (r >> 2) & 1 : 0] + shv[n][M16 ? (r >> 2) & 1 : 0];
                    return relu ? fmaxf(v, 0.0f) : v;
                };
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int r0 = 4 * g + 2 * p;
#pragma unroll
                        for (int m = 0; m < MREP; m += 2) {
                            float v = fmaxf(fmaxf(val(m, r0), val(m, r0 + 1)), fmaxf(val(m, r0 + 8), val(m, r0 + 9)));
                            v = fmaxf(v, fmaxf(fmaxf(val(m + 1, r0), val(m + 1, r0 + 1)), fmaxf(val(m + 1, r0 + 8), val(m + 1, r0 + 9))));
                            const int x = M16 ? ox0 + 4 * rq16 + 2 * p : ox0 + 8 * g + 4 * half + 2 * p, y = oy0 + 2 * wy, z = oz0 + m;
                            const int pco = M16 ? cb * 64 + n * 32 + 16 * g + col16 : co;
                            store_split_pair(pb + srec(tile, nco, (size_t)Dp * Hp * Wp, 0, (((size_t)(z / 2)) * Hp + y / 2) * Wp + x / 2), (size_t)Dp * Hp * Wp * 64,
                                             pco, v, M16 ? (g ? cvalid1 : cvalid) : cvalid, a.range_flag);
                        }
                    }
            }
        }
    }
    OAI_STAMPB(5);
    if (head) {
        int hlo[3], hhi[3];
        tile_box(a.head_boxes, tile, blo, bhi, hlo, hhi);
        const size_t evox = (size_t)a.head_e[0] * a.head_e[1] * a.head_e[2];
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            const int vox = j * 256 + tid;
            const int oz = oz0 + vox / (kTX * kTY), oy = oy0 + (vox / kTX) % kTY, ox = ox0 + vox % kTX;
            if (oz >= hlo[0] && oz < hhi[0] && oy >= hlo[1] && oy < hhi[1] && ox >= hlo[2] && ox < hhi[2]) {
                const size_t o = ((size_t)(oz - a.head_k[0]) * a.head_e[1] + (oy - a.head_k[1])) * a.head_e[2] + (ox - a.head_k[2]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < a.head_ncls) {
                        const float l = hacc[j][k] + head_b[k];
                        float r = l;
                        if (a.head_mode != 2) {
                            const float pr = 1.0f / (1.0f + expf(-l));
                            r = a.head_mode == 1 ? (pr > 0.5f ? 1.0f : 0.0f) : pr;
                        }
                        a.head_out[((size_t)tile * a.head_ncls + k) * evox + o] = r;
                    }
            }
        }
    }
    if constexpr (FIRST) census_note(a.first_census, a.range_flag, first_max, seen_first);
    census_note(a.census, a.range_flag, vmax, seen);
#ifdef OAI_DIAG
    OAI_STAMP(6);
    OAI_STAMPB(6);
    if (a.stamps && lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, (unsigned long long)st_sum[i]);
        atomicAdd(a.stamps + 8, 1ull);
        atomicAdd(a.stamps + 9, (unsigned long long)st_chunks);
    }
#endif
#undef OAI_STAMP
#undef OAI_STAMPB
#undef OAI_STAMP_
}

// ---- k2s2 up-conv, split-resident in and out ------------------------------------------------------------------------------
// GEMM [voxels x Cin] x [Cin x 8*Cout] (column = parity * Cout + co), A rows read straight from format S, B from the
// pre-split panel of pack_up_panel_f16.  The first version gave a workgroup 64 voxels x 256 columns with every wave
// re-reading all 64 A rows: 37 GB of L2->L1 traffic per 32-tile pass for 5 GB of output, i.e. bound by L2 bandwidth
// (dc9: 11.5 TB/s) at 0.7-1.6 TB/s of writes.  Now a workgroup owns 128 voxels x 256 columns as 2 x 2 waves of
// 64 voxels x 128 columns (MREP 2, NREP 4): each A row and each B fragment is fetched by two waves of the same
// workgroup at the same time (one L2 request), 17.6 GB per pass.
// Operands go through a three-stage LDS ring filled by LDS-DMA (per 16-deep k step: 128 A records = 8 KB, swizzled like the
// conv halo, + the four 4-KiB weight slabs of the workgroup's 256 columns, already in fragment order), two steps ahead of the
// MFMAs with counted vmcnt waits and ONE barrier per step: with direct register loads each of the 8-32 k steps waited a
// full L2 round trip with only two waves per SIMD to hide it (1.9 TB/s of writes on dc3).
__global__ void __launch_bounds__(256, 2) upconv2_igemm_sres(const UpArgs a) {
#ifdef OAI_DIAG
    // phase stamps (scripts/stamp_phases.py): [16] prologue, [17] k loop, [18] epilogue barriers, [19] split + LDS image, [20] copy-out stores, [24] waves
    unsigned ust[5] = {0, 0, 0, 0, 0};
    unsigned long long ulast = __builtin_amdgcn_s_memtime();
#define OAI_USTAMP(i) do { if (a.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ust[i] += (unsigned)(now_ - ulast); ulast = now_; } } while (0)
#else
#define OAI_USTAMP(i) do { } while (0)
#endif
    constexpr int kStage = 24 * 1024, kRing = 3 * kStage;                            // [A 8 KB | B 4 x 4 KB] per stage
    __shared__ __attribute__((aligned(16))) unsigned char ulds[kRing + 512];          // ring (the 64-KB epilogue image reuses it) + voxel table
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int id = a.xcd_group ? xcd_block_id(a.nblocks, a.xcd_group) : (int)blockIdx.x;
    if (id < 0) return;
    // (round 6) a workgroup walks `nbw` column blocks of its 128 voxels one after the other: the voxel table, the A-row plan and the workgroup launch
    // itself -- a quarter of dc3's time with one column block per workgroup (profiles/r02_conv_per_layer.md: "skeleton") -- are paid once per `nbw`,
    // and the A rows of the later column blocks come out of a hot L2.  Same arithmetic per (voxel, column): bit-identical.
    const int nbw = a.nbw > 0 ? a.nbw : 1, nnbg = (a.nnb + nbw - 1) / nbw;
    const int nbg = id % nnbg; id /= nnbg;
    const int mb = id % a.nmb; id /= a.nmb;
    const int tile = id;
    const int N = 8 * a.Cout;
    const int row = lane & 31, half = lane >> 5;
    const int rz = a.hi[0] - a.lo[0], ry = a.hi[1] - a.lo[1], rx = a.hi[2] - a.lo[2];
    const int nvox = rz * ry * rx;
    int blo[3], bhi[3];
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    {
        const int zf = a.lo[0] + (mb * 128) / (rx * ry), zl = a.lo[0] + min(mb * 128 + 127, nvox - 1) / (rx * ry);
        if (zl < blo[0] || zf >= bhi[0]) return;
    }
    const size_t plane = (size_t)a.D * a.H * a.W;
    const int nks = (a.Cin + 15) / 16;
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    const int nco = (a.Cout + 15) / 16;
    unsigned* vtab = reinterpret_cast<unsigned*>(ulds + kRing);        // [128 block voxels]: output voxel index of parity 0, or ~0u
    if (tid < 128) {
        const int v = mb * 128 + tid;
        unsigned e = ~0u;
        if (v < nvox) {
            const int x = v % rx, t = v / rx, y = t % ry, z = t / ry;
            const int iz = a.lo[0] + z, iy = a.lo[1] + y, ix = a.lo[2] + x;
            if (iz >= blo[0] && iz < bhi[0] && iy >= blo[1] && iy < bhi[1] && ix >= blo[2] && ix < bhi[2])
                e = (unsigned)(((2 * iz) * Ho + 2 * iy) * Wo + 2 * ix);
        }
        vtab[tid] = e;
    }
    // ---- DMA plan.  A: pieces p = it*256 + tid (it = 0,1) = slot (p & 3) of block voxel p >> 2; the slot holds logical slot
    // (p & 3) ^ key(voxel) (the DMA writes lane-linearly, so the swizzle is applied to the source address).  B: piece tid of
    // each of the four 64-column groups; a group's k-step slab [term][nr][lane] is 4 KiB contiguous in the panel.
    const unsigned char* asrc[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int p = it * 256 + tid, vl = p >> 2, v = mb * 128 + vl;
        if (v < nvox) {
            const int x = v % rx, y = (v / rx) % ry, z = v / (rx * ry);
            asrc[it] = reinterpret_cast<const unsigned char*>(a.src) +
                       srec(tile, nks, plane, 0, ((size_t)(a.lo[0] + z) * a.H + (a.lo[1] + y)) * a.W + (a.lo[2] + x)) + (((p & 3) ^ ((vl >> 2) & 3)) << 4);
        } else asrc[it] = nullptr;
    }
    unsigned char* outb = reinterpret_cast<unsigned char*>(a.out) + srec(tile, nco, 8 * plane, 0, 0);      // chunk c of output voxel v at + (c * 8 plane + v) * 64
    const unsigned seen = census_peek(a.census);
    unsigned umax = 0;
    f32x16 acc[2][4];
    bool looped = true;
    int nb = nbg * nbw, ncol0 = nb * 256 + wn * 128;
    for (int nbi = 0; nbi < nbw && nb < a.nnb; ++nbi, ++nb) {
    ncol0 = nb * 256 + wn * 128;
    if (nbi > 0) __syncthreads();                                           // everybody is done with the previous column block's image: the ring may be refilled
    const int ngroups = (N + 63) / 64;
    const unsigned char* bsrc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bsrc[g] = nb * 4 + g < ngroups ? reinterpret_cast<const unsigned char*>(a.wpanel + (size_t)(nb * 4 + g) * nks * 4 * 64) + tid * 16 : nullptr;
    const unsigned ring_addr = lds_addr_of(ulds);
    auto issue = [&](int st, int ks) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ring_addr + st * kStage + wave * 1024);
#pragma unroll
        for (int it = 0; it < 2; ++it)
            lds_dma16(asrc[it] ? asrc[it] + (size_t)ks * plane * 64 : a.zero, base + it * 4096);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            lds_dma16(bsrc[g] ? bsrc[g] + (size_t)ks * 4096 : a.zero, base + 8192 + g * 4096);
    };
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const bool active = ncol0 < N;                                        // (an idle wave still stages and copies out)
    int aoff[2][2];                                                       // [term][m]: this lane's A slot inside a stage
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int vl = wm * 64 + m * 32 + row;
#pragma unroll
        for (int k = 0; k < 2; ++k) aoff[k][m] = vl * 64 + (((k * 2 + half) ^ ((vl >> 2) & 3)) << 4);
    }
    const int boff = 8192 + wn * 8192 + lane * 16;                         // this wave's two column groups, fragment (k, nr) at + (k*2+nr)*1024
    OAI_USTAMP(0);
    issue(0, 0);
    if (nks > 1) issue(1, 1);
    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    for (int ks = 0; ks < nks; ++ks) {
        // stage ks has landed (the younger stage's six pieces may still fly) and this wave's LDS reads of stage ks-1 are done ...
        if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // ... for everybody (a bare barrier: __syncthreads() would drain vmcnt)
        asm volatile("" ::: "memory");
        if (ks + 2 < nks) issue((ks + 2) % 3, ks + 2);
        if (active) {
            const unsigned char* sp = ulds + (ks % 3) * kStage;
            float4 at[2][2], bf[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m) at[k][m] = *reinterpret_cast<const float4*>(sp + aoff[k][m]);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[k][n] = *reinterpret_cast<const float4*>(sp + boff + (n >> 1) * 4096 + (k * 2 + (n & 1)) * 1024);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = mfma_16bit<true>(at[PA[p]][m], bf[PB[p]][n], acc[m][n]);
        }
    }
    OAI_USTAMP(1);
    if (a.Cout % 16 == 0) {
        // ---- epilogue through LDS, one 64-voxel half (m) at a time: records [voxel 64][16 column chunks][64 B], copied out
        // 16 B per lane: every store instruction writes whole 64-byte records, 1 KiB per wave
        const int q = tid & 63;                                            // this thread's 16-byte piece of every voxel row
        const int cg = nb * 256 + (q >> 2) * 16;                             // first global column of its record
        const bool qok = cg < N;
        const int par = qok ? cg / a.Cout : 0, cchunk = qok ? (cg - par * a.Cout) >> 4 : 0;
        const unsigned poff = (unsigned)(((par >> 2) * Ho + ((par >> 1) & 1)) * Wo + (par & 1));
        const size_t inrow = (size_t)cchunk * 8 * plane * 64 + (q & 3) * 16;
        // this lane's four column scales / shifts, loaded ONCE here: inside the loops below hipcc waits for them with vmcnt(0), which in
        // the second half also drains the first half's sixteen copy-out stores all the way to memory
        float scn[4], shn[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = ncol0 + n * 32 + row;
            const bool cok = col < N;
            const int co = cok ? col % a.Cout : 0;
            scn[n] = cok ? a.scale[co] : 0.0f; shn[n] = cok ? a.shift[co] : 0.0f;
        }
        // a use on the unconditional path: the wait for the eight loads lands HERE (otherwise the wait-count pass, which merges the
        // skipped-branch paths of the first half, puts a vmcnt(0) at their first use in the second half -- behind the stores)
        asm volatile("" : "+v"(scn[0]), "+v"(scn[1]), "+v"(scn[2]), "+v"(scn[3]), "+v"(shn[0]), "+v"(shn[1]), "+v"(shn[2]), "+v"(shn[3]));
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            __syncthreads();                                                 // voxel table written / previous half copied out
            OAI_USTAMP(2);
            if (active) {
                unsigned vmask = 0;                                          // which of this lane's 16 voxel rows are real outputs
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    vmask |= (vtab[wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] != ~0u ? 1u : 0u) << r;
                const bool odd = row & 1;
                const unsigned sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int col = ncol0 + n * 32 + row;
                    const float sc = scn[n], sh = shn[n];
                    // (round 5: 1 700 VALU operations of this epilogue stood against 192 MFMAs of dc3's k loop; the per-value tests are gone -- the ReLU a plain max,
                    //  the row / column tests one mask word applied by v_bfe_i32 + v_and, the census maximum on bit patterns: 1 444 operations, same values bit
                    //  for bit -- and the up-convs' time did not move (same-box A/B 127.3-128.0 ms per pass either way): vector issue is not their bound either)
                    unsigned om = col < N ? vmask : 0u;
                    asm volatile("" : "+v"(om));
                    // this lane's image row of element (r, e = odd): voxel row (r + odd) & 3 + 8 (r >> 2) + 4 half -- r is even: + odd
                    unsigned char* lrow = ulds + ((wn * 128 + n * 32 + row) >> 4) * 64 + ((row & 15) >> 1) * 4 + (wm * 32 + 4 * half + (odd ? 1 : 0)) * 1024;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float v[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float x = fmaxf(acc[m][n][r + e] * sc + sh, 0.0f);                   // (ConvTranspose3d [+ BN] + ReLU: networks.py:94-107; launch_up launches no other form)
                            unsigned keep;                                     // 0 or ~0: rows outside the box read unwritten memory (as asm: the builtin is folded back into a compare + select)
                            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(om), "n"(r + e));
                            v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & keep);
                        }
                        // (behind the ReLU every value is >= +0: the largest |v| is the largest BIT PATTERN -- one v_max3_u32 per pair)
                        umax = max(umax, max(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1])));
                        unsigned w_hi, w_lo;
                        split_two_voxels(v[0], v[1], sel, w_hi, w_lo);
                        const int vl = (r & 3) + 8 * (r >> 2);
                        *reinterpret_cast<unsigned*>(lrow + vl * 1024) = w_hi;
                        *reinterpret_cast<unsigned*>(lrow + vl * 1024 + 32) = w_lo;
                    }
                }
            }
            OAI_USTAMP(3);
            __syncthreads();
            OAI_USTAMP(2);
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int vl = it * 4 + (tid >> 6);                          // image row: block voxel (vl >> 5) * 64 + m * 32 + (vl & 31)
                const unsigned e = vtab[(vl >> 5) * 64 + m * 32 + (vl & 31)];
                if (qok && e != ~0u)
                    *reinterpret_cast<float4*>(outb + (size_t)(e + poff) * 64 + inrow) = *reinterpret_cast<const float4*>(ulds + vl * 1024 + q * 16);
            }
            OAI_USTAMP(4);
        }
        continue;                                                             // the next column block of this workgroup
    }
    looped = false;                                                           // (narrow network: nbw is 1, the store path below runs once)
    break;
    }
    if (looped) {
        census_note(a.census, a.range_flag, __builtin_bit_cast(float, umax), seen);
#ifdef OAI_DIAG
        OAI_USTAMP(4);
        if (a.stamps && lane == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(a.stamps + 16 + i, (unsigned long long)ust[i]);
            atomicAdd(a.stamps + 24, 1ull);
        }
#endif
        return;
    }
    const bool active = ncol0 < N;
    // ---- narrow test networks (Cout not a multiple of 16): dword stores straight from the accumulators
    if (!active) return;                                                  // (wave-uniform)
    {   // range census in a pass of its own over the accumulators (tracked inside the store loop below it cost 58 VGPRs and 168 B of scratch)
        float nmax = 0.0f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = ncol0 + n * 32 + row;
            const bool cok = col < N;
            const int co = cok ? col % a.Cout : 0;
            const float sc = cok ? a.scale[co] : 0.0f, sh = cok ? a.shift[co] : 0.0f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned e = vtab[wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                    float val = acc[m][n][r] * sc + sh;
                    if (a.relu) val = fmaxf(val, 0.0f);
                    if (cok && e != ~0u) nmax = fmaxf(nmax, fabsf(val));
                }
        }
        census_note(a.census, a.range_flag, nmax);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = ncol0 + n * 32 + row;
        const bool cok = col < N;
        const int par = cok ? col / a.Cout : 0, co = cok ? col - par * a.Cout : 0;
        const float sc = cok ? a.scale[co] : 0.0f, sh = cok ? a.shift[co] : 0.0f;
        const unsigned poff = (unsigned)(((par >> 2) * Ho + ((par >> 1) & 1)) * Wo + (par & 1));
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned e = vtab[wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                float val = acc[m][n][r] * sc + sh;
                if (a.relu) val = fmaxf(val, 0.0f);
                // lanes of a pair (co even/odd) share parity and voxel because Cout is even and 32 | lane groups
                store_split_pair(outb + (size_t)(e == ~0u ? 0 : e + poff) * 64, 8 * plane * 64, co, val, cok && e != ~0u, a.range_flag);
            }
    }
}

// ---- ec0 (1 -> COUT, gather-fused) writing format S ---------------------------------------------------------------------------
template <int COUT>
__global__ void __launch_bounds__(256) conv3_first_sres_kernel(const TileSource s, const float* __restrict__ wk, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, unsigned char* __restrict__ out, int relu,
                                                               int* __restrict__ range_flag, unsigned* __restrict__ census, int shell) {
    // shell > 0: only voxels closer than `shell` to a face of the tile are computed (the rest of ec0 comes from the pass over the whole
    // volume, see run_batch).
    // Round 6: a workgroup WALKS its tile's voxel pairs with a grid stride (the host launches option "first_blocks" = 24 workgroups per tile), and the 36 inputs
    // of the NEXT pair are gathered while the current pair's 1 728 FMAs run.  The counters had shown the VALU 35 % busy at 2.5 waves per SIMD
    // (profiles/r06_ec1_shell.md): a wave's life was a gather round trip out of L2 / HBM, then its FMAs, then its stores, with too few waves per SIMD to
    // cover one another -- neither the store pattern, nor the instruction count, nor the LDS weight reads moved the time.  Same fmaf chains: bit-identical.
    __shared__ __attribute__((aligned(16))) float wl[27 * COUT];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) wl[i] = wk[i];
    __syncthreads();
    const size_t plane = (size_t)s.td * s.th * s.tw;
    const int local_tile = blockIdx.y;
    const int hw = s.tw >> 1;
    const int sh2 = 2 * shell, ps = (shell + 1) / 2;
    const size_t nA = shell > 0 ? (size_t)sh2 * s.th * hw : 0, nB = shell > 0 ? (size_t)(s.td - sh2) * sh2 * hw : 0;
    const size_t npairs = shell <= 0 ? plane / 2 : nA + nB + (size_t)(s.td - sh2) * (s.th - sh2) * 2 * ps;
    int tk = 0, tj = 0, ti = 0;
    if (s.vol) {
        const int t = s.tile_begin + local_tile;
        tk = t % s.gx; tj = (t / s.gx) % s.gy; ti = t / (s.gx * s.gy);
    }
    const float* const base = s.vol ? s.vol : s.tiles + (size_t)local_tile * plane;
    // voxel pair p -> (x even, y, z); shell mode: the pairs closer than `shell` to a face, enumerated slab by slab (see first_shell_pairs): two z slabs of
    // `shell` slices, between them two y slabs of `shell` rows, between those the first and last ceil(shell / 2) x pairs of every row
    auto decode = [&](size_t p, int& x, int& y, int& z) __attribute__((always_inline)) {
        if (shell <= 0) {
            x = 2 * (int)(p % hw); y = (int)((p / hw) % s.th); z = (int)(p / ((size_t)hw * s.th));
        } else if (p < nA) {
            const int zi = (int)(p / ((size_t)s.th * hw));
            z = zi < shell ? zi : s.td - sh2 + zi; y = (int)((p / hw) % s.th); x = 2 * (int)(p % hw);
        } else if (p < nA + nB) {
            const size_t q = p - nA;
            const int yi = (int)((q / hw) % sh2);
            z = shell + (int)(q / ((size_t)sh2 * hw)); y = yi < shell ? yi : s.th - sh2 + yi; x = 2 * (int)(q % hw);
        } else {
            const size_t q = p - nA - nB;
            const int xi = (int)(q % (2 * ps));
            z = shell + (int)(q / ((size_t)(s.th - sh2) * 2 * ps)); y = shell + (int)((q / (2 * ps)) % (s.th - sh2));
            x = xi < ps ? 2 * xi : s.tw - 2 * ps + 2 * (xi - ps);
        }
    };
    // the 3 x 3 x 4 inputs of the pair at (x, y, z): Partition's reflect-padded gather out of the volume, or the tile tensor; outside the tile: Conv3d's zeros
    auto gather = [&](int x, int y, int z, float (&inr)[9][4]) __attribute__((always_inline)) {
        int iz[3], iy[3], ix[4];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int zz = z + d - 1, yy = y + d - 1;
            if (s.vol) {
                iz[d] = (unsigned)zz < (unsigned)s.td ? reflect_index(ti * s.ez + zz - s.oz, s.D) * s.H * s.W : -1;
                iy[d] = (unsigned)yy < (unsigned)s.th ? reflect_index(tj * s.ey + yy - s.oy, s.H) * s.W : -1;
            } else {
                iz[d] = (unsigned)zz < (unsigned)s.td ? zz * s.th * s.tw : -1;
                iy[d] = (unsigned)yy < (unsigned)s.th ? yy * s.tw : -1;
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int xx = x + d - 1;
            if (s.vol) ix[d] = (unsigned)xx < (unsigned)s.tw ? reflect_index(tk * s.ex + xx - s.ox, s.W) : -1;
            else ix[d] = (unsigned)xx < (unsigned)s.tw ? xx : -1;
        }
#pragma unroll
        for (int zy = 0; zy < 9; ++zy) {
            const int a = iz[zy / 3], b = iy[zy % 3];
#pragma unroll
            for (int d = 0; d < 4; ++d) inr[zy][d] = (a | b | ix[d]) >= 0 ? base[(size_t)a + b + ix[d]] : 0.0f;
        }
    };
    float rmax = 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    int x = 0, y = 0, z = 0;
    float inn[9][4];
    if (p < npairs) { decode(p, x, y, z); gather(x, y, z, inn); }
    while (p < npairs) {
        float inr[9][4];
#pragma unroll
        for (int zy = 0; zy < 9; ++zy)
#pragma unroll
            for (int d = 0; d < 4; ++d) inr[zy][d] = inn[zy][d];
        const size_t v = ((size_t)z * s.th + y) * s.tw + x;
        p += stride;
        if (p < npairs) { decode(p, x, y, z); gather(x, y, z, inn); }      // the next pair's inputs: in flight under this pair's FMAs
        __builtin_amdgcn_sched_barrier(0);
        float acc[2][COUT];
#pragma unroll
        for (int j = 0; j < COUT; ++j) { acc[0][j] = 0.0f; acc[1][j] = 0.0f; }
#pragma unroll
        for (int zy = 0; zy < 9; ++zy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* w = &wl[(zy * 3 + dx) * COUT];
#pragma unroll
                for (int j = 0; j < COUT; j += 4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(w + j);
                    acc[0][j] = fmaf(inr[zy][dx], w4.x, acc[0][j]);         acc[1][j] = fmaf(inr[zy][dx + 1], w4.x, acc[1][j]);
                    acc[0][j + 1] = fmaf(inr[zy][dx], w4.y, acc[0][j + 1]); acc[1][j + 1] = fmaf(inr[zy][dx + 1], w4.y, acc[1][j + 1]);
                    acc[0][j + 2] = fmaf(inr[zy][dx], w4.z, acc[0][j + 2]); acc[1][j + 2] = fmaf(inr[zy][dx + 1], w4.z, acc[1][j + 2]);
                    acc[0][j + 3] = fmaf(inr[zy][dx], w4.w, acc[0][j + 3]); acc[1][j + 3] = fmaf(inr[zy][dx + 1], w4.w, acc[1][j + 3]);
                }
            }
        }
        constexpr int NCH = (COUT + 15) / 16;
#pragma unroll
        for (int vv = 0; vv < 2; ++vv) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                unsigned char* o = out + srec(local_tile, NCH, plane, ch, v + vv);
                u16x8 hi[2], lo[2];                                  // the whole 64-byte record in registers: four 16-byte stores
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = ch * 16 + j;
                    float r = 0.0f;
                    if (c < COUT) { r = acc[vv][c < COUT ? c : 0] * scale[c < COUT ? c : 0] + shift[c < COUT ? c : 0]; if (relu) r = fmaxf(r, 0.0f); }
                    rmax = fmaxf(rmax, fabsf(r));
                    unsigned l;
                    hi[j >> 3][j & 7] = (unsigned short)split2_f16(r, l);
                    lo[j >> 3][j & 7] = (unsigned short)l;
                }
                *reinterpret_cast<u16x8*>(o) = hi[0];
                *reinterpret_cast<u16x8*>(o + 16) = hi[1];
                *reinterpret_cast<u16x8*>(o + 32) = lo[0];
                *reinterpret_cast<u16x8*>(o + 48) = lo[1];
            }
        }
    }
    census_note(census, range_flag, rmax);
}

// ---- MaxPool3d(2) on format S (fallback when the pooling cannot ride in the conv epilogue) -----------------------------------
// shell_only: just the pooled voxels on the faces of the tile (their 2 x 2 x 2 windows are what the per-tile shell launches of ec1 wrote)
__global__ void __launch_bounds__(256) maxpool2_sres_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                            int D, int H, int W, int nch, size_t total /*out voxels * nch * 4*/, int shell_only) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        size_t rec = i >> 2;
        const int ch = (int)(rec % nch);
        size_t v = rec / nch;
        int x, y, z;
        size_t tile;
        if (!shell_only) {
            x = (int)(v % Wo); v /= Wo;
            y = (int)(v % Ho); v /= Ho;
            z = (int)(v % Do);
            tile = v / Do;
        } else {                                   // `total` counts face voxels only: two z faces, between them two y faces, between those two x faces
            const size_t nA = (size_t)2 * Ho * Wo, nB = (size_t)(Do - 2) * 2 * Wo, nC = (size_t)(Do - 2) * (Ho - 2) * 2;
            const size_t skipA = shell_only == 2 ? nA : 0;      // 2: the z faces were pooled in the epilogue of their own launches
            size_t f = v % (nA + nB + nC - skipA) + skipA;
            tile = v / (nA + nB + nC - skipA);
            if (f < nA) { z = f < (size_t)Ho * Wo ? 0 : Do - 1; y = (int)((f / Wo) % Ho); x = (int)(f % Wo); }
            else if (f < nA + nB) { f -= nA; z = 1 + (int)(f / (2 * Wo)); y = (f / Wo) % 2 ? Ho - 1 : 0; x = (int)(f % Wo); }
            else { f -= nA + nB; z = 1 + (int)(f / ((size_t)(Ho - 2) * 2)); y = 1 + (int)((f / 2) % (Ho - 2)); x = f % 2 ? Wo - 1 : 0; }
        }
        // The pooled record is the stored (hi, lo) PAIR of the window's largest element -- not a re-split of the joined maximum: the fused
        // pool of the conv epilogue splits max(x_i) of the unrounded values, which is exactly the pair the arg-max element was stored as;
        // re-splitting h0 + h1 can return the other representation of the same value when h1 is half an ulp of h0 (a tie), and the a1.b1 term
        // the arithmetic drops then differs.  Equal joined values: the larger hi came from the larger unrounded value (it rounded up).
        float m[4];
        u16x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned char* p = in + srec(tile, nch, (size_t)D * H * W, ch, (((size_t)(2 * z + (k >> 2))) * H + 2 * y + ((k >> 1) & 1)) * W + 2 * x + (k & 1));
            const u16x4 h = *reinterpret_cast<const u16x4*>(p + q * 8), l = *reinterpret_cast<const u16x4*>(p + 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = join2_f16(h[j], l[j]);
                const bool better = k == 0 || v > m[j] || (v == m[j] && (float)__builtin_bit_cast(_Float16, h[j]) > (float)__builtin_bit_cast(_Float16, hi[j]));
                if (better) { m[j] = v; hi[j] = h[j]; lo[j] = l[j]; }
            }
        }
        unsigned char* o = out + srec(tile, nch, (size_t)Do * Ho * Wo, ch, ((size_t)z * Ho + y) * Wo + x);
        *reinterpret_cast<u16x4*>(o + q * 8) = hi;
        *reinterpret_cast<u16x4*>(o + 32 + q * 8) = lo;
    }
}

// ---- interior of a per-tile pooled tensor out of the volume-wide one (shared ec0 -> ec1 pass): 16 bytes per thread -------------------------
// out[tile][chunk][z][y][x] = shared[chunk][tz/2 + z][ty/2 + y][tx/2 + x] for the pooled voxels NOT on a face of the tile
__global__ void __launch_bounds__(256) pooled_gather_kernel(const unsigned char* __restrict__ shared, unsigned char* __restrict__ out, TileSource g,
                                                            int Dp, int Hp, int Wp, int sDp, int sHp, int sWp, int nch, size_t total /*tiles * voxels * nch * 4*/) {
    const size_t pvox = (size_t)Dp * Hp * Wp, spl = (size_t)sDp * sHp * sWp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        size_t rec = i >> 2;
        const int x = (int)(rec % Wp); rec /= Wp;
        const int y = (int)(rec % Hp); rec /= Hp;
        const int z = (int)(rec % Dp); rec /= Dp;
        const int ch = (int)(rec % nch);
        const size_t tile = rec / nch;
        if (z == 0 || z == Dp - 1 || y == 0 || y == Hp - 1 || x == 0 || x == Wp - 1) continue;      // faces: written by the per-tile shell pass
        const int t = g.tile_begin + (int)tile;
        const int tz = (t / (g.gx * g.gy)) * g.ez / 2, ty = ((t / g.gx) % g.gy) * g.ey / 2, tx = (t % g.gx) * g.ex / 2;
        const uint4 v = *reinterpret_cast<const uint4*>(shared + ((size_t)ch * spl + ((size_t)(tz + z) * sHp + ty + y) * sWp + tx + x) * 64 + q * 16);
        *reinterpret_cast<uint4*>(out + srec(tile, nch, pvox, ch, ((size_t)z * Hp + y) * Wp + x) + q * 16) = v;
    }
}

// ---- head on a format-S input ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_sres_kernel(const unsigned char* __restrict__ in, int Cin, int D, int H, int W,
                                                        int lz, int ly, int lx, int bz, int by, int bx,
                                                        int kz, int ky, int kx, int ez, int ey, int ex,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int ncls, int out_mode, float* __restrict__ blocks, const int* __restrict__ boxes) {
    const size_t nvox = (size_t)bz * by * bx;
    const int tile = blockIdx.y;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const int x = lx + (int)(v % bx), y = ly + (int)((v / bx) % by), z = lz + (int)(v / ((size_t)bx * by));
    if (boxes) {
        const int* b = boxes + 6 * tile;
        if (z < b[0] || z >= b[3] || y < b[1] || y >= b[4] || x < b[2] || x >= b[5]) return;
    }
    const int nch = (Cin + 15) / 16;
    const size_t hplane = (size_t)D * H * W;
    const unsigned char* p = in + srec(tile, nch, hplane, 0, ((size_t)z * H + y) * W + x);
    float acc[4] = {0, 0, 0, 0};
    for (int ch = 0; ch < nch; ++ch)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u16x4 hi = *reinterpret_cast<const u16x4*>(p + ch * hplane * 64 + q * 8), lo = *reinterpret_cast<const u16x4*>(p + ch * hplane * 64 + 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = ch * 16 + 4 * q + j;
                if (c < Cin) {
                    const float xv = join2_f16(hi[j], lo[j]);
                    for (int k = 0; k < ncls; ++k) acc[k] = fmaf(xv, w[k * Cin + c], acc[k]);
                }
            }
        }
    const size_t evox = (size_t)ez * ey * ex;
    const size_t o = ((size_t)(z - kz) * ey + (y - ky)) * ex + (x - kx);
    for (int k = 0; k < ncls; ++k) {
        const float l = acc[k] + bias[k];
        float r = l;
        if (out_mode != 2) {
            const float pr = 1.0f / (1.0f + expf(-l));
            r = out_mode == 1 ? (pr > 0.5f ? 1.0f : 0.0f) : pr;
        }
        blocks[((size_t)tile * ncls + k) * evox + o] = r;
    }
}

}  // namespace oai
