// conv3_wino_sres: the split-resident 3x3x3 conv with the x axis in Winograd F(2,3) form -- two thirds of the matrix work.
//
// The fp16x3 conv is bound by the matrix pipe (MFMA-busy 0.86, clock at the power wall: profiles/r03_sq_summary.md); what moves its time
// is removing MFMAs.  Along x an output pair (X, X+1) of a k3 correlation needs the four inputs d0..d3 = in[X-1 .. X+2] and, per
// (dz, dy, cin), SIX products; in Winograd's minimal form FOUR:
//     t0 = d0 - d2   t1 = d1 + d2   t2 = d2 - d1   t3 = d1 - d3            (input transform: +-1 only)
//     u0 = g0   u1 = (g0 + g1 + g2) / 2   u2 = (g0 - g1 + g2) / 2   u3 = g2 (weights: host, in double, before the fp16 split)
//     m_f = sum over (dz, dy, cin) of t_f u_f                               (four GEMMs with K = 9 Cin instead of one with K = 27 Cin)
//     out[X] = (m0 + m1) + m2       out[X+1] = (m1 - m2) - m3               (output transform: +-1 only)
// The t_f are sums of two 22-bit values, formed in fp32 and split into an fp16 pair again exactly like a stored activation, so each
// product keeps the fp16x3 precision; the error against fp64 grows ~1.3 x over the direct form and stays at fp32-conv level
// (scripts/study/winograd_x_error.py).  y and z stay direct: each transformed axis doubles the accumulators, and 2 x is what fits.
//
// Workgroup = NG cout groups of four waves, ONE workgroup per CU (it holds 100 KB of LDS).  Block = 4 z x 8 y x 8 x outputs = 128
// x pairs; wave w of a group owns frequency f = w: four 32-row tiles (one per z slice; row = 8 y x 4 pairs) x 64 couts = 128
// accumulators, the same register shape as conv3_igemm_sres -- and the same tap loop, with 9 taps (dz, dy) per 16-channel chunk.
//
// Per chunk:  [transform: raw halo (LDS) -> T (LDS), one (z, y, pair, channel-half) unit per thread: 8 x 16 B in, ~180 VALU, 8 x 16 B out]
//             [barrier] [9 taps: A fragments from T, weight fragments from L2, 24 MFMAs each; the raw halo of chunk c+1 arrives by
//             LDS-DMA meanwhile, one piece per tap] [barrier].
// The raw halo (6 x 10 x 10 records) is single-buffered: it is dead once transformed, i.e. before the taps start.  T holds the four
// frequencies of the halo box: record (hz, f, hy, pair) at ((hz * 4 + f) * 10 + hy) * 4 + pair, slots XOR-swizzled by ((hy * NP + pair) >> 2) & 3
// (= hy & 3 for the main shape: 16 lanes of an A-fragment read = 4 y x 4 pairs hit 16 different 16-byte bank groups).  The raw box is laid out for the transform's
// reads, not as records: piece L = (hz * 10 + hy) * 41 + (term * 10 + hx) * 2 + half (one pad piece per row: 16 lanes = 4 y x 4 pairs
// read 16 different bank groups); global_load_lds writes lane-linearly, so the permutation is applied to the SOURCE address.
//
// Epilogue: the four frequencies of an output sit in four waves.  Per cout half the waves of a group exchange accumulators through
// LDS (wave w keeps z slice w: it sends three slices and receives three frequencies, 48 KB per group), apply the output transform,
// scale / shift / ReLU / split as conv3_igemm_sres does, build the block's image in LDS and copy it out 16 B per lane; ec3 / ec5:
// MaxPool3d(2) from that image (main shape, whole-tile boxes).  A transformed input beyond fp16's range turns its outputs NaN: the
// outputs inside the box are tested for finiteness before the ReLU and raise the overflow bit of the range flag.
//
// Measured (profiles/r03_winograd.md): 0.76-0.82 of the direct kernel's time per layer at 128 couts per workgroup, nothing at 64; what
// is left is the weight-fragment stream (36 / 27 of the direct form's bytes for 2 / 3 of its MFMAs) at the power wall.
//
// Every value of an output depends only on the parity of its x (pairs start at even tile coordinates: the host aligns the launch box)
// -- not on the block grid, the batch or the launch box: results do not depend on how tiles are batched.
#pragma once
#include "unet_sres2.h"
namespace oai {

// Block shapes: TY rows x NP pairs with TY * NP = 32 -- 8 x 4 (the main shape, figures above), 4 x 8 (y remainders of <= 4 rows) and
// 16 x 2 (x remainders of <= 4 columns): trimmed boxes are e.g. 12 x 52 x 52, and a block that is half outside its box costs all of its MFMAs.
// MS = 2 (NG = 1, layers with ONE block of 64 couts): eight waves all the same, wave (zp, f) owns frequency f of the z slices 2 zp, 2 zp + 1
// -- two accumulator tiles x 64 couts; the two waves of a frequency fetch the same weight fragments (the second from L1).  Four waves with
// four tiles each (NG = 1, MS = 1) leave every SIMD with one wave: measured no faster than the direct kernel.
// WS (NG = 1, MS = 1; one block of 64 couts): the eight waves are SPECIALISED.  Waves 0-3 only multiply -- wave = frequency, four tiles x 64
// couts per weight fragment set, the register shape of the two-group form -- and waves 4-7 only stage: each owns a quarter of the halo rows,
// requests their pieces, transforms them into the OTHER of two T buffers while the multipliers work on the current one, and requests the next
// chunk's pieces into the same places (no wave ever reads what another wave's DMA writes: no barrier inside the staging).  One barrier per
// chunk; no transform phase and no halo piece in front of a weight fragment on the multipliers' path.  2 x 60 KB of T + 40 KB of raw rows =
// the CU's 160 KB exactly.  (MS = 2 measured 0.43 MFMA-busy on dc2: its vector-memory path carries every weight fragment twice.)
// M16 (round 4; NG = 2, MS = 1, not WS): the same block, staging, transform and epilogue with the taps on v_mfma_f32_16x16x32_f16 instead of
// 32x32x16.  Why: this loop runs at the power wall, and at equal cycles per FLOP the 16x16x32 shape holds a ~12 % higher clock on split-fp16
// data with the structure of this tap loop (scripts/micro/mfma_shape.hip, profiles/r04_wino_stream.md: 1.66 -> 1.86 GHz, 1309 -> 1465 TFLOP/s
// executed at the same MFMA-busy share).  K = 32 is a PAIR OF TAPS x 16 channels, so every operand is a natural record read / panel load:
//     step j < 4:  taps (2j, 2j+1);  lanes 0-31 carry tap 2j, lanes 32-63 tap 2j+1 (lane group g = lane >> 4: channel half g & 1, tap g >> 1)
//         pass A  [a0(t) | a0(t')] . X'   X' = [b0(t) | b0(t')]        pass B  [a0 | a0'] . Y'   Y' = [b1(t) | b1(t')]        pass C  [a1 | a1'] . X'
//     step 4:      the ninth tap alone -- lanes 32-63 carry its LOW terms:  pass A  [a0 | a1] . [b0 | b0] = a0.b0 + a1.b0;  pass B  [a0 | a1] . [b1 | 0]
// = 14 K-32 MFMA steps per chunk where 13.5 would be exact (+3.7 %).  A wave's tile is the same 128 rows x 64 couts as 8 x 4 tiles of 16 x 16
// (m2 = 2 m + p: rows 16 p .. 16 p + 15 of slice m; n2 = 2 n + q): the same 128 accumulator registers, element (p, q, i) of the old (m, n)
// tile at row 16 p + 4 (lane >> 4) + i, cout column 16 q + (lane & 15).  Weight fragments: pack_wino16_panel, 40 KiB per chunk and 64 couts
// (36 in the 32x32 form), X' double-buffered, Y' reloaded behind pass B: 48 registers.  Every accumulator sees a0.b0 (two taps at once),
// a0.b1, a1.b0 per step: another summation order than the 32x32 form's, same arithmetic class (tests: the Winograd gates).
// PS (round 5; WS only): PERSISTENT workgroups.  One workgroup per CU pulls blocks from a per-XCD counter (ConvArgs::ps_plan, wino_plan_kernel) until none is
// left, and the staging waves run ONE BLOCK AHEAD: during the last two chunks of a block they request and transform chunk 0 of the next block, so the
// multipliers go from a block's epilogue straight into the next block's taps -- no prologue (request, L2 / HBM round trip, transform: ~5 us of a ~70-us
// block) and no workgroup launch in between (~4 us of an idle CU per block: profiles/r04_wino_stream.md section 6).  Same arithmetic, same order: bit-identical.
template <int NG, int TY, int NP, int MS = 1, bool WS = false, bool M16 = false, bool PS = false>
__global__ void __launch_bounds__(WS ? 512 : 256 * NG * MS, 1) conv3_wino_sres(const ConvArgs a, const unsigned char* __restrict__ zero_rec) {
    static_assert(!PS || (WS && !M16), "persistent workgroups: the specialised form");
    constexpr int NT = 256 * NG * MS, MREP = 4 / MS, NREP = 2;
    static_assert((MS == 1 || MS == 2) && NG * MS <= 2, "eight waves at most");
    static_assert(!M16 || MS == 1 || (NG == 1 && !WS), "the 16x16x32 taps: the four-slice forms (two groups, or specialised waves) and the slice-split 64-cout form");
    static_assert(!WS || (NG == 1 && MS == 1), "specialised waves: four multiply, four stage");
    constexpr int TZ = 4, TX = 2 * NP, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int RS = 4 * HX + 1;                                // 16-byte pieces per raw row (hz, hy): [term][hx][half] + 1 pad
    constexpr int PIECES = HZ * HY * RS;                          // 2460 for 8 x 4
    constexpr int NIT = (PIECES + NT - 1) / NT;                   // LDS-DMA instructions per thread per chunk: 5 (NG 2) / 10 (NG 1) for 8 x 4
    constexpr int TB = HZ * 4 * HY * NP * 64;                     // 61 440 bytes of T for 8 x 4
    constexpr int RPW = HZ * HY / 4, WNIT = (RPW * RS + 63) / 64;    // WS: halo rows and LDS-DMA instructions per staging wave and chunk (15, 10)
    constexpr int RAWB = WS ? 4 * WNIT * 1024 : NIT * NT * 16;    // 40 960 bytes of raw box (whole 1-KiB wave writes)
    constexpr int UNITS = HZ * 2 * HY * NP;                       // 480 transform units (hz, half, hy, pair)
    constexpr int XB = 3 * 4 * 4 * 1024;                          // exchange buffer of one cout group: [f][3 slices][4 row groups][lane] x 16 B
    static_assert(NG == 1 || NG == 2, "one or two cout groups");
    static_assert(TY * NP == 32 && (NP == 2 || NP == 4 || NP == 8), "32 rows per accumulator tile");
    constexpr int NTB = WS ? 2 : 1;                               // T buffers
    static_assert(!WS || (HZ * HY) % 4 == 0, "whole halo rows per staging wave");
    static_assert(NTB * TB + RAWB <= 160 * 1024 && NG * MS * XB <= TB + RAWB && TZ * TY * TX * 128 <= XB && TZ * TY * TX * 256 <= TB + RAWB, "epilogue buffers live in the idle T / raw space");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NTB * TB + RAWB];
    unsigned char* const Tl = lds;
    unsigned char* const raw = lds + NTB * TB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef OAI_DIAG
    // phase stamps (scripts/stamp_wino.py; s_memrealtime: 100 MHz): [0] prologue, [1] chunk loop, [2] epilogue, [8] waves; [10] / [11] = earliest start /
    // latest end of any wave (the span of the launch: span x CUs - the workgroups' own time = what a CU spends between workgroups + the tail)
    // [3] exchange (send, receive, three barriers), [4] output transform + image writes + barrier, [5] copy-out (+ fused pool) -- parts of [2], both halves
    unsigned long long wst[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0}, wep[3] = {0, 0, 0}, wlast = 0;
    // PS: [0] block switch (end barrier -> first chunk), [1] chunk loop, [2] epilogue up to its end barrier, [3] of [1]: at the chunk-end barriers (multiplying wave 0);
    // [4] the staging wave 4 at the chunk-end barriers (its slack), [5] its epilogue; [8] blocks
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0}, pslast = wst[0], psblocks = 0;
#define OAI_PSTAMP(i) do { if (PS && a.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); pst[i] += now_ - pslast; pslast = now_; } } while (0)
#define OAI_WSTAMP(i) do { if (a.stamps) wst[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define OAI_WEP0() do { if (a.stamps) wlast = __builtin_amdgcn_s_memrealtime(); } while (0)
#define OAI_WEP(i) do { if (a.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); wep[i] += now_ - wlast; wlast = now_; } } while (0)
#else
#define OAI_PSTAMP(i) do { } while (0)
#define OAI_WSTAMP(i) do { } while (0)
#define OAI_WEP0() do { } while (0)
#define OAI_WEP(i) do { } while (0)
#endif
    const int grp = (MS == 2 || WS) ? 0 : wave >> 2, zp = MS == 2 ? wave >> 2 : 0, f = wave & 3, gtid = tid & 255;
    const bool stager = WS && __builtin_amdgcn_readfirstlane(wave) >= 4;      // WS: waves 4-7 stage, waves 0-3 multiply (a scalar condition: the branches on it are uniform)
    // ---- the block (PS: re-assigned per block by set_block; otherwise fixed)
    const int nch0 = (a.C0 + 15) / 16, nch1 = (a.C1 + 15) / 16, nchunks = nch0 + nch1;
    const size_t plane = (size_t)a.D * a.H * a.W;
    int tile, cb, oz0, oy0, ox0, m_lo, m_hi;
    int blo[3], bhi[3];
    const unsigned char *s0, *s1;
    // (PS: the six ints of a.boxes[tile] can be handed in -- loaded a phase earlier, so that the switch to the next block does not wait for them)
    auto set_block = [&](int tile_, int bz, int by, int bx, int cbg, const int* pre = nullptr) __attribute__((always_inline)) -> bool {
        tile = tile_;
        cb = cbg * NG + grp;                                      // this group's block of 64 couts
        oz0 = a.lo[0] + bz * TZ; oy0 = a.lo[1] + by * TY; ox0 = a.lo[2] + bx * TX;      // a.lo[2] is even (host)
        if (pre) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { blo[i] = max(a.lo[i], pre[i]); bhi[i] = min(a.hi[i], pre[3 + i]); }
            if (!(blo[0] < bhi[0] && blo[1] < bhi[1] && blo[2] < bhi[2])) return false;
        } else if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return false;
        if (oz0 >= bhi[0] || oz0 + TZ <= blo[0] || oy0 >= bhi[1] || oy0 + TY <= blo[1] || ox0 >= bhi[2] || ox0 + TX <= blo[2]) return false;
        m_lo = max(0, blo[0] - oz0); m_hi = min(TZ, bhi[0] - oz0);
        s0 = reinterpret_cast<const unsigned char*>(a.src0) + srec(tile, nch0, plane, 0, 0);
        s1 = reinterpret_cast<const unsigned char*>(a.src1) + srec(tile, nch1, plane, 0, 0);
        return true;
    };
    if constexpr (!PS) {
        int id = a.xcd_group ? xcd_block_id(a.nblocks, a.xcd_group) : (int)blockIdx.x;
        if (id < 0) return;
        const int cbg = id % a.ncb; id /= a.ncb;                  // a.ncb = Cout / (64 NG) for this kernel
        const int bx = id % a.nbx; id /= a.nbx;
        const int by = id % a.nby; id /= a.nby;
        const int bz = id % a.nbz; id /= a.nbz;
        if (!set_block(id, bz, by, bx, cbg)) return;
    }

    const int row = lane & 31, half = lane >> 5;

    f32x16 acc[M16 ? 1 : MREP][M16 ? 1 : NREP];
    f32x4 acc4[M16 ? MREP : 1][M16 ? NREP : 1][4];                    // M16: [m][n][p * 2 + q], element i at row 16 p + 4 (lane >> 4) + i, column 16 q + (lane & 15)
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
            for (int t = 0; t < 4; ++t) acc4[m][n][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // element r (0..15) of the (m, n) tile, whichever shape holds it
    auto acc_el = [&](int m, int n, int r) __attribute__((always_inline)) -> float {
        if constexpr (M16) return acc4[m][n][r >> 2][r & 3];
        else return acc[m][n][r];
    };

    // ---- staging plan: piece L = it * NT + tid of the raw box -> byte offset of its 16 bytes inside a chunk plane, or kNoPiece (outside
    // the tile = Conv3d's zero padding, pad piece, beyond the box) -> the zero record
    constexpr unsigned kNoPiece = 0xFFFFFFFFu;
    unsigned poff[NIT];
#pragma unroll
    for (int it = 0; it < (WS ? 0 : NIT); ++it) {
        const int L = it * NT + tid;
        const int rw = L / RS, c = L - rw * RS;
        const int hz = rw / HY, hy = rw - hz * HY;
        const int term = c / (2 * HX), hx = (c - term * 2 * HX) >> 1, hf = c & 1;
        const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
        const bool ok = L < PIECES && c < 4 * HX && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        poff[it] = ok ? ((unsigned)((gz * a.H + gy) * a.W + gx) << 6) | (unsigned)(term * 32 + hf * 16) : kNoPiece;
    }
    const unsigned raw0 = lds_addr_of(raw);
    // one piece of chunk `ch`; past the last chunk the zero record is fetched instead, so that every chunk issues the same number of
    // vector-memory operations (the counted waits below depend on it)
    auto issue_piece = [&](int it, int ch) __attribute__((always_inline)) {
        const bool real = ch < nchunks;
        const bool first = ch < nch0;
        const unsigned char* cbase = (first ? s0 : s1) + (size_t)(first ? ch : ch - nch0) * plane * 64;     // wave-uniform chunk plane
        const unsigned char* g = (real && poff[it] != kNoPiece) ? cbase + poff[it] : zero_rec;
        lds_dma16(g, __builtin_amdgcn_readfirstlane(raw0 + (it * NT + wave * 64) * 16));
    };
    // M16: the piece from a wave-uniform base (SGPR pair) + a 32-bit per-lane offset -- no 64-bit per-lane address, no select against the zero
    // record (hoisted out of the chunk loop those are ten more registers; spilled, their reload inside the taps drains vmcnt).  A piece outside
    // the tile reads offset 0 of the plane (valid memory, unused data) and its LDS slot is zeroed behind the chunk-end wait (zero_missing);
    // past the last chunk the last chunk is fetched again.
    auto issue_piece_s = [&](int it, int ch) __attribute__((always_inline)) {
        const int che = ch < nchunks ? ch : nchunks - 1;
        const bool first = che < nch0;
        const unsigned char* cbase = (first ? s0 : s1) + (size_t)(first ? che : che - nch0) * plane * 64;     // wave-uniform chunk plane
        const unsigned off = poff[it] != kNoPiece ? poff[it] : 0u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(cbase), "s"(__builtin_amdgcn_readfirstlane(raw0 + (it * NT + wave * 64) * 16)) : "memory");
    };
    unsigned miss = 0;                                              // bit it: piece it of this thread lies outside the tile
    if constexpr (M16 && !WS) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) miss |= poff[it] == kNoPiece ? 1u << it : 0u;
    }
    auto zero_missing = [&]() __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64(miss != 0) == 0) return;     // (interior blocks: no wave has one)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (miss & (1u << it)) *reinterpret_cast<float4*>(raw + (it * NT + tid) * 16) = float4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    // pieces requested in tap t (for the NEXT chunk): spread over the nine taps, the early taps take the remainder
    auto pieces_in_tap = [](int t) constexpr { return WS ? 0 : NIT / 9 + (t < NIT % 9 ? 1 : 0); };
    auto first_piece = [](int t) constexpr { return t * (NIT / 9) + (t < NIT % 9 ? t : NIT % 9); };
    static_assert(NIT <= 18, "at most two pieces per tap (vm_wait<0..2>)");

    // ---- input transform of the staged chunk: raw -> T
    // (a t beyond fp16's range -- inputs <= 65504, |t| <= 131008 -- becomes the pair (inf, -inf): every output that depends on it turns NaN and
    // is reported by the epilogue's finiteness test.  Not tested here: halo voxels that no output inside the box reads may be uninitialised memory.)
    // one unit: the four frequencies of (hz, hy, pair p, channel half hf) from the raw pieces at rb into the T buffer at Tdst
    auto transform_unit = [&](const unsigned char* rb, int hz, int hy, int p, int hf, unsigned char* Tdst) __attribute__((always_inline)) {
        float x[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u16x8 h0 = *reinterpret_cast<const u16x8*>(rb + (2 * i) * 16);
            const u16x8 h1 = *reinterpret_cast<const u16x8*>(rb + (2 * HX + 2 * i) * 16);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[i][c] = join2_f16(h0[c], h1[c]);
        }
        unsigned char* tw = Tdst + (((hz * 4) * HY + hy) * NP + p) * 64;
        const int key = ((hy * NP + p) >> 2) & 3;
        const int sl0 = ((hf) ^ key) * 16, sl1 = ((2 + hf) ^ key) * 16;
#pragma unroll
        for (int fq = 0; fq < 4; ++fq) {
            u16x8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                f32x2 v;
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    v[e] = fq == 0 ? x[0][c + e] - x[2][c + e] : fq == 1 ? x[1][c + e] + x[2][c + e] : fq == 2 ? x[2][c + e] - x[1][c + e] : x[1][c + e] - x[3][c + e];
                const f16x2 h = __builtin_convertvector(v, f16x2);
                const f32x2 res = v - __builtin_convertvector(h, f32x2);
                const f16x2 l = __builtin_convertvector(res, f16x2);
                const unsigned H = __builtin_bit_cast(unsigned, h), Lw = __builtin_bit_cast(unsigned, l);
                hi[c] = (unsigned short)H; hi[c + 1] = (unsigned short)(H >> 16);
                lo[c] = (unsigned short)Lw; lo[c + 1] = (unsigned short)(Lw >> 16);
            }
            *reinterpret_cast<u16x8*>(tw + fq * (HY * NP * 64) + sl0) = hi;
            *reinterpret_cast<u16x8*>(tw + fq * (HY * NP * 64) + sl1) = lo;
        }
    };
    auto transform = [&]() __attribute__((always_inline)) {
        int tq = tid;
        if constexpr (M16) asm volatile("" : "+v"(tq));             // M16: the unit's addresses are recomputed every chunk instead of living in (spilled) registers through the taps
#pragma unroll
        for (int ui = 0; ui < (UNITS + NT - 1) / NT; ++ui) {
            const int u = ui * NT + tq;
            if (u < UNITS) {
                const int p = u % NP;
                int t = u / NP;
                const int hy = t % HY; t /= HY;
                const int hf = t & 1, hz = t >> 1;
                transform_unit(raw + (((hz * HY + hy) * RS) + 4 * p + hf) * 16, hz, hy, p, hf, Tl);
            }
        }
    };
    // ---- WS, staging waves: wave sw = wave - 4 owns the halo rows r = sw + 4 j, j < RPW, stored as [j][RS pieces] in its own 1-KiB-aligned part
    // of the raw box.  stage_request(ch): its WNIT LDS-DMA instructions for chunk ch; stage_transform(ch): its rows -> T[ch & 1]
    const int sw = wave & 3;
    unsigned char* const raw_w = raw + sw * (WNIT * 1024);
    unsigned woff[WS ? WNIT : 1];
    // this staging wave's piece offsets for the block at (z0, y0, x0)
    auto stage_plan = [&](unsigned (&wo)[WS ? WNIT : 1], int z0, int y0, int x0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < (WS ? WNIT : 0); ++it) {
            const int q = it * 64 + lane;
            const int rj = q / RS, c = q - rj * RS;
            const int r = sw + 4 * rj, hz = r / HY, hy = r - hz * HY;
            const int term = c / (2 * HX), hx = (c - term * 2 * HX) >> 1, hf = c & 1;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = rj < RPW && c < 4 * HX && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            wo[WS ? it : 0] = ok ? ((unsigned)((gz * a.H + gy) * a.W + gx) << 6) | (unsigned)(term * 32 + hf * 16) : kNoPiece;
        }
    };
    if constexpr (WS && !PS) stage_plan(woff, oz0, oy0, ox0);
    // PS: the block-independent part of the plan, once per kernel: (hz, hy, hx, term / half, piece exists) per DMA instruction, packed; per block only the
    // origin is added (the divisions of stage_plan cost a staging wave ~3 us per block)
    unsigned wpk[PS ? WNIT : 1];
    if constexpr (PS) {
#pragma unroll
        for (int it = 0; it < WNIT; ++it) {
            const int q = it * 64 + lane;
            const int rj = q / RS, c = q - rj * RS;
            const int r = sw + 4 * rj, hz = r / HY, hy = r - hz * HY;
            const int term = c / (2 * HX), hx = (c - term * 2 * HX) >> 1, hf = c & 1;
            wpk[it] = (unsigned)hz | (unsigned)hy << 3 | (unsigned)hx << 8 | (unsigned)(term * 2 + hf) << 13 | (rj < RPW && c < 4 * HX ? 1u << 15 : 0u);
        }
    }
    auto stage_plan_fast = [&](unsigned (&wo)[WS ? WNIT : 1], int z0, int y0, int x0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < (PS ? WNIT : 0); ++it) {
            const unsigned k = wpk[PS ? it : 0];
            const int gz = z0 - 1 + (int)(k & 7), gy = y0 - 1 + (int)((k >> 3) & 31), gx = x0 - 1 + (int)((k >> 8) & 31);
            const bool ok = (k >> 15) && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            wo[PS ? it : 0] = ok ? ((unsigned)((gz * a.H + gy) * a.W + gx) << 6) | (((k >> 13) & 3) << 4) : kNoPiece;
        }
    };
    // PS: the lanes of a wave's last DMA instruction that lie beyond its pieces do not write -- the tail of the raw box holds the control words
    constexpr int kLastLanes = WS ? RPW * RS - (WNIT - 1) * 64 : 64;
    static_assert(!PS || kLastLanes <= 56, "PS: room for the control words behind the last piece");
    auto stage_request_of = [&](const unsigned (&wo)[WS ? WNIT : 1], const unsigned char* b0, const unsigned char* b1, int ch) __attribute__((always_inline)) {
        const bool first = ch < nch0;
        const unsigned char* cbase = (first ? b0 : b1) + (size_t)(first ? ch : ch - nch0) * plane * 64;     // wave-uniform chunk plane
        const unsigned dst = lds_addr_of(raw_w);
#pragma unroll
        for (int it = 0; it < (WS ? WNIT : 0); ++it) {
            const unsigned char* g = (wo[WS ? it : 0] != kNoPiece) ? cbase + wo[WS ? it : 0] : zero_rec;
            if (!PS || it + 1 < WNIT || lane < kLastLanes) lds_dma16(g, __builtin_amdgcn_readfirstlane(dst + it * 1024));
        }
    };
    auto stage_request = [&](int ch) __attribute__((always_inline)) { stage_request_of(woff, s0, s1, ch); };
    auto stage_transform_to = [&](unsigned char* Tdst) __attribute__((always_inline)) {
#pragma unroll
        for (int ui = 0; ui < (RPW * 2 * NP + 63) / 64; ++ui) {
            const int idx = ui * 64 + lane;
            if (idx < RPW * 2 * NP) {
                const int rj = idx / (2 * NP), rem = idx - rj * 2 * NP;
                const int p = rem % NP, hf = rem / NP;
                const int r = sw + 4 * rj, hz = r / HY, hy = r - hz * HY;
                transform_unit(raw_w + (rj * RS + 4 * p + hf) * 16, hz, hy, p, hf, Tdst);
            }
        }
    };
    auto stage_transform = [&](int ch) __attribute__((always_inline)) { stage_transform_to(Tl + (ch & 1) * TB); };

    // ---- A fragments: this lane's record for tap (dz, dy) = (0, 0), slice 0, and its swizzled slot per dy and term
    int aofs, sl[3][2];
    unsigned wlane;                                                 // this lane's 16 bytes of a weight fragment
    // (PS: recomputed per block from an opaque copy of the lane id -- kept live across the epilogue, the register peak of the block loop, they would be spilled)
    auto set_lane_consts = [&]() __attribute__((always_inline)) {
        int lq = lane;
        if constexpr (PS) asm volatile("" : "+v"(lq));
        const int rw = lq & 31, hf = lq >> 5;
        aofs = (((zp * MREP * 4 + f) * HY + rw / NP) * NP + rw % NP) * 64;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int t = 0; t < 2; ++t) sl[dy][t] = ((t * 2 + hf) ^ (((rw + dy * NP) >> 2) & 3)) * 16;
        wlane = (unsigned)lq * 16;
    };
    set_lane_consts();

    // M16: byte offset (inside T) of this lane's record slot per step and term -- lane group g = lane >> 4 reads channel half g & 1 of tap
    // 2 j + (g >> 1) (steps 0..3; term k), or of term g >> 1 of tap 8 (step 4); row r16 = lane & 15 of the 16-row tile p = 0 (p = 1: + 1024)
    // (computed at the top of every chunk's taps from an opaque copy of the lane id: kept live through the transform -- the register peak of
    // the chunk loop -- they are what tips the allocator into spilling)
    unsigned a16[M16 ? 9 : 1];
    auto compute_a16 = [&](unsigned tbofs = 0) __attribute__((always_inline)) {      // tbofs: byte offset of the T buffer to read (WS: two of them)
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const int r16 = lq & 15, hsel = (lq >> 4) & 1, tsel = lq >> 5;
        const int y16 = r16 / NP, p16 = r16 % NP;
        const unsigned rowofs = tbofs + (unsigned)(((f * HY + y16) * NP + p16) * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = 2 * j + tsel, dz = t / 3, dy = t - 3 * dz;
                const int key = ((r16 + dy * NP) >> 2) & 3;
                a16[j * 2 + k] = rowofs + (unsigned)(((dz * 4 * HY + dy) * NP) * 64 + ((k * 2 + hsel) ^ key) * 16);
            }
        {
            const int key = ((r16 + 2 * NP) >> 2) & 3;                // tap 8 = (dz 2, dy 2)
            a16[8] = rowofs + (unsigned)(((2 * 4 * HY + 2) * NP) * 64 + ((tsel * 2 + hsel) ^ key) * 16);
        }
    };
    constexpr int STEP = M16 ? 2 * 4 * 64 : 2 * NREP * 64;          // 16-byte units of weights per tap (M16: per STEP of two taps: [X' | Y'][n2 4][lane])
    constexpr int NSTEPS = M16 ? 5 : 9;                             // fragment sets per chunk
    // panel of pack_wino_panel: [cb][f][chunk][tap 9][term][nr][lane] (M16, pack_wino16_panel: [cb][f][chunk][step 5][X' | Y'][n2][lane]);
    // wave-uniform base, forced into an SGPR pair
    const size_t wp_v = (size_t)(a.wpanel + (size_t)(cb * 4 + f) * nchunks * NSTEPS * STEP);
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(
        ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(wp_v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp_v));
    const unsigned char* const wp_base = wp;                        // (PS: every block starts at the panel's first tap again)
    int pb = 0;                                                     // WS: the T buffer of the block's chunk 0 (chunk ch reads buffer (pb + ch) & 1; not PS: always 0)

    f32x4 fs[WS && M16 ? 2 : 1][8];                                  // [set][X' n2 0..3 | Y' n2 0..3]
    auto ws16_request = [&](f32x4 (&d)[8]) __attribute__((always_inline)) {
        {
            d[0] = gload16_asm<0>(wp, wlane); d[1] = gload16_asm<1024>(wp, wlane); d[2] = gload16_asm<2048>(wp, wlane); d[3] = gload16_asm<3072>(wp, wlane);
            d[4] = gload16_asm<0>(wp + 4096, wlane); d[5] = gload16_asm<1024>(wp + 4096, wlane); d[6] = gload16_asm<2048>(wp + 4096, wlane); d[7] = gload16_asm<3072>(wp + 4096, wlane);
        }
        wp += STEP * 16;
    };
    auto ws16_landed = [&](f32x4 (&d)[8]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) :: "memory");
    };
    // ---- prologue: the first raw box and the weight fragments of tap 0 (WS: the stagers also transform chunk 0 and request chunk 1)
    if constexpr (!WS) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) { if constexpr (M16) issue_piece_s(it, 0); else issue_piece(it, 0); }
    }
    // D = taps between the request of a fragment set and its use.  WS: a multiplier has its SIMD to itself -- nobody covers a wait -- and a tap
    // is 24 MFMAs = 0.37 us, less than an L2 round trip: D = 2, three register sets (9 % 3 == 0: the set of a tap does not depend on the chunk);
    // the registers come from the A fragments, see the slice-major tap below
    constexpr int D = WS ? 2 : 1, NB = D + 1;
    f32x4 bq[NB][2][NREP];                                          // [tap % NB][term][n]
    f32x4 bXlo[2], bXhi[2], bY[4];                                  // M16: X' (the high terms b0), couts n 0, 1 and n 2, 3, and Y' (the low terms b1) of the running step
    if constexpr (PS) {
        // (the persistent driver at the end of the kernel has its own prologue, behind the first block's assignment)
    } else if (M16 && !WS) {
        // (round 5) the fragments of a chunk's step 0 are requested at the chunk's top, in front of the transform: see run_chunks
        sgpr_settle(wp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first raw box has landed
        zero_missing();
    } else if (M16 && !stager) {                                    // (WS: the multipliers only) step 0 of chunk 0 into fragment set 0, landed before the dispatch over ML
        sgpr_settle(wp);
        if constexpr (WS && M16) {
            ws16_request(fs[0]);
            ws16_landed(fs[0]);
        }
    } else if (!stager) {
        sgpr_settle(wp);                                            // wp has just been made uniform by v_readfirstlane
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int n = 0; n < NREP; ++n) bq[d][k][n] = k == 0 ? (n == 0 ? gload16_asm<0>(wp, wlane) : gload16_asm<1024>(wp, wlane))
                                                                    : (n == 0 ? gload16_asm<2048>(wp, wlane) : gload16_asm<3072>(wp, wlane));
            wp += STEP * 16;
        }
        static_assert(NREP == 2, "vm_wait names four fragments");
        vm_wait<0>(bq[D - 1][0][0], bq[D - 1][0][1], bq[D - 1][1][0], bq[D - 1][1][1]);
    } else {
        stage_request(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_transform(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // its reads of the raw rows are done: they may be overwritten
        if (nchunks > 1) stage_request(1);
    }
    if constexpr (!PS) __syncthreads();
    OAI_WSTAMP(1);

    // ---- WS + M16 (round 5): the multipliers' chunks, SLICE-MAJOR on TWO whole fragment sets.  The first tap-pair version of this form (round 4) kept the
    // two-group form's pass order on 32 fragment registers: X' / Y' were re-requested 48-64 MFMAs before their next use, and a multiplier has its SIMD to
    // itself -- nobody covers the rest of an L2 round trip: +15 % clock, MFMA-busy 0.55 -> 0.50, nothing gained.  Here a step runs slice by slice -- per
    // slice m: B a0(m).Y', A a0(m).X', C a1(m).X' (24 MFMAs; every accumulator still sees B, A, C in this order: bit-identical to the pass-major order) --
    // so only ONE slice's A fragments are live (24 registers instead of 32 + 32), and the registers that frees hold a SECOND fragment set: the fragments of
    // step j + 1 are requested at the top of step j, a whole step (96 MFMAs, ~0.7 us) ahead, and waited for with vmcnt(0) at the top of step j + 1.  Five steps
    // per chunk is odd, so the set a chunk starts with alternates: the chunk loop is unrolled by two (P = the set of step 0).
    auto ws16_chunk = [&](auto ml_tag, auto p_tag, int ch) __attribute__((always_inline)) {
        constexpr int ML = decltype(ml_tag)::value, P = decltype(p_tag)::value;
        constexpr int SL = 4 * HY * NP * 64;
        auto lda = [&](unsigned off, int m, int p) __attribute__((always_inline)) {
            return *reinterpret_cast<const float4*>(Tl + off + m * SL + p * 1024);
        };
        auto mma = [&](const float4& av, const f32x4& bv, f32x4& c) __attribute__((always_inline)) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
        };
        compute_a16((unsigned)(P * TB));                              // (chunk ch reads T buffer ch & 1 = P: the loop below is unrolled by two)
        float4 a0[2], a1[2];                                            // this slice's a0 / a1 fragments
        if constexpr (ML > 0) { a0[0] = lda(a16[0], 0, 0); a0[1] = lda(a16[0], 0, 1); }
        auto step = [&](auto jtag) __attribute__((always_inline)) {
            constexpr int j = decltype(jtag)::value;
            f32x4 (&cur)[8] = fs[(P + j) & 1];
            f32x4 (&nxt)[8] = fs[(P + j + 1) & 1];
            ws16_landed(cur);                                           // requested a whole step ago
            __builtin_amdgcn_sched_barrier(0);
            ws16_request(nxt);                                          // step j + 1 (behind step 4: step 0 of the next chunk)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < ML; ++m) {
                if (j < 4) { a1[0] = lda(a16[j * 2 + 1], m, 0); a1[1] = lda(a16[j * 2 + 1], m, 1); }      // needed 16 MFMAs from here
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 2; ++p)                           // pass B: a0 . Y'   (step 4: [a0 | a1] . [b1 | 0])
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma(a0[p], cur[4 + n], acc4[m][n >> 1][p * 2 + (n & 1)]);
#pragma unroll
                for (int p = 0; p < 2; ++p)                           // pass A: a0 . X'
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma(a0[p], cur[n], acc4[m][n >> 1][p * 2 + (n & 1)]);
                __builtin_amdgcn_sched_barrier(0);
                // the next slice's a0 (the next step's slice 0 behind the last slice; nothing behind the chunk's last slice: the other T buffer)
                const bool more = m + 1 < ML || j < 4;
                if (more) {                                           // (a0 is dead behind pass A: straight into its registers, pass C covers the LDS latency)
                    const int nj = m + 1 < ML ? j : j + 1, nm = m + 1 < ML ? m + 1 : 0;
                    const unsigned off = a16[nj < 4 ? nj * 2 : 8];
                    a0[0] = lda(off, nm, 0); a0[1] = lda(off, nm, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j < 4) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)                       // pass C: a1 . X'
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma(a1[p], cur[n], acc4[m][n >> 1][p * 2 + (n & 1)]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{});
        // chunk end: the fragments of the next chunk's step 0 stay in flight (set P ^ 1: waited for at its top); the barrier: the stagers have finished
        // the other T buffer, and everybody is done reading this one
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto run_chunks = [&](auto ml_tag) __attribute__((always_inline)) {
        constexpr int ML = decltype(ml_tag)::value;
        if constexpr (WS && M16) {
            int ch = 0;
            for (; ch + 1 < nchunks; ch += 2) {
                ws16_chunk(ml_tag, std::integral_constant<int, 0>{}, ch);
                ws16_chunk(ml_tag, std::integral_constant<int, 1>{}, ch + 1);
            }
            if (ch < nchunks) ws16_chunk(ml_tag, std::integral_constant<int, 0>{}, ch);
            return;
        }
        for (int ch = 0; ch < nchunks; ++ch) {
            if constexpr (!WS) {
                if constexpr (M16) {
                    // (round 5) The weight fragments of this chunk's step 0, requested HERE -- they land under the transform (~1 us) -- instead of
                    // during the previous chunk's last step, where hi(0) was requested 16 MFMAs before the chunk-end wait: every chunk ended with
                    // an exposed L2 round trip, for all eight waves (the barrier waits for the slowest).  Same basic block as their first use.
                    {
#pragma unroll
                        for (int n = 0; n < 4; ++n) bY[n] = n == 0 ? gload16_asm<0>(wp + 4096, wlane) : n == 1 ? gload16_asm<1024>(wp + 4096, wlane) : n == 2 ? gload16_asm<2048>(wp + 4096, wlane) : gload16_asm<3072>(wp + 4096, wlane);
                        bXlo[0] = gload16_asm<0>(wp, wlane); bXlo[1] = gload16_asm<1024>(wp, wlane);
                        bXhi[0] = gload16_asm<2048>(wp, wlane); bXhi[1] = gload16_asm<3072>(wp, wlane);
                    }
                    wp += STEP * 16;
                }
                transform();
                if constexpr (M16) compute_a16((unsigned)(zp * MREP) * (unsigned)(4 * HY * NP * 64));      // (MS = 2: this wave's slice pair)   behind the transform's register peak, in front of the barrier: under the wait for the slowest wave
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                        // T is complete; the raw box is free for the next chunk's pieces
                asm volatile("" ::: "memory");
            }
            const unsigned char* abase = Tl + (WS ? ((ch + pb) & 1) * TB : 0) + aofs;
            if constexpr (M16 && WS) {
                // (round 5: the chunk loop of this form is ws16_chunks below -- slice-major steps on two whole fragment sets)
                (void)abase;
            } else if constexpr (M16) {
                // Five steps, pass order B, A, C -- a0 . Y', a0 . X', a1 . X'.  Y' is dead after the first pass: its registers take Y' of the
                // next step a whole step ahead.  X' is live through the last two passes; so that its successor has more than pass B to land
                // in, passes A and C run the LOW couts (n 0, 1) over all slices and then the HIGH couts (n 2, 3): X' lo is dead after C lo and
                // re-requested there, X' hi at the end of the step, each 48 MFMAs (the 32x32 form's lead) before its first use in A lo / A hi of
                // the next step.  32 fragment registers.  Vector-memory order per step: Y'(j+1) | lo(j+1) | hi(j+1) | pieces(j); the counted
                // waits leave exactly the younger requests in flight.
                constexpr int SL = 4 * HY * NP * 64;                    // bytes between the T images of consecutive z slices
                auto lda = [&](unsigned off, int m, int p) __attribute__((always_inline)) {
                    return *reinterpret_cast<const float4*>(Tl + off + m * SL + p * 1024);
                };
                auto mma = [&](const float4& av, const f32x4& bv, f32x4& c) __attribute__((always_inline)) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
                };
                // pieces of the NEXT chunk's raw box: requested at the ends of steps 0..3 (step 4 requests nothing: what the chunk-end wait covers was
                // requested at least one step -- 96 MFMAs -- earlier)
                auto np_of = [](int j) constexpr { int c = 0; for (int q = 0; q < NIT; ++q) c += (j < 4 && q % 4 == j) ? 1 : 0; return c; };
                static_assert(NIT >= 4 && NIT <= 12, "one to three pieces per step 0..3 (the waits below name 1, 2 or 3)");
                float4 af[MREP][2];                                     // [m][p]: the A fragments of the running pass
#pragma unroll
                for (int m = 0; m < ML; ++m)
#pragma unroll
                    for (int p = 0; p < 2; ++p) af[m][p] = lda(a16[0], m, p);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                        // vector-memory operations younger than Y'(j) at the top of step j: lo(j), hi(j) and the pieces requested at the end of step j - 1
                    const int npp = j > 0 ? np_of(j - 1) : 0;           // (step 0: the fragments were requested at the chunk's top, nothing behind them)
                    if (npp == 0) vm_wait<4>(bY[0], bY[1], bY[2], bY[3]); else if (npp == 1) vm_wait<5>(bY[0], bY[1], bY[2], bY[3]); else if (npp == 2) vm_wait<6>(bY[0], bY[1], bY[2], bY[3]); else vm_wait<7>(bY[0], bY[1], bY[2], bY[3]);
                    __builtin_amdgcn_sched_barrier(0);
                    // pass B: a0 . Y'  (step 4: [a0 | a1] . [b1 | 0])
#pragma unroll
                    for (int m = 0; m < ML; ++m)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int n = 0; n < 4; ++n) mma(af[m][p], bY[n], acc4[m][n >> 1][p * 2 + (n & 1)]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 4) {               // Y' of the next step
#pragma unroll
                        for (int n = 0; n < 4; ++n) bY[n] = n == 0 ? gload16_asm<0>(wp + 4096, wlane) : n == 1 ? gload16_asm<1024>(wp + 4096, wlane) : n == 2 ? gload16_asm<2048>(wp + 4096, wlane) : gload16_asm<3072>(wp + 4096, wlane);
                    }
                    // lo(j) has landed; younger: hi(j) 2, pieces(j - 1) npp, Y'(j + 1) 4 (steps 0..3)
                    {
                        constexpr int kD2 = 0; (void)kD2;
                        const int y = 2 + npp + (j < 4 ? 4 : 0);
                        if (y == 9) asm volatile("s_waitcnt vmcnt(9)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        else if (y == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        else if (y == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        else if (y == 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        else if (y == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        else if (y == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(bXlo[0]), "+v"(bXlo[1]) :: "memory");
                        static_assert(true, "");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // pass A, low couts: a0 . X'[0, 1] over all slices
#pragma unroll
                    for (int m = 0; m < ML; ++m)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int n = 0; n < 2; ++n) mma(af[m][p], bXlo[n], acc4[m][0][p * 2 + n]);
                    __builtin_amdgcn_sched_barrier(0);                  // (the wait below must not rise above these MFMAs: they are its lead)
                    // hi(j) has landed; younger: pieces(j - 1), Y'(j + 1)
                    {
                        const int y = npp + (j < 4 ? 4 : 0);
                        if (y == 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                        else if (y == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                        else if (y == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                        else if (y == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                        else if (y == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                        else if (y == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" : "+v"(bXhi[0]), "+v"(bXhi[1]) :: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // pass A, high couts; behind each slice the a1 fragments of pass C take its registers
#pragma unroll
                    for (int m = 0; m < ML; ++m) {
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int n = 0; n < 2; ++n) mma(af[m][p], bXhi[n], acc4[m][1][p * 2 + n]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (j < 4) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) af[m][p] = lda(a16[j * 2 + 1], m, p);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (j < 4) {
                        // pass C, low couts: a1 . X'[0, 1] over all slices
#pragma unroll
                        for (int m = 0; m < ML; ++m)
#pragma unroll
                            for (int p = 0; p < 2; ++p)
#pragma unroll
                                for (int n = 0; n < 2; ++n) mma(af[m][p], bXlo[n], acc4[m][0][p * 2 + n]);
                        __builtin_amdgcn_sched_barrier(0);
                        {                    // the low couts of the next step's X'
                            bXlo[0] = gload16_asm<0>(wp, wlane); bXlo[1] = gload16_asm<1024>(wp, wlane);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // pass C, high couts; behind each slice the a0 fragments of the next step take its registers
#pragma unroll
                        for (int m = 0; m < ML; ++m) {
#pragma unroll
                            for (int p = 0; p < 2; ++p)
#pragma unroll
                                for (int n = 0; n < 2; ++n) mma(af[m][p], bXhi[n], acc4[m][1][p * 2 + n]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int p = 0; p < 2; ++p) af[m][p] = lda(a16[j < 3 ? (j + 1) * 2 : 8], m, p);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        {                    // the high couts of the next step's X'
                            bXhi[0] = gload16_asm<2048>(wp, wlane); bXhi[1] = gload16_asm<3072>(wp, wlane);
                        }
                        wp += STEP * 16;
                        {
#pragma unroll
                            for (int q = 0; q < NIT; ++q)
                                if (q % 4 == j) issue_piece_s(q, ch + 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // chunk end: the next raw box has landed for this thread (its last pieces were requested a whole step ago); the barrier says so
                // for everybody -- and that everybody is done reading T.  No weight fragment is in flight (step 4 requested none).
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                zero_missing();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (M16) continue;
            else {                                                   // (discarded for M16: its accumulators have another type)
            if constexpr (WS) {
                // the multiplier's chunk: 9 taps, slice-major -- per z slice m: a0[m].b0, a0[m].b1, a1[m].b0 for both cout halves (six MFMAs; every
                // accumulator still sees a0.b0, a0.b1, a1.b0 in this order), with the fragment pair of the NEXT slice requested in front of
                // them: one pair live + one in flight = 16 VGPRs instead of 32, the difference pays the third weight-fragment set
                auto load_pair = [&](float4 (&dst)[2], int t, int m) __attribute__((always_inline)) {
                    const int dz = t / 3, dy = t % 3;
#pragma unroll
                    for (int k = 0; k < 2; ++k) dst[k] = *reinterpret_cast<const float4*>(abase + (((m + dz) * 4 * HY + dy) * NP) * 64 + sl[dy][k]);
                };
                float4 aa[2][2];                                     // [step parity][term]
                if constexpr (ML > 0) load_pair(aa[0], 0, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    f32x4 (&bc)[2][NREP] = bq[t % 3];
                    f32x4 (&bn)[2][NREP] = bq[(t + 2) % 3];
                    vm_wait<4>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);      // in flight behind B(t): the fragments of tap t+1
                    {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int n = 0; n < NREP; ++n)
                                bn[k][n] = k == 0 ? (n == 0 ? gload16_asm<0>(wp, wlane) : gload16_asm<1024>(wp, wlane))
                                                  : (n == 0 ? gload16_asm<2048>(wp, wlane) : gload16_asm<3072>(wp, wlane));
                    }
                    wp += STEP * 16;
#pragma unroll
                    for (int m = 0; m < ML; ++m) {
                                const int step = t * ML + m;                     // aa[step & 1] holds (t, m)
                        __builtin_amdgcn_sched_barrier(0);
                        if (m + 1 < ML) load_pair(aa[(step + 1) & 1], t, m + 1);
                        else if (t + 1 < 9) load_pair(aa[(step + 1) & 1], t + 1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        float4 (&ac)[2] = aa[step & 1];
#pragma unroll
                        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(ac[0], __builtin_bit_cast(float4, bc[0][n]), acc[m][n]);      // a0.b0
#pragma unroll
                        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(ac[0], __builtin_bit_cast(float4, bc[1][n]), acc[m][n]);      // a0.b1
#pragma unroll
                        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(ac[1], __builtin_bit_cast(float4, bc[0][n]), acc[m][n]);      // a1.b0
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // chunk end: the fragments of the next chunk's taps 0 and 1 stay in flight (waited for there); the barrier: the stagers have
                // finished the other T buffer, and everybody is done reading this one
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                OAI_PSTAMP(1);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                OAI_PSTAMP(3);
                continue;
            }
            auto load_a = [&](float4 (&dst)[MREP], int t, int k) __attribute__((always_inline)) {
                const int dz = t / 3, dy = t % 3;
#pragma unroll
                for (int m = 0; m < ML; ++m)
                    dst[m] = *reinterpret_cast<const float4*>(abase + (((m + dz) * 4 * HY + dy) * NP) * 64 + sl[dy][k]);      // (abase: slice zp * MREP)
            };
            float4 acur[2][MREP];
            load_a(acur[0], 0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x4 (&bc)[2][NREP] = bq[t & 1];
                f32x4 (&bn)[2][NREP] = bq[(t + 1) & 1];
                // B(t) was requested in tap t-1, in front of that tap's pieces: they may stay in flight
                {
                        const int younger = t > 0 ? pieces_in_tap(t - 1) : 0;
                    if (younger == 0) vm_wait<0>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                    else if (younger == 1) vm_wait<1>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                    else vm_wait<2>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                }
                load_a(acur[1], t, 1);
                {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int n = 0; n < NREP; ++n)
                            bn[k][n] = k == 0 ? (n == 0 ? gload16_asm<0>(wp, wlane) : gload16_asm<1024>(wp, wlane))
                                              : (n == 0 ? gload16_asm<2048>(wp, wlane) : gload16_asm<3072>(wp, wlane));
                }
                wp += STEP * 16;
                {
#pragma unroll
                    for (int q = 0; q < pieces_in_tap(t); ++q) issue_piece(first_piece(t) + q, ch + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 2; ++p)                              // a0.b0, a0.b1
#pragma unroll
                    for (int m = 0; m < ML; ++m)
#pragma unroll
                        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[0][m], __builtin_bit_cast(float4, bc[p][n]), acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 9) load_a(acur[0], t + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < ML; ++m)                             // a1.b0
#pragma unroll
                    for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[1][m], __builtin_bit_cast(float4, bc[0][n]), acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // 9 is odd: the fragments of the next chunk's tap 0 are in bq[1]; the wait also covers every piece of the next raw box (older):
            // it has landed for this thread, the barrier says so for everybody -- and that everybody is done reading T
            vm_wait<0>(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int n = 0; n < NREP; ++n) bq[0][k][n] = bq[1][k][n];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            }
        }
    };
    int mlb, ml;
    auto set_ml = [&]() __attribute__((always_inline)) {
        mlb = m_lo == 0 ? m_hi : TZ;                                // live z slices of the block (workgroup-uniform) ...
        ml = min(MREP, max(0, mlb - zp * MREP));                    // ... and of this wave (wave-uniform; MS = 2: the upper pair of a 1- or 2-slice block idles through the taps)
    };
    if constexpr (!PS) set_ml();
    // copy-out of cout half n of the block's output image (LDS, 128 B per voxel) to the output tensor: pieces [it0, it1) of this thread
    constexpr int TV = TZ * TY * TX;                                // 256 voxels
    constexpr int EIT = TV * 8 / 256;                               // 16-byte pieces per thread and cout half
    unsigned char* xb = lds + grp * XB;                             // exchange buffer, then output image, of this group (PS: the T buffer that the block's last chunk has read)
    unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
    const int nco = (a.Cout + 15) / 16;
    auto copy_out = [&](int n, int it0, int it1, const int (&clo)[3], const int (&chi)[3]) __attribute__((always_inline)) {
        const int t5 = gtid >> 3, q = gtid & 7;
        const int x_lane = t5 % TX, y_lane = t5 / TX;
        const bool cok = cb * 4 + n * 2 + (q >> 2) < nco;
        unsigned char* ob = outb + ((size_t)tile * nco + cb * 4 + n * 2) * plane * 64;                        // wave-uniform
        const unsigned lane_off = (unsigned)(((size_t)(q >> 2) * plane + (size_t)y_lane * a.W + x_lane) * 64 + (q & 3) * 16);
        const int oyl = oy0 + y_lane, oxl = ox0 + x_lane;
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            if (it < it0 || it >= it1) continue;
            const int zc = it >> 1, yc = (it & 1) * (32 / TX);
            const int ozc = oz0 + zc, oy = oyl + yc;
            if (cok && ozc >= clo[0] && ozc < chi[0] && oy >= clo[1] && oy < chi[1] && oxl >= clo[2] && oxl < chi[2]) {
                const unsigned uni = (unsigned)(((ozc * a.H + oy0 + yc) * a.W + ox0) * 64);                  // wave-uniform
                float4* dstp = reinterpret_cast<float4*>(ob + (size_t)(lane_off + uni));
                const float4 val = *reinterpret_cast<const float4*>(xb + (it * 256 + gtid) * 16);
                __builtin_nontemporal_store(val.x, &dstp->x); __builtin_nontemporal_store(val.y, &dstp->y);
                __builtin_nontemporal_store(val.z, &dstp->z); __builtin_nontemporal_store(val.w, &dstp->w);
            }
        }
    };
    auto store_box = [&](int (&clo)[3], int (&chi)[3]) __attribute__((always_inline)) {      // what the copy-out writes: the tile's box cut down to what the consumer reads
#pragma unroll
        for (int i = 0; i < 3; ++i) { clo[i] = blo[i]; chi[i] = bhi[i]; }
        if (a.store_boxes) {
            const int* sb = a.store_boxes + 6 * tile;
#pragma unroll
            for (int i = 0; i < 3; ++i) { clo[i] = max(clo[i], sb[i] - a.store_grow); chi[i] = min(chi[i], sb[3 + i] + a.store_grow); }
        }
    };
    // ---- epilogue (its constants are loaded behind the chunk loop: load_epilogue_consts)
    unsigned seen = 0;
    float vmax = 0.0f;
    bool nonfinite = false;                                         // an output inside the box that is inf / NaN (an overflowed t): fmaxf would drop the NaN silently
    unsigned umax = 0;                                              // (MS = 1: the same as the largest |v| BIT PATTERN inside the box -- NaN > inf > every finite value as unsigned)
    const float relu_floor = a.relu ? 0.0f : -__builtin_inff();
    // this lane's cout column: lane & 31 of the 32-cout half n -- M16: columns 16 q + (lane & 15), q = 0, 1 (element (p, q, i) = r of the tile)
    const int col16 = lane & 15, rq16 = lane >> 4;
    float scv[NREP][M16 ? 2 : 1], shv[NREP][M16 ? 2 : 1];
    int clo[3], chi[3];
    auto load_epilogue_consts = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int q = 0; q < (M16 ? 2 : 1); ++q) {
                const int co = cb * 64 + n * 32 + (M16 ? q * 16 + col16 : row);
                scv[n][q] = co < a.Cout ? a.scale[co] : 0.0f; shv[n][q] = co < a.Cout ? a.shift[co] : 0.0f;
                asm volatile("" : "+v"(scv[n][q]), "+v"(shv[n][q]));
            }
        store_box(clo, chi);
    };
    // wave F (= frequency F during the taps) finishes z slice F
    auto finish = [&](auto ftag) __attribute__((always_inline)) {
        constexpr int F = decltype(ftag)::value;
        // Which of this lane's 16 x 2 outputs lie inside the tile's box: ONE mask register, made once per block.  (As compares inside the loop below the
        // same tests were ~100 SGPR-pair mask operations per cout half, enough of them live at once that the compiler spilled SGPRs through
        // v_writelane / v_readlane: 1450 instructions per half where ~600 do the work.  Cout % 64 == 0 for every layer of this kernel: no cout test.)
        unsigned okmask = 0;
        {
            const bool zok = oz0 + F >= blo[0] && oz0 + F < bhi[0];
            const unsigned ylen = (unsigned)(bhi[1] - blo[1]), xlen = (unsigned)(bhi[2] - blo[2]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = M16 ? 16 * (r >> 3) + 4 * rq16 + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ty = rr / NP, tx = 2 * (rr % NP);
                const bool yok = (unsigned)(oy0 + ty - blo[1]) < ylen;
#pragma unroll
                for (int e = 0; e < 2; ++e) okmask |= (yok && (unsigned)(ox0 + tx + e - blo[2]) < xlen ? 1u : 0u) << (2 * r + e);
            }
            okmask = zok ? okmask : 0u;
        }
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            OAI_WEP0();
            __syncthreads();                                          // T reads / the previous half's copy-out are done
            // send: slice m of frequency F to wave m
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                if (m == F) continue;
                const int slot = F * 3 + (m > F ? m - 1 : m);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = {acc_el(m, n, 4 * j), acc_el(m, n, 4 * j + 1), acc_el(m, n, 4 * j + 2), acc_el(m, n, 4 * j + 3)};
                    *reinterpret_cast<f32x4*>(xb + ((slot * 4 + j) * 64 + lane) * 16) = v;
                }
            }
            __syncthreads();
            // receive: slice F of the other three frequencies
            float M[4][16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == F) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) M[g][r] = acc_el(F, n, r);
                } else {
                    const int slot = g * 3 + (F > g ? F - 1 : F);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((slot * 4 + j) * 64 + lane) * 16);
                        M[g][4 * j] = v[0]; M[g][4 * j + 1] = v[1]; M[g][4 * j + 2] = v[2]; M[g][4 * j + 3] = v[3];
                    }
                }
            }
            __syncthreads();                                          // everybody has its frequencies: the buffer becomes the output image
            OAI_WEP(0);
            unsigned om = okmask;
            asm volatile("" : "+v"(om));                              // (per half: otherwise the compiler turns the bit tests into 32 compare masks in the first half and keeps them for the second)
            const int ccol = M16 ? col16 : row;                       // cout column inside its 16-column (M16) / 32-column tile
            const bool odd = ccol & 1;
            const unsigned sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // C/D row and 16-cout record of element r: 32x32 -- register r of the tile; 16x16 -- element i = r & 3 of tile (p, q) = (r >> 3, (r >> 2) & 1)
                const int rr = M16 ? 16 * (r >> 3) + 4 * rq16 + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * half;
                const int qrec = M16 ? (r >> 2) & 1 : row >> 4;
                const float sc = scv[n][M16 ? (r >> 2) & 1 : 0], sh = shv[n][M16 ? (r >> 2) & 1 : 0];
                unsigned char* lrow = xb + qrec * 64 + ((ccol & 15) >> 1) * 4;
                const int ty = rr / NP, tx = 2 * (rr % NP);
                const float y0 = (M[0][r] + M[1][r]) + M[2][r];
                const float y1 = (M[1][r] - M[2][r]) - M[3][r];
                float v[2] = {y0 * sc + sh, y1 * sc + sh};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // voxels outside the box are never copied out: their value is 0 (bitwise AND with the sign-extended mask bit) -- for the census, the
                    // finiteness test (before the ReLU: fmaxf(NaN, 0) = 0) and the fused pool
                    const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)om, 2 * r + e, 1);      // 0 or ~0 (v_bfe_i32: no compare, no SGPR mask)
                    umax = max(umax, __builtin_bit_cast(unsigned, v[e]) & keep & 0x7FFFFFFFu);
                    v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, fmaxf(v[e], relu_floor)) & keep);    // (ReLU on the computed value: no canonicalising max in front of it)
                }
                vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1])));
                unsigned w_hi, w_lo;
                split_two_voxels(v[0], v[1], sel, w_hi, w_lo);
                const int vox = (F * TY + ty) * TX + tx + (odd ? 1 : 0);
                unsigned char* dst = lrow + vox * 128;
                *reinterpret_cast<unsigned*>(dst) = w_hi;
                *reinterpret_cast<unsigned*>(dst + 32) = w_lo;
            }
            __syncthreads();
            OAI_WEP(1);
            if constexpr (PS) {
                // the last half's image is written: the accumulators are dead.  The weight fragments of the NEXT block's taps 0 .. D - 1, from the panel's first
                // tap again; they land under the copy-out and are waited for behind it -- before any control flow (unet_sres.h: asm loads and branches)
                if (n == NREP - 1) {
                    const unsigned char* wq = wp_base;                // (a local: `wp` assigned inside the branches on the wave's frequency would stop being uniform for the compiler)
                    sgpr_settle(wq);
                    set_lane_consts();
#pragma unroll
                    for (int d = 0; d < D; ++d) {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int nn = 0; nn < NREP; ++nn) bq[d][k][nn] = k == 0 ? (nn == 0 ? gload16_asm<0>(wq, wlane) : gload16_asm<1024>(wq, wlane))
                                                                                  : (nn == 0 ? gload16_asm<2048>(wq, wlane) : gload16_asm<3072>(wq, wlane));
                        wq += STEP * 16;
                    }
                }
            }
            // (WS: the staging waves, idle since the last chunk, take the other half of the pieces.  PS, measured: ALL pieces of the last half on the staging waves, so
            //  that the multipliers' wait for the next block's fragments below is not a wait for copy-out stores as well: +1 ms per pass -- the end barrier then waits
            //  for the staging waves' sixteen pieces)
            copy_out(n, 0, WS ? EIT / 2 : EIT, clo, chi);
            if constexpr (PS) {
                if (n == NREP - 1)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0][0]), "+v"(bq[0][0][1]), "+v"(bq[0][1][0]), "+v"(bq[0][1][1]),
                                 "+v"(bq[D - 1][0][0]), "+v"(bq[D - 1][0][1]), "+v"(bq[D - 1][1][0]), "+v"(bq[D - 1][1][1]) :: "memory");
            }
            if constexpr (TY == 8 && NP == 4) if (a.pool_out) {       // (main shape only: the host asks for it where the box is whole blocks of it)
                // MaxPool3d(2) fused (ec3 / ec5; networks.py:117,122), from the block's image: thread = (pooled voxel, 4 channels of one of the two
                // 16-channel records); the pooled record keeps the (h0, h1) PAIR of the window's largest joined value -- the rule of
                // maxpool2_sres_kernel (on equal values the pair with the larger h0: what splitting the fp32 maximum would have produced).
                // Block origins are even (host) and blocks lie inside the tile: every window is whole.
                static_assert(TZ * TY * TX / 8 == 32, "32 pooled voxels x 8 quarter records = the 256 threads of a cout group");
                const int pv = gtid >> 3, q = gtid & 7;
                const int px = pv % (TX / 2), py = (pv / (TX / 2)) % (TY / 2), pz = pv / ((TX / 2) * (TY / 2));
                u16x4 mh, ml;
                {
                    // The host takes this kernel for a pooled layer only behind a ReLU: every record is >= +0 (h0 >= +0, h0 + h1 >= +0, no -0: fmaxf(x, +0),
                    // x - x = +0), and for such values "larger joined value, then larger h0" is the unsigned order of (bits of the joined fp32 : bits of h0)
                    // -- one 64-bit compare and two selects per candidate instead of three float compares, their mask logic and three selects.  (Equal
                    // joined value and equal h0 means equal h1.)
                    unsigned long long mkey[4];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int vox = ((2 * pz + (k >> 2)) * TY + 2 * py + ((k >> 1) & 1)) * TX + 2 * px + (k & 1);
                        const unsigned char* rec = xb + vox * 128 + (q >> 2) * 64 + (q & 3) * 8;
                        const u16x4 h = *reinterpret_cast<const u16x4*>(rec), l = *reinterpret_cast<const u16x4*>(rec + 32);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned, join2_f16(h[j], l[j])) << 32) | ((unsigned)h[j] << 16) | l[j];
                            mkey[j] = k == 0 ? key : max(mkey[j], key);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { mh[j] = (unsigned short)(mkey[j] >> 16); ml[j] = (unsigned short)mkey[j]; }
                }
                const int Dp = a.D / 2, Hp = a.H / 2, Wp = a.W / 2;
                const int gz = oz0 / 2 + pz, gy = oy0 / 2 + py, gx = ox0 / 2 + px;
                if (cb * 4 + n * 2 + (q >> 2) < nco && gz < Dp && gy < Hp && gx < Wp) {
                    unsigned char* dst = reinterpret_cast<unsigned char*>(a.pool_out) +
                                         srec(tile, nco, (size_t)Dp * Hp * Wp, cb * 4 + n * 2 + (q >> 2), ((size_t)gz * Hp + gy) * Wp + gx) + (q & 3) * 8;
                    *reinterpret_cast<u16x4*>(dst) = mh;
                    *reinterpret_cast<u16x4*>(dst + 32) = ml;
                }
            }
            OAI_WEP(2);
        }
    };
    // MS = 2: wave (zp, F) holds frequency F of the units e = (local slice s, cout half n) = 2 s + n of its slice pair; it finishes unit F.
    // One exchange (48 KB per slice pair), then ONE image of the whole block, 256 B per voxel (four 16-cout records), copied out by all 512 threads.
    auto finish2 = [&](auto ftag) __attribute__((always_inline)) {
        constexpr int F = decltype(ftag)::value;
        constexpr int S = F >> 1, N = F & 1;
        unsigned char* const xz = lds + zp * XB;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e == F) continue;
            constexpr int kDummy = 0; (void)kDummy;
            const int slot = F * 3 + (e > F ? e - 1 : e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = {acc_el(e >> 1, e & 1, 4 * j), acc_el(e >> 1, e & 1, 4 * j + 1), acc_el(e >> 1, e & 1, 4 * j + 2), acc_el(e >> 1, e & 1, 4 * j + 3)};
                *reinterpret_cast<f32x4*>(xz + ((slot * 4 + j) * 64 + lane) * 16) = v;
            }
        }
        __syncthreads();
        float M[4][16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g == F) {
#pragma unroll
                for (int r = 0; r < 16; ++r) M[g][r] = acc_el(S, N, r);
            } else {
                const int slot = g * 3 + (F > g ? F - 1 : F);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xz + ((slot * 4 + j) * 64 + lane) * 16);
                    M[g][4 * j] = v[0]; M[g][4 * j + 1] = v[1]; M[g][4 * j + 2] = v[2]; M[g][4 * j + 3] = v[3];
                }
            }
        }
        __syncthreads();                                              // both slice pairs have their frequencies: the buffers become the output image
        const int ccol = M16 ? col16 : row;                           // cout column inside its 16-column (M16) / 32-column tile (see finish)
        const bool odd = ccol & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;
        const int zs = zp * 2 + S;
        const int oz = oz0 + zs;
        const bool zok = oz >= blo[0] && oz < bhi[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = M16 ? 16 * (r >> 3) + 4 * rq16 + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * half;
            const int qrec = M16 ? (r >> 2) & 1 : row >> 4;
            const bool cvalid = cb * 64 + N * 32 + qrec * 16 + (ccol & 15) < nco * 16;
            const float sc = scv[N][M16 ? (r >> 2) & 1 : 0], sh = shv[N][M16 ? (r >> 2) & 1 : 0];
            unsigned char* lrow = lds + (N * 2 + qrec) * 64 + ((ccol & 15) >> 1) * 4;
            const int ty = rr / NP, tx = 2 * (rr % NP);
            const float y0 = (M[0][r] + M[1][r]) + M[2][r];
            const float y1 = (M[1][r] - M[2][r]) - M[3][r];
            float v[2] = {y0 * sc + sh, y1 * sc + sh};
            const int oy = oy0 + ty;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ox = ox0 + tx + e;
                const bool ok = cvalid && zok && ox >= blo[2] && ox < bhi[2] && oy >= blo[1] && oy < bhi[1];
                nonfinite |= ok && !(fabsf(v[e]) <= 3.0e38f);      // (before the ReLU: fmaxf(NaN, 0) = 0)
                if (a.relu) v[e] = fmaxf(v[e], 0.0f);
                v[e] = ok ? v[e] : 0.0f;
            }
            vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1])));
            unsigned w_hi, w_lo;
            split_two_voxels(v[0], v[1], sel, w_hi, w_lo);
            const int vox = (zs * TY + ty) * TX + tx + (odd ? 1 : 0);
            unsigned char* dst = lrow + vox * 256;
            *reinterpret_cast<unsigned*>(dst) = w_hi;
            *reinterpret_cast<unsigned*>(dst + 32) = w_lo;
        }
        __syncthreads();
        {
            const int t6 = tid >> 4, q = tid & 15;
            const int x_lane = t6 % TX, y_lane = t6 / TX;
            const bool cok = cb * 4 + (q >> 2) < nco;
            unsigned char* ob = outb + ((size_t)tile * nco + cb * 4) * plane * 64;                                    // wave-uniform
            const unsigned lane_off = (unsigned)(((size_t)(q >> 2) * plane + (size_t)y_lane * a.W + x_lane) * 64 + (q & 3) * 16);   // 4 * plane * 64 < 2^32 (host check)
            const int oyl = oy0 + y_lane, oxl = ox0 + x_lane;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int zc = it >> 1, yc = (it & 1) * (32 / TX);
                const int ozc = oz0 + zc, oy = oyl + yc;
                if (cok && ozc >= clo[0] && ozc < chi[0] && oy >= clo[1] && oy < chi[1] && oxl >= clo[2] && oxl < chi[2]) {
                    const unsigned uni = (unsigned)(((ozc * a.H + oy0 + yc) * a.W + ox0) * 64);                      // wave-uniform
                    float4* dstp = reinterpret_cast<float4*>(ob + (size_t)(lane_off + uni));
                    const float4 val = *reinterpret_cast<const float4*>(lds + (it * 512 + tid) * 16);
                    __builtin_nontemporal_store(val.x, &dstp->x); __builtin_nontemporal_store(val.y, &dstp->y);
                    __builtin_nontemporal_store(val.z, &dstp->z); __builtin_nontemporal_store(val.w, &dstp->w);
                }
            }
        }
    };
    auto finish_block = [&]() __attribute__((always_inline)) {
        if constexpr (MS == 1) {
            if (f == 0) finish(std::integral_constant<int, 0>{});
            else if (f == 1) finish(std::integral_constant<int, 1>{});
            else if (f == 2) finish(std::integral_constant<int, 2>{});
            else finish(std::integral_constant<int, 3>{});
        } else {
            if (f == 0) finish2(std::integral_constant<int, 0>{});
            else if (f == 1) finish2(std::integral_constant<int, 1>{});
            else if (f == 2) finish2(std::integral_constant<int, 2>{});
            else finish2(std::integral_constant<int, 3>{});
        }
    };
    if constexpr (PS) {
        // ---- PS: the persistent driver.  Control words (LDS, behind the last staging wave's pieces): two slots of {tile, bz, by, bx, cbg, valid}
        int* const ctl = reinterpret_cast<int*>(raw + RAWB - 64);
        int* const planw = a.ps_plan;                               // [0..7] the XCDs' counters, [8] blocks, [9] tiles, [16 + t] blocks in front of tile t, [288 + 8 t] tile t's {bz0, nbz, by0, nby, bx0, nbx}
        const int ps_home = (int)blockIdx.x & 7;                    // (workgroups are dealt round-robin to the XCDs: neighbours in the list meet in one L2)
        // The next block of this workgroup -> ctl[slot]: an eighth of the list per XCD, pulled in order; an XCD that runs dry takes from the next one.  The
        // fetch is a chain of three dependent memory round trips (counter, prefix table, the tile's sub-box); the fetching wave walks it ONE STEP PER CHUNK,
        // behind its staging work: every step's loads are waited for by the vmcnt(0) that opens the next chunk's staging anyway.
        int f_state = 0, f_tries = 0, f_k = 0, f_g = -1, f_p[4] = {0, 0, 0, 0}, f_tl = 0, f_p0 = 0, f_sb[6] = {0, 0, 0, 0, 0, 0};
        auto fetch_issue_atomic = [&]() __attribute__((always_inline)) {
            const int x = (ps_home + f_tries) & 7;
            f_k = lane == 0 ? atomicAdd(planw + x, 1) : 0;
        };
        // one step; returns true when ctl[slot] has been written
        auto fetch_step = [&](int slot) __attribute__((always_inline)) -> bool {
            const int n = planw[8], nt = planw[9];
            if (f_state == 0) { fetch_issue_atomic(); f_state = 1; return false; }
            if (f_state == 1) {
                const int x = (ps_home + f_tries) & 7;
                const int st = (int)(((long long)n * x) >> 3), en = (int)(((long long)n * (x + 1)) >> 3);
                const int k = __builtin_amdgcn_readfirstlane(f_k);
                if (k < en - st) {
                    f_g = st + k;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int t = q * 64 + lane; f_p[q] = t < nt ? planw[16 + t] : 0x7FFFFFFF; }
                    f_state = 2;
                    return false;
                }
                if (++f_tries < 8) { fetch_issue_atomic(); return false; }
                f_g = -1; f_state = 3;                               // every range is dry: no block
            }
            if (f_state == 2) {
                int cnt = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) cnt += __popcll(__builtin_amdgcn_ballot_w64(f_p[q] <= f_g));
                f_tl = cnt - 1;
                f_p0 = planw[16 + f_tl];
#pragma unroll
                for (int i = 0; i < 6; ++i) f_sb[i] = planw[288 + f_tl * 8 + i];
                f_state = 3;
                return false;
            }
            int o[6] = {0, 0, 0, 0, 0, 0};
            if (f_g >= 0) {
                int r = f_g - f_p0;
                o[4] = r % a.ncb; r /= a.ncb;
                o[3] = f_sb[4] + r % f_sb[5]; r /= f_sb[5];
                o[2] = f_sb[2] + r % f_sb[3];
                o[1] = f_sb[0] + r / f_sb[3];
                o[0] = f_tl; o[5] = 1;
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) ctl[slot * 8 + i] = o[i];
            }
            f_state = 0;
            return true;
        };
        int nb[6];
        auto read_ctl = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 6; ++i) nb[i] = __builtin_amdgcn_readfirstlane(ctl[slot * 8 + i]);
        };
        const bool fetcher = __builtin_amdgcn_readfirstlane(wave) == 4;
        if (fetcher) { while (!fetch_step(0)) { } }
        __syncthreads();
        read_ctl(0);
        if (!nb[5]) return;                                         // (more workgroups than blocks)
        set_block(nb[0], nb[1], nb[2], nb[3], nb[4]);
        seen = census_peek(a.census);
        // the first block's prologue: as in the plain form
        if (stager) {
            stage_plan_fast(woff, oz0, oy0, ox0);
            stage_request(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage_transform_to(Tl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (nchunks > 1) stage_request(1);
        } else {
            sgpr_settle(wp);
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) bq[d][k][n] = k == 0 ? (n == 0 ? gload16_asm<0>(wp, wlane) : gload16_asm<1024>(wp, wlane))
                                                                        : (n == 0 ? gload16_asm<2048>(wp, wlane) : gload16_asm<3072>(wp, wlane));
                wp += STEP * 16;
            }
            vm_wait<0>(bq[D - 1][0][0], bq[D - 1][0][1], bq[D - 1][1][0], bq[D - 1][1][1]);
        }
        __syncthreads();
        int slot = 0;
        int nbox[6];                                                // a.boxes[next tile], loaded a phase before the switch
        auto load_nbox = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 6; ++i) nbox[i] = a.boxes ? a.boxes[6 * nb[0] + i] : (i < 3 ? a.lo[i] : a.hi[i - 3]);
        };
        // (two loops, one per role: in ONE loop with a branch on the role everything a role keeps across blocks is live through the other role's code too)
        if (stager) {
            for (;;) {
                const int pe = (pb + nchunks) & 1;                   // the T buffer of the NEXT block's chunk 0; the other one is free behind the last chunk: the epilogue's buffer
                bool have_next = false, fetched = false;
                // One block ahead: chunk ch + 1 of this block is transformed while the multipliers run chunk ch; behind the block's last chunk come chunk 0 of the
                // NEXT block (requested during chunk nchunks - 2, transformed during chunk nchunks - 1) and the request of its chunk 1.  (host: nchunks >= 6)
                unsigned woffN[WS ? WNIT : 1];
                const unsigned char *s0N = nullptr, *s1N = nullptr;
                for (int ch = 0; ch < nchunks; ++ch) {
                    unsigned char* const tgt = Tl + ((pb + ch + 1) & 1) * TB;
                    if (ch == nchunks - 2) {                         // the next block is known (written by chunk nchunks - 3 at the latest, one barrier ago)
                        read_ctl(slot ^ 1);
                        have_next = nb[5] != 0;
                    }
                    if (ch + 1 < nchunks) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        stage_transform_to(tgt);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (ch + 2 < nchunks) stage_request(ch + 2);
                        else if (have_next) {
                            stage_plan_fast(woffN, a.lo[0] + nb[1] * TZ, a.lo[1] + nb[2] * TY, a.lo[2] + nb[3] * TX);
                            s0N = reinterpret_cast<const unsigned char*>(a.src0) + srec(nb[0], nch0, plane, 0, 0);
                            s1N = reinterpret_cast<const unsigned char*>(a.src1) + srec(nb[0], nch1, plane, 0, 0);
                            stage_request_of(woffN, s0N, s1N, 0);
                            load_nbox();
                        }
                    } else if (have_next) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        stage_transform_to(tgt);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        stage_request_of(woffN, s0N, s1N, 1);
                    }
                    if (fetcher && !fetched) {                       // (behind this wave's staging of the chunk: the multipliers are still in their taps)
                        fetched = fetch_step(slot ^ 1);
                        if (ch >= nchunks - 3) { while (!fetched) fetched = fetch_step(slot ^ 1); }      // (the deadline: only when an XCD's range ran dry on the way)
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    OAI_PSTAMP(0);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    OAI_PSTAMP(4);
                }
                // the multipliers' epilogue: its four barriers per cout half, then this wave's half of the image's copy-out
                xb = Tl + (pe ^ 1) * TB;
                int sclo[3], schi[3];
                store_box(sclo, schi);
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    for (int i = 0; i < 4; ++i) __syncthreads();
                    copy_out(n, EIT / 2, EIT, sclo, schi);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image has been read: the next transform may overwrite it
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                OAI_PSTAMP(5);
#ifdef OAI_DIAG
                if (!have_next && a.stamps && fetcher && lane == 0) { atomicAdd(a.stamps + 4, pst[4]); atomicAdd(a.stamps + 5, pst[5]); atomicAdd(a.stamps + 6, pst[0]); }
#endif
                if (!have_next) return;
                set_block(nb[0], nb[1], nb[2], nb[3], nb[4], nbox);
#pragma unroll
                for (int it = 0; it < (WS ? WNIT : 1); ++it) woff[it] = woffN[it];
                pb = pe; slot ^= 1;
            }
        }
        for (;;) {
            const int pe = (pb + nchunks) & 1;
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int n = 0; n < NREP; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
            set_ml();
            OAI_PSTAMP(0);
            if (ml == 4) run_chunks(std::integral_constant<int, 4>{});
            else if (ml == 3) run_chunks(std::integral_constant<int, 3>{});
            else if (ml == 2) run_chunks(std::integral_constant<int, 2>{});
            else run_chunks(std::integral_constant<int, 1>{});
            // (the fragments requested past the last tap: see the plain form)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0][0]), "+v"(bq[0][0][1]), "+v"(bq[0][1][0]), "+v"(bq[0][1][1]), "+v"(bq[1][0][0]), "+v"(bq[1][0][1]), "+v"(bq[1][1][0]), "+v"(bq[1][1][1]),
                         "+v"(bq[NB - 1][0][0]), "+v"(bq[NB - 1][0][1]), "+v"(bq[NB - 1][1][0]), "+v"(bq[NB - 1][1][1]) :: "memory");
            read_ctl(slot ^ 1);                                      // (written before the barrier of chunk nchunks - 3)
            const bool have_next = nb[5] != 0;
            if (have_next) load_nbox();                              // (needed behind the epilogue)
            xb = Tl + (pe ^ 1) * TB;
            load_epilogue_consts();
            finish_block();                                          // (requests the next block's first fragment sets and waits for them)
            wp = wp_base + D * STEP * 16;
            sgpr_settle(wp);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            OAI_PSTAMP(2);
#ifdef OAI_DIAG
            ++psblocks;
#endif
            if (!have_next) break;
            set_block(nb[0], nb[1], nb[2], nb[3], nb[4], nbox);
            pb = pe; slot ^= 1;
        }
#ifdef OAI_DIAG
        if (a.stamps && wave == 0 && lane == 0) {
            atomicAdd(a.stamps + 0, pst[0]); atomicAdd(a.stamps + 1, pst[1]); atomicAdd(a.stamps + 2, pst[2]); atomicAdd(a.stamps + 3, pst[3]);
            atomicAdd(a.stamps + 8, psblocks);
        }
#endif
    } else {
        if (stager) {
            // WS, waves 4-7: while the multipliers run the taps of chunk ch, transform the rows of chunk ch + 1 (requested a chunk ago: the wait is
            // short) into the other T buffer, then request chunk ch + 2 into the same rows; one barrier per chunk, the multipliers' chunk-end one
            for (int ch = 0; ch < nchunks; ++ch) {
                if (ch + 1 < nchunks) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    stage_transform(ch + 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ch + 2 < nchunks) stage_request(ch + 2);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // the multipliers' epilogue: its four barriers per cout half, then this wave's half of the image's copy-out (the next half's first barrier
            // waits for these LDS reads before the exchange overwrites the image)
            int sclo[3], schi[3];
            store_box(sclo, schi);
    #pragma unroll
            for (int n = 0; n < NREP; ++n) {
                for (int i = 0; i < 4; ++i) __syncthreads();
                copy_out(n, EIT / 2, EIT, sclo, schi);
            }
            return;
        }
        if constexpr (MS == 1) {
            if (ml == 4) run_chunks(std::integral_constant<int, 4>{});
            else if (ml == 3) run_chunks(std::integral_constant<int, 3>{});
            else if (ml == 2) run_chunks(std::integral_constant<int, 2>{});
            else run_chunks(std::integral_constant<int, 1>{});
        } else {
            if (ml == 2) run_chunks(std::integral_constant<int, 2>{});
            else if (ml == 1) run_chunks(std::integral_constant<int, 1>{});
            else run_chunks(std::integral_constant<int, 0>{});
        }

        // fragments requested past the last tap / step: never used, but still in flight INTO their registers -- the wait names them, so that they stay
        // allocated until it has executed (a bare wait lets the compiler reuse the "dead" registers between the loop exit and the wait: unet_sres.h)
        if constexpr (WS && M16) {
            if (!stager) { ws16_landed(fs[0]); ws16_landed(fs[1]); }
        } else if constexpr (D > 1) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0][0]), "+v"(bq[0][0][1]), "+v"(bq[0][1][0]), "+v"(bq[0][1][1]), "+v"(bq[1][0][0]), "+v"(bq[1][0][1]), "+v"(bq[1][1][0]), "+v"(bq[1][1][1]),
                         "+v"(bq[NB - 1][0][0]), "+v"(bq[NB - 1][0][1]), "+v"(bq[NB - 1][1][0]), "+v"(bq[NB - 1][1][1]) :: "memory");
        }
        OAI_WSTAMP(2);
        seen = census_peek(a.census);                               // (waited for under the epilogue)
        load_epilogue_consts();
        finish_block();
    }
    if (nonfinite || umax > __builtin_bit_cast(unsigned, 3.0e38f)) atomicOr(a.range_flag, 1);
    census_note(a.census, a.range_flag, vmax, seen);
#ifdef OAI_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (the stamped epilogue includes the round trip of its last stores)
    OAI_WSTAMP(3);
    if (!PS && a.stamps && lane == 0) {
        atomicAdd(a.stamps + 0, wst[1] - wst[0]); atomicAdd(a.stamps + 1, wst[2] - wst[1]); atomicAdd(a.stamps + 2, wst[3] - wst[2]);
        atomicAdd(a.stamps + 3, wep[0]); atomicAdd(a.stamps + 4, wep[1]); atomicAdd(a.stamps + 5, wep[2]);
        atomicAdd(a.stamps + 8, 1ull);
        atomicMax(a.stamps + 10, ~wst[0]); atomicMax(a.stamps + 11, wst[3]);   // [10] holds the complement of the earliest start
    }
#endif
#undef OAI_WSTAMP
#undef OAI_PSTAMP
#undef OAI_WEP0
#undef OAI_WEP
}

}  // namespace oai
