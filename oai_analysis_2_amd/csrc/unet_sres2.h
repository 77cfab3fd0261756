// conv3_igemm_sres2: the split-resident 3x3x3 conv for layers with Cout % 128 == 0 -- a workgroup that OWNS its CU.
//
// conv3_igemm_sres (unet_sres.h) runs two 4-wave workgroups per CU, each with its own 68-KB halo box: a chunk is
// [barrier] [issue 17 LDS-DMA pieces] [wait for them] [barrier] [27 taps], and a wave spends 39 % of its life outside the taps
// (profiles/r02_conv_phases.md) -- blocked in vector-memory issue while its partner workgroup, if it happens to be in its taps, has
// the matrix pipe to itself at 74-89 % of its rate.  For Cout >= 128 the two workgroups of a CU are moreover often the two cout
// blocks of ONE spatial block and stage the same halo twice.
//
// Here ONE workgroup of 8 waves holds the CU: waves 0-3 compute couts [0, 64) of the block, waves 4-7 couts [64, 128), from ONE halo
// box that is DOUBLE-BUFFERED (2 x 72 KB of the CU's 160 KB): the pieces of chunk c+1 are requested one per tap during the first
// nine taps of chunk c, by all eight waves, and have the rest of the chunk to land.  A chunk is [one barrier] [27 taps]: both waves
// of every SIMD are in their taps all the time, the halo is fetched once per 128 couts, and the chunk loop contains no wait for
// staging at all.
//
// vmcnt is an in-order counter, so everything the chunk loop loads is issued from inline assembly and waited for with hand-counted
// s_waitcnt (the compiler's own waits know nothing of asm loads: it would wait vmcnt(0) for its weight fragments, i.e. for the DMA
// piece requested a moment ago as well -- see lds_dma16 in unet_sres.h).  Order inside tap t: wait for B(t) -- at most one younger
// op, the DMA piece of tap t-1, may stay in flight --, request B(t+1), request one DMA piece (taps 0-8), MFMAs.  A piece therefore
// has two taps (~3 k cycles) before an in-order wait can stall on it.
//
// Same arithmetic in the same order as conv3_igemm_sres (chunks, taps, passes a0.b0, a0.b1, a1.b0; same epilogue code): results are
// bit-identical, which tests/test_unet_gpu.py asserts with option "wide" 0 / 1.
#pragma once
#include "unet_sres.h"

namespace oai {

// 16 bytes per lane from global memory, from inline assembly: invisible to the compiler's wait-count pass; the consumer must pass the
// registers through vm_wait<N>() first.
typedef float f32x4 __attribute__((ext_vector_type(4)));          // a native vector: inline asm can tie it to a 128-bit VGPR tuple (HIP's float4 is a struct)
// address = wave-uniform base (SGPR pair) + per-lane 32-bit offset + immediate: no per-load 64-bit VALU address arithmetic, no address VGPRs
template <int IMM>
__device__ __forceinline__ f32x4 gload16_asm(const void* sbase, unsigned voff) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
    return v;
}
// "VALU writes an SGPR -> VMEM reads it" needs five wait states, and the compiler's hazard recognizer does not look inside inline assembly:
// a uniform base made by v_readfirstlane (or reloaded from an SGPR spill lane by v_readlane) right in front of gload16_asm would be read
// STALE -- a wild address (conv3_wino_sres faulted that way on some layers).  sgpr_settle(base) in front of the first load behind such a
// write; the build refuses a library in which any candidate is left (build.py: check_no_sgpr_hazard, scripts/sgpr_hazard_scan.py).
__device__ __forceinline__ void sgpr_settle(const unsigned char*& sbase) { asm volatile("s_nop 4" : "+s"(sbase) :: "memory"); }     // (tied to the SGPR pair: its producer stays in front, its readers behind)
// s_waitcnt vmcnt(N) that "produces" the four weight fragments: every later use of them is ordered behind the wait
template <int N>
__device__ __forceinline__ void vm_wait(f32x4& b0, f32x4& b1, f32x4& b2, f32x4& b3) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N) : "memory");
}

// VAR (diagnostic builds pick it with OAI_WIDE_VAR; production = kWideVar): where the DMA pieces of the next box are requested and how far
// ahead the weight fragments are --
//   0  one piece per tap in taps 0..8, fragments one tap ahead        1  no DMA after the prologue (TIMING ONLY: wrong results)
//   2  one piece every third tap (0, 3, .., 24), one tap ahead        3  one piece per tap in taps 0..8, fragments TWO taps ahead
//   4  = 1 and no weight-fragment loads either (TIMING ONLY)           5  = 1 with the second cout group started 384 cycles late per chunk (TIMING ONLY)
//   6  = 0 with the second cout group started 384 cycles late per chunk     7  = 3 without DMA (TIMING ONLY)
//   9 / 10  = 0 with the MFMAs of the first two passes ordered (pass, n, m) / (m, pass, n): which operand stays put
//   8  = 1 with the weight pointer never advanced: the same 4 KiB every tap, L1-resident (TIMING ONLY)
template <int RX, int RY, int WY, int WX, int VAR>
__global__ void __launch_bounds__(512, 1) conv3_igemm_sres2(const ConvArgs a, const unsigned char* __restrict__ zero_rec) {
    static_assert(RX * RY == 32 && WY * WX == 4, "bad tile shape");
    constexpr int MREP = 4, NREP = 2, TZ = MREP;
    constexpr int kTY = WY * RY, kTX = WX * RX, HY = kTY + 2, HX = kTX + 2, HZ = TZ + 2;
    constexpr int HVOX = HZ * HY * HX;
    constexpr int PIECES = HVOX * 4;                             // 16-byte slots of the halo box
    constexpr int NIT = (PIECES + 511) / 512;                    // LDS-DMA instructions per thread per chunk (9 for 6 x 10 x 18)
    constexpr int BUF = NIT * 512 * 16;                          // bytes of one halo buffer (whole 1-KiB wave writes)
    static_assert(NIT <= 10 && 2 * BUF <= 160 * 1024, "one piece per tap in taps 0..NIT-1 (< 27); two halo buffers must fit the CU's LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;   // cout group, wave and thread inside the group
    int id = a.xcd_group ? xcd_block_id(a.nblocks, a.xcd_group) : (int)blockIdx.x;
    if (id < 0) return;
    const int cb2 = id % a.ncb; id /= a.ncb;                     // a.ncb = Cout / 128 for this kernel
    const int bx = id % a.nbx; id /= a.nbx;
    const int by = id % a.nby; id /= a.nby;
    const int bz = id % a.nbz; id /= a.nbz;
    const int tile = id;
    const int cb = cb2 * 2 + grp;                                 // this group's block of 64 couts
    const int oz0 = a.lo[0] + bz * TZ, oy0 = a.lo[1] + by * kTY, ox0 = a.lo[2] + bx * kTX;
    int blo[3], bhi[3];
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    if (oz0 >= bhi[0] || oz0 + TZ <= blo[0] || oy0 >= bhi[1] || oy0 + kTY <= blo[1] || ox0 >= bhi[2] || ox0 + kTX <= blo[2]) return;

    const int wy = gw / WX, wx = gw % WX;
    const int row = lane & 31, half = lane >> 5;
    const int lx = wx * RX + row % RX, ly = wy * RY + row / RX;
    const int m_lo = max(0, blo[0] - oz0), m_hi = min(MREP, bhi[0] - oz0);

    f32x16 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int nch0 = (a.C0 + 15) / 16, nch1 = (a.C1 + 15) / 16, nchunks = nch0 + nch1;
    const size_t plane = (size_t)a.D * a.H * a.W;
    const unsigned char* s0 = reinterpret_cast<const unsigned char*>(a.src0) + srec(tile, nch0, plane, 0, 0);
    const unsigned char* s1 = reinterpret_cast<const unsigned char*>(a.src1) + srec(tile, nch1, plane, 0, 0);

    // staging plan (see conv3_igemm_sres): piece P = it * 512 + tid = slot (P & 3) of halo record P >> 2.  The 32-bit byte offset of a
    // piece inside a chunk plane is RECOMPUTED when the piece is requested (~20 VALU, nine times per chunk) instead of being held in
    // nine VGPRs for the whole kernel: the registers go to the second set of weight fragments.
    constexpr unsigned kNoPiece = 0xFFFFFFFFu;
    const int pslot = tid & 3;
    auto piece_off = [&](int it) __attribute__((always_inline)) -> unsigned {
        const int r = (it * 512 + tid) >> 2;
        const int hx = r % HX, t2 = r / HX, hy = t2 % HY, hz = t2 / HY;
        const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
        const bool ok = r < HVOX && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        return ok ? ((unsigned)((gz * a.H + gy) * a.W + gx) << 6) | (unsigned)((pslot ^ ((hx >> 2) & 3)) << 4) : kNoPiece;
    };
    const unsigned lds0 = lds_addr_of(lds);
    // one piece of chunk `ch` into buffer `bufsel`; past the last chunk the zero record is fetched instead, so that every chunk
    // issues the same number of vector-memory operations (the counted waits below depend on it)
    auto issue_piece = [&](int it, int ch, int bufsel) __attribute__((always_inline)) {
        const bool real = ch < nchunks;
        const bool first = ch < nch0;
        const unsigned char* cbase = (first ? s0 : s1) + (size_t)(first ? ch : ch - nch0) * plane * 64;     // wave-uniform chunk plane
        const unsigned po = piece_off(it);
        const unsigned char* g = (real && po != kNoPiece) ? cbase + po : zero_rec;
        lds_dma16(g, __builtin_amdgcn_readfirstlane(lds0 + bufsel * BUF + (it * 512 + wave * 64) * 16));
    };

    // A fragments: byte offset of this lane's voxel for tap (0,0,0) inside a buffer, and its swizzled slot per dx and term
    const int aofs = (ly * HX + lx) * 64;
    int sl[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int t = 0; t < 2; ++t) sl[dx][t] = ((t * 2 + half) ^ (((lx + dx) >> 2) & 3)) * 16;

    constexpr int STEP = 2 * NREP * 64;                             // 16-byte units of weights per tap: [term][nr][lane]
    // wave-uniform (cb comes from the wave index): forced into an SGPR pair
    const size_t wp_v = (size_t)(a.wpanel + (size_t)cb * nchunks * 27 * STEP);
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(
        ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(wp_v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp_v));
    const unsigned wlane = lane * 16;

    // piece schedule: piece_at(t) = index of the piece requested in tap t, or -1
    constexpr int D = (VAR == 3 || VAR == 7) ? 2 : 1;                              // taps between the request of a weight fragment and its use
    constexpr int NB = D + 1;                                        // fragment register sets (27 % 3 == 0: with three sets the set of a tap does not depend on the chunk)
    auto piece_at = [](int t) constexpr { return (VAR == 1 || VAR == 4 || VAR == 5 || VAR == 7 || VAR == 8) ? -1 : VAR == 2 ? (t % 3 == 0 && t / 3 < NIT ? t / 3 : -1) : (t < NIT ? t : -1); };
    // ---- prologue: the whole first halo box, then the weight fragments of the first D taps
#pragma unroll
    for (int it = 0; it < NIT; ++it) issue_piece(it, 0, 0);
    f32x4 bq[NB][2][NREP];                                          // [tap % NB][term][n]
    sgpr_settle(wp);                                                // wp has just been made uniform by v_readfirstlane
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
    __syncthreads();

    auto run_chunks = [&](auto ml_tag) __attribute__((always_inline)) {
        constexpr int ML = decltype(ml_tag)::value;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int cur = ch & 1;
            const unsigned char* abase = lds + cur * BUF + aofs;
            auto load_a = [&](float4 (&dst)[MREP], int t, int k) __attribute__((always_inline)) {
                const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
#pragma unroll
                for (int m = 0; m < ML; ++m)
                    dst[m] = *reinterpret_cast<const float4*>(abase + (((m + dz) * HY + dy) * HX + dx) * 64 + sl[dx][k]);
            };
            float4 acur[2][MREP];
            load_a(acur[0], 0, 0);
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                // fragment sets: D == 1: parity of the tap, with a copy at the chunk end (27 is odd); D == 2: t % 3
                f32x4 (&bc)[2][NREP] = bq[D == 1 ? (t & 1) : t % 3];
                f32x4 (&bn)[2][NREP] = bq[D == 1 ? ((t + 1) & 1) : (t + 2) % 3];
                // B(t) was requested in tap t-D (of the previous chunk for t < D).  Younger operations that may stay in flight: the
                // fragments of the taps in between (4 each) and the pieces requested in taps t-D .. t-1 (behind the fragments there)
                {
                    constexpr int kDummy = 0; (void)kDummy;
                    int younger = 4 * (D - 1);
                    for (int u = t - D; u < t; ++u) younger += (u >= 0 && piece_at(u) >= 0) ? 1 : 0;      // (pieces are requested in taps 0..24 at most: none wraps from the previous chunk)
                    if (younger == 0) vm_wait<0>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                    else if (younger == 1) vm_wait<1>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                    else if (younger == 4) vm_wait<4>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                    else if (younger == 5) vm_wait<5>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                    else vm_wait<6>(bc[0][0], bc[0][1], bc[1][0], bc[1][1]);
                }
                load_a(acur[1], t, 1);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int n = 0; n < NREP; ++n)
                        bn[k][n] = VAR == 4 ? bc[k][n] : k == 0 ? (n == 0 ? gload16_asm<0>(wp, wlane) : gload16_asm<1024>(wp, wlane))
                                                                : (n == 0 ? gload16_asm<2048>(wp, wlane) : gload16_asm<3072>(wp, wlane));
                if constexpr (VAR != 8) wp += STEP * 16;
                if (piece_at(t) >= 0) issue_piece(piece_at(t), ch + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (VAR == 9) {                                // operand-order experiment: B fixed for four MFMAs
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int n = 0; n < NREP; ++n)
#pragma unroll
                            for (int m = 0; m < ML; ++m) acc[m][n] = mfma_16bit<true>(acur[0][m], __builtin_bit_cast(float4, bc[p][n]), acc[m][n]);
                } else if constexpr (VAR == 10) {                        // A fixed for four MFMAs (per accumulator still a0.b0 before a0.b1)
#pragma unroll
                    for (int m = 0; m < ML; ++m)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[0][m], __builtin_bit_cast(float4, bc[p][n]), acc[m][n]);
                } else {
#pragma unroll
                for (int p = 0; p < 2; ++p)                              // a0.b0, a0.b1
#pragma unroll
                    for (int m = 0; m < ML; ++m)
#pragma unroll
                        for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[0][m], __builtin_bit_cast(float4, bc[p][n]), acc[m][n]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 27) load_a(acur[0], t + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < ML; ++m)                             // a1.b0
#pragma unroll
                    for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_16bit<true>(acur[1][m], __builtin_bit_cast(float4, bc[0][n]), acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (D == 1) {
                // 27 is odd: the fragments of the next chunk's tap 0 were requested into bq[1] at the top of tap 26 (24 MFMAs ago: an L2 hit
                // has landed); wait for them, then move them to bq[0] (16 register moves per chunk).  The wait also covers every piece of
                // the next box: it has landed for this thread; the barrier says so for everybody -- and that everybody is done reading
                // the current box
                vm_wait<0>(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) bq[0][k][n] = bq[1][k][n];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                // in flight: the fragments of the next chunk's taps 0 and 1 (eight loads); every piece of the next box is older
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (VAR == 5 || VAR == 6) { if (grp) __builtin_amdgcn_s_sleep(6); }
        }
    };
    const int ml = m_lo == 0 ? m_hi : MREP;                         // workgroup-uniform
    if (ml == 4) run_chunks(std::integral_constant<int, 4>{});
    else if (ml == 3) run_chunks(std::integral_constant<int, 3>{});
    else if (ml == 2) run_chunks(std::integral_constant<int, 2>{});
    else run_chunks(std::integral_constant<int, 1>{});
    if constexpr (D > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // fragments requested past the last tap: never used, drained before their registers are

    // ---- epilogue: conv3_igemm_sres's, per cout group; group g builds its output image in halo buffer g
    constexpr int TV = TZ * kTY * kTX;
    constexpr int EIT = TV * 8 / 256;
    static_assert(TV * 128 <= BUF && (TV * 8) % 256 == 0, "output image must fit a halo buffer");
    static_assert(kTX * kTY == 128, "the index split of the copy-out");
    unsigned char* img = lds + grp * BUF;
    unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
    const int nco = (a.Cout + 15) / 16;
    float vmax = 0.0f;
    float scv[NREP], shv[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int co = cb * 64 + n * 32 + row;
        scv[n] = co < a.Cout ? a.scale[co] : 0.0f; shv[n] = co < a.Cout ? a.shift[co] : 0.0f;
    }
    asm volatile("" : "+v"(scv[0]), "+v"(scv[1]), "+v"(shv[0]), "+v"(shv[1]));
    int clo[3], chi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { clo[i] = blo[i]; chi[i] = bhi[i]; }
    if (a.store_boxes) {
        const int* sb = a.store_boxes + 6 * tile;
#pragma unroll
        for (int i = 0; i < 3; ++i) { clo[i] = max(clo[i], sb[i] - a.store_grow); chi[i] = min(chi[i], sb[3 + i] + a.store_grow); }
    }
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int co = cb * 64 + n * 32 + row;
        const bool cvalid = co < nco * 16;
        const float sc = scv[n], sh = shv[n];
        const bool odd = row & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;
        unsigned char* lrow = img + (row >> 4) * 64 + ((row & 15) >> 1) * 4;
        __syncthreads();                                              // halo reads / the previous half's copy-out are done
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int oz = oz0 + m;
            const bool zok = oz >= blo[0] && oz < bhi[0];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v[2];
                int vox[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rr = ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * half;
                    const int tx = wx * RX + rr % RX, ty = wy * RY + rr / RX;
                    const int ox = ox0 + tx, oy = oy0 + ty;
                    float x = acc[m][n][r + e] * sc + sh;
                    if (a.relu) x = fmaxf(x, 0.0f);
                    const bool ok = cvalid && zok && ox >= blo[2] && ox < bhi[2] && oy >= blo[1] && oy < bhi[1];
                    v[e] = ok ? x : 0.0f;
                    vox[e] = (m * kTY + ty) * kTX + tx;
                }
                vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1])));
                unsigned w_hi, w_lo;
                split_two_voxels(v[0], v[1], sel, w_hi, w_lo);
                unsigned char* dst = lrow + (odd ? vox[1] : vox[0]) * 128;
                *reinterpret_cast<unsigned*>(dst) = w_hi;
                *reinterpret_cast<unsigned*>(dst + 32) = w_lo;
            }
        }
        __syncthreads();
        {
            constexpr bool kWide = kTX >= 32;
            const int t5 = gtid >> 3, q = gtid & 7;
            const int x_lane = kWide ? t5 : t5 % kTX, y_lane = kWide ? 0 : t5 / kTX;
            const bool cok = cb * 4 + n * 2 + (q >> 2) < nco;
            unsigned char* ob = outb + ((size_t)tile * nco + cb * 4 + n * 2) * plane * 64;                        // wave-uniform
            const unsigned lane_off = (unsigned)(((size_t)(q >> 2) * plane + (size_t)y_lane * a.W + x_lane) * 64 + (q & 3) * 16);
            const int oyl = oy0 + y_lane, oxl = ox0 + x_lane;
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int c = (it & 3) * 32, zc = it >> 2;
                const int xc = kWide ? c % kTX : 0, yc = c / kTX;
                const int oz = oz0 + zc, oy = oyl + yc, ox = oxl + xc;
                if (cok && oz >= clo[0] && oz < chi[0] && oy >= clo[1] && oy < chi[1] && ox >= clo[2] && ox < chi[2]) {
                    const unsigned uni = (unsigned)(((oz * a.H + oy0 + yc) * a.W + ox0 + xc) * 64);                // wave-uniform
                    float4* dstp = reinterpret_cast<float4*>(ob + (size_t)(lane_off + uni));
                    const float4 val = *reinterpret_cast<const float4*>(img + (it * 256 + gtid) * 16);
                    __builtin_nontemporal_store(val.x, &dstp->x); __builtin_nontemporal_store(val.y, &dstp->y);
                    __builtin_nontemporal_store(val.z, &dstp->z); __builtin_nontemporal_store(val.w, &dstp->w);
                }
            }
        }
        if constexpr (RX == 16 && RY == 2 && WY == 4 && WX == 1) {
            if (a.pool_out) {                                     // MaxPool3d(2) fused (see conv3_igemm_sres)
                unsigned char* pb = reinterpret_cast<unsigned char*>(a.pool_out);
                const int Dp = a.D / 2, Hp = a.H / 2, Wp = a.W / 2;
                const bool relu = a.relu != 0;
                auto val = [&](int m, int r) { const float v = acc[m][n][r] * sc + sh; return relu ? fmaxf(v, 0.0f) : v; };
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int r0 = 4 * g + 2 * p;
#pragma unroll
                        for (int m = 0; m < MREP; m += 2) {
                            float v = fmaxf(fmaxf(val(m, r0), val(m, r0 + 1)), fmaxf(val(m, r0 + 8), val(m, r0 + 9)));
                            v = fmaxf(v, fmaxf(fmaxf(val(m + 1, r0), val(m + 1, r0 + 1)), fmaxf(val(m + 1, r0 + 8), val(m + 1, r0 + 9))));
                            const int x = ox0 + 8 * g + 4 * half + 2 * p, y = oy0 + 2 * wy, z = oz0 + m;
                            store_split_pair(pb + srec(tile, nco, (size_t)Dp * Hp * Wp, 0, (((size_t)(z / 2)) * Hp + y / 2) * Wp + x / 2), (size_t)Dp * Hp * Wp * 64,
                                             co, v, cvalid, a.range_flag);
                        }
                    }
            }
        }
    }
    census_note(a.census, a.range_flag, vmax);
}

}  // namespace oai
