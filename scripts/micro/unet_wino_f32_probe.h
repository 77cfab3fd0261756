// PROBE COPY of oai_analysis_2_amd/csrc/unet_wino_f32.h (round 6, the version that shipped) with the EXP timing switches of scripts/micro/wino_f32_ablate.hip:
// EXP != 0 computes garbage.  Not part of the library; kept out of the product source on purpose.
// conv3_wino_f32_probe: the exact-fp32 3x3x3 conv (precision OAI_PREC_F32: the reference-precision path, networks.py:43-64) with the x axis in Winograd
// F(2,3) form -- two thirds of the fp32 MFMAs (round 6).
//
// conv3_igemm_f32 runs at 0.90 of the fp32 MFMA peak (v_mfma_f32_32x32x2_f32: 64 cycles per instruction and SIMD): only fewer MFMAs make that path
// faster.  Along x an output pair (X, X + 1) of a k3 correlation needs the inputs d0..d3 = in[X - 1 .. X + 2] and, per (dz, dy, cin), six products;
// in Winograd's minimal form four:
//     t0 = d0 - d2   t1 = d1 + d2   t2 = d2 - d1   t3 = d1 - d3            (input transform, fp32)
//     u0 = g0   u1 = (g0 + g1 + g2) / 2   u2 = (g0 - g1 + g2) / 2   u3 = g2 (weights: host, in double, rounded to fp32 once: pack_wino_f32_panel)
//     m_f = sum over (dz, dy, cin) of t_f u_f                               (four GEMMs with K = 9 Cin instead of one with K = 27 Cin)
//     out[X] = (m0 + m1) + m2       out[X + 1] = (m1 - m2) - m3
// Products are exact fp32 x fp32 -> fp32 MFMA products as in conv3_igemm_f32, and the accumulation is the same TWO-LEVEL scheme: the 36 MFMAs of one
// 8-channel chunk (9 taps x 4 k-steps) run into a fresh partial sum that is folded into the running sum by one add.  Emulated against float64
// (scripts/study/winograd_x_f32_error.py: one layer, K = 576 ... 6 912): rms error 0.93-0.99 x the shipped two-level direct form's -- the shorter chains
// pay for the transform's roundings.
//
// Workgroup = four waves = the four frequencies; block = 2 z x TY y x 2 NP x outputs (TY x NP = 32 pair rows = the M of the 32x32 MFMA tile per slice)
// x 64 couts; wave f holds m_f for the whole block: 2 slices x 2 cout halves x 16 registers = 64 accumulators + 64 for the partial sums, the register
// shape of conv3_igemm_f32.  Per chunk: [halo registers -> raw box (LDS)] [barrier] [transform raw -> T, one (hz, hy, pair, channel quad) unit per
// thread] [barrier] [next chunk's halo loads issued] [9 taps: A fragments from T, weight fragments from L2 one tap ahead].  Epilogue: the four
// frequencies of an output sit in four waves: every wave writes its 64 accumulators to LDS (64 KB, the raw box and T are dead), wave w then finalises
// rows 8 w .. 8 w + 7 of every tile: output transform, scale / shift / ReLU, fp32 channels-last stores (32 consecutive couts = 128 bytes per lane row).
// An output's bits depend on the parity of its x only (pairs start at even tile coordinates: the host aligns the launch box) -- not on blocks or batches.
#pragma once
#include "../../oai_analysis_2_amd/csrc/unet_kernels.h"

namespace oai {

// EXP (128: the input loads fetch full 128-byte lines once per four chunks): timing probes of scripts/micro/wino_f32_ablate.hip (1: no fold, 2: no weight loads in the loop, 4: no input loads / transform, 8: no barrier per
// chunk, 16: no A reads per tap, 32: the weight loads of every tap read the same address, 64: the input loads all read zero16 -- every one of them computes garbage); the library instantiates EXP = 0 only.
template <int TY, int NP, int EXP = 0>
__global__ void __launch_bounds__(256, 2) conv3_wino_f32_probe(const ConvArgs a, const float* __restrict__ zero16) {
    static_assert(TY * NP == 32, "32 pair rows per accumulator tile");
    constexpr int KC = 8, MREP = 2, NREP = 2, TZ = MREP, HZ = TZ + 2, TX = 2 * NP, HY = TY + 2;
    constexpr int STRIDE = KC + 4;                    // floats per record (16-byte pad: bank spread, as conv3_igemm_f32)
    constexpr int Q = KC / 4;
    constexpr int UNITS = HZ * HY * NP * Q;           // one unit = (hz, hy, pair p, channel quad q): four inputs along x -> four frequencies
    constexpr int NU = (UNITS + 255) / 256;
    constexpr int FS = HY * NP * STRIDE;              // one frequency plane of a slice
    constexpr int TF = HZ * 4 * FS;                   // T [hz][f][hy][pair][STRIDE], two of them (this chunk's and the next one's)
    constexpr int XF = 4 * MREP * NREP * 16 * 64;     // epilogue exchange [f][m][n][r][lane]
    constexpr int LDSF = 2 * TF > XF ? 2 * TF : XF;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63, f = tid >> 6;
    int id = blockIdx.x;
    const int cb = id % a.ncb; id /= a.ncb;
    const int bx = id % a.nbx; id /= a.nbx;
    const int by = id % a.nby; id /= a.nby;
    const int bz = id % a.nbz; id /= a.nbz;
    const int tile = id;
    const int oz0 = a.lo[0] + bz * TZ, oy0 = a.lo[1] + by * TY, ox0 = a.lo[2] + bx * TX;      // a.lo[2] is even (host)
    // EXP 256 / 512: stagger the FIRST generation of workgroups (two share a CU and, equally long, would otherwise enter prologue and epilogue together):
    // 256 = a pseudo-random delay of 0..17 x 8 128 cycles, 512 = the second 256 workgroups wait half a block time
    if constexpr ((EXP & 256) != 0) {
        if (blockIdx.x < 512u) { const unsigned n = ((blockIdx.x * 2654435761u) >> 16) % 18u; for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127); }
    }
    if constexpr ((EXP & 512) != 0) {
        if (blockIdx.x >= 256u && blockIdx.x < 512u) { for (unsigned i = 0; i < 9; ++i) __builtin_amdgcn_s_sleep(127); }
    }
    int blo[3], bhi[3];
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    if (oz0 >= bhi[0] || oz0 + TZ <= blo[0] || oy0 >= bhi[1] || oy0 + TY <= blo[1] || ox0 >= bhi[2] || ox0 + TX <= blo[2]) return;

    f32x16 acc[MREP][NREP], part[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.0f; part[m][n][r] = 0.0f; }

    const int row = lane & 31, half = lane >> 5;
    const int a_off = ((f * HY + row / NP) * NP + row % NP) * STRIDE + 4 * half;      // this lane's record for tap (0, 0), slice 0
    constexpr int SLICE = 4 * FS, DYS = NP * STRIDE;

    const int nch0 = (a.C0 + KC - 1) / KC, nch1 = (a.C1 + KC - 1) / KC, nchunks = nch0 + nch1;
    // panel of pack_wino_f32_panel: [cb][f][chunk][tap 9][nr 2][lane] float4
    const float4* wp = a.wpanel + (size_t)(cb * 4 + f) * nchunks * 9 * NREP * 64 + lane;
    const size_t plane = (size_t)a.D * a.H * a.W;

    // this thread's units: the voxel of d0, which of d0..d3 exist (bits 0..3; zero padding outside the image), the unit's place in T
    int vox[NU], tw_off[NU], q4[NU];
    unsigned okm[NU];
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
        const int u = ui * 256 + tid;
        const int q = u % Q;
        int t = u / Q;
        const int p = t % NP; t /= NP;
        const int hy = t % HY, hz = t / HY;
        const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + 2 * p;
        const bool rok = u < UNITS && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H;
        unsigned mk = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mk |= (rok && (unsigned)(gx + k) < (unsigned)a.W) ? 1u << k : 0u;
        okm[ui] = mk;
        vox[ui] = (gz * a.H + gy) * a.W + gx;
        q4[ui] = 4 * q;
        tw_off[ui] = u < UNITS ? ((hz * 4 * HY + hy) * NP + p) * STRIDE + 4 * q : -1;
    }

    // Every thread issues the same NUMBER of loads whatever its units are (a missing unit, a voxel outside the image and the chunk behind the last one read
    // the 16 zero bytes `zero16`): with loads under a branch the compiler can no longer count them and turns the weights' counted waits into full ones.
    float4 hreg[NU][4];
    auto unit_load = [&](int ch, bool real) __attribute__((always_inline)) {
        const bool first = ch < nch0;
        const float* src = first ? a.src0 : a.src1;
        const int C = first ? a.C0 : a.C1;
        const int c0 = (first ? ch : ch - nch0) * KC;
        const float* sbase = src + (size_t)tile * plane * C + c0;
        if constexpr ((EXP & 128) != 0) {
            // EXP 128: what would FULL-LINE input fetches buy?  The halo of FOUR chunks (32 channels = one 128-byte line per voxel) is read once per four
            // chunks, eight lanes per voxel: the same bytes out of HBM, a quarter of the L2 -> L1 line requests (13 loads per thread and four chunks instead of 32)
            constexpr int HXp = TX + 2, HVOXp = HZ * HY * HXp;
            const int phase = ch & 3, n0 = phase == 0 ? 0 : 8, n1 = phase == 0 ? 8 : phase == 1 ? 13 : 8;
            const float* gbase = src + (size_t)tile * plane * C + (c0 & ~31);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int slot = tid + (n0 + i) * 256, hv = slot >> 3, piece = slot & 7;
                const int hx = hv % HXp, t2 = hv / HXp, hy = t2 % HY, hz = t2 / HY;
                const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
                const bool ok = real && n0 + i < n1 && hv < HVOXp && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && (c0 & ~31) + 4 * piece < C;
                const float* p = ok ? gbase + ((ptrdiff_t)(gz * a.H + gy) * a.W + gx) * C + 4 * piece : zero16;
                hreg[i >> 2][i & 3] = *reinterpret_cast<const float4*>(p);
            }
            return;
        }
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
            const float* rp = sbase + (ptrdiff_t)vox[ui] * C + q4[ui];
            const bool cok = real && c0 + q4[ui] < C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* p = (cok && ((okm[ui] >> k) & 1)) ? rp + (ptrdiff_t)k * C : zero16;
                hreg[ui][k] = *reinterpret_cast<const float4*>(p);
            }
        }
    };
    // frequency fq of unit ui -> Tb:   t0 = d0 - d2   t1 = d1 + d2   t2 = d2 - d1   t3 = d1 - d3
    auto unit_piece = [&](float* Tb, int ui, int fq) __attribute__((always_inline)) {
        if (tw_off[ui] >= 0) {
            const float4 x = fq == 0 ? hreg[ui][0] : fq == 2 ? hreg[ui][2] : hreg[ui][1];
            const float4 y = fq == 1 ? hreg[ui][2] : fq == 2 ? hreg[ui][1] : fq == 3 ? hreg[ui][3] : hreg[ui][2];
            const float4 v = fq == 1 ? make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w) : make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
            *reinterpret_cast<float4*>(Tb + tw_off[ui] + fq * FS) = v;
        }
    };

    // weight fragments: a ring of three taps, requested two taps ahead (L2 latency is longer than one tap's 16 MFMAs); 9 taps per chunk = 3 turns of the ring
    float4 bq[3][NREP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NREP; ++n) bq[i][n] = wp[(i * NREP + n) * 64];
    wp += 2 * NREP * 64;
    if constexpr ((EXP & 4096) == 0) {        // EXP 4096: no first-chunk loads / transform in the prologue
    unit_load(0, true);
#pragma unroll
    for (int ui = 0; ui < NU; ++ui)
#pragma unroll
        for (int fq = 0; fq < 4; ++fq) unit_piece(lds, ui, fq);
    }
    __syncthreads();

    // One barrier per chunk: the next chunk's inputs are requested at the head of this chunk's taps (straight from global memory: every unit reads its own
    // four voxels -- no raw staging box, the 2 x overlap of neighbouring pairs comes out of L1 / L2), transformed BETWEEN the MFMAs of tap kTapX into the
    // other T buffer (two MFMAs, then the four VALU ops and the one LDS write of one frequency of one unit: the transform rides in the MFMAs' shadow).
    constexpr int kTapX = 5;
    static_assert(NU * 4 <= 8, "the pieces of a thread's units fit between the MFMA pairs of one tap");
    if constexpr ((EXP & 1024) != 0) __builtin_amdgcn_s_setprio(3);      // EXP 1024: the chunk loop at high wave priority, prologue / epilogue at 0
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* const Tc = lds + (ch & 1) * TF + a_off;
        float* const Tn = lds + ((ch + 1) & 1) * TF;
        const bool more = ch + 1 < nchunks;
        float4 acur[MREP], anext[MREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) acur[m] = *reinterpret_cast<const float4*>(Tc + m * SLICE);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // the next tap's fragments are requested before this tap's 16 MFMAs: A from T (LDS), the weights from L2 two taps ahead (the panel has slack behind its end)
            if (t + 1 < 9) {
                const int dz = (t + 1) / 3, dy = (t + 1) % 3;
#pragma unroll
                for (int m = 0; m < MREP; ++m) anext[m] = (EXP & 16) ? acur[m] : *reinterpret_cast<const float4*>(Tc + (m + dz) * SLICE + dy * DYS);
            }
#pragma unroll
            for (int n = 0; n < NREP; ++n) bq[(t + 2) % 3][n] = (EXP & 2) ? bq[t % 3][n] : wp[n * 64];
            if (!(EXP & 32)) wp += NREP * 64;
            // the next chunk's inputs: requested BEHIND tap 2's weights -- vmcnt counts in order, the first wait that covers them is tap 3's
            if (t == 0 && !(EXP & 4)) unit_load(more ? ch + 1 : ch, more && !(EXP & 64));
            __builtin_amdgcn_sched_barrier(0);
            // (both z slices always: a block that straddles its box in z computes a slice it does not store -- a guard per MFMA costs more than it saves)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    const float av = s == 0 ? acur[m].x : s == 1 ? acur[m].y : s == 2 ? acur[m].z : acur[m].w;
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        const float4 b4 = bq[t % 3][n];
                        const float bv = s == 0 ? b4.x : s == 1 ? b4.y : s == 2 ? b4.z : b4.w;
                        part[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, part[m][n], 0, 0, 0);
                    }
                    if (t == kTapX) {
                        const int piece = s * MREP + m;                       // 0..7
                        __builtin_amdgcn_sched_barrier(0);
                        if (more && piece < NU * 4 && !(EXP & 4)) unit_piece(Tn, piece / 4, piece % 4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) {
#pragma unroll
                for (int m = 0; m < MREP; ++m) acur[m] = anext[m];
            }
        }
        // fold the chunk's partial sums into the running sums (one rounding per element and chunk)
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (EXP & 1) { if (ch + 1 == nchunks) acc[m][n][r] = part[m][n][r]; }
                    else { acc[m][n][r] += part[m][n][r]; part[m][n][r] = 0.0f; }
                }
        if constexpr (!(EXP & 8)) __syncthreads();           // T[next] is written, T[this] is read by every wave
    }

    if constexpr ((EXP & 1024) != 0) __builtin_amdgcn_s_setprio(0);
    if constexpr ((EXP & 2048) != 0) {        // EXP 2048: no epilogue (one impossible store keeps the accumulators alive)
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
        if (sum == 123.456f) a.out[tid] = sum;
        return;
    }
    // ---- epilogue: exchange the frequencies through LDS, output transform, scale / shift / ReLU, stores
    __syncthreads();
    float* const XB = lds;
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) XB[(((f * MREP + m) * NREP + n) * 16 + r) * 64 + lane] = acc[m][n][r];
    __syncthreads();
    // wave w = f finalises registers r = 4 w + j of every tile: C/D row (r & 3) + 8 (r >> 2) + 4 half = j + 8 w + 4 half (a pair row), column = lane & 31.
    // For TY x NP = 8 x 4 that is y = 2 w + half, pair j: the lane holds both z slices and both x of a 2 x 2 x 2 pooling window, its partner lane ^ 32 the other
    // y row -- the fused MaxPool3d(2) (ec1 / ec3 / ec5, networks.py:113,117,122; the host passes pool_out only where the box is whole blocks) is one cross-lane max.
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int co = cb * 64 + n * 32 + row;
        const bool cok = co < a.Cout;
        float sc = cok ? a.scale[co] : 0.0f, sh = cok ? a.shift[co] : 0.0f;
        asm volatile("" : "+v"(sc), "+v"(sh));      // (the wait for the two loads lands here, once: see conv3_igemm_f32)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * f + j;
            const int pr = j + 8 * f + 4 * half;
            const int oy = oy0 + pr / NP, ox = ox0 + 2 * (pr % NP);
            float pv = -3.0e38f;
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int oz = oz0 + m;
                const float* x = XB + ((m * NREP + n) * 16 + r) * 64 + lane;
                constexpr int FSX = MREP * NREP * 16 * 64;
                const float m0 = x[0], m1 = x[FSX], m2 = x[2 * FSX], m3 = x[3 * FSX];
                float o0 = ((m0 + m1) + m2) * sc + sh, o1 = ((m1 - m2) - m3) * sc + sh;
                if (a.relu) { o0 = fmaxf(o0, 0.0f); o1 = fmaxf(o1, 0.0f); }
                pv = fmaxf(pv, fmaxf(o0, o1));
                if (cok && oz >= blo[0] && oz < bhi[0] && oy >= blo[1] && oy < bhi[1]) {
                    float* dst = a.out + ((size_t)tile * plane + ((size_t)oz * a.H + oy) * a.W + ox) * a.Cout + co;
                    if (ox >= blo[2] && ox < bhi[2]) dst[0] = o0;
                    if (ox + 1 >= blo[2] && ox + 1 < bhi[2]) dst[a.Cout] = o1;
                }
            }
            if constexpr (TY == 8 && NP == 4) {
                if (a.pool_out) {                                          // (wave-uniform)
                    pv = fmaxf(pv, __shfl_xor(pv, 32, 64));              // the window's other y row
                    if (cok && half == 0) {
                        const int Dp = a.D / 2, Hp = a.H / 2, Wp = a.W / 2;
                        a.pool_out[((((size_t)tile * Dp + oz0 / 2) * Hp + oy / 2) * Wp + ox / 2) * a.Cout + co] = pv;
                    }
                }
            }
        }
    }
}

}  // namespace oai
