// Device kernels of the 3D U-Net segmentation path (gfx950).  Included by unet.hip only.
//
// Layout: every activation is channels-last  [tile][z][y][x][c]  fp32, so that the GEMM K
// dimension (input channels of one tap) is contiguous and an output row of 32 couts is one
// 128-byte store.  Tiles are never materialised from the volume: ec0 gathers its 27 inputs
// straight from the resident volume with reflect-pad index math (Partition.__call__,
// image_transforms.py:395-455), and the head writes only the kept centre block of each tile.
#pragma once
#include <hip/hip_runtime.h>

namespace oai {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM 3x3x3 convolution on v_mfma_f32_32x32x2_f32  (M = voxels, N = cout, K = 27*cin).
//
// Workgroup = 4 waves, output tile  TZ(=MREP) x 8 x 16 voxels  x  64 couts.
//   wave w owns the y-pair {2w, 2w+1}; its MREP row blocks are the z slices of the tile; one MFMA
//   row block (32 rows) = 16 x by 2 y voxels.  Each wave holds MREP x 2 accumulators of 32x32.
// K loop: chunks of KC input channels.  Per chunk the (TZ+2) x 10 x 18 halo box of the input is
//   staged once into LDS (zero-filled outside the tile: Conv3d padding=1 at the TILE border,
//   networks.py:82) and reused by all 27 taps and both cout blocks.
// A fragments: one ds_read_b128 per row block gives the lane 4 consecutive channels = 4 MFMA
//   k-steps (lane half h supplies channels 4h..4h+3, so k is permuted identically in A and B).
// B fragments: weights are pre-packed on the host into per-lane float4 panels in exactly the order
//   the loop consumes them; every wave streams them from L2 with coalesced 1-KiB loads, one step
//   ahead of the MFMAs that use them.  No LDS, no barrier for B.
// Numerics: exact fp32 products, fp32 accumulation (an fmaf chain in k order), like the
//   reference's fp32 CPU path up to summation order.
// ---------------------------------------------------------------------------------------------

// Where a tile's input comes from (ec0's gather; also carried by ConvArgs for the ec0-fused ec1 of the split-resident path).
struct TileSource {
    const float* vol;        // resident volume [D][H][W]  (or null)
    const float* tiles;      // explicit tiles [n][td][th][tw] (B3 seam) when vol == null
    int D, H, W;
    int td, th, tw;          // tile size
    int ez, ey, ex;          // effective (kept) size = tile - 2*overlap
    int oz, oy, ox;          // overlap
    int gy, gx;              // tile grid (y, x); z-major tile order ind = (i*gy + j)*gx + k
    int tile_begin;          // global index of local tile 0
};

__device__ __forceinline__ int reflect_index(int v, int n) {
    // numpy.pad(mode='reflect'): period 2(n-1), no edge repeat
    const int p = 2 * (n - 1);
    int m = v % p;
    if (m < 0) m += p;
    return m < n ? m : p - m;
}

struct ConvArgs {
    const float* src0; const float* src1;   // concat (src0, src1) along channels; src1 may be null
    int C0, C1;
    float* out; int Cout;
    const float4* wpanel;                    // packed weights (see pack_conv3_panel)
    const float4* wpanel16 = nullptr;        // fp16x3: the same weights in the order of conv3_igemm_sres<..., M16>'s tap pairs (pack_conv3_m16_panel), or null
    const float* scale; const float* shift;  // per-cout epilogue: relu(acc*scale + shift)
    int D, H, W;                             // spatial dims of this level (per tile)
    int lo[3], hi[3];                        // output box to compute, [lo,hi) in z,y,x
    int nbz, nby, nbx, ncb;                  // spatial blocks, cout blocks of 64
    int relu;
    const int* boxes;                        // optional [tile][6] (lo z,y,x, hi z,y,x): the part of [lo,hi) THIS tile needs
                                             // (tiles at the volume border need less: their kept centre is partly zeroed)
    float* pool_out;                         // optional: MaxPool3d(2) of the output, [tile][D/2][H/2][W/2][Cout] (main shape only)
    int* range_flag;                         // split-fp16 only: set to 1 if an activation is outside fp16's range (|x| > 65504)
    // split-resident kernel: which part of its output some LATER layer reads -- [tile][6] rows of the consumer's output box, grown by
    // store_grow (its 3x3x3 halo).  Copy-out stores outside it are skipped: a skip tensor is written in full by the encoder but read by
    // the decoder only around the trimmed box (syn0 = ec1's output: dc2 reads 20 x 100 x 100 of 32 x 128 x 128 voxels, 38 %).  The fused
    // max-pool output (pool_out) is not affected.  Null = store everything inside the tile's box.
    const int* store_boxes = nullptr;
    int store_grow = 0;
    // split-resident kernel, instantiation FIRST, shared encoder pass (the "tile" is the whole reflect-padded volume): instead of ONE output
    // tensor the copy-out SCATTERS every voxel into the per-tile tensors out[t] of the tiles t in [sc_tile0, sc_tile0 + sc_ntiles) whose
    // consumer reads it: tile (iz, iy, ix) of the sc_g grid starts at (iz, iy, ix) * sc_e and is sc_t large; sc_boxes[t - sc_tile0] = the
    // consumer's (dc2's) output box in tile coordinates, read with a 1-voxel halo.  Null = the ordinary copy-out.
    const int* sc_boxes = nullptr;
    int sc_ntiles = 0, sc_tile0 = 0;
    int sc_g[3] = {0, 0, 0}, sc_e[3] = {0, 0, 0}, sc_t[3] = {0, 0, 0};
    unsigned* census = nullptr;              // split-resident kernels: 16 words of this layer's max |stored activation| (float bits, atomicMax;
                                             // see census_note).  Feeds the per-layer activation exponents and the low-range flag
    unsigned* first_census = nullptr;        // ... of the fused ec0 (instantiation FIRST)
    int reserved0 = 0;                       // (keeps the kernel-argument layout of earlier rounds: the slot of the removed diagnostic switch word)
    unsigned long long* stamps = nullptr;    // -DOAI_DIAG builds: device array of phase cycle sums (oai_diag_stamps); never set in production
    int nblocks = 0, xcd_group = 0;          // split-resident kernel: true workgroup count and the XCD dealing granularity (see xcd_block_id)
    int* ps_plan = nullptr;                  // conv3_wino_sres<..., PS>: the launch's block plan (wino_plan_kernel): counters, block count, per-tile prefix and sub-boxes
    // split-resident kernel only: dc0 (1x1x1 conv) + sigmoid / threshold + centre crop fused into dc1's epilogue.  When head_w is
    // set the layer's own output is NOT written; every block voxel inside head_boxes[tile] goes to the kept-centre blocks instead.
    const float* head_w = nullptr;           // [ncls][Cout]
    const float* head_b = nullptr;           // [ncls]
    const int* head_boxes = nullptr;         // optional [tile][6]
    float* head_out = nullptr;               // blocks [tile][ncls][ez][ey][ex]
    int head_ncls = 0, head_mode = 0;        // out_mode of oai_segment_tiles: 0 probability, 1 mask, 2 logit
    int head_k[3] = {0, 0, 0}, head_e[3] = {0, 0, 0};   // origin (in tile coordinates) and extent of a kept-centre block
    // split-resident kernel only, instantiation FIRST: ec0 (Conv3d 1 -> 32, k3 p1, + ReLU, gather-fused) is computed INTO the halo box
    // from the raw volume instead of being read back from memory (src0 unused); see conv3_igemm_sres.
    const float* first_w = nullptr;          // ec0 weights [27][32] (Layer::plain)
    const float* first_scale = nullptr; const float* first_shift = nullptr;
    TileSource first_src = {};
};

// Workgroup ids are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2.  In launch order the cout blocks of one
// spatial block and its x neighbours -- which read the same halo -- land on eight different L2s.  Re-deal in groups: XCD k
// takes the logical blocks [(8 i + k) G, (8 i + k + 1) G) for i = 0, 1, ...: neighbours share an L2, and the groups are small
// enough that the XCDs finish together (whole eighths of the grid were 5 % slower: border tiles are trimmed).  The launch grid
// is rounded up to a multiple of 8 G; returns -1 for the padding.
__device__ __forceinline__ int xcd_block_id(int nblocks, int G) {
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int id = ((j / G) * 8 + xcd) * G + j % G;
    return id < nblocks ? id : -1;
}

// intersection of the launch box with the tile's own box; false if the block [o, o+t) misses it entirely
__device__ __forceinline__ bool tile_box(const int* boxes, int tile, const int (&llo)[3], const int (&lhi)[3],
                                         int (&lo)[3], int (&hi)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { lo[i] = llo[i]; hi[i] = lhi[i]; }
    if (boxes) {
        const int* b = boxes + 6 * tile;
#pragma unroll
        for (int i = 0; i < 3; ++i) { lo[i] = max(lo[i], b[i]); hi[i] = min(hi[i], b[3 + i]); }
    }
    return lo[0] < hi[0] && lo[1] < hi[1] && lo[2] < hi[2];
}


// MaxPool3d(2) fused into the conv epilogue (networks.py:112,117,122: pool(relu(conv))).  With the main tile shape a
// lane's 16 accumulator rows are x in {0..3, 8..11} (+4 for the upper half-wave) on 2 y rows, and its MREP accumulators
// are consecutive z slices, so every 2x2x2 pooling window lies in one lane's registers: no cross-lane traffic.
// Requires even block origins and an even number of z slices (host checks); v = epilogue(acc) is applied before the max.
template <int MREP, typename F>
__device__ __forceinline__ void pooled_store(const ConvArgs& a, int tile, int co, int oz0, int oy, int ox0, int half, F&& value) {
    const int Dp = a.D / 2, Hp = a.H / 2, Wp = a.W / 2;
#pragma unroll
    for (int m = 0; m < MREP; m += 2)
#pragma unroll
        for (int g = 0; g < 2; ++g)              // x group: rows 0..3 / 8..11 (+4*half)
#pragma unroll
            for (int p = 0; p < 2; ++p) {        // x pair inside the group
                const int r0 = 4 * g + 2 * p;    // registers r0, r0+1 (y row 0) and r0+8, r0+9 (y row 1)
                float v = fmaxf(fmaxf(value(m, r0), value(m, r0 + 1)), fmaxf(value(m, r0 + 8), value(m, r0 + 9)));
                v = fmaxf(v, fmaxf(fmaxf(value(m + 1, r0), value(m + 1, r0 + 1)), fmaxf(value(m + 1, r0 + 8), value(m + 1, r0 + 9))));
                const int x = ox0 + 8 * g + 4 * half + 2 * p;
                a.pool_out[((((size_t)tile * Dp + (oz0 + m) / 2) * Hp + oy / 2) * Wp + x / 2) * a.Cout + co] = v;
            }
}

// Tile shape: a 32-row MFMA block is RX x RY voxels (x, y); the 4 waves are arranged WY x WX, so a workgroup covers
// MREP x (WY*RY) x (WX*RX) voxels.  The main shape is <16,2,4,1> (4 x 8 x 16); the other shapes exist for the thin
// remainder strips of trimmed output boxes (e.g. 98 = 6*16 + 2), see launch_conv3 in unet.hip.  Every shape
// accumulates each output voxel in the same k order, so the result does not depend on the shape used.
template <int MREP, int KC, int RX, int RY, int WY, int WX>
__global__ void __launch_bounds__(256, 2) conv3_igemm_f32(const ConvArgs a) {
    static_assert(RX * RY == 32 && WY * WX == 4, "bad tile shape");
    constexpr int kConvTY = WY * RY, kConvTX = WX * RX, kConvHY = kConvTY + 2, kConvHX = kConvTX + 2;
    constexpr int NREP = 2;
    constexpr int TZ = MREP, HZ = TZ + 2;
    constexpr int STRIDE = KC + 4;                 // floats per halo voxel (16-byte pad: bank spread)
    constexpr int HVOX = HZ * kConvHY * kConvHX;
    constexpr int Q = KC / 4;                      // float4 per voxel per chunk
    constexpr int KG = KC / 8;                     // k-groups (8 channels = 4 MFMA steps) per chunk
    constexpr int NSLOT = (HVOX * Q + 255) / 256;  // halo float4 slots per thread
    constexpr int NSTEP = 27 * KG;                 // (tap, k-group) steps per chunk
    __shared__ __attribute__((aligned(16))) float lds[HVOX * STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int id = blockIdx.x;
    const int cb = id % a.ncb; id /= a.ncb;
    const int bx = id % a.nbx; id /= a.nbx;
    const int by = id % a.nby; id /= a.nby;
    const int bz = id % a.nbz; id /= a.nbz;
    const int tile = id;
    const int oz0 = a.lo[0] + bz * TZ, oy0 = a.lo[1] + by * kConvTY, ox0 = a.lo[2] + bx * kConvTX;
    int blo[3], bhi[3];                            // what this tile needs of the launch box (wave-uniform)
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    if (oz0 >= bhi[0] || oz0 + TZ <= blo[0] || oy0 >= bhi[1] || oy0 + kConvTY <= blo[1] || ox0 >= bhi[2] || ox0 + kConvTX <= blo[2]) return;
    const int m_lo = max(0, blo[0] - oz0), m_hi = min(MREP, bhi[0] - oz0);   // z slices of this block that are needed

    // Two-level accumulation (round 5): v_mfma_f32_32x32x2_f32 adds TWO products per instruction into its fp32
    // accumulator, so one running sum over K = 27 Cin is a chain of up to 10 368 roundings (dc8) -- measured 2.9-3.5 x as far from the
    // reference network's float64 run as the reference's own fp32 run is, worse than the fp16x3 path (1.7-2.0 x: its MFMA sums 16 / 32
    // products inside the instruction).  Here the MFMAs of one chunk (27 taps x KC channels = 108 instructions at KC 8) run into a FRESH
    // accumulator set `part`, which is folded into the running sum `acc` by one fp32 add per element at the chunk end: a chain of 108 +
    // Cin / 8 roundings instead of 27 Cin / 2 (dc8: 204 instead of 10 368).  The second set costs MREP x NREP x 16 registers, which is
    // why the exact-fp32 path now runs two z slices per block (MREP 2: 64 + 64 accumulator registers, the 128 of the MREP 4 form; the
    // weight fragments per MFMA double, at 1/16 of the fp16 kernels' MFMA rate that is nothing).
    f32x16 acc[MREP][NREP], part[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.0f; part[m][n][r] = 0.0f; }

    const int row = lane & 31, half = lane >> 5;
    const int wy = wave / WX, wx = wave % WX;
    const int lx = wx * RX + row % RX, ly = wy * RY + row / RX;
    // LDS float offset of this lane's voxel for row block 0, tap (0,0,0)
    const float* a_ptr = &lds[(ly * kConvHX + lx) * STRIDE + 4 * half];

    const int nch0 = (a.C0 + KC - 1) / KC, nch1 = (a.C1 + KC - 1) / KC;
    const int nchunks = nch0 + nch1;
    const float4* wp = a.wpanel + (size_t)cb * nchunks * NSTEP * NREP * 64 + lane;
    const size_t plane = (size_t)a.D * a.H * a.W;

    // ---- halo staging, split in two: issue the global loads early, write LDS late (T14) ----------------
    float4 hreg[NSLOT];
    auto halo_load = [&](int ch) {
        const bool first = ch < nch0;
        const float* src = first ? a.src0 : a.src1;
        const int C = first ? a.C0 : a.C1;
        const int c0 = (first ? ch : ch - nch0) * KC;
        const float* sbase = src + (size_t)tile * plane * C + c0;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int slot = tid + i * 256;
            const int hv = slot / Q, q = slot - hv * Q;
            const int hx = hv % kConvHX;
            const int t2 = hv / kConvHX;
            const int hy = t2 % kConvHY, hz = t2 / kConvHY;
            const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < HVOX * Q && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H &&
                (unsigned)gx < (unsigned)a.W && c0 + 4 * q < C)
                v = *reinterpret_cast<const float4*>(sbase + (((size_t)gz * a.H + gy) * a.W + gx) * C + 4 * q);
            hreg[i] = v;
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int slot = tid + i * 256;
            const int hv = slot / Q, q = slot - hv * Q;
            if (slot < HVOX * Q) *reinterpret_cast<float4*>(&lds[hv * STRIDE + 4 * q]) = hreg[i];
        }
    };

    float4 bcur[NREP], bnext[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) bcur[n] = wp[n * 64];
    wp += NREP * 64;
    halo_load(0);

    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                                     // every wave is done reading the previous chunk
        halo_store();
        __syncthreads();
        if (ch + 1 < nchunks) halo_load(ch + 1);   // in flight behind this chunk's 27*KG*MREP*8 MFMAs
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int t = st / KG, kg = st % KG;
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            // this step's A fragments (LDS, ~100 cycles) and the NEXT step's B fragments (L2, ~600 cycles),
            // pinned above this step's MFMAs so that the weight loads have a whole step to land
            float4 acur[MREP];
#pragma unroll
            for (int m = 0; m < MREP; ++m)
                acur[m] = *reinterpret_cast<const float4*>(
                    a_ptr + (((m + dz) * kConvHY + dy) * kConvHX + dx) * STRIDE + 8 * kg);
#pragma unroll
            for (int n = 0; n < NREP; ++n) bnext[n] = wp[n * 64];
            wp += NREP * 64;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    if (m >= m_lo && m < m_hi) {
                        const float av = s == 0 ? acur[m].x : s == 1 ? acur[m].y : s == 2 ? acur[m].z : acur[m].w;
#pragma unroll
                        for (int n = 0; n < NREP; ++n) {
                            const float bv = s == 0 ? bcur[n].x : s == 1 ? bcur[n].y : s == 2 ? bcur[n].z : bcur[n].w;
                            part[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, part[m][n], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NREP; ++n) bcur[n] = bnext[n];
        }
        // fold the chunk's partial sums into the running sums (one rounding per element and chunk)
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[m][n][r] += part[m][n][r]; part[m][n][r] = 0.0f; }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int co = cb * 64 + n * 32 + row;
        if (co >= a.Cout) continue;
        float sc = a.scale[co], sh = a.shift[co];
        asm volatile("" : "+v"(sc), "+v"(sh));      // the wait for these two loads lands HERE, once: otherwise the wait-count pass re-waits (vmcnt(0)) at the top of every
                                                    // conditional store block below, i.e. every store waits for the previous one to reach memory (unet_sres.h, lds_dma16 notes)
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int oz = oz0 + m;
            if (oz < blo[0] || oz >= bhi[0]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ox = ox0 + wx * RX + rr % RX, oy = oy0 + wy * RY + rr / RX;
                if (ox >= blo[2] && ox < bhi[2] && oy >= blo[1] && oy < bhi[1]) {
                    float v = acc[m][n][r] * sc + sh;
                    if (a.relu) v = fmaxf(v, 0.0f);
                    a.out[((size_t)tile * plane + ((size_t)oz * a.H + oy) * a.W + ox) * a.Cout + co] = v;
                }
            }
        }
        if constexpr (RX == 16 && RY == 2 && WY == 4 && WX == 1 && MREP % 2 == 0) {
            if (a.pool_out) {
                const bool relu = a.relu != 0;
                pooled_store<MREP>(a, tile, co, oz0, oy0 + 2 * wy, ox0, half, [&](int m, int r) {
                    const float v = acc[m][n][r] * sc + sh;
                    return relu ? fmaxf(v, 0.0f) : v;
                });
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 variant of the same implicit GEMM on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
//
// Every fp32 operand x is split into NS bf16 terms x = x0 + x1 (+ x2) (+ O(2^-8NS) residual), each term the
// round-to-nearest bf16 of what the previous terms left; a product a*b is the sum of the term products whose
// combined order is < NS:  NS=2 -> a0b0 + a0b1 + a1b0            (3 MFMA passes, error ~2^-17 per product)
//                          NS=3 -> + a0b2 + a2b0 + a1b1           (6 MFMA passes, error ~2^-24: fp32 grade)
// Products are exact in the MFMA and accumulated in fp32, so NS=3 reproduces the fp32 kernel to fp32 rounding
// noise while issuing 6 x 32 cycles per 16-deep k step instead of 8 x 64.
// Activations stay fp32 in HBM: the split happens once per halo voxel when the chunk is staged into LDS (27x
// reuse); weights are split on the host at ingest.  Same tile shapes, same epilogue as conv3_igemm_f32.
// ---------------------------------------------------------------------------------------------

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// x = t0 + t1 (+ t2) + residual, each term the round-to-nearest 16-bit float (bf16: 8-bit mantissa; fp16: 11-bit) of
// what the previous terms left.  With fp16 two terms carry 22 mantissa bits: 3 MFMA passes (a0b0 + a0b1 + a1b0) are
// already fp32 grade (measured 6e-7 relative on the logits), provided |x| < 65504 (activations after ReLU / BatchNorm
// are O(1); weights are pre-scaled per output channel by an exact power of two so that their low term stays normal).
template <int NS, bool FP16>
__device__ __forceinline__ void split_terms(const float4 v, u16x4 (&t)[NS]) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (FP16) {
                const _Float16 b = (_Float16)r[j];        // round to nearest even
                t[k][j] = __builtin_bit_cast(unsigned short, b);
                r[j] -= (float)b;                           // exact in fp32
            } else {
                const __bf16 b = (__bf16)r[j];
                t[k][j] = __builtin_bit_cast(unsigned short, b);
                r[j] -= (float)b;
            }
        }
}

template <bool FP16>
__device__ __forceinline__ f32x16 mfma_16bit(const float4 a, const float4 b, const f32x16 c) {
    if constexpr (FP16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NS, bool FP16, int MREP, int RX, int RY, int WY, int WX>
__global__ void __launch_bounds__(256, 2) conv3_igemm_bf16s(const ConvArgs a) {
    static_assert(RX * RY == 32 && WY * WX == 4 && (NS == 2 || NS == 3), "bad configuration");
    constexpr int KC = 16, NREP = 2, Q = 4;
    constexpr int kTY = WY * RY, kTX = WX * RX, HY = kTY + 2, HX = kTX + 2;
    constexpr int TZ = MREP, HZ = TZ + 2;
    constexpr int REC = NS == 2 ? 80 : 96;            // bytes per halo voxel: NS x 16 bf16 (+16 pad when it fits)
    constexpr int HVOX = HZ * HY * HX;
    constexpr int NSLOT = (HVOX * Q + 255) / 256;
    constexpr int NPASS = NS == 2 ? 3 : 6;
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
    __shared__ __attribute__((aligned(16))) unsigned char lds[HVOX * REC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int id = blockIdx.x;
    const int cb = id % a.ncb; id /= a.ncb;
    const int bx = id % a.nbx; id /= a.nbx;
    const int by = id % a.nby; id /= a.nby;
    const int bz = id % a.nbz; id /= a.nbz;
    const int tile = id;
    const int oz0 = a.lo[0] + bz * TZ, oy0 = a.lo[1] + by * kTY, ox0 = a.lo[2] + bx * kTX;
    int blo[3], bhi[3];
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    if (oz0 >= bhi[0] || oz0 + TZ <= blo[0] || oy0 >= bhi[1] || oy0 + kTY <= blo[1] || ox0 >= bhi[2] || ox0 + kTX <= blo[2]) return;
    const int m_lo = max(0, blo[0] - oz0), m_hi = min(MREP, bhi[0] - oz0);

    f32x16 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int row = lane & 31, half = lane >> 5;
    const int wy = wave / WX, wx = wave % WX;
    const int lx = wx * RX + row % RX, ly = wy * RY + row / RX;
    const unsigned char* a_ptr = &lds[(ly * HX + lx) * REC + 16 * half];

    const int nch0 = (a.C0 + KC - 1) / KC, nch1 = (a.C1 + KC - 1) / KC;
    const int nchunks = nch0 + nch1;
    constexpr int STEP = NS * NREP * 64;               // float4 (16-byte) units of weights per tap
    const float4* wp = a.wpanel + (size_t)cb * nchunks * 27 * STEP + lane;
    const size_t plane = (size_t)a.D * a.H * a.W;

    float4 hreg[NSLOT];
    auto halo_load = [&](int ch) {
        const bool first = ch < nch0;
        const float* src = first ? a.src0 : a.src1;
        const int C = first ? a.C0 : a.C1;
        const int c0 = (first ? ch : ch - nch0) * KC;
        const float* sbase = src + (size_t)tile * plane * C + c0;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int slot = tid + i * 256;
            const int hv = slot / Q, q = slot - hv * Q;
            const int hx = hv % HX;
            const int t2 = hv / HX;
            const int hy = t2 % HY, hz = t2 / HY;
            const int gz = oz0 - 1 + hz, gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < HVOX * Q && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H &&
                (unsigned)gx < (unsigned)a.W && c0 + 4 * q < C)
                v = *reinterpret_cast<const float4*>(sbase + (((size_t)gz * a.H + gy) * a.W + gx) * C + 4 * q);
            hreg[i] = v;
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int slot = tid + i * 256;
            const int hv = slot / Q, q = slot - hv * Q;
            if (slot < HVOX * Q) {
                u16x4 t[NS];
                if constexpr (FP16) {      // an activation fp16 cannot hold would silently become inf: report it instead
                    const float4 v = hreg[i];
                    if (!(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) <= 65504.0f)) atomicOr(a.range_flag, 1);
                }
                split_terms<NS, FP16>(hreg[i], t);
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<u16x4*>(&lds[hv * REC + k * 32 + q * 8]) = t[k];
            }
        }
    };

    float4 bcur[NS][NREP], bnext[NS][NREP];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int n = 0; n < NREP; ++n) bcur[k][n] = wp[(k * NREP + n) * 64];
    wp += STEP;
    halo_load(0);

    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        halo_store();
        __syncthreads();
        if (ch + 1 < nchunks) halo_load(ch + 1);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            float4 acur[NS][MREP];
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int m = 0; m < MREP; ++m)
                    acur[k][m] = *reinterpret_cast<const float4*>(a_ptr + (((m + dz) * HY + dy) * HX + dx) * REC + k * 32);
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int n = 0; n < NREP; ++n) bnext[k][n] = wp[(k * NREP + n) * 64];
            wp += STEP;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    if (m >= m_lo && m < m_hi) {
#pragma unroll
                        for (int n = 0; n < NREP; ++n)
                            acc[m][n] = mfma_16bit<FP16>(acur[PA[p]][m], bcur[PB[p]][n], acc[m][n]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int n = 0; n < NREP; ++n) bcur[k][n] = bnext[k][n];
        }
    }

#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int co = cb * 64 + n * 32 + row;
        if (co >= a.Cout) continue;
        float sc = a.scale[co], sh = a.shift[co];
        asm volatile("" : "+v"(sc), "+v"(sh));      // the wait for these two loads lands HERE, once: otherwise the wait-count pass re-waits (vmcnt(0)) at the top of every
                                                    // conditional store block below, i.e. every store waits for the previous one to reach memory (unet_sres.h, lds_dma16 notes)
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int oz = oz0 + m;
            if (oz < blo[0] || oz >= bhi[0]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ox = ox0 + wx * RX + rr % RX, oy = oy0 + wy * RY + rr / RX;
                if (ox >= blo[2] && ox < bhi[2] && oy >= blo[1] && oy < bhi[1]) {
                    float v = acc[m][n][r] * sc + sh;
                    if (a.relu) v = fmaxf(v, 0.0f);
                    a.out[((size_t)tile * plane + ((size_t)oz * a.H + oy) * a.W + ox) * a.Cout + co] = v;
                }
            }
        }
        if constexpr (RX == 16 && RY == 2 && WY == 4 && WX == 1 && MREP % 2 == 0) {
            if (a.pool_out) {
                const bool relu = a.relu != 0;
                pooled_store<MREP>(a, tile, co, oz0, oy0 + 2 * wy, ox0, half, [&](int m, int r) {
                    const float v = acc[m][n][r] * sc + sh;
                    return relu ? fmaxf(v, 0.0f) : v;
                });
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose3d(k=2, s=2): out[2i+a][2j+b][2k+c][co] = sum_ci x[i][j][k][ci] * W[ci][co][a][b][c]
// (networks.py:56,59,62).  One GEMM with M = input voxels, K = cin, N = 8*cout (column n =
// parity*cout + co), the store scatters each column to its parity's output voxel.
// A and B fragments come straight from global/L2 (no reuse worth staging: 3 % of the FLOPs).
// Workgroup = 4 waves sharing the same 64 input voxels, each wave a different 64-column slab.
// ---------------------------------------------------------------------------------------------

struct UpArgs {
    const float* src; int Cin;
    float* out; int Cout;
    const float4* wpanel;               // [N/64][Cin/8][2][64] float4
    const float* scale; const float* shift;
    int D, H, W;                        // INPUT level dims
    int lo[3], hi[3];                   // INPUT box needed, z,y,x
    int nmb;                            // row blocks of 64 voxels per tile
    int nnb;                            // column groups of 256
    int relu;
    const int* boxes;                   // optional [tile][6]: the part of the INPUT box this tile needs
    int* range_flag;                    // split-fp16: set when an input is outside fp16's range
    unsigned* census = nullptr;         // split-resident kernel: this layer's 16 census words (see ConvArgs::census)
    const unsigned char* zero = nullptr; // split-resident kernel: 64 zero bytes, the LDS-DMA source of rows / columns that do not exist
    unsigned long long* stamps = nullptr;   // -DOAI_DIAG builds: phase cycle sums of the up-conv kernel at stamps[16..31]
    int reserved0 = 0;                  // (keeps the kernel-argument layout of earlier rounds)
    int nblocks = 0, xcd_group = 0;     // split-resident kernel: true workgroup count and the XCD dealing granularity (xcd_block_id): the column
                                        // blocks of one row block read the same A rows and should meet in one L2
    int nbw = 1;                        // split-resident kernel: column blocks a workgroup walks one after the other (1 = one column block per workgroup)
};

// SPLIT = false: exact fp32 MFMA.  SPLIT = true: split-fp16, 3 passes (see conv3_igemm_bf16s): the A rows are split in
// registers as they arrive (each lane owns 8 consecutive channels of its row per 16-deep k step), the weight panel is
// pre-split; the MFMA work drops 5x and the kernel becomes what it should be, bound by its output writes.
template <bool SPLIT>
__global__ void __launch_bounds__(256, 2) upconv2_igemm(const UpArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int id = blockIdx.x;
    const int nb = id % a.nnb; id /= a.nnb;
    const int mb = id % a.nmb; id /= a.nmb;
    const int tile = id;
    const int N = 8 * a.Cout;
    const int ncol0 = nb * 256 + wave * 64;
    if (ncol0 >= N) return;
    const int row = lane & 31, half = lane >> 5;
    const int rz = a.hi[0] - a.lo[0], ry = a.hi[1] - a.lo[1], rx = a.hi[2] - a.lo[2];
    const int nvox = rz * ry * rx;
    int blo[3], bhi[3];
    if (!tile_box(a.boxes, tile, a.lo, a.hi, blo, bhi)) return;
    {   // rows are linear over the launch box: skip the block if its z range misses this tile's box
        const int zf = a.lo[0] + (mb * 64) / (rx * ry), zl = a.lo[0] + min(mb * 64 + 63, nvox - 1) / (rx * ry);
        if (zl < blo[0] || zf >= bhi[0]) return;
    }
    const size_t plane = (size_t)a.D * a.H * a.W;

    // this lane's A rows (two row blocks of 32 voxels)
    const float* ap[2];
    bool av[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int v = mb * 64 + m * 32 + row;
        av[m] = v < nvox;
        const int vv = av[m] ? v : 0;
        const int x = vv % rx, y = (vv / rx) % ry, z = vv / (rx * ry);
        ap[m] = a.src + ((size_t)tile * plane + ((size_t)(a.lo[0] + z) * a.H + (a.lo[1] + y)) * a.W + (a.lo[2] + x)) * a.Cin + 4 * half;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  if constexpr (!SPLIT) {
    const int nkg = (a.Cin + 7) / 8;
    const float4* wp = a.wpanel + (size_t)(ncol0 / 64) * nkg * 2 * 64 + lane;
    auto load_a = [&](int kg, float4 (&af)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
            af[m] = (av[m] && kg * 8 + 4 * half < a.Cin) ? *reinterpret_cast<const float4*>(ap[m] + kg * 8)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 af[2], bf[2], afn[2], bfn[2];
    load_a(0, af);
#pragma unroll
    for (int n = 0; n < 2; ++n) bf[n] = wp[n * 64];
    for (int kg = 0; kg < nkg; ++kg) {
        // next k-group's fragments in flight behind this k-group's 16 MFMAs
        load_a(kg + 1 < nkg ? kg + 1 : kg, afn);
#pragma unroll
        for (int n = 0; n < 2; ++n) bfn[n] = wp[((kg + 1 < nkg ? kg + 1 : kg) * 2 + n) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float x = s == 0 ? af[m].x : s == 1 ? af[m].y : s == 2 ? af[m].z : af[m].w;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float y = s == 0 ? bf[n].x : s == 1 ? bf[n].y : s == 2 ? bf[n].z : bf[n].w;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m][n], 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m) af[m] = afn[m];
#pragma unroll
        for (int n = 0; n < 2; ++n) bf[n] = bfn[n];
    }
  } else {
    // k step = 16 channels; this lane's part of row m: channels 16*ks + 8*half .. +7 (ap already points at 4*half: undo)
    const int nks = (a.Cin + 15) / 16;
    const float4* wp = a.wpanel + (size_t)(ncol0 / 64) * nks * 4 * 64 + lane;       // [ks][term 2][nr 2][lane]
    auto load_a = [&](int ks, float4 (&raw)[2][2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = ks * 16 + 8 * half + 4 * q;
                raw[m][q] = (av[m] && c < a.Cin) ? *reinterpret_cast<const float4*>(ap[m] - 4 * half + c)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    };
    float4 raw[2][2], rawn[2][2], bf[2][2], bfn[2][2];
    load_a(0, raw);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n) bf[k][n] = wp[(k * 2 + n) * 64];
    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    for (int ks = 0; ks < nks; ++ks) {
        const int nx = ks + 1 < nks ? ks + 1 : ks;
        load_a(nx, rawn);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n) bfn[k][n] = wp[((size_t)nx * 4 + k * 2 + n) * 64];
        // split this step's A rows: term k of row block m = 8 fp16 = one 16-byte MFMA operand
        float4 at[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u16x4 lo4[2], hi4[2];
            split_terms<2, true>(raw[m][0], lo4);
            split_terms<2, true>(raw[m][1], hi4);
            if (!(fmaxf(fmaxf(fabsf(raw[m][0].x), fabsf(raw[m][0].y)), fmaxf(fabsf(raw[m][0].z), fabsf(raw[m][0].w))) <= 65504.0f) ||
                !(fmaxf(fmaxf(fabsf(raw[m][1].x), fabsf(raw[m][1].y)), fmaxf(fabsf(raw[m][1].z), fabsf(raw[m][1].w))) <= 65504.0f))
                atomicOr(a.range_flag, 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                u16x8 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo4[k][j]; v[4 + j] = hi4[k][j]; }
                at[k][m] = __builtin_bit_cast(float4, v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma_16bit<true>(at[PA[p]][m], bf[PB[p]][n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) raw[m][q] = rawn[m][q];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n) bf[k][n] = bfn[k][n];
    }
  }
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    // rows of this lane: v = mb*64 + m*32 + 8*g + 4*half + j  (g = r>>2, j = r&3).  Decompose the 8 group bases
    // once (two integer divisions each) and walk j with carries instead of dividing per element.
    int vx[2][4], vy[2][4], vz[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int v = mb * 64 + m * 32 + 8 * g + 4 * half;
            vx[m][g] = v % rx;
            const int t = v / rx;
            vy[m][g] = t % ry;
            vz[m][g] = t / ry;
        }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = ncol0 + n * 32 + row;
        if (col >= N) continue;
        const int par = col / a.Cout, co = col - par * a.Cout;
        const int pa = par >> 2, pb = (par >> 1) & 1, pc = par & 1;
        float sc = a.scale[co], sh = a.shift[co];
        asm volatile("" : "+v"(sc), "+v"(sh));      // the wait for these two loads lands HERE, once: otherwise the wait-count pass re-waits (vmcnt(0)) at the top of every
                                                    // conditional store block below, i.e. every store waits for the previous one to reach memory (unet_sres.h, lds_dma16 notes)
        float* obase = a.out + (size_t)tile * 8 * plane * a.Cout + co;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int x = vx[m][g], y = vy[m][g], z = vz[m][g];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int iz = a.lo[0] + z, iy = a.lo[1] + y, ix = a.lo[2] + x;
                    if (z < rz && iz >= blo[0] && iz < bhi[0] && iy >= blo[1] && iy < bhi[1] && ix >= blo[2] && ix < bhi[2]) {
                        const int oz = 2 * iz + pa, oy = 2 * iy + pb, ox = 2 * ix + pc;
                        float val = acc[m][n][4 * g + j] * sc + sh;
                        if (a.relu) val = fmaxf(val, 0.0f);
                        obase[(((size_t)oz * Ho + oy) * Wo + ox) * a.Cout] = val;
                    }
                    if (++x == rx) { x = 0; if (++y == ry) { y = 0; ++z; } }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// ec0: Conv3d(1 -> cout, k3, p1) + ReLU, fused with the overlap-tile gather.  Memory/VALU bound
// (K = 27): one thread = one voxel x 8 couts, weights through the scalar cache.
// ---------------------------------------------------------------------------------------------

// One thread = two x-adjacent voxels x all 32 (COUT) couts, so every voxel's 128-byte channel row is written
// whole.  The per-axis neighbour indices (reflect-padded volume index, or -1 for a neighbour outside the TILE =
// Conv3d's zero padding) are computed once; weights [27][COUT] sit in LDS and are read as broadcasts.
template <int COUT>
__global__ void __launch_bounds__(256) conv3_first_kernel(const TileSource s, const float* __restrict__ wk /*[27][COUT]*/,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out, int relu) {
    __shared__ __attribute__((aligned(16))) float wl[27 * COUT];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) wl[i] = wk[i];
    __syncthreads();
    const size_t plane = (size_t)s.td * s.th * s.tw;
    const int local_tile = blockIdx.y;
    const int hw = s.tw >> 1;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;      // voxel-pair index
    if (p >= plane / 2) return;
    const int x = 2 * (int)(p % hw), y = (int)((p / hw) % s.th), z = (int)(p / ((size_t)hw * s.th));
    int iz[3], iy[3], ix[4];
    const float* base;
    if (s.vol) {
        const int t = s.tile_begin + local_tile;
        const int tk = t % s.gx, tj = (t / s.gx) % s.gy, ti = t / (s.gx * s.gy);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int zz = z + d - 1, yy = y + d - 1;
            iz[d] = (unsigned)zz < (unsigned)s.td ? reflect_index(ti * s.ez + zz - s.oz, s.D) * s.H * s.W : -1;
            iy[d] = (unsigned)yy < (unsigned)s.th ? reflect_index(tj * s.ey + yy - s.oy, s.H) * s.W : -1;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int xx = x + d - 1;
            ix[d] = (unsigned)xx < (unsigned)s.tw ? reflect_index(tk * s.ex + xx - s.ox, s.W) : -1;
        }
        base = s.vol;
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int zz = z + d - 1, yy = y + d - 1;
            iz[d] = (unsigned)zz < (unsigned)s.td ? zz * s.th * s.tw : -1;
            iy[d] = (unsigned)yy < (unsigned)s.th ? yy * s.tw : -1;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int xx = x + d - 1;
            ix[d] = (unsigned)xx < (unsigned)s.tw ? xx : -1;
        }
        base = s.tiles + (size_t)local_tile * plane;
    }
    float acc[2][COUT];
#pragma unroll
    for (int j = 0; j < COUT; ++j) { acc[0][j] = 0.0f; acc[1][j] = 0.0f; }
#pragma unroll 1
    for (int zy = 0; zy < 9; ++zy) {
        const int dz = zy / 3, dy = zy - 3 * dz;
        const int a = dz == 0 ? iz[0] : dz == 1 ? iz[1] : iz[2];
        const int b = dy == 0 ? iy[0] : dy == 1 ? iy[1] : iy[2];
        float in[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) in[d] = (a | b | ix[d]) >= 0 ? base[(size_t)a + b + ix[d]] : 0.0f;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float* w = &wl[(zy * 3 + dx) * COUT];
#pragma unroll
            for (int j = 0; j < COUT; j += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(w + j);
                acc[0][j] = fmaf(in[dx], w4.x, acc[0][j]);         acc[1][j] = fmaf(in[dx + 1], w4.x, acc[1][j]);
                acc[0][j + 1] = fmaf(in[dx], w4.y, acc[0][j + 1]); acc[1][j + 1] = fmaf(in[dx + 1], w4.y, acc[1][j + 1]);
                acc[0][j + 2] = fmaf(in[dx], w4.z, acc[0][j + 2]); acc[1][j + 2] = fmaf(in[dx + 1], w4.z, acc[1][j + 2]);
                acc[0][j + 3] = fmaf(in[dx], w4.w, acc[0][j + 3]); acc[1][j + 3] = fmaf(in[dx + 1], w4.w, acc[1][j + 3]);
            }
        }
    }
    const size_t v = ((size_t)z * s.th + y) * s.tw + x;
    float* o = out + ((size_t)local_tile * plane + v) * COUT;
#pragma unroll
    for (int vv = 0; vv < 2; ++vv)
#pragma unroll
        for (int j = 0; j < COUT; j += 4) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = acc[vv][j + k] * scale[j + k] + shift[j + k];
                if (relu) r[k] = fmaxf(r[k], 0.0f);
            }
            *reinterpret_cast<float4*>(o + vv * COUT + j) = make_float4(r[0], r[1], r[2], r[3]);
        }
}

// MaxPool3d(2) on channels-last: one thread = one output voxel x 4 channels
__global__ void __launch_bounds__(256) maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       int D, int H, int W, int C, size_t total4) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, C4 = C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t v = i / C4;
        const int x = (int)(v % Wo); v /= Wo;
        const int y = (int)(v % Ho); v /= Ho;
        const int z = (int)(v % Do);
        const size_t tile = v / Do;
        const float* p = in + (((tile * D + 2 * z) * H + 2 * y) * W + 2 * x) * (size_t)C + 4 * c4;
        float4 m = *reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float4 q = *reinterpret_cast<const float4*>(p + ((size_t)((k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1)) * C);
            m.x = fmaxf(m.x, q.x); m.y = fmaxf(m.y, q.y); m.z = fmaxf(m.z, q.z); m.w = fmaxf(m.w, q.w);
        }
        *reinterpret_cast<float4*>(out + i * 4) = m;
    }
}

// dc0 (Conv3d 1x1x1, no ReLU, networks.py:66,148) + torch.sigmoid / > 0.5 (segmenter.py:121-124)
// + the kept-centre crop of Partition.assemble (image_transforms.py:500-503).
// blocks[tile][class][bz][by][bx], one thread per kept voxel.
__global__ void __launch_bounds__(256) head_kernel(const float* __restrict__ in, int Cin, int D, int H, int W,
                                                   int lz, int ly, int lx, int bz, int by, int bx,
                                                   int kz, int ky, int kx, int ez, int ey, int ex,
                                                   const float* __restrict__ w /*[ncls][Cin]*/, const float* __restrict__ bias,
                                                   int ncls, int out_mode, float* __restrict__ blocks, const int* __restrict__ boxes) {
    // launch box [l, l+b) in tile coordinates; the kept-centre block is [k, k+e) and blocks[] is laid out over it
    const size_t nvox = (size_t)bz * by * bx;
    const int tile = blockIdx.y;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const int x = lx + (int)(v % bx), y = ly + (int)((v / bx) % by), z = lz + (int)(v / ((size_t)bx * by));
    if (boxes) {            // voxels of the kept centre that stitch zeroes (volume frame) or trims are left unwritten
        const int* b = boxes + 6 * tile;
        if (z < b[0] || z >= b[3] || y < b[1] || y >= b[4] || x < b[2] || x >= b[5]) return;
    }
    const float* p = in + (((size_t)tile * D + z) * H + y) * (size_t)W * Cin + (size_t)x * Cin;
    float acc[4] = {0, 0, 0, 0};
    for (int c = 0; c < Cin; c += 4) {
        const float4 q = *reinterpret_cast<const float4*>(p + c);
        for (int k = 0; k < ncls; ++k) {
            acc[k] = fmaf(q.x, w[k * Cin + c], acc[k]);
            acc[k] = fmaf(q.y, w[k * Cin + c + 1], acc[k]);
            acc[k] = fmaf(q.z, w[k * Cin + c + 2], acc[k]);
            acc[k] = fmaf(q.w, w[k * Cin + c + 3], acc[k]);
        }
    }
    const size_t evox = (size_t)ez * ey * ex;
    const size_t o = ((size_t)(z - kz) * ey + (y - ky)) * ex + (x - kx);
    for (int k = 0; k < ncls; ++k) {
        float l = acc[k] + bias[k];
        float r = l;
        if (out_mode != 2) {
            const float pr = 1.0f / (1.0f + expf(-l));          // fp32 sigmoid, IEEE divide
            r = out_mode == 1 ? (pr > 0.5f ? 1.0f : 0.0f) : pr;  // literally sigmoid(x) > 0.5 (SURVEY D-4)
        }
        blocks[((size_t)tile * ncls + k) * evox + o] = r;
    }
}

// Where the block of tile t lives in a buffer that an all_gather of per-rank tile ranges left behind (oai_stitch_blocks_ranged): rank r's
// range [bound[r], bound[r + 1]) sits in slots [r * stride, r * stride + its length) -- ragged ranges are padded to `stride` blocks per rank
// by the collective, and the stitch reads through this table instead of a compacting copy.  n == 0: slot = tile (one contiguous list).
struct StitchRanges { int n, stride; int bound[65]; };
__device__ __forceinline__ size_t stitch_slot(const StitchRanges& rg, size_t t) {
    if (rg.n == 0) return t;
    int r = 0;
    while (r + 1 < rg.n && (int)t >= rg.bound[r + 1]) ++r;
    return (size_t)r * rg.stride + (t - (size_t)rg.bound[r]);
}

// Partition.assemble: blocks of all tiles -> maps[class][D][H][W] with trim + zeroed frame
__global__ void __launch_bounds__(256) stitch_kernel(const float* __restrict__ blocks, int ncls, int D, int H, int W,
                                                     int ez, int ey, int ex, int gy, int gx, int cz, int cy, int cx,
                                                     float* __restrict__ maps, const StitchRanges rg) {
    const size_t plane = (size_t)D * H * W;
    const size_t bvox = (size_t)ez * ey * ex;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane * ncls; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i / plane);
        const size_t v = i - (size_t)k * plane;
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
        float r = 0.0f;
        const bool frame = (cz > 0 && (z < cz || z >= D - cz)) || (cy > 0 && (y < cy || y >= H - cy)) ||
                           (cx > 0 && (x < cx || x >= W - cx));
        if (!frame) {
            const int ti = z / ez, tj = y / ey, tk = x / ex;
            const size_t t = stitch_slot(rg, ((size_t)ti * gy + tj) * gx + tk);
            r = blocks[(t * ncls + k) * bvox + ((size_t)(z - ti * ez) * ey + (y - tj * ey)) * ex + (x - tk * ex)];
        }
        maps[i] = r;
    }
}

}  // namespace oai
