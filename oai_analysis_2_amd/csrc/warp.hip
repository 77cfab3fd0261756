// HBM-bound registration kernels for gfx950: trilinear gather (grid_sample3d), displacement
// compose, 2x average pool, trilinear resize, phi -> ITK displacement, and the fused
// prob-map resample through phi.
//
// Reference semantics restated (see include/oai_hip.h for the call sites):
//   torch.nn.functional.grid_sample(mode='bilinear', padding_mode='border', align_corners=True)
//   behind icon_registration.mermaidlite.compute_warped_image_multiNC (scale_map: g = 2*c - 1,
//   channel reversal to xyz).  One lane per output voxel, the 8 corners fetched as 4 x-adjacent pairs.
#include <type_traits>

#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float identity_coord(int i, double inv_nm1) {
    // mermaidlite.identity_map: float32(index * spacing) with spacing = 1/(n-1) in float64
    return (float)((double)i * inv_nm1);
}

// PyTorch grid_sampler_compute_source_index (align_corners=True) + border clip
__device__ __forceinline__ float unnormalize_border(float c01, int size) {
    float g = c01 * 2.0f - 1.0f;                       // scale_map
    float ix = (g + 1.0f) * (0.5f * (float)(size - 1));  // == ((g + 1) / 2) * (size - 1) bit for bit: the halving is exact
    return fminf((float)(size - 1), fmaxf(ix, 0.0f));
}

// The 8 corners as 4 x-adjacent PAIRS: one 8-byte load fetches (x0, x0+1).  x0 is clamped to w-2 so that the pair
// exists; at the upper border (ix == w-1 exactly) all weight moves to the second element, which is what PyTorch's
// skipped out-of-range corner (weight 0) amounts to.  Same for y and z through clamped row/plane indices.
struct Taps {
    int o[4];      // offsets of the pairs (z0,y0) (z0,y1) (z1,y0) (z1,y1) inside one [d][h][w] plane
    float wx0, wx1, wy0, wy1, wz0, wz1;
};

__device__ __forceinline__ void make_taps(float cz, float cy, float cx, int d, int h, int w, Taps& t) {
    const float iz = unnormalize_border(cz, d), iy = unnormalize_border(cy, h), ix = unnormalize_border(cx, w);
    const float fz0 = floorf(iz), fy0 = floorf(iy), fx0 = floorf(ix);
    const int z0 = (int)fz0, y0 = (int)fy0;
    int x0 = (int)fx0;
    float wx1 = ix - fx0, wx0 = (fx0 + 1.0f) - ix;      // PyTorch's weights: (ix - floor), (floor + 1 - ix)
    if (x0 > w - 2) { x0 = w - 2; wx0 = 0.0f; wx1 = 1.0f; }   // ix == w-1: the in-range corner carries weight 1
    t.wx0 = wx0; t.wx1 = wx1;
    t.wy1 = iy - fy0; t.wy0 = (fy0 + 1.0f) - iy;
    t.wz1 = iz - fz0; t.wz0 = (fz0 + 1.0f) - iz;
    const int z1 = min(z0 + 1, d - 1), y1 = min(y0 + 1, h - 1);   // clamped twin carries weight 0 at the border
    t.o[0] = (z0 * h + y0) * w + x0;
    t.o[1] = (z0 * h + y1) * w + x0;
    t.o[2] = (z1 * h + y0) * w + x0;
    t.o[3] = (z1 * h + y1) * w + x0;
}

struct __attribute__((packed, aligned(4))) pair_f32 { float a, b; };      // 8-byte load at 4-byte alignment (dwordx2)

// (Tried and rejected: fetching only x0 and taking x0+1 from the neighbour lane by ds_bpermute -- the cross-lane moves
// and the masked fix-up load cost more than the second dword: 36 us instead of 23 us for the 160^3 warp.)
__device__ __forceinline__ float gather8(const float* __restrict__ plane, const Taps& t) {
    const pair_f32 p00 = *reinterpret_cast<const pair_f32*>(plane + t.o[0]);
    const pair_f32 p01 = *reinterpret_cast<const pair_f32*>(plane + t.o[1]);
    const pair_f32 p10 = *reinterpret_cast<const pair_f32*>(plane + t.o[2]);
    const pair_f32 p11 = *reinterpret_cast<const pair_f32*>(plane + t.o[3]);
    // PyTorch order and weight products: tnw tne tsw tse bnw bne bsw bse, weight = wx * wy * wz
    float acc = 0.0f;
    acc += p00.a * (t.wx0 * t.wy0 * t.wz0);
    acc += p00.b * (t.wx1 * t.wy0 * t.wz0);
    acc += p01.a * (t.wx0 * t.wy1 * t.wz0);
    acc += p01.b * (t.wx1 * t.wy1 * t.wz0);
    acc += p10.a * (t.wx0 * t.wy0 * t.wz1);
    acc += p10.b * (t.wx1 * t.wy0 * t.wz1);
    acc += p11.a * (t.wx0 * t.wy1 * t.wz1);
    acc += p11.b * (t.wx1 * t.wy1 * t.wz1);
    return acc;
}

// MODE 0: out[c] = sample(src[c], coords)            (image / field warp, C channels)
// MODE 1: out[c] = coords[c] + sample(src[c], coords) (compose, C == 3)
// coords == nullptr -> identity map of the output grid.
// One lane = one output voxel, consecutive lanes = consecutive x: for near-identity maps a wave's 64 pair loads fall in
// two or three 128-byte lines (a lane-per-4-voxels layout touches 8 lines per load and is 3x slower).
// One lane = one output voxel.  A block is a 32(x) x 4(y) x 2(z) brick: lanes of a wave are 32 consecutive x on two
// rows (coalesced 128-byte coordinate reads / result writes, pair loads falling in two or three lines), and the four
// waves' source rows overlap, so the 2x2 (y,z) re-use of every source row is served by the CU's L1 instead of L2.
#ifndef OAI_WARP_U
#define OAI_WARP_U 2
#endif
template <int MODE, typename IDX, int CT>
__global__ void __launch_bounds__(kThreads)
sample_kernel(const float* __restrict__ src, int C_rt, int d, int h, int w,
              const float* __restrict__ coords, int D, int H, int W, float* __restrict__ out, int nbx, int nby, int nbz,
              double inz, double iny, double inx /* 1/(D-1), 1/(H-1), 1/(W-1): the identity map's spacing, from the host */) {
    const IDX plane_out = (IDX)D * H * W;
    const IDX plane_src = (IDX)d * h * w;
    const int C = CT > 0 ? CT : C_rt;              // 1 (image) and 3 (field) are compiled unrolled
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) & 3, tz = threadIdx.x >> 7;
    constexpr int U = OAI_WARP_U;       // z-adjacent bricks per block: their coordinate loads, then their gathers, are in
                                        // flight together, and they share source rows in L1.  32-bit offsets (IDX).
    // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Give every XCD one contiguous run of
    // bricks (a z-slab of the output, hence -- for the near-identity maps of registration -- of the source), so that a
    // source line is fetched into ONE L2 instead of up to eight (17.8 -> 16.6 us on the 160^3 warp, scripts/micro/warp_probe.hip).
    const int nb = nbx * nby * nbz;
    const int per = (nb + 7) >> 3;
    const int logical = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || logical >= nb) return;
    const int bx = logical % nbx, by = (logical / nbx) % nby, bz = logical / (nbx * nby);
    const int x = bx * 32 + tx, y = by * 4 + ty;
    IDX lin[U];
    bool ok[U];
    float cz[U], cy[U], cx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int z = (bz * U + u) * 2 + tz;
        ok[u] = x < W && y < H && z < D;
        lin[u] = ok[u] ? ((IDX)z * H + y) * W + x : 0;
        if (coords) {
            cz[u] = coords[lin[u]]; cy[u] = coords[plane_out + lin[u]]; cx[u] = coords[2 * plane_out + lin[u]];
        } else {
            cz[u] = identity_coord(z, inz); cy[u] = identity_coord(y, iny); cx[u] = identity_coord(x, inx);
        }
    }
    Taps t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) make_taps(cz[u], cy[u], cx[u], d, h, w, t[u]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = gather8(src + c * plane_src, t[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (MODE == 1) r[u] += (c == 0 ? cz[u] : (c == 1 ? cy[u] : cx[u]));
            if (ok[u]) __builtin_nontemporal_store(r[u], out + c * plane_out + lin[u]);     // written once, read by a later kernel
        }
    }
}

// ---- brick form (round 5): the source BOX of an output brick staged in LDS ----------------------------------------------
// sample_kernel is bound by the CU's L1 tag pipeline (TCP_GATE_EN1 97 % of the kernel, 2.35 line accesses per voxel on the spec's +-14-voxel
// field): every lane's four pair loads are separate tag lookups.  Here a block owns a 16 (x) x 8 (y) x 4 (z) output brick, computes its taps,
// reduces the bounding box of their corners over the block (DPP wave reductions + 24 words of LDS), and -- when the box fits kBrickCap floats
// (a per-brick test; on the spec's field a 512-voxel brick's box is ~3.5 x the brick) -- copies the box into LDS by LDS-DMA (the box as one linear
// list, 64 lanes per global_load_lds_dword: a row of the box is one or two cache lines, ~3 x fewer tag lookups per voxel than the gathers) and
// takes the eight corners from LDS (ds_read2_b32 pairs), in gather8's order: bit-identical results.  A brick whose box does not fit (a fold, a
// strong shear) runs gather8 on global memory as before.  Brick 16 x 8 x 4: coordinate / result rows are 64-byte runs, four rows per wave.
constexpr int kBrickX = 16, kBrickY = 8, kBrickZ = 4;
constexpr int kBrickCap = 3584;                                  // floats of LDS per block for the box (14 KB: eight blocks = 32 waves per CU; 24 KB measured the same)

struct Corner { int x0, y0, y1, z0, z1; float wx0, wx1, wy0, wy1, wz0, wz1; };

__device__ __forceinline__ void make_corner(float cz, float cy, float cx, int d, int h, int w, Corner& t) {      // make_taps, keeping the indices
    const float iz = unnormalize_border(cz, d), iy = unnormalize_border(cy, h), ix = unnormalize_border(cx, w);
    const float fz0 = floorf(iz), fy0 = floorf(iy), fx0 = floorf(ix);
    t.z0 = (int)fz0; t.y0 = (int)fy0;
    int x0 = (int)fx0;
    float wx1 = ix - fx0, wx0 = (fx0 + 1.0f) - ix;
    if (x0 > w - 2) { x0 = w - 2; wx0 = 0.0f; wx1 = 1.0f; }
    t.x0 = x0; t.wx0 = wx0; t.wx1 = wx1;
    t.wy1 = iy - fy0; t.wy0 = (fy0 + 1.0f) - iy;
    t.wz1 = iz - fz0; t.wz0 = (fz0 + 1.0f) - iz;
    t.z1 = min(t.z0 + 1, d - 1); t.y1 = min(t.y0 + 1, h - 1);
}

// maximum of an int over the wave (every lane gets it): six DPP steps + a readlane (see wave_max_nonneg in unet_sres.h)
__device__ __forceinline__ int wave_max_i32(int v) {
    auto step = [](int x, auto ctrl, auto rmask) __attribute__((always_inline)) {
        return max(x, __builtin_amdgcn_update_dpp(x, x, decltype(ctrl)::value, decltype(rmask)::value, 0xF, false));
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});       // quad_perm 1,0,3,2
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});       // quad_perm 2,3,0,1
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});      // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});      // row_mirror
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});      // row_bcast15 into rows 1, 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});      // row_bcast31 into rows 2, 3: lane 63 = the wave's maximum
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ float corners8(float a00, float b00, float a01, float b01, float a10, float b10, float a11, float b11, const Corner& t) {
    float acc = 0.0f;                                   // gather8's order and weight products
    acc += a00 * (t.wx0 * t.wy0 * t.wz0);
    acc += b00 * (t.wx1 * t.wy0 * t.wz0);
    acc += a01 * (t.wx0 * t.wy1 * t.wz0);
    acc += b01 * (t.wx1 * t.wy1 * t.wz0);
    acc += a10 * (t.wx0 * t.wy0 * t.wz1);
    acc += b10 * (t.wx1 * t.wy0 * t.wz1);
    acc += a11 * (t.wx0 * t.wy1 * t.wz1);
    acc += b11 * (t.wx1 * t.wy1 * t.wz1);
    return acc;
}

template <int MODE, int CT>
__global__ void __launch_bounds__(kThreads)
sample_brick_kernel(const float* __restrict__ src, int d, int h, int w, const float* __restrict__ coords, int D, int H, int W,
                    float* __restrict__ out, int nbx, int nby, int nbz, double inz, double iny, double inx) {
    __shared__ float box[kBrickCap];
    __shared__ int red[4][6];
    const int plane_out = D * H * W, plane_src = d * h * w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = nbx * nby * nbz;
    const int per = (nb + 7) >> 3;                                   // one contiguous run of bricks per XCD (see sample_kernel)
    const int logical = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || logical >= nb) return;
    const int bx = logical % nbx, by = (logical / nbx) % nby, bz = logical / (nbx * nby);
    const int x = bx * kBrickX + (tid & 15), y = by * kBrickY + ((tid >> 4) & 7);
    constexpr int U = 2;                                             // voxels per thread: z = 2 (tid >> 7) + u
    int lin[U];
    bool ok[U];
    float cz[U], cy[U], cx[U];
    Corner t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int z = bz * kBrickZ + 2 * (tid >> 7) + u;
        ok[u] = x < W && y < H && z < D;
        lin[u] = ok[u] ? (z * H + y) * W + x : 0;
        if (coords) { cz[u] = coords[lin[u]]; cy[u] = coords[plane_out + lin[u]]; cx[u] = coords[2 * plane_out + lin[u]]; }
        else { cz[u] = identity_coord(z, inz); cy[u] = identity_coord(y, iny); cx[u] = identity_coord(x, inx); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) make_corner(cz[u], cy[u], cx[u], d, h, w, t[u]);
    // ---- bounding box of the corners of the brick's REAL voxels: [xlo, xhi] x [ylo, yhi] x [zlo, zhi] (all maxima: lows negated)
    constexpr int kNone = -(1 << 30);
    int m[6] = {kNone, kNone, kNone, kNone, kNone, kNone};
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (ok[u]) {
            m[0] = max(m[0], -t[u].x0); m[1] = max(m[1], t[u].x0 + 1);
            m[2] = max(m[2], -t[u].y0); m[3] = max(m[3], t[u].y1);
            m[4] = max(m[4], -t[u].z0); m[5] = max(m[5], t[u].z1);
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = wave_max_i32(m[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) red[wave][i] = m[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = max(max(red[0][i], red[1][i]), max(red[2][i], red[3][i]));
    const int xlo = -m[0], ylo = -m[2], zlo = -m[4];
    const int nx = m[1] - xlo + 1, ny = m[3] - ylo + 1, nz = m[5] - zlo + 1;
    const bool staged = m[1] != kNone && nx * ny * nz <= kBrickCap;      // block-uniform
    const int C = CT;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const float* plane = src + c * plane_src;
        float r[U];
        if (staged) {
            if (c > 0) __syncthreads();                              // the previous channel's box has been read
            // the box as ONE linear list of nx * ny * nz floats, element e = it * 256 + tid -> (ez, ey, ex): every LDS-DMA instruction moves 64 lanes
            // (a first version issued one instruction per box row, ~25 of 64 lanes active, ~170 instructions per brick where the gather form
            // issues 32: 43.8 us against the gather form's 24.0 -- vector-memory ISSUE, not the tag pipeline, was then the limit)
            {
                const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)box;
                const int vol = nx * ny * nz, nxy = nx * ny;
                const float inv_nxy = 1.0f / (float)nxy, inv_nx = 1.0f / (float)nx;
                for (int e0 = 0; e0 < vol; e0 += kThreads) {
                    const int e = e0 + tid;
                    int ez = (int)((float)e * inv_nxy);                       // e / nxy for e < 2^13: the float quotient is within 1 of it
                    int rem = e - ez * nxy;
                    if (rem < 0) { --ez; rem += nxy; } else if (rem >= nxy) { ++ez; rem -= nxy; }
                    int ey = (int)((float)rem * inv_nx);
                    int ex = rem - ey * nx;
                    if (ex < 0) { --ey; ex += nx; } else if (ex >= nx) { ++ey; ex -= nx; }
                    if (e < vol) {
                        const float* g = plane + ((zlo + ez) * h + (ylo + ey)) * w + xlo + ex;
                        const unsigned la = __builtin_amdgcn_readfirstlane(lbase + (unsigned)(e0 + (tid & ~63)) * 4u);
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(g), "s"(la) : "memory");
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int bxo = t[u].x0 - xlo;
                const int r00 = ((t[u].z0 - zlo) * ny + (t[u].y0 - ylo)) * nx + bxo, r01 = ((t[u].z0 - zlo) * ny + (t[u].y1 - ylo)) * nx + bxo;
                const int r10 = ((t[u].z1 - zlo) * ny + (t[u].y0 - ylo)) * nx + bxo, r11 = ((t[u].z1 - zlo) * ny + (t[u].y1 - ylo)) * nx + bxo;
                r[u] = ok[u] ? corners8(box[r00], box[r00 + 1], box[r01], box[r01 + 1], box[r10], box[r10 + 1], box[r11], box[r11 + 1], t[u]) : 0.0f;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const pair_f32 p00 = *reinterpret_cast<const pair_f32*>(plane + (t[u].z0 * h + t[u].y0) * w + t[u].x0);
                const pair_f32 p01 = *reinterpret_cast<const pair_f32*>(plane + (t[u].z0 * h + t[u].y1) * w + t[u].x0);
                const pair_f32 p10 = *reinterpret_cast<const pair_f32*>(plane + (t[u].z1 * h + t[u].y0) * w + t[u].x0);
                const pair_f32 p11 = *reinterpret_cast<const pair_f32*>(plane + (t[u].z1 * h + t[u].y1) * w + t[u].x0);
                r[u] = corners8(p00.a, p00.b, p01.a, p01.b, p10.a, p10.b, p11.a, p11.b, t[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (MODE == 1) r[u] += (c == 0 ? cz[u] : (c == 1 ? cy[u] : cx[u]));
            if (ok[u]) __builtin_nontemporal_store(r[u], out + c * plane_out + lin[u]);
        }
    }
}

// SURVEY K15 "fuse chains phi(psi(x))" / K18: a whole compose chain per output voxel, in registers:
//     c  = identity [+ start]                         (start: a displacement on the OUTPUT grid -- the isIdentity shortcut)
//     c  = c + sample(field_i, c)     i = 0 .. nf-1   (each field on its own, usually half-resolution, grid)
//     out = WARP ? sample(image, c) : c
// i.e. TwoStepRegistration / DownsampleRegistration's closures of icon_registration composed without materialising c1..c4.
// Every sample is gather8 with the arithmetic and order of sample_kernel, and an intermediate map that the unfused launches
// round to fp32 in memory is the same fp32 value in a register, so the result is bit-identical to the chain of
// oai_compose / oai_grid_sample3d calls it replaces (tests/test_warp_gpu.py), while the traffic drops from ~25 B per voxel and
// link to the compulsory 12 B (start) + 12 x (field voxels / output voxels) per field + 12 (or 4 + 4 for WARP) B written.
constexpr int kMaxChainFields = OAI_WARP_CHAIN_MAX_FIELDS;     // include/oai_hip.h
struct ChainArgs {
    const float* start;                  // [3][D][H][W] or nullptr
    const float* field[kMaxChainFields]; // [3][fd][fh][fw]
    int fd[kMaxChainFields], fh[kMaxChainFields], fw[kMaxChainFields];
    int nf;
    const float* image;                  // WARP: [id][ih][iw]
    int id, ih, iw;
    float* out;                          // WARP: [D][H][W]; else [3][D][H][W]
    int D, H, W, nbx, nby, nbz;
    double inz, iny, inx;
};

template <bool WARP>
__global__ void __launch_bounds__(kThreads)
chain_kernel(const ChainArgs a) {
    constexpr int U = OAI_WARP_U;
    const int D = a.D, H = a.H, W = a.W;
    const int plane_out = D * H * W;
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) & 3, tz = threadIdx.x >> 7;
    const int nb = a.nbx * a.nby * a.nbz;
    const int per = (nb + 7) >> 3;
    const int logical = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || logical >= nb) return;
    const int bx = logical % a.nbx, by = (logical / a.nbx) % a.nby, bz = logical / (a.nbx * a.nby);
    const int x = bx * 32 + tx, y = by * 4 + ty;
    int lin[U];
    bool ok[U];
    float c[3][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int z = (bz * U + u) * 2 + tz;
        ok[u] = x < W && y < H && z < D;
        lin[u] = ok[u] ? (z * H + y) * W + x : 0;
        c[0][u] = identity_coord(z, a.inz); c[1][u] = identity_coord(y, a.iny); c[2][u] = identity_coord(x, a.inx);
        if (a.start) {
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k][u] = c[k][u] + a.start[k * plane_out + lin[u]];      // add_identity_kernel: id + disp
        }
    }
    // one link per field, in application order (a step tree of N FunctionFromVectorFields flattens to N links, icon.hip); the loop
    // is not unrolled: the index into the kernel arguments is wave-uniform (scalar loads)
#pragma unroll 1
    for (int i = 0; i < a.nf; ++i) {
        {
            const int plane = a.fd[i] * a.fh[i] * a.fw[i];
            Taps t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) make_taps(c[0][u], c[1][u], c[2][u], a.fd[i], a.fh[i], a.fw[i], t[u]);
            float g[3][U];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < U; ++u) g[k][u] = gather8(a.field[i] + k * plane, t[u]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < U; ++u) c[k][u] = g[k][u] + c[k][u];                               // sample_kernel MODE 1: r += coord
        }
    }
    if constexpr (WARP) {
        Taps t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) make_taps(c[0][u], c[1][u], c[2][u], a.id, a.ih, a.iw, t[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float r = gather8(a.image, t[u]);
            if (ok[u]) __builtin_nontemporal_store(r, a.out + lin[u]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) __builtin_nontemporal_store(c[k][u], a.out + k * plane_out + lin[u]);
    }
}

// out = identity + disp (same grid): the package's isIdentity shortcut
__global__ void __launch_bounds__(kThreads)
add_identity_kernel(const float* __restrict__ disp, int D, int H, int W, float* __restrict__ out) {
    const long long plane = (long long)D * H * W;
    const double inz = 1.0 / (D - 1), iny = 1.0 / (H - 1), inx = 1.0 / (W - 1);
    for (long long lin = (long long)blockIdx.x * kThreads + threadIdx.x; lin < plane; lin += (long long)gridDim.x * kThreads) {
        const int x = (int)(lin % W);
        const int y = (int)((lin / W) % H);
        const int z = (int)(lin / ((long long)W * H));
        out[lin] = identity_coord(z, inz) + disp[lin];
        out[plane + lin] = identity_coord(y, iny) + disp[plane + lin];
        out[2 * plane + lin] = identity_coord(x, inx) + disp[2 * plane + lin];
    }
}

// avg_pool3d(k=2, s=2, ceil_mode=True): clipped windows divide by the number of valid voxels
__global__ void __launch_bounds__(kThreads)
avgpool2_kernel(const float* __restrict__ in, int C, int D, int H, int W, float* __restrict__ out, int Do, int Ho, int Wo) {
    const long long n = (long long)C * Do * Ho * Wo;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const int xo = (int)(i % Wo);
        const int yo = (int)((i / Wo) % Ho);
        const int zo = (int)((i / ((long long)Wo * Ho)) % Do);
        const int c = (int)(i / ((long long)Wo * Ho * Do));
        const int z0 = 2 * zo, y0 = 2 * yo, x0 = 2 * xo;
        const int z1 = min(z0 + 2, D), y1 = min(y0 + 2, H), x1 = min(x0 + 2, W);
        const float* p = in + (long long)c * D * H * W;
        float s = 0.0f;
        for (int z = z0; z < z1; ++z)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) s += p[((long long)z * H + y) * W + x];
        out[i] = s / (float)((z1 - z0) * (y1 - y0) * (x1 - x0));
    }
}

// PyTorch area_pixel_compute_source_index(align_corners=False) for linear modes
__device__ __forceinline__ void linear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

__global__ void __launch_bounds__(kThreads)
resize_trilinear_kernel(const float* __restrict__ in, int C, int d, int h, int w,
                        float* __restrict__ out, int D, int H, int W) {
    const float sz = (float)d / (float)D, sy = (float)h / (float)H, sx = (float)w / (float)W;
    const long long n = (long long)C * D * H * W;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int z = (int)((i / ((long long)W * H)) % D);
        const int c = (int)(i / ((long long)W * H * D));
        int z0, z1, y0, y1, x0, x1;
        float a0, a1, b0, b1, c0, c1;
        linear_src(z, sz, d, z0, z1, a0, a1);
        linear_src(y, sy, h, y0, y1, b0, b1);
        linear_src(x, sx, w, x0, x1, c0, c1);
        const float* p = in + (long long)c * d * h * w;
        auto at = [&](int zz, int yy, int xx) { return p[((long long)zz * h + yy) * w + xx]; };
        // ATen upsample_trilinear3d accumulation order
        out[i] = a0 * (b0 * (c0 * at(z0, y0, x0) + c1 * at(z0, y0, x1)) + b1 * (c0 * at(z0, y1, x0) + c1 * at(z0, y1, x1))) +
                 a1 * (b0 * (c0 * at(z1, y0, x0) + c1 * at(z1, y0, x1)) + b1 * (c0 * at(z1, y1, x0) + c1 * at(z1, y1, x1)));
    }
}

__global__ void __launch_bounds__(kThreads)
phi_to_disp_kernel(const float* __restrict__ phi, int D, int H, int W, double* __restrict__ disp) {
    const long long plane = (long long)D * H * W;
    const double inz = 1.0 / (D - 1), iny = 1.0 / (H - 1), inx = 1.0 / (W - 1);
    for (long long lin = (long long)blockIdx.x * kThreads + threadIdx.x; lin < plane; lin += (long long)gridDim.x * kThreads) {
        const int x = (int)(lin % W);
        const int y = (int)((lin / W) % H);
        const int z = (int)(lin / ((long long)W * H));
        // fp32 like the reference: (phi - ident) then *= (shape - 1), then .double()
        const float dz = (phi[lin] - identity_coord(z, inz)) * (float)(D - 1);
        const float dy = (phi[plane + lin] - identity_coord(y, iny)) * (float)(H - 1);
        const float dx = (phi[2 * plane + lin] - identity_coord(x, inx)) * (float)(W - 1);
        double* o = disp + 3 * lin;
        o[0] = (double)dx; o[1] = (double)dy; o[2] = (double)dz;
    }
}

struct Affine { double A[9]; double b[3]; };

__device__ __forceinline__ void apply(const Affine& t, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = t.A[0] * x + t.A[1] * y + t.A[2] * z + t.b[0];
    oy = t.A[3] * x + t.A[4] * y + t.A[5] * z + t.b[1];
    oz = t.A[6] * x + t.A[7] * y + t.A[8] * z + t.b[2];
}

__device__ __forceinline__ void clamp_split(double c, int n, int& i0, int& i1, double& f) {
    c = fmin(fmax(c, 0.0), (double)(n - 1));
    const double fl = floor(c);
    i0 = (int)fl;
    i1 = min(i0 + 1, n - 1);
    f = c - fl;
}

// fused K19: B index -> network space -> + trilinear(disp) -> A index -> trilinear(prob), fp64 coordinates
__global__ void __launch_bounds__(kThreads)
resample_kernel(const float* __restrict__ prob, int nzA, int nyA, int nxA,
                const double* __restrict__ disp, int Dn, int Hn, int Wn,
                Affine b2n, Affine n2a, float* __restrict__ out, int nzB, int nyB, int nxB) {
    const long long n = (long long)nzB * nyB * nxB;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const int xb = (int)(i % nxB);
        const int yb = (int)((i / nxB) % nyB);
        const int zb = (int)(i / ((long long)nxB * nyB));
        double nx_, ny_, nz_;
        apply(b2n, (double)xb, (double)yb, (double)zb, nx_, ny_, nz_);
        const bool inside = nx_ >= -0.5 && nx_ < Wn - 0.5 && ny_ >= -0.5 && ny_ < Hn - 0.5 && nz_ >= -0.5 && nz_ < Dn - 0.5;
        if (inside) {
            int x0, x1, y0, y1, z0, z1;
            double fx, fy, fz;
            clamp_split(nx_, Wn, x0, x1, fx);
            clamp_split(ny_, Hn, y0, y1, fy);
            clamp_split(nz_, Dn, z0, z1, fz);
            double acc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                auto at = [&](int zz, int yy, int xx) { return disp[(((long long)zz * Hn + yy) * Wn + xx) * 3 + c]; };
                const double c00 = at(z0, y0, x0) * (1 - fx) + at(z0, y0, x1) * fx;
                const double c01 = at(z0, y1, x0) * (1 - fx) + at(z0, y1, x1) * fx;
                const double c10 = at(z1, y0, x0) * (1 - fx) + at(z1, y0, x1) * fx;
                const double c11 = at(z1, y1, x0) * (1 - fx) + at(z1, y1, x1) * fx;
                acc[c] = (c00 * (1 - fy) + c01 * fy) * (1 - fz) + (c10 * (1 - fy) + c11 * fy) * fz;
            }
            nx_ += acc[0]; ny_ += acc[1]; nz_ += acc[2];
        }
        double ax, ay, az;
        apply(n2a, nx_, ny_, nz_, ax, ay, az);
        float r = 0.0f;
        if (ax >= -0.5 && ax < nxA - 0.5 && ay >= -0.5 && ay < nyA - 0.5 && az >= -0.5 && az < nzA - 0.5) {
            int x0, x1, y0, y1, z0, z1;
            double fx, fy, fz;
            clamp_split(ax, nxA, x0, x1, fx);
            clamp_split(ay, nyA, y0, y1, fy);
            clamp_split(az, nzA, z0, z1, fz);
            auto at = [&](int zz, int yy, int xx) { return (double)prob[((long long)zz * nyA + yy) * nxA + xx]; };
            const double c00 = at(z0, y0, x0) * (1 - fx) + at(z0, y0, x1) * fx;
            const double c01 = at(z0, y1, x0) * (1 - fx) + at(z0, y1, x1) * fx;
            const double c10 = at(z1, y0, x0) * (1 - fx) + at(z1, y0, x1) * fx;
            const double c11 = at(z1, y1, x0) * (1 - fx) + at(z1, y1, x1) * fx;
            r = (float)((c00 * (1 - fy) + c01 * fy) * (1 - fz) + (c10 * (1 - fy) + c11 * fy) * fz);
        }
        out[i] = r;
    }
}

// K18 fused into K19, for all maps of a volume at once: the displacement the ITK transform would hold,
//   disp[zyx][c] = double( (phi[2-c] - identity) * (n - 1) )   (fp32 arithmetic, then widened: exactly phi_to_disp_kernel),
// is rebuilt from phi's fp32 planes at the 8 corners instead of being read back as 24-byte fp64 triples, and the NM probability
// maps share one coordinate computation (fp64, same operations in the same order as resample_kernel: bit-identical results).
// Algorithmic bytes per atlas voxel: NM x (4 read + 4 written) + 12 x (network voxels / atlas voxels) of phi.
// One lane per atlas voxel, consecutive lanes = consecutive x; blocks are dealt to the 8 XCDs in contiguous z-runs so that a
// source line of the near-identity map lands in one L2 (as in sample_kernel).
// Round 6: the two x corners of a row are ONE 8-byte gather (x1 = x0 + 1 except at the clamped end of a row, where the pair starts one
// element earlier and both corners select its upper half): 12 + 8 NM load instructions per voxel instead of 24 + 16 NM.  The kernel is
// bound by the L1's request path (profiles/r05_registration.md: 413 MB in 0.44 ms = 0.12 of 8 TB/s), not by the fp64 chain or HBM; the
// values loaded and every arithmetic operation on them are unchanged: bit-identical outputs.  Needs >= 2 elements per row (host check).
struct PairSel { int base; bool lo_hi, hi_hi; };          // pair = p[base], p[base + 1]; corner x0 = lo_hi ? pair.y : pair.x, corner x1 = hi_hi ? pair.y : pair.x
__device__ __forceinline__ PairSel pair_of(int x0, int x1, int n) {
    const int base = min(x0, n - 2);
    return PairSel{base, x0 != base, x1 != base};
}
__device__ __forceinline__ float2 load_pair(const float* p) {          // 4-byte aligned 8-byte load (global memory: unaligned dwordx2 is legal)
    float2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Index arithmetic (round 6): a block is a run of kRsThreads voxels of ONE output row -- (row, x block) from the block id by wave-uniform
// 32-bit divisions -- instead of a 64-bit linear index divided per lane, and the three fp64 reciprocals 1 / (n - 1) of the identity map
// arrive as arguments (the host computes the same correctly rounded quotients): the kernel was bound by its own instruction count
// (1117 per wave, a third of them integer and fp64 DIVISION sequences), not by memory (profiles/r06_registration.md).
constexpr int kRsThreads = 128;
struct RsInv { double inz, iny, inx; };

template <int NM>
__global__ void __launch_bounds__(kRsThreads)
resample_maps_kernel(const float* __restrict__ prob, int nzA, int nyA, int nxA,
                     const float* __restrict__ phi, int Dn, int Hn, int Wn,
                     Affine b2n, Affine n2a, float* __restrict__ out, int nzB, int nyB, int nxB, unsigned nblocks, unsigned nxblk, RsInv inv) {
    const unsigned per = (nblocks + 7u) >> 3;
    const unsigned logical = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || logical >= nblocks) return;
    const unsigned rowid = logical / nxblk, xblk = logical - rowid * nxblk;       // wave-uniform
    const int zb = (int)(rowid / (unsigned)nyB), yb = (int)(rowid - (unsigned)zb * (unsigned)nyB);
    const int xb = (int)(xblk * kRsThreads + threadIdx.x);
    if (xb >= nxB) return;
    const long long n = (long long)nzB * nyB * nxB;
    const long long i = (long long)rowid * nxB + xb;
    double nx_, ny_, nz_;
    apply(b2n, (double)xb, (double)yb, (double)zb, nx_, ny_, nz_);
    const bool inside = nx_ >= -0.5 && nx_ < Wn - 0.5 && ny_ >= -0.5 && ny_ < Hn - 0.5 && nz_ >= -0.5 && nz_ < Dn - 0.5;
    if (inside) {
        int x0, x1, y0, y1, z0, z1;
        double fx, fy, fz;
        clamp_split(nx_, Wn, x0, x1, fx);
        clamp_split(ny_, Hn, y0, y1, fy);
        clamp_split(nz_, Dn, z0, z1, fz);
        const long long plane = (long long)Dn * Hn * Wn;
        const double inz = inv.inz, iny = inv.iny, inx = inv.inx;
        const int o00 = (z0 * Hn + y0) * Wn, o01 = (z0 * Hn + y1) * Wn, o10 = (z1 * Hn + y0) * Wn, o11 = (z1 * Hn + y1) * Wn;
        const PairSel ps = pair_of(x0, x1, Wn);
        float2 q[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {                          // all twelve pair loads first
            const float* p = phi + (long long)(2 - c) * plane + ps.base;
            q[c][0] = load_pair(p + o00); q[c][1] = load_pair(p + o01); q[c][2] = load_pair(p + o10); q[c][3] = load_pair(p + o11);
        }
        double acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {                          // ITK component c (x, y, z) = phi channel 2 - c (w, h, d)
            const float sc = (float)((c == 0 ? Wn : c == 1 ? Hn : Dn) - 1);
            // identity coordinate of the corner along this component's axis: two values per axis
            const float ia = c == 0 ? identity_coord(x0, inx) : c == 1 ? identity_coord(y0, iny) : identity_coord(z0, inz);
            const float ib = c == 0 ? identity_coord(x1, inx) : c == 1 ? identity_coord(y1, iny) : identity_coord(z1, inz);
            auto at = [&](int r, bool zhi, bool yhi, bool xhi) {
                const float id = c == 0 ? (xhi ? ib : ia) : c == 1 ? (yhi ? ib : ia) : (zhi ? ib : ia);
                const float v = xhi ? (ps.hi_hi ? q[c][r].y : q[c][r].x) : (ps.lo_hi ? q[c][r].y : q[c][r].x);
                return (double)((v - id) * sc);
            };
            const double c00 = at(0, 0, 0, 0) * (1 - fx) + at(0, 0, 0, 1) * fx;
            const double c01 = at(1, 0, 1, 0) * (1 - fx) + at(1, 0, 1, 1) * fx;
            const double c10 = at(2, 1, 0, 0) * (1 - fx) + at(2, 1, 0, 1) * fx;
            const double c11 = at(3, 1, 1, 0) * (1 - fx) + at(3, 1, 1, 1) * fx;
            acc[c] = (c00 * (1 - fy) + c01 * fy) * (1 - fz) + (c10 * (1 - fy) + c11 * fy) * fz;
        }
        nx_ += acc[0]; ny_ += acc[1]; nz_ += acc[2];
    }
    double ax, ay, az;
    apply(n2a, nx_, ny_, nz_, ax, ay, az);
    float r[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) r[m] = 0.0f;
    if (ax >= -0.5 && ax < nxA - 0.5 && ay >= -0.5 && ay < nyA - 0.5 && az >= -0.5 && az < nzA - 0.5) {
        int x0, x1, y0, y1, z0, z1;
        double fx, fy, fz;
        clamp_split(ax, nxA, x0, x1, fx);
        clamp_split(ay, nyA, y0, y1, fy);
        clamp_split(az, nzA, z0, z1, fz);
        const long long planeA = (long long)nzA * nyA * nxA;
        const PairSel ps = pair_of(x0, x1, nxA);
        const long long o00 = ((long long)z0 * nyA + y0) * nxA + ps.base, o01 = ((long long)z0 * nyA + y1) * nxA + ps.base;
        const long long o10 = ((long long)z1 * nyA + y0) * nxA + ps.base, o11 = ((long long)z1 * nyA + y1) * nxA + ps.base;
        float2 v[NM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m) {                           // all loads first: 4 x NM pair requests in flight per lane
            const float* p = prob + m * planeA;
            v[m][0] = load_pair(p + o00); v[m][1] = load_pair(p + o01); v[m][2] = load_pair(p + o10); v[m][3] = load_pair(p + o11);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            auto lo = [&](int k) { return (double)(ps.lo_hi ? v[m][k].y : v[m][k].x); };
            auto hi = [&](int k) { return (double)(ps.hi_hi ? v[m][k].y : v[m][k].x); };
            const double c00 = lo(0) * (1 - fx) + hi(0) * fx;
            const double c01 = lo(1) * (1 - fx) + hi(1) * fx;
            const double c10 = lo(2) * (1 - fx) + hi(2) * fx;
            const double c11 = lo(3) * (1 - fx) + hi(3) * fx;
            r[m] = (float)((c00 * (1 - fy) + c01 * fy) * (1 - fz) + (c10 * (1 - fy) + c11 * fy) * fz);
        }
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) __builtin_nontemporal_store(r[m], out + m * n + i);
}

inline unsigned grid_for(long long work_items) {
    long long blocks = (work_items + kThreads - 1) / kThreads;
    const long long cap = 256LL * 16;            // 256 CUs x 16 blocks: grid-stride beyond that
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

int g_warp_brick = 0;        // option "brick" (oai_warp_set_option): 1 = image warps / composes through sample_brick_kernel when the grids allow it

template <int MODE>
int launch_sample(const float* src, int C, int d, int h, int w, const float* coords, int D, int H, int W,
                  float* out, hipStream_t s) {
    if (g_warp_brick && (C == 1 || C == 3) && (long long)C * D * H * W < (1LL << 31) && (long long)C * d * h * w < (1LL << 31) && w >= 2) {
        const int nbx = (W + kBrickX - 1) / kBrickX, nby = (H + kBrickY - 1) / kBrickY, nbz = (D + kBrickZ - 1) / kBrickZ;
        const long long nb = (long long)nbx * nby * nbz;
        if (nb <= (1LL << 28)) {
            const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
            const double inz = 1.0 / (D - 1), iny = 1.0 / (H - 1), inx = 1.0 / (W - 1);
            if (C == 1) sample_brick_kernel<MODE, 1><<<grid, kThreads, 0, s>>>(src, d, h, w, coords, D, H, W, out, nbx, nby, nbz, inz, iny, inx);
            else sample_brick_kernel<MODE, 3><<<grid, kThreads, 0, s>>>(src, d, h, w, coords, D, H, W, out, nbx, nby, nbz, inz, iny, inx);
            OAI_CHECK_LAUNCH();
            return OAI_OK;
        }
    }
    const int nbx = (W + 31) / 32, nby = (H + 3) / 4, nbz = ((D + 1) / 2 + OAI_WARP_U - 1) / OAI_WARP_U;
    const long long nb = (long long)nbx * nby * nbz;
    if (nb > (1LL << 28)) return oai::set_error(OAI_ERR_ARG, "volume too large for the sample grid");
    const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);                  // 8 XCD runs of ceil(nb/8) bricks
    const long long big = (long long)C * D * H * W > (long long)C * d * h * w ? (long long)C * D * H * W : (long long)C * d * h * w;
    const double inz = 1.0 / (D - 1), iny = 1.0 / (H - 1), inx = 1.0 / (W - 1);
#define OAI_SAMPLE(IDX, CT) sample_kernel<MODE, IDX, CT><<<grid, kThreads, 0, s>>>(src, C, d, h, w, coords, D, H, W, out, nbx, nby, nbz, inz, iny, inx)
    if (big < (1LL << 31)) { if (C == 1) OAI_SAMPLE(int, 1); else if (C == 3) OAI_SAMPLE(int, 3); else OAI_SAMPLE(int, 0); }
    else OAI_SAMPLE(long long, 0);
#undef OAI_SAMPLE
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

}  // namespace

extern "C" {

int oai_warp_set_option(const char* name, int value) {
    OAI_CHECK_ARG(name, "oai_warp_set_option: null name");
    if (!strcmp(name, "brick")) {
        OAI_CHECK_ARG(value == 0 || value == 1, "oai_warp_set_option: brick must be 0 or 1");
        g_warp_brick = value;
        return OAI_OK;
    }
    return oai::set_error(OAI_ERR_ARG, "oai_warp_set_option: unknown option '%s'", name);
}

int oai_grid_sample3d(const float* src, int C, int d, int h, int w, const float* coords, int D, int H, int W,
                      float* out, void* stream) {
    OAI_CHECK_ARG(src && out, "oai_grid_sample3d: null pointer");
    OAI_CHECK_ARG(C > 0 && d > 1 && h > 1 && w > 1 && D > 1 && H > 1 && W > 1, "oai_grid_sample3d: sizes must be > 1");
    return launch_sample<0>(src, C, d, h, w, coords, D, H, W, out, (hipStream_t)stream);
}

int oai_compose(const float* disp, int d, int h, int w, const float* coords, int D, int H, int W, int shortcut,
                float* out, void* stream) {
    OAI_CHECK_ARG(disp && out, "oai_compose: null pointer");
    OAI_CHECK_ARG(d > 1 && h > 1 && w > 1 && D > 1 && H > 1 && W > 1, "oai_compose: sizes must be > 1");
    if (!coords && shortcut && d == D && h == H && w == W) {
        add_identity_kernel<<<grid_for((long long)D * H * W), kThreads, 0, (hipStream_t)stream>>>(disp, D, H, W, out);
        OAI_CHECK_LAUNCH();
        return OAI_OK;
    }
    return launch_sample<1>(disp, 3, d, h, w, coords, D, H, W, out, (hipStream_t)stream);
}

int oai_warp_chain(const float* start, int D, int H, int W, int n_fields, const float* const* fields, const int* field_dims,
                   const float* image, int id, int ih, int iw, float* out, void* stream) {
    OAI_CHECK_ARG(out, "oai_warp_chain: null output");
    OAI_CHECK_ARG(D > 1 && H > 1 && W > 1, "oai_warp_chain: output axes must be > 1");
    OAI_CHECK_ARG(n_fields >= 0 && n_fields <= kMaxChainFields, "oai_warp_chain: 0..%d fields per call", kMaxChainFields);
    OAI_CHECK_ARG(n_fields == 0 || (fields && field_dims), "oai_warp_chain: null field list");
    OAI_CHECK_ARG((long long)3 * D * H * W < (1LL << 31), "oai_warp_chain: output grid too large (32-bit offsets)");
    ChainArgs a{};
    a.start = start; a.nf = n_fields;
    for (int i = 0; i < n_fields; ++i) {
        OAI_CHECK_ARG(fields[i], "oai_warp_chain: null field");
        a.field[i] = fields[i];
        a.fd[i] = field_dims[3 * i]; a.fh[i] = field_dims[3 * i + 1]; a.fw[i] = field_dims[3 * i + 2];
        OAI_CHECK_ARG(a.fd[i] > 1 && a.fh[i] > 1 && a.fw[i] > 1 && (long long)3 * a.fd[i] * a.fh[i] * a.fw[i] < (1LL << 31), "oai_warp_chain: bad field size");
    }
    a.image = image; a.id = id; a.ih = ih; a.iw = iw;
    if (image) OAI_CHECK_ARG(id > 1 && ih > 1 && iw > 1 && (long long)id * ih * iw < (1LL << 31), "oai_warp_chain: bad image size");
    a.out = out; a.D = D; a.H = H; a.W = W;
    a.nbx = (W + 31) / 32; a.nby = (H + 3) / 4; a.nbz = ((D + 1) / 2 + OAI_WARP_U - 1) / OAI_WARP_U;
    a.inz = 1.0 / (D - 1); a.iny = 1.0 / (H - 1); a.inx = 1.0 / (W - 1);
    const long long nb = (long long)a.nbx * a.nby * a.nbz;
    const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
    if (image) chain_kernel<true><<<grid, kThreads, 0, (hipStream_t)stream>>>(a);
    else chain_kernel<false><<<grid, kThreads, 0, (hipStream_t)stream>>>(a);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_avgpool2_3d(const float* in, int C, int D, int H, int W, float* out, void* stream) {
    OAI_CHECK_ARG(in && out && C > 0 && D > 0 && H > 0 && W > 0, "oai_avgpool2_3d: bad arguments");
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    avgpool2_kernel<<<grid_for((long long)C * Do * Ho * Wo), kThreads, 0, (hipStream_t)stream>>>(in, C, D, H, W, out, Do, Ho, Wo);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_resize_trilinear(const float* in, int C, int d, int h, int w, float* out, int D, int H, int W, void* stream) {
    OAI_CHECK_ARG(in && out && C > 0 && d > 0 && h > 0 && w > 0 && D > 0 && H > 0 && W > 0, "oai_resize_trilinear: bad arguments");
    resize_trilinear_kernel<<<grid_for((long long)C * D * H * W), kThreads, 0, (hipStream_t)stream>>>(in, C, d, h, w, out, D, H, W);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_phi_to_itk_displacement(const float* phi, int D, int H, int W, double* disp, void* stream) {
    OAI_CHECK_ARG(phi && disp && D > 1 && H > 1 && W > 1, "oai_phi_to_itk_displacement: bad arguments");
    phi_to_disp_kernel<<<grid_for((long long)D * H * W), kThreads, 0, (hipStream_t)stream>>>(phi, D, H, W, disp);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_resample_through_disp(const float* prob, int nzA, int nyA, int nxA, const double* disp, int Dn, int Hn, int Wn,
                              const oai_affine* b2n, const oai_affine* n2a, float* out, int nzB, int nyB, int nxB,
                              void* stream) {
    OAI_CHECK_ARG(prob && disp && b2n && n2a && out, "oai_resample_through_disp: null pointer");
    OAI_CHECK_ARG(nzA > 0 && nyA > 0 && nxA > 0 && Dn > 0 && Hn > 0 && Wn > 0 && nzB > 0 && nyB > 0 && nxB > 0,
                  "oai_resample_through_disp: bad sizes");
    Affine a, b;
    memcpy(&a, b2n, sizeof(Affine));
    memcpy(&b, n2a, sizeof(Affine));
    resample_kernel<<<grid_for((long long)nzB * nyB * nxB), kThreads, 0, (hipStream_t)stream>>>(
        prob, nzA, nyA, nxA, disp, Dn, Hn, Wn, a, b, out, nzB, nyB, nxB);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_resample_maps_through_phi(const float* probs, int n_maps, int nzA, int nyA, int nxA, const float* phi, int Dn, int Hn, int Wn,
                                  const oai_affine* b2n, const oai_affine* n2a, float* out, int nzB, int nyB, int nxB, void* stream) {
    OAI_CHECK_ARG(probs && phi && b2n && n2a && out, "oai_resample_maps_through_phi: null pointer");
    OAI_CHECK_ARG(n_maps >= 1 && n_maps <= 4, "oai_resample_maps_through_phi: 1..4 maps per call");
    OAI_CHECK_ARG(nzA > 0 && nyA > 0 && nxA > 0 && Dn > 1 && Hn > 1 && Wn > 1 && nzB > 0 && nyB > 0 && nxB > 0,
                  "oai_resample_maps_through_phi: bad sizes");
    OAI_CHECK_ARG((long long)Dn * Hn * Wn < (1LL << 31), "oai_resample_maps_through_phi: network grid too large");
    OAI_CHECK_ARG(nxA >= 2, "oai_resample_maps_through_phi: maps need at least two voxels per row (the x corners are loaded as pairs)");
    Affine a, b;
    memcpy(&a, b2n, sizeof(Affine));
    memcpy(&b, n2a, sizeof(Affine));
    const long long nxblk = (nxB + kRsThreads - 1) / kRsThreads;
    const long long nblocks = (long long)nzB * nyB * nxblk;                       // one block = <= kRsThreads voxels of one output row
    OAI_CHECK_ARG(nblocks < (1LL << 31) - 8 && (long long)nzB * nyB < (1LL << 31), "oai_resample_maps_through_phi: output grid too large");
    const unsigned grid = (unsigned)(((nblocks + 7) / 8) * 8);
    hipStream_t st = (hipStream_t)stream;
    const RsInv inv{1.0 / (Dn - 1), 1.0 / (Hn - 1), 1.0 / (Wn - 1)};             // mermaidlite.identity_map's spacing, in double
#define OAI_RS(NM) resample_maps_kernel<NM><<<grid, kRsThreads, 0, st>>>(probs, nzA, nyA, nxA, phi, Dn, Hn, Wn, a, b, out, nzB, nyB, nxB, (unsigned)nblocks, (unsigned)nxblk, inv)
    if (n_maps == 1) OAI_RS(1); else if (n_maps == 2) OAI_RS(2); else if (n_maps == 3) OAI_RS(3); else OAI_RS(4);
#undef OAI_RS
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

}  // extern "C"
