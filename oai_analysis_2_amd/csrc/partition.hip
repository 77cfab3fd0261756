// Partition as standalone device operations (oai_analysis/segmentation/image_transforms.py:371-519), for callers that use the
// class directly: materialised tiles (Partition.__call__, :395-455) and the label-vote branch of Partition.assemble (:466-491).
// The prediction path itself never materialises tiles (the first conv gathers with the same index math, unet_kernels.h) and
// assembles through oai_stitch_blocks.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int v, int n) {      // numpy.pad(mode='reflect'): period 2(n-1), no edge repeat
    const int p = 2 * (n - 1);
    int m = v % p;
    if (m < 0) m += p;
    return m < n ? m : p - m;
}

struct PartGeom {
    int D, H, W;
    int tz, ty, tx;      // tile
    int oz, oy, ox;      // overlap
    int ez, ey, ex;      // effective = tile - 2 overlap
    int gz, gy, gx;      // grid
};

// tiles[t - t0][z][y][x] = padded_volume[i ez + z][j ey + y][k ex + x], padded = reflect pad with lo = overlap (:409-434)
__global__ void __launch_bounds__(256) partition_kernel(const float* __restrict__ vol, PartGeom g, int t0, int n, float* __restrict__ out) {
    const long long tvox = (long long)g.tz * g.ty * g.tx;
    const long long total = tvox * n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = t0 + (int)(i / tvox);
        const long long r = i % tvox;
        const int x = (int)(r % g.tx), y = (int)((r / g.tx) % g.ty), z = (int)(r / ((long long)g.tx * g.ty));
        const int tk = t % g.gx, tj = (t / g.gx) % g.gy, ti = t / (g.gx * g.gy);
        const int vz = reflect_idx(ti * g.ez + z - g.oz, g.D), vy = reflect_idx(tj * g.ey + y - g.oy, g.H), vx = reflect_idx(tk * g.ex + x - g.ox, g.W);
        out[i] = vol[((long long)vz * g.H + vy) * g.W + vx];
    }
}

// Vote branch of assemble: every tile votes with ALL its voxels (overlaps included) on the padded canvas; the result at an image
// voxel is the argmax over the label planes (lowest index wins a tie, np.argmax), cropped to [overlap, overlap + size) (:466-484).
template <int MAXL>
__global__ void __launch_bounds__(256) vote_kernel(const int* __restrict__ labels, PartGeom g, int nlab, unsigned char* __restrict__ out) {
    const long long total = (long long)g.D * g.H * g.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((long long)g.W * g.H));
        const int cz = z + g.oz, cy = y + g.oy, cx = x + g.ox;           // canvas coordinate
        int votes[MAXL];
#pragma unroll
        for (int l = 0; l < MAXL; ++l) votes[l] = 0;
        // tiles whose extent [t e, t e + tile) contains the canvas coordinate
        const int i1 = min(cz / g.ez, g.gz - 1), j1 = min(cy / g.ey, g.gy - 1), k1 = min(cx / g.ex, g.gx - 1);
        for (int ti = i1; ti >= 0 && ti * g.ez + g.tz > cz; --ti)
            for (int tj = j1; tj >= 0 && tj * g.ey + g.ty > cy; --tj)
                for (int tk = k1; tk >= 0 && tk * g.ex + g.tx > cx; --tk) {
                    const long long t = ((long long)ti * g.gy + tj) * g.gx + tk;
                    const int lz = cz - ti * g.ez, ly = cy - tj * g.ey, lx = cx - tk * g.ex;
                    const int lab = labels[((t * g.tz + lz) * g.ty + ly) * g.tx + lx];
#pragma unroll
                    for (int l = 0; l < MAXL; ++l) votes[l] += (l == lab);
                }
        int best = 0;
#pragma unroll
        for (int l = 1; l < MAXL; ++l)
            if (l < nlab && votes[l] > votes[best]) best = l;
        out[i] = (unsigned char)best;
    }
}

int make_geom(int D, int H, int W, const int tile[3], const int overlap[3], PartGeom& g) {
    OAI_CHECK_ARG(D > 1 && H > 1 && W > 1, "volume axes must be > 1 (reflect padding)");
    g.D = D; g.H = H; g.W = W;
    g.tz = tile[0]; g.ty = tile[1]; g.tx = tile[2];
    g.oz = overlap[0]; g.oy = overlap[1]; g.ox = overlap[2];
    g.ez = g.tz - 2 * g.oz; g.ey = g.ty - 2 * g.oy; g.ex = g.tx - 2 * g.ox;
    OAI_CHECK_ARG(g.oz >= 0 && g.oy >= 0 && g.ox >= 0 && g.ez > 0 && g.ey > 0 && g.ex > 0, "overlap too large for the tile");
    g.gz = (D + g.ez - 1) / g.ez; g.gy = (H + g.ey - 1) / g.ey; g.gx = (W + g.ex - 1) / g.ex;
    return OAI_OK;
}

unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b));
}

}  // namespace

extern "C" {

int oai_partition_tiles(const float* vol, int D, int H, int W, const int tile[3], const int overlap[3], int tile_begin, int tile_end,
                        float* tiles_out, void* stream) {
    OAI_CHECK_ARG(vol && tile && overlap && tiles_out, "oai_partition_tiles: null pointer");
    PartGeom g;
    if (int rc = make_geom(D, H, W, tile, overlap, g)) return rc;
    const int n_all = g.gz * g.gy * g.gx;
    OAI_CHECK_ARG(0 <= tile_begin && tile_begin <= tile_end && tile_end <= n_all, "oai_partition_tiles: tile range [%d,%d) outside [0,%d)", tile_begin, tile_end, n_all);
    if (tile_end == tile_begin) return OAI_OK;
    const long long total = (long long)(tile_end - tile_begin) * g.tz * g.ty * g.tx;
    partition_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(vol, g, tile_begin, tile_end - tile_begin, tiles_out);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_assemble_vote(const int* tile_labels, int n_labels, int D, int H, int W, const int tile[3], const int overlap[3],
                      unsigned char* out, void* stream) {
    OAI_CHECK_ARG(tile_labels && tile && overlap && out, "oai_assemble_vote: null pointer");
    OAI_CHECK_ARG(n_labels >= 1 && n_labels <= 16, "oai_assemble_vote: 1..16 label classes");
    PartGeom g;
    if (int rc = make_geom(D, H, W, tile, overlap, g)) return rc;
    const unsigned grid = grid_for((long long)D * H * W);
    if (n_labels <= 4) vote_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(tile_labels, g, n_labels, out);
    else vote_kernel<16><<<grid, 256, 0, (hipStream_t)stream>>>(tile_labels, g, n_labels, out);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

}  // extern "C"
