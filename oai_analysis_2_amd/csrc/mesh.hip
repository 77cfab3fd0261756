// Iso-surface + smoothing + closest-point distance for gfx950: the step after the hot path.
//
// Replaces, on the device, the three library calls of oai_analysis/mesh_processing.py:
//   skimage.measure.marching_cubes(img, level=0.5, spacing, step_size=1)      (:325-336)  -> oai_mc_count / oai_mc_emit
//   vtkSmoothPolyDataFilter(num_iterations=150)                                 (:298-307)  -> oai_mesh_smooth
//   vtkDistancePolyDataFilter(SignedDistanceOff, ComputeSecondDistance)         (:310-322)  -> oai_mesh_point_distance
// None of skimage / vtk is installed, so the published algorithms are restated (see oracle/mesh.py for the conventions):
// marching cubes with a case table generated from first principles (face-consistent, watertight), Jacobi Laplacian
// smoothing over the edge graph, exact closest point on a triangle (Ericson 5.1.5).
//
// All three are HBM / VALU streaming kernels; nothing here is GEMM-shaped.
#include "common.h"

#include <mutex>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// case table, generated on the host (same construction as oracle/mesh.py: tests compare the two tables entry by entry)
// ---------------------------------------------------------------------------------------------------------------------
const int kFaces[6][4] = {{0, 4, 6, 2}, {1, 3, 7, 5}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 2, 3, 1}, {4, 5, 7, 6}};   // ccw seen from outside

int edge_of(int c0, int c1) {
    const int d = c0 ^ c1, lo = c0 < c1 ? c0 : c1;
    const int x = lo & 1, y = (lo >> 1) & 1, z = lo >> 2;
    if (d == 1) return 0 * 4 + y + 2 * z;
    if (d == 2) return 1 * 4 + x + 2 * z;
    return 2 * 4 + x + 2 * y;
}

// do two cube edges lie on a common cube face?  edge = axis * 4 + j, j's two bits = the coordinates along the other two axes
bool share_face(int e1, int e2) {
    auto faces = [](int e, int (&f)[2]) {
        const int axis = e >> 2, j = e & 3;
        const int o0 = axis == 0 ? 1 : 0, o1 = axis == 2 ? 1 : 2;
        f[0] = o0 * 2 + (j & 1);
        f[1] = o1 * 2 + (j >> 1);
    };
    int a[2], b[2];
    faces(e1, a);
    faces(e2, b);
    return a[0] == b[0] || a[0] == b[1] || a[1] == b[0] || a[1] == b[1];
}

struct Tables {
    signed char tri[256][16];
    unsigned char ntri[256];
};

const Tables& host_tables() {
    static Tables t;
    static std::once_flag once;
    std::call_once(once, [] {
        for (int cs = 0; cs < 256; ++cs) {
            int nxt[12];
            for (int e = 0; e < 12; ++e) nxt[e] = -1;
            for (const auto& quad : kFaces) {
                int enters[2], exits[2], ne = 0, nx = 0;
                for (int k = 0; k < 4; ++k) {
                    const int a = (cs >> quad[k]) & 1, b = (cs >> quad[(k + 1) & 3]) & 1;
                    if (a && !b) exits[nx++] = k;
                    else if (!a && b) enters[ne++] = k;
                }
                for (int i = 0; i < ne; ++i) {
                    // the inside arc entered here ends at the first exit after it (ccw): ambiguous faces isolate inside corners
                    int best = -1, bd = 5;
                    for (int j = 0; j < nx; ++j) {
                        const int dd = (exits[j] - enters[i] + 4) & 3;
                        if (dd < bd) { bd = dd; best = exits[j]; }
                    }
                    nxt[edge_of(quad[enters[i]], quad[(enters[i] + 1) & 3])] = edge_of(quad[best], quad[(best + 1) & 3]);
                }
            }
            bool seen[12] = {};
            int n = 0;
            for (int e = 0; e < 16; ++e) t.tri[cs][e] = -1;
            for (int start = 0; start < 12; ++start) {
                if (nxt[start] < 0 || seen[start]) continue;
                int loop[12], len = 0;
                for (int e = start; !seen[e]; e = nxt[e]) { seen[e] = true; loop[len++] = e; }
                // fan apex: the first rotation with no diagonal lying in a cube face (the neighbouring cube could draw the same
                // diagonal on the shared face: four triangles on one edge)
                int best = 0, best_bad = 1 << 30;
                for (int r = 0; r < len; ++r) {
                    int bad = 0;
                    for (int i = 2; i < len - 1; ++i) bad += share_face(loop[r], loop[(r + i) % len]) ? 1 : 0;
                    if (bad < best_bad) { best_bad = bad; best = r; }
                }
                for (int i = 1; i + 1 < len; ++i) {
                    t.tri[cs][n++] = (signed char)loop[best];
                    t.tri[cs][n++] = (signed char)loop[(best + i) % len];
                    t.tri[cs][n++] = (signed char)loop[(best + i + 1) % len];
                }
            }
            t.ntri[cs] = (unsigned char)(n / 3);
        }
    });
    return t;
}

__constant__ signed char c_tri[256][16];
__constant__ unsigned char c_ntri[256];

int upload_tables() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    // per device would be more general; the library targets one GPU per process
    std::call_once(once, [] {
        const Tables& t = host_tables();
        err = hipMemcpyToSymbol(HIP_SYMBOL(c_tri), t.tri, sizeof(t.tri));
        if (err == hipSuccess) err = hipMemcpyToSymbol(HIP_SYMBOL(c_ntri), t.ntri, sizeof(t.ntri));
    });
    if (err != hipSuccess) return oai::set_error(OAI_ERR_HIP, "marching-cubes table upload failed: %s", hipGetErrorString(err));
    return OAI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// exclusive scan of int32 (counts per voxel), three levels of 1024-element blocks
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kScanBlock = 1024;          // elements per workgroup (256 threads x 4)

__global__ void __launch_bounds__(256) scan_block_kernel(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ block_sums, long long n) {
    __shared__ int wsum[4];
    const long long base = (long long)blockIdx.x * kScanBlock + threadIdx.x * 4;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = base + i < n ? in[base + i] : 0;
    const int tsum = v[0] + v[1] + v[2] + v[3];
    int incl = tsum;                                                   // inclusive scan over the wave (DPP-free: shuffles)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + incl - tsum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (block_sums && threadIdx.x == 255) block_sums[blockIdx.x] = woff + incl;
}

__global__ void __launch_bounds__(256) scan_add_kernel(int* __restrict__ out, const int* __restrict__ block_offsets, long long n) {
    const long long base = (long long)blockIdx.x * kScanBlock + threadIdx.x * 4;
    const int off = block_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < n) out[base + i] += off;
}

size_t scan_scratch_ints(long long n) {          // block sums of every level
    size_t total = 0;
    while (n > kScanBlock) {
        n = (n + kScanBlock - 1) / kScanBlock;
        total += (size_t)((n + 63) / 64 * 64);
    }
    return total + 64;
}

// out[i] = sum in[0..i); returns through *total_dev (device int, may alias scratch tail) nothing -- the caller reads
// in[n-1] + out[n-1].  scratch: scan_scratch_ints(n) ints.
int exclusive_scan(const int* in, int* out, long long n, int* scratch, hipStream_t st) {
    const long long nb = (n + kScanBlock - 1) / kScanBlock;
    if (nb <= 1) {
        scan_block_kernel<<<1, 256, 0, st>>>(in, out, nullptr, n);
        OAI_CHECK_LAUNCH();
        return OAI_OK;
    }
    int* sums = scratch;
    scan_block_kernel<<<(unsigned)nb, 256, 0, st>>>(in, out, sums, n);
    OAI_CHECK_LAUNCH();
    if (int rc = exclusive_scan(sums, sums, nb, scratch + (nb + 63) / 64 * 64, st)) return rc;      // in place
    scan_add_kernel<<<(unsigned)nb, 256, 0, st>>>(out, sums, n);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// marching cubes
// ---------------------------------------------------------------------------------------------------------------------
// per voxel: bits 0..2 = the grid edge from this voxel along +x / +y / +z changes sign; vcount = popcount; tcount = triangles of
// the cell whose lowest corner is this voxel (0 on the last plane / row / column)
__global__ void __launch_bounds__(256) mc_classify_kernel(const float* __restrict__ vol, int D, int H, int W, float iso,
                                                          unsigned char* __restrict__ flags, int* __restrict__ vcount, int* __restrict__ tcount) {
    const long long n = (long long)D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((long long)W * H));
    const long long sy = W, sz = (long long)W * H;
    const bool in0 = vol[i] > iso;
    const bool hx = x + 1 < W, hy = y + 1 < H, hz = z + 1 < D;
    const bool i1 = hx && vol[i + 1] > iso, i2 = hy && vol[i + sy] > iso, i4 = hz && vol[i + sz] > iso;
    unsigned f = 0;
    if (hx && i1 != in0) f |= 1;
    if (hy && i2 != in0) f |= 2;
    if (hz && i4 != in0) f |= 4;
    flags[i] = (unsigned char)f;
    vcount[i] = __popc(f);
    int nt = 0;
    if (hx && hy && hz) {
        unsigned cs = (in0 ? 1u : 0u) | (i1 ? 2u : 0u) | (i2 ? 4u : 0u) | (i4 ? 16u : 0u);
        cs |= vol[i + sy + 1] > iso ? 8u : 0u;
        cs |= vol[i + sz + 1] > iso ? 32u : 0u;
        cs |= vol[i + sz + sy] > iso ? 64u : 0u;
        cs |= vol[i + sz + sy + 1] > iso ? 128u : 0u;
        nt = c_ntri[cs];
    }
    tcount[i] = nt;
}

__global__ void __launch_bounds__(256) mc_emit_kernel(const float* __restrict__ vol, int D, int H, int W, float iso, float sx, float sy_, float sz_,
                                                      const unsigned char* __restrict__ flags, const int* __restrict__ voff, const int* __restrict__ toff,
                                                      float* __restrict__ verts, int* __restrict__ faces) {
    const long long n = (long long)D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((long long)W * H));
    const long long sy = W, sz = (long long)W * H;
    const unsigned f = flags[i];
    const float v0 = vol[i];
    // ---- vertices owned by this voxel, in axis order
    if (f) {
        int o = voff[i];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
            if (f & (1u << ax)) {
                const float v1 = vol[i + (ax == 0 ? 1 : ax == 1 ? sy : sz)];
                const float t = (iso - v0) / (v1 - v0);
                float p[3] = {(float)x, (float)y, (float)z};
                p[ax] += t;
                verts[3 * (long long)o + 0] = p[0] * sx;
                verts[3 * (long long)o + 1] = p[1] * sy_;
                verts[3 * (long long)o + 2] = p[2] * sz_;
                ++o;
            }
    }
    // ---- triangles of the cell
    if (x + 1 < W && y + 1 < H && z + 1 < D) {
        unsigned cs = v0 > iso ? 1u : 0u;
        cs |= vol[i + 1] > iso ? 2u : 0u;
        cs |= vol[i + sy] > iso ? 4u : 0u;
        cs |= vol[i + sy + 1] > iso ? 8u : 0u;
        cs |= vol[i + sz] > iso ? 16u : 0u;
        cs |= vol[i + sz + 1] > iso ? 32u : 0u;
        cs |= vol[i + sz + sy] > iso ? 64u : 0u;
        cs |= vol[i + sz + sy + 1] > iso ? 128u : 0u;
        const int nt = c_ntri[cs];
        if (nt) {
            const long long t0 = toff[i];
            for (int k = 0; k < 3 * nt; ++k) {
                const int e = c_tri[cs][k];
                const int ax = e >> 2, j = e & 3, a = j & 1, b = j >> 1;
                const int ox = ax == 0 ? 0 : a, oy = ax == 0 ? a : (ax == 1 ? 0 : b), oz = ax == 2 ? 0 : b;
                const long long owner = i + ox + oy * sy + oz * sz;
                const unsigned of = flags[owner];
                faces[3 * t0 + k] = voff[owner] + __popc(of & ((1u << ax) - 1u));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Laplacian smoothing (Jacobi sweep over the CSR edge graph)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) smooth_kernel(const float* __restrict__ in, float* __restrict__ out, long long n,
                                                     const int* __restrict__ off, const int* __restrict__ nbr, float relax) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = off[i], e = off[i + 1];
    const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    if (e == b) { out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z; return; }
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (int k = b; k < e; ++k) {
        const long long j = nbr[k];
        sx += in[3 * j]; sy += in[3 * j + 1]; sz += in[3 * j + 2];
    }
    const float d = (float)(e - b);
    out[3 * i] = x + relax * (sx / d - x);
    out[3 * i + 1] = y + relax * (sy / d - y);
    out[3 * i + 2] = z + relax * (sz / d - z);
}

// ---------------------------------------------------------------------------------------------------------------------
// unsigned distance from points to a triangle mesh: one lane per point, triangles streamed through LDS
// ---------------------------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// squared distance from p to triangle (a, a + ab, a + ac): Ericson, Real-Time Collision Detection 5.1.5
__device__ __forceinline__ float tri_dist2(V3 p, V3 a, V3 ab, V3 ac) {
    const V3 ap = p - a;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    V3 q;
    if (d1 <= 0.0f && d2 <= 0.0f) q = a;
    else {
        const V3 bp = ap - ab;
        const float d3 = dot(ab, bp), d4 = dot(ac, bp);
        if (d3 >= 0.0f && d4 <= d3) q = a + ab;
        else {
            const float vc = d1 * d4 - d3 * d2;
            if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) q = a + ab * (d1 / (d1 - d3));
            else {
                const V3 cp = ap - ac;
                const float d5 = dot(ab, cp), d6 = dot(ac, cp);
                if (d6 >= 0.0f && d5 <= d6) q = a + ac;
                else {
                    const float vb = d5 * d2 - d1 * d6;
                    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) q = a + ac * (d2 / (d2 - d6));
                    else {
                        const float va = d3 * d6 - d5 * d4;
                        if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
                            const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
                            q = a + ab + (ac - ab) * w;
                        } else {
                            const float denom = 1.0f / (va + vb + vc);
                            q = a + ab * (vb * denom) + ac * (vc * denom);
                        }
                    }
                }
            }
        }
    }
    const V3 d = p - q;
    return dot(d, d);
}

constexpr int kTriTile = 512;

__global__ void __launch_bounds__(256) point_distance_kernel(const float* __restrict__ pts, long long np, const float* __restrict__ verts,
                                                             const int* __restrict__ faces, long long nt, float* __restrict__ dist) {
    __shared__ float tri[kTriTile][9];          // a, ab, ac
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < np;
    const V3 p = live ? V3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]} : V3{0.f, 0.f, 0.f};
    float best = 3.4e38f;
    for (long long t0 = 0; t0 < nt; t0 += kTriTile) {
        const int cnt = (int)(nt - t0 < kTriTile ? nt - t0 : kTriTile);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const int* f = faces + 3 * (t0 + k);
            const float* a = verts + 3 * (long long)f[0];
            const float* b = verts + 3 * (long long)f[1];
            const float* c = verts + 3 * (long long)f[2];
            tri[k][0] = a[0]; tri[k][1] = a[1]; tri[k][2] = a[2];
            tri[k][3] = b[0] - a[0]; tri[k][4] = b[1] - a[1]; tri[k][5] = b[2] - a[2];
            tri[k][6] = c[0] - a[0]; tri[k][7] = c[1] - a[1]; tri[k][8] = c[2] - a[2];
        }
        __syncthreads();
        if (live)
            for (int k = 0; k < cnt; ++k)          // LDS broadcast reads: every lane wants the same triangle
                best = fminf(best, tri_dist2(p, V3{tri[k][0], tri[k][1], tri[k][2]}, V3{tri[k][3], tri[k][4], tri[k][5]}, V3{tri[k][6], tri[k][7], tri[k][8]}));
    }
    if (live) dist[i] = sqrtf(best);
}


// ---------------------------------------------------------------------------------------------------------------------
// the same distance with a uniform-grid broad phase: triangles are binned into the cells their bounding boxes overlap
// (count -> prefix sum -> fill), a point walks the cells around it ring by ring and stops when the best distance found is
// within the radius already covered.  Exact (same closest-point routine, every triangle that could be closer is visited).
// ---------------------------------------------------------------------------------------------------------------------
struct GridDesc {
    float lo[3];
    float inv_h, h;
    int n[3];
};

__device__ __forceinline__ int cell_coord(float p, float lo, float inv_h, int n) {
    const int c = (int)floorf((p - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

template <bool FILL>
__global__ void __launch_bounds__(256) grid_bin_kernel(const float* __restrict__ verts, const int* __restrict__ faces, long long nt, GridDesc g,
                                                       int* __restrict__ cell_count /*FILL: running fill cursor*/, const int* __restrict__ cell_start,
                                                       int* __restrict__ tri_list) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    int lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = verts[3 * (long long)faces[3 * t] + k], b = verts[3 * (long long)faces[3 * t + 1] + k], c = verts[3 * (long long)faces[3 * t + 2] + k];
        lo[k] = cell_coord(fminf(a, fminf(b, c)), g.lo[k], g.inv_h, g.n[k]);
        hi[k] = cell_coord(fmaxf(a, fmaxf(b, c)), g.lo[k], g.inv_h, g.n[k]);
    }
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const int cell = (z * g.n[1] + y) * g.n[0] + x;
                const int slot = atomicAdd(&cell_count[cell], 1);
                if (FILL) tri_list[cell_start[cell] + slot] = (int)t;
            }
}

__global__ void __launch_bounds__(256) grid_distance_kernel(const float* __restrict__ pts, long long np, const float* __restrict__ verts,
                                                            const int* __restrict__ faces, GridDesc g, const int* __restrict__ cell_start,
                                                            const int* __restrict__ tri_list, float* __restrict__ dist) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const V3 p = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    const int cx = cell_coord(p.x, g.lo[0], g.inv_h, g.n[0]), cy = cell_coord(p.y, g.lo[1], g.inv_h, g.n[1]), cz = cell_coord(p.z, g.lo[2], g.inv_h, g.n[2]);
    const int rmax = max(max(max(cx, g.n[0] - 1 - cx), max(cy, g.n[1] - 1 - cy)), max(cz, g.n[2] - 1 - cz));
    float best = 3.4e38f;
    for (int r = 0; r <= rmax; ++r) {
        // triangles not visited after ring r-1 lie in cells at Chebyshev distance >= r: at least (r-1) * h away from the point's
        // projection onto the grid box, hence from the point
        const float covered = (float)(r - 1) * g.h;
        if (r > 0 && best <= covered * covered) break;
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.n[2] - 1), y0 = max(cy - r, 0), y1 = min(cy + r, g.n[1] - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.n[0] - 1);
        auto visit = [&](int x, int y, int z) {
            const int cell = (z * g.n[1] + y) * g.n[0] + x;
            for (int k = cell_start[cell]; k < cell_start[cell + 1]; ++k) {
                const int* f = faces + 3 * (long long)tri_list[k];
                const float* a = verts + 3 * (long long)f[0];
                const float* b = verts + 3 * (long long)f[1];
                const float* c = verts + 3 * (long long)f[2];
                const V3 A = {a[0], a[1], a[2]};
                best = fminf(best, tri_dist2(p, A, V3{b[0], b[1], b[2]} - A, V3{c[0], c[1], c[2]} - A));
            }
        };
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                if (z == cz - r || z == cz + r || y == cy - r || y == cy + r) {      // a face of the ring's cube: the whole row
                    for (int x = x0; x <= x1; ++x) visit(x, y, z);
                } else {                                                             // otherwise only the two end cells are new
                    if (cx - r >= 0) visit(cx - r, y, z);
                    if (cx + r < g.n[0]) visit(cx + r, y, z);
                }
            }
    }
    dist[i] = sqrtf(best);
}

struct GridLayout { size_t count, start, list, scratch, total; };
GridLayout grid_layout(long long ncells, long long nt) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    GridLayout l;
    size_t o = 0;
    l.count = o; o += al((size_t)(ncells + 1) * 4);
    l.start = o; o += al((size_t)(ncells + 1) * 4);
    l.list = o; o += al((size_t)nt * 8 * 4);
    l.scratch = o; o += al(scan_scratch_ints(ncells + 1) * 4);
    l.total = o;
    return l;
}

struct McLayout {
    size_t flags, vcount, tcount, voff, toff, scratch, total;
};

McLayout mc_layout(long long n) {
    McLayout l;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t o = 0;
    l.flags = o; o += al((size_t)n);
    l.vcount = o; o += al((size_t)n * 4);
    l.tcount = o; o += al((size_t)n * 4);
    l.voff = o; o += al((size_t)n * 4);
    l.toff = o; o += al((size_t)n * 4);
    l.scratch = o; o += al(scan_scratch_ints(n) * 4);
    l.total = o;
    return l;
}

}  // namespace

extern "C" {

int oai_mc_table(signed char* out_256x16) {
    OAI_CHECK_ARG(out_256x16, "oai_mc_table: null pointer");
    std::memcpy(out_256x16, host_tables().tri, sizeof(host_tables().tri));
    return OAI_OK;
}

size_t oai_mc_workspace_bytes(int D, int H, int W) {
    if (D < 2 || H < 2 || W < 2) return 0;
    return mc_layout((long long)D * H * W).total;
}

int oai_mc_count(const float* vol_dev, int D, int H, int W, float iso, void* ws_dev, size_t ws_bytes,
                 long long* n_verts, long long* n_tris, void* stream) {
    OAI_CHECK_ARG(vol_dev && ws_dev && n_verts && n_tris, "oai_mc_count: null pointer");
    OAI_CHECK_ARG(D >= 2 && H >= 2 && W >= 2, "oai_mc_count: every axis needs at least 2 voxels");
    const long long n = (long long)D * H * W;
    OAI_CHECK_ARG(n < (1LL << 31) / 3, "oai_mc_count: volume too large for 32-bit vertex ids");
    const McLayout l = mc_layout(n);
    if (ws_bytes < l.total) return oai::set_error(OAI_ERR_WORKSPACE, "oai_mc_count: workspace %zu B < %zu B", ws_bytes, l.total);
    if (int rc = upload_tables()) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_dev;
    int* vcount = (int*)(ws + l.vcount); int* tcount = (int*)(ws + l.tcount);
    int* voff = (int*)(ws + l.voff); int* toff = (int*)(ws + l.toff);
    mc_classify_kernel<<<oai::cdiv(n, 256), 256, 0, st>>>(vol_dev, D, H, W, iso, (unsigned char*)(ws + l.flags), vcount, tcount);
    OAI_CHECK_LAUNCH();
    if (int rc = exclusive_scan(vcount, voff, n, (int*)(ws + l.scratch), st)) return rc;
    if (int rc = exclusive_scan(tcount, toff, n, (int*)(ws + l.scratch), st)) return rc;
    int last[4];
    OAI_CHECK_HIP(hipMemcpyAsync(&last[0], vcount + n - 1, 4, hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipMemcpyAsync(&last[1], voff + n - 1, 4, hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipMemcpyAsync(&last[2], tcount + n - 1, 4, hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipMemcpyAsync(&last[3], toff + n - 1, 4, hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipStreamSynchronize(st));                    // the caller sizes its output arrays from the counts
    *n_verts = (long long)last[0] + last[1];
    *n_tris = (long long)last[2] + last[3];
    return OAI_OK;
}

int oai_mc_emit(const float* vol_dev, int D, int H, int W, float iso, const float spacing_xyz[3], const void* ws_dev,
                float* verts_dev, int* faces_dev, void* stream) {
    OAI_CHECK_ARG(vol_dev && ws_dev && spacing_xyz, "oai_mc_emit: null pointer");
    OAI_CHECK_ARG(D >= 2 && H >= 2 && W >= 2, "oai_mc_emit: every axis needs at least 2 voxels");
    const long long n = (long long)D * H * W;
    const McLayout l = mc_layout(n);
    const char* ws = (const char*)ws_dev;
    mc_emit_kernel<<<oai::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(vol_dev, D, H, W, iso, spacing_xyz[0], spacing_xyz[1], spacing_xyz[2],
                                                                        (const unsigned char*)(ws + l.flags), (const int*)(ws + l.voff),
                                                                        (const int*)(ws + l.toff), verts_dev, faces_dev);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

int oai_mesh_smooth(const float* verts_in_dev, long long n_verts, const int* offsets_dev, const int* nbrs_dev, int iterations,
                    float relaxation, float* tmp_dev, float* verts_out_dev, void* stream) {
    OAI_CHECK_ARG(verts_in_dev && offsets_dev && nbrs_dev && tmp_dev && verts_out_dev, "oai_mesh_smooth: null pointer");
    OAI_CHECK_ARG(n_verts >= 0 && iterations >= 0, "oai_mesh_smooth: negative size");
    hipStream_t st = (hipStream_t)stream;
    if (n_verts == 0) return OAI_OK;
    if (iterations == 0) {
        OAI_CHECK_HIP(hipMemcpyAsync(verts_out_dev, verts_in_dev, (size_t)n_verts * 12, hipMemcpyDeviceToDevice, st));
        return OAI_OK;
    }
    // ping-pong so that the last sweep lands in verts_out
    const float* src = verts_in_dev;
    for (int it = 0; it < iterations; ++it) {
        float* dst = ((iterations - it) & 1) ? verts_out_dev : tmp_dev;
        smooth_kernel<<<oai::cdiv(n_verts, 256), 256, 0, st>>>(src, dst, n_verts, offsets_dev, nbrs_dev, relaxation);
        OAI_CHECK_LAUNCH();
        src = dst;
    }
    return OAI_OK;
}

int oai_mesh_point_distance(const float* points_dev, long long n_points, const float* verts_dev, const int* faces_dev,
                            long long n_tris, float* dist_dev, void* stream) {
    OAI_CHECK_ARG(points_dev && verts_dev && faces_dev && dist_dev, "oai_mesh_point_distance: null pointer");
    OAI_CHECK_ARG(n_points >= 0 && n_tris > 0, "oai_mesh_point_distance: needs at least one triangle");
    if (n_points == 0) return OAI_OK;
    point_distance_kernel<<<oai::cdiv(n_points, 256), 256, 0, (hipStream_t)stream>>>(points_dev, n_points, verts_dev, faces_dev, n_tris, dist_dev);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

size_t oai_mesh_grid_workspace_bytes(const int grid_dims_xyz[3], long long n_tris) {
    if (!grid_dims_xyz || n_tris <= 0) return 0;
    return grid_layout((long long)grid_dims_xyz[0] * grid_dims_xyz[1] * grid_dims_xyz[2], n_tris).total;
}

int oai_mesh_point_distance_grid(const float* points_dev, long long n_points, const float* verts_dev, const int* faces_dev, long long n_tris,
                                 const float grid_lo_xyz[3], float cell_size, const int grid_dims_xyz[3],
                                 void* workspace_dev, size_t workspace_bytes, float* dist_dev, void* stream) {
    OAI_CHECK_ARG(points_dev && verts_dev && faces_dev && dist_dev && grid_lo_xyz && grid_dims_xyz && workspace_dev, "oai_mesh_point_distance_grid: null pointer");
    OAI_CHECK_ARG(n_points >= 0 && n_tris > 0 && cell_size > 0.0f, "oai_mesh_point_distance_grid: needs triangles and a positive cell size");
    OAI_CHECK_ARG(grid_dims_xyz[0] > 0 && grid_dims_xyz[1] > 0 && grid_dims_xyz[2] > 0, "oai_mesh_point_distance_grid: empty grid");
    const long long ncells = (long long)grid_dims_xyz[0] * grid_dims_xyz[1] * grid_dims_xyz[2];
    OAI_CHECK_ARG(ncells < (1LL << 30), "oai_mesh_point_distance_grid: grid too fine");
    const GridLayout l = grid_layout(ncells, n_tris);
    if (workspace_bytes < l.total) return oai::set_error(OAI_ERR_WORKSPACE, "oai_mesh_point_distance_grid: workspace %zu B < %zu B", workspace_bytes, l.total);
    if (n_points == 0) return OAI_OK;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace_dev;
    int* count = (int*)(ws + l.count); int* start = (int*)(ws + l.start); int* list = (int*)(ws + l.list);
    GridDesc g;
    for (int k = 0; k < 3; ++k) { g.lo[k] = grid_lo_xyz[k]; g.n[k] = grid_dims_xyz[k]; }
    g.h = cell_size; g.inv_h = 1.0f / cell_size;
    OAI_CHECK_HIP(hipMemsetAsync(count, 0, (size_t)(ncells + 1) * 4, st));
    grid_bin_kernel<false><<<oai::cdiv(n_tris, 256), 256, 0, st>>>(verts_dev, faces_dev, n_tris, g, count, nullptr, nullptr);
    OAI_CHECK_LAUNCH();
    if (int rc = exclusive_scan(count, start, ncells + 1, (int*)(ws + l.scratch), st)) return rc;
    int total = 0;
    OAI_CHECK_HIP(hipMemcpyAsync(&total, start + ncells, 4, hipMemcpyDeviceToHost, st));
    OAI_CHECK_HIP(hipStreamSynchronize(st));
    if ((long long)total > n_tris * 8)          // a triangle longer than a cell overlaps more than 8 cells: the caller's cell size is too small
        return oai::set_error(OAI_ERR_ARG, "oai_mesh_point_distance_grid: %d triangle-cell pairs > 8 per triangle; cell_size must be >= the longest edge", total);
    OAI_CHECK_HIP(hipMemsetAsync(count, 0, (size_t)(ncells + 1) * 4, st));
    grid_bin_kernel<true><<<oai::cdiv(n_tris, 256), 256, 0, st>>>(verts_dev, faces_dev, n_tris, g, count, start, list);
    OAI_CHECK_LAUNCH();
    grid_distance_kernel<<<oai::cdiv(n_points, 256), 256, 0, st>>>(points_dev, n_points, verts_dev, faces_dev, g, start, list, dist_dev);
    OAI_CHECK_LAUNCH();
    return OAI_OK;
}

}  // extern "C"
