// Diagnostic: where do the 23 us of the 160^3 image warp go?  Streams the same buffers with and without the gather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
extern "C" int oai_grid_sample3d(const float* src, int C, int d, int h, int w, const float* coords, int D, int H, int W, float* out, void* stream);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct __attribute__((packed, aligned(4))) pair_f32 { float a, b; };

template <int VAR, bool XCD = false>
__global__ void __launch_bounds__(256) k(const float* __restrict__ src, const float* __restrict__ coords, float* __restrict__ out, int D, int H, int W) {
    const int plane = D * H * W;
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) & 3, tz = threadIdx.x >> 7;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (XCD) {   // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD one contiguous z-slab of bricks
        const int nb = gridDim.x * gridDim.y * gridDim.z, L = bx + gridDim.x * (by + gridDim.y * bz);
        const int per = (nb + 7) >> 3, logical = (L & 7) * per + (L >> 3);
        if (logical >= nb) return;
        bx = logical % gridDim.x; by = (logical / gridDim.x) % gridDim.y; bz = logical / (gridDim.x * gridDim.y);
    }
    const int x = bx * 32 + tx, y = by * 4 + ty;
    constexpr int U = 2;
    int lin[U]; float cz[U], cy[U], cx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int z = (XCD ? bz * U + u : bz + u * gridDim.z) * 2 + tz;
        lin[u] = (z * H + y) * W + x;
        cz[u] = coords[lin[u]]; cy[u] = coords[plane + lin[u]]; cx[u] = coords[2 * plane + lin[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float r;
        if (VAR == 0) r = cz[u] + cy[u] + cx[u];
        else {
            const float iz = fminf(D - 1.f, fmaxf(cz[u] * (D - 1), 0.f)), iy = fminf(H - 1.f, fmaxf(cy[u] * (H - 1), 0.f)), ix = fminf(W - 1.f, fmaxf(cx[u] * (W - 1), 0.f));
            const float fz = floorf(iz), fy = floorf(iy), fx = floorf(ix);
            int z0 = (int)fz, y0 = (int)fy, x0 = min((int)fx, W - 2);
            const int z1 = min(z0 + 1, D - 1), y1 = min(y0 + 1, H - 1);
            const float wx1 = ix - x0, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1, wz1 = iz - fz, wz0 = 1.f - wz1;
            if (VAR == 1) {
                const pair_f32 p00 = *reinterpret_cast<const pair_f32*>(src + (z0 * H + y0) * W + x0);
                const pair_f32 p01 = *reinterpret_cast<const pair_f32*>(src + (z0 * H + y1) * W + x0);
                const pair_f32 p10 = *reinterpret_cast<const pair_f32*>(src + (z1 * H + y0) * W + x0);
                const pair_f32 p11 = *reinterpret_cast<const pair_f32*>(src + (z1 * H + y1) * W + x0);
                r = ((p00.a * wx0 + p00.b * wx1) * wy0 + (p01.a * wx0 + p01.b * wx1) * wy1) * wz0 + ((p10.a * wx0 + p10.b * wx1) * wy0 + (p11.a * wx0 + p11.b * wx1) * wy1) * wz1;
            } else {   // VAR 2: one dword per corner pair (half the gather bytes), tests whether the gather is the cost
                const float p00 = src[(z0 * H + y0) * W + x0], p01 = src[(z0 * H + y1) * W + x0], p10 = src[(z1 * H + y0) * W + x0], p11 = src[(z1 * H + y1) * W + x0];
                r = (p00 * wy0 + p01 * wy1) * wz0 + (p10 * wy0 + p11 * wy1) * wz1 + wx0 * 0.f;
            }
        }
        out[lin[u]] = r;
    }
}

// persistent, software-pipelined: each workgroup walks its XCD's z-slab brick by brick; the next brick's coordinates are
// requested before the current brick's gathers are consumed
template <bool NT>
__global__ void __launch_bounds__(256) kp(const float* __restrict__ src, const float* __restrict__ coords, float* __restrict__ out, int D, int H, int W) {
    const int plane = D * H * W;
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) & 3, tz = threadIdx.x >> 7;
    const int nbx = W / 32, nby = H / 4, nbz = D / 2, nb = nbx * nby * nbz;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int per = (nb + 7) >> 3, b0 = xcd * per, b1 = min(nb, b0 + per);
    auto lin_of = [&](int b) { const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby); return ((bz * 2 + tz) * H + by * 4 + ty) * W + bx * 32 + tx; };
    int b = b0 + slot;
    if (b >= b1) return;
    int lin = lin_of(b);
    float cz = coords[lin], cy = coords[plane + lin], cx = coords[2 * plane + lin];
    while (true) {
        const int bn = b + nslot;
        const bool more = bn < b1;
        const int linn = more ? lin_of(bn) : lin;
        const float ncz = coords[linn], ncy = coords[plane + linn], ncx = coords[2 * plane + linn];
        const float iz = fminf(D - 1.f, fmaxf(cz * (D - 1), 0.f)), iy = fminf(H - 1.f, fmaxf(cy * (H - 1), 0.f)), ix = fminf(W - 1.f, fmaxf(cx * (W - 1), 0.f));
        const float fz = floorf(iz), fy = floorf(iy), fx = floorf(ix);
        int z0 = (int)fz, y0 = (int)fy, x0 = min((int)fx, W - 2);
        const int z1 = min(z0 + 1, D - 1), y1 = min(y0 + 1, H - 1);
        const float wx1 = ix - x0, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1, wz1 = iz - fz, wz0 = 1.f - wz1;
        const pair_f32 p00 = *reinterpret_cast<const pair_f32*>(src + (z0 * H + y0) * W + x0);
        const pair_f32 p01 = *reinterpret_cast<const pair_f32*>(src + (z0 * H + y1) * W + x0);
        const pair_f32 p10 = *reinterpret_cast<const pair_f32*>(src + (z1 * H + y0) * W + x0);
        const pair_f32 p11 = *reinterpret_cast<const pair_f32*>(src + (z1 * H + y1) * W + x0);
        const float r = ((p00.a * wx0 + p00.b * wx1) * wy0 + (p01.a * wx0 + p01.b * wx1) * wy1) * wz0 + ((p10.a * wx0 + p10.b * wx1) * wy0 + (p11.a * wx0 + p11.b * wx1) * wy1) * wz1;
        if (NT) __builtin_nontemporal_store(r, out + lin); else out[lin] = r;
        if (!more) break;
        b = bn; lin = linn; cz = ncz; cy = ncy; cx = ncx;
    }
}

// brick-shape study: BX x BY x BZ lanes per block, XCD-contiguous, U z-adjacent bricks
template <int BX, int BY, int BZ, int U>
__global__ void __launch_bounds__(256) kb(const float* __restrict__ src, const float* __restrict__ coords, float* __restrict__ out, int D, int H, int W) {
    static_assert(BX * BY * BZ == 256, "256 lanes");
    const int plane = D * H * W;
    const int tx = threadIdx.x % BX, ty = (threadIdx.x / BX) % BY, tz = threadIdx.x / (BX * BY);
    const int nbx = W / BX, nby = H / BY, nbz = D / (BZ * U), nb = nbx * nby * nbz;
    const int per = (nb + 7) >> 3, logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || logical >= nb) return;
    const int bx = logical % nbx, by = (logical / nbx) % nby, bz = logical / (nbx * nby);
    const int x = bx * BX + tx, y = by * BY + ty;
    int lin[U]; float cz[U], cy[U], cx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int z = (bz * U + u) * BZ + tz;
        lin[u] = (z * H + y) * W + x;
        cz[u] = coords[lin[u]]; cy[u] = coords[plane + lin[u]]; cx[u] = coords[2 * plane + lin[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float iz = fminf(D - 1.f, fmaxf(cz[u] * (D - 1), 0.f)), iy = fminf(H - 1.f, fmaxf(cy[u] * (H - 1), 0.f)), ix = fminf(W - 1.f, fmaxf(cx[u] * (W - 1), 0.f));
        const float fz = floorf(iz), fy = floorf(iy), fx = floorf(ix);
        int z0 = (int)fz, y0 = (int)fy, x0 = min((int)fx, W - 2);
        const int z1 = min(z0 + 1, D - 1), y1 = min(y0 + 1, H - 1);
        const float wx1 = ix - x0, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1, wz1 = iz - fz, wz0 = 1.f - wz1;
        const pair_f32 p00 = *reinterpret_cast<const pair_f32*>(src + (z0 * H + y0) * W + x0);
        const pair_f32 p01 = *reinterpret_cast<const pair_f32*>(src + (z0 * H + y1) * W + x0);
        const pair_f32 p10 = *reinterpret_cast<const pair_f32*>(src + (z1 * H + y0) * W + x0);
        const pair_f32 p11 = *reinterpret_cast<const pair_f32*>(src + (z1 * H + y1) * W + x0);
        const float r = ((p00.a * wx0 + p00.b * wx1) * wy0 + (p01.a * wx0 + p01.b * wx1) * wy1) * wz0 + ((p10.a * wx0 + p10.b * wx1) * wy0 + (p11.a * wx0 + p11.b * wx1) * wy1) * wz1;
        __builtin_nontemporal_store(r, out + lin[u]);
    }
}

// VAR 3: plain linear streaming, float4 per lane, grid-stride (the copy-like floor for these bytes)
__global__ void __launch_bounds__(256) stream4(const float4* __restrict__ c, float4* __restrict__ out, int n4) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 a = c[i], b = c[n4 + i], d = c[2 * n4 + i];
        out[i] = make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w);
    }
}

int main() {
    const int N = 160, V = N * N * N;
    std::vector<float> hc(3 * (size_t)V), hs(V);
    for (int z = 0; z < N; ++z) for (int y = 0; y < N; ++y) for (int x = 0; x < N; ++x) {
        const size_t i = ((size_t)z * N + y) * N + x;
        hs[i] = sinf(0.1f * x) * cosf(0.07f * y) + 0.01f * z;
        hc[i] = z / (N - 1.f) + 0.02f * sinf(0.05f * y); hc[V + i] = y / (N - 1.f) + 0.02f * sinf(0.04f * x); hc[2 * (size_t)V + i] = x / (N - 1.f) + 0.02f * cosf(0.05f * z);
    }
    if (FILE* f = fopen("/tmp/warp_coords.bin", "rb")) { size_t n = fread(hc.data(), 4, hc.size(), f); fclose(f); printf("coords from file (%zu floats)\n", n); }
    if (FILE* f = fopen("/tmp/warp_src.bin", "rb")) { size_t n = fread(hs.data(), 4, hs.size(), f); fclose(f); (void)n; }
    float *src, *coords, *out;
    CK(hipMalloc(&src, V * 4)); CK(hipMalloc(&coords, 3 * (size_t)V * 4)); CK(hipMalloc(&out, V * 4));
    CK(hipMemcpy(src, hs.data(), V * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(coords, hc.data(), 3 * (size_t)V * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid(N / 32, N / 4, N / 4);
    auto run = [&](int var) {
        for (int it = 0; it < 55; ++it) {
            if (it == 5) hipEventRecord(e0, 0);
            if (var == 0) k<0><<<grid, 256>>>(src, coords, out, N, N, N);
            else if (var == 1) k<1><<<grid, 256>>>(src, coords, out, N, N, N);
            else if (var == 2) k<2><<<grid, 256>>>(src, coords, out, N, N, N);
            else if (var == 3) stream4<<<256 * 8, 256>>>((const float4*)coords, (float4*)out, V / 4);
            else if (var == 5) k<1, true><<<grid, 256>>>(src, coords, out, N, N, N);
            else if (var == 6) kp<false><<<2048, 256>>>(src, coords, out, N, N, N);
            else if (var == 7) kp<true><<<2048, 256>>>(src, coords, out, N, N, N);
            else if (var == 8) kp<false><<<4096, 256>>>(src, coords, out, N, N, N);
            else if (var == 9) kb<32, 4, 2, 2><<<((N / 32) * (N / 4) * (N / 4) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 10) kb<32, 2, 4, 2><<<((N / 32) * (N / 2) * (N / 8) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 11) kb<16, 4, 4, 2><<<((N / 16) * (N / 4) * (N / 8) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 12) kb<16, 8, 2, 2><<<((N / 16) * (N / 8) * (N / 4) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 13) kb<8, 8, 4, 2><<<((N / 8) * (N / 8) * (N / 8) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 14) kb<32, 8, 1, 4><<<((N / 32) * (N / 8) * (N / 4) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 15) kb<32, 4, 2, 1><<<((N / 32) * (N / 4) * (N / 2) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 16) kb<32, 4, 2, 4><<<((N / 32) * (N / 4) * (N / 8) + 7) / 8 * 8, 256>>>(src, coords, out, N, N, N);
            else if (var == 4) oai_grid_sample3d(src, 1, N, N, N, coords, N, N, N, out, nullptr);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms / 50 * 1e3;
    };
    const char* names[] = {"brick stream (12 B in, 4 B out), no gather", "brick + 4 pair gathers (the shipped scheme)", "brick + 4 dword gathers", "linear float4 stream, grid-stride", "liboai_hip oai_grid_sample3d, same buffers", "pair gathers + XCD-contiguous z-slabs", "persistent pipelined, 2048 WGs", "persistent pipelined, 2048 WGs, nt stores", "persistent pipelined, 4096 WGs", "kb 32x4x2 U2", "kb 32x2x4 U2", "kb 16x4x4 U2", "kb 16x8x2 U2", "kb 8x8x4 U2", "kb 32x8x1 U4", "kb 32x4x2 U1", "kb 32x4x2 U4"};
    for (int v = 0; v < 17; ++v) { const double us = run(v); printf("%-48s %7.1f us  %7.1f GB/s (of 16 or 20 B/voxel)\n", names[v], us, (v == 0 || v == 3 ? 16.0 : 20.0) * V / us / 1e3); }
    return 0;
}
