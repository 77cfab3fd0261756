// Intensity windowing (the step just before the hot path):
//   image_normalize(image, 0.1, 99.9, 0, 1)   oai_analysis/dask_processing.py:10-26, called at :75 and :177
//     window_min/max = np.percentile(array, q)            (exact order statistics + linear interpolation)
//     itk.IntensityWindowingImageFilter[F,F]              (x<wmin -> omin; x>wmax -> omax; else x*factor+offset in double)
// On the device the two percentiles are found EXACTLY by a 4-pass 8-bit radix select over the order-preserving
// integer image of the floats (LDS-privatised histograms, 4 ranks at once: k_lo, k_lo+1, k_hi, k_hi+1), then one
// streaming pass applies the window.  HBM-bound: 5 reads + 1 write of the volume.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRanks = 4;

struct SelectState {                 // lives in the caller's workspace
    unsigned prefix[kRanks];         // key bits fixed so far (high bits)
    unsigned long long rank[kRanks]; // remaining rank inside the current prefix bucket
    unsigned hist[kRanks][256];
    float value[kRanks];             // result: the order statistics
    float window[2];                 // interpolated percentiles (wmin, wmax)
};

__device__ __forceinline__ unsigned key_of(float f) {      // monotone float -> uint map
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void select_init_kernel(SelectState* st, unsigned long long r0, unsigned long long r1, unsigned long long r2, unsigned long long r3) {
    const int t = threadIdx.x;
    if (t < kRanks) { st->prefix[t] = 0; st->rank[t] = t == 0 ? r0 : t == 1 ? r1 : t == 2 ? r2 : r3; }
    for (int i = t; i < kRanks * 256; i += blockDim.x) st->hist[i / 256][i % 256] = 0;
}

// pass p (0 = most significant byte): histogram of byte p among elements whose higher bytes equal prefix[r]
__global__ void __launch_bounds__(kThreads) select_hist_kernel(const float* __restrict__ x, size_t n, int pass, SelectState* st) {
    __shared__ unsigned h[kRanks][256];
    for (int i = threadIdx.x; i < kRanks * 256; i += kThreads) h[i / 256][i % 256] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned mask = pass == 0 ? 0u : 0xffffffffu << (shift + 8);
    unsigned pre[kRanks];
#pragma unroll
    for (int r = 0; r < kRanks; ++r) pre[r] = st->prefix[r];
    const bool same01 = pre[0] == pre[1], same23 = pre[2] == pre[3], same02 = pre[0] == pre[2];
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const unsigned k = key_of(x[i]);
        const unsigned hi = k & mask, d = (k >> shift) & 255u;
        // ranks that share a prefix share a histogram row (k and k+1 almost always do): count once, copy later
        if (hi == pre[0]) atomicAdd(&h[0][d], 1u);
        if (!same01 && hi == pre[1]) atomicAdd(&h[1][d], 1u);
        if (!same02 && hi == pre[2]) atomicAdd(&h[2][d], 1u);
        if (!same23 && !(pre[3] == pre[0]) && hi == pre[3]) atomicAdd(&h[3][d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kRanks * 256; i += kThreads) {
        const unsigned v = h[i / 256][i % 256];
        if (v) atomicAdd(&st->hist[i / 256][i % 256], v);
    }
}

// one block: per rank, find the bin holding the rank, extend the prefix, clear the histograms for the next pass
__global__ void select_scan_kernel(int pass, SelectState* st) {
    __shared__ unsigned pre[kRanks];
    if (threadIdx.x < kRanks) pre[threadIdx.x] = st->prefix[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < kRanks) {
        const int r = threadIdx.x;
        // the histogram row this rank's prefix was counted in (see select_hist_kernel)
        int row = r;
        if (r == 1 && pre[1] == pre[0]) row = 0;
        if (r == 2 && pre[2] == pre[0]) row = 0;
        if (r == 3) row = pre[3] == pre[0] ? 0 : (pre[3] == pre[2] ? (pre[2] == pre[0] ? 0 : 2) : 3);
        unsigned long long rem = st->rank[r];
        int d = 0;
        for (; d < 255; ++d) {
            const unsigned c = st->hist[row][d];
            if (rem < c) break;
            rem -= c;
        }
        const int shift = 24 - 8 * pass;
        st->rank[r] = rem;
        st->prefix[r] = pre[r] | ((unsigned)d << shift);
        if (pass == 3) st->value[r] = float_of(pre[r] | (unsigned)d);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kRanks * 256; i += blockDim.x) st->hist[i / 256][i % 256] = 0;
}

// numpy's _lerp in the array dtype (float32): a + (b-a)*t for t < 0.5, else b - (b-a)*(1-t); no FMA contraction
__global__ void window_params_kernel(SelectState* st, float g_lo, float g_hi) {
    if (threadIdx.x < 2) {
        const float a = st->value[2 * threadIdx.x], b = st->value[2 * threadIdx.x + 1];
        const float t = threadIdx.x == 0 ? g_lo : g_hi;
        const float diff = __fsub_rn(b, a);
        st->window[threadIdx.x] = t < 0.5f ? __fadd_rn(a, __fmul_rn(diff, t)) : __fsub_rn(b, __fmul_rn(diff, __fsub_rn(1.0f, t)));
    }
}

__global__ void __launch_bounds__(kThreads) window_apply_kernel(const float* __restrict__ x, size_t n, const SelectState* st,
                                                                float omin, float omax, float* __restrict__ out) {
    const float wmin = st->window[0], wmax = st->window[1];
    const double factor = ((double)omax - (double)omin) / ((double)wmax - (double)wmin);
    const double offset = (double)omin - (double)wmin * factor;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = r[j] < wmin ? omin : (r[j] > wmax ? omax : (float)((double)r[j] * factor + offset));
        reinterpret_cast<float4*>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const float v = x[i];
        out[i] = v < wmin ? omin : (v > wmax ? omax : (float)((double)v * factor + offset));
    }
}

// np.percentile(a, q) on a float32 array, numpy >= 2 semantics: the quantile, the virtual index and gamma are float32
void numpy_virtual_index(size_t n, float pct, unsigned long long& k0, unsigned long long& k1, float& gamma) {
    const float q = pct / 100.0f;                       // np.true_divide(q, a.dtype.type(100))
    const float vi = (float)(n - 1) * q;                // (n - 1) * quantiles
    float fl = floorf(vi);
    if (fl < 0.0f) fl = 0.0f;
    unsigned long long k = (unsigned long long)fl;
    gamma = vi - fl;
    if (k >= n - 1) { k = n - 1; gamma = 0.0f; }        // virtual_indexes >= n-1 -> the last element
    k0 = k;
    k1 = k + 1 < n ? k + 1 : n - 1;
}

}  // namespace

extern "C" {

size_t oai_image_normalize_workspace_bytes(void) { return (sizeof(SelectState) + 255) / 256 * 256; }

int oai_image_normalize(const float* in, size_t n, float pct_lo, float pct_hi, float out_min, float out_max,
                        float* out, float* window_out_dev, void* ws, size_t ws_bytes, void* stream) {
    OAI_CHECK_ARG(in && out && ws, "oai_image_normalize: null pointer");
    OAI_CHECK_ARG(n >= 2, "oai_image_normalize: need at least 2 voxels");
    OAI_CHECK_ARG(pct_lo >= 0.0f && pct_hi <= 100.0f && pct_lo < pct_hi, "oai_image_normalize: percentiles must satisfy 0 <= lo < hi <= 100");
    OAI_CHECK_ARG((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "oai_image_normalize: buffers must be 16-byte aligned");
    if (ws_bytes < oai_image_normalize_workspace_bytes())
        return oai::set_error(OAI_ERR_WORKSPACE, "oai_image_normalize: workspace %zu B < %zu B", ws_bytes, oai_image_normalize_workspace_bytes());
    hipStream_t st = (hipStream_t)stream;
    SelectState* s = reinterpret_cast<SelectState*>(ws);
    unsigned long long k[4];
    float g_lo, g_hi;
    numpy_virtual_index(n, pct_lo, k[0], k[1], g_lo);
    numpy_virtual_index(n, pct_hi, k[2], k[3], g_hi);
    select_init_kernel<<<1, 256, 0, st>>>(s, k[0], k[1], k[2], k[3]);
    OAI_CHECK_LAUNCH();
    size_t blocks = (n + kThreads * 16 - 1) / (kThreads * 16);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    for (int pass = 0; pass < 4; ++pass) {
        select_hist_kernel<<<(unsigned)blocks, kThreads, 0, st>>>(in, n, pass, s);
        OAI_CHECK_LAUNCH();
        select_scan_kernel<<<1, 256, 0, st>>>(pass, s);
        OAI_CHECK_LAUNCH();
    }
    window_params_kernel<<<1, 64, 0, st>>>(s, g_lo, g_hi);
    OAI_CHECK_LAUNCH();
    window_apply_kernel<<<(unsigned)blocks, kThreads, 0, st>>>(in, n, s, out_min, out_max, out);
    OAI_CHECK_LAUNCH();
    if (window_out_dev) OAI_CHECK_HIP(hipMemcpyAsync(window_out_dev, s->window, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return OAI_OK;
}

}  // extern "C"
