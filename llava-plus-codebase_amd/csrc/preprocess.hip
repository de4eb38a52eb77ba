// Image preprocessing on the device: process_images / expand2square (llava/mm_utils.py:16-44) + HF's
// CLIPImageProcessor (resize shortest edge with PIL BICUBIC -> center crop -> rescale 1/255 -> normalise) for one decoded RGB
// image already in HBM as uint8 [H][W][3].  The CPU path costs 10-20 ms per image on the host (PIL resize + numpy), in front of
// every multimodal request's prefill; here it is two small launches.
//
// Bit-exact with Pillow's 8-bit resampler (src/libImaging/Resample.c) by construction: the host computes Pillow's coefficient
// tables with the same double-precision expressions (bicubic a = -0.5, support 2 x max(scale, 1), normalised, 22-bit fixed point
// with round-half-away), the kernels do the integer accumulation with the 1 << 21 rounding term and clip to uint8 after EACH pass,
// horizontal pass first — including Pillow's rule that a pass whose size does not change is skipped (no rounding).  Only the
// columns / rows inside the center-crop window are produced.  expand2square is a virtual canvas: reads outside the pasted image
// return the fill colour.  The float tail is (u8 * (1/255) - mean) / std in fp32, then one rounding to the model dtype.
#include <cmath>
#include <vector>
#include "common.h"
#include "kernels.h"

namespace lmx {

constexpr int PRECISION_BITS = 32 - 8 - 2;

static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output indices [o0, o0 + on) of an in_size -> out_size resample
static int precompute(int in_size, int out_size, int o0, int on, std::vector<int>& bounds, std::vector<int>& kk) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)on * 2, 0);
    kk.assign((size_t)on * ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w((size_t)ksize);
    for (int i = 0; i < on; ++i) {
        const int xx = o0 + i;
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = bicubic_filter((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            kk[(size_t)i * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[(size_t)i * 2] = xmin; bounds[(size_t)i * 2 + 1] = xmax;
    }
    return ksize;
}

struct PreArgs {
    const uint8_t* src; int H, W;            // decoded image
    int CH, CW, oy, ox;                      // canvas (== image unless padded to a square) and paste offset
    int fill[3];
    int size;                                // output side (crop window)
    int left, top;                           // crop origin inside the resized image
    int need_h, need_v, ksize_h, ksize_v;
    const int *bounds_h, *kk_h, *bounds_v, *kk_v;
    uint8_t* tmp;                            // [CH][size][3] after the horizontal pass
    float mean[3], std[3];
};

__device__ __forceinline__ int canvas_px(const PreArgs& a, int y, int x, int c) {
    const int sy = y - a.oy, sx = x - a.ox;
    if (sy < 0 || sy >= a.H || sx < 0 || sx >= a.W) return a.fill[c];
    return a.src[((size_t)sy * a.W + sx) * 3 + c];
}
__device__ __forceinline__ int clip8(int v) { v >>= PRECISION_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass over every canvas row, crop-window columns only
__global__ __launch_bounds__(256) void pre_resize_h_kernel(PreArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.CH * a.size) return;
    const int y = idx / a.size, i = idx % a.size;
    uint8_t* o = a.tmp + ((size_t)y * a.size + i) * 3;
    if (!a.need_h) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)canvas_px(a, y, a.left + i, c);
        return;
    }
    const int x0 = a.bounds_h[i * 2], n = a.bounds_h[i * 2 + 1];
    const int* k = a.kk_h + (size_t)i * a.ksize_h;
    int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int t = 0; t < n; ++t) {
        const int kv = k[t];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += canvas_px(a, y, x0 + t, c) * kv;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)clip8(acc[c]);
}

// vertical pass (crop-window rows) + rescale + normalise + channel-first store
template <typename T>
__global__ __launch_bounds__(256) void pre_resize_v_norm_kernel(PreArgs a, T* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.size * a.size) return;
    const int j = idx / a.size, i = idx % a.size;
    int px[3];
    if (!a.need_v) {
        const uint8_t* s = a.tmp + ((size_t)(a.top + j) * a.size + i) * 3;
        px[0] = s[0]; px[1] = s[1]; px[2] = s[2];
    } else {
        const int y0 = a.bounds_v[j * 2], n = a.bounds_v[j * 2 + 1];
        const int* k = a.kk_v + (size_t)j * a.ksize_v;
        int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
        for (int t = 0; t < n; ++t) {
            const uint8_t* s = a.tmp + ((size_t)(y0 + t) * a.size + i) * 3;
            const int kv = k[t];
            acc[0] += s[0] * kv; acc[1] += s[1] * kv; acc[2] += s[2] * kv;
        }
        px[0] = clip8(acc[0]); px[1] = clip8(acc[1]); px[2] = clip8(acc[2]);
    }
    {
        // three separately rounded fp32 operations, like the host processor: no fused multiply-add, IEEE division
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float scaled = (float)px[c] * (1.0f / 255.0f);
            const float centred = scaled - a.mean[c];
            const float v = __fdiv_rn(centred, a.std[c]);
            out[((size_t)c * a.size + j) * a.size + i] = from_f32<T>(v);
        }
    }
}

// host-only view of the coefficient tables (CPU tests pin them to the oracle / Pillow without a GPU)
int preprocess_coeffs(int in_size, int out_size, int o0, int on, int* bounds_out, int* kk_out, int kk_cap) {
    std::vector<int> b, k;
    const int ksize = precompute(in_size, out_size, o0, on, b, k);
    if (bounds_out) for (size_t i = 0; i < b.size(); ++i) bounds_out[i] = b[i];
    if (kk_out && (int)k.size() <= kk_cap) for (size_t i = 0; i < k.size(); ++i) kk_out[i] = k[i];
    return ksize;
}

// scratch layout: [tables][tmp]; returns bytes needed
size_t launch_preprocess(int dtype, const uint8_t* rgb, int H, int W, int size, int pad_to_square, const float* mean, const float* std,
                         void* out, void* scratch, size_t scratch_bytes, hipStream_t st) {
    LMX_REQUIRE(H > 0 && W > 0 && size > 0, "preprocess: bad geometry");
    PreArgs a{};
    a.src = rgb; a.H = H; a.W = W; a.size = size;
    a.CH = H; a.CW = W; a.oy = 0; a.ox = 0;
    if (pad_to_square && H != W) {
        const int S = H > W ? H : W;
        a.CH = S; a.CW = S; a.oy = (S - H) / 2; a.ox = (S - W) / 2;
    }
    for (int c = 0; c < 3; ++c) { a.fill[c] = (int)(mean[c] * 255); a.mean[c] = mean[c]; a.std[c] = std[c]; }
    // HF get_resize_output_image_size(shortest_edge = size, default_to_square = False)
    int nh, nw;
    if (a.CW <= a.CH) { nw = size; nh = (int)((double)size * a.CH / a.CW); } else { nh = size; nw = (int)((double)size * a.CW / a.CH); }
    a.top = (nh - size) / 2; a.left = (nw - size) / 2;
    a.need_h = nw != a.CW; a.need_v = nh != a.CH;
    std::vector<int> bh, kh, bv, kv;
    a.ksize_h = a.need_h ? precompute(a.CW, nw, a.left, size, bh, kh) : 0;
    a.ksize_v = a.need_v ? precompute(a.CH, nh, a.top, size, bv, kv) : 0;
    const size_t n_tab = bh.size() + kh.size() + bv.size() + kv.size();
    const size_t tab_bytes = (n_tab * sizeof(int) + 255) / 256 * 256;
    const size_t tmp_bytes = (size_t)a.CH * size * 3;
    const size_t need = tab_bytes + tmp_bytes;
    if (!scratch || scratch_bytes < need) return need;
    std::vector<int> host;
    host.reserve(n_tab);
    host.insert(host.end(), bh.begin(), bh.end()); host.insert(host.end(), kh.begin(), kh.end());
    host.insert(host.end(), bv.begin(), bv.end()); host.insert(host.end(), kv.begin(), kv.end());
    int* d = static_cast<int*>(scratch);
    if (n_tab) LMX_CHECK_HIP(hipMemcpyAsync(d, host.data(), n_tab * sizeof(int), hipMemcpyHostToDevice, st));
    a.bounds_h = d; a.kk_h = d + bh.size(); a.bounds_v = a.kk_h + kh.size(); a.kk_v = a.bounds_v + bv.size();
    a.tmp = static_cast<uint8_t*>(scratch) + tab_bytes;
    hipLaunchKernelGGL(pre_resize_h_kernel, dim3(cdiv(a.CH * size, 256)), dim3(256), 0, st, a);
#define L(TT) hipLaunchKernelGGL(pre_resize_v_norm_kernel<TT>, dim3(cdiv(size * size, 256)), dim3(256), 0, st, a, (TT*)out)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
    LMX_CHECK_HIP(hipStreamSynchronize(st));     // the host coefficient tables must outlive their upload
    return need;
}

}  // namespace lmx
