// Row-wise / elementwise kernels of the LLaVA forward path (all HBM-bound: 16-byte vector accesses, one pass).
//
//   rmsnorm_kernel        LlamaRMSNorm                      HF5:models/llama/modeling_llama.py:53-67
//   layernorm_kernel      CLIP LayerNorm (eps 1e-5)         HF5:models/clip/modeling_clip.py:353-384,605-607
//   rope_kv_kernel        rotate_half RoPE + KV-cache write HF5:models/llama/modeling_llama.py:130-160,243-281
//   gather_embed_kernel   embed_tokens + image-feature splice, device half of llava/model/llava_arch.py:150-225
//   im2col / clip_embed_ln  CLIPVisionEmbeddings + pre_layrnorm   HF5:models/clip/modeling_clip.py:138-218,642
//   argmax_kernel         greedy sampling (temperature <= 0.001, llava/serve/model_worker.py:161)
#include "common.h"
#include "kernels.h"
#include "engine.h"

namespace lmx {

// ---------------------------------------------------------------------------------------------------------------
// RMSNorm: y = w * round_T(x * rsqrt(mean(x^2) + eps))   (fp32 statistics, HF rounding points)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y,
                                                      int H, int ldx, int ldy, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const T* xr = x + (size_t)row * ldx;
    T* yr = y + (size_t)row * ldy;
    const int HC = H >> 3;
    float ss = 0.f;
    for (int c = tid; c < HC; c += 256) {
        float v[8]; load8<T>(xr + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    }
    ss = block_sum<4>(ss, red);
    const float inv = rsqrtf(ss / (float)H + eps);
    for (int c = tid; c < HC; c += 256) {
        float v[8], g[8]; load8<T>(xr + c * 8, v); load8<T>(w + c * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * g[e];
        store8<T>(yr + c * 8, v);
    }
}

// The same kernel with the row held in registers between the two passes (H <= 8192: at most four 16-byte pieces per thread): ONE read of x instead of two — the second
// loop above re-read the row through L1 / L2, a dependent round trip in a launch that is all latency at 32 rows (a decode batch: 65 launches per step) and a third of the
// traffic at 32768 (a training step).  Same summation order (a thread adds its pieces in the order tid, tid + 256, ...), same bits.
template <typename T, int NP>
__global__ __launch_bounds__(256) void rmsnorm_reg_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int H, int ldx, int ldy, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const T* xr = x + (size_t)row * ldx;
    T* yr = y + (size_t)row * ldy;
    const int HC = H >> 3;
    float v[NP][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = tid + i * 256;
        if (c < HC) {
            load8<T>(xr + c * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
        }
    }
    ss = block_sum<4>(ss, red);
    const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = tid + i * 256;
        if (c < HC) {
            float g[8], o[8]; load8<T>(w + c * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = round_to<T>(v[i][e] * inv) * g[e];
            store8<T>(yr + c * 8, o);
        }
    }
}

void launch_rmsnorm(int dtype, const void* x, const void* w, void* y, int rows, int H, int ldx, int ldy, float eps, hipStream_t st) {
    LMX_REQUIRE(H % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "rmsnorm: hidden size / strides must be multiples of 8");
    if (rows <= 0) return;
    if (H <= 8192) {
        const int np = (H / 8 + 255) / 256;            // 16-byte pieces per thread: 1 .. 4
#define R(TT, NPV) hipLaunchKernelGGL((rmsnorm_reg_kernel<TT, NPV>), dim3(rows), dim3(256), 0, st, (const TT*)x, (const TT*)w, (TT*)y, H, ldx, ldy, eps)
#define RT(TT) do { if (np == 1) R(TT, 1); else if (np == 2) R(TT, 2); else if (np == 3) R(TT, 3); else R(TT, 4); } while (0)
        if (dtype == kBF16) RT(bf16_t); else if (dtype == kF16) RT(f16_t); else RT(float);
#undef RT
#undef R
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
#define L(TT) hipLaunchKernelGGL(rmsnorm_kernel<TT>, dim3(rows), dim3(256), 0, st, (const TT*)x, (const TT*)w, (TT*)y, H, ldx, ldy, eps)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: y = (x - mean) * rsqrt(var + eps) * w + b   (fp32 statistics, one rounding at the end like torch)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b,
                                                        T* __restrict__ y, int H, int ldx, int ldy, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const T* xr = x + (size_t)row * ldx;
    T* yr = y + (size_t)row * ldy;
    const int HC = H >> 3;
    float s = 0.f;
    for (int c = tid; c < HC; c += 256) {
        float v[8]; load8<T>(xr + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    const float mean = block_sum<4>(s, red) / (float)H;
    float q = 0.f;
    for (int c = tid; c < HC; c += 256) {
        float v[8]; load8<T>(xr + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(block_sum<4>(q, red) / (float)H + eps);
    for (int c = tid; c < HC; c += 256) {
        float v[8], g[8], bb[8]; load8<T>(xr + c * 8, v); load8<T>(w + c * 8, g); load8<T>(b + c * 8, bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * g[e] + bb[e];
        store8<T>(yr + c * 8, v);
    }
}

// The CLIP tower's widths (H = 512 NC elements, NC <= 4: ViT-L/14 has 1024): ONE WAVE per row, four rows per workgroup, the row in registers between the passes, no
// barrier.  The 577-row launches of the tower are latency, not bandwidth (46 per image, 6.6 us each in the in-situ profile for 2.4 MB of traffic): a row per
// 256-thread workgroup left half the threads without an element and paid four workgroup barriers.  Same bits as layernorm_kernel: lane l holds the chunks
// l, l + 64, ... that threads l, l + 64, ... of that kernel hold; each chunk group is reduced by the same wave_sum, the groups are added in wave order as
// block_sum adds its waves (empty waves add 0).
template <typename T, int NC>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b, T* __restrict__ y,
                                                             int rows, int ldx, int ldy, float eps) {
    constexpr int H = 512 * NC;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * ldx;
    float v[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) load8<T>(xr + (c * 64 + lane) * 8, v[c]);
    float tot = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e];
        tot += wave_sum(s);
    }
    const float mean = tot / (float)H;
    float qt = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
        qt += wave_sum(q);
    }
    const float rstd = rsqrtf(qt / (float)H + eps);
    T* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float g[8], bb[8], o[8];
        load8<T>(w + (c * 64 + lane) * 8, g); load8<T>(b + (c * 64 + lane) * 8, bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd * g[e] + bb[e];
        store8<T>(yr + (c * 64 + lane) * 8, o);
    }
}

void launch_layernorm(int dtype, const void* x, const void* w, const void* b, void* y, int rows, int H, int ldx, int ldy, float eps, hipStream_t st) {
    LMX_REQUIRE(H % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm: hidden size / strides must be multiples of 8");
    if (rows <= 0) return;
    if (dtype != kF32 && H % 512 == 0 && H <= 2048 && H != 1536) {
#define LW(TT, NC) hipLaunchKernelGGL((layernorm_wave_kernel<TT, NC>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const TT*)x, (const TT*)w, (const TT*)b, (TT*)y, rows, ldx, ldy, eps)
#define LWN(TT) do { if (H == 512) LW(TT, 1); else if (H == 1024) LW(TT, 2); else LW(TT, 4); } while (0)
        if (dtype == kBF16) LWN(bf16_t); else LWN(f16_t);
#undef LWN
#undef LW
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
#define L(TT) hipLaunchKernelGGL(layernorm_kernel<TT>, dim3(rows), dim3(256), 0, st, (const TT*)x, (const TT*)w, (const TT*)b, (TT*)y, H, ldx, ldy, eps)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE + KV-cache write. grid = (ceil(T/64), n_heads + 2*n_kv_heads); 256 threads.
//   head slot < n_heads                : q, rotated in place
//   n_heads <= slot < n_heads + n_kv   : k, rotated, written to K cache [kvh][pos][D]
//   otherwise                          : v, transposed through LDS into Vᵀ cache [kvh][d][pos] (128-byte key runs)
// HF rounding chain (model dtype T): cos,sin -> T ; out = T( T(x*cos) + T(rot(x)*sin) ).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void rope_kv_kernel(RopeKvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int tok0 = blockIdx.x * 64;
    const int slot = blockIdx.y;
    const int pos0 = a.pos_ptr ? *a.pos_ptr : a.pos0;
    T* qkv = reinterpret_cast<T*>(a.QKV);
    constexpr int HALF = D / 2;
    constexpr int GPR = HALF / 8;              // 8-pair groups per token row
    const int ntok = (a.T - tok0) < 64 ? (a.T - tok0) : 64;

    if (slot < a.n_heads + a.n_kv_heads) {
        const bool is_k = slot >= a.n_heads;
        const int col0 = is_k ? (a.n_heads * D + (slot - a.n_heads) * D) : slot * D;
        T* kc = reinterpret_cast<T*>(a.K) + (size_t)(is_k ? slot - a.n_heads : 0) * a.s_max * D;
        for (int g = tid; g < 64 * GPR; g += 256) {
            const int t = g / GPR, i0 = (g % GPR) * 8;
            if (t >= ntok) continue;
            const int pos = pos0 + tok0 + t;
            T* xp = qkv + (size_t)(tok0 + t) * a.qkv_stride + col0;
            float x1[8], x2[8];
            load8<T>(xp + i0, x1); load8<T>(xp + HALF + i0, x2);
            float o1[8], o2[8];
            if (a.cos_sin) {
                const float* cs = a.cos_sin + (size_t)pos * D;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float c = round_to<T>(cs[i0 + e]), s = round_to<T>(cs[HALF + i0 + e]);
                    o1[e] = rope_term<T>(x1[e], c, -x2[e], s);
                    o2[e] = rope_term<T>(x2[e], c, x1[e], s);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { o1[e] = x1[e]; o2[e] = x2[e]; }
            }
            if (is_k) {
                T* kp = kc + (size_t)pos * D;
                store8<T>(kp + i0, o1); store8<T>(kp + HALF + i0, o2);
                if (a.k_inplace && a.cos_sin) { store8<T>(xp + i0, o1); store8<T>(xp + HALF + i0, o2); }
            } else if (a.cos_sin) {
                store8<T>(xp + i0, o1); store8<T>(xp + HALF + i0, o2);
            }
        }
        return;
    }

    // ---- V: [tok][D] -> Vᵀ[d][pos] --------------------------------------------------------------------------------
    const int kvh = slot - a.n_heads - a.n_kv_heads;
    constexpr int PITCH = D + 2;               // elements; breaks the power-of-two column stride
    T* tile = reinterpret_cast<T*>(smem);      // [64][PITCH]
    const int col0 = (a.n_heads + a.n_kv_heads) * D + kvh * D;
    for (int g = tid; g < 64 * (D / 8); g += 256) {
        const int t = g / (D / 8), c = (g % (D / 8)) * 8;
        if (t >= ntok) continue;
        float v[8]; load8<T>(qkv + (size_t)(tok0 + t) * a.qkv_stride + col0 + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[t * PITCH + c + e] = from_f32<T>(v[e]);
    }
    __syncthreads();
    T* vt = reinterpret_cast<T*>(a.VT) + (size_t)kvh * D * a.s_max;
    const int t = tid & 63;
    if (t < ntok) {
        const int pos = pos0 + tok0 + t;
        for (int d = tid >> 6; d < D; d += 4) vt[(size_t)d * a.s_max + pos] = tile[t * PITCH + d];
    }
}

void launch_rope_kv(int dtype, int D, const RopeKvArgs& a, hipStream_t st) {
    LMX_REQUIRE(D == 64 || D == 128, "rope_kv: head_dim must be 64 or 128");
    LMX_REQUIRE(a.qkv_stride % 8 == 0, "rope_kv: qkv stride must be a multiple of 8");
    if (a.T <= 0) return;
    const dim3 grid(cdiv(a.T, 64), a.n_heads + 2 * a.n_kv_heads);
    const size_t smem = (size_t)64 * (D + 2) * dtype_size(dtype);
#define L(TT, DD) hipLaunchKernelGGL((rope_kv_kernel<TT, DD>), grid, dim3(256), smem, st, a)
    if (dtype == kBF16) { if (D == 128) L(bf16_t, 128); else L(bf16_t, 64); }
    else if (dtype == kF16) { if (D == 128) L(f16_t, 128); else L(f16_t, 64); }
    else { if (D == 128) L(float, 128); else L(float, 64); }
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// embedding gather / splice
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_embed_kernel(const int* __restrict__ src, const T* __restrict__ table,
                                                           const T* __restrict__ feats, T* __restrict__ out, int H) {
    const int row = blockIdx.x;
    const int s = src[row];
    T* o = out + (size_t)row * H;
    constexpr int VE = 16 / sizeof(T);
    const int HC = H / VE;
    if (s == -1) {
        for (int c = threadIdx.x; c < HC; c += 256) reinterpret_cast<uint4*>(o)[c] = make_uint4(0, 0, 0, 0);
        return;
    }
    const T* p = s >= 0 ? table + (size_t)s * H : feats + (size_t)(-2 - s) * H;
    for (int c = threadIdx.x; c < HC; c += 256) reinterpret_cast<uint4*>(o)[c] = reinterpret_cast<const uint4*>(p)[c];
}

void launch_gather_embed(int dtype, const int* src, const void* table, const void* feats, void* out, int rows, int H, hipStream_t st) {
    LMX_REQUIRE(H % 8 == 0, "gather_embed: hidden size must be a multiple of 8");
    if (rows <= 0) return;
#define L(TT) hipLaunchKernelGGL(gather_embed_kernel<TT>, dim3(rows), dim3(256), 0, st, src, (const TT*)table, (const TT*)feats, (TT*)out, H)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(256) void gather_token_kernel(const int64_t* __restrict__ tok, const T* __restrict__ table,
                                                           T* __restrict__ out, int H, int vocab) {
    int64_t s = *tok;
    if (s < 0) s = 0;
    if (s >= vocab) s = vocab - 1;
    constexpr int VE = 16 / sizeof(T);
    const int HC = H / VE;
    const T* p = table + (size_t)s * H;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < HC; c += gridDim.x * 256)
        reinterpret_cast<uint4*>(out)[c] = reinterpret_cast<const uint4*>(p)[c];
}

void launch_gather_token(int dtype, const int64_t* tok_ptr, const void* table, void* out, int H, int vocab, hipStream_t st) {
#define L(TT) hipLaunchKernelGGL(gather_token_kernel<TT>, dim3(2), dim3(256), 0, st, tok_ptr, (const TT*)table, (TT*)out, H, vocab)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// CLIP embeddings
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ pix, T* __restrict__ out, int S, int ps, int kpad) {
    const int G = S / ps;                          // patches per side
    const int P = G * G;
    const int n = blockIdx.x / P, p = blockIdx.x % P;
    const int gy = p / G, gx = p % G;
    const int kk = 3 * ps * ps;
    const T zero = from_f32<T>(0.f);
    T* o = out + (size_t)blockIdx.x * kpad;
    for (int k = threadIdx.x; k < kpad; k += 256) {
        T v = zero;
        if (k < kk) {
            const int c = k / (ps * ps), r = k % (ps * ps);
            const int py = r / ps, px = r % ps;
            v = pix[(((size_t)n * 3 + c) * S + gy * ps + py) * S + gx * ps + px];
        }
        o[k] = v;
    }
}

void launch_im2col(int dtype, const void* pix, void* out, int N, int S, int ps, int kpad, hipStream_t st) {
    LMX_REQUIRE(S % ps == 0 && kpad >= 3 * ps * ps, "im2col: bad geometry");
    const int P = (S / ps) * (S / ps);
    if (N <= 0) return;
#define L(TT) hipLaunchKernelGGL(im2col_kernel<TT>, dim3(N * P), dim3(256), 0, st, (const TT*)pix, (TT*)out, S, ps, kpad)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// y[n][t] = LN( T(e + pos[t]) ), e = class embedding (t == 0) or patch GEMM output (t >= 1)
template <typename T>
__global__ __launch_bounds__(256) void clip_embed_ln_kernel(const T* __restrict__ patch, const T* __restrict__ cls, const T* __restrict__ pos,
                                                            const T* __restrict__ w, const T* __restrict__ b, T* __restrict__ y,
                                                            int P, int Dm, float eps) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* buf = reinterpret_cast<float*>(smem);         // [Dm]
    const int Tt = P + 1;
    const int n = blockIdx.x / Tt, t = blockIdx.x % Tt;
    const T* e = t == 0 ? cls : patch + ((size_t)n * P + (t - 1)) * Dm;
    const T* pp = pos + (size_t)t * Dm;
    float s = 0.f;
    for (int i = threadIdx.x; i < Dm; i += 256) {
        const float v = round_to<T>(to_f32(e[i]) + to_f32(pp[i]));
        buf[i] = v; s += v;
    }
    const float mean = block_sum<4>(s, red) / (float)Dm;
    float q = 0.f;
    for (int i = threadIdx.x; i < Dm; i += 256) { const float d = buf[i] - mean; q += d * d; }
    const float rstd = rsqrtf(block_sum<4>(q, red) / (float)Dm + eps);
    T* yr = y + (size_t)blockIdx.x * Dm;
    for (int i = threadIdx.x; i < Dm; i += 256) yr[i] = from_f32<T>((buf[i] - mean) * rstd * to_f32(w[i]) + to_f32(b[i]));
}

void launch_clip_embed_ln(int dtype, const void* patch, const void* cls, const void* pos, const void* w, const void* b,
                          void* y, int N, int P, int Dm, float eps, hipStream_t st) {
    if (N <= 0) return;
    const size_t smem = (size_t)Dm * 4;
#define L(TT) hipLaunchKernelGGL(clip_embed_ln_kernel<TT>, dim3(N * (P + 1)), dim3(256), smem, st, (const TT*)patch, (const TT*)cls, (const TT*)pos, (const TT*)w, (const TT*)b, (TT*)y, P, Dm, eps)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int Ttot, int tok0, int P, int Dm) {
    const int n = blockIdx.x / P, p = blockIdx.x % P;
    const T* s = src + ((size_t)n * Ttot + tok0 + p) * Dm;
    T* d = dst + (size_t)blockIdx.x * Dm;
    constexpr int VE = 16 / sizeof(T);
    for (int c = threadIdx.x; c < Dm / VE; c += 256) reinterpret_cast<uint4*>(d)[c] = reinterpret_cast<const uint4*>(s)[c];
}

void launch_copy_rows(int dtype, const void* src, void* dst, int N, int Ttot, int tok0, int P, int Dm, hipStream_t st) {
    LMX_REQUIRE(Dm % 8 == 0, "copy_rows: width must be a multiple of 8");
    if (N * P <= 0) return;
#define L(TT) hipLaunchKernelGGL(copy_rows_kernel<TT>, dim3(N * P), dim3(256), 0, st, (const TT*)src, (TT*)dst, Ttot, tok0, P, Dm)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// greedy sampling + decode-loop bookkeeping
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void argmax_kernel(const T* __restrict__ logits, int V, int64_t* __restrict__ out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY; int idx = 0x7fffffff;
    const int V8 = V & ~7;
    for (int i = threadIdx.x * 8; i < V8; i += 1024 * 8) {          // 16-byte loads; ascending ids inside a thread keep "first wins"
        float v[8]; load8<T>(logits + i, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (v[j] > best) { best = v[j]; idx = i + j; }
    }
    for (int i = V8 + threadIdx.x; i < V; i += 1024) {
        const float v = to_f32(logits[i]);
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i)
            if (bv[i] > best || (bv[i] == best && bi[i] < idx)) { best = bv[i]; idx = bi[i]; }
        *out = idx == 0x7fffffff ? 0 : idx;
    }
}

void launch_argmax(int dtype, const void* logits, int V, int64_t* out_tok, hipStream_t st) {
#define L(TT) hipLaunchKernelGGL(argmax_kernel<TT>, dim3(1), dim3(1024), 0, st, (const TT*)logits, V, out_tok)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

__global__ void advance_kernel(int* len_ptr, const int64_t* tok_ptr, int64_t* out_tokens, int* n_out_ptr, int max_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *len_ptr += 1;
        const int n = *n_out_ptr;
        if (out_tokens && n < max_out) out_tokens[n] = *tok_ptr;
        *n_out_ptr = n + 1;
    }
}

void launch_advance(int* len_ptr, const int64_t* tok_ptr, int64_t* out_tokens, int* n_out_ptr, int max_out, hipStream_t st) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, len_ptr, tok_ptr, out_tokens, n_out_ptr, max_out);
    LMX_CHECK_HIP(hipGetLastError());
}

// ---- decode batch: the same two steps for every member of a batch in one launch each --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_tokens_batch_kernel(const SeqStateRef* __restrict__ tab, const T* __restrict__ table,
                                                                  T* __restrict__ out, int H, int vocab) {
    const int i = blockIdx.y;
    int64_t s = *tab[i].tok;
    if (s < 0) s = 0;
    if (s >= vocab) s = vocab - 1;
    constexpr int VE = 16 / sizeof(T);
    const int HC = H / VE;
    const T* p = table + (size_t)s * H;
    T* o = out + (size_t)i * H;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < HC; c += gridDim.x * 256)
        reinterpret_cast<uint4*>(o)[c] = reinterpret_cast<const uint4*>(p)[c];
}

void launch_gather_tokens_batch(int dtype, const SeqStateRef* tab, int n, const void* table, void* out, int H, int vocab, hipStream_t st) {
#define L(TT) hipLaunchKernelGGL(gather_tokens_batch_kernel<TT>, dim3(2, n), dim3(256), 0, st, tab, (const TT*)table, (TT*)out, H, vocab)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// (the per-member pick + advance kernel of the decode batch lives in sampling.hip: greedy or sampled per member)

// ---------------------------------------------------------------------------------------------------------------
// weight re-layout
// ---------------------------------------------------------------------------------------------------------------
// fused gate|up weight: source row j of gate (half=0) / up (half=1) goes to fused row 64*(j/32) + (j%32) + 32*half, so a
// wave's two adjacent 32-row MFMA tiles hold gate[j] and up[j] in the same lane/register (SiLU·mul in the GEMM epilogue).
template <typename T>
__global__ __launch_bounds__(256) void interleave_half_kernel(const T* __restrict__ src, T* __restrict__ dst, int K, int half) {
    const int j = blockIdx.x;
    const T* s = src + (size_t)j * K;
    T* d = dst + (size_t)(64 * (j >> 5) + (j & 31) + 32 * half) * K;
    constexpr int VE = 16 / sizeof(T);
    for (int c = threadIdx.x; c < K / VE; c += 256) reinterpret_cast<uint4*>(d)[c] = reinterpret_cast<const uint4*>(s)[c];
}

void launch_interleave_half(int dtype, const void* src, void* dst, int I, int K, int half, hipStream_t st) {
    LMX_REQUIRE(I % 32 == 0 && K % 8 == 0, "interleave: intermediate size must be a multiple of 32");
#define L(TT) hipLaunchKernelGGL(interleave_half_kernel<TT>, dim3(I), dim3(256), 0, st, (const TT*)src, (TT*)dst, K, half)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

__global__ void set_state_kernel(int* len_ptr, int len, int64_t* tok_ptr, int64_t tok, int set_tok, int* nout_ptr, int set_nout) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (len >= 0) *len_ptr = len;
        if (set_tok) *tok_ptr = tok;
        if (set_nout >= 0) *nout_ptr = set_nout;
    }
}
void launch_set_state(int* len_ptr, int len, int64_t* tok_ptr, int64_t tok, int set_tok, int* nout_ptr, int set_nout, hipStream_t st) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(64), 0, st, len_ptr, len, tok_ptr, tok, set_tok, nout_ptr, set_nout);
    LMX_CHECK_HIP(hipGetLastError());
}


// 128-bit content hash of a device buffer (image-feature cache key: the LLaVA-Plus tool loop re-sends the same image with every re-prompt,
// llava/serve/gradio_web_server_llava_plus.py:612-637).  Two independent 64-bit sums of mixed 16-byte chunks: each chunk is mixed with its index, the sums are
// commutative (atomic adds in any order give the same value), so the hash is a pure function of the bytes.  n16 = number of 16-byte chunks; `tail` bytes (< 16)
// after them are hashed by thread 0 of block 0.  out[2 * blockIdx.y + {0, 1}], zeroed by the caller; blockIdx.y = image.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__global__ __launch_bounds__(256) void hash128_kernel(const uint8_t* __restrict__ base, size_t bytes_per_item, unsigned long long* __restrict__ out) {
    const uint8_t* p = base + (size_t)blockIdx.y * bytes_per_item;
    const size_t n16 = bytes_per_item / 16, tail = bytes_per_item - n16 * 16;
    uint64_t h1 = 0, h2 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + i * 16);
        const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
        h1 += mix64(lo ^ mix64(hi + 0x9e3779b97f4a7c15ull * (i + 1)));
        h2 += mix64(hi ^ mix64(lo + 0xc2b2ae3d27d4eb4full * (i + 1)) ^ 0x165667b19e3779f9ull);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t t = bytes_per_item;
        for (size_t k = 0; k < tail; ++k) t = mix64(t ^ ((uint64_t)p[n16 * 16 + k] << (8 * (k & 7))) ^ (k + 1));
        h1 += mix64(t); h2 += mix64(t ^ 0x27d4eb2f165667c5ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 2 * blockIdx.y, (unsigned long long)h1);
        atomicAdd(out + 2 * blockIdx.y + 1, (unsigned long long)h2);
    }
}
void launch_hash128(const void* base, size_t bytes_per_item, int items, uint64_t* out_dev, hipStream_t st) {
    LMX_REQUIRE(items >= 1 && bytes_per_item > 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (items == 1 || bytes_per_item % 16 == 0),
                "hash128: every item must start 16-byte aligned");
    LMX_CHECK_HIP(hipMemsetAsync(out_dev, 0, (size_t)items * 16, st));
    const size_t n16 = bytes_per_item / 16;
    int gx = (int)((n16 + 255) / 256); gx = gx < 1 ? 1 : (gx > 128 ? 128 : gx);
    hipLaunchKernelGGL(hash128_kernel, dim3(gx, items), dim3(256), 0, st, static_cast<const uint8_t*>(base), bytes_per_item, reinterpret_cast<unsigned long long*>(out_dev));
    LMX_CHECK_HIP(hipGetLastError());
}

// the stop rule travels by value (kernel argument): no host buffer to keep alive, no host synchronisation in lmx_seq_set_stop
__global__ void set_stop_kernel(StopSpec* dst, StopSpec v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
void launch_set_stop(StopSpec* dst, const StopSpec& v, hipStream_t st) {
    hipLaunchKernelGGL(set_stop_kernel, dim3(1), dim3(64), 0, st, dst, v);
    LMX_CHECK_HIP(hipGetLastError());
}

// the pick of a prefill (launch_argmax / launch_sample left it in *tok_ptr) joins the sequence's token log; its stop rule (may be null) is applied to it
__global__ void log_token_kernel(const int64_t* tok_ptr, int64_t* log, int* n_out_ptr, int max_out, StopSpec* stop) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (stop && stop->done != 0) return;
        const int n = *n_out_ptr;
        const int64_t t = *tok_ptr;
        if (n < max_out) log[n] = t;
        *n_out_ptr = n + 1;
        if (stop && stop_rule_fires(stop, t, log, n + 1, max_out)) stop->done = 1;
    }
}
void launch_log_token(const int64_t* tok_ptr, int64_t* log, int* n_out_ptr, int max_out, StopSpec* stop, hipStream_t st) {
    hipLaunchKernelGGL(log_token_kernel, dim3(1), dim3(64), 0, st, tok_ptr, log, n_out_ptr, max_out, stop);
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename S, typename Dt>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, Dt* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = from_f32<Dt>(to_f32<S>(s[i]));
}

void launch_cast(int sd, int dd, const void* src, void* dst, size_t n, hipStream_t st) {
    if (n == 0) return;
    const int grid = (int)(cdiv64((int64_t)n, 256) < 4096 ? cdiv64((int64_t)n, 256) : 4096);
#define L(SS, DD) hipLaunchKernelGGL((cast_kernel<SS, DD>), dim3(grid), dim3(256), 0, st, (const SS*)src, (DD*)dst, n)
    if (sd == kF32 && dd == kBF16) L(float, bf16_t);
    else if (sd == kF32 && dd == kF16) L(float, f16_t);
    else if (sd == kBF16 && dd == kF32) L(bf16_t, float);
    else if (sd == kF16 && dd == kF32) L(f16_t, float);
    else if (sd == kF32 && dd == kF32) L(float, float);
    else if (sd == kBF16 && dd == kBF16) L(bf16_t, bf16_t);
    else if (sd == kF16 && dd == kF16) L(f16_t, f16_t);
    else if (sd == kBF16 && dd == kF16) L(bf16_t, f16_t);
    else if (sd == kF16 && dd == kBF16) L(f16_t, bf16_t);
    else throw Error{"cast: unsupported dtype pair"};
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
