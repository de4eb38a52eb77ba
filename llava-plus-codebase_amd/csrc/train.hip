// Training-step slices for the visual-instruction-tuning path (BASELINE config 5) — parity-first versions.
//
// What the reference runs in a finetuning step (llava/train/train.py:805-1000 -> HF Trainer -> LlavaLlamaForCausalLM.forward with labels,
// llava_llama.py:56-99 -> LlamaForCausalLM's shifted cross-entropy; attention through llava/train/llama_flash_attn_monkey_patch.py:68-91)
// and what each kernel here replaces:
//   ce_loss_fwd / ce_loss_bwd     CrossEntropyLoss over logits[..., :-1, :] vs labels[..., 1:], ignore_index = IGNORE_INDEX (-100,
//                                 llava/constants.py:7; labels built by llava_arch.py:181,200), mean over the counted positions, fp32 math
//   rmsnorm_bwd_dx / _dw          autograd of LlamaRMSNorm (HF5:models/llama/modeling_llama.py:53-67)
//   swiglu_bwd                    autograd of down_proj's input  silu(gate) * up  (LlamaMLP :163-176)
//   rope_bwd                      autograd of apply_rotary_pos_emb (:138-160): dx = dy * cos + rot^T(dy * sin)
//   transpose2d                   operand re-layout so that dgrad (dX = dY · W) and wgrad (dW = dYᵀ · X) run on the forward GEMM kernels,
//                                 which contract over the contiguous dimension of both operands
//   attn_bwd                      causal softmax attention backward with recomputation (the contract of flash_attn_unpadded_qkvpacked_func
//                                 used at llama_flash_attn_monkey_patch.py:68-91: dropout 0, scale 1/sqrt(d), causal): dQ, dK, dV
//   elementwise / col_sum / embed_bwd / sumsq / adamw   the remaining pieces of one optimisation step (bottom of the file); ZeRO-2's partitioning
//                                 and collectives live in llava_mi355x/train.py (torch.distributed: RCCL on the GPUs)
// These are correctness-first kernels (fp32 accumulation, simple tiling).
#include <algorithm>
#include <cmath>

#include <mutex>
#include "common.h"
#include "kernels.h"
#include "engine.h"

namespace lmx {

// ---------------------------------------------------------------------------------------------------------------
// shifted cross-entropy.  Row r = (b, t), t < T-1: logits row b*T + t against label[b*T + t + 1].
//   fwd: lse[r] = logsumexp(logits row) ; row_loss[r] = lse - logit[label] (0 for ignored rows)
//   reduce: out[0] = mean over counted rows, out[1] = number of counted rows
//   bwd: dlogits[b*T + t] = (softmax - onehot) * grad / n_counted for counted rows, 0 otherwise (row T-1 of every b: 0)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, int ld, const int64_t* __restrict__ labels, int Tlen, int V,
                                                     int64_t ignore, float* __restrict__ lse, float* __restrict__ row_loss) {
    __shared__ float red[4];
    const int r = blockIdx.x, b = r / (Tlen - 1), t = r % (Tlen - 1);
    const T* row = logits + (size_t)(b * Tlen + t) * ld;
    const int64_t lab = labels[(size_t)b * Tlen + t + 1];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) mx = fmaxf(mx, to_f32(row[i]));
    mx = block_max<4>(mx, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(to_f32(row[i]) - mx);
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const float l = logf(s) + mx;
        lse[r] = l;
        row_loss[r] = (lab == ignore || lab < 0 || lab >= V) ? 0.f : l - to_f32(row[lab]);
    }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, int B, int Tlen, int V,
                                                        int64_t ignore, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f, n = 0.f;
    const int R = B * (Tlen - 1);
    for (int r = threadIdx.x; r < R; r += 256) {
        const int b = r / (Tlen - 1), t = r % (Tlen - 1);
        const int64_t lab = labels[(size_t)b * Tlen + t + 1];
        if (!(lab == ignore || lab < 0 || lab >= V)) { s += row_loss[r]; n += 1.f; }
    }
    s = block_sum<4>(s, red);
    n = block_sum<4>(n, red);
    if (threadIdx.x == 0) { out[0] = s / n; out[1] = n; }      // 0 / 0 = NaN, as torch's mean over no elements
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, int ld, const int64_t* __restrict__ labels, int Tlen, int V,
                                                     int64_t ignore, const float* __restrict__ lse, const float* __restrict__ out, float grad,
                                                     T* __restrict__ dlogits, int ldd) {
    const int row_id = blockIdx.x, b = row_id / Tlen, t = row_id % Tlen;      // one workgroup per logits row (incl. the last position)
    T* d = dlogits + (size_t)row_id * ldd;
    int64_t lab = ignore;
    if (t < Tlen - 1) lab = labels[(size_t)b * Tlen + t + 1];
    if (lab == ignore || lab < 0 || lab >= V) {
        for (int i = threadIdx.x; i < V; i += 256) d[i] = from_f32<T>(0.f);
        return;
    }
    const T* row = logits + (size_t)row_id * ld;
    const float l = lse[b * (Tlen - 1) + t], g = grad / out[1];
    for (int i = threadIdx.x; i < V; i += 256) {
        const float p = expf(to_f32(row[i]) - l);
        d[i] = from_f32<T>((p - (i == lab ? 1.f : 0.f)) * g);
    }
}

void launch_ce_loss_fwd(int dtype, const void* logits, int ld, const int64_t* labels, int B, int Tlen, int V, int64_t ignore, float* lse,
                        float* row_loss, float* out2, hipStream_t st) {
    LMX_REQUIRE(B >= 1 && Tlen >= 2 && V >= 1, "ce_loss: need at least two positions per sequence");
    const int R = B * (Tlen - 1);
#define L(TT) hipLaunchKernelGGL(ce_fwd_kernel<TT>, dim3(R), dim3(256), 0, st, (const TT*)logits, ld, labels, Tlen, V, ignore, lse, row_loss)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, labels, B, Tlen, V, ignore, out2);
    LMX_CHECK_HIP(hipGetLastError());
}

void launch_ce_loss_bwd(int dtype, const void* logits, int ld, const int64_t* labels, int B, int Tlen, int V, int64_t ignore, const float* lse,
                        const float* out2, float grad, void* dlogits, int ldd, hipStream_t st) {
#define L(TT) hipLaunchKernelGGL(ce_bwd_kernel<TT>, dim3(B * Tlen), dim3(256), 0, st, (const TT*)logits, ld, labels, Tlen, V, ignore, lse, out2, grad, (TT*)dlogits, ldd)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// RMSNorm backward.  y = w * xhat, xhat = x * inv, inv = rsqrt(mean(x^2) + eps)   (the forward's rounding of xhat is the identity for autograd)
//   dx = inv * (g - xhat * mean(g * xhat)),  g = dy * w           dw = sum_rows dy * xhat
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ dy, T* __restrict__ dx,
                                                             int H, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const T* xr = x + (size_t)row * H; const T* gr = dy + (size_t)row * H; T* dr = dx + (size_t)row * H;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) { const float v = to_f32(xr[i]); ss += v * v; }
    ss = block_sum<4>(ss, red);
    const float inv = rsqrtf(ss / (float)H + eps);
    float dot = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) dot += to_f32(gr[i]) * to_f32(w[i]) * to_f32(xr[i]) * inv;
    dot = block_sum<4>(dot, red) / (float)H;
    for (int i = threadIdx.x; i < H; i += 256) {
        const float xh = to_f32(xr[i]) * inv;
        dr[i] = from_f32<T>(inv * (to_f32(gr[i]) * to_f32(w[i]) - xh * dot));
    }
}

// dw[c] = sum_r dy[r][c] * x[r][c] * inv[r]: one workgroup per 64 columns, 4 row lanes x 64 columns, inv recomputed per row by a pre-pass
template <typename T>
__global__ __launch_bounds__(256) void rms_inv_kernel(const T* __restrict__ x, float* __restrict__ inv, int H, float eps) {
    __shared__ float red[4];
    const T* xr = x + (size_t)blockIdx.x * H;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) { const float v = to_f32(xr[i]); ss += v * v; }
    ss = block_sum<4>(ss, red);
    if (threadIdx.x == 0) inv[blockIdx.x] = rsqrtf(ss / (float)H + eps);
}
// two deterministic stages: (64 columns x a block of `rpb` rows) per workgroup -> fp32 partial rows, then a column sum over the row blocks in block order.
// (Round 2 walked ALL rows with 64 workgroups: 662 us per call at 8192 rows for 134 MB of input — 6 % of a training step.)
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_dw_part_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ inv, float* __restrict__ part,
                                                                  int rows, int H, int rpb) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
    float s = 0.f;
    if (c < H)
        for (int r = r0 + rl; r < r1; r += 4) s += to_f32(dy[(size_t)r * H + c]) * to_f32(x[(size_t)r * H + c]) * inv[r];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < H) part[(size_t)blockIdx.y * H + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void rmsnorm_bwd_dw_sum_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ dw) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * H + c];
    dw[c] = s;
}

// One pass over x and dy for 16-bit rows of H = NCH * 512 (round 6; the three launches above read x three times and dy twice: 774 us per norm at 32768 x 4096,
// 2.8 % of a 16 x 2048 training step, profiles/r06_config5_kernel_stats_wgrad.csv).  One wave per row, 16 rows per wave in turn, 8 waves = 128 rows per workgroup:
// a lane holds its 8 NCH elements of x, dy (and the residual) packed, computes inv, mean(g * xhat), dx, and keeps the row's dy * x * inv in NCH * 8 fp32
// accumulators; after its rows the eight waves add their accumulators in wave order through LDS and the workgroup writes ONE partial row of dw (the same
// [ceil(rows / 128)][H] partials the two-stage form used; rmsnorm_bwd_dw_sum_kernel adds them in block order: deterministic).
// `res` (optional): dx = T(res + T(dx)) — the residual branch's gradient joins here instead of in an elementwise launch of its own (two roundings, as that launch).
template <typename T> __device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
    v[0] = unpack_lo<T>(u.x); v[1] = unpack_hi<T>(u.x); v[2] = unpack_lo<T>(u.y); v[3] = unpack_hi<T>(u.y);
    v[4] = unpack_lo<T>(u.z); v[5] = unpack_hi<T>(u.z); v[6] = unpack_lo<T>(u.w); v[7] = unpack_hi<T>(u.w);
}
template <typename T, int NCH>
__global__ __launch_bounds__(512) void rmsnorm_bwd_fused_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ dy, const T* __restrict__ res,
                                                                T* __restrict__ dx, float* __restrict__ part, int rows, float eps) {
    constexpr int H = NCH * 512, RPB = 128, RPW = 16;
    __shared__ float sums[H];
    __shared__ __attribute__((aligned(16))) T wl[H];                       // the weight row, read back per chunk (registers go to the dw accumulators)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < H / 8; i += 512) reinterpret_cast<uint4*>(wl)[i] = reinterpret_cast<const uint4*>(w)[i];
    __syncthreads();
    float acc[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    const int r0 = blockIdx.x * RPB + wave * RPW;
#pragma unroll 1
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = r0 + rr;
        if (r >= rows) break;                                              // wave-uniform
        const size_t off = (size_t)r * H + lane * 8;
        uint4 xv[NCH], gv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) { xv[j] = *reinterpret_cast<const uint4*>(x + off + j * 512); gv[j] = *reinterpret_cast<const uint4*>(dy + off + j * 512); }
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float xf[8]; unpack8<T>(xv[j], xf);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += xf[e] * xf[e];
        }
        ss = wave_sum(ss);
        const float inv = rsqrtf(ss / (float)H + eps);
        // the packed words stay the only copy of the row between the passes: without these the compiler keeps the unpacked floats of one pass alive for the next
        // (128 more registers: spills)
#define LMX_OPAQUE(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
        for (int j = 0; j < NCH; ++j) LMX_OPAQUE(xv[j]);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float xf[8], gf[8], wf[8]; unpack8<T>(xv[j], xf); unpack8<T>(gv[j], gf);
            unpack8<T>(*reinterpret_cast<const uint4*>(wl + j * 512 + lane * 8), wf);
#pragma unroll
            for (int e = 0; e < 8; ++e) dot += gf[e] * wf[e] * xf[e] * inv;
        }
        dot = wave_sum(dot) / (float)H;
#pragma unroll
        for (int j = 0; j < NCH; ++j) { LMX_OPAQUE(xv[j]); LMX_OPAQUE(gv[j]); }
#undef LMX_OPAQUE
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float xf[8], gf[8], wf[8], o[8]; unpack8<T>(xv[j], xf); unpack8<T>(gv[j], gf);
            unpack8<T>(*reinterpret_cast<const uint4*>(wl + j * 512 + lane * 8), wf);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = xf[e] * inv;
                o[e] = inv * (gf[e] * wf[e] - xh * dot);
                acc[j][e] += gf[e] * xf[e] * inv;
            }
            if (res) {
                float rf[8]; unpack8<T>(*reinterpret_cast<const uint4*>(res + off + j * 512), rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rf[e] + round_to<T>(o[e]);
            }
            uint4 u;
            u.x = pack2<T>(o[0], o[1]); u.y = pack2<T>(o[2], o[3]); u.z = pack2<T>(o[4], o[5]); u.w = pack2<T>(o[6], o[7]);
            *reinterpret_cast<uint4*>(dx + off + j * 512) = u;
        }
    }
    if (!part) return;                                                     // uniform: dx only
    // the workgroup's partial row of dw: waves add their accumulators in wave order (deterministic)
#pragma unroll 1
    for (int wv_ = 0; wv_ < 8; ++wv_) {
        if (wave == wv_) {
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float* p = sums + j * 512 + lane * 8 + e;
                    *p = wv_ == 0 ? acc[j][e] : *p + acc[j][e];
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < H; c += 512) part[(size_t)blockIdx.x * H + c] = sums[c];
}

void launch_rmsnorm_bwd(int dtype, const void* x, const void* w, const void* dy, const void* residual, void* dx, float* dw, float* inv_scratch, int rows, int H,
                        float eps, hipStream_t st) {
    if (rows <= 0) return;
    // partial rows of the weight gradient live in the CALLER's scratch, behind the [rows] inverse norms: inv_scratch holds
    // rows + cdiv(rows, 128) * H floats (lmx_op_rmsnorm_bwd's contract) — nothing process-wide, so steps on different streams / devices cannot meet
    constexpr int RPB = 128;
    const int nblk = cdiv(rows, RPB);
    float* part = dw ? inv_scratch + (((size_t)rows + 63) / 64) * 64 : nullptr;
    const bool aligned = (((uintptr_t)x | (uintptr_t)w | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)residual) & 15) == 0;
    if (dtype != kF32 && aligned && (H == 1024 || H == 2048 || H == 4096)) {
#define F(TT, NC) hipLaunchKernelGGL((rmsnorm_bwd_fused_kernel<TT, NC>), dim3(nblk), dim3(512), 0, st, (const TT*)x, (const TT*)w, (const TT*)dy, (const TT*)residual, \
                                     (TT*)dx, part, rows, eps)
#define FT(TT) do { if (H == 1024) F(TT, 2); else if (H == 2048) F(TT, 4); else F(TT, 8); } while (0)
        if (dtype == kBF16) FT(bf16_t); else FT(f16_t);
#undef FT
#undef F
        if (dw) hipLaunchKernelGGL(rmsnorm_bwd_dw_sum_kernel, dim3(cdiv(H, 256)), dim3(256), 0, st, part, nblk, H, dw);
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
#define L(TT)                                                                                                                              \
    do {                                                                                                                                   \
        hipLaunchKernelGGL(rmsnorm_bwd_dx_kernel<TT>, dim3(rows), dim3(256), 0, st, (const TT*)x, (const TT*)w, (const TT*)dy, (TT*)dx, H, eps);   \
        if (dw) {                                                                                                                          \
            hipLaunchKernelGGL(rms_inv_kernel<TT>, dim3(rows), dim3(256), 0, st, (const TT*)x, inv_scratch, H, eps);                       \
            hipLaunchKernelGGL(rmsnorm_bwd_dw_part_kernel<TT>, dim3(cdiv(H, 64), nblk), dim3(256), 0, st, (const TT*)x, (const TT*)dy, inv_scratch, part, rows, H, RPB); \
            hipLaunchKernelGGL(rmsnorm_bwd_dw_sum_kernel, dim3(cdiv(H, 256)), dim3(256), 0, st, part, nblk, H, dw);                        \
        }                                                                                                                                  \
    } while (0)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
    if (residual) launch_elementwise(dtype, 3 /* add */, residual, dx, dx, (size_t)rows * H, st);      // the shapes the fused kernel does not take: dx = residual + dx
}

// ---------------------------------------------------------------------------------------------------------------
// SwiGLU backward: act = silu(g) * u ;  dg = dact * u * sigma(g) * (1 + g * (1 - sigma(g))) ;  du = dact * silu(g)
// gu = [rows][2I] with the engine's fused row order is NOT assumed here: plain [g | u] halves, leading dim ldgu.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ u, const T* __restrict__ dact, T* __restrict__ dg,
                                                         T* __restrict__ du, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gv = to_f32(g[i]), uv = to_f32(u[i]), d = to_f32(dact[i]);
    const float sg = 1.f / (1.f + expf(-gv));
    dg[i] = from_f32<T>(d * uv * sg * (1.f + gv * (1.f - sg)));
    du[i] = from_f32<T>(d * gv * sg);
}
// 16-bit, 8 elements per lane (16-byte loads / stores): the scalar form above moved 2 bytes per lane and ran at 3.7 TB/s over its five streams
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd8_kernel(const T* __restrict__ g, const T* __restrict__ u, const T* __restrict__ dact, T* __restrict__ dg,
                                                          T* __restrict__ du, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float gv[8], uv[8], dv[8], og[8], ou[8];
    load8<T>(g + i * 8, gv); load8<T>(u + i * 8, uv); load8<T>(dact + i * 8, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float sg = 1.f / (1.f + expf(-gv[e]));
        og[e] = dv[e] * uv[e] * sg * (1.f + gv[e] * (1.f - sg));
        ou[e] = dv[e] * gv[e] * sg;
    }
    store8<T>(dg + i * 8, og); store8<T>(du + i * 8, ou);
}
void launch_swiglu_bwd(int dtype, const void* g, const void* u, const void* dact, void* dg, void* du, size_t n, hipStream_t st) {
    if (!n) return;
    if (dtype != kF32 && ((((uintptr_t)g | (uintptr_t)u | (uintptr_t)dact | (uintptr_t)dg | (uintptr_t)du) & 15) == 0)) {
        const size_t n8 = n / 8, tail = n - n8 * 8;
#define V(TT) do { if (n8) hipLaunchKernelGGL(swiglu_bwd8_kernel<TT>, dim3((unsigned)cdiv64((int64_t)n8, 256)), dim3(256), 0, st, (const TT*)g, (const TT*)u, (const TT*)dact, \
                                              (TT*)dg, (TT*)du, n8);                                                                                                   \
                   if (tail) hipLaunchKernelGGL(swiglu_bwd_kernel<TT>, dim3(1), dim3(256), 0, st, (const TT*)g + n8 * 8, (const TT*)u + n8 * 8, (const TT*)dact + n8 * 8, \
                                                (TT*)dg + n8 * 8, (TT*)du + n8 * 8, tail); } while (0)
        if (dtype == kBF16) V(bf16_t); else V(f16_t);
#undef V
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
#define L(TT) hipLaunchKernelGGL(swiglu_bwd_kernel<TT>, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, st, (const TT*)g, (const TT*)u, (const TT*)dact, (TT*)dg, (TT*)du, n)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE backward on [T][heads][D] (row stride ld): dx = dy * cos + rot^T(dy * sin), rot(x) = (-x2, x1)  =>  rot^T(v) = (v2, -v1)
// cos_sin table: [pos][D]: cos half | sin half of the D/2 frequencies (as lmx_set_rope_table)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, const float* __restrict__ cs, int pos0, int heads, int D,
                                                       int ld, int Tn) {
    const int half = D >> 1;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // one thread per (t, head, i < half)
    const size_t total = (size_t)Tn * heads * half;
    if (idx >= total) return;
    const int i = (int)(idx % half), h = (int)((idx / half) % heads), t = (int)(idx / ((size_t)half * heads));
    const float c = cs[(size_t)(pos0 + t) * D + i], s = cs[(size_t)(pos0 + t) * D + half + i];
    const T* src = dy + (size_t)t * ld + (size_t)h * D; T* dst = dx + (size_t)t * ld + (size_t)h * D;
    // forward rounds cos / sin to T and every product to T (HF rounding points); the backward applies the same rounded factors
    const float cr = round_to<T>(c), sr = round_to<T>(s);
    const float d1 = to_f32(src[i]), d2 = to_f32(src[half + i]);
    dst[i] = from_f32<T>(d1 * cr + d2 * sr);
    dst[half + i] = from_f32<T>(d2 * cr - d1 * sr);
}
void launch_rope_bwd(int dtype, const void* dy, void* dx, const float* cos_sin, int pos0, int Tn, int heads, int D, int ld, hipStream_t st) {
    const size_t total = (size_t)Tn * heads * (D / 2);
    if (!total) return;
#define L(TT) hipLaunchKernelGGL(rope_bwd_kernel<TT>, dim3((unsigned)cdiv64((int64_t)total, 256)), dim3(256), 0, st, (const TT*)dy, (TT*)dx, cos_sin, pos0, heads, D, ld, Tn)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// 2-D transpose through LDS (64 x 64 tiles, padded): dst[c][r] = src[r][c]
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int lds_, int rows, int cols, T* __restrict__ dst, int ldd) {
    __shared__ T tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (r0 + r < rows && c0 + c < cols) tile[r][c] = src[(size_t)(r0 + r) * lds_ + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < rows && c0 + c < cols) dst[(size_t)(c0 + c) * ldd + r0 + r] = tile[r][c];
    }
}
// 16-bit elements, every dimension / leading dimension a multiple of 8 and 16-byte aligned bases (the training step's operands: rows padded to the contraction
// quantum): 16-byte loads of the source rows, 16-byte stores of the destination rows — the scalar form above moved 2 bytes per lane and ran at 1.5 TB/s, 7 % of a
// training step (profiles/r06_config5_kernel_stats_before.csv).  64 (rows) x 128 (cols) source tile per workgroup: 256-byte runs on the read side, 128-byte runs
// on the write side.  LDS image swizzled, not padded: 16-byte chunk q of row r sits at chunk q ^ ((r >> 3) & 7), so the 8 lanes of a wave that gather one column
// from 8 different row groups hit 32 different banks; with a padded pitch (any multiple of 8 elements: 8 rows = a multiple of 128 bytes) they shared 4 banks, and
// that, not HBM, set the rate: 3.7 - 4.1 TB/s (read + write) for the 128 x 64 padded tile of the round's first form, 4.5 - 5.0 for this one on the 32768-row
// operands, 3.5 -> 6.2 - 6.5 on weight-sized ones (tools/probes/transpose_bw.hip, profiles/r06_transpose_forms.jsonl).
__global__ __launch_bounds__(256) void transpose16_vec_kernel(const uint16_t* __restrict__ src, int lds_, int rows, int cols, uint16_t* __restrict__ dst, int ldd) {
    constexpr int TR = 64, TC = 128, CPR = TC / 8, SPR = TR / 8;
    __shared__ __attribute__((aligned(16))) uint16_t tile[TR * TC];
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC, tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < TR * TC / 8 / 256; ++it) {                       // 4 passes: 16 rows x 16 chunks of 8 columns
        const int i = tid + it * 256, r = i / CPR, ch = i % CPR;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + r) * lds_ + c0 + ch * 8);
        *reinterpret_cast<uint4*>(tile + r * TC + ((ch ^ ((r >> 3) & 7)) * 8)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < TR * TC / 8 / 256; ++it) {                       // output row = source column c, 8 segments of 8 source rows
        const int i = tid + it * 256, c = i / SPR, sg = i % SPR;
        if (c0 + c >= cols || r0 + sg * 8 >= rows) continue;
        const int col = (((c >> 3) ^ sg) << 3) + (c & 7);
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = tile[(sg * 8 + j) * TC + col];
        uint4 v;
        v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16); v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
        *reinterpret_cast<uint4*>(dst + (size_t)(c0 + c) * ldd + r0 + sg * 8) = v;
    }
}
void launch_transpose(int dtype, const void* src, int ld, int rows, int cols, void* dst, int ldd, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return;
    const bool vec = dtype != kF32 && rows % 8 == 0 && cols % 8 == 0 && ld % 8 == 0 && ldd % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
    if (vec) {
        hipLaunchKernelGGL(transpose16_vec_kernel, dim3(cdiv(cols, 128), cdiv(rows, 64)), dim3(256), 0, st, (const uint16_t*)src, ld, rows, cols, (uint16_t*)dst, ldd);
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 64));
#define L(TT) hipLaunchKernelGGL(transpose_kernel<TT>, grid, dim3(256), 0, st, (const TT*)src, ld, rows, cols, (TT*)dst, ldd)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// causal attention backward with recomputation, two passes, no atomics (deterministic).  q / k / v / do: [T][heads][D] (row strides ld*),
// k / v of kv head h / (heads / kv_heads).
//   p_ij = exp(scale q_i . k_j - lse_i), j <= i ;  dp_ij = do_i . v_j ;  delta_i = sum_j p_ij dp_ij ;  ds_ij = p_ij (dp_ij - delta_i)
//   dq_i = scale * sum_j ds_ij k_j ;  dk_j = scale * sum_i ds_ij q_i ;  dv_j = sum_i p_ij do_i       (sums over the heads of a GQA group too)
// pass 1 (one workgroup per (block of QB query rows, head)): score rows in LDS -> softmax statistics lse_i, delta_i (kept for pass 2) and dq_i
// pass 2 (one workgroup per (block of KB keys, kv head)): walks the query rows i >= j in chunks of 256 — phase a: one thread per row
//   recomputes p and ds for the KB keys; phase b: one thread per output element accumulates dv / dk over the chunk (coalesced rows)
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int D, int QB>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dO,
                                                          T* __restrict__ dq, float* __restrict__ lse, float* __restrict__ delta_out, int Tn, int heads,
                                                          int kv_heads, int ldq, int ldk, int ldo, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* p = reinterpret_cast<float*>(smem_raw);          // [QB][Tn] scores -> ds
    float* dp = p + (size_t)QB * Tn;                        // [QB][Tn]
    __shared__ float qs[QB][D], dos[QB][D], red[4], part[QB][256];
    const int i0 = blockIdx.x * QB, h = blockIdx.y, hk = h / (heads / kv_heads), tid = threadIdx.x;
    for (int idx = tid; idx < QB * D; idx += 256) {
        const int c = idx / D, d = idx % D, i = i0 + c;
        qs[c][d] = i < Tn ? to_f32(q[(size_t)i * ldq + (size_t)h * D + d]) : 0.f;
        dos[c][d] = i < Tn ? to_f32(dO[(size_t)i * ldo + (size_t)h * D + d]) : 0.f;
    }
    __syncthreads();
    const int imax = (i0 + QB - 1 < Tn ? i0 + QB - 1 : Tn - 1);
    const int n = imax + 1;                                 // keys 0..imax cover every row of the block (row c uses keys 0..i0+c)
    // sweep over the keys: one thread per key, the block's QB query rows against it
    for (int j = tid; j < n; j += 256) {
        const T* kj = k + (size_t)j * ldk + (size_t)hk * D; const T* vj = v + (size_t)j * ldk + (size_t)hk * D;
        float s[QB], dd[QB];
#pragma unroll
        for (int c = 0; c < QB; ++c) { s[c] = 0.f; dd[c] = 0.f; }
        for (int d0 = 0; d0 < D; d0 += 8) {
            float kk[8], vv[8];
            load8<T>(kj + d0, kk); load8<T>(vj + d0, vv);
#pragma unroll
            for (int c = 0; c < QB; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[c] = fmaf(qs[c][d0 + e], kk[e], s[c]); dd[c] = fmaf(dos[c][d0 + e], vv[e], dd[c]); }
        }
#pragma unroll
        for (int c = 0; c < QB; ++c) { p[(size_t)c * Tn + j] = j <= i0 + c ? s[c] * scale : -INFINITY; dp[(size_t)c * Tn + j] = dd[c]; }
    }
    __syncthreads();
    // per row: log-sum-exp, delta, ds (in place)
#pragma unroll
    for (int c = 0; c < QB; ++c) {
        const int i = i0 + c;
        if (i >= Tn) break;                                  // uniform
        float* pr = p + (size_t)c * Tn; const float* dr = dp + (size_t)c * Tn;
        float mx = -INFINITY;
        for (int j = tid; j <= i; j += 256) mx = fmaxf(mx, pr[j]);
        mx = block_max<4>(mx, red);
        float sum = 0.f;
        for (int j = tid; j <= i; j += 256) sum += expf(pr[j] - mx);
        sum = block_sum<4>(sum, red);
        const float l = mx + logf(sum);
        float delta = 0.f;
        for (int j = tid; j <= i; j += 256) { const float pj = expf(pr[j] - l); pr[j] = pj; delta += pj * dr[j]; }
        delta = block_sum<4>(delta, red);
        if (tid == 0) { lse[(size_t)i * heads + h] = l; delta_out[(size_t)i * heads + h] = delta; }
        for (int j = tid; j < n; j += 256) pr[j] = j <= i ? pr[j] * (dr[j] - delta) : 0.f;       // ds; keys past the row's diagonal contribute 0
    }
    __syncthreads();
    // dq_i[d] = scale * sum_j ds_ij k_j[d]: 256 / D thread groups stride over j (each k element loaded once for the QB rows)
    constexpr int NG = 256 / D;
    const int g = tid / D, d = tid % D;
    float a[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) a[c] = 0.f;
    const T* kcol = k + (size_t)hk * D + d;
#pragma unroll 4
    for (int j = g; j < n; j += NG) {
        const float kv = to_f32(kcol[(size_t)j * ldk]);
#pragma unroll
        for (int c = 0; c < QB; ++c) a[c] = fmaf(p[(size_t)c * Tn + j], kv, a[c]);
    }
#pragma unroll
    for (int c = 0; c < QB; ++c) part[c][tid] = a[c];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < QB; ++c) {
            const int i = i0 + c;
            if (i >= Tn) continue;
            float r = a[c];
#pragma unroll
            for (int x = 1; x < NG; ++x) r += part[c][x * D + d];
            dq[(size_t)i * ldq + (size_t)h * D + d] = from_f32<T>(r * scale);
        }
    }
}

template <typename T, int D, int KB>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
                                                           int Tn, int heads, int kv_heads, int ldq, int ldk, int ldo, float scale) {
    __shared__ float ks[KB][D], vs[KB][D];
    __shared__ float ps[KB][256], dss[KB][256];
    constexpr int NG = 256 / D, NS = NG / 2;                // thread groups of D: role dv / dk, and NS interleaved slices of the rows
    __shared__ float comb[NG][KB][D];
    const int j0 = blockIdx.x * KB, hk = blockIdx.y, tid = threadIdx.x, group = heads / kv_heads;
    for (int idx = tid; idx < KB * D; idx += 256) {
        const int c = idx / D, d = idx % D, j = j0 + c;
        ks[c][d] = j < Tn ? to_f32(k[(size_t)j * ldk + (size_t)hk * D + d]) : 0.f;
        vs[c][d] = j < Tn ? to_f32(v[(size_t)j * ldk + (size_t)hk * D + d]) : 0.f;
    }
    __syncthreads();
    const int g = tid / D, d = tid % D, role = g & 1, slice = g >> 1;
    float acc[KB];
#pragma unroll
    for (int c = 0; c < KB; ++c) acc[c] = 0.f;
    for (int h = hk * group; h < (hk + 1) * group; ++h) {
        for (int i0 = j0; i0 < Tn; i0 += 256) {
            const int i = i0 + tid;
            float s[KB], dd[KB];
#pragma unroll
            for (int c = 0; c < KB; ++c) { s[c] = 0.f; dd[c] = 0.f; }
            if (i < Tn) {
                const T* qi = q + (size_t)i * ldq + (size_t)h * D; const T* doi = dO + (size_t)i * ldo + (size_t)h * D;
                for (int d0 = 0; d0 < D; d0 += 8) {
                    float qq[8], oo[8];
                    load8<T>(qi + d0, qq); load8<T>(doi + d0, oo);
#pragma unroll
                    for (int c = 0; c < KB; ++c)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s[c] = fmaf(qq[e], ks[c][d0 + e], s[c]); dd[c] = fmaf(oo[e], vs[c][d0 + e], dd[c]); }
                }
                const float l = lse[(size_t)i * heads + h], de = delta[(size_t)i * heads + h];
#pragma unroll
                for (int c = 0; c < KB; ++c) {
                    const int j = j0 + c;
                    const float pj = (j <= i && j < Tn) ? expf(s[c] * scale - l) : 0.f;
                    ps[c][tid] = pj; dss[c][tid] = pj * (dd[c] - de) * scale;
                }
            } else {
#pragma unroll
                for (int c = 0; c < KB; ++c) { ps[c][tid] = 0.f; dss[c][tid] = 0.f; }
            }
            __syncthreads();
            const int n = Tn - i0 < 256 ? Tn - i0 : 256;
            const T* col = role == 0 ? dO + (size_t)h * D + d : q + (size_t)h * D + d;
            const int ldc = role == 0 ? ldo : ldq;
            for (int ii = slice; ii < n; ii += NS) {
                const float x = to_f32(col[(size_t)(i0 + ii) * ldc]);
#pragma unroll
                for (int c = 0; c < KB; ++c) acc[c] = fmaf(role == 0 ? ps[c][ii] : dss[c][ii], x, acc[c]);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int c = 0; c < KB; ++c) comb[g][c][d] = acc[c];
    __syncthreads();
    if (slice == 0) {
#pragma unroll
        for (int c = 0; c < KB; ++c) {
            const int j = j0 + c;
            if (j >= Tn) continue;
            float a = acc[c];
#pragma unroll
            for (int x = 1; x < NS; ++x) a += comb[2 * x + role][c][d];
            T* out = role == 0 ? dv : dk;
            out[(size_t)j * ldk + (size_t)hk * D + d] = from_f32<T>(a);                 // dk / dv rows have the stride of k / v (column windows of a q|k|v buffer included)
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = from_f32<T>(src[i]);
}

void launch_attn_bwd(int dtype, int D, const void* q, const void* k, const void* v, const void* dO, void* dq, float* dk32, float* dv32, void* dk, void* dv,
                     int Tn, int heads, int kv_heads, int ldq, int ldk, int ldo, float scale, hipStream_t st) {
    LMX_REQUIRE(D == 64 || D == 128, "attn_bwd: head_dim must be 64 or 128");
    LMX_REQUIRE(heads % kv_heads == 0 && Tn >= 1, "attn_bwd: bad geometry");
    if (attn_bwd_mfma_wanted(dtype, D) && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0) {      // 16-bit models: the matrix-core kernels of attn_bwd.hip
        launch_attn_bwd_mfma(dtype, D, q, k, v, dO, dq, dk, dv, Tn, heads, kv_heads, ldq, ldk, ldo, scale, st);
        return;
    }
    LMX_REQUIRE(kv_heads * D >= heads, "attn_bwd: the scratch arrays ([T][kv_heads][D] floats) hold the per-(row, head) statistics");
    float* lse = dk32; float* delta = dv32;                 // per (query row, head): log-sum-exp and sum_j p dp
    const int QB = (size_t)2 * 4 * Tn * sizeof(float) <= 100 * 1024 ? 4 : 1;          // query rows per workgroup of pass 1 (score rows live in LDS)
    const size_t smem = (size_t)2 * QB * Tn * sizeof(float);
    LMX_REQUIRE(smem <= 120 * 1024, "attn_bwd: sequence too long for the parity kernel (<= 15360 positions)");
    constexpr int KB = 4;
#define L2(TT, DD)                                                                                                                                  \
    do {                                                                                                                                           \
        auto kern = QB == 4 ? attn_bwd_dq_kernel<TT, DD, 4> : attn_bwd_dq_kernel<TT, DD, 1>;                                                       \
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));          \
        hipLaunchKernelGGL(kern, dim3(cdiv(Tn, QB), heads), dim3(256), smem, st, (const TT*)q, (const TT*)k, (const TT*)v, (const TT*)dO, (TT*)dq, lse, delta, Tn, \
                           heads, kv_heads, ldq, ldk, ldo, scale);                                                                                 \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<TT, DD, KB>), dim3(cdiv(Tn, KB), kv_heads), dim3(256), 0, st, (const TT*)q, (const TT*)k, (const TT*)v,  \
                           (const TT*)dO, lse, delta, (TT*)dk, (TT*)dv, Tn, heads, kv_heads, ldq, ldk, ldo, scale);                                \
    } while (0)
#define L(TT) do { if (D == 128) L2(TT, 128); else L2(TT, 64); } while (0)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
#undef L2
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// Whole-step pieces (llava_mi355x/train.py composes them): elementwise forward ops with HF's rounding points, the gradients of the
// embedding gather / splice and of the projector's bias + GELU, the global gradient norm, and AdamW with fp32 master weights.
// ---------------------------------------------------------------------------------------------------------------
enum { kEwSwiglu = 0, kEwGelu = 1, kEwGeluBwd = 2, kEwAdd = 3 };

// op 0: out = round(round(silu(a)) * b)      LlamaMLP: act_fn(gate_proj(x)) * up_proj(x), each op rounded to T
// op 1: out = gelu_erf(a)                    nn.GELU() of the mlp2x_gelu projector (llava/model/multimodal_projector/builder.py:33-51)
// op 2: out = b * gelu'(a)                   its autograd: 0.5 (1 + erf(a / sqrt2)) + a exp(-a^2 / 2) / sqrt(2 pi)
// op 3: out = a + b                          residual / branch-gradient sum
template <typename T, int OP>
__global__ __launch_bounds__(256) void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = to_f32(a[i]);
    float r;
    if (OP == kEwSwiglu) r = round_to<T>(x / (1.f + expf(-x))) * to_f32(b[i]);
    else if (OP == kEwGelu) r = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    else if (OP == kEwGeluBwd) r = to_f32(b[i]) * (0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f);
    else r = x + to_f32(b[i]);
    out[i] = from_f32<T>(r);
}
// the same arithmetic on 8 elements per lane (16-byte loads / stores of 16-bit tensors): the one-element form ran the residual adds of the backward at 1 TB/s
template <typename T, int OP>
__global__ __launch_bounds__(256) void ew8_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 av = reinterpret_cast<const uint4*>(a)[i];
    uint4 bv = {0u, 0u, 0u, 0u};
    if (OP != kEwGelu) bv = reinterpret_cast<const uint4*>(b)[i];
    const uint32_t aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
    uint32_t ow[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float r[2];
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const float x = hl ? unpack_hi<T>(aw[c]) : unpack_lo<T>(aw[c]);
            const float y = hl ? unpack_hi<T>(bw[c]) : unpack_lo<T>(bw[c]);
            if (OP == kEwSwiglu) r[hl] = round_to<T>(x / (1.f + expf(-x))) * y;
            else if (OP == kEwGelu) r[hl] = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
            else if (OP == kEwGeluBwd) r[hl] = y * (0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f);
            else r[hl] = x + y;
        }
        ow[c] = pack2<T>(r[0], r[1]);
    }
    reinterpret_cast<uint4*>(out)[i] = uint4{ow[0], ow[1], ow[2], ow[3]};
}
void launch_elementwise(int dtype, int op, const void* a, const void* b, void* out, size_t n, hipStream_t st) {
    if (!n) return;
    LMX_REQUIRE(op >= 0 && op <= 3, "elementwise: unknown op");
    const bool vec = dtype != kF32 && (((uintptr_t)a | (uintptr_t)out | (uintptr_t)(b ? b : a)) & 15) == 0;
    const size_t n8 = vec ? n / 8 : 0, done = n8 * 8;
    if (n8) {
        const dim3 g8((unsigned)cdiv64((int64_t)n8, 256));
#define V2(TT, OP) hipLaunchKernelGGL((ew8_kernel<TT, OP>), g8, dim3(256), 0, st, (const TT*)a, (const TT*)b, (TT*)out, n8)
#define V(TT) do { if (op == 0) V2(TT, 0); else if (op == 1) V2(TT, 1); else if (op == 2) V2(TT, 2); else V2(TT, 3); } while (0)
        if (dtype == kBF16) V(bf16_t); else V(f16_t);
#undef V
#undef V2
    }
    if (done < n) {
        const size_t es = dtype == kF32 ? 4 : 2, rest = n - done;
        const char* a2 = static_cast<const char*>(a) + done * es; const char* b2 = b ? static_cast<const char*>(b) + done * es : nullptr; char* o2 = static_cast<char*>(out) + done * es;
        const dim3 grid((unsigned)cdiv64((int64_t)rest, 256));
#define L2(TT, OP) hipLaunchKernelGGL((ew_kernel<TT, OP>), grid, dim3(256), 0, st, (const TT*)a2, (const TT*)b2, (TT*)o2, rest)
#define L(TT) do { if (op == 0) L2(TT, 0); else if (op == 1) L2(TT, 1); else if (op == 2) L2(TT, 2); else L2(TT, 3); } while (0)
        if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
#undef L2
    }
    LMX_CHECK_HIP(hipGetLastError());
}

void launch_cast_f32(int dtype, const float* src, void* dst, size_t n, hipStream_t st) {
    if (!n) return;
#define L(TT) hipLaunchKernelGGL(cast_f32_kernel<TT>, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, st, src, (TT*)dst, n)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// column sums of dy [rows][cols] -> fp32 [cols] (bias gradient).  One thread per column, rows walked in order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void col_sum_kernel(const T* __restrict__ dy, int ld, int rows, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += to_f32(dy[(size_t)r * ld + c]);
    out[c] = s;
}
void launch_col_sum(int dtype, const void* dy, int ld, int rows, int cols, float* out, hipStream_t st) {
    if (cols <= 0) return;
#define L(TT) hipLaunchKernelGGL(col_sum_kernel<TT>, dim3(cdiv(cols, 256)), dim3(256), 0, st, (const TT*)dy, ld, rows, cols, out)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// gradient of the embedding gather + image-feature splice (gather_embed_kernel's plan: src >= 0 token id, -1 pad row, -2-k feature row k):
// token rows add into the fp32 table gradient (a token can occur many times: atomics), feature rows are written once each.
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ src, const T* __restrict__ d, float* __restrict__ dtable, T* __restrict__ dfeats, int H) {
    const int row = blockIdx.x, s = src[row];
    if (s == -1) return;
    const T* dr = d + (size_t)row * H;
    if (s >= 0) { if (dtable) for (int c = threadIdx.x; c < H; c += 256) atomicAdd(dtable + (size_t)s * H + c, to_f32(dr[c])); }
    else if (dfeats) for (int c = threadIdx.x; c < H; c += 256) dfeats[(size_t)(-2 - s) * H + c] = dr[c];
}
void launch_embed_bwd(int dtype, const int* src, const void* d, float* dtable, void* dfeats, int rows, int H, hipStream_t st) {
    if (rows <= 0) return;
#define L(TT) hipLaunchKernelGGL(embed_bwd_kernel<TT>, dim3(rows), dim3(256), 0, st, src, (const TT*)d, dtable, (TT*)dfeats, H)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// acc[0] += sum(x^2)  (global gradient norm, clip_grad_norm_ of the HF Trainer; max_grad_norm = 1.0 by default)
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ x, size_t n, float* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t n8 = n / 8;                                 // 8 elements per load (the flat gradient shards are 64-element aligned)
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n8; c += (size_t)gridDim.x * 256) {
        float v[8]; load8<T>(x + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(v[e], v[e], s);
    }
    if (blockIdx.x == 0) for (size_t i = n8 * 8 + threadIdx.x; i < n; i += 256) { const float v = to_f32(x[i]); s += v * v; }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
void launch_sumsq(int dtype, const void* x, size_t n, float* acc, hipStream_t st) {
    if (!n) return;
    LMX_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0, "sumsq: x must be 16-byte aligned");
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64((int64_t)n, 256 * 8 * 4), 4096);
#define L(TT) hipLaunchKernelGGL(sumsq_kernel<TT>, dim3(grid), dim3(256), 0, st, (const TT*)x, n, acc)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// AdamW (torch.optim.AdamW's update order; DeepSpeed bf16 keeps fp32 master weights and moments, scripts/zero2.json "bf16": auto):
//   g = grad * clip,  clip = min(1, max_norm / (sqrt(gnorm_sq[0]) + 1e-6)) when max_norm > 0
//   p32 *= 1 - lr * wd ;  m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2
//   p32 -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps) ;  param = T(p32)
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T* __restrict__ param, const T* __restrict__ grad, float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ gnorm_sq, float max_norm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float clip = 1.f;
    if (gnorm_sq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f));
    const float g = to_f32(grad[i]) * clip;
    float p = master[i];
    p *= 1.f - lr * wd;
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    p -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    master[i] = p;
    param[i] = from_f32<T>(p);
}
void launch_adamw(int dtype, void* param, const void* grad, float* master, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                  int step, const float* gnorm_sq, float max_norm, hipStream_t st) {
    if (!n) return;
    LMX_REQUIRE(step >= 1, "adamw: step counts from 1");
    const float bc1 = 1.f - powf(b1, (float)step), bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
#define L(TT) hipLaunchKernelGGL(adamw_kernel<TT>, dim3((unsigned)cdiv64((int64_t)n, 256)), dim3(256), 0, st, (TT*)param, (const TT*)grad, master, m, v, n, lr, b1, b2, \
                                 eps, wd, bc1, bc2_sqrt, gnorm_sq, max_norm)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
