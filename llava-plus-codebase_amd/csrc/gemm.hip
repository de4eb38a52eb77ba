// Linear-layer kernels for gfx950:  C[M,N] = act(X[M,K] · W[N,K]ᵀ + bias) (+ residual)
//
// Replaces the cuBLAS calls reached through torch.nn.Linear in the reference's third-party stack:
//   LLaMA q/k/v/o_proj, gate/up/down_proj, lm_head   (HF5:models/llama/modeling_llama.py:163-176,243-281)
//   CLIP q/k/v/out_proj, fc1/fc2, patch-embed conv   (HF5:models/clip/modeling_clip.py:138-218,259-350)
//   mm_projector Linear/GELU/Linear                   (llava/model/multimodal_projector/builder.py:33-51)
//
// Three kernels:
//   gemm_mfma_kernel   16-bit (bf16 | f16) prefill GEMM on v_mfma_f32_32x32x16_{bf16,f16}; LDS double-buffered,
//                      128-byte rows XOR-swizzled so every ds_read_b128 lane group is conflict-free; tiles are
//                      staged with global_load_lds (LDS-DMA) or, as a cross-check variant, through registers.
//   gemm_f32_kernel    fp32 verification-mode GEMM (VALU, exact fp32 accumulate) — parity mode, not a perf path.
//   gemv_kernel        decode-time M<=4 weight-streaming kernel (HBM-bound): weights straight to VGPRs with
//                      non-temporal 16-byte loads, x staged once per block in LDS, optional fused RMSNorm.
//
// Operand roles in the MFMA: the *weight* fragment is the A operand (rows = n) and the activation fragment is the
// B operand (cols = m), so lane l ends up holding 4 consecutive n for one m per accumulator quad — an 8-byte
// packed store per quad, and bias / SiLU·mul pairing are per-register constants.
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include <cstring>
#include "gemm_common.h"
#include "wstream.h"

namespace lmx {

template <typename T, int BM, int BN, int WM, int WN, bool GLDS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_mfma_kernel(GemmArgs a) {
    constexpr int NW = WM * WN;
    constexpr int NT_ = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN;       // per-wave tile
    constexpr int MT = TM / 32, NTL = TN / 32;      // 32x32 MFMA tiles per wave
    constexpr int ROWS = BM + BN;                   // X rows then W rows in one LDS image
    constexpr int PIECES = ROWS / 8;                // 1-KiB pieces (8 rows x 128 B)
    constexpr int PPW = PIECES / NW;                // pieces per wave
    static_assert(PIECES % NW == 0, "tile rows must split evenly over waves");
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tile");
    constexpr int BUF_BYTES = ROWS * GEMM_ROWB;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;

    const int mtiles = (a.M + BM - 1) / BM;
    const int ntiles = (a.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, mtiles * ntiles);
    const int tile_n = lid / mtiles, tile_m = lid % mtiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
    const T* __restrict__ W = reinterpret_cast<const T*>(a.W);

    // ---- per-lane global source pointers for the PPW pieces this wave stages ------------------------------
    // piece p covers LDS rows 8p..8p+7 linearly; lane l writes row 8p + (l>>3), physical slot (l&7); the data it
    // fetches is logical chunk slot ^ ((row>>1)&7)  (swizzle applied on the SOURCE side: LDS-DMA writes are
    // lane-linear, so the permutation has to live in the global address).
    const T* gsrc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + NW * i;
        const int row = p * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ (row >> 1)) & 7;
        if (row < BM) {
            int m = m0 + row; m = m < a.M ? m : a.M - 1;
            gsrc[i] = X + (size_t)m * a.ldx + chunk * 8;
        } else {
            int n = n0 + row - BM; n = n < a.N ? n : a.N - 1;
            gsrc[i] = W + (size_t)n * a.ldw + chunk * 8;
        }
    }

    f32x16 acc[NTL][MT];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.K / GEMM_BK;

    // fragment row bases (bytes) inside one buffer
    int xrow[MT], wrow[NTL];
#pragma unroll
    for (int j = 0; j < MT; ++j) xrow[j] = wm * TM + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < NTL; ++i) wrow[i] = BM + wn * TN + i * 32 + l31;

    auto compute = [&](const char* buf) {
#pragma unroll
        for (int ks = 0; ks < GEMM_BK / 16; ++ks) {
            uint4 wf[NTL], xf[MT];
            const int chunk = ks * 2 + hi;
#pragma unroll
            for (int i = 0; i < NTL; ++i) wf[i] = *reinterpret_cast<const uint4*>(buf + lds_chunk_off(wrow[i], chunk));
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const uint4*>(buf + lds_chunk_off(xrow[j], chunk));
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = Mfma32x32x16<T>::run(wf[i], xf[j], acc[i][j]);
        }
    };

    if constexpr (GLDS) {
        auto stage = [&](int kt, char* buf) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int p = wave + NW * i;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(gsrc[i] + (size_t)kt * GEMM_BK),
                    (__attribute__((address_space(3))) void*)(buf + p * 1024), 16, 0, 0);
            }
        };
        stage(0, smem);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();   // LDS-DMA of tile kt drained (vmcnt) + every wave is done reading tile kt-1
            if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * BUF_BYTES);
            compute(smem + (kt & 1) * BUF_BYTES);
        }
    } else {
        uint4 regs[PPW];
        auto gload = [&](int kt) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) regs[i] = *reinterpret_cast<const uint4*>(gsrc[i] + (size_t)kt * GEMM_BK);
        };
        auto swrite = [&](char* buf) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int p = wave + NW * i;
                *reinterpret_cast<uint4*>(buf + p * 1024 + lane * 16) = regs[i];
            }
        };
        gload(0);
        swrite(smem);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
            if (kt + 1 < nk) gload(kt + 1);
            compute(smem + (kt & 1) * BUF_BYTES);
            if (kt + 1 < nk) swrite(smem + ((kt + 1) & 1) * BUF_BYTES);
        }
    }

    gemm_epilogue<T, MT, NTL>(a, acc, m0 + wm * TM, n0 + wn * TN, l31, hi);
}

// ---------------------------------------------------------------------------------------------
// Pipelined 16-bit MFMA GEMM: 8 waves, 3-slot LDS ring, TWO K-tiles of LDS-DMA in flight across every barrier.
//   iteration t:  s_waitcnt vmcnt(PPW)  -> this wave's pieces of tile t have landed (tile t+1 may still be flying)
//                 s_barrier             -> everyone's pieces landed AND everyone finished reading slot (t-1)%3
//                 stage tile t+2 into slot (t+2)%3 == (t-1)%3
//                 MFMA over slot t%3
// Raw s_barrier + counted vmcnt (inline asm) instead of __syncthreads(), which would drain the DMA queue (vmcnt(0)).
// Same LDS image / swizzle / fragment mapping / epilogue as gemm_mfma_kernel.
// ---------------------------------------------------------------------------------------------
// LDS offset of 16-byte chunk `chunk` of tile row `row` for a K-slab of BK elements (row pitch BK*2 bytes).
//   BK=64: 8 chunks/row, 2 rows per 256-B bank row  -> chunk ^ (row>>1)&7
//   BK=32: 4 chunks/row, 4 rows per 256-B bank row  -> chunk ^ (row>>2)&3
// Either way the 16 distinct rows of a ds_read_b128 lane group land on 16 distinct 16-byte slots.
template <int BK> __device__ __forceinline__ int lds_chunk_off_bk(int row, int chunk) {
    if constexpr (BK == 64) return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4);
    else return row * 64 + (((chunk ^ (row >> 2)) & 3) << 4);
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, int BK>
__global__ __launch_bounds__(WM * WN * 64) void gemm_pipe_kernel(GemmArgs a) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MT = TM / 32, NTL = TN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int ROWB = BK * 2;                   // bytes per tile row
    constexpr int CPR = BK / 8;                    // 16-byte chunks per row
    constexpr int RPP = 1024 / ROWB;               // rows per 1-KiB LDS-DMA piece
    constexpr int PIECES = ROWS / RPP;
    constexpr int PPW = PIECES / NW;
    static_assert(PIECES % NW == 0, "tile rows must split evenly over waves");
    static_assert(BK == 32 || BK == 64, "K slab");
    static_assert(NSTAGE >= 2 && NSTAGE <= 6, "ring depth");
    static_assert((NSTAGE - 2) * PPW <= 63, "vmcnt is a 6-bit counter");
    constexpr int BUF_BYTES = ROWS * ROWB;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;

    const int mtiles = (a.M + BM - 1) / BM;
    const int ntiles = (a.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, mtiles * ntiles);
    const int tile_n = lid / mtiles, tile_m = lid % mtiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
    const T* __restrict__ W = reinterpret_cast<const T*>(a.W);

    const T* gsrc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + NW * i;
        const int row = p * RPP + lane / CPR;
        const int sw = BK == 64 ? (row >> 1) : (row >> 2);
        const int chunk = ((lane % CPR) ^ sw) & (CPR - 1);
        if (row < BM) {
            int m = m0 + row; m = m < a.M ? m : a.M - 1;
            gsrc[i] = X + (size_t)m * a.ldx + chunk * 8;
        } else {
            int n = n0 + row - BM; n = n < a.N ? n : a.N - 1;
            gsrc[i] = W + (size_t)n * a.ldw + chunk * 8;
        }
    }

    f32x16 acc[NTL][MT];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.K / BK;
    int xrow[MT], wrow[NTL];
#pragma unroll
    for (int j = 0; j < MT; ++j) xrow[j] = wm * TM + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < NTL; ++i) wrow[i] = BM + wn * TN + i * 32 + l31;

    // LDS-DMA issued from inline asm: hipcc must not see these as LDS writes, or it drains the whole DMA queue
    // (s_waitcnt vmcnt(0)) in front of the first ds_read of every K-step because it cannot prove the slot being filled
    // differs from the slot being read.  Ordering is ours: counted vmcnt + s_barrier below.  M0 carries the wave-uniform
    // LDS destination (lane l lands at base + 16*l); it is compiler-reserved, so it is saved/restored in the statement.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto stage = [&](int kt, unsigned buf_off) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf_off + p * 1024);
            const T* src = gsrc[i] + (size_t)kt * BK;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    auto compute = [&](const char* buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            uint4 wf[NTL], xf[MT];
            const int chunk = ks * 2 + hi;
#pragma unroll
            for (int i = 0; i < NTL; ++i) wf[i] = *reinterpret_cast<const uint4*>(buf + lds_chunk_off_bk<BK>(wrow[i], chunk));
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const uint4*>(buf + lds_chunk_off_bk<BK>(xrow[j], chunk));
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = Mfma32x32x16<T>::run(wf[i], xf[j], acc[i][j]);
        }
    };

    // prologue: NSTAGE-1 tiles in flight
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i)
        if (i < nk) stage(i, i * BUF_BYTES);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tiles allowed to stay in flight past this wait: min(NSTAGE-2, tiles remaining after kt)
        const int rem = nk - 1 - kt;
        if (NSTAGE >= 6 && rem >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 6 ? 4 * PPW : 0) : "memory");
        else if (NSTAGE >= 5 && rem >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 5 ? 3 * PPW : 0) : "memory");
        else if (NSTAGE >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTAGE >= 4 ? 2 * PPW : 0) : "memory");
        else if (NSTAGE >= 3 && rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // tile kt visible to every wave; everyone is done reading slot (kt-1) % NSTAGE
        if (kt + NSTAGE - 1 < nk) {
            int s2 = slot + NSTAGE - 1; s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
            stage(kt + NSTAGE - 1, s2 * BUF_BYTES);
        }
        compute(smem + slot * BUF_BYTES);
        slot = slot + 1 == NSTAGE ? 0 : slot + 1;
    }
    gemm_epilogue<T, MT, NTL>(a, acc, m0 + wm * TM, n0 + wn * TN, l31, hi);
}

// ---------------------------------------------------------------------------------------------
// fp32 verification-mode GEMM (VALU; exact fp32 products, k-ordered accumulation per thread)
// 64x64 tile, BK=16, 256 threads, 4x4 outputs per thread. Same epilogue contract as the MFMA kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
    __shared__ float Xs[16][64 + 4];
    __shared__ float Ws[16][64 + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;   // tx -> n (4 each), ty -> m (4 each)
    const int mtiles = (a.M + 63) / 64;
    const int tile_n = blockIdx.x / mtiles, tile_m = blockIdx.x % mtiles;
    const int m0 = tile_m * 64, n0 = tile_n * 64;
    const float* __restrict__ X = reinterpret_cast<const float*>(a.X);
    const float* __restrict__ W = reinterpret_cast<const float*>(a.W);

    const int lr = tid >> 2, lk = (tid & 3) * 4;
    int xm = m0 + lr; xm = xm < a.M ? xm : a.M - 1;
    int wn_ = n0 + lr; wn_ = wn_ < a.N ? wn_ : a.N - 1;
    const float* xp = X + (size_t)xm * a.ldx + lk;
    const float* wp = W + (size_t)wn_ * a.ldw + lk;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < a.K; k0 += 16) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + k0);
        const float4 wv = *reinterpret_cast<const float4*>(wp + k0);
        __syncthreads();
        Xs[lk + 0][lr] = xv.x; Xs[lk + 1][lr] = xv.y; Xs[lk + 2][lr] = xv.z; Xs[lk + 3][lr] = xv.w;
        Ws[lk + 0][lr] = wv.x; Ws[lk + 1][lr] = wv.y; Ws[lk + 2][lr] = wv.z; Ws[lk + 3][lr] = wv.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float xr[4], wr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xr[i] = Xs[k][ty * 4 + i]; wr[i] = Ws[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xr[i], wr[j], acc[i][j]);
        }
    }

    float* __restrict__ C = reinterpret_cast<float*>(a.C);
    const float* __restrict__ bias = reinterpret_cast<const float*>(a.bias);
    const float* R = reinterpret_cast<const float*>(a.R);
    const int act = a.act;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= a.N) continue;
            float v = acc[i][j];
            if (bias) v += bias[n];
            if (act == kActSiluMul) {
                // [32 gate | 32 up] interleave: this thread holds gate when (n % 64) < 32; its partner column n+32
                // lives in thread tx+8 of the same row -> exchange through LDS is avoided by recomputing ownership:
                // handled below by a second pass (see after loop)
                acc[i][j] = v;
                continue;
            }
            v = apply_act(v, act);
            if (R) v += R[(size_t)m * a.ldr + n];
            C[(size_t)m * a.ldc + n] = v;
        }
    }
    if (act == kActSiluMul) {
        // exchange through LDS: write the biased 64x64 tile, then gate threads combine with up values.
        __shared__ float Ts[64][64 + 1];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Ts[ty * 4 + i][tx * 4 + j] = acc[i][j];
        __syncthreads();
        if (tx < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + ty * 4 + i;
                if (m >= a.M) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = tx * 4 + j;          // 0..31 gate column inside this 64 group
                    if (n0 + nl >= a.N) continue;
                    const float g = Ts[ty * 4 + i][nl], u = Ts[ty * 4 + i][nl + 32];
                    C[(size_t)m * a.ldc + n0 / 2 + nl] = act_silu(g) * u;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// decode GEMV: out[mb][n] = act( sum_k xhat[mb][k] * W[n][k] + bias[n] ) (+ residual)
//   xhat = x, or RMSNorm(x) * g (fused, HF rounding points: normalised value rounded to T, then * g rounded to T;
//   HF5:models/llama/modeling_llama.py:53-67).
// One block = 4 waves; each wave owns R weight rows at a time and streams them with 16-byte non-temporal loads.
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
typedef float f32x4_v __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void load8_nt(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8_nt<float>(const float* p, float (&v)[8]) {
    const f32x4_v* q = reinterpret_cast<const f32x4_v*>(p);
    f32x4_v a = __builtin_nontemporal_load(q);
    f32x4_v b = __builtin_nontemporal_load(q + 1);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8_nt<bf16_t>(const bf16_t* p, float (&v)[8]) {
    u32x4_v u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(p));
    v[0] = unpack_lo<bf16_t>(u.x); v[1] = unpack_hi<bf16_t>(u.x);
    v[2] = unpack_lo<bf16_t>(u.y); v[3] = unpack_hi<bf16_t>(u.y);
    v[4] = unpack_lo<bf16_t>(u.z); v[5] = unpack_hi<bf16_t>(u.z);
    v[6] = unpack_lo<bf16_t>(u.w); v[7] = unpack_hi<bf16_t>(u.w);
}
template <> __device__ __forceinline__ void load8_nt<f16_t>(const f16_t* p, float (&v)[8]) {
    u32x4_v u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(p));
    v[0] = unpack_lo<f16_t>(u.x); v[1] = unpack_hi<f16_t>(u.x);
    v[2] = unpack_lo<f16_t>(u.y); v[3] = unpack_hi<f16_t>(u.y);
    v[4] = unpack_lo<f16_t>(u.z); v[5] = unpack_hi<f16_t>(u.z);
    v[6] = unpack_lo<f16_t>(u.w); v[7] = unpack_hi<f16_t>(u.w);
}

// 8 weight elements held raw (as loaded) so the conversion happens at use and the load can be issued early
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    u32x4_v v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(p)); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        f[0] = unpack_lo<bf16_t>(v.x); f[1] = unpack_hi<bf16_t>(v.x); f[2] = unpack_lo<bf16_t>(v.y); f[3] = unpack_hi<bf16_t>(v.y);
        f[4] = unpack_lo<bf16_t>(v.z); f[5] = unpack_hi<bf16_t>(v.z); f[6] = unpack_lo<bf16_t>(v.w); f[7] = unpack_hi<bf16_t>(v.w);
    }
};
template <> struct Raw8<f16_t> {
    u32x4_v v;
    __device__ __forceinline__ void load(const f16_t* p) { v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(p)); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        f[0] = unpack_lo<f16_t>(v.x); f[1] = unpack_hi<f16_t>(v.x); f[2] = unpack_lo<f16_t>(v.y); f[3] = unpack_hi<f16_t>(v.y);
        f[4] = unpack_lo<f16_t>(v.z); f[5] = unpack_hi<f16_t>(v.z); f[6] = unpack_lo<f16_t>(v.w); f[7] = unpack_hi<f16_t>(v.w);
    }
};
template <> struct Raw8<float> {
    f32x4_v a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_v*>(p)); b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_v*>(p) + 1);
    }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};

template <typename T, int MB, int R, int P = 2>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                       // [MB][K]
    float* red = reinterpret_cast<float*>(smem + (size_t)MB * a.K * sizeof(T));   // 4 floats

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, KC = K >> 3;
    const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
    const T* __restrict__ W = reinterpret_cast<const T*>(a.W);
    const bool silu = a.act == kActSiluMul;

    // rows this wave owns
    const int slot0 = (blockIdx.x * 4 + wave) * R;            // first "row slot"
    int rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int f;
        if (silu) { const int j = (slot0 + r) >> 1; f = 64 * (j >> 5) + (j & 31) + 32 * ((slot0 + r) & 1); }
        else f = slot0 + r;
        rows[r] = f < a.N ? f : a.N - 1;
    }

    // ---- issue the first two rounds of weight loads NOW: they are independent of x, so the HBM latency of the stream's
    //      head overlaps the x staging / RMSNorm prologue below (plain loads stay in flight across __syncthreads) --------
    const T* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = W + (size_t)rows[r] * a.ldw;
    // P rounds of 16-byte loads per row stay in flight per lane (P = 2: the original wa / wb pair; deeper rings for the one-row launches,
    // whose waves otherwise keep only 2 KB each on the wire)
    Raw8<T> buf[P][R];
    auto issue = [&](int p, int c) {
        if (c < KC) {
#pragma unroll
            for (int r = 0; r < R; ++r) buf[p][r].load(wrow[r] + c * 8);
        }
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue(p, lane + 64 * p);

    // ---- stage x into LDS: plain copy | RMS-normalised ---------------------------------------------------------------
    for (int mb = 0; mb < MB; ++mb) {
        const T* xr = X + (size_t)mb * a.ldx;
        float inv = 1.f;
        if (a.norm_w) {
            float ss = 0.f;
            for (int c = tid; c < KC; c += 256) {
                float v[8]; load8<T>(xr + c * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
            }
            ss = block_sum<4>(ss, red);
            inv = rsqrtf(ss / (float)K + a.eps);
        }
        const T* g = reinterpret_cast<const T*>(a.norm_w);
        for (int c = tid; c < KC; c += 256) {
            float v[8]; load8<T>(xr + c * 8, v);
            if (g) {
                float gv[8]; load8<T>(g + c * 8, gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * gv[e];
            }
            store8<T>(xs + (size_t)mb * K + c * 8, v);
        }
    }
    __syncthreads();

    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[r][mb] = 0.f;

    auto consume = [&](const Raw8<T> (&w)[R], int c) {
        float wv[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r) w[r].unpack(wv[r]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float xv[8]; load8<T>(xs + (size_t)mb * K + c * 8, xv);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r][mb] = fmaf(wv[r][e], xv[e], acc[r][mb]);
        }
    };
    for (int c = lane; c < KC; c += 64 * P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int cc = c + 64 * p;
            if (cc < KC) {
                consume(buf[p], cc);
                issue(p, cc + 64 * P);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[r][mb] = wave_sum(acc[r][mb]);

    if (lane != 0) return;
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const T* bias = reinterpret_cast<const T*>(a.bias);
    const T* Rr = reinterpret_cast<const T*>(a.R);
    if (silu) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            const int j = (slot0 + r) >> 1;
            if (j >= a.N / 2) continue;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float g = acc[r][mb], u = acc[r + 1][mb];
                if (bias) { g += to_f32(bias[rows[r]]); u += to_f32(bias[rows[r + 1]]); }
                C[(size_t)mb * a.ldc + j] = from_f32<T>(act_silu(g) * u);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = slot0 + r;
        if (n >= a.N) continue;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float v = acc[r][mb];
            if (bias) v += to_f32(bias[n]);
            v = apply_act(v, a.act);
            if (Rr) v += to_f32(Rr[(size_t)mb * a.ldr + n]);
            C[(size_t)mb * a.ldc + n] = from_f32<T>(v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gemv2_kernel: the single-row decode linear with a hand-counted weight stream (wstream.h).  Same block / wave / lane mapping, same staging of x
// (RMSNorm fused, HF rounding points), same per-lane accumulation order, reduction and epilogue as gemv_kernel<T, 1, R> — bit-identical results —
// but R x P loads stay on the wire for the whole row: a round is consumed after `s_waitcnt vmcnt(R (P - 1))` and refilled at once, where hipcc's own
// schedule drains to vmcnt(0) before every consume.  16-bit weights only (a lane's 16 bytes = 8 elements).
// ---------------------------------------------------------------------------------------------
// GEMV2_NX: 16-byte chunks of x per thread (K <= 2048 NX): 2 for the hidden-width inputs of the 7B model, 4 / 6 / 8 up to 8192 / 12288 / 16384
template <typename T, int R, int P, int GEMV2_NX>
__global__ __launch_bounds__(256) void gemv2_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                       // [K]
    float* red = reinterpret_cast<float*>(smem + (size_t)a.K * sizeof(T));   // 4 floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, KC = K >> 3;
    const int NR = (KC + 63) >> 6;                            // load rounds per row
    const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
    const bool silu = a.act == kActSiluMul;

    const int slot0 = (blockIdx.x * 4 + wave) * R;            // first "row slot" of this wave
    int rows[R];
    uint32_t roff[R];                                         // byte offset of each row (wave-uniform: scalar registers)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int f;
        if (silu) { const int j = (slot0 + r) >> 1; f = 64 * (j >> 5) + (j & 31) + 32 * ((slot0 + r) & 1); }
        else f = slot0 + r;
        rows[r] = f < a.N ? f : a.N - 1;
        roff[r] = (uint32_t)__builtin_amdgcn_readfirstlane(rows[r]) * (uint32_t)a.ldw * (uint32_t)sizeof(T);
    }
    const ws_v4i rsW = ws_make_rsrc(a.W, 0x7fffffffu);
    // the residual values this wave will add at the very end: requested FIRST (ahead of the weight stream in the wave's load queue, so the hand-made
    // vmcnt counts below stay exact) instead of as a dependent load after the last reduction (~1 us of L2 latency in front of the store)
    const T* Rr = reinterpret_cast<const T*>(a.R);
    const ws_v4i rsR = ws_make_rsrc(Rr ? a.R : a.W, 0x7fffffffu);
    uint32_t rraw[R];                                         // always issued (no residual: a hot line of W), so the counts below do not depend on it
#pragma unroll
    for (int r = 0; r < R; ++r) ws_load_u16(rraw[r], 0u, rsR, (Rr && slot0 + r < a.N) ? (uint32_t)(slot0 + r) * 2u : 0u);

    // ---- x (one short row, L2-resident) is requested ahead of the weights, in the same counted queue: it lands first and is staged / normalised while
    //      the first P weight rounds are still on the wire.  (As plain loads behind the weight issue its wait was a vmcnt(0): staging started only after
    //      the first P rounds had landed too.)
    const ws_v4i rsX = ws_make_rsrc(a.X, 0x7fffffffu);
    ws_u32x4 xraw[GEMV2_NX];
#pragma unroll
    for (int i = 0; i < GEMV2_NX; ++i) {
        const int c = tid + 256 * i;
        if (c < KC) ws_load_plain(xraw[i], (uint32_t)c * 16u, rsX, 0u);
    }
    const ws_v4i rsG = ws_make_rsrc(a.norm_w ? a.norm_w : a.W, 0x7fffffffu);
    ws_u32x4 graw[GEMV2_NX];                                  // RMSNorm weights of the same chunks (same queue position: older than every weight load)
    if (a.norm_w) {
#pragma unroll
        for (int i = 0; i < GEMV2_NX; ++i) {
            const int c = tid + 256 * i;
            if (c < KC) ws_load_plain(graw[i], (uint32_t)c * 16u, rsG, 0u);
        }
    }

    // ---- the first P rounds go out NOW: they do not depend on x --------------------------------------------------------
    ws_u32x4 buf[P][R];
    auto issue = [&](int p, int j) {                          // round j (wave-uniform) into buffer p; past the row: a dummy load of one hot line
        const int c = lane + 64 * j;
        const uint32_t vo = j < NR ? (uint32_t)(c < KC ? c : KC - 1) * 16u : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) ws_load(buf[p][r], vo, rsW, j < NR ? roff[r] : 0u);
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue(p, p);

    // ---- stage x into LDS: plain copy | RMS-normalised (gemv_kernel's arithmetic; x is read from memory ONCE: the chunks go to LDS raw, the statistics
    //      are taken on the way, and each thread normalises in place the chunks it wrote itself — no second global round trip behind the weight loads) ------
    {
        const T* g = reinterpret_cast<const T*>(a.norm_w);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < GEMV2_NX; ++i) {
            const int c = tid + 256 * i;
            if (c < KC) {
                ws_wait1<P * R>(xraw[i]);                     // everything older than the P x R weight loads = every x load
                const ws_u32x4 raw = xraw[i];
                *reinterpret_cast<ws_u32x4*>(xs + c * 8) = raw;
                if (g) {
                    float v[8]; ws_unpack8<T>(raw, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
                }
            }
        }
        if (g) {
            ss = block_sum<4>(ss, red);
            const float inv = rsqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int i = 0; i < GEMV2_NX; ++i) {
                const int c = tid + 256 * i;
                if (c < KC) {
                    float v[8], gv[8];
                    load8<T>(xs + c * 8, v);
                    ws_wait1<P * R>(graw[i]);
                    ws_unpack8<T>(graw[i], gv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * gv[e];
                    store8<T>(xs + c * 8, v);
                }
            }
        }
    }
    __syncthreads();

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int j0 = 0; j0 < NR; j0 += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int j = j0 + p;                             // wave-uniform
            if (j < NR) {
                ws_wait<R * (P - 1), R>(buf[p]);
                const int cc = lane + 64 * j;
                if (cc < KC) {
                    float xv[8]; load8<T>(xs + cc * 8, xv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float wv[8]; ws_unpack8<T>(buf[p][r], wv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r] = fmaf(wv[e], xv[e], acc[r]);
                    }
                }
                issue(p, j + P);
            }
        }
    }
    ws_drain<P, R>(buf);
    float rres[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ws_landed(rraw[r]); T t; t.x = (uint16_t)rraw[r]; rres[r] = to_f32(t); }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);

    if (lane != 0) return;
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const T* bias = reinterpret_cast<const T*>(a.bias);
    if (silu) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if constexpr (R >= 2) {
                const int j = (slot0 + r) >> 1;
                if (j >= a.N / 2) continue;
                float g = acc[r], u = acc[r + 1];
                if (bias) { g += to_f32(bias[rows[r]]); u += to_f32(bias[rows[r + 1]]); }
                C[j] = from_f32<T>(act_silu(g) * u);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = slot0 + r;
        if (n >= a.N) continue;
        float v = acc[r];
        if (bias) v += to_f32(bias[n]);
        v = apply_act(v, a.act);
        if (Rr) v += rres[r];
        C[n] = from_f32<T>(v);
    }
}

// ---------------------------------------------------------------------------------------------
// gemv2m_kernel: o_proj of a decode step whose attention ran SPLIT over 128-key chunks (decode_flow.hip: flow_attn, attn_form 3).  The attention launch
// stops at the per-chunk partials {o[D], max, sum}; the merge of a head's chunks — which as a hand-over between workgroups inside the attention launch
// cost ~6 us of store -> ticket -> load round trips per layer (EXPERIMENTS.md r3-D) — happens HERE, across the kernel boundary, in every workgroup's
// staging of x: the partial rows (n chunks x K floats, L2-resident after the first workgroup of an XCD touched them) are requested first, the first P
// weight rounds right behind them, and the merged, rounded row goes to LDS while the weights are on the wire.  512 threads = 8 waves x 2 rows, so that
// N / 16 workgroups (one per CU for the 7B / 13B widths) share the extra L2 traffic and a thread's partial loads fit the 6-bit vmcnt.
// Measured (EXPERIMENTS.md r3-I): attention 15.8 -> 11.7 us, o_proj 6.6 -> 10.3 us per layer — EQUAL in sum.  The merge is cheap; what costs is that each of
// the 256 workgroups pulls all n x K partial floats (150 - 260 KB) through its CU's 64 B / clk L1 path before its weights.  Opt-in: LMX_ATTN_MERGE=1.
// Arithmetic: decode_fused_body's merge, operation for operation (groups of chunks s % (256 / D) summed in chunk order, then the groups in order;
// HF eager-attention rounding point = the T-rounded attention output), then gemv2_kernel's stream -> results bit-identical to the two-launch form.
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int NCH, int NSB>
__global__ __launch_bounds__(512) void gemv2m_kernel(GemvArgs a) {
    constexpr int R = 2, P = 4, NT = 512, WS = D + 4, NG = 256 / D, NML = 2;
    constexpr int NO = NCH * NSB * 2, NW = P * R;
    static_assert(NML + NO + NW + R <= 63, "vmcnt is a 6-bit field");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.K, KC = K >> 3;
    const int NR = (KC + 63) >> 6;
    T* xs = reinterpret_cast<T*>(smem);                                                       // [K]
    float* ml = reinterpret_cast<float*>(smem + (((size_t)K * sizeof(T) + 15) & ~(size_t)15)); // [heads * n][2] = {max, sum} of every partial

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ns = a.merge_n, npair = (K / D) * ns;

    const int slot0 = (blockIdx.x * 8 + wave) * R;
    uint32_t roff[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int f = slot0 + r < a.N ? slot0 + r : a.N - 1;
        roff[r] = (uint32_t)__builtin_amdgcn_readfirstlane(f) * (uint32_t)a.ldw * (uint32_t)sizeof(T);
    }
    const ws_v4i rsW = ws_make_rsrc(a.W, 0x7fffffffu);
    const ws_v4i rsP = ws_make_rsrc(a.merge_ws, 0x7fffffffu);
    const T* Rr = reinterpret_cast<const T*>(a.R);
    const ws_v4i rsR = ws_make_rsrc(Rr ? a.R : a.W, 0x7fffffffu);
    uint32_t rraw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ws_load_u16(rraw[r], 0u, rsR, (Rr && slot0 + r < a.N) ? (uint32_t)(slot0 + r) * 2u : 0u);

    // ---- 1. the {max, sum} tails of all partials (pair index = head * n + chunk = the row index of the workspace) ------------------------------------
    ws_u32x4 mlb[NML];
#pragma unroll
    for (int q = 0; q < NML; ++q) {
        const int pi = tid + NT * q;
        ws_load_plain(mlb[q], (uint32_t)((pi < npair ? pi : 0) * WS + D) * 4u, rsP, 0u);
    }
    // ---- 2. this thread's 8 columns of every chunk's partial row (chunks past the live ones: a repeat of the last, never used) ------------------------
    ws_u32x4 ob[NCH][NSB][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = tid + NT * ch, cc = c < KC ? c : KC - 1;
        const int head = (cc * 8) / D, d0 = (cc * 8) % D;
#pragma unroll
        for (int s = 0; s < NSB; ++s) {
            const uint32_t off = (uint32_t)((head * ns + (s < ns ? s : ns - 1)) * WS + d0) * 4u;
            ws_load_plain(ob[ch][s][0], off, rsP, 0u);
            ws_load_plain(ob[ch][s][1], off + 16u, rsP, 0u);
        }
    }
    // ---- 3. the first P weight rounds ----------------------------------------------------------------------------------------------------------------
    ws_u32x4 buf[P][R];
    auto issue = [&](int p, int j) {
        const int c = lane + 64 * j;
        const uint32_t vo = j < NR ? (uint32_t)(c < KC ? c : KC - 1) * 16u : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) ws_load(buf[p][r], vo, rsW, j < NR ? roff[r] : 0u);
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue(p, p);

    // debug (LMX_ATTN_PROBE=1, last layer): in-kernel clock of workgroups 0 and 128 -> a.ts[16 + 8 b + k]
    const int probe_base = (a.ts && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 128)) ? 16 + (blockIdx.x ? 8 : 0) : -1;
    auto probe = [&](int k) { if (probe_base >= 0) a.ts[probe_base + k] = __builtin_amdgcn_s_memrealtime(); };
    probe(0);

    // ---- 4. tails -> LDS ------------------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < NML; ++q) {
        ws_wait1<NO + NW>(mlb[q]);
        const int pi = tid + NT * q;
        if (pi < npair) *reinterpret_cast<float2*>(ml + 2 * pi) = make_float2(__uint_as_float(mlb[q].x), __uint_as_float(mlb[q].y));
    }
    probe(1);
    __syncthreads();
    probe(2);

    // ---- 5. merge (decode_fused_body's arithmetic), round, stage -------------------------------------------------------------------------------------
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = tid + NT * ch, cc = c < KC ? c : KC - 1;
        const int head = (cc * 8) / D;
        const float* mh = ml + 2 * head * ns;                      // the LDS array is padded by NSB pairs: slots past the live chunks read junk, never used
        float mv[NSB], wv[NSB], ev[NSB];
#pragma unroll
        for (int s = 0; s < NSB; ++s) {
            const float2 t = *reinterpret_cast<const float2*>(mh + 2 * s);
            mv[s] = s < ns ? t.x : -INFINITY; wv[s] = t.y;
        }
        float M = -INFINITY;
#pragma unroll
        for (int s = 0; s < NSB; ++s) M = fmaxf(M, mv[s]);
        float l = 0.f;
#pragma unroll
        for (int s = 0; s < NSB; ++s) {
            ev[s] = __builtin_amdgcn_exp2f(mv[s] - M);
            if (mv[s] != -INFINITY) l += ev[s] * wv[s];
        }
        float acc[NG][8];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
#pragma unroll
        for (int s = 0; s < NSB; ++s) {
            ws_wait1<NW>(ob[ch][s][0]);
            ws_wait1<NW>(ob[ch][s][1]);
            if (mv[s] != -INFINITY) {
                const float w = ev[s];
                const ws_u32x4 lo = ob[ch][s][0], hi = ob[ch][s][1];
                float* o = acc[s % NG];
                o[0] += w * __uint_as_float(lo.x); o[1] += w * __uint_as_float(lo.y); o[2] += w * __uint_as_float(lo.z); o[3] += w * __uint_as_float(lo.w);
                o[4] += w * __uint_as_float(hi.x); o[5] += w * __uint_as_float(hi.y); o[6] += w * __uint_as_float(hi.z); o[7] += w * __uint_as_float(hi.w);
            }
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = acc[0][e];
#pragma unroll
            for (int g = 1; g < NG; ++g) o += acc[g][e];
            v[e] = l > 0.f ? o / l : 0.f;
        }
        if (c < KC) store8<T>(xs + c * 8, v);
    }
    probe(3);
    __syncthreads();
    probe(4);

    // ---- 6. the weight stream (gemv2_kernel) ----------------------------------------------------------------------------------------------------------
    float accw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) accw[r] = 0.f;
    for (int j0 = 0; j0 < NR; j0 += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int j = j0 + p;
            if (j < NR) {
                ws_wait<R * (P - 1), R>(buf[p]);
                const int cc = lane + 64 * j;
                if (cc < KC) {
                    float xv[8]; load8<T>(xs + cc * 8, xv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float wv[8]; ws_unpack8<T>(buf[p][r], wv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) accw[r] = fmaf(wv[e], xv[e], accw[r]);
                    }
                }
                issue(p, j + P);
            }
        }
    }
    ws_drain<P, R>(buf);
    probe(5);
    if (a.ts && tid == 0) __hip_atomic_fetch_max(a.ts + 15, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // latest workgroup
    float rres[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ws_landed(rraw[r]); T t; t.x = (uint16_t)rraw[r]; rres[r] = to_f32(t); }
#pragma unroll
    for (int r = 0; r < R; ++r) accw[r] = wave_sum(accw[r]);
    if (lane != 0) return;
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const T* bias = reinterpret_cast<const T*>(a.bias);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = slot0 + r;
        if (n >= a.N) continue;
        float v = accw[r];
        if (bias) v += to_f32(bias[n]);
        v = apply_act(v, a.act);
        if (Rr) v += rres[r];
        C[n] = from_f32<T>(v);
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool GLDS>
static void launch_gemm_cfg(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * GEMM_ROWB;
    auto kern = gemm_mfma_kernel<T, BM, BN, WM, WN, GLDS>;
    static bool attr_set = false;
    if (!attr_set) {
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int mtiles = cdiv(a.M, BM), ntiles = cdiv(a.N, BN);
    LMX_LAUNCH(kern, dim3(mtiles * ntiles), dim3(WM * WN * 64), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, int BK = 64>
static void launch_gemm_pipe(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = NSTAGE * (BM + BN) * BK * 2;
    static_assert(smem <= 160 * 1024, "LDS ring does not fit");
    if (a.act == kActSiluMul) LMX_REQUIRE((BN / WN / 32) % 2 == 0, "gemm: this tile cannot pair gate/up rows (SiLU·mul needs an even number of 32-wide n tiles per wave)");
    auto kern = gemm_pipe_kernel<T, BM, BN, WM, WN, NSTAGE, BK>;
    static bool attr_set = false;
    if (!attr_set) {
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int mtiles = cdiv(a.M, BM), ntiles = cdiv(a.N, BN);
    LMX_LAUNCH(kern, dim3(mtiles * ntiles), dim3(WM * WN * 64), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T>
static void launch_gemm16(const GemmArgs& a, int variant, hipStream_t st) {
    // variant: 0 = auto, 1 = 128x128 glds, 2 = 128x128 reg-staged (cross-check), 4 = 64x128 glds, 5 = 64x64 glds,
    //          7 / 9 / 12 = LDS-ring kernels with counted vmcnt (128x256x64 3-slot, 256x256x64 2-slot, 256x256x32 3-slot; 8 waves),
    //          14 / 15 = small-tile ring kernels (64x128, 64x64; 4 waves, 4-slot ring), 18 = 128x128x64 2-slot 4 waves,
    //          23 .. 26 = deeper-ring arms of round 4 (measured: no gain on the 577-row CLIP shapes — those launches are not bound by bytes in flight)
    if (variant == 0) {
        // Large problems: the ping-pong 256x256x64 kernel (gemm8p.hip), K-sliced when N = hidden leaves CUs idle and the caller brought
        // the partial-tile scratch (the engine does, per sequence; LMX_GEMM8P=0 switches the kernel off for A/B runs).
        static const bool use8p = [] { const char* e = getenv("LMX_GEMM8P"); return !(e && atoi(e) == 0); }();
        if (use8p) {
            const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
            const int S = (a.skw && a.skc) ? gemm8p_pick_split(a.M, a.N, a.K) : 1;
            if (tiles * S >= 160) { GemmArgs b = a; b.split_k = S; launch_gemm8p(TypeInfo<T>::id, b, 0, st); return; }
        }
        // Tile choice from the round-1 microbenchmarks (profiles/r01_microbench.jsonl).  The pipelined ring kernels are
        // bound by L2->LDS bandwidth (~12 TB/s), so the biggest tile that still fills the chip wins; when even 128-row
        // tiles cannot give every CU work (CLIP-sized problems) fall back to the small-tile kernels.
        const long t256 = (long)cdiv(a.M, 256) * cdiv(a.N, 256);
        const long t128x256 = (long)cdiv(a.M, 128) * cdiv(a.N, 256);
        const long t64 = (long)cdiv(a.M, 64) * cdiv(a.N, 128);
        const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
        if (t256 >= 200) variant = 9;                          // 256x256, 2-slot ring   (qkv, gate|up at T~1k; TP=2 gate|up: 215 tiles)
        else if (t128 >= 192 && t128 <= 512)
            variant = 18;                                      // 128x128, 64 KB: 192..512 tiles, all co-resident (two per CU), beat fewer big
                                                               // tiles (o_proj 58 vs 64 us, down_proj 146 vs 152 us at T=1087; TP-rank shapes
                                                               // 1087x3072x4096 44 vs 52 us, 1087x2816x4096 46 vs 51 us: profiles/EXPERIMENTS.md)
        else if (t128x256 >= 128) variant = 7;                 // 128x256, 3-slot ring   (o_proj, down_proj)
        else if (t64 <= 320 && a.act != kActSiluMul) {                  // (<= : CLIP fc1 577 x 4096 x 1024 has exactly 320 -> 64x64 2-slot kernel, 14.6 vs 16.5 us for 64x128)
            // CLIP-sized problems are latency-bound: with few 64x64 tiles (<= 2 per CU) keep three K-slabs in flight per
            // workgroup (4-slot ring, 64 KB LDS); with more tiles the 2-slot kernel's higher occupancy (32 KB) wins.
            const long t64x64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
            variant = t64x64 <= 512 ? 15 : 5;
        }
        else variant = 4;
    }
    switch (variant) {
        case 1: launch_gemm_cfg<T, 128, 128, 2, 2, true>(a, st); break;
        case 2: launch_gemm_cfg<T, 128, 128, 2, 2, false>(a, st); break;
        case 4: launch_gemm_cfg<T, 64, 128, 2, 2, true>(a, st); break;
        case 5: LMX_REQUIRE(a.act != kActSiluMul, "gemm: the 64x64 tile has no SiLU·mul epilogue"); launch_gemm_cfg<T, 64, 64, 2, 2, true>(a, st); break;
        case 7: launch_gemm_pipe<T, 128, 256, 2, 4, 3>(a, st); break;          // 128x256x64, 3-slot ring (two slabs in flight)
        case 9: launch_gemm_pipe<T, 256, 256, 2, 4, 2>(a, st); break;          // 256x256x64, 2-slot ring, 128 FLOP per L2 byte
        case 12: launch_gemm_pipe<T, 256, 256, 2, 4, 3, 32>(a, st); break;     // 256x256x32, 3-slot ring
        case 14: launch_gemm_pipe<T, 64, 128, 2, 2, 4>(a, st); break;          // small tiles, 4 waves, 4-slot ring: latency-bound shapes
        case 15: launch_gemm_pipe<T, 64, 64, 2, 2, 4>(a, st); break;
        // N = hidden-size outputs at T ~ 1k (o_proj, down_proj): 128x256 tiles give only 144 workgroups for 256 CUs; 128x128 gives
        // 288 workgroups small enough (64 KB LDS) for two to share a CU, so every CU has work for the whole kernel.  (64x256x{64,32}
        // and 128x128x32 3-slot were slower: 70 / 88 / 64 us on o_proj.)
        case 18: launch_gemm_pipe<T, 128, 128, 2, 2, 2>(a, st); break;
        // round-4 arms for the CLIP-sized (577-row) and tensor-parallel rank shapes: deeper rings (more bytes in flight per workgroup: these launches are bound by
        // LDS-DMA latency x bytes in flight, not by MFMA issue)
        case 25: launch_gemm_pipe<T, 128, 128, 2, 2, 4>(a, st); break;         // 128x128x64, 4-slot ring, 128 KB: one workgroup per CU, three slabs in flight
        case 26: launch_gemm_pipe<T, 64, 64, 2, 2, 6>(a, st); break;           // 64x64x64, 6-slot ring, 96 KB
        case 23: launch_gemm_pipe<T, 64, 128, 2, 2, 6>(a, st); break;          // 64x128x64, 6-slot ring, 144 KB
        case 24: launch_gemm_pipe<T, 128, 128, 2, 2, 3>(a, st); break;         // 128x128x64, 3-slot ring, 96 KB
         // 128x128x64, 2-slot, 4 waves (64x64 each), 64 KB
        default: throw Error{"gemm: unknown variant " + std::to_string(variant)};
    }
}

int gemm_norm_mode() {
    const char* fe = getenv("LMX_FUSE_NORM");
    const int m = fe ? atoi(fe) : 0;
    return m < 0 || m > 3 ? 0 : m;
}

bool gemm_fuses_norm(int dtype, int M, int N, int K) {
    if (dtype != kBF16 && dtype != kF16) return false;
    static const bool use8p = [] { const char* e = getenv("LMX_GEMM8P"); return !(e && atoi(e) == 0); }();
    // LMX_FUSE_NORM (read per call: a test switches it inside one process): unset / 0 = separate rmsnorm launches (shipping); 1 = the row-owning reduction of
    // round 3 (fused o_proj + down_proj 6.66 ms vs 5.43 ms + 0.51 ms of rmsnorm launches unfused: it gathers 64-byte sectors); 2 = the tile-shaped fused reduction of
    // round 4 (splitk_reduce_rows_norm_kernel: the N-tiles of a row block exchange their partial sums of squares inside the launch — o_proj + down_proj 5.89 ms
    // vs 5.20 ms + 0.51 ms: the exchange turns the reduction into load phase / wait / store phase, +11 us per launch against the 7.5 us a norm launch costs;
    // EXPERIMENTS.md r4-B); 3 = row-major slabs + a row-owning reduction with coalesced reads (split_mode 7, splitk_reduce_rowmajor_kernel): the norm needs
    // nothing from another workgroup.
    const bool fuse = gemm_norm_mode() != 0;
    if (!use8p || !fuse || !gemm8p_boundary_reduce() || M <= 0 || K % 64 != 0 || N % 8 != 0 || N > 8192) return false;
    const int tiles = cdiv(M, 256) * cdiv(N, 256);
    const int S = gemm8p_pick_split(M, N, K);
    return S > 1 && tiles * S >= 160;
}

bool gemm_fuses_qkv(int dtype, int M, int K, int D, int nh, int nkv, int pos0, int s_max, bool has_bias) {
    if (dtype != kBF16 && dtype != kF16) return false;
    static const bool use8p = [] { const char* e = getenv("LMX_GEMM8P"); return !(e && atoi(e) == 0); }();
    const char* fe = getenv("LMX_FUSE_ROPE");                // read per call: a test switches it inside one process
    if (fe && atoi(fe) == 0) return false;
    const int N = (nh + 2 * nkv) * D;
    if (!use8p || gemm8p_tail_split_applies(M, N, K) || has_bias || M <= 0 || K % 64 != 0 || !(D == 64 || D == 128) || (nh * D) % 256 != 0 || (nkv * D) % 256 != 0 || pos0 % 8 != 0 || s_max % 8 != 0)
        return false;
    return cdiv(M, 256) * cdiv(N, 256) >= 160 && gemm8p_pick_split(M, N, K) == 1;      // launch_gemm16's own rule for the un-split ping-pong kernel
}

void launch_gemm(int dtype, const GemmArgs& a, int variant, hipStream_t st) {
    LMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
    if (a.qf_kc) LMX_REQUIRE(variant == 0 && gemm_fuses_qkv(dtype, a.M, a.K, a.qf_D, a.qf_nh, a.qf_nkv, a.qf_pos0, a.qf_smax, a.bias != nullptr),
                             "gemm: the fused q|k|v epilogue needs the un-split ping-pong launch (gemm_fuses_qkv)");
    if (a.norm_w) LMX_REQUIRE(variant == 0 && a.skw && a.skc && gemm_fuses_norm(dtype, a.M, a.N, a.K), "gemm: a fused RMSNorm needs the K-sliced ping-pong launch (gemm_fuses_norm)");
    if (variant == 20) { launch_skinny_gemm(dtype, a, st); return; }
    // 30: ping-pong kernel, K slices chosen by gemm8p_pick_split; 31 / 32: A/B arms (no s_setprio / wave groups in lock-step), unsplit;
    // 33 / 34 / 35: 2 / 3 / 1 slices forced (35: plain order, no tail split, no M-tail); 36: one slice with the tail-split order (K-halves) forced;
    // 37: one slice with the M-tail order (128-row halves) forced
    if (variant >= 30 && variant <= 37) {
        GemmArgs b = a;
        if (variant == 31 || variant == 32 || variant >= 35) b.split_k = 1;
        else if (variant >= 33) b.split_k = variant - 31;
        launch_gemm8p(dtype, b, (variant == 31 || variant == 32) ? variant - 30 : variant == 36 ? 3 : variant == 35 ? 4 : variant == 37 ? 5 : 0, st);
        return;
    }
    LMX_REQUIRE(a.N % 8 == 0, "gemm: N must be a multiple of 8");
    if (a.act == kActSiluMul) LMX_REQUIRE(a.N % 64 == 0, "gemm: SiLU·mul needs N (fused gate|up rows) % 64 == 0");
    if (dtype == kF32) {
        LMX_REQUIRE(a.K % 16 == 0, "gemm f32: K must be a multiple of 16");
        const int mtiles = cdiv(a.M, 64), ntiles = cdiv(a.N, 64);
        LMX_LAUNCH(gemm_f32_kernel, dim3(mtiles * ntiles), dim3(256), 0, st, a);
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
    LMX_REQUIRE(a.K % GEMM_BK == 0, "gemm: K must be a multiple of 64 (pad the operand)");
    LMX_REQUIRE(a.ldx % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 4 == 0, "gemm: leading dims must keep 16-byte row alignment");
    if (dtype == kBF16) launch_gemm16<bf16_t>(a, variant, st);
    else if (dtype == kF16) launch_gemm16<f16_t>(a, variant, st);
    else throw Error{"gemm: bad dtype"};
}

template <typename T, int MB, int R, int P = 2>
static void launch_gemv_r(const GemvArgs& a, hipStream_t st) {
    const size_t smem = (size_t)MB * a.K * sizeof(T) + 16;
    auto kern = gemv_kernel<T, MB, R, P>;
    static bool attr_set = false;
    if (!attr_set) { LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    LMX_LAUNCH(kern, dim3(cdiv(a.N, 4 * R)), dim3(256), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T, int R, int P, int NX>
static void launch_gemv2_rx(const GemvArgs& a, hipStream_t st) {
    const size_t smem = (size_t)a.K * sizeof(T) + 16;
    auto kern = gemv2_kernel<T, R, P, NX>;
    static bool attr_set = false;
    if (!attr_set) { LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    LMX_LAUNCH(kern, dim3(cdiv(a.N, 4 * R)), dim3(256), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}
template <typename T, int R, int P>
static void launch_gemv2_r(const GemvArgs& a, hipStream_t st) {
    if (a.K <= 4096) launch_gemv2_rx<T, R, P, 2>(a, st);
    else if (a.K <= 8192) launch_gemv2_rx<T, R, P, 4>(a, st);
    else if (a.K <= 12288) launch_gemv2_rx<T, R, P, 6>(a, st);
    else launch_gemv2_rx<T, R, P, 8>(a, st);
}

// The single-row decode linears of 16-bit models take the hand-counted stream.  (R rows per wave, P rounds in flight) per shape from the in-situ sweep
// of round 3 (tools/mb_decode.py, LLaVA-1.5-7B, per launch incl. ~1.4 us of event pair; gemv_kernel -> gemv2_kernel): q|k|v 20.5 -> 19.1 us (R 2, P 4),
// gate|up 30.9 -> 29.5 (2, 4), down 19.9 -> 17.8 (1, 8: K = 11008 keeps 8 KB per wave on the wire), lm_head 45.8 -> 42.2 (2, 4), o_proj 8.7 -> 8.8 (1, 2).
// LMX_GEMV2 = 0 switches back to gemv_kernel, "P" or "P,R" forces one configuration for every shape (sweeps).
template <typename T>
static bool launch_gemv2(const GemvArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) != 2) return false;
    else {
        static const int conf_p = [] { const char* e = getenv("LMX_GEMV2"); return e ? atoi(e) : -1; }();
        static const int conf_r = [] { const char* e = getenv("LMX_GEMV2"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
        if (conf_p == 0) return false;
        if (((size_t)a.ldw * sizeof(T)) % 16 != 0 || (size_t)a.N * a.ldw * sizeof(T) >= ((size_t)1 << 32)) return false;      // 16-byte rows, 32-bit row offsets
        if (a.K > 16384 || reinterpret_cast<uintptr_t>(a.X) % 16 != 0) return false;
        int R, P;
        if (a.act == kActSiluMul) { R = 2; P = 4; }
        else if (a.K >= 8192) { R = 1; P = 8; }
        else if ((size_t)a.N * a.K <= ((size_t)1 << 25)) { R = 1; P = 2; }
        else { R = 2; P = 4; }
        if (conf_p == 2 || conf_p == 4 || conf_p == 8) P = conf_p;
        if (conf_r == 1 || conf_r == 2 || conf_r == 4) R = conf_r;
        if (R == 1 && a.act == kActSiluMul) R = 2;
        if (R == 4 && P == 8) P = 4;                                      // 32 loads x 4 registers would not leave room for the rest
#define G2(RR, PP) launch_gemv2_r<T, RR, PP>(a, st)
        if (R == 4) { if (P == 2) G2(4, 2); else G2(4, 4); }
        else if (R == 2) { if (P == 2) G2(2, 2); else if (P == 4) G2(2, 4); else G2(2, 8); }
        else { if (P == 2) G2(1, 2); else if (P == 4) G2(1, 4); else G2(1, 8); }
#undef G2
        return true;
    }
}

// (chunk slots NSB) x (16-byte chunks of x per thread NCH) instantiated: {8, 16} x 1 and 8 x 2 — 32 partial loads per thread at most
static bool gemv2m_shape(int K, int D, int n, int* nch, int* nsb) {
    if (!(D == 64 || D == 128) || K % D != 0 || n < 1 || n > 16) return false;
    const int KC = K / 8;
    *nch = KC <= 512 ? 1 : 2;
    *nsb = n <= 8 ? 8 : 16;
    return KC <= 1024 && *nch * *nsb <= 16 && (K / D) * n <= 1024;
}

bool gemv_can_merge(int dtype, int K, int D, int n) {
    int nch, nsb;
    return (dtype == kBF16 || dtype == kF16) && gemv2m_shape(K, D, n, &nch, &nsb);
}

template <typename T>
static void launch_gemv2m(const GemvArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) != 2) throw Error{"gemv: merged staging is for 16-bit models"};
    else {
        int nch = 0, nsb = 0;
        LMX_REQUIRE(gemv2m_shape(a.K, a.merge_D, a.merge_n, &nch, &nsb), "gemv: merged staging does not cover this shape (gemv_can_merge)");
        LMX_REQUIRE(a.act != kActSiluMul && !a.norm_w, "gemv: merged staging takes a plain linear");
        LMX_REQUIRE(((size_t)a.ldw * sizeof(T)) % 16 == 0 && (size_t)a.N * a.ldw * sizeof(T) < ((size_t)1 << 32), "gemv: merged staging needs 16-byte rows and 32-bit row offsets");
        const size_t smem = (((size_t)a.K * sizeof(T) + 15) & ~(size_t)15) + ((size_t)(a.K / a.merge_D) * a.merge_n + 16) * 8;
#define GM(DD, CH, SB) LMX_LAUNCH((gemv2m_kernel<T, DD, CH, SB>), dim3(cdiv(a.N, 16)), dim3(512), smem, st, a)
#define GMD(DD) do { if (nch == 2) GM(DD, 2, 8); else if (nsb == 8) GM(DD, 1, 8); else GM(DD, 1, 16); } while (0)
        if (a.merge_D == 128) GMD(128); else GMD(64);
#undef GMD
#undef GM
        LMX_CHECK_HIP(hipGetLastError());
    }
}

template <typename T, int MB>
static void launch_gemv_mb(const GemvArgs& a, hipStream_t st) {
    if constexpr (MB == 1) { if (launch_gemv2<T>(a, st)) return; }
    // R weight rows per wave: more rows = more independent 16-byte loads in flight per lane, fewer = more workgroups.
    static const int r_override = [] { const char* e = getenv("LMX_GEMV_R"); return e ? atoi(e) : 0; }();
    // cold-cache sweep (tools/mb_gemv_cold.py, 7B shapes): o_proj 8.8 us at R=1 vs 9.9 at R=2; gate|up 31.3 at R=2 vs 32.5 at R=4;
    // qkv / down / lm_head are flat between R=2 and R=4; R=8 loses everywhere (too few workgroups).
    int R = a.act == kActSiluMul ? 2 : ((size_t)a.N * a.K <= ((size_t)1 << 24) ? 1 : (a.N >= 8192 ? 4 : 2));
    if (r_override == 1 || r_override == 2 || r_override == 4) R = r_override;
    if (R == 1 && a.act == kActSiluMul) R = 2;            // SiLU·mul pairs a gate row with its up row inside one wave
    // load-pipeline depth per row (single-row batches only: the decode step); LMX_GEMV_P overrides for the microbenchmarks
    static const int p_override = [] { const char* e = getenv("LMX_GEMV_P"); return e ? atoi(e) : 0; }();
    int P = 2;
    if (MB == 1 && (p_override == 2 || p_override == 4 || p_override == 6)) P = p_override;
    if constexpr (MB == 1) {
        if (P == 4) { if (R == 4) launch_gemv_r<T, 1, 4, 4>(a, st); else if (R == 2) launch_gemv_r<T, 1, 2, 4>(a, st); else launch_gemv_r<T, 1, 1, 4>(a, st); return; }
        if (P == 6) { if (R == 4) launch_gemv_r<T, 1, 4, 6>(a, st); else if (R == 2) launch_gemv_r<T, 1, 2, 6>(a, st); else launch_gemv_r<T, 1, 1, 6>(a, st); return; }
    }
    if (R == 4) launch_gemv_r<T, MB, 4>(a, st);
    else if (R == 2) launch_gemv_r<T, MB, 2>(a, st);
    else launch_gemv_r<T, MB, 1>(a, st);
}

template <typename T>
static void launch_gemv_t(const GemvArgs& a, int MB, hipStream_t st) {
    switch (MB) {
        case 1: launch_gemv_mb<T, 1>(a, st); break;
        case 2: launch_gemv_mb<T, 2>(a, st); break;
        case 3: launch_gemv_mb<T, 3>(a, st); break;
        case 4: launch_gemv_mb<T, 4>(a, st); break;
        default: throw Error{"gemv: batch rows must be 1..4"};
    }
}

void launch_gemv(int dtype, const GemvArgs& a, int MB, hipStream_t st) {
    LMX_REQUIRE(a.K % 8 == 0, "gemv: K must be a multiple of 8");
    LMX_REQUIRE((size_t)MB * a.K * dtype_size(dtype) + 16 <= 160 * 1024, "gemv: x does not fit LDS");
    if (a.act == kActSiluMul) LMX_REQUIRE(a.N % 64 == 0, "gemv: SiLU·mul needs N % 64 == 0");
    if (a.merge_ws) {
        LMX_REQUIRE(MB == 1, "gemv: merged staging is for the single-row decode step");
        if (dtype == kBF16) launch_gemv2m<bf16_t>(a, st);
        else if (dtype == kF16) launch_gemv2m<f16_t>(a, st);
        else throw Error{"gemv: merged staging is for 16-bit models"};
        return;
    }
    if (dtype == kBF16) launch_gemv_t<bf16_t>(a, MB, st);
    else if (dtype == kF16) launch_gemv_t<f16_t>(a, MB, st);
    else if (dtype == kF32) launch_gemv_t<float>(a, MB, st);
    else throw Error{"gemv: bad dtype"};
}

}  // namespace lmx
