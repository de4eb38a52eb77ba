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
#include "gemv2.h"

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

// KS > 1 (round 5): the K range split over KS wave GROUPS inside the workgroup — every group is a complete NW-wave team with its own ring and its own K slice,
// the groups' fp32 accumulators are added in group order through LDS at the end (deterministic), group 0 runs the epilogue.  For problems with fewer tiles than
// CUs whose time is their dependent chain of K-steps (CLIP's N = 1024 linears at 577 rows: 160 tiles, 16 / 64 K-steps): the chain is KS times shorter and the
// CU holds KS times the waves.
template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, int BK, int KS = 1>
__global__ __launch_bounds__(WM * WN * 64 * KS) void gemm_pipe_kernel(GemmArgs a) {
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
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KS > 1 ? wave_all / NW : 0;     // K group of this wave
    const int wave = KS > 1 ? wave_all % NW : wave_all;
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

    const int nk_all = a.K / BK;
    const int nk_g = (nk_all + KS - 1) / KS;        // loop trips of every group (the barriers are workgroup-wide)
    const int k_first = grp * nk_g;
    const int nk = nk_all - k_first < nk_g ? (nk_all - k_first > 0 ? nk_all - k_first : 0) : nk_g;      // K-steps of THIS group
    int xrow[MT], wrow[NTL];
#pragma unroll
    for (int j = 0; j < MT; ++j) xrow[j] = wm * TM + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < NTL; ++i) wrow[i] = BM + wn * TN + i * 32 + l31;

    // LDS-DMA issued from inline asm: hipcc must not see these as LDS writes, or it drains the whole DMA queue
    // (s_waitcnt vmcnt(0)) in front of the first ds_read of every K-step because it cannot prove the slot being filled
    // differs from the slot being read.  Ordering is ours: counted vmcnt + s_barrier below.  M0 carries the wave-uniform
    // LDS destination (lane l lands at base + 16*l); it is compiler-reserved, so it is saved/restored in the statement.
    char* ring = smem + (size_t)grp * NSTAGE * BUF_BYTES;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    auto stage = [&](int kt, unsigned buf_off) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf_off + p * 1024);
            const T* src = gsrc[i] + (size_t)(k_first + kt) * BK;
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
    for (int kt = 0; kt < (KS > 1 ? nk_g : nk); ++kt) {
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
        if (KS == 1 || kt < nk) compute(ring + slot * BUF_BYTES);
        slot = slot + 1 == NSTAGE ? 0 : slot + 1;
    }
    if constexpr (KS > 1) {
        // groups 1 .. KS - 1 hand their accumulators to group 0 through the (now dead) rings: [group - 1][wave][i][j][register][lane] floats, added in group order
        constexpr int ACC_F = NTL * MT * 16 * 64;
        static_assert((KS - 1) * NW * ACC_F * 4 <= KS * NSTAGE * BUF_BYTES, "accumulator exchange must fit the rings");
        float* xch = reinterpret_cast<float*>(smem);
        __syncthreads();                                    // every wave has left its ring (all LDS-DMA landed: the loop's last wait was vmcnt(0))
        if (grp > 0) {
            float* dst = xch + ((size_t)(grp - 1) * NW + wave) * ACC_F;
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((i * MT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float* src = xch + ((size_t)(g - 1) * NW + wave) * ACC_F;
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * MT + j) * 16 + r) * 64 + lane];
        }
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
// gemv2_kernel: the single-row decode linear of 16-bit models with a hand-counted weight stream — body in gemv2.h (shared with the decode step's fused
// k|v projection + attention launch, decode_attn.hip)
// ---------------------------------------------------------------------------------------------
template <typename T, int R, int P, int GEMV2_NX>
__global__ __launch_bounds__(256) void gemv2_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemv2_body<T, R, P, GEMV2_NX, false>(a, (int)blockIdx.x, smem, nullptr, 0u);
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

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, int BK = 64, int KS = 1>
static void launch_gemm_pipe(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = KS * NSTAGE * (BM + BN) * BK * 2;
    static_assert(smem <= 160 * 1024, "LDS ring does not fit");
    if (a.act == kActSiluMul) LMX_REQUIRE((BN / WN / 32) % 2 == 0, "gemm: this tile cannot pair gate/up rows (SiLU·mul needs an even number of 32-wide n tiles per wave)");
    auto kern = gemm_pipe_kernel<T, BM, BN, WM, WN, NSTAGE, BK, KS>;
    static bool attr_set = false;
    if (!attr_set) {
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int mtiles = cdiv(a.M, BM), ntiles = cdiv(a.N, BN);
    LMX_LAUNCH(kern, dim3(mtiles * ntiles), dim3(WM * WN * 64 * KS), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T>
static void launch_gemm16(const GemmArgs& a, int variant, hipStream_t st) {
    // variant: 0 = auto, 2 = 128x128 reg-staged (kept as an independent cross-check of the LDS-DMA path), 4 = 64x128 glds, 5 = 64x64 glds,
    //          7 / 9 = LDS-ring kernels with counted vmcnt (128x256x64 3-slot, 256x256x64 2-slot; 8 waves), 15 = 64x64 4-slot ring (4 waves),
    //          18 = 128x128x64 2-slot 4 waves.  (Round 5 removed the arms auto never picks: 1, 12, 14 and the deeper rings 23 - 26 of round 4, EXPERIMENTS.md r4-C.)
    if (variant == 0) {
        // Large problems: the ping-pong 256x256x64 kernel (gemm8p.hip), K-sliced when N = hidden leaves CUs idle and the caller brought
        // the partial-tile scratch (the engine does, per sequence; LMX_GEMM8P=0 switches the kernel off: the ring kernels below as an A/B and cross-check).
        if (gemm8p_enabled()) {
            const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
            const int S = a.skw ? gemm8p_pick_split(a.M, a.N, a.K) : 1;
            if (tiles * S >= 160) { GemmArgs b = a; b.split_k = S; launch_gemm8p(TypeInfo<T>::id, b, st); return; }
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
            // fewer tiles than CUs and a long K (CLIP fc2: 577 x 1024 x 4096, 160 tiles x 64 K-steps): two K groups per workgroup — 20.5 -> 18.2 us
            // (profiles/r05_vis_gemm_ksplit.jsonl; at K = 1024 the forms tie, with more tiles than CUs the 8-wave workgroups lose)
            variant = t64x64 <= 256 && a.K >= 2048 ? 16 : (t64x64 <= 512 ? 15 : 5);
        }
        else variant = 4;
    }
    switch (variant) {
        case 2: launch_gemm_cfg<T, 128, 128, 2, 2, false>(a, st); break;
        case 4: launch_gemm_cfg<T, 64, 128, 2, 2, true>(a, st); break;
        case 5: LMX_REQUIRE(a.act != kActSiluMul, "gemm: the 64x64 tile has no SiLU·mul epilogue"); launch_gemm_cfg<T, 64, 64, 2, 2, true>(a, st); break;
        case 7: launch_gemm_pipe<T, 128, 256, 2, 4, 3>(a, st); break;          // 128x256x64, 3-slot ring (two slabs in flight)
        case 9: launch_gemm_pipe<T, 256, 256, 2, 4, 2>(a, st); break;          // 256x256x64, 2-slot ring, 128 FLOP per L2 byte
        case 15: LMX_REQUIRE(a.act != kActSiluMul, "gemm: the 64x64 tile has no SiLU·mul epilogue"); launch_gemm_pipe<T, 64, 64, 2, 2, 4>(a, st); break;   // small tiles, 4 waves, 4-slot ring: latency-bound shapes
        // in-workgroup split-K forms of the same tile (round 5): 16 = 2 groups x 4-slot rings (128 KB), 17 = 4 groups x 2-slot rings (128 KB, 16 waves)
        case 16: LMX_REQUIRE(a.act != kActSiluMul, "gemm: the 64x64 tile has no SiLU·mul epilogue"); launch_gemm_pipe<T, 64, 64, 2, 2, 4, 64, 2>(a, st); break;
        case 17: LMX_REQUIRE(a.act != kActSiluMul, "gemm: the 64x64 tile has no SiLU·mul epilogue"); launch_gemm_pipe<T, 64, 64, 2, 2, 2, 64, 4>(a, st); break;
        // N = hidden-size outputs at T ~ 1k (o_proj, down_proj) without split-K scratch and the tensor-parallel rank shapes: 128x128 gives 288 workgroups small
        // enough (64 KB LDS) for two to share a CU, so every CU has work for the whole kernel.
        case 18: launch_gemm_pipe<T, 128, 128, 2, 2, 2>(a, st); break;
        default: throw Error{"gemm: unknown variant " + std::to_string(variant)};
    }
}

bool gemm8p_enabled() { static const bool on = [] { const char* e = getenv("LMX_GEMM8P"); return !(e && atoi(e) == 0); }(); return on; }

bool gemm_fuses_qkv(int dtype, int M, int K, int D, int nh, int nkv, int pos0, int s_max, bool has_bias) {
    if (dtype != kBF16 && dtype != kF16) return false;
    const int N = (nh + 2 * nkv) * D;
    if (!gemm8p_enabled() || has_bias || M <= 0 || K % 64 != 0 || !(D == 64 || D == 128) || (nh * D) % 256 != 0 || (nkv * D) % 256 != 0 || pos0 % 8 != 0 || s_max % 8 != 0)
        return false;
    return cdiv(M, 256) * cdiv(N, 256) >= 160 && gemm8p_pick_split(M, N, K) == 1;      // launch_gemm16's own rule for the un-split ping-pong kernel
}

void launch_gemm(int dtype, const GemmArgs& a, int variant, hipStream_t st) {
    LMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
    if (a.qf_kc) LMX_REQUIRE(variant == 0 && gemm_fuses_qkv(dtype, a.M, a.K, a.qf_D, a.qf_nh, a.qf_nkv, a.qf_pos0, a.qf_smax, a.bias != nullptr),
                             "gemm: the fused q|k|v epilogue needs the un-split ping-pong launch (gemm_fuses_qkv)");
    if (variant == 20) { launch_skinny_gemm(dtype, a, st); return; }
    // 30: ping-pong kernel, K slices chosen by gemm8p_pick_split; 33 / 34 / 35: 2 / 3 / 1 slices forced
    if (variant == 30 || (variant >= 33 && variant <= 35)) {
        GemmArgs b = a;
        if (variant == 35) b.split_k = 1;
        else if (variant >= 33) b.split_k = variant - 31;
        launch_gemm8p(dtype, b, st);
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

template <typename T, int MB, int R>
static void launch_gemv_r(const GemvArgs& a, hipStream_t st) {
    const size_t smem = (size_t)MB * a.K * sizeof(T) + 16;
    auto kern = gemv_kernel<T, MB, R, 2>;
    static bool attr_set = false;
    if (!attr_set) { LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    LMX_LAUNCH(kern, dim3(cdiv(a.N, 4 * R)), dim3(256), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T, int R, int P, int NX>
static void launch_gemv2_rx(const GemvArgs& a, hipStream_t st) {
    const size_t smem = gemv2_smem_bytes(a.K, (int)sizeof(T));
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

// does the single-row linear of this shape take the hand-counted stream (gemv2_body)?  16-byte rows, 32-bit row offsets, K <= 16384, an aligned x row
bool gemv2_applies(int dtype, const GemvArgs& a) {
    if (dtype != kBF16 && dtype != kF16) return false;
    if (((size_t)a.ldw * 2) % 16 != 0 || (size_t)a.N * a.ldw * 2 >= ((size_t)1 << 32)) return false;
    return a.K <= 16384 && reinterpret_cast<uintptr_t>(a.X) % 16 == 0;
}

// The single-row decode linears of 16-bit models take the hand-counted stream.  (R rows per wave, P rounds in flight) per shape from the in-situ sweep
// of round 3 (tools/mb_decode.py, LLaVA-1.5-7B, per launch incl. ~1.4 us of event pair; gemv_kernel -> gemv2_kernel): q|k|v 20.5 -> 19.1 us (R 2, P 4),
// gate|up 30.9 -> 29.5 (2, 4), down 19.9 -> 17.8 (1, 8: K = 11008 keeps 8 KB per wave on the wire), lm_head 45.8 -> 42.2 (2, 4), o_proj 8.7 -> 8.8 (1, 2).
template <typename T>
static bool launch_gemv2(const GemvArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) != 2) return false;
    else {
        if (!gemv2_applies(TypeInfo<T>::id, a)) return false;
        if (a.act == kActSiluMul) launch_gemv2_r<T, 2, 4>(a, st);
        else if (a.K >= 8192) launch_gemv2_r<T, 1, 8>(a, st);
        // small linears (<= 32 M weights): o_proj keeps one row per wave (1024 short-lived workgroups); with the RMSNorm fused — the split-q step's q projection — every
        // workgroup also normalises the whole input row, so half as many workgroups with two rows per wave win: 8.1 -> 7.5 us (profiles/r05_small_gemv_rp.txt)
        else if ((size_t)a.N * a.K <= ((size_t)1 << 25)) { if (a.norm_w) launch_gemv2_r<T, 2, 4>(a, st); else launch_gemv2_r<T, 1, 2>(a, st); }
        else launch_gemv2_r<T, 2, 4>(a, st);
        return true;
    }
}

template <typename T, int MB>
static void launch_gemv_mb(const GemvArgs& a, hipStream_t st) {
    if constexpr (MB == 1) { if (launch_gemv2<T>(a, st)) return; }
    // R weight rows per wave: more rows = more independent 16-byte loads in flight per lane, fewer = more workgroups.
    // cold-cache sweep (tools/mb_gemv_cold.py, 7B shapes): o_proj 8.8 us at R=1 vs 9.9 at R=2; gate|up 31.3 at R=2 vs 32.5 at R=4;
    // qkv / down / lm_head are flat between R=2 and R=4; R=8 loses everywhere (too few workgroups).
    const int R = a.act == kActSiluMul ? 2 : ((size_t)a.N * a.K <= ((size_t)1 << 24) ? 1 : (a.N >= 8192 ? 4 : 2));      // SiLU·mul pairs a gate row with its up row inside one wave
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
    if (dtype == kBF16) launch_gemv_t<bf16_t>(a, MB, st);
    else if (dtype == kF16) launch_gemv_t<f16_t>(a, MB, st);
    else if (dtype == kF32) launch_gemv_t<float>(a, MB, st);
    else throw Error{"gemv: bad dtype"};
}

}  // namespace lmx
