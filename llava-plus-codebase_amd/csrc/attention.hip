// Attention kernels for gfx950.
//
// Contract reproduced (reference = third-party HF code the reference repo calls, plus its own statement of the fused
// contract): scores = q·kᵀ / sqrt(D) (+ causal mask), softmax in fp32, probabilities · V.
//   HF5:models/llama/modeling_llama.py:191-214 (eager_attention_forward), :243-281 (LlamaAttention.forward)
//   HF5:models/clip/modeling_clip.py:259-335  (CLIP attention, non-causal, d=64)
//   llava/train/llama_flash_attn_monkey_patch.py:79-91 (causal=True, softmax_scale=1/sqrt(d), dropout 0)
//
// flash_prefill2_kernel (16-bit, MFMA 32x32x16): one workgroup = 8 waves on 128 query rows; 64-key tiles of K and Vᵀ go by LDS-DMA into an XOR-swizzled
//   ring.  Both products keep the query index on the MFMA *column* (lane&31): Sᵀ = K·Qᵀ and Oᵀ = Vᵀ·Pᵀ, so the running max /
//   sum / rescale are per-lane scalars and P goes from the score accumulators straight into the next MFMA's B operand
//   with no cross-lane traffic (the key order inside each 16-key MFMA step is permuted identically for P and Vᵀ).
//   (The 4-wave first-generation kernel, 36.8 vs 27.5 us per layer, was removed in round 5.)
// decode_attn_kernel    (any dtype, VALU): one workgroup per (row, head, key-split); used for single-token decode
//   (HBM-bound KV streaming) and as the fp32 verification-mode attention for whole prompts.
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "attention_decode.h"
#include "attention_mfma.h"
#include "attention_batch.h"

namespace lmx {

// flash_prefill2_kernel: the KEYS of a query block split over two wave groups.
// One workgroup = 8 waves: wave w works on query rows 32 (w & 3) .. +32 and on the key tiles of parity w >> 2.  The heaviest causal block of a 1087-row
// prompt walks 17 key tiles alone in the 4-wave kernel (34-39 us per layer while the MFMA work of the launch is ~7 us: one wave per SIMD cannot overlap
// its softmax VALU work, LDS reads and barrier waits with anything); here every wave walks half the tiles and each SIMD holds two waves — one of each
// group — whose MFMA and VALU phases interleave.  Tiles are staged in ROUNDS (tile 2r for group 0, 2r + 1 for group 1; each group's four waves DMA
// their own tile) into a ring of NRND rounds, one workgroup barrier per round.  At the end group 1 hands its (O, m, l) to group 0 through the (then
// free) ring and group 0 merges the two online-softmax states per lane: m = max(m0, m1), O = O0 2^(m0 - m) + O1 2^(m1 - m), l likewise.
template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(512) void flash_prefill2_kernel(FlashArgs a) {
    constexpr int KSTEPS = D / 16;        // MFMA k-steps over the head dim (QKᵀ)
    constexpr int DB = D / 32;            // 32-row blocks of Oᵀ
    constexpr int K_BYTES = FA_KT * D * 2;
    constexpr int V_BYTES = D * FA_KT * 2;
    constexpr int BUF_BYTES = K_BYTES + V_BYTES;
    constexpr int NRND = D == 128 ? 2 : 3;        // rounds in the ring: 2 x 2 x 32 KiB (D = 128) / 3 x 2 x 16 KiB (D = 64)
    constexpr int KROWS_PP = 1024 / (D * 2);      // K rows per 1-KiB DMA piece (4 for D=128, 8 for D=64)
    constexpr int KCPR = D / 8;                   // 16-byte chunks per K row
    constexpr int KPPW = K_BYTES / 1024 / 4;      // K pieces per wave per tile
    constexpr int VPPW = V_BYTES / 1024 / 4;      // Vᵀ pieces per wave per tile (8 rows each)
    constexpr int PPW = KPPW + VPPW;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;        // query-row wave, key group
    const int hi = lane >> 5, l31 = lane & 31;
    // 1-D grid ordered by work: ALL heads' heaviest (latest) causal query block first, then the next one, ... so the workgroups
    // that do not fit the first round (grid > CUs) are the lightest ones, not a few heads' full set
    const int nqb = gridDim.x / a.n_heads;
    const int head = blockIdx.x % a.n_heads;
    const int kvh = head / (a.n_heads / a.n_kv_heads);
    const int qb = nqb - 1 - blockIdx.x / a.n_heads;
    const int q0 = qb * FA_QB + wave * 32;               // this wave's first query row
    const int qrow = q0 + l31;                           // this lane's query row (column of both products)
    const int krow_pi = fa_key_perm(l31);                // key row this lane feeds to QKᵀ as tile row l31

    const T* __restrict__ Q = reinterpret_cast<const T*>(a.Q);
    const T* __restrict__ Kc = reinterpret_cast<const T*>(a.K) + (size_t)kvh * a.s_max * D;
    const T* __restrict__ Vt = reinterpret_cast<const T*>(a.VT) + (size_t)kvh * D * a.s_max;

    // ---- Q fragments: B operand, lane holds Q[qrow][s*16 + hi*8 .. +8) ----------------------------------------
    uint4 qf[KSTEPS];
    {
        const int qr = qrow < a.q_len ? qrow : a.q_len - 1;
        const T* qp = Q + (size_t)qr * a.q_stride + head * D + hi * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = *reinterpret_cast<const uint4*>(qp + s * 16);
        // These are the only vector-memory loads the COMPILER knows of (the tile DMAs below are inline asm).  Left pending into the key loop, its wait-count
        // pass guards their first use there with s_waitcnt vmcnt(KSTEPS - 1) ... vmcnt(0) — counts that, on the hardware's single in-order counter, also cover
        // the DMAs of the NEXT round issued just before: every round waited for its own prefetch inside the S = K Q chain (seen in the ISA, EXPERIMENTS r5-I).
        // Waiting here, with an instruction the pass sees, leaves the loop with the one vmcnt wait per round that is meant.  (Time: unchanged — the tiles come
        // from L2 faster than a round computes — but the ring now works as written.)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0); expcnt / lgkmcnt untouched
    }

    // number of key tiles this workgroup needs
    int kv_end = a.kv_len;
    if (CAUSAL) {
        const int last_q = qb * FA_QB + FA_QB - 1;
        const int lim = a.q_pos0 + (last_q < a.q_len ? last_q : a.q_len - 1) + 1;
        kv_end = lim < kv_end ? lim : kv_end;
    }
    const int ntiles = (kv_end + FA_KT - 1) / FA_KT;

    f32x16 oacc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sc = a.scale * 1.4426950408889634f;      // work in the log2 domain
    const int my_pos = a.q_pos0 + qrow;                   // causal limit of this lane's row

    // ---- K / Vᵀ tiles by LDS-DMA into a 3-slot ring (inline asm: see gemm_pipe_kernel for why) -----------------------
    // per-lane source offsets (elements) inside a tile; the swizzle lives in the SOURCE address, the LDS image is lane-linear
    int ksrc[KPPW], vsrc[VPPW];
#pragma unroll
    for (int i = 0; i < KPPW; ++i) {
        const int p = wave + 4 * i;
        const int row = p * KROWS_PP + lane / KCPR;                       // key row inside the tile
        const int sw = D == 128 ? row : (row >> 1);
        const int chunk = ((lane % KCPR) ^ sw) & (KCPR - 1);
        ksrc[i] = row * D + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < VPPW; ++i) {
        const int p = wave + 4 * i;
        const int row = p * 8 + (lane >> 3);                              // d row
        const int chunk = ((lane & 7) ^ (row >> 1)) & 7;                  // 8-key group
        vsrc[i] = row * a.s_max + chunk * 8;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto dma = [&](const T* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto stage = [&](int t, int slot) {
        const unsigned base = lds_base + slot * BUF_BYTES;
        const T* kt = Kc + (size_t)t * FA_KT * D;
        const T* vt = Vt + (size_t)t * FA_KT;
#pragma unroll
        for (int i = 0; i < KPPW; ++i) dma(kt + ksrc[i], __builtin_amdgcn_readfirstlane(base + (wave + 4 * i) * 1024));
#pragma unroll
        for (int i = 0; i < VPPW; ++i) dma(vt + vsrc[i], __builtin_amdgcn_readfirstlane(base + K_BYTES + (wave + 4 * i) * 1024));
    };

    const int nrounds = (ntiles + 1) >> 1;
    // this wave's group stages tile 2 r + grp of round r into slot (r % NRND) * 2 + grp (nothing if that tile does not exist)
    auto stage_round = [&](int r) { const int t = 2 * r + grp; if (t < ntiles) stage(t, (r % NRND) * 2 + grp); };
#pragma unroll
    for (int r = 0; r < NRND - 1; ++r)
        if (r < nrounds) stage_round(r);
    for (int r = 0; r < nrounds; ++r) {
        const int t = 2 * r + grp;
        // everything this wave staged for round r has landed; with a 3-round ring round r + 1 may still be on the wire
        if (NRND == 3 && r + 1 < nrounds && 2 * (r + 1) + grp < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // round r visible to every wave; everyone is done with round r - 1 (its slots are free)
        if (r + NRND - 1 < nrounds) stage_round(r + NRND - 1);
        if (t >= ntiles) continue;             // odd tile count: group 1 sits out the last round (it still takes the barriers)
        const char* kb_ = smem + ((r % NRND) * 2 + grp) * BUF_BYTES;
        const char* vb_ = kb_ + K_BYTES;

        // ---- Sᵀ = K · Qᵀ : sacc[kb][r] = S[key = t*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hi][q = qrow] ------------
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
        // the two 32-key blocks are independent accumulator chains: alternate them so back-to-back MFMAs never depend
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint4 kf = *reinterpret_cast<const uint4*>(kb_ + k_lds_off<D>(kb * 32 + krow_pi, s * 2 + hi));
                sacc[kb] = Mfma32<T>::run(kf, qf[s], sacc[kb]);
            }
        }

        // ---- online softmax (per lane = per query row) -------------------------------------------------------
        // masking only on tiles that can contain an invisible key for some row of this wave (wave-uniform test)
        float p[2][16];
        float tmax = -INFINITY;
        const int tile_last = t * FA_KT + FA_KT - 1;
        const bool need_mask = tile_last >= a.kv_len || (CAUSAL && tile_last > a.q_pos0 + q0);
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * FA_KT + kb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
                    bool ok = key < a.kv_len;
                    if (CAUSAL) ok = ok && (key <= my_pos);
                    const float v = ok ? sacc[kb][r] * sc : -INFINITY;
                    p[kb][r] = v;
                    tmax = fmaxf(tmax, v);
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = sacc[kb][r] * sc;
                    p[kb][r] = v;
                    tmax = fmaxf(tmax, v);
                }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // defer-max: keep the old running max (no O / l rescale) while no row of the wave grew by more than 2^FA_DEFER;
        // P is then bounded by 2^FA_DEFER instead of 1, which fp32 sums and 16-bit P fragments absorb.  m_run starts at
        // -1e30, so the first tile always takes the rescale branch (alpha = 0 on zero accumulators).
        float m_new = m_run;
        if (!__all(tmax - m_run <= FA_DEFER)) {
            m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(p[kb][r] - m_new);
                p[kb][r] = e;
                psum += e;
            }
        l_run += psum;

        // ---- P -> 16-bit B fragments: slot e of step (kb, s2) = p[kb][8*s2 + e] ---------------------------------
        uint4 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                pf[kb][s2].x = pack2<T>(p[kb][8 * s2 + 0], p[kb][8 * s2 + 1]);
                pf[kb][s2].y = pack2<T>(p[kb][8 * s2 + 2], p[kb][8 * s2 + 3]);
                pf[kb][s2].z = pack2<T>(p[kb][8 * s2 + 4], p[kb][8 * s2 + 5]);
                pf[kb][s2].w = pack2<T>(p[kb][8 * s2 + 6], p[kb][8 * s2 + 7]);
            }

        // ---- Oᵀ += Vᵀ · Pᵀ : A slot e of step (kb,s2) is key kb*32 + 16*s2 + 8*hi + e (see fa_key_perm: one 16-byte read) -------
        // key steps outer, d blocks inner: DB independent accumulator chains per step
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int chunk = kb * 4 + s2 * 2 + hi;          // 16-byte chunk = this lane's 8 consecutive keys
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const uint4 vf = *reinterpret_cast<const uint4*>(vb_ + vt_lds_chunk(db * 32 + l31, chunk));
                    oacc[db] = Mfma32<T>::run(vf, pf[kb][s2], oacc[db]);
                }
            }
    }

    // ---- merge the two key groups (group 1 -> LDS -> group 0), then O[q][d] = oacc / l ---------------------------------------------------
    __builtin_amdgcn_s_barrier();              // every wave is done with the ring
    float* mg = reinterpret_cast<float*>(smem) + (size_t)wave * (DB * 16 + 2) * 64;       // [reg][lane] of this query-row wave
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[(i * 16 + r) * 64 + lane] = oacc[i][r];
        mg[(DB * 16) * 64 + lane] = m_run;
        mg[(DB * 16 + 1) * 64 + lane] = l_run;
    }
    __syncthreads();
    if (grp == 1) return;
    {
        const float m1 = mg[(DB * 16) * 64 + lane], l1 = mg[(DB * 16 + 1) * 64 + lane];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
        l_run = l_run * a0 + l1 * a1;
        m_run = m;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = oacc[i][r] * a0 + mg[(i * 16 + r) * 64 + lane] * a1;
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (a.lse && hi == 0 && qrow < a.q_len) a.lse[(size_t)head * a.lse_stride + qrow] = m_run + __builtin_amdgcn_logf(l_tot);      // v_log_f32 = log2
    if (qrow < a.q_len) {
        T* op = reinterpret_cast<T*>(a.O) + (size_t)qrow * a.o_stride + head * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int d = db * 32 + 8 * q4 + 4 * hi;
                uint2 u;
                u.x = pack2<T>(oacc[db][4 * q4 + 0] * inv, oacc[db][4 * q4 + 1] * inv);
                u.y = pack2<T>(oacc[db][4 * q4 + 2] * inv, oacc[db][4 * q4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + d) = u;
            }
    }
}

void launch_flash_prefill(int dtype, int D, const FlashArgs& a, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "flash prefill is the 16-bit path (fp32 verification uses decode_attn)");
    LMX_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
    LMX_REQUIRE(a.s_max % 64 == 0, "KV cache length must be a multiple of 64");
    LMX_REQUIRE(a.kv_len <= a.s_max && a.q_len > 0, "bad lengths");
    LMX_REQUIRE(a.q_stride % 8 == 0 && a.o_stride % 4 == 0, "q/o strides must keep 16-byte alignment");
    const dim3 grid(cdiv(a.q_len, FA_QB) * a.n_heads, 1, 1);
    {
        // two key groups per query block (flash_prefill2_kernel): 8 waves, ring of 2 (D = 128) / 3 (D = 64) rounds of two tiles
        const int smem2 = (D == 128 ? 2 : 3) * 2 * (FA_KT * D * 2 + D * FA_KT * 2);
#define LMX_FA2_LAUNCH(TT, DD, CC)                                                                                  \
    do {                                                                                                            \
        auto kern = flash_prefill2_kernel<TT, DD, CC>;                                                              \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem2)); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, grid, dim3(512), smem2, st, a);                                                    \
    } while (0)
        if (dtype == kBF16) {
            if (D == 128) { if (a.causal) LMX_FA2_LAUNCH(bf16_t, 128, true); else LMX_FA2_LAUNCH(bf16_t, 128, false); }
            else          { if (a.causal) LMX_FA2_LAUNCH(bf16_t, 64, true);  else LMX_FA2_LAUNCH(bf16_t, 64, false); }
        } else {
            if (D == 128) { if (a.causal) LMX_FA2_LAUNCH(f16_t, 128, true); else LMX_FA2_LAUNCH(f16_t, 128, false); }
            else          { if (a.causal) LMX_FA2_LAUNCH(f16_t, 64, true);  else LMX_FA2_LAUNCH(f16_t, 64, false); }
        }
#undef LMX_FA2_LAUNCH
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// decode / verification attention: grid (n_heads, n_split, n_rows), 256 threads.
//   phase 1  scores: D/8 lanes per key (16-byte K loads), shuffle-reduced, written to LDS in the log2 domain
//   phase 2  block max, exp2, block sum
//   phase 3  o[d] = sum_k p[k] * Vᵀ[d][k]: 8 lanes per d row read one 128-byte line of 64 keys per step
//   partial (m, l, o[D]) -> workspace; combine kernel merges the splits.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_attn_kernel(DecodeAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc_lds = reinterpret_cast<float*>(smem);      // [lds_keys] scores / probabilities of this split

    const int tid = threadIdx.x, lane = tid & 63;
    const int head = blockIdx.x, split = blockIdx.y, row = blockIdx.z;
    const int kvh = head / (a.n_heads / a.n_kv_heads);
    const int pos0 = a.pos_ptr ? *a.pos_ptr : a.pos0;
    const int kv_len = a.causal ? (pos0 + row + 1) : a.kv_total;
    // key range of this split: chunks are multiples of 64 keys
    const int chunk = ((((kv_len + a.n_split - 1) / a.n_split) + 63) / 64) * 64;
    const int k_begin = split * chunk;
    int k_end = k_begin + chunk; k_end = k_end < kv_len ? k_end : kv_len;
    const int nk = k_end > k_begin ? k_end - k_begin : 0;
    // LDS carve: scores live in [0, lds_keys) floats, reduction scratch after them
    const int lds_keys = a.causal ? a.s_max : ((a.kv_total + 63) & ~63);
    float* red = sc_lds + lds_keys;

    const T* __restrict__ Q = reinterpret_cast<const T*>(a.Q) + (size_t)row * a.q_stride + head * D;
    const T* __restrict__ Kc = reinterpret_cast<const T*>(a.K) + (size_t)kvh * a.s_max * D;
    const T* __restrict__ Vt = reinterpret_cast<const T*>(a.VT) + (size_t)kvh * D * a.s_max;
    float* ws = a.ws + ((size_t)(row * a.n_heads + head) * a.n_split + split) * (D + 2);

    constexpr int LPK = D / 8;            // lanes per key
    constexpr int KPW = 64 / LPK;         // keys per wave-iteration
    const float scl = a.scale * 1.4426950408889634f;

    // ---- phase 1: scores -------------------------------------------------------------------------------------
    {
        const int sub = lane % LPK, kslot = lane / LPK;
        float qv[8]; load8<T>(Q + sub * 8, qv);
        const int wave = tid >> 6;
        for (int k0 = wave * KPW; k0 < nk; k0 += 4 * KPW) {
            const int kl = k0 + kslot;
            const int key = k_begin + (kl < nk ? kl : nk - 1);
            float kv[8]; load8<T>(Kc + (size_t)key * D + sub * 8, kv);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kv[e], s);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (sub == 0 && kl < nk) sc_lds[kl] = s * scl;
        }
    }
    __syncthreads();

    // ---- phase 2: softmax statistics ---------------------------------------------------------------------------
    float mx = -INFINITY;
    for (int k = tid; k < nk; k += 256) mx = fmaxf(mx, sc_lds[k]);
    mx = block_max<4>(mx, red);
    float sum = 0.f;
    const int nk_pad = (nk + 63) & ~63;
    for (int k = tid; k < nk_pad; k += 256) {
        float e = 0.f;
        if (k < nk) e = __builtin_amdgcn_exp2f(sc_lds[k] - mx);
        sc_lds[k] = e;                          // zero-fills the tail of the last 64-key line
        sum += e;
    }
    sum = block_sum<4>(sum, red);
    __syncthreads();

    // ---- phase 3: o = P · V ---------------------------------------------------------------------------------------
    {
        const int sub = tid & 7, drow = tid >> 3;            // 32 d rows per pass
#pragma unroll
        for (int db = 0; db < D / 32; ++db) {
            const int d = db * 32 + drow;
            float acc = 0.f;
            for (int kb = 0; kb < nk_pad; kb += 64) {
                float vv[8]; load8<T>(Vt + (size_t)d * a.s_max + k_begin + kb + sub * 8, vv);
                const float4 p0 = *reinterpret_cast<const float4*>(sc_lds + kb + sub * 8);
                const float4 p1 = *reinterpret_cast<const float4*>(sc_lds + kb + sub * 8 + 4);
                acc = fmaf(p0.x, vv[0], acc); acc = fmaf(p0.y, vv[1], acc);
                acc = fmaf(p0.z, vv[2], acc); acc = fmaf(p0.w, vv[3], acc);
                acc = fmaf(p1.x, vv[4], acc); acc = fmaf(p1.y, vv[5], acc);
                acc = fmaf(p1.z, vv[6], acc); acc = fmaf(p1.w, vv[7], acc);
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (sub == 0) ws[2 + d] = acc;
        }
    }
    if (tid == 0) { ws[0] = nk > 0 ? mx : -INFINITY; ws[1] = sum; }
}

template <typename T, int D>
__global__ __launch_bounds__(D) void decode_attn_combine_kernel(DecodeAttnArgs a) {
    const int head = blockIdx.x, row = blockIdx.y, d = threadIdx.x;
    const float* ws = a.ws + (size_t)(row * a.n_heads + head) * a.n_split * (D + 2);
    float M = -INFINITY;
    for (int s = 0; s < a.n_split; ++s) M = fmaxf(M, ws[s * (D + 2)]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < a.n_split; ++s) {
        const float m = ws[s * (D + 2)];
        if (m == -INFINITY) continue;
        const float w = __builtin_amdgcn_exp2f(m - M);
        l += w * ws[s * (D + 2) + 1];
        o += w * ws[s * (D + 2) + 2 + d];
    }
    T* op = reinterpret_cast<T*>(a.O) + (size_t)row * a.o_stride + head * D;
    op[d] = from_f32<T>(l > 0.f ? o / l : 0.f);
}

size_t decode_attn_ws_floats(int n_rows, int n_heads, int n_split, int D) {
    return (size_t)n_rows * n_heads * n_split * (D + 2);
}

template <typename T, int D>
static void launch_decode_attn_t(const DecodeAttnArgs& a, hipStream_t st) {
    const int lds_keys = a.causal ? a.s_max : ((a.kv_total + 63) & ~63);
    const size_t smem = (size_t)lds_keys * 4 + 64;
    auto kern = decode_attn_kernel<T, D>;
    static bool attr_set = false;
    if (!attr_set) {
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.n_heads, a.n_split, a.n_rows), dim3(256), smem, st, a);
    LMX_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((decode_attn_combine_kernel<T, D>), dim3(a.n_heads, a.n_rows), dim3(D), 0, st, a);
    LMX_CHECK_HIP(hipGetLastError());
}

void launch_decode_attn(int dtype, int D, const DecodeAttnArgs& a, hipStream_t st) {
    LMX_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
    LMX_REQUIRE(a.s_max % 64 == 0, "KV cache length must be a multiple of 64");
    LMX_REQUIRE(a.n_split >= 1 && a.ws != nullptr, "decode attention needs a workspace");
    LMX_REQUIRE((size_t)(a.causal ? a.s_max : a.kv_total + 64) * 4 + 64 <= 160 * 1024, "score row does not fit LDS");
#define LMX_DA(TT)                                                                                     \
    do { if (D == 128) launch_decode_attn_t<TT, 128>(a, st); else launch_decode_attn_t<TT, 64>(a, st); } while (0)
    if (dtype == kBF16) LMX_DA(bf16_t);
    else if (dtype == kF16) LMX_DA(f16_t);
    else if (dtype == kF32) LMX_DA(float);
    else throw Error{"decode_attn: bad dtype"};
#undef LMX_DA
}

}  // namespace lmx

// ---------------------------------------------------------------------------------------------------------------
// Fused single-token decode attention: RoPE(q, k_new) + KV append + split-K partials in one launch.
// grid (n_heads, n_split), 256 threads.  The newest key/value never round-trips through the cache inside this launch:
// every block whose key range contains `pos` rotates k_new / reads v_new straight from the qkv row, and exactly one block
// per kv head (first q head of the group, owning split) writes them to the caches for later steps.
// ---------------------------------------------------------------------------------------------------------------
namespace lmx {

template <typename T, int D>
__global__ __launch_bounds__(256) void decode_fused_kernel(DecodeFusedArgs a) {
    decode_fused_body<T, D>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

size_t decode_fused_ws_floats(int n_heads, int n_split, int D) { return (size_t)n_heads * n_split * (D + 4); }

void launch_decode_fused(int dtype, int D, const DecodeFusedArgs& a, hipStream_t st) {
    LMX_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
    LMX_REQUIRE(a.s_max % DF_CHUNK == 0 || a.s_max % 64 == 0, "KV cache length must be a multiple of 64");
    // n_split = the caller's count of 128-key chunks to visit: at least every chunk that holds a key of the longest sequence (the host mirrors the positions)
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && (a.n_split - 1) * DF_CHUNK < a.s_max, "decode_fused: n_split must be 1..32 chunks of 128 keys inside the cache");
    LMX_REQUIRE(a.cos_sin && a.O && (a.tab ? a.n_seq >= 1 : (a.ws && a.pos_ptr && a.counters)), "decode_fused: bad arguments");
    if (a.tab && D == 128 && (dtype == kBF16 || dtype == kF16) && a.s_max % BA_PIECE == 0) {
        // the decode batch of 16-bit models with head_dim 128 (attention_batch.h): one workgroup per (sequence, head), waves stream the keys — 44.8 -> 38.9 us per
        // layer at 8 sequences, 143 -> 134 at 32 against the chunked launch below (profiles/r05_batch_attn_wave.jsonl)
        if (dtype == kBF16) launch_decode_attn_wave_t<bf16_t>(a, st); else launch_decode_attn_wave_t<f16_t>(a, st);
        LMX_CHECK_HIP(hipGetLastError());
        return;
    }
#define LMX_DF(TT, DD) hipLaunchKernelGGL((decode_fused_kernel<TT, DD>), dim3(a.n_heads, a.n_split, a.tab ? a.n_seq : 1), dim3(256), 0, st, a)
    if (dtype == kBF16) { if (D == 128) LMX_DF(bf16_t, 128); else LMX_DF(bf16_t, 64); }
    else if (dtype == kF16) { if (D == 128) LMX_DF(f16_t, 128); else LMX_DF(f16_t, 64); }
    else if (dtype == kF32) { if (D == 128) LMX_DF(float, 128); else LMX_DF(float, 64); }
    else throw Error{"decode_fused: bad dtype"};
#undef LMX_DF
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx

// ---------------------------------------------------------------------------------------------------------------
// output_attentions: one workgroup per (query row, head); wave w takes keys w, w + 4, ... (a key row is one coalesced 64-lane read), the scores wait in LDS for
// the row's softmax.  Rounding points of the eager path (kernels.h: AttnProbsArgs).
// ---------------------------------------------------------------------------------------------------------------
namespace lmx {

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_probs_kernel(AttnProbsArgs a) {
    extern __shared__ float ap_sc[];                          // [kv_total] scores, then 4 floats for the reductions
    float* red = ap_sc + a.kv_total;
    const int i = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kvh = h / (a.n_heads / a.n_kv_heads);
    const int n_keys = min(a.pos0 + i + 1, a.kv_total);
    constexpr int E = D / 64;
    const T* q = reinterpret_cast<const T*>(a.Q) + (size_t)i * a.q_stride + (size_t)h * D + lane * E;
    const T* K = reinterpret_cast<const T*>(a.K) + (size_t)kvh * a.s_max * D + lane * E;
    float qv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) qv[e] = to_f32(q[e]);
    for (int j = wave; j < n_keys; j += 4) {
        const T* kr = K + (size_t)j * D;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) s = fmaf(qv[e], to_f32(kr[e]), s);
        s = wave_sum(s);
        if (lane == 0) ap_sc[j] = round_to<T>(round_to<T>(s) * a.scale);
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = tid; j < n_keys; j += 256) m = fmaxf(m, ap_sc[j]);
    m = block_max<4>(m, red);
    float l = 0.f;
    for (int j = tid; j < n_keys; j += 256) { const float p = expf(ap_sc[j] - m); ap_sc[j] = p; l += p; }
    l = block_sum<4>(l, red);
    T* out = reinterpret_cast<T*>(a.P) + ((size_t)h * a.rows_total + a.row0 + i) * a.kv_total;
    for (int j = tid; j < a.kv_total; j += 256) out[j] = from_f32<T>(j < n_keys ? ap_sc[j] / l : 0.f);
}

void launch_attn_probs(int dtype, int D, const AttnProbsArgs& a, hipStream_t st) {
    LMX_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
    LMX_REQUIRE(a.Q && a.K && a.P && a.n_rows >= 1 && a.row0 >= 0 && a.row0 + a.n_rows <= a.rows_total, "attn_probs: bad arguments");
    LMX_REQUIRE(a.pos0 >= 0 && a.pos0 + a.n_rows <= a.kv_total && a.kv_total <= a.s_max, "attn_probs: rows past the key range");
    LMX_REQUIRE(a.n_kv_heads >= 1 && a.n_heads % a.n_kv_heads == 0, "attn_probs: heads must be a multiple of kv heads");
    const size_t smem = ((size_t)a.kv_total + 8) * sizeof(float);
    LMX_REQUIRE(smem <= 64 * 1024, "attn_probs: score row does not fit LDS");
#define LMX_AP(TT, DD) hipLaunchKernelGGL((attn_probs_kernel<TT, DD>), dim3(a.n_rows, a.n_heads), dim3(256), smem, st, a)
    if (dtype == kBF16) { if (D == 128) LMX_AP(bf16_t, 128); else LMX_AP(bf16_t, 64); }
    else if (dtype == kF16) { if (D == 128) LMX_AP(f16_t, 128); else LMX_AP(f16_t, 64); }
    else if (dtype == kF32) { if (D == 128) LMX_AP(float, 128); else LMX_AP(float, 64); }
    else throw Error{"attn_probs: bad dtype"};
#undef LMX_AP
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
