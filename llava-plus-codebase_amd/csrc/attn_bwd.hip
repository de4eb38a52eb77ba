// Causal attention backward on the matrix cores (16-bit models): the gradient of the reference's training attention,
//   llava/train/llama_flash_attn_monkey_patch.py:68-91 (flash_attn_unpadded_qkvpacked_func(..., softmax_scale=None, causal=True), dropout 0),
// i.e. of  o_i = sum_{j<=i} p_ij v_j,  p_ij = softmax_j(scale q_i . k_j):
//   dp_ij = do_i . v_j ;  delta_i = sum_j p_ij dp_ij ;  ds_ij = p_ij (dp_ij - delta_i)
//   dq_i = scale sum_j ds_ij k_j ;  dk_j = scale sum_i ds_ij q_i ;  dv_j = sum_i p_ij do_i        (sums over the heads of a GQA group too)
// Deterministic (no atomics), recomputing (nothing but q, k, v, do comes in; train.hip's launch_attn_bwd keeps the two-pass VALU kernels for fp32).
//
// Both kernels are the flash FORWARD kernel's skeleton (attention.hip: flash_prefill_kernel) — 64-row tiles staged by LDS-DMA into a ring, one side of
// every product held per lane in registers, the index that the softmax statistics belong to on the MFMA column — with more products per tile:
//
//   attn_bwd_dq_mfma_kernel   workgroup = 128 query rows x 1 head (4 waves x 32 rows), lane = query.  Tiles: K, V (row-major).
//       sweep 1 over the visible key tiles:  S^T = K Q^T, dP^T = V dO^T  ->  online max / sum / sum(p dp): lse_i and delta_i (written out for the second kernel);
//                                            SKIPPED when the forward's lse and rowsum(dO o O) come in (lmx_op_attn_bwd_lse: FlashAttention-2's form)
//       sweep 2:                             S^T, dP^T again, dS^T = P^T o (dP^T - delta) as 16-bit fragments, dQ^T += (K tile)^T . dS^T
//   attn_bwd_dkv_mfma_kernel  workgroup = 128 keys x 1 kv head, lane = key.  Tiles (per head of the group, per query tile at or after the keys): Q, dO
//       (row-major), lse / delta of the tile's 64 rows.   S = Q K^T, dP = dO V^T (rows = queries, in registers),  P, dS with the row's lse / delta,
//       dV^T += (dO tile)^T . P,  dK^T += (Q tile)^T . dS.
//
// Round 6: NO transposed copies.  Through round 5 K^T, Q^T and dO^T were made by three launch_transpose calls per sample and staged as tiles of their own beside
// the row-major ones (3 / 4 tiles per ring slot: 144 / 132 KiB of LDS, ONE 4-wave workgroup per CU, one wave per SIMD — every exp / pack / LDS phase of a wave
// left its matrix pipe idle: 0.16 of peak).  The second products' A operands (rows = d, contraction = the tile's 64 rows) now come out of the ROW-MAJOR tile the
// first products already use, through ds_read_b64_tr_b16 (tools/probes/tr_b16_probe.hip: 16 lanes read a [4 rows][16 columns] block, lane i receives column i).
// For that read to be conflict-free on the k_lds_off image (16-byte chunk c of row r at c ^ r) the four rows of a block must differ in bits 2 - 3 of the row index,
// so the order in which tile rows are fed to the first product is bw_perm (bit pairs (0,1) and (2,3) of the row swapped) instead of the forward's fa_key_perm:
// accumulator register r of lane half hi then holds row 4 (r & 3) + 2 ((r >> 2) & 1) + hi + 16 (r >> 3) of the 32-row block, and k-slots 0 .. 3 / 4 .. 7 of a
// second-product step are rows {0, 4, 8, 12} + const — one transpose read each.  Any permutation of the low four row bits keeps the first product's ds_read_b128
// conflict-free (a service group's 16 lanes cover all 16 values of l31 mod 16).  Ring slots shrink to K | V (dq) and Q | dO | statistics (dkv), two slots each:
// 64 / 66 KiB, so TWO workgroups share a CU and each SIMD has a second wave to run while the first is between MFMA chains.
// P and dS are rounded to the model dtype for the MFMAs (as FlashAttention-2 does); sums are fp32.
#include <cstdlib>
#include <map>
#include <mutex>

#include "common.h"
#include "kernels.h"
#include "engine.h"
#include "attention_mfma.h"

namespace lmx {

namespace {

struct BwdArgs {
    const void *Q, *K, *V, *dO;       // [T][ld]: q / do head h at column h * D, k / v kv head g at column g * D
    void *dQ, *dK, *dV;
    float *lse, *delta;               // [heads][Tp]: log2-domain log-sum-exp of the scaled scores, sum_j p dp
    int T, Tp, heads, kv_heads, ldq, ldk, ldo;
    float scale;
    int lse_stride;                   // row stride of lse ([heads][lse_stride]; Tp for the workspace copy)
    int have_stats;                   // lse (the forward's) and delta (= rowsum(dO o O)) are inputs: the dq kernel skips its statistics sweep
};

__device__ __forceinline__ void dma16(const void* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// order in which the rows of a 32-row block are fed to the first product as MFMA A rows: bit pairs (0, 1) and (2, 3) swapped (an involution)
__device__ __forceinline__ int bw_perm(int l31) { return ((l31 & 3) << 2) | ((l31 >> 2) & 3) | (l31 & 16); }
// row inside a 32-row block that accumulator register r of lane half hi holds: bw_perm of the MFMA row (r & 3) + 4 hi + 8 ((r >> 2) & 1) + 16 (r >> 3)
__device__ __forceinline__ int blk_row(int r, int hi) { return 4 * (r & 3) + 2 * ((r >> 2) & 1) + hi + 16 * (r >> 3); }

// stage a ROW-MAJOR tile: rows t * 64 .. + 63 (clamped to T - 1) of a [T][ld] array, D columns from col0, into the k_lds_off image at `dst`
template <typename T, int D>
__device__ __forceinline__ void stage_rows(const T* base, int ld, int col0, int t, int Tn, unsigned dst, int wave, int lane) {
    constexpr int ROWS_PP = 1024 / (D * 2), CPR = D / 8, PPW = (64 * D * 2) / 1024 / 4;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + 4 * i;
        const int row = p * ROWS_PP + lane / CPR;
        const int sw = D == 128 ? row : (row >> 1);
        const int chunk = ((lane % CPR) ^ sw) & (CPR - 1);
        int rg = t * 64 + row; rg = rg < Tn ? rg : Tn - 1;
        dma16(base + (size_t)rg * ld + col0 + chunk * 8, __builtin_amdgcn_readfirstlane(dst + p * 1024));
    }
}
// acc[kb] = tile rows (32-row block kb, bw_perm order) . this lane's row fragments                (the forward's S^T = K Q^T)
template <typename T, int D>
__device__ __forceinline__ void rows_times_lane(const char* tile, const uint4 (&frag)[D / 16], int krow_pi, int hi, f32x16 (&acc)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < D / 16; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const uint4 f = *reinterpret_cast<const uint4*>(tile + k_lds_off<D>(kb * 32 + krow_pi, s * 2 + hi));
            acc[kb] = Mfma32<T>::run(f, frag[s], acc[kb]);
        }
}
// out[db] += (row-major tile)^T . 16-bit fragments of the 64 tile rows: A operand row = column d of the tile, its 8 k-values = tile rows
// kb * 32 + 16 s2 + 4 j + 2 q + hi (j = 0 .. 3: k-slots 4 q + j), fetched as two transpose reads.  16-lane group g = lane >> 4 (hi = g >> 1, columns
// db * 32 + 16 (g & 1) ..): lane i supplies row j = i >> 2, columns 4 (i & 3) .. + 3 of the block and receives column i.  `toff[db][q]` = this lane's byte offset
// for (db, q) at kb = s2 = 0 (tr_offsets below); (kb, s2) add whole rows.
typedef short bw_v4s_t __attribute__((ext_vector_type(4)));
template <int D>
__device__ __forceinline__ void tr_offsets(int lane, int (&toff)[D / 32][2]) {
    const int g = lane >> 4, i = lane & 15, hi = g >> 1;
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = 4 * (i >> 2) + 2 * q + hi;
            const int chunk = db * 4 + 2 * (g & 1) + ((i & 3) >> 1);
            toff[db][q] = k_lds_off<D>(row, chunk) + (i & 1) * 8;
        }
}
template <typename T, int D>
__device__ __forceinline__ void cols_times_frag(const char* tile, const uint4 (&pf)[2][2], const int (&toff)[D / 32][2], f32x16 (&out)[D / 32]) {
    typedef __attribute__((address_space(3))) bw_v4s_t lds_v4s;
    constexpr int ROWB = D * 2;                                       // bytes per tile row; rows 16 apart keep the swizzle (it uses the row's low 4 / 3 bits + bit 0 via row >> 1 for D = 64)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const char* base = tile + (kb * 32 + s2 * 16) * ROWB;
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                const bw_v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(__attribute__((address_space(3))) char*)(base + toff[db][0]));
                const bw_v4s_t up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(__attribute__((address_space(3))) char*)(base + toff[db][1]));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), u2 = __builtin_bit_cast(uint2, up);
                out[db] = Mfma32<T>::run(uint4{l2.x, l2.y, u2.x, u2.y}, pf[kb][s2], out[db]);
            }
        }
}
template <typename T>
__device__ __forceinline__ void pack_frag(const float (&p)[2][16], uint4 (&pf)[2][2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            pf[kb][s2].x = pack2<T>(p[kb][8 * s2 + 0], p[kb][8 * s2 + 1]);
            pf[kb][s2].y = pack2<T>(p[kb][8 * s2 + 2], p[kb][8 * s2 + 3]);
            pf[kb][s2].z = pack2<T>(p[kb][8 * s2 + 4], p[kb][8 * s2 + 5]);
            pf[kb][s2].w = pack2<T>(p[kb][8 * s2 + 6], p[kb][8 * s2 + 7]);
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma_kernel(BwdArgs a) {
    constexpr int KSTEPS = D / 16, DB = D / 32;
    constexpr int RT = 64 * D * 2;                  // bytes of one row-major tile
    constexpr int BUF = 2 * RT;                     // K | V
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int nqb = gridDim.x / a.heads;
    const int head = blockIdx.x % a.heads;
    const int kvh = head / (a.heads / a.kv_heads);
    // heaviest (latest) causal blocks first.  (With two workgroups per CU all 512 of a 2048-row sample are resident at once and launch order is placement; pairing the
    // p-th heaviest with the p-th lightest item on a CU — tried for both plausible placement orders, (w, w + 256) and (w, w + 8) — changed nothing: 97.4 / 99.0 us
    // against 97.8: two co-resident workgroups each run at half speed, the CU's throughput is set by what they share, not by either one's latency chain.)
    const int qb = nqb - 1 - blockIdx.x / a.heads;
    const int q0 = qb * 128 + wave * 32;
    const int qrow = q0 + l31;
    const int qr = qrow < a.T ? qrow : a.T - 1;
    const int krow_pi = bw_perm(l31);

    const T* __restrict__ Kp = reinterpret_cast<const T*>(a.K);
    const T* __restrict__ Vp = reinterpret_cast<const T*>(a.V);

    uint4 qf[KSTEPS], dof[KSTEPS];
    {
        const T* qp = reinterpret_cast<const T*>(a.Q) + (size_t)qr * a.ldq + head * D + hi * 8;
        const T* dp = reinterpret_cast<const T*>(a.dO) + (size_t)qr * a.ldo + head * D + hi * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) { qf[s] = *reinterpret_cast<const uint4*>(qp + s * 16); dof[s] = *reinterpret_cast<const uint4*>(dp + s * 16); }
    }
    int toff[DB][2];
    tr_offsets<D>(lane, toff);
    const int last_q = (qb * 128 + 127 < a.T ? qb * 128 + 127 : a.T - 1);
    const int ntiles = last_q / 64 + 1;
    const float sc = a.scale * 1.4426950408889634f;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    auto stage = [&](int t, int slot) {
        const unsigned base = lds_base + slot * BUF;
        stage_rows<T, D>(Kp, a.ldk, kvh * D, t, a.T, base, wave, lane);
        stage_rows<T, D>(Vp, a.ldk, kvh * D, t, a.T, base + RT, wave, lane);
    };

    // ---- sweep 1: softmax statistics and delta (skipped when the forward's lse and rowsum(dO o O) came in: FlashAttention-2's form) ------------------
    float lse2, delta;
    if (a.have_stats) {
        lse2 = a.lse[(size_t)head * a.lse_stride + qr];
        delta = a.delta[(size_t)head * a.Tp + qr];
    } else {
        float m_run = -1e30f, l_run = 0.f, pd_run = 0.f;
        stage(0, 0);
        for (int t = 0; t < ntiles; ++t) {
            const int slot = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // tile t visible; everyone is done with the other slot
            if (t + 1 < ntiles) stage(t + 1, slot ^ 1);
            const char* kb_ = smem + slot * BUF;
            f32x16 sacc[2], dpacc[2];
            rows_times_lane<T, D>(kb_, qf, krow_pi, hi, sacc);
            rows_times_lane<T, D>(kb_ + RT, dof, krow_pi, hi, dpacc);
            float tmax = -INFINITY;
            float sv[2][16];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kb * 32 + blk_row(r, hi);
                    const float v = key <= qr ? sacc[kb][r] * sc : -INFINITY;
                    sv[kb][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);                  // finite from the first tile on: key 0 is visible to every row
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f, pds = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sv[kb][r] - m_new);     // exp2(-inf) = 0 for masked keys
                    ps += e;
                    pds = fmaf(e, dpacc[kb][r], pds);
                }
            l_run = l_run * alpha + ps;
            pd_run = pd_run * alpha + pds;
            m_run = m_new;
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float pd_tot = pd_run + __shfl_xor(pd_run, 32, 64);
        lse2 = m_run + __builtin_amdgcn_logf(l_tot);                  // v_log_f32 = log2
        delta = pd_tot / l_tot;
        if (hi == 0 && qrow < a.T) { a.lse[(size_t)head * a.lse_stride + qrow] = lse2; a.delta[(size_t)head * a.Tp + qrow] = delta; }
        __builtin_amdgcn_s_barrier();                                 // every wave is out of the ring of sweep 1
    }

    // ---- sweep 2: dQ^T += K^T . dS^T (K^T = the K tile read through the transpose read) ----------------------------------------------------------------
    f32x16 acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    stage(0, 0);
    for (int t = 0; t < ntiles; ++t) {
        const int slot = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) stage(t + 1, slot ^ 1);
        const char* kb_ = smem + slot * BUF;
        f32x16 sacc[2], dpacc[2];
        rows_times_lane<T, D>(kb_, qf, krow_pi, hi, sacc);
        rows_times_lane<T, D>(kb_ + RT, dof, krow_pi, hi, dpacc);
        float ds[2][16];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + kb * 32 + blk_row(r, hi);
                const float p = __builtin_amdgcn_exp2f(sacc[kb][r] * sc - lse2);
                ds[kb][r] = key <= qr ? p * (dpacc[kb][r] - delta) : 0.f;
            }
        uint4 dsf[2][2];
        pack_frag<T>(ds, dsf);
        cols_times_frag<T, D>(kb_, dsf, toff, acc);
    }
    if (qrow < a.T) {
        T* op = reinterpret_cast<T*>(a.dQ) + (size_t)qrow * a.ldq + head * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int d = db * 32 + 8 * q4 + 4 * hi;
                uint2 u;
                u.x = pack2<T>(acc[db][4 * q4 + 0] * a.scale, acc[db][4 * q4 + 1] * a.scale);
                u.y = pack2<T>(acc[db][4 * q4 + 2] * a.scale, acc[db][4 * q4 + 3] * a.scale);
                *reinterpret_cast<uint2*>(op + d) = u;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(BwdArgs a) {
    constexpr int KSTEPS = D / 16, DB = D / 32;
    constexpr int RT = 64 * D * 2;
    constexpr int BUF = 2 * RT + 512;               // Q | dO | lse[64] | delta[64] (the statistics in bw_perm order: see stage)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int kvh = blockIdx.x % a.kv_heads;
    const int kblk = blockIdx.x / a.kv_heads;                   // key block 0 sees every query: heaviest first
    const int group = a.heads / a.kv_heads;
    const int krow = kblk * 128 + wave * 32 + l31;
    const int kr = krow < a.T ? krow : a.T - 1;
    const int qrow_pi = bw_perm(l31);

    const T* __restrict__ Qp = reinterpret_cast<const T*>(a.Q);
    const T* __restrict__ dOp = reinterpret_cast<const T*>(a.dO);

    uint4 kf[KSTEPS], vf[KSTEPS];
    {
        const T* kp = reinterpret_cast<const T*>(a.K) + (size_t)kr * a.ldk + kvh * D + hi * 8;
        const T* vp = reinterpret_cast<const T*>(a.V) + (size_t)kr * a.ldk + kvh * D + hi * 8;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) { kf[s] = *reinterpret_cast<const uint4*>(kp + s * 16); vf[s] = *reinterpret_cast<const uint4*>(vp + s * 16); }
    }
    int toff[DB][2];
    tr_offsets<D>(lane, toff);
    const int t_first = kblk * 2;                               // first 64-row query tile that can see a key of this block
    const int nt = (a.T + 63) / 64 - t_first;                   // >= 1
    const int n_it = group * nt;
    const float sc = a.scale * 1.4426950408889634f;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // LDS position p of the 64 statistics holds query row (p & 32) + bw_perm(p & 31): the four rows {x, x + 4, x + 8, x + 12} that accumulator registers
    // 4 g .. 4 g + 3 hold sit side by side (one 16-byte read)
    const int stat_row = (lane & 32) | bw_perm(lane & 31);

    auto stage = [&](int it, int slot) {
        const int head = kvh * group + it / nt, t = t_first + it % nt;
        const unsigned base = lds_base + slot * BUF;
        stage_rows<T, D>(Qp, a.ldq, head * D, t, a.T, base, wave, lane);
        stage_rows<T, D>(dOp, a.ldo, head * D, t, a.T, base + RT, wave, lane);
        // one more DMA per wave keeps the per-wave count uniform: waves 0 / 1 bring lse / delta of the tile's 64 rows, waves 2 / 3 repeat them
        const float* src = ((wave & 1) ? a.delta + (size_t)head * a.Tp : a.lse + (size_t)head * a.lse_stride) + t * 64 + stat_row;
        dma4(src, __builtin_amdgcn_readfirstlane(base + 2 * RT + (wave & 1) * 256));
    };

    f32x16 accK[DB], accV[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accK[i][r] = 0.f; accV[i][r] = 0.f; }

    stage(0, 0);
    for (int it = 0; it < n_it; ++it) {
        const int slot = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // tile `it` visible; everyone is done with the other slot
        if (it + 1 < n_it) stage(it + 1, slot ^ 1);
        const int t = t_first + it % nt;
        const char* qb_ = smem + slot * BUF;
        const float* lse_s = reinterpret_cast<const float*>(qb_ + 2 * RT);
        const float* del_s = lse_s + 64;
        f32x16 sacc[2], dpacc[2];
        rows_times_lane<T, D>(qb_, kf, qrow_pi, hi, sacc);
        rows_times_lane<T, D>(qb_ + RT, vf, qrow_pi, hi, dpacc);
        float p[2][16], ds[2][16];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // registers 4 g .. 4 g + 3 hold query rows blk_row(4 g, hi) + {0, 4, 8, 12}: positions bw_perm(blk_row(4 g, hi)) .. + 3 of the staged statistics
                const int pos = kb * 32 + 4 * hi + 8 * (g & 1) + 16 * (g >> 1);
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + pos);
                const float4 d4 = *reinterpret_cast<const float4*>(del_s + pos);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int q = t * 64 + kb * 32 + blk_row(r, hi);
                    const bool ok = q < a.T && q >= kr;
                    const float pe = __builtin_amdgcn_exp2f(sacc[kb][r] * sc - lv[e]);
                    p[kb][r] = ok ? pe : 0.f;
                    ds[kb][r] = ok ? pe * (dpacc[kb][r] - dv[e]) : 0.f;
                }
            }
        uint4 pf[2][2], dsf[2][2];
        pack_frag<T>(p, pf);
        pack_frag<T>(ds, dsf);
        cols_times_frag<T, D>(qb_ + RT, pf, toff, accV);          // dV^T += dO^T . P
        cols_times_frag<T, D>(qb_, dsf, toff, accK);              // dK^T += Q^T . dS
    }
    if (krow < a.T) {
        T* kp = reinterpret_cast<T*>(a.dK) + (size_t)krow * a.ldk + kvh * D;
        T* vp = reinterpret_cast<T*>(a.dV) + (size_t)krow * a.ldk + kvh * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int d = db * 32 + 8 * q4 + 4 * hi;
                uint2 u, w;
                u.x = pack2<T>(accK[db][4 * q4 + 0] * a.scale, accK[db][4 * q4 + 1] * a.scale);
                u.y = pack2<T>(accK[db][4 * q4 + 2] * a.scale, accK[db][4 * q4 + 3] * a.scale);
                w.x = pack2<T>(accV[db][4 * q4 + 0], accV[db][4 * q4 + 1]);
                w.y = pack2<T>(accV[db][4 * q4 + 2], accV[db][4 * q4 + 3]);
                *reinterpret_cast<uint2*>(kp + d) = u;
                *reinterpret_cast<uint2*>(vp + d) = w;
            }
    }
}

struct BwdWs { std::mutex mu; std::map<hipStream_t, DevBuf> buf; };      // one grow-only statistics buffer PER STREAM: calls on different streams may overlap
BwdWs g_ws;

}  // namespace

// delta[head][row] = sum_d dO[row][head][d] * O[row][head][d]  (= sum_j p_ij dp_ij: FlashAttention-2's form of the softmax-gradient correction); one wave per (row, head);
// rows T .. Tp - 1 of the output are zeroed (the dkv kernel stages whole 64-row tiles of it)
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ dO, const T* __restrict__ O, float* __restrict__ delta, int Tn, int Tp, int heads, int ldo,
                                                         int ldout) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)Tp * heads) return;
    const int row = (int)(item / heads), head = (int)(item % heads);
    float s = 0.f;
    if (row < Tn && lane * 8 < D) {                                   // D / 8 lanes hold the head's row as 16-byte pieces
        float a[8], b[8];
        load8<T>(dO + (size_t)row * ldo + head * D + lane * 8, a); load8<T>(O + (size_t)row * ldout + head * D + lane * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(a[e], b[e], s);
    }
    s = wave_sum(s);
    if (lane == 0) delta[(size_t)head * Tp + row] = s;
}

bool attn_bwd_mfma_wanted(int dtype, int D) {
    static const bool on = [] { const char* e = getenv("LMX_ATTN_BWD_MFMA"); return !(e && atoi(e) == 0); }();
    return on && (dtype == kBF16 || dtype == kF16) && (D == 64 || D == 128);
}

// Same contract as launch_attn_bwd (train.hip); dk / dv rows have the k / v row stride ldk.  The statistics (lse when it is not an input, delta) live in a grow-only
// workspace per stream: calls on one stream are ordered, calls on different streams (the training step spreads its samples over a few) do not share it.
// out / lse_in (both or neither): the forward's output rows [T][ldout] and its log2-domain log-sum-exp [heads][lse_stride >= T rounded up to 64] (lmx_op_flash_attn_lse) —
// the dq kernel then skips its statistics sweep (2 of its 5 products) and delta comes from rowsum(dO o O), as in FlashAttention-2 (the reference's training attention,
// llava/train/llama_flash_attn_monkey_patch.py:68-91).
void launch_attn_bwd_mfma(int dtype, int D, const void* q, const void* k, const void* v, const void* dO, void* dq, void* dk, void* dv, int Tn, int heads,
                          int kv_heads, int ldq, int ldk, int ldo, float scale, hipStream_t st, const void* out, int ldout, const float* lse_in, int lse_stride) {
    LMX_REQUIRE(attn_bwd_mfma_wanted(dtype, D), "attn_bwd_mfma: 16-bit dtypes, head_dim 64 or 128");
    LMX_REQUIRE(heads % kv_heads == 0 && Tn >= 1 && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0, "attn_bwd_mfma: bad geometry (row strides must keep 16-byte alignment)");
    const int Tp = (Tn + 63) / 64 * 64;
    const size_t n_st = (size_t)heads * Tp;
    const size_t bytes = 2 * n_st * sizeof(float);
    std::lock_guard<std::mutex> lk(g_ws.mu);
    DevBuf& wsb = g_ws.buf[st];
    if (wsb.bytes < bytes) { LMX_CHECK_HIP(hipStreamSynchronize(st)); wsb.ensure(bytes); }
    float* lse = wsb.as<float>(); float* delta = lse + n_st;
    const bool have = out && lse_in;
    LMX_REQUIRE(!have || (lse_stride >= Tp && ldout % 8 == 0), "attn_bwd_mfma: the forward's lse needs a row stride of at least T rounded up to 64, its output 16-byte aligned rows");
    BwdArgs a{q, k, v, dO, dq, dk, dv, have ? const_cast<float*>(lse_in) : lse, delta, Tn, Tp, heads, kv_heads, ldq, ldk, ldo, scale, have ? lse_stride : Tp,
              have ? 1 : 0};
    if (have) {
        const unsigned g = (unsigned)(((long)Tp * heads + 3) / 4);
        if (dtype == kBF16) { if (D == 128) hipLaunchKernelGGL((attn_delta_kernel<bf16_t, 128>), dim3(g), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)out, delta, Tn, Tp, heads, ldo, ldout);
                              else hipLaunchKernelGGL((attn_delta_kernel<bf16_t, 64>), dim3(g), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)out, delta, Tn, Tp, heads, ldo, ldout); }
        else { if (D == 128) hipLaunchKernelGGL((attn_delta_kernel<f16_t, 128>), dim3(g), dim3(256), 0, st, (const f16_t*)dO, (const f16_t*)out, delta, Tn, Tp, heads, ldo, ldout);
               else hipLaunchKernelGGL((attn_delta_kernel<f16_t, 64>), dim3(g), dim3(256), 0, st, (const f16_t*)dO, (const f16_t*)out, delta, Tn, Tp, heads, ldo, ldout); }
    }
    const int nblk = (Tn + 127) / 128;
    const int rt = 64 * D * 2;
    const int smem_dq = 2 * 2 * rt, smem_dkv = 2 * (2 * rt + 512);          // two ring slots each: 64 / 66.5 KiB at D = 128 -> two workgroups per CU
#define LB(TT, DD)                                                                                                                              \
    do {                                                                                                                                       \
        static bool attr = false;                                                                                                              \
        if (!attr) {                                                                                                                           \
            LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_mfma_kernel<TT, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
            LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_mfma_kernel<TT, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr = true;                                                                                                                       \
        }                                                                                                                                      \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<TT, DD>), dim3(nblk * heads), dim3(256), smem_dq, st, a);                                  \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<TT, DD>), dim3(nblk * kv_heads), dim3(256), smem_dkv, st, a);                             \
    } while (0)
    if (dtype == kBF16) { if (D == 128) LB(bf16_t, 128); else LB(bf16_t, 64); }
    else { if (D == 128) LB(f16_t, 128); else LB(f16_t, 64); }
#undef LB
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
