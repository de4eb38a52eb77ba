// Ping-pong prefill GEMM for gfx950: C[M,N] = act(X[M,K] · W[N,K]ᵀ + bias) (+ residual), 16-bit operands, fp32 accumulate.
//
// Replaces the same torch.nn.Linear calls as gemm.hip (HF5:models/llama/modeling_llama.py:163-176,243-281) for the large
// decoder linears of a prefill (q|k|v, gate|up, o_proj, down_proj at T ~ 1k).
//
// Structure (one 512-thread workgroup per 256x256 output tile, one workgroup per CU, 128 KiB of LDS):
//   * 8 waves = 2 (M) x 4 (N); a wave owns a 128 (M) x 64 (N) sub-tile = 4 x 2 accumulators of v_mfma_f32_32x32x16.
//   * Waves w and w+4 share a SIMD.  The two wave GROUPS (w < 4, w >= 4) run the same instruction stream one barrier
//     apart, so on every SIMD one wave is in a pure-MFMA segment while its partner is in a load segment (ds_read of the
//     next fragments + LDS-DMA issue).  The matrix pipe always has a wave to serve; s_setprio(1) covers the MFMA segment.
//   * A K-step (64 k) is consumed in 4 phases, one 64x32 quadrant of the wave tile each (8 MFMAs):
//         phase 0: X half 0 x W half 0     phase 1: X half 0 x W half 1     phase 2: X half 1 x W half 1     phase 3: X half 1 x W half 0
//     so each phase needs at most one new operand half: 12 / 4 / 8 / 0 ds_read_b128 per wave (W half 0 stays in registers).
//   * Operand staging is a ring of 8 HALF-tiles (128 rows x 64 k = 16 KiB: X half = rows {0..63} (+64*half) of both wave rows,
//     W half = rows {0..31} (+32*half) of all four wave columns), filled by LDS-DMA (buffer_load_dwordx4 ... lds, 2 per wave per
//     half-tile) SIX phases ahead of their first use: one half-tile is issued per phase, two stay in flight across the
//     once-per-K-step `s_waitcnt vmcnt(4)`; nothing in the main loop ever waits for vmcnt(0).
//   * Hazards (MI355X_MICROARCH.md "Two waves per SIMD" item 7, guide §5 "Read a staged buffer one phase AFTER the wait"):
//       RAW  a half-tile is read only after every wave's counted vmcnt that covers it AND a barrier both groups have passed:
//            the wait sits in phase 3 of K-step kt (before that phase's first barrier) and covers all four half-tiles of
//            K-step kt+1, whose first read is in phase 0 of kt+1 — two barrier instances later for either group.
//       WAR  half-tile h goes into the ring slot of half-tile h-8, whose last ds_read is in phase h-8 (or h-9); h is issued in
//            phase h-6, i.e. two phases (four barrier instances) after those reads were retired by lgkmcnt(0) in both groups.
//   * LDS image per half-tile: 128 rows x 128 B, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the 16-lane
//     groups of a ds_read_b128 fragment read hit 16 distinct 16-byte slots (conflict-free).  LDS-DMA writes are lane-linear,
//     so the permutation is applied to the per-lane SOURCE address.
//   * Rows beyond M / N are clamped to the last valid row at load time (their accumulators are never stored).
//   * split-K (o_proj / down_proj: N = hidden gives only 80 tiles at T ~ 1k): S workgroups share a tile, each takes a K range,
//     writes its fp32 partial tile in accumulator order (1-KiB coalesced WRITE-THROUGH stores), drains them and takes a ticket;
//     the last arriver acquires (agent scope), reads all S partial tiles back, adds them in slice order (deterministic) and runs
//     the epilogue.
//
// The weight fragment is the MFMA A operand (rows = n), the activation fragment the B operand (cols = m): same accumulator
// layout and register epilogue as gemm.hip (gemm_common.h).
#include <cstdlib>
#include <mutex>
#include <set>

#include "common.h"
#include "kernels.h"
#include "gemm_common.h"

namespace lmx {

namespace {

constexpr int P8_HALF = 128 * 128;         // bytes per half-tile buffer
constexpr int P8_LDS = 8 * P8_HALF;        // ring of 8 half-tiles
constexpr int P8_SLAB_FLOATS = 256 * 256;  // fp32 partial tile of one split-K slice

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

template <int N> struct IC { static constexpr int value = N; };

// buffer resource for raw (stride 0) 32-bit-offset addressing: base, num_records (bytes), gfx9-family dword 3
__device__ __forceinline__ v4i_t make_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// q|k|v epilogue with RoPE + KV-cache append (GemmArgs::qf_*).  The 256 x 256 tile covers whole heads of ONE of the three regions (the launcher checks the
// alignment).  It is rounded to T — the rounding point of the unfused path, where the GEMM stores T and rope_kv_kernel reads it back — into the LDS the
// operand ring no longer needs (256 x 264 elements), then leaves row-wise:
//   q / k tile: LDS image [row][col]; work item = (row, head, group of 8 rotation pairs): x1 = cols d .. d + 7, x2 = cols d + D/2 ..; cos / sin of the row's
//               position from the fp32 table, rounded to T; out = T( T(x cos) + T(rot(x) sin) ) as rope_kv_kernel; 16-byte stores into the q columns of C /
//               the K-cache row of the position.
//   v tile:     LDS image TRANSPOSED [col][row] (2-byte LDS writes: a lane owns one row), so that a work item (d, 8 consecutive positions) is one 16-byte
//               LDS read and one 16-byte store into V^T[kv head][d][pos ..] (pos0 and the row group are multiples of 8).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int QF_PITCH = 264;                               // elements per LDS row (528 B: 16-byte aligned rows, 4-bank shift per row)
constexpr int QF_LDS = 256 * QF_PITCH * 2;                  // bytes (16-bit T)

template <typename T>
__device__ __forceinline__ void qkv_rope_epilogue(const GemmArgs& a, f32x16 (&acc)[2][4], char* smem, int m0, int n0, int wm, int wn, int l31, int hi, int tid) {
    T* tile = reinterpret_cast<T*>(smem);
    const int D = a.qf_D, HALF = D >> 1;
    const int q_cols = a.qf_nh * D, k_cols = a.qf_nkv * D;
    const int region = n0 < q_cols ? 0 : (n0 < q_cols + k_cols ? 1 : 2);
    __syncthreads();                                        // every wave has left the operand ring
    if (region < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = wm * 128 + j * 32 + l31, col = wn * 64 + i * 32 + 8 * q + 4 * hi;
                    uint2 u;
                    u.x = pack2<T>(acc[i][j][4 * q], acc[i][j][4 * q + 1]); u.y = pack2<T>(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    *reinterpret_cast<uint2*>(tile + row * QF_PITCH + col) = u;
                }
        __syncthreads();
        const int GPR = HALF >> 3, HPT = 256 / D;            // groups of 8 pairs per head row, heads per tile
        // 256 x HPT x GPR = 4096 items, 8 per thread: item = tid + 512 it -> the same pair group g and head hh every time, rows 32 apart.  The cos / sin
        // values of four rows at a time are requested BEFORE the batch's first store (round 4): in the one-item-at-a-time loop every item's table loads waited behind
        // the previous item's stores (possible aliasing), eight dependent L2 round trips = ~10 us at the tail of every q|k|v workgroup.
        constexpr int NIT = 4;                               // items per batch (two batches: 64 registers of table values at a time)
        const int g = tid % GPR, hh = (tid / GPR) % HPT, i0 = g * 8;
        const int rstep = 512 / (GPR * HPT);                 // 32
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
        const int r0 = tid / (GPR * HPT) + half * NIT * rstep;
        f32x4 cs4[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int m = m0 + r0 + it * rstep; m = m < a.M ? m : a.M - 1;
            const float* cs = a.qf_rope + (size_t)(a.qf_pos0 + m) * D;
            cs4[it][0] = *reinterpret_cast<const f32x4*>(cs + i0); cs4[it][1] = *reinterpret_cast<const f32x4*>(cs + i0 + 4);
            cs4[it][2] = *reinterpret_cast<const f32x4*>(cs + HALF + i0); cs4[it][3] = *reinterpret_cast<const f32x4*>(cs + HALF + i0 + 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + it * rstep;
            const int m = m0 + r;
            if (m >= a.M) continue;
            const int pos = a.qf_pos0 + m;
            float x1[8], x2[8], o1[8], o2[8];
            load8<T>(tile + r * QF_PITCH + hh * D + i0, x1);
            load8<T>(tile + r * QF_PITCH + hh * D + HALF + i0, x2);
            const float cv[8] = {cs4[it][0].x, cs4[it][0].y, cs4[it][0].z, cs4[it][0].w, cs4[it][1].x, cs4[it][1].y, cs4[it][1].z, cs4[it][1].w};
            const float sv[8] = {cs4[it][2].x, cs4[it][2].y, cs4[it][2].z, cs4[it][2].w, cs4[it][3].x, cs4[it][3].y, cs4[it][3].z, cs4[it][3].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = round_to<T>(cv[e]), sn = round_to<T>(sv[e]);
                o1[e] = rope_term<T>(x1[e], c, -x2[e], sn);
                o2[e] = rope_term<T>(x2[e], c, x1[e], sn);
            }
            T* dst;
            if (region == 0) dst = reinterpret_cast<T*>(a.C) + (size_t)m * a.ldc + n0 + hh * D;
            else dst = reinterpret_cast<T*>(a.qf_kc) + ((size_t)((n0 - q_cols) / D + hh) * a.qf_smax + pos) * D;
            store8<T>(dst + i0, o1);
            store8<T>(dst + HALF + i0, o2);
        }
        }
    } else {
        uint16_t* tt = reinterpret_cast<uint16_t*>(smem);     // [col][row]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 128 + j * 32 + l31, col = wn * 64 + i * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
                    tt[col * QF_PITCH + row] = (uint16_t)(pack2<T>(acc[i][j][r], 0.f) & 0xffffu);      // the conversion gemm_epilogue's store4 uses
                }
        __syncthreads();
        T* vt = reinterpret_cast<T*>(a.qf_vt);
        const int kvh0 = (n0 - q_cols - k_cols) / D;
        for (int item = tid; item < 256 * 32; item += 512) {
            const int rg = item & 31, c = item >> 5;
            const int m = m0 + rg * 8;
            if (m >= a.M) continue;
            T* dst = vt + ((size_t)(kvh0 + c / D) * D + (c % D)) * a.qf_smax + a.qf_pos0 + m;
            const uint4 v = *reinterpret_cast<const uint4*>(tt + c * QF_PITCH + rg * 8);
            if (m + 8 <= a.M) *reinterpret_cast<uint4*>(dst) = v;
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {                   // the last, partial row group of the prompt
                    const uint32_t wk = k < 2 ? v.x : (k < 4 ? v.y : (k < 6 ? v.z : v.w));
                    if (m + k < a.M) reinterpret_cast<uint16_t*>(dst)[k] = (uint16_t)((k & 1) ? (wk >> 16) : (wk & 0xffffu));
                }
            }
        }
    }
}


// split_mode 7 (round 4): the fp32 partial tile leaves the workgroup in ROW-MAJOR order ([256][256] floats per (tile, slice) slab), turned through the LDS the
// operand ring no longer needs: two halves of 128 rows x 256 columns (pitch 260 floats, 133 KB), accumulator-order 16-byte LDS writes by the wave row that
// owns the half, then every wave streams 16 rows out as 1-KiB runs.  Costs the GEMM ~2 us; buys a reduction that can OWN ROWS with fully coalesced reads
// (splitk_reduce_rowmajor_kernel below): LlamaRMSNorm of the finished rows then needs nothing from another workgroup.
constexpr int RM_PITCH = 260;
static_assert(128 * RM_PITCH * 4 <= QF_LDS, "row-major slab staging must fit the LDS the launcher asks for");
__device__ __forceinline__ void store_slab_rowmajor(const GemmArgs& a, f32x16 (&acc)[2][4], char* smem, float* slab, int m0, int wm, int wn, int l31, int hi, int tid) {
    float* sums = reinterpret_cast<float*>(smem);
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                    // half 0: every wave has left the operand ring; half 1: half 0 has been read out
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(sums + (j * 32 + l31) * RM_PITCH + wn * 64 + i * 32 + 8 * q + 4 * hi) =
                            f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = wave * 16 + it;                   // row of the half; a wave-instruction moves one whole 1-KiB row
            if (m0 + half * 128 + r < a.M)                  // rows of padding are never read back
                *reinterpret_cast<f32x4*>(slab + (size_t)(half * 128 + r) * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(sums + r * RM_PITCH + lane * 4);
        }
    }
}

// PRIO / STAGGER: the two levers of the schedule, kept as template arms for the microbenchmarks (profiles/EXPERIMENTS.md: without
// s_setprio 830 TF, groups in lock-step 838 TF, both 1016 TF on the q|k|v shape).  SPLIT: K slices per tile (1, 2 or 3).
// INLAUNCH (K-sliced instantiations): false = the launch-boundary forms only (split_mode 5 / 7: store the slab and exit) — the in-launch reduction's loads and
// sums are not compiled in, which keeps the K-sliced kernel free of scratch (with them: 532 bytes of spills per lane at SPLIT 3).
template <typename T, bool PRIO, bool STAGGER, int SPLIT, bool INLAUNCH = true>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const bool group1 = STAGGER && wave >= 4;

    constexpr int S = SPLIT;                          // K slices per tile (compile time: 1, 2 or 3)
    const int mtiles = (a.M + 255) >> 8;
    const int ntiles = (a.N + 255) >> 8;
    int tile, slice, tile_n, tile_m, s_eff = S;       // tile: index of the partial-tile slabs / arrival counter of a K-sliced tile
    bool hybrid = false;
    if constexpr (SPLIT == 2) hybrid = a.hyb_unsplit > 0;
    int mhalf = -1;                                   // M-tail order: 0 / 1 = this item is the upper / lower 128 rows of a full tile, -1 = a whole tile
    bool mtail = false;
    if constexpr (SPLIT == 1) mtail = a.mt_whole > 0;
    if (mtail) {
        // M-TAIL order (more full tiles than CUs, un-split launch: 7B gate|up at 1087 rows = 344 full + 86 ragged tiles on 256 CUs, i.e. a second round that
        // two thirds of the chip sit out).  Every XCD owns a contiguous range of N-tiles and walks, in dispatch order: its first `mt_whole` full tiles
        // whole, the remaining full tiles as TWO 128-ROW HALVES each, and last the ragged M-tiles.  A half item keeps the 8-wave ping-pong: wave row wm takes
        // rows 64 wm .. 64 wm + 63 of the half (its blocks j = 0, 1, i.e. X half 0 only), so phases 0 and 1 of every K-step run as in a whole tile and
        // phases 2 and 3 are barriers only.  No partial sums leave the workgroup (unlike the K-halves of the tail-split order above): every output element
        // is accumulated over K in the same order as in a whole tile, so the result is bit-identical to the plain order.
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int n_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, n_cnt = q + (xcd < r ? 1 : 0);
        const int mf = (a.M & 255) ? mtiles - 1 : mtiles;          // full M-tiles
        const int full = n_cnt * mf, rag = n_cnt * (mtiles - mf);
        const int whole = full < a.mt_whole ? full : a.mt_whole;
        const int nsplit = full - whole;
        int f;
        if (idx < whole) f = idx;
        else if (idx < whole + 2 * nsplit) { const int j = idx - whole; f = whole + (j >> 1); mhalf = j & 1; }
        else {
            const int rr = idx - whole - 2 * nsplit;
            if (rr >= rag) return;                                // this XCD has fewer items than the widest one
            f = -1; tile_n = n_lo + rr; tile_m = mf;
        }
        if (f >= 0) { tile_n = n_lo + f / mf; tile_m = f - (f / mf) * mf; }
        tile = 0; slice = 0;
    } else if (!hybrid) {
        const int lid = xcd_remap(blockIdx.x, mtiles * ntiles * S);
        tile = lid / S; slice = lid - tile * S;
        tile_n = tile / mtiles; tile_m = tile - tile_n * mtiles;
    } else {
        // TAIL-SPLIT order (more full tiles than CUs: the second round would run half empty).  Every XCD owns a contiguous range of N-tiles and walks,
        // in dispatch order: its first `hyb_unsplit` (= CUs of an XCD) full M-tiles whole, further full tiles whole until only `hyb_split` are left,
        // those as TWO K-halves each (fp32 partial tiles + in-launch reduction, as split-K), and last the cheap ragged M-tiles — so the tail of the
        // launch is made of half-length and quarter-filled items instead of whole tiles on a third of the CUs.
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int n_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, n_cnt = q + (xcd < r ? 1 : 0);
        const int mf = (a.M & 255) ? mtiles - 1 : mtiles;          // full M-tiles
        const int full = n_cnt * mf, rag = n_cnt * (mtiles - mf);
        const int nsplit = full - a.hyb_unsplit < a.hyb_split ? (full - a.hyb_unsplit > 0 ? full - a.hyb_unsplit : 0) : a.hyb_split;
        const int whole = full - nsplit;
        int f;
        if (idx < whole) { f = idx; slice = 0; s_eff = 1; tile = 0; }
        else if (idx < whole + 2 * nsplit) { const int j = idx - whole; f = whole + (j >> 1); slice = j & 1; s_eff = 2; tile = xcd * a.hyb_split + (j >> 1); }
        else {
            const int rr = idx - whole - 2 * nsplit;
            if (rr >= rag) return;                                // this XCD has fewer items than the widest one
            f = -1; slice = 0; s_eff = 1; tile = 0;
            tile_n = n_lo + rr; tile_m = mf;
        }
        if (f >= 0) { tile_n = n_lo + f / mf; tile_m = f - (f / mf) * mf; }
    }
    const int m0 = tile_m << 8, n0 = tile_n << 8;
    const int nk_all = a.K >> 6;
    const int kt_begin = (int)((long)nk_all * slice / s_eff), kt_end = (int)((long)nk_all * (slice + 1) / s_eff);
    const int nk = kt_end - kt_begin;                 // K-steps of this workgroup (>= 1: the launcher keeps S <= nk_all)

    // ---- LDS-DMA sources ---------------------------------------------------------------------------------------------------
    // piece i (0,1) of a half-tile covers buffer rows 8*wave + 64*i .. +7; lane l: row += l >> 3, physical chunk l & 7,
    // logical chunk (l & 7) ^ ((row >> 1) & 7).  X half mh: buffer row r <-> tile row (r >> 6) * 128 + mh * 64 + (r & 63);
    // W half nh: buffer row r <-> tile column (r >> 5) * 64 + nh * 32 + (r & 31).
    const v4i_t rsX = make_rsrc(a.X, (uint32_t)((size_t)a.M * a.ldx * sizeof(T)));
    const v4i_t rsW = make_rsrc(a.W, (uint32_t)((size_t)a.N * a.ldw * sizeof(T)));
    uint32_t voX[2][2], voW[2][2];                    // [piece][half] byte offsets of this lane's 16 bytes at k = 0
    {
        const int rb = 8 * wave + (lane >> 3);                        // buffer row of piece 0 (piece 1: + 64)
        const int chunk = ((lane & 7) ^ (rb >> 1)) & 7;               // (rb + 64) >> 1 has the same low 3 bits
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int m = m0 + i * 128 + h * 64 + rb;                   // rb < 64
                if (mhalf >= 0) m = m0 + mhalf * 128 + (h == 0 ? i * 64 + rb : 0);      // half item: X half 0 = its 128 rows, X half 1 is never multiplied
                m = m < a.M ? m : a.M - 1;
                voX[i][h] = (uint32_t)(((size_t)m * a.ldx + chunk * 8) * sizeof(T));
                const int r = rb + 64 * i;
                int n = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
                n = n < a.N ? n : a.N - 1;
                voW[i][h] = (uint32_t)(((size_t)n * a.ldw + chunk * 8) * sizeof(T));
            }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);

    // half-tile sequence: index 4*kt + j, j = 0: X half 0, 1: W half 0, 2: W half 1, 3: X half 1; ring slot = (kt & 1) * 4 + j.
    // Issued from inline asm so hipcc neither sees an LDS write (it would drain vmcnt(0) before the next ds_read) nor counts these
    // loads; M0 = wave-uniform LDS destination, lane l lands at M0 + 16 l.  M0 is compiler-reserved: saved / restored in the statement.
    auto stage = [&](int kt_abs, int j, int slot) {
        const unsigned d0 = lds_wave + slot * P8_HALF, d1 = d0 + 8192;
        const unsigned so = (unsigned)kt_abs * 128u;                 // 64 k * 2 B per K-step
        const bool isx = (j == 0) || (j == 3);
        const int h = (j == 0 || j == 1) ? 0 : 1;
        const uint32_t v0 = isx ? voX[0][h] : voW[0][h], v1 = isx ? voX[1][h] : voW[1][h];
        unsigned keep;
        if (isx)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                         "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(d0), "s"(d1), "v"(v0), "v"(v1), "s"(rsX), "s"(so) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                         "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(d0), "s"(d1), "v"(v0), "v"(v1), "s"(rsW), "s"(so) : "memory");
    };
    // leave at most `halves` half-tiles (2 LDS-DMA instructions each) of this wave in flight
    auto wait_halves = [&](int halves) {
        if (halves >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (halves == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (halves == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (halves == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- fragment read offsets -------------------------------------------------------------------------------------------------
    // X fragment (half, jj, ks): buffer row wm*64 + jj*32 + l31; W fragment (half, ks): buffer row wn*32 + l31; 16-byte chunk
    // 2 ks + hi, stored at chunk ^ ((row >> 1) & 7) — the low bits of row >> 1 depend on l31 only.
    int xo[4], wo[4];
    {
        const int sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ((2 * ks + hi) ^ sw) << 4;
            xo[ks] = (wm * 64 + l31) * 128 + c;
            wo[ks] = (wn * 32 + l31) * 128 + c;
        }
    }

    f32x16 acc[2][4];                         // [n tile i][m tile j], as gemm_epilogue expects
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 xa[2][4];                           // X fragments of the current M half: [jj][ks]
    uint4 wb0[4], wb1[4];                     // W fragments of N half 0 / 1: [ks]

    auto read_x = [&](int slot) {
        const char* b = smem + slot * P8_HALF;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xa[jj][ks] = *reinterpret_cast<const uint4*>(b + jj * 4096 + xo[ks]);
    };
    auto read_w = [&](int slot, uint4 (&wb)[4]) {
        const char* b = smem + slot * P8_HALF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wb[ks] = *reinterpret_cast<const uint4*>(b + wo[ks]);
    };
    // ---- prologue: six half-tiles in flight, K-step 0 landed ----------------------------------------------------------------------
    const int n_half = 4 * nk;
#pragma unroll
    for (int h = 0; h < 6; ++h)
        if (h < n_half && ((h & 3) != 3 || a.M - m0 > 64 || a.no_skip)) stage(kt_begin + (h >> 2), h & 3, ((h >> 2) & 1) * 4 + (h & 3));
    wait_halves(nk >= 2 ? 2 : 0);
    __builtin_amdgcn_s_barrier();
    if (group1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one barrier behind group 0

    // Ragged M: the last M tile of a prompt is mostly padding (T = 1087: 63 valid rows of 256).  In such a tile a wave skips the MFMAs of
    // 32-row blocks that lie entirely beyond M (wave-uniform scalar branches; barriers and loads unchanged): the tile finishes early, which
    // relieves the tile-count quantisation (430 tiles of gate|up on 256 CUs) and stops burning power on rows that are never stored
    // (in-model A/B: q|k|v 126 -> 118 us, gate|up 221 -> 210 us).  Full tiles run the SAME loop compiled without the branches — folding
    // the checks into one loop cost the full tiles 15 % (register pressure 218 -> 250, accumulator copies) — so the loop is instantiated
    // twice and chosen once per workgroup.
    // A ragged tile with at most 64 valid rows (T = 1087: the 63 rows of the fifth M-tile) never multiplies X half 1 (tile rows 64..127, 192..255): its ragged loop
    // neither stages nor reads that half, so phases 2 and 3 of its K-steps shrink to their barriers (round 4: such a tile cost 0.9 of a full one because every
    // interval lasts as long as its load side, EXPERIMENTS.md r3-M; the counted waits stay valid — fewer loads are in flight than they allow for).
    const bool x1_needed = mhalf >= 0 || a.M - m0 > 64;
    const int mw = mhalf < 0 ? m0 + wm * 128 : m0 + mhalf * 128 + wm * 64;      // first row of this wave's 128-row block (half item: of its two 32-row blocks)
    const bool live[4] = {mw < a.M, mw + 32 < a.M, mhalf < 0 && mw + 64 < a.M, mhalf < 0 && mw + 96 < a.M};

    auto main_loop = [&](auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value != 0;
        constexpr bool SKIPX1 = decltype(ragged_c)::value == 2;       // ragged tile with <= 64 valid rows: X half 1 is neither staged, read nor multiplied
        auto mma = [&](const uint4 (&wb)[4], int nh, int mh) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            if (!RAGGED || live[2 * mh + 1]) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[nh][2 * mh] = Mfma32x32x16<T>::run(wb[ks], xa[0][ks], acc[nh][2 * mh]);
                    acc[nh][2 * mh + 1] = Mfma32x32x16<T>::run(wb[ks], xa[1][ks], acc[nh][2 * mh + 1]);
                }
            } else if (live[2 * mh]) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[nh][2 * mh] = Mfma32x32x16<T>::run(wb[ks], xa[0][ks], acc[nh][2 * mh]);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        };
        // one K-step = 4 phases; PAR = ring half (kt & 1), known at compile time so every LDS offset is an immediate.
        // Half-tile issued in phase p of K-step kt: sequence index 4 kt + p + 6 -> K-step kt+1 (j = p+2) for p < 2, K-step kt+2 (j = p-2) else.
        auto kstep = [&](auto par_c, int kt) {
            constexpr int PAR = decltype(par_c)::value;
            constexpr int SX0 = PAR * 4 + 0, SW0 = PAR * 4 + 1, SW1 = PAR * 4 + 2, SX1 = PAR * 4 + 3;
            // phase 0 -----------------------------------------------------------------------------------------------------------
            read_w(SW0, wb0);
            __builtin_amdgcn_sched_barrier(0);
            read_x(SX0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) stage(kt_begin + kt + 1, 2, (1 - PAR) * 4 + 2);
            __builtin_amdgcn_s_barrier();
            mma(wb0, 0, 0);
            __builtin_amdgcn_s_barrier();
            // phase 1 -----------------------------------------------------------------------------------------------------------
            read_w(SW1, wb1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!SKIPX1) { if (kt + 1 < nk) stage(kt_begin + kt + 1, 3, (1 - PAR) * 4 + 3); }
            __builtin_amdgcn_s_barrier();
            mma(wb1, 1, 0);
            __builtin_amdgcn_s_barrier();
            // phase 2 -----------------------------------------------------------------------------------------------------------
            if constexpr (!SKIPX1) read_x(SX1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < nk) stage(kt_begin + kt + 2, 0, PAR * 4 + 0);
            __builtin_amdgcn_s_barrier();
            if constexpr (!SKIPX1) mma(wb1, 1, 1);
            __builtin_amdgcn_s_barrier();
            // phase 3 -----------------------------------------------------------------------------------------------------------
            if (kt + 2 < nk) stage(kt_begin + kt + 2, 1, PAR * 4 + 1);
            // every half-tile of K-step kt+1 must have landed; the two of kt+2 issued in this K-step may stay in flight
            wait_halves(kt + 2 < nk ? 2 : 0);
            __builtin_amdgcn_s_barrier();
            if constexpr (!SKIPX1) mma(wb0, 0, 1);
            __builtin_amdgcn_s_barrier();
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            kstep(IC<0>{}, kt);
            kstep(IC<1>{}, kt + 1);
        }
        if (kt < nk) kstep(IC<0>{}, kt);
    };
    if (mhalf < 0 && (m0 + 256 <= a.M || a.no_skip)) main_loop(IC<0>{});
    else if (x1_needed) main_loop(IC<1>{});
    else main_loop(IC<2>{});
    if (STAGGER && !group1) __builtin_amdgcn_s_barrier();   // balance the stagger barrier

    const int m_base = mw, n_base = n0 + wn * 64;
    if (mhalf >= 0) a.M = mw + 64;                    // half item: accumulator blocks j = 2, 3 hold nothing (their rows belong to the other wave row)
    bool k_sliced = SPLIT > 1;
    if constexpr (SPLIT == 2) k_sliced = s_eff > 1;
    if constexpr (SPLIT == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.qf_kc) { qkv_rope_epilogue<T>(a, acc, smem, m0, n0, wm, wn, l31, hi, tid); return; }
    } else if (!k_sliced) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ---- split-K: publish the partial tile, last arriver reduces (guide §5 "in-launch split-K reduction") -------------------
        // accumulator order: float4 number (i*4 + j)*4 + q of lane `tid` lives at byte ((idx * 512) + tid) * 16 of the slab, so every
        // wave-instruction moves 1 KiB of contiguous memory.  The stores are WRITE-THROUGH (sc1: the bytes leave the XCD's L2 as they are
        // written), so publishing needs no L2 write-back fence afterwards (MI355X_MICROARCH.md "publish-large": 3.0 vs 8.2 us per 64 KiB
        // of partials per workgroup); the reducer takes ONE agent-scope acquire (stale lines of earlier launches) and reads with sc1 loads.
        if (a.split_mode == 7) {                          // row-major slabs for the row-owning reduction (fused RMSNorm)
            store_slab_rowmajor(a, acc, smem, static_cast<float*>(a.skw) + ((size_t)tile * S + slice) * P8_SLAB_FLOATS, m0, wm, wn, l31, hi, tid);
            return;
        }
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
        const uint32_t slab_off = (uint32_t)(((size_t)tile * S + slice) * (P8_SLAB_FLOATS * sizeof(float)));   // < 2^31: <= 256 slabs of 256 KiB
        const uint32_t lane_off = (uint32_t)tid * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // a 32-row block that lies entirely beyond M is never read back by the launch-boundary reduction (the last M tile of a 1087-row prompt
                // is 7/8 padding: 10.7 of the 61 MB of down_proj's slabs)
                if (a.split_mode == 5 && m_base + j * 32 >= a.M) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = (i * 4 + j) * 4 + q;
                    v4u_t v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                               __float_as_uint(acc[i][j][4 * q + 3])};
                    // split_mode 1 (shipping): sc1 write-through stores + agent release; the reducer takes an agent acquire and reads with sc1
                    //   loads.  Measured under uneven load (a second stream sharing the chip, 96 launches, tools/sessions/gpu_r2e.sh):
                    //   0: plain stores + release, acquire + PLAIN loads      44 of 96 launches wrong (stale partial tiles of earlier launches)
                    //   1: sc1 stores + release, acquire + sc1 loads           0 wrong, 3 % faster than 0
                    //   2: sc0 sc1 stores, otherwise as 1                       0 wrong, same time as 1
                    //   3: sc1 stores, NO release, acquire + sc1 loads         wrong (the ticket overtakes the write-through)
                    if (a.split_mode == 0 || a.split_mode == 5) __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, lane_off + idx * 8192, slab_off, 0);
                    else if (a.split_mode == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, lane_off + idx * 8192, slab_off, 17);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, lane_off + idx * 8192, slab_off, /*sc1*/ 16);
                }
            }
        if (a.split_mode == 5) return;                 // launch-boundary reduction: splitk_reduce_kernel sums the slabs and runs the epilogue
        if constexpr (!INLAUNCH) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);      // the one LDS array doubles as the broadcast word (ring is dead now)
        if (tid == 0) {
            if (a.split_mode != 3) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const int t = __hip_atomic_fetch_add(a.skc + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == S - 1);
            if (last) {
                __hip_atomic_store(a.skc + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
                // the scratch is reused launch after launch: this CU's L1 / this XCD's L2 may still hold the PREVIOUS launch's partial
                // tiles of these addresses (measured: sc1 loads alone return them) — one agent-scope acquire before the reads
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        // sum the S partial tiles in slice order (all read back from the scratch, this workgroup's own included, so the result does
        // not depend on which slice arrived last); two accumulators at a time keeps <= 3 x 32 registers of loads in flight
        const uint32_t tile_off = (uint32_t)((size_t)tile * S * (P8_SLAB_FLOATS * sizeof(float)));
        constexpr uint32_t SLAB_BYTES = P8_SLAB_FLOATS * sizeof(float);
        auto ld = [&](uint32_t off) {
            const v4u_t u = a.split_mode == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rs_slab, lane_off + off, tile_off, 0)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rs_slab, lane_off + off, tile_off, /*sc1*/ 16);
            return f32x4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
        };
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v0[8], v1[8], v2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v0[e] = ld((g * 8 + e) * 8192);
                v1[e] = ld(SLAB_BYTES + (g * 8 + e) * 8192);
            }
            if constexpr (S > 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v2[e] = ld(2 * SLAB_BYTES + (g * 8 + e) * 8192);
            }
            // fresh accumulator vectors (an element-wise update of the old ones would keep all 128 dead registers live)
            f32x16 r0, r1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f32x4 t = v0[e] + v1[e];
                if constexpr (S > 2) t += v2[e];
                const int q = e & 3;
                if (e < 4) { r0[4 * q] = t.x; r0[4 * q + 1] = t.y; r0[4 * q + 2] = t.z; r0[4 * q + 3] = t.w; }
                else       { r1[4 * q] = t.x; r1[4 * q + 1] = t.y; r1[4 * q + 2] = t.z; r1[4 * q + 3] = t.w; }
            }
            acc[g >> 1][(g & 1) * 2] = r0;            // float4 index (i*4 + j)*4 + q = g*8 + e  ->  i = g >> 1, j = (g & 1)*2 + (e >> 2)
            acc[g >> 1][(g & 1) * 2 + 1] = r1;
            __builtin_amdgcn_sched_barrier(0);        // keep the next group's loads behind this group's sums (no spills)
        }
    }
    gemm_epilogue<T, 4, 2>(a, acc, m_base, n_base, l31, hi);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Launch-boundary split-K reduction (default for K-sliced launches; LMX_SPLITK_MODE=1 keeps the in-launch reduction by the last arriver).
// In the in-launch form ONE workgroup per tile reads S x 256 KiB of partials back through one CU (~10 us at the ~60-100 GB/s a single CU pulls) after an
// agent-scope release / ticket / acquire: ~45 us of the 142 us down_proj launch.  Here the GEMM workgroups store their slabs (plain stores: the kernel
// boundary publishes them) and exit; this kernel spreads the same sums over 8 workgroups per tile — workgroup (tile, wm, j) owns the 32 x 256 block that
// accumulator column j of the GEMM's wave row wm held, so thread t re-reads exactly the float4s GEMM thread wm * 256 + t wrote (4 KiB contiguous per
// wave-instruction), adds them in slice order (deterministic, the same order as the in-launch form) and runs the shared epilogue (bias / activation /
// residual, packed stores) on its 2 x 16 values.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int S>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wn = tid >> 6;
    const int mtiles = (a.M + 255) >> 8;
    const int blk = blockIdx.x;
    const int tile = blk >> 3, wm = (blk >> 2) & 1, j = blk & 3;
    const int tile_n = tile / mtiles, tile_m = tile - tile_n * mtiles;
    const int m_base = (tile_m << 8) + wm * 128 + j * 32, n_base = (tile_n << 8) + wn * 64;
    if (m_base >= a.M) return;                                             // a row block of padding
    const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
    constexpr uint32_t SLAB_BYTES = P8_SLAB_FLOATS * sizeof(float);
    const uint32_t tile_off = (uint32_t)((size_t)tile * S * SLAB_BYTES);
    const uint32_t lane_off = (uint32_t)(wm * 256 + tid) * 16u;
    // every load of this thread goes out before the first sum (2 x 4 float4 per slice: 96 registers at S = 3): ONE round trip to the slabs instead of two
    v4u_t w[2][S][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < S; ++sl)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[i][sl][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, lane_off + (uint32_t)(((i * 4 + j) * 4 + q) * 8192) + sl * SLAB_BYTES, tile_off, 0);
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t = f32x4{__uint_as_float(w[i][0][q].x), __uint_as_float(w[i][0][q].y), __uint_as_float(w[i][0][q].z), __uint_as_float(w[i][0][q].w)};
#pragma unroll
            for (int sl = 1; sl < S; ++sl)
                t += f32x4{__uint_as_float(w[i][sl][q].x), __uint_as_float(w[i][sl][q].y), __uint_as_float(w[i][sl][q].z), __uint_as_float(w[i][sl][q].w)};
            acc[i][0][4 * q] = t.x; acc[i][0][4 * q + 1] = t.y; acc[i][0][4 * q + 2] = t.z; acc[i][0][4 * q + 3] = t.w;
        }
    gemm_epilogue<T, 1, 2>(a, acc, m_base, n_base, l31, hi);
}

// Same reduction with LlamaRMSNorm of the output rows fused in (o_proj / down_proj of a prefill are followed by the next block's RMSNorm: one launch per
// norm, 64 per 7B prefill).  A workgroup owns FOUR consecutive rows (the norm needs whole rows); lane = 4 u + r reads, for row r, the float4 piece u of
// every 256-column tile — in the accumulator-ordered slabs the same piece of 4 consecutive rows is one 64-byte sector, so a wave-instruction moves 16 full
// sectors.  (One row per workgroup with rmsnorm_kernel's mapping, 16 bytes of every sector, measured +45 us per launch.)  The row's sum of squares is
// reduced over the 64 threads of the row (shuffles over the lane's u bits, then the 4 waves through LDS): a fixed order, but not rmsnorm_kernel's — the
// normalised row can differ from the unfused sequence in the last bit of a few elements (tests/test_gemm8p_gpu.py bounds it).
template <typename T, int S>
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(GemmArgs a) {
    __shared__ float red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 3, u = tid >> 2;                                  // row within the workgroup, float4 piece within a 256-column tile (0 .. 63)
    const int m = blockIdx.x * 4 + r;
    const bool live = m < a.M;
    const int mc = live ? m : a.M - 1;
    const int mtiles = (a.M + 255) >> 8, ntiles = (a.N + 255) >> 8;
    const int tile_m = mc >> 8, wm = (mc & 255) >> 7, j = (mc & 127) >> 5, l31 = mc & 31;
    const int wn = u >> 4, i = (u >> 3) & 1, q = (u >> 1) & 3, hi = u & 1;
    const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
    constexpr uint32_t SLAB_BYTES = P8_SLAB_FLOATS * sizeof(float);
    const uint32_t in_tile = (uint32_t)(((i * 4 + j) * 4 + q) * 8192) + (uint32_t)(((wm * 4 + wn) * 64 + hi * 32 + l31) * 16);
    T* __restrict__ C = reinterpret_cast<T*>(a.C) + (size_t)mc * a.ldc;
    const T* bias = reinterpret_cast<const T*>(a.bias);
    const T* R = a.R ? reinterpret_cast<const T*>(a.R) + (size_t)mc * a.ldr : nullptr;
    constexpr int MAXT = 32;                                              // 256-column tiles per row: N <= 8192
    constexpr int TG = 4;                                                 // tiles per batch: TG x S loads are issued before the first sum (one round trip per batch)
    float hv[MAXT][4];
    float ss = 0.f;
#pragma unroll
    for (int t0 = 0; t0 < MAXT; t0 += TG) {
        if (t0 < ntiles) {                                                // wave-uniform
            v4u_t w[TG][S];
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                const int t = t0 + tt < ntiles ? t0 + tt : ntiles - 1;     // past the last tile: re-read it (masked below)
                const uint32_t base = (uint32_t)(t * mtiles + tile_m) * (uint32_t)S * SLAB_BYTES + in_tile;
#pragma unroll
                for (int sl = 0; sl < S; ++sl) w[tt][sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, base + sl * SLAB_BYTES, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                const int t = t0 + tt;
                const int n = t * 256 + u * 4;
                f32x4 acc4 = f32x4{__uint_as_float(w[tt][0].x), __uint_as_float(w[tt][0].y), __uint_as_float(w[tt][0].z), __uint_as_float(w[tt][0].w)};
#pragma unroll
                for (int sl = 1; sl < S; ++sl) acc4 += f32x4{__uint_as_float(w[tt][sl].x), __uint_as_float(w[tt][sl].y), __uint_as_float(w[tt][sl].z), __uint_as_float(w[tt][sl].w)};
                float v[4] = {acc4.x, acc4.y, acc4.z, acc4.w};
                if (t < ntiles && n < a.N) {
                    if (bias) { float bb[4]; load4<T>(bias + n, bb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bb[e]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                    if (R) { float rr[4]; load4<T>(R + n, rr);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rr[e]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[t][e] = round_to<T>(v[e]);
                    if (live) store4<T>(C + n, hv[t]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ss += hv[t][e] * hv[t][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[t][e] = 0.f;
                }
            }
        }
    }
    // sum of squares of row r: over the 16 lanes of this wave that share r (lane bits 2..5), then over the 4 waves in order
    ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64); ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
    if (lane < 4) red[wave][lane] = ss;
    __syncthreads();
    const float tot = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
    const float inv = rsqrtf(tot / (float)a.N + a.norm_eps);
    if (!live) return;
    const T* g = reinterpret_cast<const T*>(a.norm_w);
    T* __restrict__ Y = reinterpret_cast<T*>(a.norm_out) + (size_t)m * a.ld_norm;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < ntiles) {
            const int n = t * 256 + u * 4;
            if (n < a.N) {
                float gv[4], y[4]; load4<T>(g + n, gv);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = round_to<T>(hv[t][e] * inv) * gv[e];
                store4<T>(Y + n, y);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
size_t gemm8p_splitk_ws_bytes(int M, int N, int split_k) {
    if (split_k <= 1) return 0;
    const size_t tiles = (size_t)cdiv(M, 256) * cdiv(N, 256);
    return tiles * split_k * P8_SLAB_FLOATS * sizeof(float);
}
size_t gemm8p_splitk_counter_bytes(int M, int N) { return (size_t)cdiv(M, 256) * cdiv(N, 256) * sizeof(int); }

// K slices per tile for the ping-pong kernel: fill the 256 CUs when N = hidden gives too few 256x256 tiles, but keep every slice
// long enough (>= 20 K-steps with the launch-boundary reduction; the in-launch reduction needed >= 48: o_proj at K = 4096 lost to the
// 128x128 kernel) that the fp32 partial-tile round trip (S x 256 KiB written and read per tile) stays small against its main loop.
// would an un-split launch of this shape take the tail-split order (LMX_GEMM8P_TAIL=1)?  Same rule as launch_gemm8p_t.
bool gemm8p_tail_split_applies(int M, int N, int K) {
    static const int tail = [] { const char* e = getenv("LMX_GEMM8P_TAIL"); return e ? atoi(e) : 0; }();
    if (!tail) return false;
    static const int cus_x = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 > 0 ? n / 8 : 32; }();
    int cus = cus_x;
    if (const char* e = getenv("LMX_GEMM8P_TAIL_CUS")) { const int v = atoi(e); if (v >= 1 && v <= 64) cus = v; }
    const int mt = cdiv(M, 256), nt = cdiv(N, 256), mf = (M % 256) ? mt - 1 : mt;
    return mf >= 1 && nt >= 8 && ((nt + 7) / 8) * mf > cus && K / 64 >= 32;
}
bool gemm8p_boundary_reduce() { static const int mode = [] { const char* e = getenv("LMX_SPLITK_MODE"); return e ? atoi(e) : 5; }(); return mode == 5; }
int gemm8p_pick_split(int M, int N, int K) {
    const int tiles = cdiv(M, 256) * cdiv(N, 256);
    const int nk = K / 64;
    if (tiles >= 160) return 1;
    int s = 256 / tiles;
    if (s > 3) s = 3;
    static const int min_steps = [] { const char* e = getenv("LMX_SPLITK_MIN_STEPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 20; }();
    while (s > 1 && nk / s < min_steps) --s;
    return s < 1 ? 1 : s;
}

namespace {
// fallback split-K scratch for callers that bring none (lmx_op_gemm: unit tests, microbenchmarks; ONE stream at a time).
// The engine passes per-sequence scratch instead (Model::prefill), so concurrent request threads never share it.
struct FallbackWs {
    std::mutex mu;
    void* ws = nullptr; size_t ws_bytes = 0;
    int* cnt = nullptr; size_t cnt_bytes = 0;
};
FallbackWs g_fb;
}  // namespace

// splitk_reduce_rows_kernel: the same sums, with the output side turned into ROW order through LDS.  In accumulator order a lane owns 4 consecutive n of one row
// per float4, so a store instruction of gemm_epilogue scatters 16-byte pieces over 32 rows (and the residual loads gather the same way): the 18 us of
// splitk_reduce_kernel were mostly that.  Here the 32 x 256 block of sums goes to LDS (pitch 260 floats: conflict-free 16-byte writes) and is read back as
// rows: 8 consecutive lanes cover 32 consecutive n, so every residual load / output store instruction moves 64-byte runs and the wave finishes whole
// 512-byte rows; the residual is requested before the slab loads.  Epilogue arithmetic = gemm_epilogue's (bias, activation, residual, one rounding).
template <typename T, int S>
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(GemmArgs a) {
    constexpr int PITCH = 260;
    __shared__ __attribute__((aligned(16))) float sums[32 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wn = tid >> 6;
    const int mtiles = (a.M + 255) >> 8;
    const int blk = blockIdx.x;
    const int tile = blk >> 3, wm = (blk >> 2) & 1, j = blk & 3;
    const int tile_n = tile / mtiles, tile_m = tile - tile_n * mtiles;
    const int m_base = (tile_m << 8) + wm * 128 + j * 32, n0 = tile_n << 8;
    if (m_base >= a.M) return;                                             // a row block of padding
    // output side: row r_out = tid >> 3 of the block, float4 columns (tid & 7) + 8 k
    const int r_out = tid >> 3, seg = tid & 7;
    const int m_out = m_base + r_out;
    const T* R = reinterpret_cast<const T*>(a.R);
    const T* bias = reinterpret_cast<const T*>(a.bias);
    uint2 rr[8];
    if (R && m_out < a.M) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int n = n0 + (seg + 8 * k) * 4;
            rr[k] = n < a.N ? *reinterpret_cast<const uint2*>(R + (size_t)m_out * a.ldr + n) : uint2{0u, 0u};
        }
    }
    const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
    constexpr uint32_t SLAB_BYTES = P8_SLAB_FLOATS * sizeof(float);
    const uint32_t tile_off = (uint32_t)((size_t)tile * S * SLAB_BYTES);
    const uint32_t lane_off = (uint32_t)(wm * 256 + tid) * 16u;
    v4u_t w[2][S][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < S; ++sl)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[i][sl][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, lane_off + (uint32_t)(((i * 4 + j) * 4 + q) * 8192) + sl * SLAB_BYTES, tile_off, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t = f32x4{__uint_as_float(w[i][0][q].x), __uint_as_float(w[i][0][q].y), __uint_as_float(w[i][0][q].z), __uint_as_float(w[i][0][q].w)};
#pragma unroll
            for (int sl = 1; sl < S; ++sl)
                t += f32x4{__uint_as_float(w[i][sl][q].x), __uint_as_float(w[i][sl][q].y), __uint_as_float(w[i][sl][q].z), __uint_as_float(w[i][sl][q].w)};
            // accumulator (i, q) of lane (l31, hi) of wave wn: row l31, columns wn * 64 + i * 32 + 8 q + 4 hi .. + 3
            *reinterpret_cast<f32x4*>(sums + l31 * PITCH + wn * 64 + i * 32 + 8 * q + 4 * hi) = t;
        }
    __syncthreads();
    if (m_out >= a.M) return;
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const int act = a.act;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = (seg + 8 * k) * 4, n = n0 + c;
        if (n >= a.N) continue;
        const f32x4 t = *reinterpret_cast<const f32x4*>(sums + r_out * PITCH + c);
        float v[4] = {t.x, t.y, t.z, t.w};
        if (bias) {
            float b[4]; load4<T>(bias + n, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        if (R) {
            v[0] += unpack_lo<T>(rr[k].x); v[1] += unpack_hi<T>(rr[k].x); v[2] += unpack_lo<T>(rr[k].y); v[3] += unpack_hi<T>(rr[k].y);
        }
        store4<T>(C + (size_t)m_out * a.ldc + n, v);
    }
}


// splitk_reduce_rows_norm_kernel (round 4): splitk_reduce_rows_kernel + LlamaRMSNorm of the rows it produces, WITHOUT giving up the tile-shaped reduction (the
// row-owning splitk_reduce_norm_kernel gathers 64-byte sectors and lost more than the 63 rmsnorm launches of a 7B prefill cost: EXPERIMENTS.md r3-E).  A
// workgroup reduces its 32 x 256 block exactly as before, stores the residual-stream rows, and keeps the T-rounded values in registers.  What the norm needs
// from elsewhere is one float per row and N-tile: the block's partial sums of squares go to `norm_part` as 8-byte {value, launch tag} granules ([M][N tiles],
// relaxed agent-scope atomic stores), the row's first lane polls the ntiles granules of its row until all carry this launch's tag, sums them in tile order
// (deterministic) and the workgroup writes norm_out for the block it still holds.  The grid is ordered ROW BLOCK MAJOR, so the workgroups that wait for each
// other are dispatched back to back: no residency assumption beyond in-order dispatch.  No fences: an agent-scope release / acquire per workgroup walks the
// XCD's L2 (first form of this kernel: correct, +42 us per launch).  A wait that outlasts ~30 ms gives up and poisons norm_out with NaN (a lost workgroup
// must not hang the box; tests see the NaN).
template <typename T, int S>
__global__ __launch_bounds__(256) void splitk_reduce_rows_norm_kernel(GemmArgs a) {
    constexpr int PITCH = 260;
    __shared__ __attribute__((aligned(16))) float sums[32 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wn = tid >> 6;
    const int mtiles = (a.M + 255) >> 8, ntiles = (a.N + 255) >> 8;
    const int blk = blockIdx.x;
    const int rb = blk / ntiles, tile_n = blk - rb * ntiles;               // row-block major
    const int tile_m = rb >> 3, wm = (rb >> 2) & 1, j = rb & 3;
    const int tile = tile_n * mtiles + tile_m;
    const int m_base = (tile_m << 8) + wm * 128 + j * 32, n0 = tile_n << 8;
    if (m_base >= a.M) return;                                             // a row block of padding: none of its workgroups touches the counters
    const int r_out = tid >> 3, seg = tid & 7;
    const int m_out = m_base + r_out;
    const bool live = m_out < a.M;
    const T* R = reinterpret_cast<const T*>(a.R);
    const T* bias = reinterpret_cast<const T*>(a.bias);
    const T* g = reinterpret_cast<const T*>(a.norm_w);
    uint2 rr[8], gw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int n = n0 + (seg + 8 * k) * 4;
        rr[k] = (R && live && n < a.N) ? *reinterpret_cast<const uint2*>(R + (size_t)m_out * a.ldr + n) : uint2{0u, 0u};
        gw[k] = n < a.N ? *reinterpret_cast<const uint2*>(g + n) : uint2{0u, 0u};
    }
    const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
    constexpr uint32_t SLAB_BYTES = P8_SLAB_FLOATS * sizeof(float);
    const uint32_t tile_off = (uint32_t)((size_t)tile * S * SLAB_BYTES);
    const uint32_t lane_off = (uint32_t)(wm * 256 + tid) * 16u;
    v4u_t w[2][S][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < S; ++sl)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[i][sl][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, lane_off + (uint32_t)(((i * 4 + j) * 4 + q) * 8192) + sl * SLAB_BYTES, tile_off, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t = f32x4{__uint_as_float(w[i][0][q].x), __uint_as_float(w[i][0][q].y), __uint_as_float(w[i][0][q].z), __uint_as_float(w[i][0][q].w)};
#pragma unroll
            for (int sl = 1; sl < S; ++sl)
                t += f32x4{__uint_as_float(w[i][sl][q].x), __uint_as_float(w[i][sl][q].y), __uint_as_float(w[i][sl][q].z), __uint_as_float(w[i][sl][q].w)};
            *reinterpret_cast<f32x4*>(sums + l31 * PITCH + wn * 64 + i * 32 + 8 * q + 4 * hi) = t;
        }
    __syncthreads();
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const int act = a.act;
    float hv[8][4];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = (seg + 8 * k) * 4, n = n0 + c;
        const f32x4 t = *reinterpret_cast<const f32x4*>(sums + r_out * PITCH + c);
        float v[4] = {t.x, t.y, t.z, t.w};
        if (bias && n < a.N) {
            float b[4]; load4<T>(bias + n, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        if (R) { v[0] += unpack_lo<T>(rr[k].x); v[1] += unpack_hi<T>(rr[k].x); v[2] += unpack_lo<T>(rr[k].y); v[3] += unpack_hi<T>(rr[k].y); }
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[k][e] = (live && n < a.N) ? round_to<T>(v[e]) : 0.f;       // LlamaRMSNorm squares the STORED (T) values
        if (live && n < a.N) store4<T>(C + (size_t)m_out * a.ldc + n, hv[k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += hv[k][e] * hv[k][e];
    }
    // this block's sum of squares of row r_out: the 8 consecutive lanes of the row
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
    // Exchange without fences or counters: a partial travels as ONE 8-byte granule {value, tag of this launch} (relaxed agent-scope atomic store: single-copy
    // atomic, performed at the coherent level); the row's first lane polls the row's ntiles granules until every tag is this launch's.  (A first form — sc1
    // stores, agent release, arrival counter, agent acquire — was correct and cost +42 us per launch: every workgroup's release / acquire walks its XCD's L2.)
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(a.norm_part);
    const unsigned tag = a.norm_tag;
    if (seg == 0 && live) __hip_atomic_store(gran + (size_t)m_out * ntiles + tile_n, ((unsigned long long)tag << 32) | __float_as_uint(ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float tot = 0.f;
    int wait_ok = 1;
    for (int it = 0;; ++it) {
        bool ok = true;
        if (seg == 0 && live) {
            unsigned long long gv[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) gv[t] = t < ntiles ? __hip_atomic_load(gran + (size_t)m_out * ntiles + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
            tot = 0.f;
#pragma unroll
            for (int t = 0; t < 32; ++t) { ok = ok && (unsigned)(gv[t] >> 32) == tag; tot += __uint_as_float((unsigned)gv[t]); }      // tile order: deterministic
        }
        if (__syncthreads_and(ok ? 1 : 0)) break;
        __builtin_amdgcn_s_sleep(4);
        if (it > 30000) { wait_ok = 0; break; }                               // uniform: `it` is the same in every thread
    }
    tot = __shfl(tot, lane & ~7, 64);
    if (!live) return;
    float inv = rsqrtf(tot / (float)a.N + a.norm_eps);
    if (!wait_ok) inv = __builtin_nanf("");
    T* __restrict__ Y = reinterpret_cast<T*>(a.norm_out) + (size_t)m_out * a.ld_norm;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int n = n0 + (seg + 8 * k) * 4;
        if (n >= a.N) continue;
        float y[4];
        y[0] = round_to<T>(hv[k][0] * inv) * unpack_lo<T>(gw[k].x); y[1] = round_to<T>(hv[k][1] * inv) * unpack_hi<T>(gw[k].x);
        y[2] = round_to<T>(hv[k][2] * inv) * unpack_lo<T>(gw[k].y); y[3] = round_to<T>(hv[k][3] * inv) * unpack_hi<T>(gw[k].y);
        store4<T>(Y + n, y);
    }
}

// splitk_reduce_rowmajor_kernel (round 4): reduction of ROW-MAJOR slabs (split_mode 7) by workgroups that own whole rows — two rows per workgroup, 128 threads per
// row, thread u takes the float4 columns u + 128 k: every load instruction of a wave reads a 1-KiB run of one (tile, slice) slab, every store a 512-byte run of
// the output row.  Sums in slice order (same values as the tile-shaped reduction), bias / activation / residual, one rounding to T; with norm_w the workgroup
// also writes LlamaRMSNorm of the rows it just finished (HF rounding points: statistics over the stored T values in fp32, round(x * inv) * w) — the next block's
// input, so the 64 rmsnorm launches of a 7B prefill disappear without any exchange between workgroups.
template <typename T, int S>
__global__ __launch_bounds__(256) void splitk_reduce_rowmajor_kernel(GemmArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid >> 7, u = tid & 127;
    const int m = blockIdx.x * 2 + r;
    const bool live = m < a.M;
    const int mc = live ? m : a.M - 1;
    const int mtiles = (a.M + 255) >> 8;
    const int tile_m = mc >> 8, rit = mc & 255;
    const int n4 = a.N >> 2;                                             // float4 columns of a row (N % 4 == 0)
    const float* __restrict__ slabs = static_cast<const float*>(a.skw);
    T* __restrict__ C = reinterpret_cast<T*>(a.C) + (size_t)mc * a.ldc;
    const T* bias = reinterpret_cast<const T*>(a.bias);
    const T* R = a.R ? reinterpret_cast<const T*>(a.R) + (size_t)mc * a.ldr : nullptr;
    constexpr int KMAX = 16;                                             // N <= 8192
    constexpr int KB = 8;                                                // float4 columns per batch: KB x S loads in flight per thread
    float hv[KMAX][4];
    float ss = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < KMAX; k0 += KB) {
        if (k0 * 128 < n4) {                                             // workgroup-uniform
            f32x4 w[KB][S];
            uint2 rr[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int c4 = u + 128 * (k0 + kk);
                const int cc = c4 < n4 ? c4 : n4 - 1;                      // past the row's end: re-read its last piece (masked below)
                const int col = cc * 4, tile_n = col >> 8;
                const float* p = slabs + ((size_t)(tile_n * mtiles + tile_m) * S) * P8_SLAB_FLOATS + rit * 256 + (col & 255);
#pragma unroll
                for (int sl = 0; sl < S; ++sl) w[kk][sl] = *reinterpret_cast<const f32x4*>(p + (size_t)sl * P8_SLAB_FLOATS);
                rr[kk] = R ? *reinterpret_cast<const uint2*>(R + col) : uint2{0u, 0u};
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const int k = k0 + kk;
                const int c4 = u + 128 * k, col = c4 * 4;
                f32x4 t = w[kk][0];
#pragma unroll
                for (int sl = 1; sl < S; ++sl) t += w[kk][sl];
                float v[4] = {t.x, t.y, t.z, t.w};
                if (c4 < n4) {
                    if (bias) { float b[4]; load4<T>(bias + col, b);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b[e]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                    if (R) { v[0] += unpack_lo<T>(rr[kk].x); v[1] += unpack_hi<T>(rr[kk].x); v[2] += unpack_lo<T>(rr[kk].y); v[3] += unpack_hi<T>(rr[kk].y); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[k][e] = round_to<T>(v[e]);
                    if (live) store4<T>(C + col, hv[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ss += hv[k][e] * hv[k][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[k][e] = 0.f;
                }
            }
        }
    }
    if (!a.norm_w) return;                                               // workgroup-uniform
    // sum of squares of row r: its 128 threads = waves 2 r and 2 r + 1, lanes in butterfly order, then the two waves in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float tot = red[2 * r] + red[2 * r + 1];
    const float inv = rsqrtf(tot / (float)a.N + a.norm_eps);
    if (!live) return;
    const T* g = reinterpret_cast<const T*>(a.norm_w);
    T* __restrict__ Y = reinterpret_cast<T*>(a.norm_out) + (size_t)m * a.ld_norm;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c4 = u + 128 * k;
        if (c4 < n4) {
            float gv[4], y[4]; load4<T>(g + c4 * 4, gv);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = round_to<T>(hv[k][e] * inv) * gv[e];
            store4<T>(Y + c4 * 4, y);
        }
    }
}

// splitk_reduce_hyb_kernel: launch-boundary reduction of the TAIL-SPLIT order's K-halves (round 4).  Only the last full tiles of every XCD were sliced; slab
// pair `p` = (xcd, jj) belongs to the tile the GEMM kernel derived from the same numbers (see its hybrid branch), so the mapping is recomputed here.  8 workgroups
// per sliced tile, accumulator-order loads, the shared epilogue (incl. SiLU*mul: gate|up is the launch this order exists for).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_hyb_kernel(GemmArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wn = tid >> 6;
    const int mtiles = (a.M + 255) >> 8, ntiles = (a.N + 255) >> 8;
    const int blk = blockIdx.x;
    const int pair = blk >> 3, wm = (blk >> 2) & 1, j = blk & 3;
    const int xcd = pair / a.hyb_split, jj = pair - xcd * a.hyb_split;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int n_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, n_cnt = q + (xcd < r ? 1 : 0);
    const int mf = (a.M & 255) ? mtiles - 1 : mtiles;
    const int full = n_cnt * mf;
    const int nsplit = full - a.hyb_unsplit < a.hyb_split ? (full - a.hyb_unsplit > 0 ? full - a.hyb_unsplit : 0) : a.hyb_split;
    if (jj >= nsplit) return;                                             // this XCD sliced fewer tiles than the widest one
    const int f = full - nsplit + jj;
    const int tile_n = n_lo + f / mf, tile_m = f - (f / mf) * mf;
    const int m_base = (tile_m << 8) + wm * 128 + j * 32, n_base = (tile_n << 8) + wn * 64;
    const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
    constexpr uint32_t SLAB_BYTES = P8_SLAB_FLOATS * sizeof(float);
    const uint32_t tile_off = (uint32_t)((size_t)pair * 2 * SLAB_BYTES);
    const uint32_t lane_off = (uint32_t)(wm * 256 + tid) * 16u;
    v4u_t w[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                w[i][sl][qq] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, lane_off + (uint32_t)(((i * 4 + j) * 4 + qq) * 8192) + sl * SLAB_BYTES, tile_off, 0);
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            f32x4 t = f32x4{__uint_as_float(w[i][0][qq].x), __uint_as_float(w[i][0][qq].y), __uint_as_float(w[i][0][qq].z), __uint_as_float(w[i][0][qq].w)};
            t += f32x4{__uint_as_float(w[i][1][qq].x), __uint_as_float(w[i][1][qq].y), __uint_as_float(w[i][1][qq].z), __uint_as_float(w[i][1][qq].w)};
            acc[i][0][4 * qq] = t.x; acc[i][0][4 * qq + 1] = t.y; acc[i][0][4 * qq + 2] = t.z; acc[i][0][4 * qq + 3] = t.w;
        }
    gemm_epilogue<T, 1, 2>(a, acc, m_base, n_base, l31, hi);
}

template <typename T>
static void launch_gemm8p_t(GemmArgs a, int flavour, hipStream_t st) {
    LMX_REQUIRE(a.K % 64 == 0, "gemm8p: K must be a multiple of 64");
    LMX_REQUIRE((size_t)a.M * a.ldx * sizeof(T) < ((size_t)1 << 32) && (size_t)a.N * a.ldw * sizeof(T) < ((size_t)1 << 32),
                "gemm8p: operands must be addressable with 32-bit byte offsets");
    if (a.act == kActSiluMul) LMX_REQUIRE(a.N % 64 == 0, "gemm8p: SiLU·mul needs N % 64 == 0");
    const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
    int S = a.split_k > 0 ? a.split_k : gemm8p_pick_split(a.M, a.N, a.K);
    if (S > 3) S = 3;                                   // the in-launch reducer keeps three partial tiles in flight
    if (S > a.K / 64) S = a.K / 64;
    if (S < 1) S = 1;
    // Tail split (experiment arm: flavour 3 / variant 36 forces it, LMX_GEMM8P_TAIL=1 makes it automatic): with more full tiles than CUs the launch needs a second round that a third of
    // the chip sits out (7B gate|up at 1087 rows: 344 full + 86 ragged tiles on 256 CUs).  The kernel then walks, per XCD, whole tiles first, the last full
    // tiles as two K-halves, the ragged M-tiles last (see the kernel); needs the fp32 partial-tile scratch.
    int grid = tiles * S;
    a.hyb_unsplit = a.hyb_split = 0;
    {
        static const int tail = [] { const char* e = getenv("LMX_GEMM8P_TAIL"); return e ? atoi(e) : 0; }();      // r2-T (in-launch reduction, dear ragged tiles): slower; round 4 form: see EXPERIMENTS.md r4-H
        static const int cus_x = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 > 0 ? n / 8 : 32; }();
        int cus = cus_x;
        if (const char* e = getenv("LMX_GEMM8P_TAIL_CUS")) { const int v = atoi(e); if (v >= 1 && v <= 64) cus = v; }      // test knob: pretend an XCD has v CUs
        const int mt = cdiv(a.M, 256), nt = cdiv(a.N, 256), mf = (a.M % 256) ? mt - 1 : mt;
        const int n_max = (nt + 7) / 8;                       // N-tiles of the widest XCD
        const int full_max = n_max * mf;
        if (S == 1 && (flavour == 3 || (tail && flavour == 0)) && mf >= 1 && nt >= 8 && full_max > cus && a.K / 64 >= 32) {
            int nsplit = full_max - cus;
            if (nsplit > 16) nsplit = 16;                      // 8 XCDs x 16 tiles x 2 halves = 256 partial tiles = the 64 MiB scratch
            a.hyb_unsplit = cus; a.hyb_split = nsplit;
            S = 2;                                            // the kernel instance with the in-launch reduction; whole tiles skip it
            grid = 8 * (full_max + nsplit + n_max * (mt - mf));
        }
    }
    // M-tail order (flavour 5 / variant 37 forces it; automatic for the shipping flavour unless LMX_GEMM8P_MTAIL=0): an un-split launch whose last round of full
    // tiles would occupy at most half of the CUs runs those tiles as two 128-row halves each (see the kernel) — no scratch, bit-identical results.
    a.mt_whole = 0;
    {
        static const int mtail = [] { const char* e = getenv("LMX_GEMM8P_MTAIL"); return e ? atoi(e) : 0; }();
        static const int cus_x = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 > 0 ? n / 8 : 32; }();
        int cus = cus_x;
        if (const char* e = getenv("LMX_GEMM8P_TAIL_CUS")) { const int v = atoi(e); if (v >= 1 && v <= 64) cus = v; }      // test knob: pretend an XCD has v CUs
        const int mt = cdiv(a.M, 256), nt = cdiv(a.N, 256), mf = (a.M % 256) ? mt - 1 : mt;
        const int n_max = (nt + 7) / 8, full_max = n_max * mf;
        if (S == 1 && !a.hyb_unsplit && !a.qf_kc && (flavour == 5 || (mtail && flavour == 0)) && mf >= 1 && nt >= 8 && full_max > cus) {
            const int whole = flavour == 5 ? cus : (full_max / cus) * cus;
            const int rem = full_max - whole;
            if (flavour == 5 || (rem > 0 && 2 * rem <= cus)) { a.mt_whole = whole; grid = 8 * (full_max + rem + n_max * (mt - mf)); }
        }
    }
    a.split_k = S;
    { static const int mode = [] { const char* e = getenv("LMX_SPLITK_MODE"); return e ? atoi(e) : 5; }(); a.split_mode = mode; }
    // the tail-split order mixes whole and K-sliced tiles.  Round 4: its halves also go through a launch-boundary reduction (splitk_reduce_hyb_kernel);
    // LMX_GEMM8P_TAIL_INLAUNCH=1 keeps round 2's in-launch reduction by the last arriver (A/B)
    if (a.hyb_unsplit) {
        static const bool inl = [] { const char* e = getenv("LMX_GEMM8P_TAIL_INLAUNCH"); return e && atoi(e) != 0; }();
        if (a.split_mode == 7 || (inl && a.split_mode == 5)) a.split_mode = 1;
    }
    // row-major slabs + row-owning reduction (split_mode 7): what a fused RMSNorm of mode 3 rides on (gemm_norm_mode); LMX_SPLITK_MODE=7 forces it for every
    // K-sliced launch (A/B of the plain reduction)
    const bool rowmajor = S > 1 && !a.hyb_unsplit && a.act != kActSiluMul && a.N % 4 == 0 && a.N <= 8192 && a.ldc % 4 == 0 && (!a.R || a.ldr % 4 == 0) &&
                          (a.split_mode == 7 || (a.split_mode == 5 && a.norm_w && gemm_norm_mode() == 3));
    if (a.split_mode == 7 && !rowmajor) a.split_mode = 5;
    if (rowmajor) a.split_mode = 7;
    { static const int ns = [] { const char* e = getenv("LMX_GEMM8P_NOSKIP"); return e ? atoi(e) : 0; }(); a.no_skip = ns; }
    if (S > 1 && (!a.skw || !a.skc)) {
        std::lock_guard<std::mutex> lk(g_fb.mu);
        const size_t need = a.hyb_unsplit ? (size_t)256 * P8_SLAB_FLOATS * sizeof(float) : gemm8p_splitk_ws_bytes(a.M, a.N, S);
        const size_t cneed = a.hyb_unsplit ? (size_t)4096 : gemm8p_splitk_counter_bytes(a.M, a.N);
        if (need > g_fb.ws_bytes) {
            LMX_CHECK_HIP(hipStreamSynchronize(st));
            if (g_fb.ws) (void)hipFree(g_fb.ws);
            LMX_CHECK_HIP(hipMalloc(&g_fb.ws, need)); g_fb.ws_bytes = need;
        }
        if (cneed > g_fb.cnt_bytes) {
            LMX_CHECK_HIP(hipStreamSynchronize(st));
            if (g_fb.cnt) (void)hipFree(g_fb.cnt);
            LMX_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_fb.cnt), cneed)); g_fb.cnt_bytes = cneed;
            LMX_CHECK_HIP(hipMemsetAsync(g_fb.cnt, 0, cneed, st));
        }
        a.skw = g_fb.ws; a.skc = g_fb.cnt;
    }
    // A K-sliced GEMM with the launch-boundary reduction is TWO launches; an armed kernel timer (in-situ profile) then spans both: start stamped at the GEMM's
    // begin, stop at the reduction's end, so the reported duration includes the boundary between them.
    const bool two = S > 1 && (a.split_mode == 5 || a.split_mode == 7);
    KernelTimer* kt = g_kernel_timer;
    const bool timed = kt && !kt->used;
    auto launch = [&](auto kern) {
        // one opt-in per kernel instantiation (every instantiation decays to the same pointer TYPE, so the flag is keyed on the pointer value)
        static std::mutex mu; static std::set<const void*> done;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!done.count(reinterpret_cast<const void*>(kern))) {
                LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QF_LDS > P8_LDS ? QF_LDS : P8_LDS));
                done.insert(reinterpret_cast<const void*>(kern));
            }
        }
        const int lds = (a.qf_kc || a.split_mode == 7) ? (QF_LDS > P8_LDS ? QF_LDS : P8_LDS) : P8_LDS;      // split_mode 7 turns the tile through 128 x 260 floats
        if (two && timed) { kt->used = true; hipExtLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, kt->e0, nullptr, 0, a); }
        else LMX_LAUNCH(kern, dim3(grid), dim3(512), lds, st, a);
        LMX_CHECK_HIP(hipGetLastError());
    };
    if (a.qf_kc) LMX_REQUIRE(S == 1 && !a.hyb_unsplit && flavour == 0 && a.qf_rope && a.qf_vt && (a.qf_D == 64 || a.qf_D == 128) && (a.qf_nh * a.qf_D) % 256 == 0 &&
                             (a.qf_nkv * a.qf_D) % 256 == 0 && a.N == (a.qf_nh + 2 * a.qf_nkv) * a.qf_D && a.qf_pos0 % 8 == 0 && a.qf_smax % 8 == 0 && !a.bias && !a.R &&
                             a.act == kActNone && a.ldc % 8 == 0, "gemm8p: the fused q|k|v epilogue needs an un-split launch over head-aligned tiles (gemm_fuses_qkv)");
    // flavour: 0 = shipping form; 1 = no s_setprio; 2 = wave groups in lock-step; 3 = tail split (K-halves) forced, 4 = plain order (no tail split, no M-tail), 5 = M-tail forced
    // (A/B arms for tools/mb_gemm_variants.py)
    const bool boundary_only = a.split_mode == 5 || (a.split_mode == 7 && !a.hyb_unsplit);
    if (S == 3 && boundary_only) launch(gemm8p_kernel<T, true, true, 3, false>);
    else if (S == 2 && boundary_only) launch(gemm8p_kernel<T, true, true, 2, false>);
    else if (S == 3) launch(gemm8p_kernel<T, true, true, 3>);
    else if (S == 2) launch(gemm8p_kernel<T, true, true, 2>);
    else if (flavour == 1) launch(gemm8p_kernel<T, false, true, 1>);
    else if (flavour == 2) launch(gemm8p_kernel<T, true, false, 1>);
    else launch(gemm8p_kernel<T, true, true, 1>);
    if (a.norm_w) LMX_REQUIRE(two && a.norm_out && a.act != kActSiluMul && a.N % 4 == 0 && a.N <= 8192 && a.ldc % 4 == 0 && a.ld_norm % 4 == 0 && (!a.R || a.ldr % 4 == 0),
                              "gemm8p: the fused RMSNorm needs the K-sliced launch with the launch-boundary reduction (gemm_fuses_norm) and N <= 8192");
    if (two && a.hyb_unsplit) {
        const dim3 rg(8 * a.hyb_split * 8);
        if (timed) hipExtLaunchKernelGGL((splitk_reduce_hyb_kernel<T>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_hyb_kernel<T>), rg, dim3(256), 0, st, a);
        LMX_CHECK_HIP(hipGetLastError());
    } else if (two && a.split_mode == 7) {
        if (a.norm_w) LMX_REQUIRE(a.norm_out && a.ld_norm % 4 == 0, "gemm8p: fused RMSNorm needs norm_out with 8-byte aligned rows");
        const dim3 rg((a.M + 1) / 2);
        if (S == 3) { if (timed) hipExtLaunchKernelGGL((splitk_reduce_rowmajor_kernel<T, 3>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_rowmajor_kernel<T, 3>), rg, dim3(256), 0, st, a); }
        else { if (timed) hipExtLaunchKernelGGL((splitk_reduce_rowmajor_kernel<T, 2>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_rowmajor_kernel<T, 2>), rg, dim3(256), 0, st, a); }
        LMX_CHECK_HIP(hipGetLastError());
    } else if (two && a.norm_w && a.norm_part) {
        LMX_REQUIRE(a.N <= 8192 && a.norm_tag != 0, "gemm8p: the tile-shaped fused RMSNorm takes N <= 8192 and a non-zero launch tag");
        const dim3 rg(tiles * 8);
        if (S == 3) { if (timed) hipExtLaunchKernelGGL((splitk_reduce_rows_norm_kernel<T, 3>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_rows_norm_kernel<T, 3>), rg, dim3(256), 0, st, a); }
        else { if (timed) hipExtLaunchKernelGGL((splitk_reduce_rows_norm_kernel<T, 2>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_rows_norm_kernel<T, 2>), rg, dim3(256), 0, st, a); }
        LMX_CHECK_HIP(hipGetLastError());
    } else if (two && a.norm_w) {
        const dim3 rg((a.M + 3) / 4);
        if (S == 3) { if (timed) hipExtLaunchKernelGGL((splitk_reduce_norm_kernel<T, 3>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_norm_kernel<T, 3>), rg, dim3(256), 0, st, a); }
        else { if (timed) hipExtLaunchKernelGGL((splitk_reduce_norm_kernel<T, 2>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_norm_kernel<T, 2>), rg, dim3(256), 0, st, a); }
        LMX_CHECK_HIP(hipGetLastError());
    } else if (two) {
        const dim3 rg(tiles * 8);
        // row-order output through LDS (default); LMX_SPLITK_ROWS=0 keeps the accumulator-order epilogue.  SiLU.mul pairs columns 32 apart: old form.
        static const bool rows = [] { const char* e = getenv("LMX_SPLITK_ROWS"); return !(e && atoi(e) == 0); }();
        if (rows && a.act != kActSiluMul && a.N % 4 == 0 && a.ldc % 4 == 0 && (!a.R || a.ldr % 4 == 0)) {
            if (S == 3) { if (timed) hipExtLaunchKernelGGL((splitk_reduce_rows_kernel<T, 3>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_rows_kernel<T, 3>), rg, dim3(256), 0, st, a); }
            else { if (timed) hipExtLaunchKernelGGL((splitk_reduce_rows_kernel<T, 2>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_rows_kernel<T, 2>), rg, dim3(256), 0, st, a); }
        } else if (S == 3) { if (timed) hipExtLaunchKernelGGL((splitk_reduce_kernel<T, 3>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_kernel<T, 3>), rg, dim3(256), 0, st, a); }
        else { if (timed) hipExtLaunchKernelGGL((splitk_reduce_kernel<T, 2>), rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL((splitk_reduce_kernel<T, 2>), rg, dim3(256), 0, st, a); }
        LMX_CHECK_HIP(hipGetLastError());
    }
}

void launch_gemm8p(int dtype, const GemmArgs& a, int flavour, hipStream_t st) {
    if (dtype == kBF16) launch_gemm8p_t<bf16_t>(a, flavour, st);
    else if (dtype == kF16) launch_gemm8p_t<f16_t>(a, flavour, st);
    else throw Error{"gemm8p: 16-bit dtypes only"};
}

}  // namespace lmx
