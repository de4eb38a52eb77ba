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
//   * split-K (o_proj / down_proj: N = hidden gives only 80 tiles at T ~ 1k): S workgroups share a tile, each takes a K range, stores its fp32 partial
//     tile in accumulator order (1-KiB coalesced plain stores) and EXITS; a second launch (8 workgroups per tile) adds the S slabs in slice order
//     (deterministic) and runs the epilogue.
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

// SPLIT: K slices per tile (1, 2 or 3).  A K-sliced workgroup stores its fp32 partial tile and exits; the kernel boundary publishes the slabs and
// splitk_reduce_rows_kernel / splitk_reduce_kernel sum them (launch-boundary split-K, EXPERIMENTS.md r3-E).  The in-launch reduction by the last arriver, the
// tail-split / M-tail tile orders and the fused-RMSNorm reductions of rounds 2 - 4 measured slower or equal and were removed in round 5 (EXPERIMENTS.md r2-I,
// r2-T, r3-M, r4-B, r4-H name them and the commits that hold them).  The two levers of the schedule are fixed on: s_setprio around the MFMA segments and the
// one-barrier stagger of the two wave groups (without either: 830 / 838 TF against 1016 TF on the q|k|v shape).
template <typename T, int SPLIT>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const bool group1 = wave >= 4;

    constexpr int S = SPLIT;                          // K slices per tile (compile time: 1, 2 or 3)
    const int mtiles = (a.M + 255) >> 8;
    const int ntiles = (a.N + 255) >> 8;
    const int lid = xcd_remap(blockIdx.x, mtiles * ntiles * S);
    const int tile = lid / S, slice = lid - tile * S; // tile: index of the partial-tile slabs of a K-sliced tile
    int tile_m, tile_n;
    if constexpr (S == 1) gemm_tile_of(tile, mtiles, ntiles, tile_m, tile_n);        // K-sliced launches keep column-major ids: the reduction kernels decode the slab index the same way
    else { tile_n = tile / mtiles; tile_m = tile - tile_n * mtiles; }
    const int m0 = tile_m << 8, n0 = tile_n << 8;
    const int nk_all = a.K >> 6;
    const int kt_begin = (int)((long)nk_all * slice / S), kt_end = (int)((long)nk_all * (slice + 1) / S);
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
        if (h < n_half && ((h & 3) != 3 || a.M - m0 > 64)) stage(kt_begin + (h >> 2), h & 3, ((h >> 2) & 1) * 4 + (h & 3));
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
    const bool x1_needed = a.M - m0 > 64;
    const int mw = m0 + wm * 128;                     // first row of this wave's 128-row block
    const bool live[4] = {mw < a.M, mw + 32 < a.M, mw + 64 < a.M, mw + 96 < a.M};

    auto main_loop = [&](auto ragged_c) {
        constexpr bool RAGGED = decltype(ragged_c)::value != 0;
        constexpr bool SKIPX1 = decltype(ragged_c)::value == 2;       // ragged tile with <= 64 valid rows: X half 1 is neither staged, read nor multiplied
        auto mma = [&](const uint4 (&wb)[4], int nh, int mh) {
            __builtin_amdgcn_s_setprio(1);
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
            __builtin_amdgcn_s_setprio(0);
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
    if (m0 + 256 <= a.M) main_loop(IC<0>{});
    else if (x1_needed) main_loop(IC<1>{});
    else main_loop(IC<2>{});
    if (!group1) __builtin_amdgcn_s_barrier();        // balance the stagger barrier

    const int m_base = mw, n_base = n0 + wn * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (SPLIT == 1) {
        if (a.qf_kc) { qkv_rope_epilogue<T>(a, acc, smem, m0, n0, wm, wn, l31, hi, tid); return; }
        gemm_epilogue<T, 4, 2>(a, acc, m_base, n_base, l31, hi);
    } else {
        // ---- split-K: store the partial tile and exit; the kernel boundary publishes it (plain stores) ---------------------------------------
        // accumulator order: float4 number (i*4 + j)*4 + q of lane `tid` lives at byte ((idx * 512) + tid) * 16 of the slab, so every wave-instruction
        // moves 1 KiB of contiguous memory.  A 32-row block that lies entirely beyond M is never read back by the reduction (the last M tile of a
        // 1087-row prompt is 7/8 padding: 10.7 of the 61 MB of down_proj's slabs) and is not stored.
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
        const uint32_t slab_off = (uint32_t)(((size_t)tile * S + slice) * (P8_SLAB_FLOATS * sizeof(float)));   // < 2^31: <= 256 slabs of 256 KiB
        const uint32_t lane_off = (uint32_t)tid * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (m_base + j * 32 >= a.M) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = (i * 4 + j) * 4 + q;
                    v4u_t v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                               __float_as_uint(acc[i][j][4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, lane_off + idx * 8192, slab_off, 0);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Launch-boundary split-K reduction (the form every K-sliced launch takes).
// In the in-launch form of round 2 ONE workgroup per tile reads S x 256 KiB of partials back through one CU (~10 us at the ~60-100 GB/s a single CU pulls) after an
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


// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
size_t gemm8p_splitk_ws_bytes(int M, int N, int split_k) {
    if (split_k <= 1) return 0;
    const size_t tiles = (size_t)cdiv(M, 256) * cdiv(N, 256);
    return tiles * split_k * P8_SLAB_FLOATS * sizeof(float);
}

// K slices per tile for the ping-pong kernel: fill the 256 CUs when N = hidden gives too few 256x256 tiles, but keep every slice long enough (>= 20
// K-steps) that the fp32 partial-tile round trip (S x 256 KiB written and read per tile) stays small against its main loop.
int gemm8p_pick_split(int M, int N, int K) {
    const int tiles = cdiv(M, 256) * cdiv(N, 256);
    const int nk = K / 64;
    if (tiles >= 160) return 1;
    int s = 256 / tiles;
    if (s > 3) s = 3;
    while (s > 1 && nk / s < 20) --s;
    return s < 1 ? 1 : s;
}

namespace {
// fallback split-K scratch for callers that bring none (lmx_op_gemm: unit tests, microbenchmarks; ONE stream at a time).
// The engine passes per-sequence scratch instead (Model::prefill), so concurrent request threads never share it.
struct FallbackWs {
    std::mutex mu;
    void* ws = nullptr; size_t ws_bytes = 0;
};
FallbackWs g_fb;
}  // namespace

template <typename T>
static void launch_gemm8p_t(GemmArgs a, hipStream_t st) {
    LMX_REQUIRE(a.K % 64 == 0, "gemm8p: K must be a multiple of 64");
    LMX_REQUIRE((size_t)a.M * a.ldx * sizeof(T) < ((size_t)1 << 32) && (size_t)a.N * a.ldw * sizeof(T) < ((size_t)1 << 32),
                "gemm8p: operands must be addressable with 32-bit byte offsets");
    if (a.act == kActSiluMul) LMX_REQUIRE(a.N % 64 == 0, "gemm8p: SiLU·mul needs N % 64 == 0");
    const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
    int S = a.split_k > 0 ? a.split_k : gemm8p_pick_split(a.M, a.N, a.K);
    if (S > 3) S = 3;                                   // instantiated: 1, 2, 3 slices
    if (S > a.K / 64) S = a.K / 64;
    if (S < 1) S = 1;
    a.split_k = S;
    if (S > 1 && !a.skw) {
        std::lock_guard<std::mutex> lk(g_fb.mu);
        const size_t need = gemm8p_splitk_ws_bytes(a.M, a.N, S);
        if (need > g_fb.ws_bytes) {
            LMX_CHECK_HIP(hipStreamSynchronize(st));
            if (g_fb.ws) (void)hipFree(g_fb.ws);
            LMX_CHECK_HIP(hipMalloc(&g_fb.ws, need)); g_fb.ws_bytes = need;
        }
        a.skw = g_fb.ws;
    }
    // A K-sliced GEMM is TWO launches; an armed kernel timer (in-situ profile) then spans both: start stamped at the GEMM's begin, stop at the reduction's
    // end, so the reported duration includes the boundary between them.
    const bool two = S > 1;
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
        const int lds = a.qf_kc ? (QF_LDS > P8_LDS ? QF_LDS : P8_LDS) : P8_LDS;
        if (two && timed) { kt->used = true; hipExtLaunchKernelGGL(kern, dim3(tiles * S), dim3(512), lds, st, kt->e0, nullptr, 0, a); }
        else LMX_LAUNCH(kern, dim3(tiles * S), dim3(512), lds, st, a);
        LMX_CHECK_HIP(hipGetLastError());
    };
    if (a.qf_kc) LMX_REQUIRE(S == 1 && a.qf_rope && a.qf_vt && (a.qf_D == 64 || a.qf_D == 128) && (a.qf_nh * a.qf_D) % 256 == 0 &&
                             (a.qf_nkv * a.qf_D) % 256 == 0 && a.N == (a.qf_nh + 2 * a.qf_nkv) * a.qf_D && a.qf_pos0 % 8 == 0 && a.qf_smax % 8 == 0 && !a.bias && !a.R &&
                             a.act == kActNone && a.ldc % 8 == 0, "gemm8p: the fused q|k|v epilogue needs an un-split launch over head-aligned tiles (gemm_fuses_qkv)");
    // the CLIP K / V^T pack lives in gemm_epilogue only: the K-sliced launches' reduction kernels would write the k | v columns to C and leave the pack untouched
    if (a.pk_kc) LMX_REQUIRE(S == 1 && a.act != kActSiluMul && a.pk_D % 4 == 0, "gemm8p: the K / V^T pack epilogue needs an un-split launch of a plain linear");
    if (S == 3) launch(gemm8p_kernel<T, 3>);
    else if (S == 2) launch(gemm8p_kernel<T, 2>);
    else launch(gemm8p_kernel<T, 1>);
    if (two) {
        const dim3 rg(tiles * 8);
        // row-order output through LDS; SiLU.mul pairs columns 32 apart and unaligned rows keep the accumulator-order epilogue
        const bool rows = a.act != kActSiluMul && a.N % 4 == 0 && a.ldc % 4 == 0 && (!a.R || a.ldr % 4 == 0);
        auto red = [&](auto kern) {
            if (timed) hipExtLaunchKernelGGL(kern, rg, dim3(256), 0, st, nullptr, kt->e1, 0, a); else hipLaunchKernelGGL(kern, rg, dim3(256), 0, st, a);
            LMX_CHECK_HIP(hipGetLastError());
        };
        if (rows) { if (S == 3) red(splitk_reduce_rows_kernel<T, 3>); else red(splitk_reduce_rows_kernel<T, 2>); }
        else { if (S == 3) red(splitk_reduce_kernel<T, 3>); else red(splitk_reduce_kernel<T, 2>); }
    }
}

void launch_gemm8p(int dtype, const GemmArgs& a, hipStream_t st) {
    if (dtype == kBF16) launch_gemm8p_t<bf16_t>(a, st);
    else if (dtype == kF16) launch_gemm8p_t<f16_t>(a, st);
    else throw Error{"gemm8p: 16-bit dtypes only"};
}

}  // namespace lmx
