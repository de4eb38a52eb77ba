// Skinny linear layer for batched decode (continuous batching, 2..32 sequences per step):
//   C[M, N] = epilogue( X[M, K] · W[N, K]ᵀ ),   M <= 32 token rows, 16-bit operands, fp32 accumulation.
//
// One step of a decode batch reads every weight once for all M sequences, so the kernel is bound by the HBM weight stream
// exactly like the single-sequence GEMV (gemm.hip: gemv_kernel) — but M·K·N multiply-adds no longer fit the vector ALUs
// next to the bf16 unpacking (M = 8 already needs ~33 T lane-ops/s against a 39 T/s VALU), and M rows of x no longer fit
// LDS (8 × 11008 × 2 B × ... per workgroup).  So the contraction runs on v_mfma_f32_16x16x32_{bf16,f16} with BOTH operands
// fed straight from global memory into registers:
//   * A operand = weights: lane (i = lane & 15, q = lane >> 4) owns row n0+i and, per "super-step" of 128 k, the 64
//     contiguous bytes k0 + 32q .. +32  (four 16-byte loads = the A fragments of four MFMA steps).  A wave
//     therefore touches 16 rows × 256 contiguous bytes per super-step: whole 128-byte lines, no LDS staging, no swizzle.
//   * B operand = activations: lane (t = lane & 15, q) owns token t and the SAME k bytes of x — the MFMA's k order is
//     arbitrary as long as A and B agree, so the permuted k order costs nothing.  x is M·K·2 B ≤ 0.7 MB: L2-resident.
//   * the NW waves of a workgroup split K (wave w takes super-steps w, w+NW, ...: concurrent loads of a row are adjacent)
//     and reduce their partial C tiles through LDS in fixed wave order — deterministic, no atomics.
//   * from the [N][K] layout every 16-lane group of such a load touches 16 different rows; the engine therefore keeps a second copy of
//     the decoder weights in FRAGMENT ORDER (skinny_swizzle_kernel: per 16-row tile and 128-k super-step, four 1-KiB pieces, piece j =
//     the lane-linear A fragments of MFMA step j) and the SWZ instantiation reads 1 KiB of contiguous memory per load instruction.
//   * a workgroup owns 16·RT weight rows (RT = 2: for SiLU·mul the two tiles are the gate rows and the matching up rows of
//     the [32 gate | 32 up] interleaved weight, so silu(g)·u happens in the epilogue like in the GEMM / GEMV).
//   * K SLICES ACROSS WORKGROUPS (round 6, narrow layers: N = hidden).  The x rows are re-read from L2 by every workgroup, and a CU's load path does not tell an L2
//     hit from an HBM line: at 32 rows o_proj moved 33.5 MB of weights + 256 workgroups x 256 KB of x = 67 MB and down_proj 90 + 180 MB (13.1 / 30.5 us: 2.6 / 3.0 TB/s of
//     weights).  Four row tiles per workgroup cut the x traffic by four but leave N / 64 = 64 workgroups; so gridDim.y = KS workgroups share a 64-row tile, each takes a
//     contiguous K range (its 8 waves split that range as before), stores its fp32 partial tile write-through, and the LAST to arrive (one agent-scope ticket per
//     tile) adds the KS partials in slice order — deterministic — and runs the epilogue (guide §6 Guideline 16, R1: sc1 payload, drain, ticket; one acquire).
//     The same form fills the chip on tensor-parallel shards, whose linears are 48 - 96 workgroups otherwise.
// Epilogue order (bias -> activation -> residual -> round) is the GEMV's, so the batched step rounds like the single step.
// (Round 4's opt-in arm with the input rows' RMSNorm inside the launch — bit-identical, slower at full width, equal on small shards — was removed in round 5:
// profiles/EXPERIMENTS.md r4-L.)
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "wstream.h"

namespace lmx {

typedef uint32_t u32x4_s __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4_s a, u32x4_s b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b), c, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_t> {
    static __device__ __forceinline__ f32x4 run(u32x4_s a, u32x4_s b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_v, a), __builtin_bit_cast(f16x8_v, b), c, 0, 0, 0);
    }
};

constexpr int SK_SUPER = 128;    // k elements per super-step (4 lane groups × 32)

// four floats as ONE 16-byte write-through (sc1) store through a raw buffer view (guide §6 Guideline 16, R1 payload store)
__device__ __forceinline__ void ba_store16_wt(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, const float (&v)[4]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 d = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 16);
}

// STREAM (fragment-order weights, K % 128 == 0): the two stages of a wave as a HAND-COUNTED load stream (wstream.h).  hipcc's own placement puts an
// `s_waitcnt vmcnt(0)` at the top of every loop trip (ISA: the loads sit under trip-count conditions), so a wave waits for BOTH stages, multiplies, requests both
// again and waits for both again — the second stage never hides behind the first one's MFMAs.  Here every load is inline asm (buffer_load_dwordx4, weights `nt`,
// x cached), a stage is consumed after `s_waitcnt vmcnt(loads per stage)` — the OTHER stage's loads stay on the wire — and refilled at once; the last stages of a
// wave's share wait with the exact smaller counts (wave-uniform branches), so nothing is loaded past the end.  Same per-lane operation order: bit-identical.
template <typename T, int NW, int RT, int CT, bool SWZ = false, bool STREAM = false>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(GemmArgs a) {
    static_assert(!STREAM || SWZ, "the hand-counted stream reads the fragment-order copy");
    __shared__ float red[NW][RT * CT][256];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const bool silu = a.act == kActSiluMul;
    const int K = a.K;

    int rowbase[RT];
    constexpr int HALF = RT / 2;                  // SiLU·mul: HALF gate tiles + their HALF up tiles per workgroup
    if (silu) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int g = rt < HALF ? rt : rt - HALF;
            const int j = blockIdx.x * 16 * HALF + 16 * g;                       // first gate row (= output column) of this tile
            rowbase[rt] = 64 * (j >> 5) + (j & 31) + (rt < HALF ? 0 : 32);       // [32 gate | 32 up] interleaved weight rows
        }
    } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rowbase[rt] = (blockIdx.x * RT + rt) * 16;
    }
    const int nsuper = (K + SK_SUPER - 1) / SK_SUPER;
    // K slices across workgroups: slice blockIdx.y of gridDim.y takes super-steps [s_lo, s_hi)
    const int KS = (int)gridDim.y, slice = (int)blockIdx.y;
    const int s_lo = (int)((long)nsuper * slice / KS), s_hi = (int)((long)nsuper * (slice + 1) / KS);
    const T* wp[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if constexpr (SWZ) {
            // fragment-order copy: tile (rowbase / 16), super-step s, MFMA step j -> 1 KiB, lane-linear 16-byte fragments: every
            // load instruction of the wave reads 1 KiB of contiguous memory
            wp[rt] = reinterpret_cast<const T*>(static_cast<const char*>(a.Wsw) + (size_t)(rowbase[rt] >> 4) * nsuper * 4096 + lane * 16);
        } else {
            int n = rowbase[rt] + i; n = n < a.N ? n : a.N - 1;
            wp[rt] = reinterpret_cast<const T*>(a.W) + (size_t)n * a.ldw + q * 32;
        }
    }
    const T* xp[CT];
    bool xvalid[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int m = ct * 16 + i;
        xvalid[ct] = m < a.M;
        xp[ct] = reinterpret_cast<const T*>(a.X) + (size_t)(xvalid[ct] ? m : 0) * a.ldx + q * 32;
    }

    struct Stage { u32x4_s w[RT][4]; u32x4_s x[CT][4]; };
    const u32x4_s zero4 = {0u, 0u, 0u, 0u};
    auto load_stage = [&](Stage& s, int sup) {
        const int kb = sup * SK_SUPER;
        const bool kin = kb + q * 32 < K;      // K % 32 == 0: a lane's 32-element run is all in or all out
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if constexpr (SWZ) s.w[rt][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_s*>(reinterpret_cast<const char*>(wp[rt]) + ((size_t)sup * 4 + j) * 1024));
                else                 // plain (cached) loads: the four pieces of a 64-byte run share L1 lines; a streaming
                s.w[rt][j] = kin ? *(reinterpret_cast<const u32x4_s*>(wp[rt] + kb) + j) : zero4;   // hint re-fetched them (-30 %)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s.x[ct][j] = (kin && xvalid[ct]) ? *(reinterpret_cast<const u32x4_s*>(xp[ct] + kb) + j) : zero4;
    };

    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto consume = [&](Stage& s) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = Mfma16<T>::run(s.w[rt][j], s.x[ct][j], acc[rt][ct]);
    };

    Stage sa, sb;
    if constexpr (STREAM) {
        constexpr int LPS = RT * 4 + CT * 4;                 // loads per stage and lane
        static_assert(LPS <= 63, "vmcnt is a 6-bit field");
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const ws_v4i rsW = ws_make_rsrc(a.Wsw, 0x7fffffffu), rsX = ws_make_rsrc(a.X, 0x7fffffffu);
        uint32_t wvo[RT], xvo[CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wvo[rt] = (uint32_t)(rowbase[rt] >> 4) * (uint32_t)nsuper * 4096u + (uint32_t)lane * 16u;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            int m = ct * 16 + i; m = m < a.M ? m : a.M - 1;                    // rows past M: a real row's bytes — their output columns are never stored
            xvo[ct] = (uint32_t)m * (uint32_t)a.ldx * (uint32_t)sizeof(T) + (uint32_t)q * 64u;
        }
        auto issue = [&](Stage& s, int r) {                  // round r of this wave = super-step s_lo + wave + r NW
            const int sup = s_lo + wave_u + r * NW;
            const uint32_t wo = (uint32_t)sup * 4096u, xo = (uint32_t)sup * (uint32_t)(SK_SUPER * sizeof(T));
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) ws_load(s.w[rt][j], wvo[rt], rsW, wo + (uint32_t)j * 1024u);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) ws_load_plain(s.x[ct][j], xvo[ct], rsX, xo + (uint32_t)j * 16u);
        };
        auto landed = [&](Stage& s, bool younger) {          // younger: the other stage's LPS loads may stay on the wire; the stage's registers pass through: no use above
            if (younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(s.w[rt][j]));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(s.x[ct][j]));
        };
        // this wave's share: rounds 0 .. nr - 1 (wave-uniform); no load is issued past it, so every count below is exact without dummy loads
        const int nr = s_lo + wave_u < s_hi ? (s_hi - s_lo - wave_u + NW - 1) / NW : 0;
        if (nr > 0) issue(sa, 0);
        if (nr > 1) issue(sb, 1);
        for (int r = 0; r < nr; r += 2) {
            landed(sa, r + 1 < nr);
            consume(sa);
            if (r + 2 < nr) issue(sa, r + 2);
            if (r + 1 < nr) {
                landed(sb, r + 2 < nr);
                consume(sb);
                if (r + 3 < nr) issue(sb, r + 3);
            }
        }
    } else {
    if (s_lo + wave < s_hi) load_stage(sa, s_lo + wave);
    if (s_lo + wave + NW < s_hi) load_stage(sb, s_lo + wave + NW);
    for (int sup = s_lo + wave; sup < s_hi; sup += 2 * NW) {
        consume(sa);
        if (sup + 2 * NW < s_hi) load_stage(sa, sup + 2 * NW);
        if (sup + NW < s_hi) {
            consume(sb);
            if (sup + 3 * NW < s_hi) load_stage(sb, sup + 3 * NW);
        }
    }
    }

    // ---- split-K reduction across the waves (fixed order), then the epilogue -----------------------------------------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][rt * CT + ct][r * 64 + lane] = acc[rt][ct][r];
    __syncthreads();

    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const T* bias = reinterpret_cast<const T*>(a.bias);
    const T* Rr = reinterpret_cast<const T*>(a.R);
    // ---- K slices across workgroups: this workgroup's tile sums -> scratch (16-byte write-through stores), ticket; the last arriver adds the KS partial tiles in slice
    //      order into the LDS tile of wave 0 and runs the epilogue from there --------------------------------------------------------------------------------------
    int n_sum = NW;                                                        // waves whose LDS tiles the epilogue adds (1 once the slices have been combined)
    if (KS > 1) {
        __shared__ int s_last;
        constexpr int TILE_F = RT * CT * 256;                              // floats per partial tile
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.skw, 0, 0x7fffffff, 0x00020000);
        const uint32_t base = (uint32_t)(((size_t)blockIdx.x * KS) * TILE_F * sizeof(float));
        for (int e4 = tid; e4 < TILE_F / 4; e4 += NW * 64) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float4 t = *reinterpret_cast<const float4*>(&red[w][(e4 * 4) >> 8][(e4 * 4) & 255]);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
            ba_store16_wt(rs, base + (uint32_t)(slice * TILE_F + e4 * 4) * 4u, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave drains, then ONE arrival
        __syncthreads();
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(a.sk_cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == KS - 1;
            if (t == KS - 1) {
                __hip_atomic_store(a.sk_cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every arrival of this launch has been seen: re-arm
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // ONE acquire: the partial tiles of the other slices are read with plain loads below
            }
        }
        __syncthreads();
        if (!s_last) return;
        const float* parts = static_cast<const float*>(a.skw) + (size_t)blockIdx.x * KS * TILE_F;
        for (int e4 = tid; e4 < TILE_F / 4; e4 += NW * 64) {
            float4 acc4 = *reinterpret_cast<const float4*>(parts + e4 * 4);
            for (int s = 1; s < KS; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(parts + (size_t)s * TILE_F + e4 * 4);
                acc4.x += t.x; acc4.y += t.y; acc4.z += t.z; acc4.w += t.w;
            }
            *reinterpret_cast<float4*>(&red[0][(e4 * 4) >> 8][(e4 * 4) & 255]) = acc4;
        }
        __syncthreads();
        n_sum = 1;
    }
    auto tile_sum = [&](int tile, int token, int nrow) {
        const int idx = (nrow & 3) * 64 + (nrow >> 2) * 16 + token;     // D[i][j]: lane = 16*(i/4) + j, register = i % 4
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) if (w < n_sum) v += red[w][tile][idx];
        return v;
    };
    if constexpr (RT % 2 == 0) if (silu) {
        for (int e = tid; e < HALF * CT * 256; e += NW * 64) {
            const int gt = e >> 8, gi = gt / CT, ct = gt % CT, token = (e >> 4) & 15, nrow = e & 15;
            const int m = ct * 16 + token, j = blockIdx.x * 16 * HALF + 16 * gi + nrow;
            if (m >= a.M || j >= a.N / 2) continue;
            float g = tile_sum(gi * CT + ct, token, nrow), u = tile_sum((gi + HALF) * CT + ct, token, nrow);
            if (bias) { g += to_f32(bias[rowbase[gi] + nrow]); u += to_f32(bias[rowbase[gi + HALF] + nrow]); }
            C[(size_t)m * a.ldc + j] = from_f32<T>(act_silu(g) * u);
        }
        return;
    }
    for (int e = tid; e < RT * CT * 256; e += NW * 64) {
        const int tile = e >> 8, rt = tile / CT, ct = tile % CT, token = (e >> 4) & 15, nrow = e & 15;
        const int m = ct * 16 + token, n = rowbase[rt] + nrow;
        if (m >= a.M || n >= a.N) continue;
        float v = tile_sum(tile, token, nrow);
        if (bias) v += to_f32(bias[n]);
        if (a.act == kActQuickGelu) v = act_quick_gelu(v); else if (a.act == kActGeluErf) v = act_gelu_erf(v);
        if (Rr) v += to_f32(Rr[(size_t)m * a.ldr + n]);
        C[(size_t)m * a.ldc + n] = from_f32<T>(v);
    }
}

// K slices across workgroups for the 64-rows-per-workgroup form of a narrow layer: enough workgroups for every CU and a LONG K (each of a slice's 8 waves keeps at
// least two super-steps).  Measured at 32 rows, 7B (profiles/r06_skinny_kslices.jsonl): down_proj (K = 11008) 31.4 -> 23.0 us with 4 slices (3: 24.7, 2: 30.6, 8: 26.8),
// 16 rows 23.8 -> 20.7; o_proj (K = 4096) 13.7 -> 13.3 (4 slices) / 12.8 (2 slices of 32 rows): a slice of 8 super-steps is one per wave — not worth a hand-over.
int skinny_kslices(int M, int N, int K) {
    if (M <= 8 || N % 64 != 0) return 1;
    const int tiles = N / 64, nsuper = cdiv(K, SK_SUPER);
    int ks = cdiv(256, tiles);
    if (ks > 4) ks = 4;
    while (ks > 1 && nsuper / ks < 16) --ks;
    return tiles * ks >= 192 ? ks : 1;              // fewer slices than fill the chip: the 16-rows-per-workgroup form (256 workgroups) is the better one
}
size_t skinny_scratch_bytes(int N) { return (size_t)cdiv(N, 64) * 8 * (4 * 2 * 256) * sizeof(float); }      // tiles x max slices x (RT 4 x CT 2) accumulator blocks

template <typename T, bool SWZ>
static void launch_skinny_s(const GemmArgs& a, hipStream_t st) {
    constexpr int NW = 8;
    const bool silu = a.act == kActSiluMul;
    const int ct = a.M > 16 ? 2 : 1;
    // the fragment-order copy with whole 128-k super-steps takes the hand-counted load stream; same tile forms, same bits
    const bool stream = SWZ && a.K % SK_SUPER == 0;
    auto go = [&](auto rt_c, auto ct_c, int grid, int ks = 1) {
        constexpr int RTV = decltype(rt_c)::value, CTV = decltype(ct_c)::value;
        if constexpr (SWZ) { if (stream) { hipLaunchKernelGGL((skinny_gemm_kernel<T, NW, RTV, CTV, true, true>), dim3(grid, ks), dim3(NW * 64), 0, st, a); return; } }
        hipLaunchKernelGGL((skinny_gemm_kernel<T, NW, RTV, CTV, SWZ, false>), dim3(grid, ks), dim3(NW * 64), 0, st, a);
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    if (silu) {
        // more than 16 token rows: x re-reads from L2 rival the weight stream, so a workgroup takes a whole 64-row fused block
        if (ct == 2 && a.N >= 16384) go(I4{}, I2{}, a.N / 64);
        else if (ct == 1) go(I2{}, I1{}, a.N / 2 / 16);
        else go(I2{}, I2{}, a.N / 2 / 16);
    } else if (a.N % 64 == 0 && a.N >= 8192 && ct == 2) {
        // (two K slices for q|k|v's 192 tiles — 384 workgroups of 384 KB instead of 192 of 768 KB — measured SLOWER: 25.2 -> 29.1 us at 32 rows, r6-G)
        go(I4{}, I2{}, a.N / 64);
    } else if (a.N % 32 == 0 && a.N >= 8192) {
        // wide layers: two row tiles per workgroup halve the x re-reads; narrow ones keep one tile for more workgroups
        if (ct == 1) go(I2{}, I1{}, a.N / 32); else go(I2{}, I2{}, a.N / 32);
    } else if (a.skw && a.sk_cnt && a.N % 64 == 0 && skinny_kslices(a.M, a.N, a.K) > 1) {
        // narrow layers with scratch: 64 rows per workgroup (a quarter of the x re-reads) and K slices across workgroups to fill the chip
        const int ks = skinny_kslices(a.M, a.N, a.K);
        if (ct == 1) go(I4{}, I1{}, a.N / 64, ks); else go(I4{}, I2{}, a.N / 64, ks);
    } else {
        // narrow layers (N = hidden): 16 rows per workgroup — with only N/16 = 256 workgroups parallelism matters more than x re-reads
        // (32 / 64 rows per workgroup measured 21.5 / 29.6 us vs 16.7 us on o_proj at M = 32)
        if (ct == 1) go(I1{}, I1{}, cdiv(a.N, 16)); else go(I1{}, I2{}, cdiv(a.N, 16));
    }
    LMX_CHECK_HIP(hipGetLastError());
}

template <typename T>
static void launch_skinny_t(const GemmArgs& a, hipStream_t st) {
    if (a.Wsw) launch_skinny_s<T, true>(a, st); else launch_skinny_s<T, false>(a, st);
}

// ---- weight re-layout into fragment order --------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void skinny_swizzle_kernel(const T* __restrict__ W, int ldw, char* __restrict__ dst, int N, int K, int nsuper) {
    const int s = blockIdx.x, tile = blockIdx.y;
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int n = tile * 16 + i, k = s * SK_SUPER + 32 * q + 8 * j;
    u32x4_s v = {0u, 0u, 0u, 0u};
    if (n < N && k < K) v = *reinterpret_cast<const u32x4_s*>(W + (size_t)n * ldw + k);
    *reinterpret_cast<u32x4_s*>(dst + (((size_t)tile * nsuper + s) * 4 + j) * 1024 + lane * 16) = v;
}

size_t skinny_swizzled_bytes(int N, int K, int es) {
    (void)es;
    return (size_t)cdiv(N, 16) * cdiv(K, SK_SUPER) * 4096;
}

void launch_skinny_swizzle(int dtype, const void* W, int ldw, void* dst, int N, int K, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "skinny swizzle: 16-bit weights only");
    LMX_REQUIRE(K % 8 == 0 && ldw % 8 == 0, "skinny swizzle: K must be a multiple of 8");
    const dim3 grid(cdiv(K, SK_SUPER), cdiv(N, 16));
    if (dtype == kBF16) hipLaunchKernelGGL(skinny_swizzle_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)W, ldw, (char*)dst, N, K, cdiv(K, SK_SUPER));
    else hipLaunchKernelGGL(skinny_swizzle_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)W, ldw, (char*)dst, N, K, cdiv(K, SK_SUPER));
    LMX_CHECK_HIP(hipGetLastError());
}

void launch_skinny_gemm(int dtype, const GemmArgs& a, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "skinny gemm: 16-bit operands only (the fp32 verification engine batches through the GEMV)");
    LMX_REQUIRE(a.M >= 1 && a.M <= 32, "skinny gemm: 1..32 token rows");
    LMX_REQUIRE(a.N > 0 && a.K > 0 && a.K % 32 == 0, "skinny gemm: K must be a multiple of 32");
    LMX_REQUIRE(a.ldx % 8 == 0 && a.ldw % 8 == 0, "skinny gemm: leading dims must keep 16-byte row alignment");
    if (a.act == kActSiluMul) LMX_REQUIRE(a.N % 64 == 0, "skinny gemm: SiLU·mul needs N (fused gate|up rows) % 64 == 0");
    if (dtype == kBF16) launch_skinny_t<bf16_t>(a, st); else launch_skinny_t<f16_t>(a, st);
}

}  // namespace lmx
