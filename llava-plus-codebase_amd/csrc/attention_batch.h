// Decode attention of the decode BATCH (continuous batching), 16-bit models with head_dim 128: one workgroup per (sequence, head) whose waves STREAM the head's
// keys (attention.hip includes this; launch_decode_fused takes it for the batch form).  Written at the end of round 4, first run in round 5 (parity green,
// profiles/r05_batch_attn_wave.jsonl: 44.8 -> 38.9 us per layer at 8 sequences, 143 -> 134 at 32; the same body as the SINGLE request's launch lost, 16.8 -> 20.4 us:
// 32 workgroups cannot pull 19 MB fast enough — that wiring is gone, EXPERIMENTS.md r5-A).
//
// Why (profiles/EXPERIMENTS.md r4-N): the batch launch of decode_fused_kernel — one workgroup per (sequence, head, 128-key chunk), the whole chunk
// in registers, partials through memory, ticket, merge by the last arriver — moves a batch-8 layer's 142 MB of K / V^T at 3.2 TB/s (4.0 at 32 sequences): its
// resident workgroups start together, request everything at once, then compute and merge together with the HBM idle.  Contract reproduced: the single-token branch of
// llava/model/llava_arch.py:103-112 -> HF5:models/llama/modeling_llama.py:191-214, 243-281 (RoPE on q and the new key, KV append, softmax in fp32, probabilities x V),
// same inputs and outputs as decode_fused_kernel's batch form (DecodeFusedArgs with `tab`); NOT the same summation order (online softmax over 64-key pieces instead
// of 128-key partials merged in chunk order), so parity is by tolerance against the reference arithmetic, not bit-identity with the chunked launch.
//
// Shape: NWV waves per workgroup; wave w walks the 64-key pieces w, w + NWV, ... of the cached keys [0, pos): 16 K loads (4 keys x 256 B = 1 KiB contiguous per
// instruction) + 16 V^T loads (8 d-rows x 128 B per instruction) requested together, scores by 16-lane dot products, an online softmax whose statistics are
// wave-wide shuffles, the piece's 64 probabilities turned from the score layout (key 4 u + kslot) into the V^T layout (8 consecutive keys per lane) through 256
// bytes of wave-private LDS, P x V on the lane's 16 d-rows.  No workgroup barrier inside the loop, nothing goes through global memory: the waves of a CU drift apart,
// so some are always waiting on HBM while others multiply.  Loads are raw buffer loads (base + 32-bit lane offset + scalar offset): as 64-bit pointers the 32 addresses
// of a piece cost 64 registers.  All 32 loads of a piece are requested together (199 VGPRs, 8 waves per CU).  After the loop: one barrier, the NWV wave states + the NEW key (rotated from the qkv row, value from
// the qkv row) are merged by threads d < D, and one workgroup per kv head appends the new key / value to the caches.
#pragma once
#include "common.h"
#include "kernels.h"
#include "attention_decode.h"

namespace lmx {

constexpr int BA_PIECE = 64;      // keys per piece
typedef uint32_t ba_u32x4 __attribute__((ext_vector_type(4)));
// raw (stride 0) buffer view: a load is base + 32-bit lane offset + wave-uniform scalar offset.  (As 64-bit pointers hipcc kept the 32 per-load addresses of a
// piece alive across the loop: 64 registers, spilled in the two-phase form.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ba_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ ba_u32x4 ba_ld16(__amdgpu_buffer_rsrc_t rs, uint32_t lane_off, uint32_t wave_off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, wave_off, 0); }

// One (sequence, head) by the NWV waves of the calling workgroup.  (Round 4 also wrote a two-phase form — a piece's V^T lines requested only after its scores, half the
// registers, 16 waves — which measured equal at 8 sequences and slower at 32, r5-A; removed.)
// `rot`: wave w takes the pieces (w - rot) mod NWV, + NWV, ... and wave `rot` the new key — a workgroup that walks TWO heads gives the second one rot = NWV / 2, so that
// the waves with one piece more than the others are different ones for the two heads (17 pieces over 8 waves: 3 + 2 and 2 + 3 instead of 3 + 3).
// Leaves the wave states (o[D], m, l) and the new key's in `part`; attn_wave_merge turns them into the output row after a workgroup barrier.
template <typename T, int NWV>
__device__ __forceinline__ void attn_wave_stream(const T* __restrict__ qkv, T* Kc, T* Vt, const int pos, const int s_max, const float* cs,
                                                 const float scale, const int n_heads, const int n_kv_heads, const int head, const int rot,
                                                 float (&p_lds)[NWV][BA_PIECE], float (&part)[NWV + 1][128 + 2]) {
    constexpr int D = 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = n_heads / n_kv_heads;
    const int kvh = head / group;
    const __amdgpu_buffer_rsrc_t rsK = ba_rsrc(Kc), rsV = ba_rsrc(Vt);
    const T* qrow = qkv + head * D;
    const T* knew = qkv + (n_heads + kvh) * D;
    const T* vnew = qkv + (n_heads + n_kv_heads + kvh) * D;

    const int sub = lane & 15, kslot = lane >> 4;                           // K layout: 16 lanes per key (8 dims each), 4 keys per instruction
    const int s8 = lane & 7, drow8 = lane >> 3;                             // V^T layout: 8 lanes per d-row (8 keys each), 8 d-rows per instruction
    const float scl = scale * 1.4426950408889634f;                          // log2 domain

    float qv[8];
    rope8<T, D>(qrow, cs, sub * 8, qv);

    float m_run = -INFINITY, l_run = 0.f;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int n_piece = (pos + BA_PIECE - 1) / BA_PIECE;
    for (int pc = (wave - rot + NWV) % NWV; pc < n_piece; pc += NWV) {
        const int k0 = pc * BA_PIECE;
        const int nk = pos - k0 < BA_PIECE ? pos - k0 : BA_PIECE;          // >= 1
        // ---- the piece's loads first: K rows (a row past the last cached key re-reads the last one), then the V^T lines ----------------------
        ba_u32x4 kraw[16], vraw[16];
        const uint32_t k_wave = (uint32_t)k0 * (uint32_t)(D * sizeof(T));
        const uint32_t v_lane = ((uint32_t)drow8 * (uint32_t)s_max + (uint32_t)s8 * 8u) * (uint32_t)sizeof(T);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kl = 4 * u + kslot;
            kraw[u] = ba_ld16(rsK, (uint32_t)((kl < nk ? kl : nk - 1) * D + sub * 8) * (uint32_t)sizeof(T), k_wave);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) vraw[i] = ba_ld16(rsV, v_lane, ((uint32_t)(8 * i) * (uint32_t)s_max + (uint32_t)k0) * (uint32_t)sizeof(T));
        // ---- scores: after the butterfly all 16 lanes of a key hold its dot product; lane (kslot, sub) keeps the one of key 4 sub + kslot -----------------
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t w4[4] = {kraw[u][0], kraw[u][1], kraw[u][2], kraw[u][3]};
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dot = fmaf(qv[2 * c], unpack_lo<T>(w4[c]), dot);
                dot = fmaf(qv[2 * c + 1], unpack_hi<T>(w4[c]), dot);
            }
            dot += __shfl_xor(dot, 8, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 1, 64);
            if (u == sub) mine = dot;
        }
        if (nk < BA_PIECE) {
            // last, partial piece: the V^T lines run past the cached keys (stale or never-written columns).  Their probabilities are 0, but 0 x Inf / NaN is
            // NaN, so the values themselves are cleared (wave-uniform branch, once per head: keys 8 s8 + 2 c, + 1 of word c)
            uint32_t msk[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) msk[c] = (8 * s8 + 2 * c < nk ? 0x0000ffffu : 0u) | (8 * s8 + 2 * c + 1 < nk ? 0xffff0000u : 0u);
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) vraw[i][c] &= msk[c];
        }
        const int my_key = 4 * sub + kslot;                                 // this lane's key inside the piece
        const bool live = my_key < nk;
        const float s = live ? mine * scl : -INFINITY;
        // ---- online softmax over the piece (wave-wide statistics) --------------------------------------------------------------------------------------
        const float m_new = fmaxf(m_run, wave_max(s));                      // finite: the piece has at least one live key
        const float p = live ? __builtin_amdgcn_exp2f(s - m_new) : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // first piece: exp2(-inf) = 0 on zero accumulators
        l_run = l_run * alpha + wave_sum(p);
        m_run = m_new;
        p_lds[wave][my_key] = p;                                            // the wave's LDS operations retire in order: the reads below see these writes;
        __builtin_amdgcn_wave_barrier();                                    // the compiler must not move them above the writes of the OTHER lanes either
        const float4 p0 = *reinterpret_cast<const float4*>(&p_lds[wave][s8 * 8]);
        const float4 p1 = *reinterpret_cast<const float4*>(&p_lds[wave][s8 * 8 + 4]);
        // ---- o = o alpha + P V on this lane's d-rows (8 i + drow8) and keys (8 s8 .. + 8) --------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t w4[4] = {vraw[i][0], vraw[i][1], vraw[i][2], vraw[i][3]};
            float t = acc[i] * alpha;
            t = fmaf(p0.x, unpack_lo<T>(w4[0]), t); t = fmaf(p0.y, unpack_hi<T>(w4[0]), t);
            t = fmaf(p0.z, unpack_lo<T>(w4[1]), t); t = fmaf(p0.w, unpack_hi<T>(w4[1]), t);
            t = fmaf(p1.x, unpack_lo<T>(w4[2]), t); t = fmaf(p1.y, unpack_hi<T>(w4[2]), t);
            t = fmaf(p1.z, unpack_lo<T>(w4[3]), t); t = fmaf(p1.w, unpack_hi<T>(w4[3]), t);
            acc[i] = t;
        }
        __builtin_amdgcn_wave_barrier();                                    // next piece's probabilities are written only after every lane has read this one's
    }
    // ---- the wave's state: o[d] = sum over the 8 lanes of a d-row, then (o, m, l) into LDS ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float t = acc[i];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        if (s8 == 0) part[wave][8 * i + drow8] = t;
    }
    if (lane == 0) { part[wave][D] = m_run; part[wave][D + 1] = l_run; }
    // ---- the new key: rotated from the qkv row by wave `rot` (same arithmetic as decode_fused_body), its value straight from the qkv row ---------------------------
    if (wave == rot) {
        float kr[8];
        rope8<T, D>(knew, cs, sub * 8, kr);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) dot = fmaf(qv[c], kr[c], dot);
        dot += __shfl_xor(dot, 8, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 1, 64);
        if (lane == 0) { part[NWV][D] = dot * scl; part[NWV][D + 1] = 1.f; }
        if (head % group == 0 && kslot == 0) store8<T>(Kc + (size_t)pos * D + sub * 8, kr);          // one workgroup per kv head appends
    }
    if (tid >= 64 && tid < 64 + D) {
        const T v = vnew[tid - 64];
        part[NWV][tid - 64] = to_f32(v);
        if (head % group == 0) Vt[(size_t)(tid - 64) * s_max + pos] = v;
    }
}

// after a workgroup barrier: thread d < D merges the NWV wave states and the new key of one head (fixed order: deterministic)
template <typename T, int NWV>
__device__ __forceinline__ void attn_wave_merge(T* __restrict__ out, const int head, const int d, const float (&part)[NWV + 1][128 + 2]) {
    constexpr int D = 128;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w <= NWV; ++w) M = fmaxf(M, part[w][D]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w <= NWV; ++w) {
        const float m = part[w][D];
        if (m != -INFINITY) {
            const float f = __builtin_amdgcn_exp2f(m - M);
            l += f * part[w][D + 1];
            o += f * part[w][d];
        }
    }
    out[head * D + d] = from_f32<T>(l > 0.f ? o / l : 0.f);
}

// decode batch: grid (heads / HPW, sequences); K / V^T / position of sequence z from the per-layer table, q|k|v and output rows by stride.  HPW = 2 (taken when the
// launch has at least two rounds of workgroups anyway): a workgroup walks two heads back to back with rotated wave assignments and ONE barrier at the end —
// 34 pieces over 8 waves (max 5, mean 4.25) instead of twice 17 (max 3, mean 2.1): the waves' idle tails shrink from 29 % to 15 %.
template <typename T, int NWV, int HPW>
__global__ __launch_bounds__(NWV * 64) void decode_attn_wave_kernel(DecodeFusedArgs a) {
    static_assert(NWV >= 4 && NWV <= 16 && HPW >= 1 && HPW <= 4, "waves per workgroup: the cache append uses threads 64 .. 64 + D; 1 .. 4 heads per workgroup");
    __shared__ __attribute__((aligned(16))) float p_lds[NWV][BA_PIECE];      // wave-private: probabilities of the current piece, by key
    __shared__ float part[HPW][NWV + 1][128 + 2];                            // per head: wave states (o[D], m, l) + the new key's
    const int zseq = blockIdx.y;
    const DecodeFusedSeq e = a.tab[zseq];
    int pos = *e.pos_ptr;                                                   // keys [0, pos) are cached; this token's key goes to row pos
    pos = pos < a.s_max ? pos : a.s_max - 1;                                // the host keeps len + steps <= s_max; never index past the cache whatever the device word holds
    const T* qkv = reinterpret_cast<const T*>(a.QKV) + (size_t)zseq * a.qkv_stride;
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const int head = blockIdx.x * HPW + h;
        const int kvh = head / (a.n_heads / a.n_kv_heads);
        // head h starts its pieces at wave h NWV / HPW: the waves that get one piece more than the others are different ones for every head of the workgroup
        attn_wave_stream<T, NWV>(qkv, reinterpret_cast<T*>(e.K) + (size_t)kvh * a.s_max * 128, reinterpret_cast<T*>(e.VT) + (size_t)kvh * 128 * a.s_max, pos, a.s_max,
                                 a.cos_sin + (size_t)pos * 128, a.scale, a.n_heads, a.n_kv_heads, head, (h * NWV) / HPW, p_lds, part[h]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HPW * 128; i += NWV * 64)
        attn_wave_merge<T, NWV>(reinterpret_cast<T*>(a.O) + (size_t)zseq * a.o_stride, blockIdx.x * HPW + i / 128, i % 128, part[i / 128]);
}

// the batch form of launch_decode_fused through the kernel above.  A piece's 32 loads are requested together (209 VGPRs: two waves per SIMD), so a CU holds 8 waves.
// Round 6 A/B'd the workgroup forms at equal contexts (profiles/r06_batch_attn_forms.jsonl, interleaved medians in one process; ms per batched step at 8 / 16 / 24 / 32
// sequences): 8 waves x 1 head 3.75 / 5.19 / 6.51 / 7.76, 8 x 2 (round 5's form from 16 sequences on) 3.95 / 5.16 / 6.71 / 7.76, 4 x 2 4.01 / 5.10 / 6.59 / 7.62,
// **4 x 1 3.77 / 5.10 / 6.38 / 7.62**, 4 or 8 waves x 4 heads 25 - 30 % slower.  Two independent 4-wave workgroups per CU — one head each — drift apart, so one's
// merge tail and start-up run under the other's stream; walking several heads per workgroup only lengthens the tail.  One form everywhere: 4 waves, 1 head.
template <typename T>
inline void launch_decode_attn_wave_t(const DecodeFusedArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((decode_attn_wave_kernel<T, 4, 1>), dim3(a.n_heads, a.n_seq), dim3(4 * 64), 0, st, a);
}

}  // namespace lmx
