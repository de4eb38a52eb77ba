// Decode-batch attention, one workgroup per (sequence, head): the waves STREAM the head's keys (attention.hip includes this; opt-in LMX_BATCH_ATTN=1).
//
// STATUS: written and compiled at the end of round 4, NOT YET RUN ON A GPU (the round's GPU minutes were spent when the measurement that motivates it came in).
// The engine does not take it unless LMX_BATCH_ATTN=1; first thing to do with it: LMX_BATCH_ATTN=1 pytest tests/test_batching_gpu.py, then
// tools/mb_tp_batch_step.py 1 {8,32} with and without it.
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
// so some are always waiting on HBM while others multiply.  After the loop: one barrier, the NWV wave states + the NEW key (rotated from the qkv row, value from
// the qkv row) are merged by threads d < D, and one workgroup per kv head appends the new key / value to the caches.
#pragma once
#include "common.h"
#include "kernels.h"
#include "attention_decode.h"

namespace lmx {

constexpr int BA_PIECE = 64;      // keys per piece

template <typename T, int NWV>
__global__ __launch_bounds__(NWV * 64) void decode_attn_wave_kernel(DecodeFusedArgs a) {
    constexpr int D = 128;
    static_assert(NWV >= 3 && NWV <= 16, "waves per workgroup: the cache append uses threads 64 .. 64 + D");
    __shared__ __attribute__((aligned(16))) float p_lds[NWV][BA_PIECE];      // wave-private: probabilities of the current piece, by key
    __shared__ float part[NWV + 1][D + 2];                                   // wave states (o[D], m, l) + the new key's

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.x, zseq = blockIdx.y;
    const DecodeFusedSeq e = a.tab[zseq];
    const T* __restrict__ qkv = reinterpret_cast<const T*>(a.QKV) + (size_t)zseq * a.qkv_stride;
    T* __restrict__ out = reinterpret_cast<T*>(a.O) + (size_t)zseq * a.o_stride;
    const int group = a.n_heads / a.n_kv_heads;
    const int kvh = head / group;
    const int pos = *e.pos_ptr;                                             // keys [0, pos) are cached; this token's key goes to row pos
    T* Kc = reinterpret_cast<T*>(e.K) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(e.VT) + (size_t)kvh * D * a.s_max;
    const T* __restrict__ Kr = Kc;
    const T* __restrict__ Vr = Vt;
    const float* cs = a.cos_sin + (size_t)pos * D;
    const T* qrow = qkv + head * D;
    const T* knew = qkv + (a.n_heads + kvh) * D;
    const T* vnew = qkv + (a.n_heads + a.n_kv_heads + kvh) * D;

    const int sub = lane & 15, kslot = lane >> 4;                           // K layout: 16 lanes per key (8 dims each), 4 keys per instruction
    const int s8 = lane & 7, drow8 = lane >> 3;                             // V^T layout: 8 lanes per d-row (8 keys each), 8 d-rows per instruction
    const float scl = a.scale * 1.4426950408889634f;                        // log2 domain

    float qv[8];
    rope8<T, D>(qrow, cs, sub * 8, qv);

    float m_run = -INFINITY, l_run = 0.f;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int n_piece = (pos + BA_PIECE - 1) / BA_PIECE;
    for (int pc = wave; pc < n_piece; pc += NWV) {
        const int k0 = pc * BA_PIECE;
        const int nk = pos - k0 < BA_PIECE ? pos - k0 : BA_PIECE;          // >= 1
        // ---- every load of the piece first: K rows (a row past the last cached key re-reads the last one), then V^T lines -------------------------------
        uint4 kraw[16], vraw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kl = 4 * u + kslot;
            const int key = k0 + (kl < nk ? kl : nk - 1);
            kraw[u] = *reinterpret_cast<const uint4*>(Kr + (size_t)key * D + sub * 8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) vraw[i] = *reinterpret_cast<const uint4*>(Vr + (size_t)(8 * i + drow8) * a.s_max + k0 + s8 * 8);
        // ---- scores: after the butterfly all 16 lanes of a key hold its dot product; lane (kslot, sub) keeps the one of key 4 sub + kslot -----------------
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t w4[4] = {kraw[u].x, kraw[u].y, kraw[u].z, kraw[u].w};
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dot = fmaf(qv[2 * c], unpack_lo<T>(w4[c]), dot);
                dot = fmaf(qv[2 * c + 1], unpack_hi<T>(w4[c]), dot);
            }
            dot += __shfl_xor(dot, 8, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 1, 64);
            if (u == sub) mine = dot;
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
            const uint32_t w4[4] = {vraw[i].x, vraw[i].y, vraw[i].z, vraw[i].w};
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
    // ---- the new key: rotated from the qkv row by wave 0's first 16 lanes (same arithmetic as decode_fused_body), its value straight from the qkv row -----------
    if (wave == 0) {
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
        if (head % group == 0) Vt[(size_t)(tid - 64) * a.s_max + pos] = v;
    }
    __syncthreads();
    // ---- merge the NWV wave states and the new key ------------------------------------------------------------------------------------------------------------
    if (tid < D) {
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
                o += f * part[w][tid];
            }
        }
        out[head * D + tid] = from_f32<T>(l > 0.f ? o / l : 0.f);
    }
}

inline bool batch_attn_wave_on() {
    static const bool on = [] { const char* e = getenv("LMX_BATCH_ATTN"); return e && atoi(e) != 0; }();
    return on;
}

// the batch form of launch_decode_fused through the kernel above: 16-bit models with head_dim 128
template <typename T>
inline void launch_decode_attn_wave_t(const DecodeFusedArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((decode_attn_wave_kernel<T, 8>), dim3(a.n_heads, a.n_seq), dim3(8 * 64), 0, st, a);
}

}  // namespace lmx
