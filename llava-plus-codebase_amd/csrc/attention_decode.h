// Single-token decode attention body of decode_fused_kernel (attention.hip; batch / fp32 paths) + helpers shared with decode_attn.hip.
#pragma once
#include "common.h"
#include "kernels.h"

namespace lmx {

// rotated 8-element slice [i0, i0+8) of head vector x (pre-RoPE, length D) at position pos — HF rounding chain
template <typename T, int D>
__device__ __forceinline__ void rope8(const T* __restrict__ x, const float* __restrict__ cs, int i0, float (&out)[8]) {
    constexpr int HALF = D / 2;
    const bool lo = i0 < HALF;
    const int j0 = lo ? i0 : i0 - HALF;                 // index into the cos/sin half tables
    float a[8], b[8];
    load8<T>(x + i0, a);
    load8<T>(x + (lo ? i0 + HALF : i0 - HALF), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float c = round_to<T>(cs[j0 + e]), s = round_to<T>(cs[HALF + j0 + e]);
        const float rot = lo ? -b[e] : b[e];
        out[e] = rope_term<T>(a[e], c, rot, s);
    }
}

constexpr int DF_CHUNK = 128;     // keys per workgroup (fixed: every K / Vᵀ load of the chunk is issued up front)
constexpr int DF_MAX_SPLIT = 32;

// Body of decode_fused_kernel (attention.hip): the fp32 models' decode step, the decode batch of head_dim-64 / fp32 models, caches beyond 4096 slots.
template <typename T, int D>
__device__ __forceinline__ void decode_fused_body(DecodeFusedArgs a, const int head, const int split, const int zseq) {
    __shared__ __attribute__((aligned(16))) float sc_lds[DF_CHUNK];      // scores -> probabilities of this chunk
    __shared__ float red[8];                                              // reductions, new-key probability, merger flag
    __shared__ float mg_m[DF_MAX_SPLIT], mg_w[DF_MAX_SPLIT];              // merge: split maxima / weights
    __shared__ float mg_o[256];                                           // merge: cross-group partial sums

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.tab) {                                            // decode batch: this workgroup's sequence (uniform scalar loads)
        const DecodeFusedSeq e = a.tab[zseq];
        a.K = e.K; a.VT = e.VT; a.pos_ptr = e.pos_ptr; a.ws = e.ws; a.counters = e.counters;
        a.QKV = reinterpret_cast<const T*>(a.QKV) + (size_t)zseq * a.qkv_stride;
        a.O = reinterpret_cast<T*>(a.O) + (size_t)zseq * a.o_stride;
    }
    const int group = a.n_heads / a.n_kv_heads;
    const int kvh = head / group;
    const int pos = *a.pos_ptr;
    const int kv_len = pos + 1;
    const int k_begin = split * DF_CHUNK;
    int k_end = k_begin + DF_CHUNK; k_end = k_end < kv_len ? k_end : kv_len;
    const int nk = k_end > k_begin ? k_end - k_begin : 0;
    const bool has_new = pos >= k_begin && pos < k_end;    // this chunk owns the newest key
    const int nk_cached = has_new ? nk - 1 : nk;            // keys to read from the cache

    const T* __restrict__ qrow = reinterpret_cast<const T*>(a.QKV) + head * D;
    const T* __restrict__ knew = reinterpret_cast<const T*>(a.QKV) + (a.n_heads + kvh) * D;
    const T* __restrict__ vnew = reinterpret_cast<const T*>(a.QKV) + (a.n_heads + a.n_kv_heads + kvh) * D;
    T* Kc = reinterpret_cast<T*>(a.K) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(a.VT) + (size_t)kvh * D * a.s_max;
    const T* __restrict__ Kr = Kc;
    const T* __restrict__ Vr = Vt;
    const float* cs = a.cos_sin + (size_t)pos * D;
    constexpr int WS = D + 4;                               // partial row: o[D], m, l, pad, pad
    float* ws = a.ws + ((size_t)head * a.n_split + split) * WS;

    constexpr int LPK = D / 8, KPW = 64 / LPK;              // lanes per key, keys per wave-instruction
    constexpr int KU = DF_CHUNK / (4 * KPW);                // K loads per lane for the whole chunk
    constexpr int DB = D / 32;                              // Vᵀ rows per thread
    const float scl = a.scale * 1.4426950408889634f;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int s8 = tid & 7, drow = tid >> 3;

    float mx = -INFINITY, sum = 0.f;
    if (nk > 0) {
        // ---- issue every global load of this chunk first: K rows, then Vᵀ lines (they do not depend on the scores) -----
        float kv[KU][8];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = (u * 4 + wave) * KPW + kslot;
            const int key = k_begin + (kl < nk_cached ? kl : (nk_cached > 0 ? nk_cached - 1 : 0));
            load8<T>(Kr + (size_t)key * D + sub * 8, kv[u]);
        }
        float vv[2][DB][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int db = 0; db < DB; ++db) load8<T>(Vr + (size_t)(db * 32 + drow) * a.s_max + k_begin + kb * 64 + s8 * 8, vv[kb][db]);
        float qv[8];
        rope8<T, D>(qrow, cs, sub * 8, qv);

        // ---- scores ----------------------------------------------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = (u * 4 + wave) * KPW + kslot;
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kv[u][e], sdot);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
            if (sub == 0) sc_lds[kl] = kl < nk_cached ? sdot * scl : -INFINITY;
        }
        __syncthreads();
        // newest key: rotated straight from the qkv row; one workgroup per kv head appends it to the caches
        if (has_new && wave == 0 && kslot == 0) {
            float kr[8];
            rope8<T, D>(knew, cs, sub * 8, kr);
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kr[e], sdot);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
            if (sub == 0) sc_lds[nk - 1] = sdot * scl;
            if (head % group == 0) store8<T>(Kc + (size_t)pos * D + sub * 8, kr);
        }
        if (has_new && head % group == 0 && tid >= 64 && tid < 64 + D) Vt[(size_t)(tid - 64) * a.s_max + pos] = vnew[tid - 64];
        __syncthreads();

        // ---- softmax statistics (128 scores: one per thread of the first two waves) ---------------------------------------
        float sc = tid < DF_CHUNK ? sc_lds[tid] : -INFINITY;
        mx = block_max<4>(sc, red);
        float e = tid < DF_CHUNK ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;      // masked scores are -inf -> 0
        sum = block_sum<4>(e, red);
        if (has_new && tid == nk - 1) { red[4] = e; e = 0.f; }                   // newest key's value is added from registers
        if (tid < DF_CHUNK) sc_lds[tid] = e;
        __syncthreads();
        const float p_new = has_new ? red[4] : 0.f;

        // ---- o = P · V from the registers loaded above ---------------------------------------------------------------------
        float acc[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) acc[db] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 p0 = *reinterpret_cast<const float4*>(sc_lds + kb * 64 + s8 * 8);
            const float4 p1 = *reinterpret_cast<const float4*>(sc_lds + kb * 64 + s8 * 8 + 4);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                float t = acc[db];
                t = fmaf(p0.x, vv[kb][db][0], t); t = fmaf(p0.y, vv[kb][db][1], t); t = fmaf(p0.z, vv[kb][db][2], t); t = fmaf(p0.w, vv[kb][db][3], t);
                t = fmaf(p1.x, vv[kb][db][4], t); t = fmaf(p1.y, vv[kb][db][5], t); t = fmaf(p1.z, vv[kb][db][6], t); t = fmaf(p1.w, vv[kb][db][7], t);
                acc[db] = t;
            }
        }
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            float t = acc[db];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
            if (s8 == 0) {
                const int d = db * 32 + drow;
                if (has_new) t = fmaf(p_new, to_f32(vnew[d]), t);
                // write-through (agent-scope relaxed store = sc1): visible to the merging workgroup on any XCD
                if (a.debug_mode == 2) { if (t == 12345.678f) ws[d] = t; }
                else __hip_atomic_store(ws + d, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (a.debug_mode) return;
    if (tid == 0) {
        __hip_atomic_store(ws + D, nk > 0 ? mx : -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + D + 1, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- split merge by the last workgroup to arrive for this head (placement-independent counter protocol:
    //      sc1 payload stores -> every wave drains vmcnt -> barrier -> one relaxed agent-scope ticket; the merger reads
    //      the partials with agent-scope (sc1) loads, all issued in parallel) ---------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(a.counters + head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[5] = (ticket == a.n_split - 1) ? 1.f : 0.f;
    }
    __syncthreads();
    if (red[5] == 0.f) {
        return;
    }
    const float* wsh = a.ws + (size_t)head * a.n_split * WS;
    if (tid < a.n_split) {
        mg_m[tid] = __hip_atomic_load(wsh + tid * WS + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mg_w[tid] = __hip_atomic_load(wsh + tid * WS + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // l for now
    }
    constexpr int NG = 256 / D;                    // thread groups striding over the splits
    constexpr int SPG = DF_MAX_SPLIT / NG;         // splits per group (compile-time bound)
    const int g = tid / D, d = tid % D;
    float ov[SPG];
#pragma unroll
    for (int i = 0; i < SPG; ++i) {
        const int sp = g + i * NG;
        ov[i] = sp < a.n_split ? __hip_atomic_load(wsh + sp * WS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    __syncthreads();
    float M = -INFINITY;
    for (int sp = 0; sp < a.n_split; ++sp) M = fmaxf(M, mg_m[sp]);
    float l = 0.f;
    for (int sp = 0; sp < a.n_split; ++sp) { const float m = mg_m[sp]; if (m != -INFINITY) l += __builtin_amdgcn_exp2f(m - M) * mg_w[sp]; }
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < SPG; ++i) {
        const int sp = g + i * NG;
        if (sp < a.n_split) { const float m = mg_m[sp]; if (m != -INFINITY) o += __builtin_amdgcn_exp2f(m - M) * ov[i]; }
    }
    mg_o[tid] = o;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int i = 1; i < NG; ++i) o += mg_o[i * D + d];
        const T r = from_f32<T>(l > 0.f ? o / l : 0.f);
        reinterpret_cast<T*>(a.O)[head * D + d] = r;
    }
    if (tid == 0) __hip_atomic_store(a.counters + head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
}


}  // namespace lmx
