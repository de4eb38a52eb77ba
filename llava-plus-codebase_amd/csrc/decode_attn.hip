// Decode-step attention of a single request (16-bit models): RoPE(q, k_new) + KV-cache append + attention over the cached keys, split over 128-key
// chunks with the merge INSIDE the launch — the single-token branch of llava_arch.py:103-112 feeding HF5:models/llama/modeling_llama.py:191-214 (LlamaAttention
// with a one-row query; eager attention: softmax in fp32, output rounded to the model dtype).
//
// Two launches carry the same body (dec_attn_chunk):
//
//   decode_attn_step_kernel   grid = heads x live chunks.  q | k_new | v_new come from the q|k|v row a preceding GEMV launch wrote.
//
//   decode_kv_attn_kernel     (round 5, "split-q" decode step)  The step's q|k|v projection is cut in two: a first launch computes q alone (4096 of the 12288
//       rows at 7B); THIS launch holds the attention workgroups (lowest block ids: dispatched first) AND the workgroups of the k|v projection (gemv2_body).
//       Why: as its own launch the attention is a latency chain — 19 MB of K / V^T land after ~5 us, partial at ~7, merged row stored at ~13 (EXPERIMENTS
//       r3-D) — during which the HBM idles, and every form of "make the chain shorter" was measured equal (r3-D, r4-F, r4-G, r5-A).  Here the chain runs UNDER
//       the k|v weight stream (67 MB, ~13 us): the cached keys need only q, which is ready at launch; the newest key / value are the one thing that depends on
//       the k|v rows, and they reach the head's last chunk (the workgroup that owns position `pos`, which is also the one that appends to the caches) as
//       tagged 8-byte granules published by the GEMV workgroups of the same launch.  That workgroup takes its ticket last — its partial waits for the
//       granules — so it is the head's merger.  Forward progress: only the `heads` last-chunk workgroups ever wait, they wait for workgroups that never
//       wait themselves, and the wait is bounded (timeout -> status word, wrong data, no hang).
//       Arithmetic is dec_attn_chunk's in both launches and the GEMV rows are gemv2_body's: ids and logits are bit-identical to the three-launch form
//       (tests/test_decode_splitq_gpu.py).
//
// History: this file replaces decode_flow.hip (one launch per token without grid barriers, r3-B; attention + o_proj in one launch, r3-C; one workgroup per
// head, r4-F; tagged-granule merge, r3-D) — built, bit-identical, measured slower or equal, removed in round 5; profiles/EXPERIMENTS.md names the commits.
#include "attention_decode.h"
#include "common.h"
#include "gemv2.h"
#include "kernels.h"

namespace lmx {

namespace {

typedef uint32_t u32x4_w __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void unpack8w(const u32x4_w v, float (&f)[8]) {
    f[0] = unpack_lo<T>(v.x); f[1] = unpack_hi<T>(v.x); f[2] = unpack_lo<T>(v.y); f[3] = unpack_hi<T>(v.y);
    f[4] = unpack_lo<T>(v.z); f[5] = unpack_hi<T>(v.z); f[6] = unpack_lo<T>(v.w); f[7] = unpack_hi<T>(v.w);
}

constexpr uint64_t DA_TIMEOUT_TICKS = 3000000ull;          // s_memrealtime runs at 100 MHz: 30 ms

constexpr size_t dec_attn_smem_bytes(int D) { return (size_t)(DF_CHUNK + 8 + 2 * DF_MAX_SPLIT + 256) * 4 + (size_t)3 * D * 2 + 16; }

// One (head, 128-key chunk) workgroup of 256 threads.  Every chunk workgroup reduces its CACHED keys to a partial {o[D], max, sum} (decode_fused_body's score /
// softmax / P.V arithmetic on the chunk, position by value, every K / V^T load issued first, only live chunks launched).  The workgroup of the head's LAST chunk
// — the one that owns position `pos`; highest block id of the head, so everything it waits for was dispatched before it — is the head's MERGER: it waits for
// the other chunks' arrivals, merges the partials in chunk order, and only then folds in the NEWEST key as one more term:
//     s = scale (q . k_new),  M' = max(M, s),  out = (2^(M - M') o + 2^(s - M') v_new) / (2^(M - M') l + 2^(s - M'))
// so that everything except that last line is done before k_new / v_new are needed.  In the stand-alone launch they come with the row; in the split-q launch
// (SPLITQ) they arrive as tagged granules from the k | v projection's workgroups of the SAME launch, at its very end — the wait for them is the only part of
// the attention chain that is not hidden under the projection's weight stream.  Same arithmetic in both launches: bit-identical outputs.
template <typename T, int D, bool SPLITQ>
__device__ __forceinline__ void dec_attn_chunk(const DecAttnArgs& a, int item, char* smem) {
    float* sc_lds = reinterpret_cast<float*>(smem);                       // [DF_CHUNK] scores -> probabilities of this chunk
    float* red = sc_lds + DF_CHUNK;                                        // [8]
    float* mg_m = red + 8; float* mg_w = mg_m + DF_MAX_SPLIT;              // merge: split maxima / weights
    float* mg_o = mg_w + DF_MAX_SPLIT;                                     // [256] merge: cross-group partial sums; before that the merger's own partial in [0, D)
    T* qkv_s = reinterpret_cast<T*>(mg_o + 256);                           // [3 D] q | k_new | v_new of this head

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = item % a.nh, split = item / a.nh;
    const int group = a.nh / a.nkv;
    const int kvh = head / group;
    const int pos = a.pos;
    const int k_begin = split * DF_CHUNK;
    auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memrealtime(); };
    if (a.ts && tid == 0 && (head & 7) == 0) __hip_atomic_fetch_min(a.ts + 0, now(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool probe = a.ts && tid == 0 && head == 0 && split == a.n_split - 1;
    const bool merger = split == a.n_split - 1;                            // this chunk holds position `pos`
    int nk = pos - k_begin; nk = nk < DF_CHUNK ? nk : DF_CHUNK;            // CACHED keys of this chunk: 1 .. 128, the merger's 0 .. 127

    T* Kc = reinterpret_cast<T*>(a.kc) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(a.vt) + (size_t)kvh * D * a.s_max;
    const T* __restrict__ Kr = Kc;
    const T* __restrict__ Vr = Vt;
    const float* cs = a.rope + (size_t)pos * D;
    constexpr int WS = D + 4;
    float* ws = a.aws + ((size_t)head * a.n_split + split) * WS;

    constexpr int LPK = D / 8, KPW = 64 / LPK;
    constexpr int KU = DF_CHUNK / (4 * KPW);
    constexpr int DB = D / 32;
    const float scl = a.scale * 1.4426950408889634f;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int s8 = tid & 7, drow = tid >> 3;

    // ---- every global load of this chunk: K rows, then V^T lines (they depend on the position only) ----------------------------------------------------
    u32x4_w kraw[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int kl = (u * 4 + wave) * KPW + kslot;
        const int key = k_begin + (kl < nk ? kl : (nk > 0 ? nk - 1 : 0));
        kraw[u] = *reinterpret_cast<const u32x4_w*>(Kr + (size_t)key * D + sub * 8);
    }
    u32x4_w vraw[2][DB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int db = 0; db < DB; ++db) vraw[kb][db] = *reinterpret_cast<const u32x4_w*>(Vr + (size_t)(db * 32 + drow) * a.s_max + k_begin + kb * 64 + s8 * 8);

    // ---- q (| k_new | v_new) of this head from the q|k|v row an earlier LAUNCH wrote ---------------------------------------------------------------------
    if (tid < ((SPLITQ || !merger) ? 1 : 3) * D / 8) {
        const int part = tid / (D / 8), c = tid % (D / 8);
        const int col = (part == 0 ? head : part == 1 ? a.nh + kvh : a.nh + a.nkv + kvh) * D + c * 8;
        *reinterpret_cast<u32x4_w*>(qkv_s + part * D + c * 8) = *reinterpret_cast<const u32x4_w*>(reinterpret_cast<const T*>(a.qkv) + col);
    }
    __syncthreads();
    const T* qrow = qkv_s; const T* knew = qkv_s + D; const T* vnew = qkv_s + 2 * D;

    float qv[8];
    rope8<T, D>(qrow, cs, sub * 8, qv);
    // ---- scores of the cached keys --------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int kl = (u * 4 + wave) * KPW + kslot;
        float kv[8]; unpack8w<T>(kraw[u], kv);
        float sdot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kv[e], sdot);
#pragma unroll
        for (int o = LPK / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
        if (sub == 0) sc_lds[kl] = kl < nk ? sdot * scl : -INFINITY;
    }
    __syncthreads();
    // ---- softmax statistics (128 scores: one per thread of the first two waves); a chunk without cached keys (the merger at pos % 128 == 0) is empty: max -inf, sum 0
    const float sc = tid < DF_CHUNK ? sc_lds[tid] : -INFINITY;
    const float mx = block_max<4>(sc, red);
    const float e = sc != -INFINITY ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;
    const float sum = block_sum<4>(e, red);
    if (tid < DF_CHUNK) sc_lds[tid] = e;
    __syncthreads();
    // ---- o = P · V from the registers loaded above; beyond the cached keys P is 0 and the V^T bytes are cleared (never 0 x NaN) ------------------------------
    float acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) acc[db] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const float4 p0 = *reinterpret_cast<const float4*>(sc_lds + kb * 64 + s8 * 8);
        const float4 p1 = *reinterpret_cast<const float4*>(sc_lds + kb * 64 + s8 * 8 + 4);
        const int key0 = kb * 64 + s8 * 8;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u32x4_w raw = vraw[kb][db];
            if (key0 + 8 > nk) {                                           // only the chunk that holds `pos`
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[c] &= (key0 + 2 * c < nk ? 0x0000ffffu : 0u) | (key0 + 2 * c + 1 < nk ? 0xffff0000u : 0u);
            }
            float vv[8]; unpack8w<T>(raw, vv);
            float t = acc[db];
            t = fmaf(p0.x, vv[0], t); t = fmaf(p0.y, vv[1], t); t = fmaf(p0.z, vv[2], t); t = fmaf(p0.w, vv[3], t);
            t = fmaf(p1.x, vv[4], t); t = fmaf(p1.y, vv[5], t); t = fmaf(p1.z, vv[6], t); t = fmaf(p1.w, vv[7], t);
            acc[db] = t;
        }
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        float t = acc[db];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        if (s8 == 0) {
            const int d = db * 32 + drow;
            if (merger) mg_o[d] = t;                                       // the merger keeps its own partial in LDS
            else __hip_atomic_store(ws + d, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through: the merger may sit on any XCD
        }
    }
    if (!merger) {
        if (tid == 0) {
            __hip_atomic_store(ws + D, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws + D + 1, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // arrival: every wave's write-through stores are acknowledged, then ONE relaxed agent-scope add on the head's counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.cnt + head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.ts && tid == 0 && (head & 7) == 0) __hip_atomic_fetch_max(a.ts + 9, now(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    // ---- the merger: wait (bounded) for the other chunks' arrivals, merge in chunk order ------------------------------------------------------------------
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    int* flag = reinterpret_cast<int*>(red + 6);
    bool timed_out = false;
    __syncthreads();                                                       // own partial complete in mg_o[0, D)
    if (probe) a.ts[1] = now();
    if (a.n_split > 1) {
        if (tid == 0) {
            int ok = 1, it = 0;
            while (__hip_atomic_load(a.cnt + head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.n_split - 1) {
                __builtin_amdgcn_s_sleep(2);
                if ((++it & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > DA_TIMEOUT_TICKS) { ok = 0; break; }
            }
            *flag = ok;
        }
        __syncthreads();
        timed_out = *flag == 0;
    }
    const bool arrivals_missing = timed_out;
    if (probe) a.ts[2] = now();
    const float* wsh = a.aws + (size_t)head * a.n_split * WS;
    const int n_other = a.n_split - 1;
    if (tid < n_other) {
        mg_m[tid] = __hip_atomic_load(wsh + tid * WS + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mg_w[tid] = __hip_atomic_load(wsh + tid * WS + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == n_other) { mg_m[tid] = mx; mg_w[tid] = sum; }               // this chunk is split n_split - 1
    constexpr int NG = 256 / D;
    constexpr int SPG = DF_MAX_SPLIT / NG;
    const int g = tid / D, d = tid % D;
    float ov[SPG];
#pragma unroll
    for (int i = 0; i < SPG; ++i) {
        const int s2 = g + i * NG;
        ov[i] = s2 < n_other ? __hip_atomic_load(wsh + s2 * WS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (s2 == n_other ? mg_o[d] : 0.f);
    }
    __syncthreads();                                                       // mg_m / mg_w complete; everyone has read its own-partial value out of mg_o
    float M = -INFINITY;
    for (int s2 = 0; s2 < a.n_split; ++s2) M = fmaxf(M, mg_m[s2]);
    float l = 0.f;
    for (int s2 = 0; s2 < a.n_split; ++s2) { const float m = mg_m[s2]; if (m != -INFINITY) l += __builtin_amdgcn_exp2f(m - M) * mg_w[s2]; }
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < SPG; ++i) {
        const int s2 = g + i * NG;
        if (s2 < a.n_split) { const float m = mg_m[s2]; if (m != -INFINITY) o += __builtin_amdgcn_exp2f(m - M) * ov[i]; }
    }
    mg_o[tid] = o;
    // re-arm for the next launch (every arrival has been seen).  NOT after a timeout: late arrivals may still be counting — the host clears the tickets when it
    // reports the status word (Seq::check_wait_status)
    if (tid == 0 && !arrivals_missing) __hip_atomic_store(a.cnt + head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if (probe) a.ts[3] = now();
    // ---- the newest key / value ----------------------------------------------------------------------------------------------------------------------------
    if constexpr (SPLITQ) {
        // k_new / v_new of this kv head arrive from the k | v projection's workgroups of THIS launch (gemv2_body<PUBLISH>), one 8-byte {bits, tag} granule per
        // row; 2 D threads poll one granule each (relaxed agent-scope loads bypass the L1) until all carry this launch's tag
        const bool mine = tid < 2 * D;
        const unsigned long long* src = a.kv_gran + (tid < D ? (size_t)kvh * D + tid : (size_t)(a.nkv + kvh) * D + (tid - D));
        for (int it = 0;; ++it) {
            unsigned long long gr = 0; bool ok = true;
            if (mine) { gr = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = (unsigned)(gr >> 32) == a.tag; }
            if (__syncthreads_and(ok ? 1 : 0)) {
                if (mine) reinterpret_cast<unsigned short*>(qkv_s)[D + tid] = (unsigned short)gr;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
            if ((it & 63) == 63) {                                         // bounded: one thread reads the clock, everybody leaves together
                if (tid == 0) *flag = (__builtin_amdgcn_s_memrealtime() - t0 > DA_TIMEOUT_TICKS) ? 0 : 1;
                __syncthreads();
                const int give_up = *flag == 0;
                __syncthreads();
                if (give_up) {
                    if (mine) reinterpret_cast<unsigned short*>(qkv_s)[D + tid] = 0;
                    timed_out = true;
                    break;
                }
            }
        }
    }
    if (timed_out && tid == 0 && a.status) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();                                                       // k_new | v_new in LDS (SPLITQ), mg_o complete
    if (probe) a.ts[4] = now();
    if (wave == 0 && kslot == 0) {                                         // rotated straight from the row; one workgroup per kv head appends it to the cache
        float kr[8];
        rope8<T, D>(knew, cs, sub * 8, kr);
        float sdot = 0.f;
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) sdot = fmaf(qv[e2], kr[e2], sdot);
#pragma unroll
        for (int o2 = LPK / 2; o2 > 0; o2 >>= 1) sdot += __shfl_xor(sdot, o2, 64);
        if (sub == 0) red[4] = sdot * scl;
        if (head % group == 0) store8<T>(Kc + (size_t)pos * D + sub * 8, kr);
    }
    if (head % group == 0 && tid >= 64 && tid < 64 + D) Vt[(size_t)(tid - 64) * a.s_max + pos] = vnew[tid - 64];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int i = 1; i < NG; ++i) o += mg_o[i * D + d];
        const float s_new = red[4];
        const float M2 = fmaxf(M, s_new);
        const float wa = M != -INFINITY ? __builtin_amdgcn_exp2f(M - M2) : 0.f, wb = __builtin_amdgcn_exp2f(s_new - M2);
        const float num = fmaf(wb, to_f32(vnew[d]), wa * o), den = fmaf(wa, l, wb);
        reinterpret_cast<T*>(a.attn)[head * D + d] = from_f32<T>(num / den);      // read by the NEXT launch (o_proj): a plain store
    }
    if (probe) a.ts[5] = now();
    if (a.ts && tid == 0 && (head & 7) == 0) __hip_atomic_fetch_max(a.ts + 8, now(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void decode_attn_step_kernel(DecAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dec_attn_chunk<T, D, false>(a, (int)blockIdx.x, smem);
}

// attention workgroups [0, n_attn) + the k|v projection's workgroups behind them (R = 2 rows per wave, P = 4 rounds in flight: the q|k|v shape's setting)
template <typename T, int D, int NX>
__global__ __launch_bounds__(256) void decode_kv_attn_kernel(DecAttnArgs a, GemvArgs g, int n_attn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n_attn) dec_attn_chunk<T, D, true>(a, (int)blockIdx.x, smem);
    else gemv2_body<T, 2, 4, NX, true>(g, (int)blockIdx.x - n_attn, smem, a.kv_gran, a.pub_tag ? a.pub_tag : a.tag, a.ts);
}

void check_dec_attn(int dtype, int D, const DecAttnArgs& a, const char* who) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, std::string(who) + ": 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, std::string(who) + ": head_dim must be 64 or 128");
    LMX_REQUIRE(a.pos >= 0 && a.pos < a.s_max && a.nkv >= 1 && a.nh % a.nkv == 0, std::string(who) + ": 0 <= pos < s_max, heads a multiple of kv heads");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK > a.pos && (a.n_split - 1) * DF_CHUNK <= a.pos && a.n_split * DF_CHUNK <= a.s_max,
                std::string(who) + ": n_split must be the number of live 128-key chunks (pos / 128 + 1, at most 32) and s_max a multiple of 128");
}

}  // namespace

void launch_decode_attn_step(int dtype, int D, const DecAttnArgs& a, hipStream_t st) {
    check_dec_attn(dtype, D, a, "decode_attn_step");
    const size_t smem = dec_attn_smem_bytes(D);
#define LA(TT, DD) LMX_LAUNCH((decode_attn_step_kernel<TT, DD>), dim3((unsigned)(a.nh * a.n_split)), dim3(256), smem, st, a)
    if (dtype == kBF16) { if (D == 128) LA(bf16_t, 128); else LA(bf16_t, 64); }
    else { if (D == 128) LA(f16_t, 128); else LA(f16_t, 64); }
#undef LA
    LMX_CHECK_HIP(hipGetLastError());
}

bool decode_kv_attn_applies(int dtype, int D, const GemvArgs& g) {
    return (D == 64 || D == 128) && gemv2_applies(dtype, g) && g.K <= 8192 && g.N % 8 == 0 && !g.bias && !g.R && g.act == kActNone;
}

// Workgroups of the split-q launch the device can hold at once (occupancy of the kernel x compute units of THIS device: CU masking and smaller parts included).
// The engine takes the split-q form only while every sequence's waiters (one per head) together fill at most half of them (Model::splitq_allowed).
int decode_kv_attn_resident_slots(int dtype, int D, int K) {
    const size_t smem_g = gemv2_smem_bytes(K, 2), smem_a = dec_attn_smem_bytes(D);
    const size_t smem = smem_g > smem_a ? smem_g : smem_a;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    LMX_CHECK_HIP(hipGetDevice(&dev));
    LMX_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
#define OC(TT, DD, NX) LMX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_kv_attn_kernel<TT, DD, NX>, 256, smem))
#define OCD(TT, DD) do { if (K <= 4096) OC(TT, DD, 2); else OC(TT, DD, 4); } while (0)
    if (dtype == kBF16) { if (D == 128) OCD(bf16_t, 128); else OCD(bf16_t, 64); }
    else { if (D == 128) OCD(f16_t, 128); else OCD(f16_t, 64); }
#undef OCD
#undef OC
    return per_cu * prop.multiProcessorCount;
}

void launch_decode_kv_attn(int dtype, int D, const DecAttnArgs& a, const GemvArgs& g, hipStream_t st) {
    check_dec_attn(dtype, D, a, "decode_kv_attn");
    LMX_REQUIRE(decode_kv_attn_applies(dtype, D, g), "decode_kv_attn: the k|v projection must be a plain 16-bit linear the hand-counted stream takes, K <= 8192");
    LMX_REQUIRE(g.N == 2 * a.nkv * D && a.kv_gran && a.tag != 0, "decode_kv_attn: the projection's rows are k | v of every kv head; granules and a non-zero tag");
    const size_t smem_g = gemv2_smem_bytes(g.K, 2), smem_a = dec_attn_smem_bytes(D);
    const size_t smem = smem_g > smem_a ? smem_g : smem_a;
    const int n_attn = a.nh * a.n_split;
    // projection part: 2 rows per wave, 4 rounds in flight — the q|k|v shape's setting; (4, 4) / (4, 2) / (2, 8) / (1, 8) measured 21.1 -> 22.5 / 23.8 / 21.5 / 22.9 us
    // per launch (profiles/r05_kv_attn_rp.txt): the launch is bound by its 86 MB at the streaming rate, not by workgroup count or bytes in flight
    const dim3 grid((unsigned)(n_attn + cdiv(g.N, 8)));
#define LK(TT, DD, NX) LMX_LAUNCH((decode_kv_attn_kernel<TT, DD, NX>), grid, dim3(256), smem, st, a, g, n_attn)
#define LKD(TT, DD) do { if (g.K <= 4096) LK(TT, DD, 2); else LK(TT, DD, 4); } while (0)
    if (dtype == kBF16) { if (D == 128) LKD(bf16_t, 128); else LKD(bf16_t, 64); }
    else { if (D == 128) LKD(f16_t, 128); else LKD(f16_t, 64); }
#undef LKD
#undef LK
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
