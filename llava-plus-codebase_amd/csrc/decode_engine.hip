// Persistent decode step with data-tagged hand-overs: ONE launch per generated token for a single sequence (tensor-parallel world 1).
//
// What it replaces: the 161 launches of Model::decode_step_launch — per layer {qkv GEMV (+RMSNorm), RoPE + KV append + split attention, o_proj GEMV
// (+residual), gate|up GEMV (+RMSNorm, SiLU*mul), down GEMV (+residual)} and the lm_head GEMV (+final norm): the decoder half of LlamaModel.forward for
// one new token (HF5:models/llama/modeling_llama.py:367-418 via llava_llama.py:88-99).  The pick kernel stays a second launch.
//
// Third attempt at removing the kernel boundaries of a decode step (EXPERIMENTS.md r2-Q: grid barriers, 3.7-3.9 ms; r3-B: dataflow launch with
// completion counters, 3.15 ms; separate launches 2.88 ms).  What the first two taught: (1) the weight stream needs hand-counted waits and >= 16 loads
// per wave on the wire (wstream.h), (2) a hand-over through a counter is >= 4 dependent memory round trips (drain, atomic, poll, read).  So here:
//   * G = 4 workgroups per CU stay resident for the whole token (grid sized by the occupancy query); every wave walks the steps.
//   * A step's rows are split EVENLY over the first n_part waves (n_part chosen per step so that the slots divide: the stream is HBM-bound, 2048-3072
//     waves x 16 KiB on the wire saturate it); a wave's slots form one hand-counted stream; the first P rounds of a step go out BEFORE its input row
//     is gathered, so the hand-over runs under loads that are already on the wire.
//   * Hand-over = the data itself.  An activation row is an array of 8-byte granules {two 16-bit elements, tag}, tag = launch number x steps + step,
//     written with ONE sc1 store each (fire and forget: no drain, no barrier, no atomic) and gathered by every consuming workgroup with sc1 loads that
//     are retried until every tag matches (MI355X_MICROARCH "handoff-1to1": data-tagged granules are the cheapest cross-CU primitive).  The chain is
//     store -> gather instead of store -> vmcnt(0) -> barrier -> atomic -> poll -> check -> read.
//   * Attention: flow_attn2's body (decode_flow.hip): K / V^T of the chunk requested before the q | k | v slices are gathered, partials as tagged
//     granules merged by the head's last chunk.
//   * Every retry loop is bounded (~50 ms): a timeout raises the status word the host checks and the workgroup carries on.
// Arithmetic: the per-lane accumulation order, rounding points and reductions of gemv_kernel / decode_fused_kernel, so ids and logits are bit-identical
// to the separate launches (tests/test_decode_engine_gpu.py).
#include "attention_decode.h"
#include "common.h"
#include "kernels.h"
#include "wstream.h"

namespace lmx {

namespace {

typedef uint32_t u32x4_e __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr uint64_t ENG_TIMEOUT_TICKS = 5000000ull;        // s_memrealtime runs at 100 MHz: 50 ms

__device__ __forceinline__ __amdgpu_buffer_rsrc_t eng_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4_e eng_ld16_coh(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, /*sc1*/ 16);
}
__device__ __forceinline__ void eng_timeout(const EngArgs& a, int code) {
    __hip_atomic_store(a.abort_word, (unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.status, (unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- gather an input row into LDS ----------------------------------------------------------------------------------------------------------------------
// Source: a granule array (gran != null: two elements + tag per 8 bytes, retried until every tag == `tag`) or a plain row an earlier LAUNCH wrote.
// Thread t owns the 8-element chunks t, t + 256, ... (gemv_kernel's mapping, so the RMSNorm statistics add up in the same order); with norm_w the row
// is normalised in place (HF rounding points).  All 256 threads call it.
template <typename T>
__device__ __forceinline__ void eng_gather(const EngArgs& a, const u64* gran, const T* plain, unsigned tag, int K, const T* norm_w, T* xs, float* red, int code) {
    const int tid = threadIdx.x, KC = K >> 3;
    constexpr int MAXC = 8;                                                // chunks per thread: K <= 16384
    __syncthreads();                                                       // every wave of the workgroup is done with the previous step's row in LDS
    if (gran) {
        const __amdgpu_buffer_rsrc_t rg = eng_rsrc(gran);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int* toflag = reinterpret_cast<int*>(red + 6);
        for (int it = 0;; ++it) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = tid + 256 * i;
                if (c < KC) {
                    const u32x4_e g0 = eng_ld16_coh(rg, (uint32_t)c * 32u), g1 = eng_ld16_coh(rg, (uint32_t)c * 32u + 16u);
                    ok = ok && g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag;
                    u32x4_e v; v.x = g0.x; v.y = g0.z; v.z = g1.x; v.w = g1.z;
                    *reinterpret_cast<u32x4_e*>(xs + c * 8) = v;          // (rewritten on a retry; nobody reads xs before the verdict below)
                }
            }
            if (__syncthreads_and(ok ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((it & 63) == 63) {                                         // bounded: one thread reads the clock, everybody leaves together
                if (tid == 0) *toflag = (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS || __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
                __syncthreads();
                const int timed_out = *toflag;
                __syncthreads();
                if (timed_out) { if (tid == 0) eng_timeout(a, code); break; }
            }
        }
    } else {
        for (int c = tid; c < KC; c += 256) *reinterpret_cast<u32x4_e*>(xs + c * 8) = *reinterpret_cast<const u32x4_e*>(plain + c * 8);
    }
    if (norm_w) {
        float ss = 0.f;
        for (int c = tid; c < KC; c += 256) {                              // each thread re-reads exactly the chunks it wrote
            float v[8]; load8<T>(xs + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
        }
        ss = block_sum<4>(ss, red);
        const float inv = rsqrtf(ss / (float)K + a.eps);
        for (int c = tid; c < KC; c += 256) {
            float v[8], gv[8];
            load8<T>(xs + c * 8, v);
            load8<T>(norm_w + c * 8, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * gv[e];
            store8<T>(xs + c * 8, v);
        }
    }
    __syncthreads();
}

// one granule = elements (n, n + 1) of a row + tag; n even
template <typename T> __device__ __forceinline__ void eng_put2(u64* gran, int n, float lo, float hi, unsigned tag) {
    const uint32_t data = pack2<T>(lo, hi);
    __hip_atomic_store(gran + (n >> 1), ((u64)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ float eng_get1(const u64* gran, int n) {      // element n of a COMPLETE granule row (no tag check: see callers)
    const u64 g = __hip_atomic_load(gran + (n >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t d = (uint32_t)g;
    return (n & 1) ? unpack_hi<T>(d) : unpack_lo<T>(d);
}

// ---- one linear step: out = act(norm(x) W^T) (+ residual); this wave owns the slots gw, gw + n_part, ... (slot = R rows) ---------------------------------------
template <typename T, int R, int P, bool SILU>
__device__ __forceinline__ void eng_linear(const EngArgs& a, const EngStep& sp, int step, char* smem) {
    T* xs = reinterpret_cast<T*>(smem);
    float* red = reinterpret_cast<float*>(smem + a.xs_bytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * 4 + wave;                                  // global wave id
    const int N = sp.N, K = sp.K, KC = K >> 3;
    const int NR = (KC + 63) >> 6;                                         // load rounds per slot
    const int nslots = (N + R - 1) / R;
    const int stride = sp.n_part;                                          // participating waves
    const int nloc = (gw < stride && gw < nslots) ? (nslots - gw + stride - 1) / stride : 0;        // slots of this wave (wave-uniform)
    const ws_v4i rw = ws_make_rsrc(sp.W, 0x7fffffffu);
    auto row_off = [&](int sl, int r) -> uint32_t {
        int f;
        if (SILU) { const int j = (sl * R + r) >> 1; f = 64 * (j >> 5) + (j & 31) + 32 * ((sl * R + r) & 1); }
        else f = sl * R + r;
        f = f < N ? f : N - 1;
        return (uint32_t)__builtin_amdgcn_readfirstlane(f) * (uint32_t)K * (uint32_t)sizeof(T);
    };
    ws_u32x4 buf[P][R];
    int i_sl = gw, i_j = 0, i_left = nloc;                                 // issue cursor (wave-uniform)
    auto issue = [&](int p) {
        const bool live = i_left > 0;
        const int c = lane + 64 * i_j;
        const uint32_t vo = live ? (uint32_t)(c < KC ? c : KC - 1) * 16u : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) ws_load(buf[p][r], vo, rw, live ? row_off(i_sl, r) : 0u);
        if (live && ++i_j == NR) { i_j = 0; i_sl += stride; --i_left; }
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue(p);                                  // on the wire while the input row is gathered
    const bool probe = a.ts && blockIdx.x == (unsigned)a.probe_block && tid == 0;
    if (probe) a.ts[3 * step] = __builtin_amdgcn_s_memrealtime();

    eng_gather<T>(a, reinterpret_cast<const u64*>(sp.x_gran), reinterpret_cast<const T*>(sp.x_plain), a.tag0 + (unsigned)sp.x_step, K,
                  reinterpret_cast<const T*>(sp.norm_w), xs, red, step + 1);

    if (probe) a.ts[3 * step + 1] = __builtin_amdgcn_s_memrealtime();
    constexpr int OPS = SILU ? R / 2 : R;                                  // outputs per slot (even: one granule per pair)
    static_assert(OPS % 2 == 0, "granules hold element pairs");
    float keep = 0.f;
    {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        int c_j = 0, c_n = 0;
        const int total = nloc * NR;
        for (int q0 = 0; q0 < total; q0 += P) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (q0 + p < total) {
                    ws_wait<R * (P - 1), R>(buf[p]);
                    const int cc = lane + 64 * c_j;
                    if (cc < KC) {
                        float xv[8]; load8<T>(xs + cc * 8, xv);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float wv[8]; ws_unpack8<T>(buf[p][r], wv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[r] = fmaf(wv[e], xv[e], acc[r]);
                        }
                    }
                    issue(p);
                    if (++c_j == NR) {
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
                        if (SILU) {
#pragma unroll
                            for (int r = 0; r < R; r += 2) if (lane == c_n * OPS + r / 2) keep = act_silu(acc[r]) * acc[r + 1];
                        } else {
#pragma unroll
                            for (int r = 0; r < R; ++r) if (lane == c_n * OPS + r) keep = acc[r];
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = 0.f;
                        c_j = 0; ++c_n;
                    }
                }
            }
        }
    }
    ws_drain<P, R>(buf);
    if (probe) a.ts[3 * step + 2] = __builtin_amdgcn_s_memrealtime();
    // lane k holds output k % OPS of this wave's slot k / OPS: add the residual, round, and let the even lane of a pair publish the granule
    {
        const int k = lane;
        const bool mine = k < nloc * OPS;
        const int sl = gw + (k / OPS) * stride;
        const int n = sl * OPS + k % OPS;                                  // SiLU*mul: index into the [N / 2] output
        const int NO = SILU ? N / 2 : N;
        float v = keep;
        if (mine && n < NO && !SILU && (sp.res_gran || sp.res_plain)) {
            // the residual row is complete: its producer finished before this step's INPUT existed (o_proj: the stream before the qkv step; down: o_proj)
            v += sp.res_gran ? eng_get1<T>(reinterpret_cast<const u64*>(sp.res_gran), n) : to_f32(reinterpret_cast<const T*>(sp.res_plain)[n]);
        }
        v = round_to<T>(v);
        const float hi = __shfl_down(v, 1, 64);
        if (mine && n < NO && (k & 1) == 0) {
            if (sp.out_gran) eng_put2<T>(reinterpret_cast<u64*>(sp.out_gran), n, v, hi, a.tag0 + (unsigned)step);
            else {                                                         // last step: plain logits for the pick kernel (next launch)
                T* C = reinterpret_cast<T*>(sp.out_plain);
                C[n] = from_f32<T>(v);
                if (n + 1 < NO) C[n + 1] = from_f32<T>(hi);
            }
        }
    }
}

// ---- attention step: flow_attn2's arithmetic (decode_flow.hip), q | k | v from the qkv granule row, output as granules -----------------------------------------
template <typename T, int D>
__device__ __forceinline__ void eng_attn(const EngArgs& a, const EngStep& sp, int item, int step, char* smem) {
    float* sc_lds = reinterpret_cast<float*>(smem);
    float* red = sc_lds + DF_CHUNK;
    float* mg_m = red + 8; float* mg_w = mg_m + DF_MAX_SPLIT;
    float* mg_o = mg_w + DF_MAX_SPLIT;
    T* qkv_s = reinterpret_cast<T*>(mg_o + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = item % a.nh, split = item / a.nh;
    const int group = a.nh / a.nkv;
    const int kvh = head / group;
    const int pos = a.pos;
    const int kv_len = pos + 1;
    const int k_begin = split * DF_CHUNK;
    int k_end = k_begin + DF_CHUNK; k_end = k_end < kv_len ? k_end : kv_len;
    const int nk = k_end - k_begin;
    const bool has_new = pos >= k_begin && pos < k_end;                    // == (split == n_split - 1): this workgroup also merges the head
    const int nk_cached = has_new ? nk - 1 : nk;
    const unsigned tag = a.tag0 + (unsigned)step;

    T* Kc = reinterpret_cast<T*>(sp.kc) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(sp.vt) + (size_t)kvh * D * a.s_max;
    constexpr int WSG = D + 4;
    u64* wsg = reinterpret_cast<u64*>(a.aws) + ((size_t)head * a.n_split + split) * WSG;

    constexpr int LPK = D / 8, KPW = 64 / LPK;
    constexpr int KU = DF_CHUNK / (4 * KPW);
    constexpr int DB = D / 32;
    const float scl = a.scale * 1.4426950408889634f;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int s8 = tid & 7, drow = tid >> 3;

    const ws_v4i rsK = ws_make_rsrc(Kc, (uint32_t)((size_t)a.s_max * D * sizeof(T)));
    const ws_v4i rsV = ws_make_rsrc(Vt, (uint32_t)((size_t)a.s_max * D * sizeof(T)));
    const ws_v4i rsC = ws_make_rsrc(a.rope + (size_t)pos * D, (uint32_t)(D * 4));
    ws_u32x4 kraw[KU], vraw[2 * DB], csr[4];
    // the KV chunk and this lane's cos / sin runs go on the wire before the qkv row is gathered
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int kl = (u * 4 + wave) * KPW + kslot;
        const int key = k_begin + (kl < nk_cached ? kl : (nk_cached > 0 ? nk_cached - 1 : 0));
        ws_load_plain(kraw[u], (uint32_t)(key * D + sub * 8) * (uint32_t)sizeof(T), rsK, 0u);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
            ws_load_plain(vraw[kb * DB + db], (uint32_t)((db * 32 + drow) * a.s_max + k_begin + kb * 64 + s8 * 8) * (uint32_t)sizeof(T), rsV, 0u);
    constexpr int HALF = D / 2;
    const int i0 = sub * 8, j0 = i0 < HALF ? i0 : i0 - HALF;
    ws_load_plain(csr[0], (uint32_t)j0 * 4u, rsC, 0u);
    ws_load_plain(csr[1], (uint32_t)j0 * 4u + 16u, rsC, 0u);
    ws_load_plain(csr[2], (uint32_t)(HALF + j0) * 4u, rsC, 0u);
    ws_load_plain(csr[3], (uint32_t)(HALF + j0) * 4u + 16u, rsC, 0u);

    const bool probe = a.ts && blockIdx.x == (unsigned)a.probe_block && tid == 0;
    if (probe) a.ts[3 * step] = __builtin_amdgcn_s_memrealtime();
    // ---- q | k_new | v_new of this head from the qkv granule row (threads 0 .. 3 D / 8 - 1: 8 elements = 4 granules each), retried until tagged ------------
    __syncthreads();                                                       // every wave of the workgroup is done with the previous step's LDS
    {
        const __amdgpu_buffer_rsrc_t rq = eng_rsrc(sp.x_gran);
        const unsigned qtag = a.tag0 + (unsigned)sp.x_step;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int* toflag = reinterpret_cast<int*>(red + 6);
        const int t = tid < 3 * D / 8 ? tid : 0;
        const int part = t / (D / 8), c = t % (D / 8);
        const int col = (part == 0 ? head : part == 1 ? a.nh + kvh : a.nh + a.nkv + kvh) * D + c * 8;
        for (int it = 0;; ++it) {
            const u32x4_e g0 = eng_ld16_coh(rq, (uint32_t)col * 4u), g1 = eng_ld16_coh(rq, (uint32_t)col * 4u + 16u);
            const bool ok = g0.y == qtag && g0.w == qtag && g1.y == qtag && g1.w == qtag;
            if (tid < 3 * D / 8) { u32x4_e v; v.x = g0.x; v.y = g0.z; v.z = g1.x; v.w = g1.z; *reinterpret_cast<u32x4_e*>(qkv_s + tid * 8) = v; }
            if (__syncthreads_and(ok ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((it & 63) == 63) {
                if (tid == 0) *toflag = (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS || __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
                __syncthreads();
                const int timed_out = *toflag;
                __syncthreads();
                if (timed_out) { if (tid == 0) eng_timeout(a, step + 1); break; }
            }
        }
    }
    if (probe) a.ts[3 * step + 1] = __builtin_amdgcn_s_memrealtime();
    // hipcc drained its own (granule) loads above with vmcnt(0): everything this wave requested has landed; the statements below only tie the registers
    ws_wait<0, 4>(csr);
    ws_wait<0, KU>(kraw);
    ws_wait<0, DB>(*reinterpret_cast<ws_u32x4(*)[DB]>(&vraw[0]));
    ws_wait<0, DB>(*reinterpret_cast<ws_u32x4(*)[DB]>(&vraw[DB]));
    const T* qrow = qkv_s; const T* knew = qkv_s + D; const T* vnew = qkv_s + 2 * D;

    float cv[8], sv[8];
    cv[0] = __uint_as_float(csr[0].x); cv[1] = __uint_as_float(csr[0].y); cv[2] = __uint_as_float(csr[0].z); cv[3] = __uint_as_float(csr[0].w);
    cv[4] = __uint_as_float(csr[1].x); cv[5] = __uint_as_float(csr[1].y); cv[6] = __uint_as_float(csr[1].z); cv[7] = __uint_as_float(csr[1].w);
    sv[0] = __uint_as_float(csr[2].x); sv[1] = __uint_as_float(csr[2].y); sv[2] = __uint_as_float(csr[2].z); sv[3] = __uint_as_float(csr[2].w);
    sv[4] = __uint_as_float(csr[3].x); sv[5] = __uint_as_float(csr[3].y); sv[6] = __uint_as_float(csr[3].z); sv[7] = __uint_as_float(csr[3].w);
    auto rope_reg = [&](const T* x, float (&out)[8]) {
        const bool lo = i0 < HALF;
        float av[8], bv[8];
        load8<T>(x + i0, av);
        load8<T>(x + (lo ? i0 + HALF : i0 - HALF), bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = round_to<T>(cv[e]), s_ = round_to<T>(sv[e]);
            const float rot = lo ? -bv[e] : bv[e];
            out[e] = rope_term<T>(av[e], c, rot, s_);
        }
    };
    float qv[8];
    rope_reg(qrow, qv);
    float kr[8];
    const bool new_lane = has_new && wave == 0 && kslot == 0;
    if (new_lane) rope_reg(knew, kr);

    float mx, sum;
    {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = (u * 4 + wave) * KPW + kslot;
            float kv[8]; ws_unpack8<T>(kraw[u], kv);
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kv[e], sdot);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
            if (sub == 0) sc_lds[kl] = kl < nk_cached ? sdot * scl : -INFINITY;
        }
        __syncthreads();
        if (new_lane) {
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

        float sc = tid < DF_CHUNK ? sc_lds[tid] : -INFINITY;
        mx = block_max<4>(sc, red);
        float e = tid < DF_CHUNK ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;
        sum = block_sum<4>(e, red);
        if (has_new && tid == nk - 1) { red[4] = e; e = 0.f; }
        if (tid < DF_CHUNK) sc_lds[tid] = e;
        __syncthreads();
        const float p_new = has_new ? red[4] : 0.f;

        float acc[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) acc[db] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 p0 = *reinterpret_cast<const float4*>(sc_lds + kb * 64 + s8 * 8);
            const float4 p1 = *reinterpret_cast<const float4*>(sc_lds + kb * 64 + s8 * 8 + 4);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                float vv[8]; ws_unpack8<T>(vraw[kb * DB + db], vv);
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
                if (has_new) t = fmaf(p_new, to_f32(vnew[d]), t);
                if (has_new) mg_o[d] = t;
                else __hip_atomic_store(wsg + d, ((u64)tag << 32) | __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!has_new) {
        if (tid == 0) {
            __hip_atomic_store(wsg + D, ((u64)tag << 32) | __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(wsg + D + 1, ((u64)tag << 32) | __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (probe) a.ts[3 * step + 2] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();                                                   // LDS of this item is free for the workgroup's next piece of work
        return;
    }

    // ---- the head's last chunk merges: poll the other chunks' granules until every tag is this step's -----------------------------------------------------
    __syncthreads();
    const u64* wsh = reinterpret_cast<const u64*>(a.aws) + (size_t)head * a.n_split * WSG;
    const int n_other = a.n_split - 1;
    constexpr int NG = 256 / D;
    constexpr int SPG = DF_MAX_SPLIT / NG;
    const int g = tid / D, d = tid % D;
    float ov[SPG];
    float st_m = -INFINITY, st_l = 0.f;
    {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int* toflag = reinterpret_cast<int*>(red + 6);
        for (int it = 0;; ++it) {
            bool ok = true;
            if (tid < n_other) {
                const u64 gm = __hip_atomic_load(wsh + tid * WSG + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 gl = __hip_atomic_load(wsh + tid * WSG + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(gm >> 32) == tag && (unsigned)(gl >> 32) == tag;
                st_m = __uint_as_float((unsigned)gm); st_l = __uint_as_float((unsigned)gl);
            }
#pragma unroll
            for (int i = 0; i < SPG; ++i) {
                const int s2 = g + i * NG;
                ov[i] = 0.f;
                if (s2 < n_other) {
                    const u64 gv = __hip_atomic_load(wsh + s2 * WSG + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(gv >> 32) == tag;
                    ov[i] = __uint_as_float((unsigned)gv);
                } else if (s2 == n_other) ov[i] = mg_o[d];
            }
            if (__syncthreads_and(ok ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((it & 63) == 63) {
                if (tid == 0) *toflag = (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS || __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
                __syncthreads();
                const int timed_out = *toflag;
                __syncthreads();
                if (timed_out) { if (tid == 0) eng_timeout(a, step + 1); break; }
            }
        }
    }
    if (tid < n_other) { mg_m[tid] = st_m; mg_w[tid] = st_l; }
    if (tid == n_other) { mg_m[tid] = mx; mg_w[tid] = sum; }
    __syncthreads();
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
    __syncthreads();
    mg_o[tid] = o;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int i = 1; i < NG; ++i) o += mg_o[i * D + d];
        const float r = round_to<T>(l > 0.f ? o / l : 0.f);
        const float hi = __shfl_down(r, 1, 64);                            // d and d + 1 sit in neighbouring lanes of one wave (D is a multiple of 64)
        if ((d & 1) == 0) eng_put2<T>(reinterpret_cast<u64*>(sp.out_gran), head * D + d, r, hi, tag);
    }
    if (probe) a.ts[3 * step + 2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
}

}  // namespace

template <typename T, int D>
__global__ __launch_bounds__(256) void decode_engine_kernel(EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // an earlier launch timed out: do nothing
    for (int step = 0; step < a.n_steps; ++step) {
        const EngStep sp = a.steps[step];
        if (sp.kind == 2) {
            const int items = a.nh * a.n_split;
            for (int item = blockIdx.x; item < items; item += gridDim.x) eng_attn<T, D>(a, sp, item, step, smem);
            continue;
        }
        if ((int)blockIdx.x * 4 >= sp.n_part) continue;                    // none of this workgroup's waves takes part in the step
        if (sp.kind == 1) eng_linear<T, 4, 4, true>(a, sp, step, smem);
        else if (sp.R == 2) eng_linear<T, 2, 8, false>(a, sp, step, smem);
        else eng_linear<T, 4, 4, false>(a, sp, step, smem);
    }
}

size_t decode_engine_smem(const EngArgs& a, int D, int es) {
    const size_t lin = (size_t)a.xs_bytes + 8 * 4 + 16;
    const size_t att = (size_t)(DF_CHUNK + 8 + 2 * DF_MAX_SPLIT + 256) * 4 + (size_t)3 * D * es + 16;
    return lin > att ? lin : att;
}

template <typename T, int D>
static int engine_occupancy_t(const EngArgs& a) {
    auto kern = decode_engine_kernel<T, D>;
    const size_t smem = decode_engine_smem(a, D, sizeof(T));
    LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int occ = 0;
    LMX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, smem));
    return occ;
}

int decode_engine_occupancy(int dtype, int D, const EngArgs& a) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_engine: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_engine: head_dim must be 64 or 128");
    if (dtype == kBF16) return D == 128 ? engine_occupancy_t<bf16_t, 128>(a) : engine_occupancy_t<bf16_t, 64>(a);
    return D == 128 ? engine_occupancy_t<f16_t, 128>(a) : engine_occupancy_t<f16_t, 64>(a);
}

void launch_decode_engine(int dtype, int D, const EngArgs& a, int grid, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_engine: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_engine: head_dim must be 64 or 128");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK > a.pos && a.n_split * DF_CHUNK <= a.s_max,
                "decode_engine: n_split must be the number of live 128-key chunks");
    LMX_REQUIRE(grid >= 1 && a.tag0 != 0, "decode_engine: bad grid / tag");
    const size_t smem = decode_engine_smem(a, D, 2);
#define LE(TT, DD) LMX_LAUNCH((decode_engine_kernel<TT, DD>), dim3((unsigned)grid), dim3(256), smem, st, a)
    if (dtype == kBF16) { if (D == 128) LE(bf16_t, 128); else LE(bf16_t, 64); }
    else { if (D == 128) LE(f16_t, 128); else LE(f16_t, 64); }
#undef LE
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
