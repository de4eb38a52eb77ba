// Dataflow decode step: ONE launch per generated token for a single sequence (tensor-parallel world 1), NO grid barriers.
//
// What it replaces: the 161 launches of Model::decode_step_launch — per layer {qkv GEMV (+RMSNorm), fused RoPE + KV append + split attention,
// o_proj GEMV (+residual), gate|up GEMV (+RMSNorm, SiLU*mul), down GEMV (+residual)} and the lm_head GEMV (+final norm) — the decoder half of
// LlamaModel.forward for one new token (HF5:models/llama/modeling_llama.py:367-418 via llava_llama.py:88-99).  The pick kernel (argmax | draw, state
// advance, next embedding row) stays a second launch.
//
// Why: a batch-1 decode step is a chain of HBM-bound weight streams separated by all-to-all hand-overs of one activation row.  As separate launches every
// link costs the drain of one kernel, the boundary and the ramp of the next (~3-6 us of idle HBM per link, 5 links per layer), and the attention launch in
// the middle is a pure latency chain that leaves the memory system idle.  The persistent kernel with grid barriers (round 2's decode_persist.hip, removed) paid MORE per
// link than a kernel boundary.  Here the grid is NOT persistent and there is no barrier:
//
//   * The grid is the concatenation of every step's workgroups in dependency order (block id -> (layer, step, item)).  The hardware dispatches
//     workgroups in block order, so at any time the chip holds the workgroups of the running step plus as many workgroups of the FOLLOWING steps as
//     fit.  Those start by issuing their first rounds of weight loads (and, for attention, every K / V^T load of their key chunk) — none of which
//     depends on the activation row — and only then wait for the previous step's completion counter.  The tail of step k, the hand-over and the head of
//     step k+1 therefore overlap with HBM traffic of steps k+1 / k+2 instead of idling the memory system.
//   * Hand-over (placement independent, guide §6 G16 / MI355X_MICROARCH "Workgroup dispatch ... visibility"): producers store the activation row
//     write-through (sc1), every wave drains vmcnt, barrier, ONE relaxed agent-scope fetch_add on the step's counter per workgroup; consumers poll that
//     counter from one lane (relaxed agent-scope loads + s_sleep) and read the row with sc1 loads.  Weights and the KV cache of earlier tokens were
//     written by earlier launches and use plain / non-temporal loads.
//   * Forward progress: a workgroup only ever waits for workgroups with LOWER block ids, which were dispatched before it (in order, per XCD), so the
//     lowest unfinished workgroup is always resident and runnable; several sequences' grids may share the chip.  Every wait is bounded (~50 ms): a
//     timeout raises a status word the host checks, and the workgroup carries on (wrong data, no hang).
//   * Counters are per sequence, double-buffered by launch parity; item 0 of every step zeroes the other parity's counter for the next launch.
//
// Arithmetic is IDENTICAL to the separate kernels (same per-lane accumulation order and rounding points as gemv_kernel / decode_fused_kernel), so
// ids and logits are bit-identical to the separate launches (tests/test_decode_flow_gpu.py).
#include "attention_decode.h"
#include "common.h"
#include "kernels.h"
#include "wstream.h"

namespace lmx {

namespace {

typedef uint32_t u32x4_w __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void unpack8(const u32x4_w v, float (&f)[8]) {
    f[0] = unpack_lo<T>(v.x); f[1] = unpack_hi<T>(v.x); f[2] = unpack_lo<T>(v.y); f[3] = unpack_hi<T>(v.y);
    f[4] = unpack_lo<T>(v.z); f[5] = unpack_hi<T>(v.z); f[6] = unpack_lo<T>(v.w); f[7] = unpack_hi<T>(v.w);
}

// raw (stride 0) buffer view of an activation row: 16-byte loads / stores with the sc1 bit (agent scope: bypass the non-coherent levels / write through)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t flow_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4_w ld16_coh(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, /*sc1*/ 16);
}
template <typename T> __device__ __forceinline__ float ld_coh(const T* p) {
    const unsigned short u = __hip_atomic_load(reinterpret_cast<const unsigned short*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    T t; *reinterpret_cast<unsigned short*>(&t) = u;
    return to_f32(t);
}

constexpr uint64_t FLOW_TIMEOUT_TICKS = 5000000ull;        // s_memrealtime runs at 100 MHz: 50 ms

// Completion counters are SHARDED: FLOW_NSUB words per (parity, step), each on its own 128-byte line; workgroup `item` of a step arrives on shard
// item % FLOW_NSUB, so a shard of an n-workgroup step is complete at (n - shard + NSUB - 1) / NSUB arrivals.  One counter per step measured ~50 ns per
// arrival (every arrival queues behind the polls of ~1000 waiting workgroups on the same line): 8.4 ms/token against 3.1 ms for the separate launches.
__device__ __forceinline__ unsigned* flow_counter(const FlowArgs& a, int par, int step, int shard) {
    return a.done + (((size_t)par * a.n_steps + step) * FLOW_NSUB + shard) * FLOW_SUB_STRIDE;
}
__device__ __forceinline__ int flow_quota(int n, int shard) { return (n - shard + FLOW_NSUB - 1) / FLOW_NSUB; }

// ---- wait for the previous step's `need` workgroups (bounded) -----------------------------------------------------------------------------------------
// Lane 0 polls this workgroup's own shard (item % NSUB: the pollers spread over the lines), then the first NSUB lanes check every shard at once.
// debug timeline: earliest of a step's first 8 workgroups / latest of every 8th workgroup to reach a point (slot: see FlowArgs::ts)
__device__ __forceinline__ void flow_stamp_min(const FlowArgs& a, int slot, int item) {
    if (a.ts && a.done && threadIdx.x == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (item < 8) __hip_atomic_fetch_min(a.ts + slot, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((item & 7) == 0) __hip_atomic_fetch_max(a.ts + slot + 2 * a.n_steps, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void flow_wait(const FlowArgs& a, int step, int need, int item) {
    if (need <= 0) return;                                          // uniform: the first step of the launch reads what earlier launches wrote
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        bool give_up = false;
        if (lane == 0) {
            const int my = item % FLOW_NSUB;
            const unsigned* p = flow_counter(a, a.par, step - 1, my);
            const int q = flow_quota(need, my);
            int it = 0;
            while ((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < q) {
                __builtin_amdgcn_s_sleep(4);
                if ((++it & 63) == 0) {
                    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { give_up = true; break; }   // another workgroup gave up
                    if (__builtin_amdgcn_s_memrealtime() - t0 > FLOW_TIMEOUT_TICKS) {
                        __hip_atomic_store(a.abort_word, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(a.status, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        give_up = true; break;
                    }
                }
            }
        }
        give_up = __builtin_amdgcn_readfirstlane((int)give_up) != 0;
        if (!give_up) {
            const int sh = lane < FLOW_NSUB ? lane : 0;
            const unsigned* p = flow_counter(a, a.par, step - 1, sh);
            const int q = flow_quota(need, sh);
            int it = 0;
            while (true) {
                const bool ok = (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= q;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(4);
                if ((++it & 63) == 0) {
                    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > FLOW_TIMEOUT_TICKS) {
                        if (lane == 0) {
                            __hip_atomic_store(a.abort_word, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(a.status, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// ---- this workgroup's write-through stores are acknowledged -> one arrival on its shard of this step's counter ------------------------------------------
__device__ __forceinline__ void flow_signal(const FlowArgs& a, int step, int mine, int item) {
    if (!a.done) return;                                            // stand-alone attention launch: the kernel boundary is the hand-over
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int sh = item % FLOW_NSUB;
        unsigned* c = flow_counter(a, a.par, step, sh);
        if (a.ts) {                                                 // debug timeline: the last arrival of a shard stamps the clock (max over the shards)
            const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)old + 1 == flow_quota(mine, sh)) __hip_atomic_fetch_max(a.ts + 1 + step, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- one linear step: C = act(norm(x) W^T) (+ residual) -------------------------------------------------------------------------------------------------
// The step has `mine` workgroups (the host's choice, about one or two per CU: every workgroup of a step should be resident before the step is released);
// wave w of workgroup `item` owns the slots item * 4 + w, + 4 * mine, + 8 * mine, ...  A slot is R weight rows, mapped and summed exactly as by
// gemv_kernel<T, 1, R> (gemm.hip): slot s = rows [s R, (s + 1) R) (SiLU*mul: R / 2 gate/up pairs of the fused [32 gate | 32 up] layout), lane l
// accumulates the 8-element chunks l, l + 64, l + 128, ... in that order, wave_sum, lane 0 writes.  A wave's slots form ONE stream of load rounds
// (round = the R rows' 16 bytes at one chunk): P rounds are always in flight, across slot boundaries, and the first P are issued before the wait.
// Addressing: one raw buffer view of W per step; the lane's chunk offset is the only address VGPR (shared by the R rows), each row's byte offset is
// wave-uniform and rides in the instruction's scalar offset, so R x P loads in flight cost R x P x 4 data registers and nothing else.
template <typename T, int R, int P, bool SILU>
__device__ __forceinline__ void flow_linear(const FlowArgs& a, const FlowStep& sp, int item, int step, int need, int mine, char* smem) {
    T* xs = reinterpret_cast<T*>(smem);
    float* red = reinterpret_cast<float*>(smem + a.xs_bytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = sp.N, K = sp.K, KC = K >> 3;
    const int NR = (KC + 63) >> 6;                                         // load rounds per slot
    const int nslots = (N + R - 1) / R;
    const int stride = mine * 4;                                           // waves of this step
    const int slot0 = item * 4 + wave;
    const int nloc = slot0 < nslots ? (nslots - slot0 + stride - 1) / stride : 0;        // slots of this wave (wave-uniform)
    const ws_v4i rw = ws_make_rsrc(sp.W, 0x7fffffffu);
    auto row_off = [&](int sl, int r) -> uint32_t {                        // byte offset of row r of slot sl (wave-uniform -> SGPRs)
        int f;
        if (SILU) { const int j = (sl * R + r) >> 1; f = 64 * (j >> 5) + (j & 31) + 32 * ((sl * R + r) & 1); }
        else f = sl * R + r;
        f = f < N ? f : N - 1;
        return (uint32_t)__builtin_amdgcn_readfirstlane(f) * (uint32_t)K * (uint32_t)sizeof(T);
    };
    // The wave's slots form ONE hand-counted stream (wstream.h): R x P loads always on the wire, across slot boundaries; past the last round dummy loads of
    // one hot line keep the count exact.
    ws_u32x4 buf[P][R];
    int i_sl = slot0, i_j = 0, i_left = nloc;                              // issue cursor: slot, round within the slot, slots left (all wave-uniform)
    auto issue = [&](int p) {
        const bool live = i_left > 0;
        const int c = lane + 64 * i_j;
        const uint32_t vo = live ? (uint32_t)(c < KC ? c : KC - 1) * 16u : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) ws_load(buf[p][r], vo, rw, live ? row_off(i_sl, r) : 0u);
        if (live && ++i_j == NR) { i_j = 0; i_sl += stride; --i_left; }
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue(p);                                  // on the wire while this workgroup waits for its input row
    flow_wait(a, step, need, item);
    flow_stamp_min(a, 1 + a.n_steps + step, item);

    // ---- stage x in LDS (sc1 loads; RMSNorm with HF's rounding points, statistics in fp32, same order as gemv_kernel) ------------------------------
    {
        const __amdgpu_buffer_rsrc_t rx = flow_rsrc(sp.x);
        const T* g = reinterpret_cast<const T*>(sp.norm_w);
        float ss = 0.f;
        for (int c = tid; c < KC; c += 256) {
            const u32x4_w raw = ld16_coh(rx, (uint32_t)c * 16u);
            *reinterpret_cast<u32x4_w*>(xs + c * 8) = raw;
            if (g) {
                float v[8]; unpack8<T>(raw, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
            }
        }
        if (g) {
            ss = block_sum<4>(ss, red);
            const float inv = rsqrtf(ss / (float)K + a.eps);
            for (int c = tid; c < KC; c += 256) {                          // each thread re-reads exactly the chunks it wrote: no barrier needed in between
                float v[8], gv[8];
                load8<T>(xs + c * 8, v);
                load8<T>(g + c * 8, gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * gv[e];
                store8<T>(xs + c * 8, v);
            }
        }
    }
    __syncthreads();
    flow_stamp_min(a, 1 + 2 * a.n_steps + step, item);

    // Results stay in registers until the stream has ended (lane k keeps output k of this wave): stores share the vmcnt queue with the weight loads and
    // only loads retire in order, so a store inside the stream would break the count.
    constexpr int OPS = SILU ? R / 2 : R;                                  // outputs per slot
    float keep = 0.f;
    {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        int c_j = 0, c_n = 0;                                              // consume cursor: round within the slot, slots done
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
                    if (++c_j == NR) {                                     // the slot's last round: reduce, park the outputs, next slot
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
    if (lane < nloc * OPS) {                                               // lane k: output k % OPS of this wave's slot k / OPS
        const int sl = slot0 + (lane / OPS) * stride;
        const int n = sl * OPS + lane % OPS;                               // SiLU*mul: index into the [N / 2] output
        T* C = reinterpret_cast<T*>(sp.C);
        if (SILU) {
            if (n < N / 2) store_coherent<T>(C + n, from_f32<T>(keep));
        } else if (n < N) {
            const T* res = reinterpret_cast<const T*>(sp.res);
            float v = keep;
            if (res) v += ld_coh<T>(res + n);
            store_coherent<T>(C + n, from_f32<T>(v));
        }
    }
    flow_signal(a, step, mine, item);
}

// ---- attention step: RoPE(q, k_new) + KV append + one 128-key chunk of one head + in-launch merge by the head's last workgroup -----------------------
// The arithmetic is decode_fused_body's (attention_decode.h), statement for statement; what differs is the order of the memory operations: every K / V^T
// load of the chunk is issued BEFORE the wait for the qkv row (they only depend on the position, which the host passes by value), the q / k / v slices
// of the row arrive through LDS (sc1 loads), and the grid holds only the chunks that exist (n_split = live chunks).
template <typename T, int D>
__device__ __forceinline__ void flow_attn(const FlowArgs& a, const FlowStep& sp, int item, int step, int need, int mine, char* smem) {
    float* sc_lds = reinterpret_cast<float*>(smem);                       // [DF_CHUNK] scores -> probabilities of this chunk
    float* red = sc_lds + DF_CHUNK;                                        // [8]
    float* mg_m = red + 8; float* mg_w = mg_m + DF_MAX_SPLIT;              // merge: split maxima / weights
    float* mg_o = mg_w + DF_MAX_SPLIT;                                     // [256] merge: cross-group partial sums
    T* qkv_s = reinterpret_cast<T*>(mg_o + 256);                           // [3 D] q | k_new | v_new of this head

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = item % a.nh, split = item / a.nh;
    const int group = a.nh / a.nkv;
    const int kvh = head / group;
    const int pos = a.pos;
    const int kv_len = pos + 1;
    const int k_begin = split * DF_CHUNK;
    int k_end = k_begin + DF_CHUNK; k_end = k_end < kv_len ? k_end : kv_len;
    const int nk = k_end - k_begin;                                        // >= 1: only live chunks are launched
    const bool has_new = pos >= k_begin && pos < k_end;
    const int nk_cached = has_new ? nk - 1 : nk;

    T* Kc = reinterpret_cast<T*>(sp.kc) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(sp.vt) + (size_t)kvh * D * a.s_max;
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

    // ---- every global load of this chunk: K rows, then V^T lines (independent of the qkv row) ---------------------------------------------------------
    u32x4_w kraw[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int kl = (u * 4 + wave) * KPW + kslot;
        const int key = k_begin + (kl < nk_cached ? kl : (nk_cached > 0 ? nk_cached - 1 : 0));
        kraw[u] = *reinterpret_cast<const u32x4_w*>(Kr + (size_t)key * D + sub * 8);
    }
    u32x4_w vraw[2][DB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int db = 0; db < DB; ++db) vraw[kb][db] = *reinterpret_cast<const u32x4_w*>(Vr + (size_t)(db * 32 + drow) * a.s_max + k_begin + kb * 64 + s8 * 8);

    flow_wait(a, step, need, item);
    flow_stamp_min(a, 1 + a.n_steps + step, item);
    // debug probe (stand-alone launch with a.ts set): block (head 0, split 0) stamps slots 0 (start) and 4 (partial stored)
    const bool probe_on = !a.done && a.ts && item == 0 && tid == 0;
    if (probe_on) a.ts[0] = __builtin_amdgcn_s_memrealtime();

    // ---- q / k_new / v_new of this head: the qkv step's row, fetched coherently into LDS ---------------------------------------------------------------
    if (tid < 3 * D / 8) {
        const int part = tid / (D / 8), c = tid % (D / 8);
        const int col = (part == 0 ? head : part == 1 ? a.nh + kvh : a.nh + a.nkv + kvh) * D + c * 8;
        const __amdgpu_buffer_rsrc_t rq = flow_rsrc(a.qkv);
        *reinterpret_cast<u32x4_w*>(qkv_s + part * D + c * 8) = ld16_coh(rq, (uint32_t)col * (uint32_t)sizeof(T));
    }
    __syncthreads();
    const T* qrow = qkv_s; const T* knew = qkv_s + D; const T* vnew = qkv_s + 2 * D;

    float qv[8];
    rope8<T, D>(qrow, cs, sub * 8, qv);
    float mx, sum;
    {
        // ---- scores ----------------------------------------------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = (u * 4 + wave) * KPW + kslot;
            float kv[8]; unpack8<T>(kraw[u], kv);
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kv[e], sdot);
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
                float vv[8]; unpack8<T>(vraw[kb][db], vv);
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
                if (a.attn_form >= 3) ws[d] = t;                                                    // merged by the next LAUNCH: plain store
                else __hip_atomic_store(ws + d, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through: the merger may sit on any XCD
            }
        }
    }
    if (tid == 0) {
        if (a.attn_form >= 3) { ws[D] = mx; ws[D + 1] = sum; }
        else {
            __hip_atomic_store(ws + D, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws + D + 1, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    flow_stamp_min(a, 1 + 2 * a.n_steps + step, item);                           // attention: first workgroup with its partial stored
    // form 3 (stand-alone launch only): the chunks' partials ARE the output; o_proj's staging merges them across the kernel boundary (gemm.hip:
    // gemv2m_kernel) — no ticket, no second pass, no hand-over inside this launch
    if (a.attn_form >= 3) {
        if (probe_on) a.ts[4] = __builtin_amdgcn_s_memrealtime();
        if (!a.done && a.ts && tid == 0) __hip_atomic_fetch_max(a.ts + 5, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // latest workgroup
        return;
    }

    // ---- split merge by the last workgroup to arrive for this head (decode_fused_body's protocol) ------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(a.cnt + head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[5] = (ticket == a.n_split - 1) ? 1.f : 0.f;
    }
    __syncthreads();
    if (red[5] != 0.f) {
        const float* wsh = a.aws + (size_t)head * a.n_split * WS;
        if (tid < a.n_split) {
            mg_m[tid] = __hip_atomic_load(wsh + tid * WS + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mg_w[tid] = __hip_atomic_load(wsh + tid * WS + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        constexpr int NG = 256 / D;
        constexpr int SPG = DF_MAX_SPLIT / NG;
        const int g = tid / D, d = tid % D;
        float ov[SPG];
#pragma unroll
        for (int i = 0; i < SPG; ++i) {
            const int s2 = g + i * NG;
            ov[i] = s2 < a.n_split ? __hip_atomic_load(wsh + s2 * WS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
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
        mg_o[tid] = o;
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int i = 1; i < NG; ++i) o += mg_o[i * D + d];
            store_coherent<T>(reinterpret_cast<T*>(a.attn) + head * D + d, from_f32<T>(l > 0.f ? o / l : 0.f));
        }
        if (tid == 0) __hip_atomic_store(a.cnt + head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-arm for the next launch
    }
    flow_signal(a, step, mine, item);
}

// ---- attention step, second form (default; LMX_ATTN_FORM=1 keeps flow_attn) ----------------------------------------------------------------------------
// Same arithmetic as flow_attn / decode_fused_body, statement for statement.  What changes is the LATENCY CHAIN of the step, which is what it costs (its
// 19 MB of KV at context ~1150 are 3 us of HBM time; decode_fused_kernel took 15.6 us):
//   * loads are inline asm with hand-counted waits (wstream.h).  Stand-alone launch: the few loads the step's critical path starts with (this head's
//     q | k | v slices, this lane's cos / sin) go out FIRST, the 16 K / V^T loads behind them, and the RoPE / staging work runs after `vmcnt(16)` while
//     the KV chunk is still landing; K is waited for with vmcnt(8), V^T with vmcnt(0).  hipcc's own schedule drains everything before the first use.
//     Inside the flow launch the K / V^T loads go out before the wait for the qkv step, the small loads after it.
//   * the split merge has no ticket: a partial is stored as 8-byte {value, tag} granules (one sc1 store each, tag = a number unique to this launch and
//     layer), and the workgroup of the head's LAST chunk (highest block id of the head: it only ever waits for lower ids) polls the other chunks'
//     granules until every tag matches, then merges in split order as before.  The producer side is fire-and-forget; the chain store -> vmcnt(0) ->
//     barrier -> ticket atomic -> merger's loads becomes store -> merger's poll.
template <typename T, int D>
__device__ __forceinline__ void flow_attn2(const FlowArgs& a, const FlowStep& sp, int item, int step, int need, int mine, char* smem) {
    float* sc_lds = reinterpret_cast<float*>(smem);                       // [DF_CHUNK] scores -> probabilities of this chunk
    float* red = sc_lds + DF_CHUNK;                                        // [8]
    float* mg_m = red + 8; float* mg_w = mg_m + DF_MAX_SPLIT;              // merge: split maxima / weights
    float* mg_o = mg_w + DF_MAX_SPLIT;                                     // [256] merge: cross-group partial sums
    T* qkv_s = reinterpret_cast<T*>(mg_o + 256);                           // [3 D] q | k_new | v_new of this head

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = item % a.nh, split = item / a.nh;
    const int group = a.nh / a.nkv;
    const int kvh = head / group;
    const int pos = a.pos;
    const int kv_len = pos + 1;
    const int k_begin = split * DF_CHUNK;
    int k_end = k_begin + DF_CHUNK; k_end = k_end < kv_len ? k_end : kv_len;
    const int nk = k_end - k_begin;                                        // >= 1: only live chunks are launched
    const bool has_new = pos >= k_begin && pos < k_end;                    // == (split == n_split - 1): this workgroup also merges the head
    const int nk_cached = has_new ? nk - 1 : nk;
    const unsigned tag = a.tag + (unsigned)(step / 5);                     // unique per (launch, layer)
    // debug probe (stand-alone launch with a.ts set): block (head 0, split 0) stamps slots 0.., the merging block of head 0 slots 8..
    const int probe_base = (!a.done && a.ts && head == 0) ? (split == 0 ? 0 : (has_new ? 8 : -1)) : -1;
    auto probe = [&](int k) { if (probe_base >= 0 && tid == 0) a.ts[probe_base + k] = __builtin_amdgcn_s_memrealtime(); };
    probe(0);

    T* Kc = reinterpret_cast<T*>(sp.kc) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(sp.vt) + (size_t)kvh * D * a.s_max;
    constexpr int WSG = D + 4;                                             // granules per partial: o[D], m, l, pad, pad
    unsigned long long* wsg = reinterpret_cast<unsigned long long*>(a.aws) + ((size_t)head * a.n_split + split) * WSG;

    constexpr int LPK = D / 8, KPW = 64 / LPK;
    constexpr int KU = DF_CHUNK / (4 * KPW);
    constexpr int DB = D / 32;
    static_assert(KU == 8 || KU == 4, "K loads per lane");
    const float scl = a.scale * 1.4426950408889634f;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int s8 = tid & 7, drow = tid >> 3;

    const ws_v4i rsK = ws_make_rsrc(Kc, (uint32_t)((size_t)a.s_max * D * sizeof(T)));
    const ws_v4i rsV = ws_make_rsrc(Vt, (uint32_t)((size_t)a.s_max * D * sizeof(T)));
    const ws_v4i rsQ = ws_make_rsrc(a.qkv, 0x7fffffffu);
    const ws_v4i rsC = ws_make_rsrc(a.rope + (size_t)pos * D, (uint32_t)(D * 4));
    ws_u32x4 kraw[KU], vraw[2 * DB], qp[1], csr[4];
    auto issue_kv = [&]() {
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
    };
    // q | k_new | v_new slices of this head (threads 0 .. 3 D / 8 - 1 fetch 16 bytes each; the others re-read thread 0's) and this lane's cos / sin runs
    constexpr int HALF = D / 2;
    const int i0 = sub * 8, j0 = i0 < HALF ? i0 : i0 - HALF;
    auto issue_small = [&]() {
        const int t = tid < 3 * D / 8 ? tid : 0;
        const int part = t / (D / 8), c = t % (D / 8);
        const int col = (part == 0 ? head : part == 1 ? a.nh + kvh : a.nh + a.nkv + kvh) * D + c * 8;
        ws_load_sc1(qp[0], (uint32_t)col * (uint32_t)sizeof(T), rsQ, 0u);
        ws_load_plain(csr[0], (uint32_t)j0 * 4u, rsC, 0u);
        ws_load_plain(csr[1], (uint32_t)j0 * 4u + 16u, rsC, 0u);
        ws_load_plain(csr[2], (uint32_t)(HALF + j0) * 4u, rsC, 0u);
        ws_load_plain(csr[3], (uint32_t)(HALF + j0) * 4u + 16u, rsC, 0u);
    };
    if (need > 0) {                                                        // flow launch: the KV chunk is on the wire while the qkv step finishes
        issue_kv();
        flow_wait(a, step, need, item);
        flow_stamp_min(a, 1 + a.n_steps + step, item);
        issue_small();
        ws_wait<0, 1>(qp);
    } else {                                                               // stand-alone launch: the short loads first, the KV chunk behind them
        issue_small();
        issue_kv();
        ws_wait<KU + 2 * DB, 1>(qp);
    }
    probe(1);
    ws_wait<KU + 2 * DB, 4>(csr);                                          // (already satisfied: ties the registers to the wait above)
    if (tid < 3 * D / 8) *reinterpret_cast<ws_u32x4*>(qkv_s + tid * 8) = qp[0];      // part * D + c * 8 == tid * 8
    __syncthreads();
    const T* qrow = qkv_s; const T* knew = qkv_s + D; const T* vnew = qkv_s + 2 * D;

    // RoPE with HF's rounding chain (rope8 of attention_decode.h, the cos / sin values from the registers loaded above)
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
        // ---- scores ----------------------------------------------------------------------------------------------------
        ws_wait<2 * DB, KU>(kraw);                                         // K rows landed (V^T may still be on the wire)
        probe(2);
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
        ws_wait<0, DB>(*reinterpret_cast<ws_u32x4(*)[DB]>(&vraw[0]));     // every load of this wave has landed: from here on hipcc's own bookkeeping is exact
        ws_wait<0, DB>(*reinterpret_cast<ws_u32x4(*)[DB]>(&vraw[DB]));
        probe(3);
        __syncthreads();
        // newest key: rotated straight from the qkv row; one workgroup per kv head appends it to the caches
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
                if (has_new) mg_o[d] = t;                                  // the merging workgroup keeps its own partial in LDS
                else __hip_atomic_store(wsg + d, ((unsigned long long)tag << 32) | __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    probe(4);
    if (!has_new) {
        if (tid == 0) {
            __hip_atomic_store(wsg + D, ((unsigned long long)tag << 32) | __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(wsg + D + 1, ((unsigned long long)tag << 32) | __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        flow_stamp_min(a, 1 + 2 * a.n_steps + step, item);
        flow_signal(a, step, mine, item);
        return;
    }

    // ---- the head's last chunk merges: poll the other chunks' granules until every tag is this launch's -------------------------------------------------
    __syncthreads();                                                       // mg_o (own partial) visible
    const unsigned long long* wsh = reinterpret_cast<const unsigned long long*>(a.aws) + (size_t)head * a.n_split * WSG;
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
            if (tid < n_other) {                                           // thread s: statistics of chunk s
                const unsigned long long gm = __hip_atomic_load(wsh + tid * WSG + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long gl = __hip_atomic_load(wsh + tid * WSG + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(gm >> 32) == tag && (unsigned)(gl >> 32) == tag;
                st_m = __uint_as_float((unsigned)gm); st_l = __uint_as_float((unsigned)gl);
            }
#pragma unroll
            for (int i = 0; i < SPG; ++i) {
                const int s2 = g + i * NG;
                ov[i] = 0.f;
                if (s2 < n_other) {
                    const unsigned long long gv = __hip_atomic_load(wsh + s2 * WSG + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(gv >> 32) == tag;
                    ov[i] = __uint_as_float((unsigned)gv);
                } else if (s2 == n_other) ov[i] = mg_o[d];                 // this workgroup's own partial, at its place in the split order
            }
            if (__syncthreads_and(ok ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(2);
            if ((it & 63) == 63) {                                         // bounded: one thread reads the clock, everybody leaves together
                if (tid == 0) *toflag = (__builtin_amdgcn_s_memrealtime() - t0 > FLOW_TIMEOUT_TICKS) ? 1 : 0;
                __syncthreads();
                const int timed_out = *toflag;
                __syncthreads();
                if (timed_out) {
                    if (tid == 0 && a.status) {
                        __hip_atomic_store(a.abort_word, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(a.status, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
            }
        }
    }
    probe(5);
    if (tid < n_other) { mg_m[tid] = st_m; mg_w[tid] = st_l; }
    if (tid == n_other) { mg_m[tid] = mx; mg_w[tid] = sum; }               // this chunk is split n_split - 1
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
    __syncthreads();                                                       // everyone has read mg_o[d] (own partial) before it is reused
    mg_o[tid] = o;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int i = 1; i < NG; ++i) o += mg_o[i * D + d];
        store_coherent<T>(reinterpret_cast<T*>(a.attn) + head * D + d, from_f32<T>(l > 0.f ? o / l : 0.f));
    }
    probe(6);
    flow_stamp_min(a, 1 + 2 * a.n_steps + step, item);
    flow_signal(a, step, mine, item);
}

}  // namespace

template <typename T, int D, int DEEP>
__device__ __forceinline__ void decode_flow_body(const FlowArgs& a, char* smem) {
    const int bid = blockIdx.x;
    const int lb = a.off5;                                                   // workgroups per layer
    int layer, kidx, item, need, mine;                                       // need: workgroups of the previous step, mine: of this step
    if (bid < a.L * lb) {
        layer = bid / lb;
        const int r = bid - layer * lb;
        if (r < a.off1) { kidx = 0; item = r; need = a.nb4; mine = a.off1; }
        else if (r < a.off2) { kidx = 1; item = r - a.off1; need = a.off1; mine = a.off2 - a.off1; }
        else if (r < a.off3) { kidx = 2; item = r - a.off2; need = a.off2 - a.off1; mine = a.off3 - a.off2; }
        else if (r < a.off4) { kidx = 3; item = r - a.off3; need = a.off3 - a.off2; mine = a.off4 - a.off3; }
        else { kidx = 4; item = r - a.off4; need = a.off4 - a.off3; mine = a.nb4; }
    } else { layer = a.L; kidx = 0; item = bid - a.L * lb; need = a.nb4; mine = a.nb_head; }
    const int step = layer * 5 + kidx;
    if (step == 0) need = 0;
    if (a.ts && bid == 0 && threadIdx.x == 0) a.ts[0] = __builtin_amdgcn_s_memrealtime();
    const FlowStep sp = a.steps[step];
    if (item == 0 && threadIdx.x < FLOW_NSUB) *flow_counter(a, 1 - a.par, step, threadIdx.x) = 0u;   // re-arm the other parity's shards for the next launch
    if (sp.kind == 2) { if (a.attn_form == 1) flow_attn<T, D>(a, sp, item, step, need, mine, smem); else flow_attn2<T, D>(a, sp, item, step, need, mine, smem); return; }
    // P = rounds of 16-byte loads in flight per row: R x P x 16 bytes per lane are on the wire (or landed) while the workgroup waits for its input row
    if (sp.kind == 1) {
        if (sp.R == 2) flow_linear<T, 2, 4 * DEEP, true>(a, sp, item, step, need, mine, smem);
        else flow_linear<T, 4, 2 * DEEP, true>(a, sp, item, step, need, mine, smem);
        return;
    }
    if (sp.R == 1) flow_linear<T, 1, 8 * DEEP, false>(a, sp, item, step, need, mine, smem);
    else if (sp.R == 2) flow_linear<T, 2, 4 * DEEP, false>(a, sp, item, step, need, mine, smem);
    else flow_linear<T, 4, 2 * DEEP, false>(a, sp, item, step, need, mine, smem);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void decode_flow_kernel(FlowArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    decode_flow_body<T, D, 1>(a, smem);
}
// default arm: 16 loads (64 data registers) in flight per wave; LMX_FLOW_DEEP=1 selects the 8-load kernel above
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_flow_kernel_deep2(FlowArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    decode_flow_body<T, D, 2>(a, smem);
}

// The attention step as its own launch (the separate-launch decode path of 16-bit models): flow_attn without wait / signal.  Against decode_fused_kernel
// (attention.hip) the position arrives by value (no dependent scalar load ahead of the K / V^T loads), only the live chunks are launched, K / V^T stay
// packed in registers until used; the arithmetic is the same statement for statement (bit-identical output).
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_attn_flow_kernel(FlowArgs a, FlowStep sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.attn_form != 2) flow_attn<T, D>(a, sp, blockIdx.x, 0, 0, (int)gridDim.x, smem);
    else flow_attn2<T, D>(a, sp, blockIdx.x, 0, 0, (int)gridDim.x, smem);
}

void launch_decode_attn_flow(int dtype, int D, const FlowArgs& a, const FlowStep& sp, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_attn_flow: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_attn_flow: head_dim must be 64 or 128");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK > a.pos && a.n_split * DF_CHUNK <= a.s_max && !a.done,
                "decode_attn_flow: n_split must be the number of live 128-key chunks; no completion counters");
    const size_t smem = (size_t)(DF_CHUNK + 8 + 2 * DF_MAX_SPLIT + 256) * 4 + (size_t)3 * D * 2 + 16;
#define LA(TT, DD) LMX_LAUNCH((decode_attn_flow_kernel<TT, DD>), dim3((unsigned)(a.nh * a.n_split)), dim3(256), smem, st, a, sp)
    if (dtype == kBF16) { if (D == 128) LA(bf16_t, 128); else LA(bf16_t, 64); }
    else { if (D == 128) LA(f16_t, 128); else LA(f16_t, 64); }
#undef LA
    LMX_CHECK_HIP(hipGetLastError());
}


// ---- decode attention, one workgroup per HEAD (round 4; opt-in LMX_ATTN_HEAD=1 until measured) ---------------------------------------------------------------
// decode_attn_flow_kernel spreads a head's 128-key chunks over workgroups and merges them through memory: 14.5 us per layer for 19 MB of KV, of which the
// chunk's own work ends at ~7 us (EXPERIMENTS.md r3-D) — the rest is the hand-over store -> ticket -> load across workgroups.  Here ONE 512-thread workgroup owns
// a head: its two 256-thread groups walk the head's live chunks alternately (chunk 2 i + g in iteration i), every chunk with flow_attn's arithmetic statement
// for statement, the next iteration's K / V^T loads issued before the current chunk's arithmetic; the per-chunk partials {o[D], max, sum} stay in LDS and are
// merged in chunk order by flow_attn's merge — no counter, no memory round trip, bit-identical output.  The price: a head's ~0.6 MB of KV (context ~1150)
// arrive through ONE CU's L1 path, and only n_heads of the 256 CUs work.
template <typename T, int D>
__global__ __launch_bounds__(512) void decode_attn_head_kernel(FlowArgs a, FlowStep sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WS = D + 4;
    float* sc_all = reinterpret_cast<float*>(smem);                        // [2][DF_CHUNK]
    float* red_all = sc_all + 2 * DF_CHUNK;                                // [2][8]
    float* mg_o = red_all + 16;                                            // [256]
    float* part = mg_o + 256;                                              // [DF_MAX_SPLIT][WS]: o[D], max, sum per chunk
    T* qkv_s = reinterpret_cast<T*>(part + DF_MAX_SPLIT * WS);             // [3 D]

    const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = t & 63, wave = t >> 6;
    float* sc_lds = sc_all + grp * DF_CHUNK;
    float* red = red_all + grp * 8;
    const int head = blockIdx.x;
    const int group = a.nh / a.nkv;
    const int kvh = head / group;
    const int pos = a.pos;
    const int kv_len = pos + 1;
    const int n_split = a.n_split;

    T* Kc = reinterpret_cast<T*>(sp.kc) + (size_t)kvh * a.s_max * D;
    T* Vt = reinterpret_cast<T*>(sp.vt) + (size_t)kvh * D * a.s_max;
    const T* __restrict__ Kr = Kc;
    const T* __restrict__ Vr = Vt;
    const float* cs = a.rope + (size_t)pos * D;

    constexpr int LPK = D / 8, KPW = 64 / LPK;
    constexpr int KU = DF_CHUNK / (4 * KPW);
    constexpr int DB = D / 32;
    const float scl = a.scale * 1.4426950408889634f;
    const int sub = lane % LPK, kslot = lane / LPK;
    const int s8 = t & 7, drow = t >> 3;

    // group-local block reductions with block_max<4> / block_sum<4>'s order (the barriers are workgroup-wide: both groups always reduce together)
    auto grp_max = [&](float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float r = red[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) r = fmaxf(r, red[i]);
        return r;
    };
    auto grp_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) r += red[i];
        return r;
    };
    // every load of a chunk: K rows, then V^T lines; a chunk past the last live one re-reads the last one (never used)
    auto issue = [&](int split, u32x4_w (&kraw)[KU], u32x4_w (&vraw)[2][DB]) {
        const int sp2 = split < n_split ? split : n_split - 1;
        const int k_begin = sp2 * DF_CHUNK;
        int k_end = k_begin + DF_CHUNK; k_end = k_end < kv_len ? k_end : kv_len;
        const int nk = k_end - k_begin;
        const bool has_new = pos >= k_begin && pos < k_end;
        const int nk_cached = has_new ? nk - 1 : nk;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = (u * 4 + wave) * KPW + kslot;
            const int key = k_begin + (kl < nk_cached ? kl : (nk_cached > 0 ? nk_cached - 1 : 0));
            kraw[u] = *reinterpret_cast<const u32x4_w*>(Kr + (size_t)key * D + sub * 8);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int db = 0; db < DB; ++db) vraw[kb][db] = *reinterpret_cast<const u32x4_w*>(Vr + (size_t)(db * 32 + drow) * a.s_max + k_begin + kb * 64 + s8 * 8);
    };

    u32x4_w kA[KU], vA[2][DB], kB[KU], vB[2][DB];
    issue(grp, kA, vA);
    // q / k_new / v_new of this head into LDS
    if (tid < 3 * D / 8) {
        const int prt = tid / (D / 8), c = tid % (D / 8);
        const int col = (prt == 0 ? head : prt == 1 ? a.nh + kvh : a.nh + a.nkv + kvh) * D + c * 8;
        const __amdgpu_buffer_rsrc_t rq = flow_rsrc(a.qkv);
        *reinterpret_cast<u32x4_w*>(qkv_s + prt * D + c * 8) = ld16_coh(rq, (uint32_t)col * (uint32_t)sizeof(T));
    }
    __syncthreads();
    const T* qrow = qkv_s; const T* knew = qkv_s + D; const T* vnew = qkv_s + 2 * D;
    float qv[8];
    rope8<T, D>(qrow, cs, sub * 8, qv);

    // one chunk with flow_attn's arithmetic; `valid` = this group has a live chunk in this iteration (the barriers are taken either way)
    auto chunk = [&](int split, const u32x4_w (&kraw)[KU], const u32x4_w (&vraw)[2][DB]) {
        const bool valid = split < n_split;
        const int sp2 = valid ? split : n_split - 1;
        const int k_begin = sp2 * DF_CHUNK;
        int k_end = k_begin + DF_CHUNK; k_end = k_end < kv_len ? k_end : kv_len;
        const int nk = k_end - k_begin;
        const bool has_new = valid && pos >= k_begin && pos < k_end;
        const int nk_cached = has_new ? nk - 1 : nk;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = (u * 4 + wave) * KPW + kslot;
            float kv[8]; unpack8<T>(kraw[u], kv);
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kv[e], sdot);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
            if (sub == 0) sc_lds[kl] = kl < nk_cached ? sdot * scl : -INFINITY;
        }
        __syncthreads();
        if (has_new && wave == 0 && kslot == 0) {
            float kr[8];
            rope8<T, D>(knew, cs, sub * 8, kr);
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf(qv[e], kr[e], sdot);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
            if (sub == 0) sc_lds[nk - 1] = sdot * scl;
        }
        __syncthreads();
        float sc = t < DF_CHUNK ? sc_lds[t] : -INFINITY;
        const float mx = grp_max(sc);
        float e = t < DF_CHUNK ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;
        const float sum = grp_sum(e);
        if (has_new && t == nk - 1) { red[4] = e; e = 0.f; }
        if (t < DF_CHUNK) sc_lds[t] = e;
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
                float vv[8]; unpack8<T>(vraw[kb][db], vv);
                float x = acc[db];
                x = fmaf(p0.x, vv[0], x); x = fmaf(p0.y, vv[1], x); x = fmaf(p0.z, vv[2], x); x = fmaf(p0.w, vv[3], x);
                x = fmaf(p1.x, vv[4], x); x = fmaf(p1.y, vv[5], x); x = fmaf(p1.z, vv[6], x); x = fmaf(p1.w, vv[7], x);
                acc[db] = x;
            }
        }
        float* ws = part + (size_t)sp2 * WS;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            float x = acc[db];
            x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64);
            if (s8 == 0 && valid) {
                const int d = db * 32 + drow;
                if (has_new) x = fmaf(p_new, to_f32(vnew[d]), x);
                ws[d] = x;
            }
        }
        if (t == 0 && valid) { ws[D] = mx; ws[D + 1] = sum; }
        __syncthreads();                                                    // sc_lds / red are reused by the next chunk
    };

    const int n_iter = (n_split + 1) >> 1;                                  // workgroup-uniform
    for (int it = 0; it < n_iter; it += 2) {
        issue(2 * (it + 1) + grp, kB, vB);                                  // next iteration's chunk on the wire while this one is worked on
        chunk(2 * it + grp, kA, vA);
        if (it + 1 < n_iter) {
            issue(2 * (it + 2) + grp, kA, vA);
            chunk(2 * (it + 1) + grp, kB, vB);
        }
    }
    // the cache append (one workgroup per kv head), kept out of the chunk loop: a conditional store inside it makes hipcc drain the prefetched loads
    // (EXPERIMENTS.md r3-A).  The rotated key is recomputed from the LDS copy — same bits as the score path's.
    if (head % group == 0) {
        if (tid < LPK) { float kr[8]; rope8<T, D>(knew, cs, tid * 8, kr); store8<T>(Kc + (size_t)pos * D + tid * 8, kr); }
        if (tid >= 64 && tid < 64 + D) Vt[(size_t)(tid - 64) * a.s_max + pos] = vnew[tid - 64];
    }
    __syncthreads();
    // ---- merge the chunks in order: flow_attn's merge with the partials read from LDS ---------------------------------------------------------------------
    constexpr int NG = 256 / D;
    constexpr int SPG = DF_MAX_SPLIT / NG;
    const int g = t / D, d = t % D;
    float M = -INFINITY;
    for (int s2 = 0; s2 < n_split; ++s2) M = fmaxf(M, part[s2 * WS + D]);
    float l = 0.f;
    for (int s2 = 0; s2 < n_split; ++s2) { const float m = part[s2 * WS + D]; if (m != -INFINITY) l += __builtin_amdgcn_exp2f(m - M) * part[s2 * WS + D + 1]; }
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < SPG; ++i) {
        const int s2 = g + i * NG;
        if (s2 < n_split) { const float m = part[s2 * WS + D]; if (m != -INFINITY) o += __builtin_amdgcn_exp2f(m - M) * part[s2 * WS + d]; }
    }
    if (grp == 0) mg_o[t] = o;
    __syncthreads();
    if (grp == 0 && g == 0) {
#pragma unroll
        for (int i = 1; i < NG; ++i) o += mg_o[i * D + d];
        store_coherent<T>(reinterpret_cast<T*>(a.attn) + head * D + d, from_f32<T>(l > 0.f ? o / l : 0.f));
    }
}

void launch_decode_attn_head(int dtype, int D, const FlowArgs& a, const FlowStep& sp, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_attn_head: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_attn_head: head_dim must be 64 or 128");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK > a.pos && a.n_split * DF_CHUNK <= a.s_max, "decode_attn_head: n_split must be the number of live 128-key chunks");
    const size_t smem = (size_t)(2 * DF_CHUNK + 16 + 256 + DF_MAX_SPLIT * (D + 4)) * 4 + (size_t)3 * D * 2 + 16;
#define LA(TT, DD) LMX_LAUNCH((decode_attn_head_kernel<TT, DD>), dim3((unsigned)a.nh), dim3(512), smem, st, a, sp)
    if (dtype == kBF16) { if (D == 128) LA(bf16_t, 128); else LA(bf16_t, 64); }
    else { if (D == 128) LA(f16_t, 128); else LA(f16_t, 64); }
#undef LA
    LMX_CHECK_HIP(hipGetLastError());
}

// Attention + o_proj of one layer as ONE launch (the separate-launch decode path): the attention workgroups first, then the o_proj workgroups, which put
// their WHOLE weight rows on the wire (R = 2 rows x P = 8 rounds per wave: all of a 4096-wide row) and wait for the attention step's completion counter.
// The attention launch is a latency chain that leaves the HBM idle for ~10 us; o_proj's 33.5 MB (7B) arrive in that shadow, and after the hand-over o_proj is
// FMAs on registers.  Counters: a.done = [2 parities][2 steps][FLOW_NSUB shards], a.n_steps = 2.
template <typename T, int D>
__global__ __launch_bounds__(256) void decode_attn_o_kernel(FlowArgs a, FlowStep sp_attn, FlowStep sp_o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x, nb_attn = a.off1;
    if (bid < FLOW_NSUB && threadIdx.x < 2) *flow_counter(a, 1 - a.par, (int)threadIdx.x, bid) = 0u;      // re-arm the other parity's shards for the next launch
    if (bid < nb_attn) {
        if (a.attn_form == 1) flow_attn<T, D>(a, sp_attn, bid, 0, 0, nb_attn, smem);
        else flow_attn2<T, D>(a, sp_attn, bid, 0, 0, nb_attn, smem);
    } else flow_linear<T, 2, 8, false>(a, sp_o, bid - nb_attn, 1, nb_attn, a.off2 - nb_attn, smem);
}

void launch_decode_attn_o(int dtype, int D, const FlowArgs& a, const FlowStep& sp_attn, const FlowStep& sp_o, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_attn_o: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_attn_o: head_dim must be 64 or 128");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK > a.pos && a.n_split * DF_CHUNK <= a.s_max && a.done && a.n_steps == 2 &&
                a.off1 == a.nh * a.n_split && a.off2 > a.off1, "decode_attn_o: bad arguments");
    const size_t att = (size_t)(DF_CHUNK + 8 + 2 * DF_MAX_SPLIT + 256) * 4 + (size_t)3 * D * 2 + 16;
    const size_t lin = (size_t)a.xs_bytes + 8 * 4 + 16;
    const size_t smem = att > lin ? att : lin;
#define LA(TT, DD) LMX_LAUNCH((decode_attn_o_kernel<TT, DD>), dim3((unsigned)a.off2), dim3(256), smem, st, a, sp_attn, sp_o)
    if (dtype == kBF16) { if (D == 128) LA(bf16_t, 128); else LA(bf16_t, 64); }
    else { if (D == 128) LA(f16_t, 128); else LA(f16_t, 64); }
#undef LA
    LMX_CHECK_HIP(hipGetLastError());
}

size_t decode_flow_smem(const FlowArgs& a, int D, int es) {
    const size_t lin = (size_t)a.xs_bytes + 8 * 4 + 16;
    const size_t att = (size_t)(DF_CHUNK + 8 + 2 * DF_MAX_SPLIT + 256) * 4 + (size_t)3 * D * es + 16;
    return lin > att ? lin : att;
}

template <typename T, int D>
static int flow_occupancy_t(const FlowArgs& a) {
    auto kern = decode_flow_kernel<T, D>;
    const size_t smem = decode_flow_smem(a, D, sizeof(T));
    LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int occ = 0;
    LMX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, smem));
    return occ;
}

int decode_flow_occupancy(int dtype, int D, const FlowArgs& a) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_flow: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_flow: head_dim must be 64 or 128");
    if (dtype == kBF16) return D == 128 ? flow_occupancy_t<bf16_t, 128>(a) : flow_occupancy_t<bf16_t, 64>(a);
    return D == 128 ? flow_occupancy_t<f16_t, 128>(a) : flow_occupancy_t<f16_t, 64>(a);
}

void launch_decode_flow(int dtype, int D, const FlowArgs& a, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_flow: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_flow: head_dim must be 64 or 128");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK > a.pos && a.n_split * DF_CHUNK <= a.s_max,
                "decode_flow: n_split must be the number of live 128-key chunks");
    LMX_REQUIRE(a.off1 > 0 && a.off2 - a.off1 == a.nh * a.n_split && a.off5 > a.off4 && a.nb4 == a.off5 - a.off4, "decode_flow: inconsistent step sizes");
    static const int deep = [] { const char* e = getenv("LMX_FLOW_DEEP"); return e ? atoi(e) : 2; }();
    const size_t smem = decode_flow_smem(a, D, 2);
    const long grid = (long)a.L * a.off5 + a.nb_head;
    LMX_REQUIRE(grid > 0 && grid < (1l << 31), "decode_flow: bad grid");
#define LF(TT, DD) do { if (deep == 1) LMX_LAUNCH((decode_flow_kernel<TT, DD>), dim3((unsigned)grid), dim3(256), smem, st, a); \
                        else LMX_LAUNCH((decode_flow_kernel_deep2<TT, DD>), dim3((unsigned)grid), dim3(256), smem, st, a); } while (0)
    if (dtype == kBF16) { if (D == 128) LF(bf16_t, 128); else LF(bf16_t, 64); }
    else { if (D == 128) LF(f16_t, 128); else LF(f16_t, 64); }
#undef LF
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
