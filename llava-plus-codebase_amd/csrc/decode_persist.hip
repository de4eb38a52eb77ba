// Persistent decode-step kernel: ONE launch per generated token for a single sequence (tensor-parallel world 1).
//
// What it replaces: the 161 launches of Model::decode_step_launch — per layer {qkv GEMV (+RMSNorm), fused RoPE + KV append + split attention,
// o_proj GEMV (+residual), gate|up GEMV (+RMSNorm, SiLU*mul), down GEMV (+residual)} and the lm_head GEMV (+final norm) — i.e. the decoder half of
// LlamaModel.forward for one new token (HF5:models/llama/modeling_llama.py:367-418 via llava_llama.py:88-99).  The pick kernel (argmax | draw, state
// advance, next embedding row) stays a second launch.  Arithmetic is IDENTICAL to the separate kernels (same per-lane accumulation order, same
// rounding points, the attention phase is the very same code: attention_decode.h), so the two paths produce the same bits.
//
// Status: OPT-IN experiment (LMX_DECODE_PERSIST=1), bit-identical to the separate launches (tests/test_decode_persist_gpu.py) but SLOWER on MI355X:
// 3.69-3.89 ms/token against 3.16 ms (LLaVA-1.5-7B, context 1087; profiles/EXPERIMENTS.md r2-Q).  The idea: every kernel boundary of the separate
// launches costs the drain of one kernel plus the ramp of the next, and the small o_proj GEMV cannot keep enough loads in flight on its own; here the
// grid is persistent (G workgroups of 256 threads, all co-resident: G = CUs x 2) and the phases are separated by grid barriers:
//     end of phase k:   stores drained (s_waitcnt vmcnt(0))  ->  FIRST WEIGHT LOADS OF PHASE k+1 ISSUED  ->  arrive  ->  wait  ->  stage x  ->  stream
// A wave owns R weight rows per work item and keeps P rounds of 16-byte loads in flight (R x P >= 6).  What it costs instead: the 8 XCDs' L2s are not
// coherent with each other, so every activation row crosses the barrier through the memory side (write-through stores, agent-scope loads: ~4 us of
// serialized latency per phase for arrive + poll + x staging) and the kernel is register-bound at 2 workgroups per CU (8 waves), fewer loads in
// flight per CU than the stand-alone GEMVs reach.  A kernel boundary does the same hand-over in ~3 us.
//
// Coherence: every store of an activation is write-through (sc1) and every load of one bypasses the non-coherent levels (sc1); weights and the KV
// cache of earlier tokens were written by earlier launches and use plain / non-temporal loads.  The barrier is two-level (32 workgroups per group
// counter, one root counter; agent-scope read-modify-writes on separate cache lines), monotonic across launches (the host passes the epoch) and
// bounded: a workgroup that waits longer than ~2 s raises a status word and every workgroup leaves.  Launches of different sequences are serialised
// on the device by the host (an event chain), so two persistent grids never compete for the CUs.  The per-step operands live in a device table
// (PersistStep) so that only the current step's pointers occupy registers.
#include "attention_decode.h"
#include "common.h"
#include "kernels.h"

namespace lmx {

namespace {

typedef uint32_t u32x4_p __attribute__((ext_vector_type(4)));

template <typename T> struct RawW {                 // 8 weight elements as loaded (16 bytes), converted at use
    u32x4_p v;
    __device__ __forceinline__ void load(const T* p) { v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_p*>(p)); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        f[0] = unpack_lo<T>(v.x); f[1] = unpack_hi<T>(v.x); f[2] = unpack_lo<T>(v.y); f[3] = unpack_hi<T>(v.y);
        f[4] = unpack_lo<T>(v.z); f[5] = unpack_hi<T>(v.z); f[6] = unpack_lo<T>(v.w); f[7] = unpack_hi<T>(v.w);
    }
};

// 16 bytes / one element that another XCD wrote earlier in this launch: agent-scope (sc1) loads bypass the non-coherent cache levels
template <typename T> __device__ __forceinline__ void load8_coherent(const T* p, float (&f)[8]) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f[0] = unpack_lo<T>(w[0]); f[1] = unpack_hi<T>(w[0]); f[2] = unpack_lo<T>(w[1]); f[3] = unpack_hi<T>(w[1]);
    f[4] = unpack_lo<T>(w[2]); f[5] = unpack_hi<T>(w[2]); f[6] = unpack_lo<T>(w[3]); f[7] = unpack_hi<T>(w[3]);
}
template <typename T> __device__ __forceinline__ float load_coherent(const T* p) {
    const unsigned short u = __hip_atomic_load(reinterpret_cast<const unsigned short*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    T t; *reinterpret_cast<unsigned short*>(&t) = u;
    return to_f32(t);
}

// ---- grid barrier ----------------------------------------------------------------------------------------------------------------------------
// bar[0]: root counter; bar[32 * (1 + g)]: counter of group g (32 workgroups).  All counters only ever grow: barrier number e (1-based, counted over
// the life of the model) is complete when root == e * n_groups.  Arrivals are agent-scope read-modify-writes (the ordering point the attention merge of
// decode_fused_kernel relies on as well); a workgroup that waits > 2 s raises the abort word and every workgroup leaves.
__device__ __forceinline__ bool grid_sync(const PersistArgs& a, unsigned& epoch, int* abort_lds) {
    epoch += 1;
    __syncthreads();                                       // every wave of this workgroup has drained its stores (callers wait vmcnt(0) first)
    if (threadIdx.x == 0) {
        if (a.fence_mode == 1) __atomic_thread_fence(__ATOMIC_RELEASE);        // experiment arm: full agent release (buffer_wbl2) before arriving
        const unsigned G = gridDim.x, grp = blockIdx.x >> 5, ngrp = (G + 31) >> 5;
        const unsigned gsize = (grp + 1) * 32 <= G ? 32u : G - grp * 32;
        const unsigned old = __hip_atomic_fetch_add(a.bar + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * gsize) __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * ngrp;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int abort = 0;
        while ((int)(__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                __hip_atomic_store(a.abort_word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // 2 s at 100 MHz: some workgroup of the grid is not running
                __hip_atomic_store(a.status, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                abort = 1;
                break;
            }
        }
        *abort_lds = abort;
    }
    __syncthreads();
    return *abort_lds == 0;
}

// ---- one linear phase: C = act(norm(x) W^T) (+ residual), rows spread over all waves of the grid ------------------------------------------------
// Same mapping and arithmetic as gemv_kernel<T, 1, R> (gemm.hip): slot s = rows [s R, (s + 1) R) (SiLU*mul: R / 2 gate/up pairs of the fused
// [32 gate | 32 up] layout), lane l accumulates the 8-element chunks l, l + 64, l + 128, ... in that order, wave_sum, lane 0 writes.
template <typename T, int R, int P, bool SILU>
__device__ __forceinline__ bool linear_phase(const PersistArgs& a, const T* __restrict__ W, int N, int K, const T* x, const T* norm_w, const T* res, T* C,
                                             bool sync, unsigned& epoch, T* xs, float* red, int* abort_lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KC = K >> 3;
    const int waves = gridDim.x * 4;
    const int nslots = (N + R - 1) / R;
    int slot = blockIdx.x * 4 + wave;
    const T* wrow[R];
    int rows[R];
    RawW<T> buf[P][R];
    auto set_rows = [&](int s) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int f;
            if (SILU) { const int j = (s * R + r) >> 1; f = 64 * (j >> 5) + (j & 31) + 32 * ((s * R + r) & 1); }
            else f = s * R + r;
            rows[r] = f < N ? f : N - 1;
            wrow[r] = W + (size_t)rows[r] * K;
        }
    };
    auto issue = [&](int p, int c) {
        if (c < KC) {
#pragma unroll
            for (int r = 0; r < R; ++r) buf[p][r].load(wrow[r] + (size_t)c * 8);
        }
    };
    bool has = slot < nslots;
    if (has) {
        set_rows(slot);
#pragma unroll
        for (int p = 0; p < P; ++p) issue(p, lane + 64 * p);                   // in flight across the barrier below
    }
    if (sync && !grid_sync(a, epoch, abort_lds)) return false;
    // ---- stage x in LDS (coherent loads; one pass — the chunks stay in registers between the sum of squares and the normalisation; RMSNorm with
    //      HF's rounding points) ---------------------------------------------------------------------------------------------------------------
    {
        constexpr int XC = 8;                                  // chunks of 8 elements per thread: K <= 256 * 8 * 8 = 16384
        float xv[XC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int c = tid + 256 * i;
            if (c < KC) {
                load8_coherent<T>(x + c * 8, xv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += xv[i][e] * xv[i][e];
            }
        }
        float inv = 1.f;
        if (norm_w) {
            ss = block_sum<4>(ss, red);
            inv = rsqrtf(ss / (float)K + a.eps);
        }
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int c = tid + 256 * i;
            if (c < KC) {
                if (norm_w) {
                    float gv[8]; load8<T>(norm_w + c * 8, gv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[i][e] = round_to<T>(xv[i][e] * inv) * gv[e];
                }
                store8<T>(xs + c * 8, xv[i]);
            }
        }
    }
    __syncthreads();
    while (has) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        for (int c = lane; c < KC; c += 64 * P) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int cc = c + 64 * p;
                if (cc < KC) {
                    float xv[8]; load8<T>(xs + cc * 8, xv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float wv[8]; buf[p][r].unpack(wv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r] = fmaf(wv[e], xv[e], acc[r]);
                    }
                    issue(p, cc + 64 * P);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        const int slot0 = slot * R;
        if (lane == 0) {
            if (SILU) {
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const int j = (slot0 + r) >> 1;
                    if (j < N / 2) store_coherent<T>(C + j, from_f32<T>(act_silu(acc[r]) * acc[r + 1]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = slot0 + r;
                    if (n < N) {
                        float v = acc[r];
                        if (res) v += load_coherent<T>(res + n);
                        store_coherent<T>(C + n, from_f32<T>(v));
                    }
                }
            }
        }
        slot += waves;
        has = slot < nslots;
        if (has) {
            set_rows(slot);
#pragma unroll
            for (int p = 0; p < P; ++p) issue(p, lane + 64 * p);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores are acknowledged before it arrives at the next barrier
    return true;
}

template <typename T, bool SILU>
__device__ __forceinline__ bool linear_dispatch(int R, const PersistArgs& a, const T* W, int N, int K, const T* x, const T* norm_w, const T* res, T* C, bool sync,
                                                unsigned& epoch, T* xs, float* red, int* abort_lds) {
    if (SILU) {
        switch (R) {
            case 2: return linear_phase<T, 2, 3, true>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
            case 4: return linear_phase<T, 4, 2, true>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
            default: return linear_phase<T, 6, 2, true>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
        }
    }
    switch (R) {
        case 1: return linear_phase<T, 1, 6, false>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
        case 2: return linear_phase<T, 2, 3, false>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
        case 3: return linear_phase<T, 3, 2, false>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
        default: return linear_phase<T, 4, 2, false>(a, W, N, K, x, norm_w, res, C, sync, epoch, xs, red, abort_lds);
    }
}

}  // namespace

template <typename T, int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void decode_step_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                                        // [max(H, I)]
    float* red = reinterpret_cast<float*>(smem + (size_t)a.xs_elems * sizeof(T));
    __shared__ __attribute__((aligned(16))) T qkv_s[3 * D];
    __shared__ int abort_lds;
    const int tid = threadIdx.x;
    if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // an earlier launch timed out: do nothing
    unsigned epoch = a.epoch0;
    const int G = gridDim.x;
    const int n_steps = a.dbg_steps > 0 && a.dbg_steps < a.n_steps ? a.dbg_steps : a.n_steps;
    for (int step = 0; step < n_steps; ++step) {
        const PersistStep& sp = a.steps[step];
        if (sp.kind == 2) {
            // RoPE + KV append + split attention + merge (the code of decode_fused_kernel; q / k_new / v_new fetched coherently into LDS)
            if (!grid_sync(a, epoch, &abort_lds)) return;
            T* qkv = reinterpret_cast<T*>(a.qkv);
            DecodeFusedArgs fa{qkv, sp.kc, sp.vt, a.rope, a.pos_ptr, a.nh, a.nkv, a.s_max, a.n_split, a.scale, a.aws, a.cnt, a.attn};
            const int group = a.nh / a.nkv, nvb = a.nh * a.n_split;
            for (int vb = blockIdx.x; vb < nvb; vb += G) {
                const int head = vb % a.nh, split = vb / a.nh, kvh = head / group;
                __syncthreads();                                               // LDS of the previous item is free
                if (tid < 3 * D / 8) {
                    const int part = tid / (D / 8), c = tid % (D / 8);
                    const int col = (part == 0 ? head : part == 1 ? a.nh + kvh : a.nh + a.nkv + kvh) * D + c * 8;
                    float v[8]; load8_coherent<T>(qkv + col, v);
                    store8<T>(qkv_s + part * D + c * 8, v);
                }
                __syncthreads();
                decode_fused_body<T, D, true>(fa, head, split, 0, qkv_s, qkv_s + D, qkv_s + 2 * D);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        const T* W = reinterpret_cast<const T*>(sp.W); const T* x = reinterpret_cast<const T*>(sp.x);
        const T* nw = reinterpret_cast<const T*>(sp.norm_w); const T* res = reinterpret_cast<const T*>(sp.res);
        T* C = reinterpret_cast<T*>(sp.C);
        const int N = sp.N, K = sp.K, R = sp.R;
        const bool sync = step > 0;
        bool ok;
        if (sp.kind == 1) ok = linear_dispatch<T, true>(R, a, W, N, K, x, nw, res, C, sync, epoch, xs, red, &abort_lds);
        else ok = linear_dispatch<T, false>(R, a, W, N, K, x, nw, res, C, sync, epoch, xs, red, &abort_lds);
        if (!ok) return;
    }
}

// barriers one launch performs (the host advances its epoch by this much per launch)
int decode_persist_barriers(int L) { return 5 * L; }

static size_t persist_smem(const PersistArgs& a, int es) { return (size_t)a.xs_elems * es + 16; }

template <typename T, int D>
static int persist_occupancy_t(const PersistArgs& a) {
    auto kern = decode_step_kernel<T, D>;
    const size_t smem = persist_smem(a, sizeof(T));
    LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int occ = 0;
    LMX_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, smem));
    return occ;
}

int decode_persist_occupancy(int dtype, int D, const PersistArgs& a) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_persist: 16-bit dtypes only");
    LMX_REQUIRE(D == 64 || D == 128, "decode_persist: head_dim must be 64 or 128");
    if (dtype == kBF16) return D == 128 ? persist_occupancy_t<bf16_t, 128>(a) : persist_occupancy_t<bf16_t, 64>(a);
    return D == 128 ? persist_occupancy_t<f16_t, 128>(a) : persist_occupancy_t<f16_t, 64>(a);
}

void launch_decode_persist(int dtype, int D, const PersistArgs& a, int grid, hipStream_t st) {
    LMX_REQUIRE(dtype == kBF16 || dtype == kF16, "decode_persist: 16-bit dtypes only");
    LMX_REQUIRE(a.n_split >= 1 && a.n_split <= DF_MAX_SPLIT && a.n_split * DF_CHUNK >= a.s_max, "decode_persist: n_split must cover s_max in 128-key chunks");
    const size_t smem = persist_smem(a, 2);
#define LP(TT, DD) hipLaunchKernelGGL((decode_step_kernel<TT, DD>), dim3(grid), dim3(256), smem, st, a)
    if (dtype == kBF16) { if (D == 128) LP(bf16_t, 128); else LP(bf16_t, 64); }
    else { if (D == 128) LP(f16_t, 128); else LP(f16_t, 64); }
#undef LP
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
