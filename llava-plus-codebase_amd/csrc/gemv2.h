// gemv2_body: the single-row decode linear of 16-bit models with a hand-counted weight stream (wstream.h) — C[n] = act(norm(x) . W[n]) (+ bias) (+ residual),
// the decode-step linears torch.nn.Linear runs in HF5:models/llama/modeling_llama.py:163-176,243-281 for one new token.  Same block / wave / lane mapping, same
// staging of x (RMSNorm fused, HF rounding points), same per-lane accumulation order, reduction and epilogue as gemv_kernel<T, 1, R> (gemm.hip) — bit-identical
// results — but R x P loads stay on the wire for the whole row: a round is consumed after `s_waitcnt vmcnt(R (P - 1))` and refilled at once, where hipcc's own
// schedule drains to vmcnt(0) before every consume.  16-bit weights only (a lane's 16 bytes = 8 elements).
//
// A __device__ body so that two kernels can carry it: gemv2_kernel (gemm.hip: one linear per launch) and decode_kv_attn_kernel (decode_attn.hip: the k|v
// projection of a decode step next to that step's attention workgroups).  `bid` = this workgroup's index among the linear's workgroups (4 waves x R rows
// each); PUBLISH: the outputs leave as tagged 8-byte granules (gran[n] = {bits of T, tag}) for a consumer inside the SAME launch instead of as C[n].
// GEMV2_NX: 16-byte chunks of x per thread (K <= 2048 NX): 2 for the hidden-width inputs of the 7B model, 4 / 6 / 8 up to 8192 / 12288 / 16384.
#pragma once
#include "common.h"
#include "gemm_common.h"
#include "kernels.h"
#include "wstream.h"

namespace lmx {

constexpr size_t gemv2_smem_bytes(int K, int es) { return (size_t)K * es + 16; }

template <typename T, int R, int P, int GEMV2_NX, bool PUBLISH>
__device__ __forceinline__ void gemv2_body(const GemvArgs& a, int bid, char* smem, unsigned long long* gran, unsigned tag, unsigned long long* ts = nullptr) {
    if (ts && threadIdx.x == 0 && (bid & 31) == 0) __hip_atomic_fetch_min(ts + 6, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // debug timeline (decode_attn.hip)
    static_assert(sizeof(T) == 2, "gemv2: 16-bit weights");
    T* xs = reinterpret_cast<T*>(smem);                       // [K]
    float* red = reinterpret_cast<float*>(smem + (size_t)a.K * sizeof(T));   // 4 floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, KC = K >> 3;
    const int NR = (KC + 63) >> 6;                            // load rounds per row
    const T* __restrict__ X = reinterpret_cast<const T*>(a.X);
    const bool silu = a.act == kActSiluMul;

    const int slot0 = (bid * 4 + wave) * R;            // first "row slot" of this wave
    int rows[R];
    uint32_t roff[R];                                         // byte offset of each row (wave-uniform: scalar registers)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int f;
        if (silu) { const int j = (slot0 + r) >> 1; f = 64 * (j >> 5) + (j & 31) + 32 * ((slot0 + r) & 1); }
        else f = slot0 + r;
        rows[r] = f < a.N ? f : a.N - 1;
        roff[r] = (uint32_t)__builtin_amdgcn_readfirstlane(rows[r]) * (uint32_t)a.ldw * (uint32_t)sizeof(T);
    }
    const ws_v4i rsW = ws_make_rsrc(a.W, 0x7fffffffu);
    // the residual values this wave will add at the very end: requested FIRST (ahead of the weight stream in the wave's load queue, so the hand-made
    // vmcnt counts below stay exact) instead of as a dependent load after the last reduction (~1 us of L2 latency in front of the store)
    const T* Rr = reinterpret_cast<const T*>(a.R);
    const ws_v4i rsR = ws_make_rsrc(Rr ? a.R : a.W, 0x7fffffffu);
    uint32_t rraw[R];                                         // always issued (no residual: a hot line of W), so the counts below do not depend on it
#pragma unroll
    for (int r = 0; r < R; ++r) ws_load_u16(rraw[r], 0u, rsR, (Rr && slot0 + r < a.N) ? (uint32_t)(slot0 + r) * 2u : 0u);

    // ---- x (one short row, L2-resident) is requested ahead of the weights, in the same counted queue: it lands first and is staged / normalised while
    //      the first P weight rounds are still on the wire.  (As plain loads behind the weight issue its wait was a vmcnt(0): staging started only after
    //      the first P rounds had landed too.)
    const ws_v4i rsX = ws_make_rsrc(a.X, 0x7fffffffu);
    ws_u32x4 xraw[GEMV2_NX];
#pragma unroll
    for (int i = 0; i < GEMV2_NX; ++i) {
        const int c = tid + 256 * i;
        if (c < KC) ws_load_plain(xraw[i], (uint32_t)c * 16u, rsX, 0u);
    }
    const ws_v4i rsG = ws_make_rsrc(a.norm_w ? a.norm_w : a.W, 0x7fffffffu);
    ws_u32x4 graw[GEMV2_NX];                                  // RMSNorm weights of the same chunks (same queue position: older than every weight load)
    if (a.norm_w) {
#pragma unroll
        for (int i = 0; i < GEMV2_NX; ++i) {
            const int c = tid + 256 * i;
            if (c < KC) ws_load_plain(graw[i], (uint32_t)c * 16u, rsG, 0u);
        }
    }

    // ---- the first P rounds go out NOW: they do not depend on x --------------------------------------------------------
    ws_u32x4 buf[P][R];
    auto issue = [&](int p, int j) {                          // round j (wave-uniform) into buffer p; past the row: a dummy load of one hot line
        const int c = lane + 64 * j;
        const uint32_t vo = j < NR ? (uint32_t)(c < KC ? c : KC - 1) * 16u : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) ws_load(buf[p][r], vo, rsW, j < NR ? roff[r] : 0u);
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue(p, p);

    // ---- stage x into LDS: plain copy | RMS-normalised (gemv_kernel's arithmetic; x is read from memory ONCE: the chunks go to LDS raw, the statistics
    //      are taken on the way, and each thread normalises in place the chunks it wrote itself — no second global round trip behind the weight loads) ------
    {
        const T* g = reinterpret_cast<const T*>(a.norm_w);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < GEMV2_NX; ++i) {
            const int c = tid + 256 * i;
            if (c < KC) {
                ws_wait1<P * R>(xraw[i]);                     // everything older than the P x R weight loads = every x load
                const ws_u32x4 raw = xraw[i];
                *reinterpret_cast<ws_u32x4*>(xs + c * 8) = raw;
                if (g) {
                    float v[8]; ws_unpack8<T>(raw, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
                }
            }
        }
        if (g) {
            ss = block_sum<4>(ss, red);
            const float inv = rsqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int i = 0; i < GEMV2_NX; ++i) {
                const int c = tid + 256 * i;
                if (c < KC) {
                    float v[8], gv[8];
                    load8<T>(xs + c * 8, v);
                    ws_wait1<P * R>(graw[i]);
                    ws_unpack8<T>(graw[i], gv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * gv[e];
                    store8<T>(xs + c * 8, v);
                }
            }
        }
    }
    __syncthreads();

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int j0 = 0; j0 < NR; j0 += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int j = j0 + p;                             // wave-uniform
            if (j < NR) {
                ws_wait<R * (P - 1), R>(buf[p]);
                const int cc = lane + 64 * j;
                if (cc < KC) {
                    float xv[8]; load8<T>(xs + cc * 8, xv);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float wv[8]; ws_unpack8<T>(buf[p][r], wv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r] = fmaf(wv[e], xv[e], acc[r]);
                    }
                }
                issue(p, j + P);
            }
        }
    }
    ws_drain<P, R>(buf);
    float rres[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ws_landed(rraw[r]); T t; t.x = (uint16_t)rraw[r]; rres[r] = to_f32(t); }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);

    if (lane != 0) return;
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const T* bias = reinterpret_cast<const T*>(a.bias);
    if (silu) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            if constexpr (R >= 2) {
                const int j = (slot0 + r) >> 1;
                if (j >= a.N / 2) continue;
                float g = acc[r], u = acc[r + 1];
                if (bias) { g += to_f32(bias[rows[r]]); u += to_f32(bias[rows[r + 1]]); }
                C[j] = from_f32<T>(act_silu(g) * u);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = slot0 + r;
        if (n >= a.N) continue;
        float v = acc[r];
        if (bias) v += to_f32(bias[n]);
        v = apply_act(v, a.act);
        if (Rr) v += rres[r];
        if constexpr (PUBLISH) {
            // hand-over inside the launch (decode_attn.hip): one naturally aligned 8-byte {value bits, tag} granule per output, ONE relaxed agent-scope store
            // (write-through, observed untorn: MI355X_MICROARCH.md "handoff-1to1"); the consumer polls for the tag
            const T tv = from_f32<T>(v);
            __hip_atomic_store(gran + n, ((unsigned long long)tag << 32) | (unsigned long long)tv.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ts && (bid & 31) == 31 && r == 0) __hip_atomic_fetch_max(ts + 7, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else C[n] = from_f32<T>(v);
    }
}

}  // namespace lmx
