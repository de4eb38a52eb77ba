// Token sampling on the device: temperature -> top-k -> top-p (nucleus) -> multinomial, the warper order of the reference's
// generate(do_sample=True, temperature, top_p) (llava/serve/model_worker.py:156-184 -> HF GenerationMixin sample():
// TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper, then softmax + multinomial).  One 1024-thread workgroup per
// logits row; the row (64 KB at V = 32000) is re-read from L2 in every pass instead of being staged, so any V works.
//
// Everything that decides WHICH tokens survive is exact integer arithmetic, so the result does not depend on the order in
// which threads add things up:
//   e_i   = exp((l_i - max l) / T)            fp32, in (0, 1]
//   key_i = bit pattern of e_i                monotone in e_i
//   q_i   = floor(e_i * 2^31)                 fixed-point mass (32 bits), sums in 64 bits
//   top-k : radix select (4 x 8 bits, histogram of COUNTS from the top) of the k-th largest key; ties at the threshold all stay
//           (`scores < kth` is what the reference removes)
//   top-p : radix select (histogram of MASS from the bottom) of the smallest key whose ascending inclusive mass exceeds
//           (1 - top_p) * Z  — the reference removes sorted_cumsum <= 1 - top_p, so the most likely token always stays
//   draw  : u in [0, 1) from Philox-4x32-10(seed, tokens produced so far), inverse CDF over the survivors in token-id order:
//           the first id whose inclusive prefix mass exceeds u * Z_kept.
// torch.multinomial draws differently (exponential race), so ids are not comparable with a torch run for the same seed;
// tests check the survivor set against HF's own warpers and the drawn id against the float64 inverse CDF for the same u.
#include "common.h"
#include "kernels.h"

namespace lmx {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ uint32_t philox_u32(uint32_t seed_lo, uint32_t seed_hi, uint32_t offset) {
    uint32_t c[4] = {offset, 0u, 0x6c6d7821u, 0u};
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    return c[0];
}

// block-wide (1024 threads = 16 waves) reductions on small LDS scratch
__device__ __forceinline__ float block_max_1024(float v, float* red16) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red16[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red16[i]);
    __syncthreads();
    return m;
}
__device__ __forceinline__ uint64_t block_sum_u64_1024(uint64_t v, uint64_t* red16) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red16[i];
    __syncthreads();
    return s;
}

struct SampleScratch {
    float redf[16];
    uint64_t redu[16];
    uint64_t hist[256];
    uint64_t scan[1024];
    uint32_t pick[2];
};

// Returns the sampled token id (valid in every thread).  keep_out (debug, may be null): 1 for every surviving token id.
template <typename T>
__device__ int64_t sample_row(const T* __restrict__ logits, int V, float temperature, float top_p, int top_k, uint32_t u32,
                              SampleScratch& sh, uint8_t* __restrict__ keep_out) {
    const int tid = threadIdx.x;
    const float inv_t = 1.f / fmaxf(temperature, 1e-5f);
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, to_f32(logits[i]));
    mx = block_max_1024(mx, sh.redf);
    auto e_of = [&](int i) { return expf((to_f32(logits[i]) - mx) * inv_t); };
    auto q_of = [&](float e) { return (uint64_t)(uint32_t)(e * 2147483648.f); };

    // ---- top-k: key of the k-th largest e ----------------------------------------------------------------------------------
    uint32_t thr_key = 0;                                   // survivors: key >= thr_key
    if (top_k > 0 && top_k < V) {
        uint32_t prefix = 0, remaining = (uint32_t)top_k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += 1024) sh.hist[b] = 0;
            __syncthreads();
            const uint32_t mask_hi = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int i = tid; i < V; i += 1024) {
                const uint32_t k = __float_as_uint(e_of(i));
                if ((k & mask_hi) == prefix) atomicAdd(reinterpret_cast<unsigned long long*>(&sh.hist[(k >> shift) & 255u]), 1ull);
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t cum = 0; int b = 255;
                for (; b > 0; --b) { const uint32_t c = (uint32_t)sh.hist[b]; if (cum + c >= remaining) break; cum += c; }
                sh.pick[0] = (uint32_t)b; sh.pick[1] = remaining - cum;
            }
            __syncthreads();
            prefix |= sh.pick[0] << shift; remaining = sh.pick[1];
            __syncthreads();
        }
        thr_key = prefix;
    }

    // ---- top-p over the top-k survivors -----------------------------------------------------------------------------------------
    if (top_p < 1.f) {
        uint64_t z = 0;
        for (int i = tid; i < V; i += 1024) { const float e = e_of(i); if (__float_as_uint(e) >= thr_key) z += q_of(e); }
        z = block_sum_u64_1024(z, sh.redu);
        const double cut = (1.0 - (double)top_p) * (double)z;
        const uint64_t thr_mass = cut <= 0.0 ? 0ull : (uint64_t)cut;          // remove while ascending inclusive mass <= thr_mass
        uint32_t prefix = 0; uint64_t carried = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += 1024) sh.hist[b] = 0;
            __syncthreads();
            const uint32_t mask_hi = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int i = tid; i < V; i += 1024) {
                const float e = e_of(i);
                const uint32_t k = __float_as_uint(e);
                if (k >= thr_key && (k & mask_hi) == prefix)
                    atomicAdd(reinterpret_cast<unsigned long long*>(&sh.hist[(k >> shift) & 255u]), (unsigned long long)q_of(e));
            }
            __syncthreads();
            if (tid == 0) {
                uint64_t cum = carried; int b = 0;
                for (; b < 255; ++b) { if (cum + sh.hist[b] > thr_mass) break; cum += sh.hist[b]; }
                sh.pick[0] = (uint32_t)b; sh.redu[0] = cum;
            }
            __syncthreads();
            prefix |= sh.pick[0] << shift; carried = sh.redu[0];
            __syncthreads();
        }
        if (prefix > thr_key) thr_key = prefix;
    }

    // ---- inverse-CDF draw over the survivors, token-id order ----------------------------------------------------------------------
    const int C = (V + 1023) / 1024;
    const int i0 = tid * C, i1 = (i0 + C < V) ? i0 + C : V;
    uint64_t mine = 0;
    for (int i = i0; i < i1; ++i) { const float e = e_of(i); if (__float_as_uint(e) >= thr_key) mine += q_of(e); }
    if (keep_out) for (int i = i0; i < i1; ++i) keep_out[i] = __float_as_uint(e_of(i)) >= thr_key ? 1 : 0;
    sh.scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {               // inclusive Hillis-Steele scan of the 1024 chunk masses
        const uint64_t add = tid >= off ? sh.scan[tid - off] : 0ull;
        __syncthreads();
        sh.scan[tid] += add;
        __syncthreads();
    }
    const uint64_t total = sh.scan[1023];
    // floor(u32 * total / 2^32) without 128-bit arithmetic (total < 2^48): always < total
    const uint64_t target = (uint64_t)u32 * (total >> 32) + (((uint64_t)u32 * (total & 0xFFFFFFFFull)) >> 32);
    const uint64_t incl = sh.scan[tid], excl = incl - mine;
    if (tid == 0) sh.pick[0] = 0;
    __syncthreads();
    if (mine > 0 && excl <= target && target < incl) {
        uint64_t run = excl; int tok = i0;
        for (int i = i0; i < i1; ++i) {
            const float e = e_of(i);
            if (__float_as_uint(e) >= thr_key) { run += q_of(e); if (run > target) { tok = i; break; } }
        }
        sh.pick[0] = (uint32_t)tok;
    }
    __syncthreads();
    const int64_t r = (int64_t)sh.pick[0];
    __syncthreads();
    return r;
}

// ---- fast path: V <= FAST_NT * CPT — every thread keeps its CPT consecutive e_i in registers for the whole kernel ----------------------
// The row is read once (16-byte loads); the two thresholds come from bisection over the 32-bit key space with block-wide integer
// reductions (count >= k for top-k, fixed-point mass >= need for top-p): 2 x 30 reductions of ~0.1 us instead of histogram passes
// with contended LDS atomics.  Same survivor set as the generic path by construction (same keys, same masses, same criteria).
constexpr int FAST_NT = 1024;       // threads of the register-cached path
struct FastScratch { uint64_t red[2][16]; uint64_t scan[FAST_NT]; uint32_t pick[2]; float redf[16]; };

template <typename T, int CPT>
__device__ int64_t sample_row_cached(const T* __restrict__ logits, int V, float temperature, float top_p, int top_k, uint32_t u32,
                                     FastScratch& sh, uint8_t* __restrict__ keep_out) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = tid * CPT;
    const float inv_t = 1.f / fmaxf(temperature, 1e-5f);
    float e[CPT];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CPT; c += 8) {
        const int i = i0 + c;
        if (i + 8 <= V) {
            float v[8]; load8<T>(logits + i, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { e[c + j] = v[j]; mx = fmaxf(mx, v[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { e[c + j] = i + j < V ? to_f32(logits[i + j]) : -INFINITY; mx = fmaxf(mx, e[c + j]); }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) sh.redf[wv] = mx;
    __syncthreads();
    mx = sh.redf[0];
#pragma unroll
    for (int w = 1; w < FAST_NT / 64; ++w) mx = fmaxf(mx, sh.redf[w]);
    // per element: key = bit pattern of e (monotone), q = floor(e * 2^31) fixed-point mass (32 bits; sums in 64)
    uint32_t key[CPT], q[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const float x = i0 + c < V ? expf((e[c] - mx) * inv_t) : 0.f;
        key[c] = __float_as_uint(x);
        q[c] = i0 + c < V ? (uint32_t)(x * 2147483648.f) : 0u;
    }

    int nred = 0;                                           // ping-pong LDS buffers: one barrier per reduction
    auto block_sum = [&](uint64_t v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
            v += ((uint64_t)hi << 32) | lo;
        }
        uint64_t* buf = sh.red[nred & 1]; ++nred;
        if (lane == 0) buf[wv] = v;
        __syncthreads();
        uint64_t s = 0;
#pragma unroll
        for (int w = 0; w < FAST_NT / 64; ++w) s += buf[w];
        return s;
    };

    uint32_t thr_key = 0;
    if (top_k > 0 && top_k < V) {
        uint32_t t = 0;
        for (int bit = 29; bit >= 0; --bit) {               // keys of e in (0, 1] are <= 0x3F800000: bits 31 / 30 are never set
            const uint32_t cand = t | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) cnt += (key[c] >= cand && i0 + c < V) ? 1u : 0u;
            if (block_sum(cnt) >= (uint64_t)top_k) t = cand;
        }
        thr_key = t;                                        // the k-th largest key (largest threshold that still keeps >= k)
    }
    if (top_p < 1.f) {
        uint64_t z = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) if (key[c] >= thr_key) z += q[c];
        z = block_sum(z);
        const double cut = (1.0 - (double)top_p) * (double)z;
        const uint64_t thr_mass = cut <= 0.0 ? 0ull : (uint64_t)cut;
        const uint64_t need = z - thr_mass;                 // keep key v iff (mass of survivors with key > v) < need
        uint32_t t = 0;
        for (int bit = 29; bit >= 0; --bit) {
            const uint32_t cand = t | (1u << bit);
            const uint32_t lo = cand > thr_key ? cand : thr_key;
            uint64_t m = 0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) if (key[c] >= lo) m += q[c];
            if (block_sum(m) >= need) t = cand;
        }
        if (t > thr_key) thr_key = t;
    }

    uint64_t mine = 0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) if (key[c] >= thr_key) mine += q[c];
    if (keep_out) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) if (i0 + c < V) keep_out[i0 + c] = key[c] >= thr_key ? 1 : 0;
    }
    __syncthreads();
    sh.scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < FAST_NT; off <<= 1) {
        const uint64_t add = tid >= off ? sh.scan[tid - off] : 0ull;
        __syncthreads();
        sh.scan[tid] += add;
        __syncthreads();
    }
    const uint64_t total = sh.scan[FAST_NT - 1];
    const uint64_t target = (uint64_t)u32 * (total >> 32) + (((uint64_t)u32 * (total & 0xFFFFFFFFull)) >> 32);
    const uint64_t incl = sh.scan[tid], excl = incl - mine;
    if (tid == 0) sh.pick[0] = 0;
    __syncthreads();
    if (mine > 0 && excl <= target && target < incl) {
        uint64_t run = excl; int tok = i0; bool found = false;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (!found && key[c] >= thr_key && q[c] > 0) { run += q[c]; if (run > target) { tok = i0 + c; found = true; } }
        }
        sh.pick[0] = (uint32_t)tok;
    }
    __syncthreads();
    const int64_t r = (int64_t)sh.pick[0];
    __syncthreads();
    return r;
}

union SampleShared { SampleScratch generic; FastScratch fast; };

// CPT = elements per thread of the register-cached path (host picks the smallest that covers V), 0 = generic streaming path
template <typename T, int CPT>
__device__ __forceinline__ int64_t sample_any(const T* logits, int V, float temperature, float top_p, int top_k, uint32_t u, SampleShared& sh,
                                              uint8_t* keep_out) {
    if constexpr (CPT > 0) return sample_row_cached<T, CPT>(logits, V, temperature, top_p, top_k, u, sh.fast, keep_out);
    else return sample_row<T>(logits, V, temperature, top_p, top_k, u, sh.generic, keep_out);
}
static int sample_cpt(int V) { return V <= FAST_NT * 32 ? 32 : 0; }

template <typename T, int CPT>
__global__ __launch_bounds__(CPT > 0 ? FAST_NT : 1024) void sample_kernel(const T* __restrict__ logits, int V, SampleParams p, const int* __restrict__ offset_ptr,
                                                      uint32_t u32_override, int use_override, int64_t* __restrict__ out_tok,
                                                      uint8_t* __restrict__ keep_out) {
    __shared__ SampleShared sh;
    const uint32_t off = offset_ptr ? (uint32_t)*offset_ptr : 0u;
    const uint32_t u = use_override ? u32_override : philox_u32(p.seed_lo, p.seed_hi, off);
    const int64_t t = sample_any<T, CPT>(logits, V, p.temperature, p.top_p, p.top_k, u, sh, keep_out);
    if (threadIdx.x == 0) *out_tok = t;
}

void launch_sample(int dtype, const void* logits, int V, const SampleParams& p, const int* offset_ptr, int64_t* out_tok,
                   const uint32_t* u32_override, uint8_t* keep_out, hipStream_t st) {
    LMX_REQUIRE(V > 0 && p.temperature > 0.f && p.top_p > 0.f, "sample: temperature and top_p must be positive");
#define L2(TT, CC) hipLaunchKernelGGL((sample_kernel<TT, CC>), dim3(1), dim3(CC > 0 ? FAST_NT : 1024), 0, st, (const TT*)logits, V, p, offset_ptr, \
                                     u32_override ? *u32_override : 0u, u32_override ? 1 : 0, out_tok, keep_out)
#define L(TT) do { if (sample_cpt(V) == 32) L2(TT, 32); else L2(TT, 0); } while (0)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
#undef L2
    LMX_CHECK_HIP(hipGetLastError());
}

// End of a decode step, one workgroup per sequence: pick the next token (greedy argmax or a draw), advance the sequence state
// (*len += 1, token log) and fetch the picked token's embedding row into the residual-stream row of the NEXT step — one launch
// instead of argmax + advance + embedding gather.  `tab` (decode batch) or `single` (one sequence, by value) describes the state.
template <typename T, int CPT>
__global__ __launch_bounds__(CPT > 0 ? FAST_NT : 1024) void pick_advance_batch_kernel(const T* __restrict__ logits_all, int V, int ld, const SeqStateRef* __restrict__ tab,
                                                                  SeqStateRef single, int64_t* __restrict__ ids_out,
                                                                  const T* __restrict__ embed, T* __restrict__ h_out, int H) {
    __shared__ SampleShared sh;
    __shared__ float bv[16];
    __shared__ int bi[16];
    const T* logits = logits_all + (size_t)blockIdx.x * ld;      // V = ids scanned (real vocabulary), ld = row pitch (padded)
    const SeqStateRef r = tab ? tab[blockIdx.x] : single;
    // stopped earlier (device-side stop rule): nothing of this sequence moves any more — workgroup-uniform, read before anyone could write it below
    if (r.stop && r.stop->done != 0) {
        if (threadIdx.x == 0 && ids_out) ids_out[blockIdx.x] = -1;
        return;
    }
    __syncthreads();
    int64_t t;
    if (r.sample.temperature > 0.f) {
        const uint32_t u = philox_u32(r.sample.seed_lo, r.sample.seed_hi, (uint32_t)*r.n_out);
        t = sample_any<T, CPT>(logits, V, r.sample.temperature, r.sample.top_p, r.sample.top_k, u, sh, nullptr);
    } else {
        float best = -INFINITY; int idx = 0x7fffffff;
        constexpr int NT = CPT > 0 ? FAST_NT : 1024;
        const int V8 = V & ~7;
        for (int i = threadIdx.x * 8; i < V8; i += NT * 8) {            // 16-byte loads; ascending ids inside a thread keep "first wins"
            float v[8]; load8<T>(logits + i, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (v[j] > best) { best = v[j]; idx = i + j; }
        }
        for (int i = V8 + threadIdx.x; i < V; i += NT) {
            const float v = to_f32(logits[i]);
            if (v > best || (v == best && i < idx)) { best = v; idx = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
        }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) { bv[w] = best; bi[w] = idx; }
        __syncthreads();
        for (int i = 0; i < (CPT > 0 ? FAST_NT : 1024) / 64; ++i)
            if (bv[i] > best || (bv[i] == best && bi[i] < idx)) { best = bv[i]; idx = bi[i]; }
        t = idx == 0x7fffffff ? 0 : idx;
    }
    if (threadIdx.x == 0) {
        *r.tok = t;
        if (ids_out) ids_out[blockIdx.x] = t;
        *r.len += 1;
        const int n = *r.n_out;
        if (r.log && n < r.log_cap) r.log[n] = t;
        *r.n_out = n + 1;
        if (r.stop && stop_rule_fires(r.stop, t, r.log, n + 1, r.log_cap)) r.stop->done = 1;      // the reference's id rules on the sequence INCLUDING t
    }
    if (embed) {                                            // next step's input row (every thread knows t)
        constexpr int NT = CPT > 0 ? FAST_NT : 1024;
        int64_t tt = t < 0 ? 0 : (t >= V ? V - 1 : t);
        const uint4* src = reinterpret_cast<const uint4*>(embed + (size_t)tt * H);
        uint4* dst = reinterpret_cast<uint4*>(h_out + (size_t)blockIdx.x * H);
        for (int c = threadIdx.x; c < H * (int)sizeof(T) / 16; c += NT) dst[c] = src[c];
    }
}

void launch_argmax_advance_batch(int dtype, const void* logits, int V, int ld, const SeqStateRef* tab, const SeqStateRef* single, int n, int64_t* ids_out,
                                 const void* embed, void* h_out, int H, hipStream_t st) {
    LMX_REQUIRE((tab != nullptr) != (single != nullptr) && (tab || n == 1), "pick: give a device table or one by-value state");
    const SeqStateRef one = single ? *single : SeqStateRef{};
#define L2(TT, CC) hipLaunchKernelGGL((pick_advance_batch_kernel<TT, CC>), dim3(n), dim3(CC > 0 ? FAST_NT : 1024), 0, st, (const TT*)logits, V, ld, tab, one, \
                                      ids_out, (const TT*)embed, (TT*)h_out, H)
#define L(TT) do { if (sample_cpt(V) == 32) L2(TT, 32); else L2(TT, 0); } while (0)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
#undef L2
    LMX_CHECK_HIP(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------------------------
// Beam search, device half (GenerationMixin.beam_search of the reference's transformers: next_token_scores = log_softmax(logits) + beam_scores[:, None],
// then topk(2 * num_beams) over the flattened [num_beams * V] scores).  One 1024-thread workgroup per beam row computes the row's log-sum-exp in
// fp32 and its K best candidates in (score descending, token id ascending) order — the union of the rows' top-K contains the global top-K, the host
// merges num_beams x K pairs.  K passes over an L2-resident row: each pass picks the best element strictly after the previous pick in that order.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void beam_topk_kernel(const T* __restrict__ logits, int ld, int V, const float* __restrict__ beam_scores, int K,
                                                         float* __restrict__ out_scores, int* __restrict__ out_ids) {
    __shared__ float redf[16];
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* x = logits + (size_t)row * ld;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, to_f32(x[i]));
    mx = block_max_1024(mx, redf);
    float sum = 0.f;
    for (int i = tid; i < V; i += 1024) sum += expf(to_f32(x[i]) - mx);
    // block sum (16 waves)
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += redf[w];
    const float lse = mx + logf(tot);
    const float base = beam_scores ? beam_scores[row] : 0.f;
    float pv = INFINITY; int pi = -1;                        // previous pick: everything is "after" (+inf, -1)
    for (int k = 0; k < K; ++k) {
        float best = -INFINITY; int besti = 0x7fffffff;
        for (int i = tid; i < V; i += 1024) {
            const float v = to_f32(x[i]);
            const bool after = v < pv || (v == pv && i > pi);
            if (after && (v > best || (v == best && i < besti))) { best = v; besti = i; }
        }
        // wave then block reduction of (value desc, index asc)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        __syncthreads();
        if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
        __syncthreads();
        best = bv[0]; besti = bi[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        if (tid == 0) {
            const bool ok = besti != 0x7fffffff;
            out_scores[(size_t)row * K + k] = ok ? (best - lse) + base : -INFINITY;
            out_ids[(size_t)row * K + k] = ok ? besti : -1;
        }
        pv = best; pi = besti;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Device half of a beam-SAMPLE step (GenerationMixin.beam_sample of the reference's transformers 4.31: num_beams > 1 with do_sample=True, reachable through
// llava/eval/run_llava.py:115-125 whose default temperature is 0.2).  HF: scores = warpers(log_softmax(logits) + beam_score) per beam row (temperature divides
// the summed score; top-k / top-p keep a survivor set per row), probs = softmax over the num_beams x V block, 2 num_beams draws WITHOUT replacement, then the
// drawn candidates are ranked by score.  Drawing m items without replacement from a categorical distribution is the Plackett-Luce order, which is exactly
// the top-m of (log-weight + Gumbel noise): per row this kernel forms s_i = (log_softmax(logits)_i + beam_score) / T for the surviving ids (keep, from the
// sampler's own warpers: sample_row's keep_out), key_i = s_i + G_i with G_i = -log(-log u_i), u_i from Philox-4x32-10(seed, counter0 + row V + i), and
// returns the K largest keys with their scores and ids in (key desc, id asc) order; the host merges the rows' lists by key (= the 2 num_beams draws).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void beam_gumbel_topk_kernel(const T* __restrict__ logits, int ld, int V, const uint8_t* __restrict__ keep, const float* __restrict__ beam_scores,
                                                                float inv_t, uint32_t seed_lo, uint32_t seed_hi, uint32_t counter0, int K,
                                                                float* __restrict__ out_keys, float* __restrict__ out_scores, int* __restrict__ out_ids) {
    __shared__ float redf[16];
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* x = logits + (size_t)row * ld;
    const uint8_t* kp = keep ? keep + (size_t)row * V : nullptr;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, to_f32(x[i]));
    mx = block_max_1024(mx, redf);
    float sum = 0.f;
    for (int i = tid; i < V; i += 1024) sum += expf(to_f32(x[i]) - mx);
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += redf[w];
    const float lse = mx + logf(tot);
    const float base = beam_scores ? beam_scores[row] : 0.f;
    auto score_of = [&](int i) { return ((to_f32(x[i]) - lse) + base) * inv_t; };
    auto key_of = [&](int i) {
        const uint32_t r = philox_u32(seed_lo, seed_hi, counter0 + (uint32_t)row * (uint32_t)V + (uint32_t)i);
        const float u = ((float)(r >> 8) + 0.5f) * (1.f / 16777216.f);            // (0, 1): 24 bits, never 0 or 1
        return score_of(i) - logf(-logf(u));
    };
    float pv = INFINITY; int pi = -1;
    for (int k = 0; k < K; ++k) {
        float best = -INFINITY; int besti = 0x7fffffff;
        for (int i = tid; i < V; i += 1024) {
            if (kp && !kp[i]) continue;
            const float v = key_of(i);
            const bool after = v < pv || (v == pv && i > pi);
            if (after && (v > best || (v == best && i < besti))) { best = v; besti = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        __syncthreads();
        if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
        __syncthreads();
        best = bv[0]; besti = bi[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        if (tid == 0) {
            const bool ok = besti != 0x7fffffff;
            out_keys[(size_t)row * K + k] = ok ? best : -INFINITY;
            out_scores[(size_t)row * K + k] = ok ? score_of(besti) : -INFINITY;
            out_ids[(size_t)row * K + k] = ok ? besti : -1;
        }
        pv = best; pi = besti;
    }
}

void launch_beam_gumbel_topk(int dtype, const void* logits, int ld, int V, int rows, const uint8_t* keep, const float* beam_scores, float temperature, uint64_t seed,
                             uint32_t counter0, int K, float* out_keys, float* out_scores, int* out_ids, hipStream_t st) {
    LMX_REQUIRE(rows >= 1 && K >= 1 && K <= 64 && V >= 1 && temperature > 0.f, "beam_gumbel_topk: bad arguments");
    const float inv_t = 1.f / fmaxf(temperature, 1e-5f);
#define L(TT) hipLaunchKernelGGL(beam_gumbel_topk_kernel<TT>, dim3(rows), dim3(1024), 0, st, (const TT*)logits, ld, V, keep, beam_scores, inv_t, (uint32_t)seed, \
                                 (uint32_t)(seed >> 32), counter0, K, out_keys, out_scores, out_ids)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

void launch_beam_topk(int dtype, const void* logits, int ld, int V, int rows, const float* beam_scores, int K, float* out_scores, int* out_ids, hipStream_t st) {
    LMX_REQUIRE(rows >= 1 && K >= 1 && K <= 64 && V >= 1, "beam_topk: bad arguments");
#define L(TT) hipLaunchKernelGGL(beam_topk_kernel<TT>, dim3(rows), dim3(1024), 0, st, (const TT*)logits, ld, V, beam_scores, K, out_scores, out_ids)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

}  // namespace lmx
