// Common device/host helpers for the MI355X (gfx950 / CDNA4) LLaVA forward path.
// Everything here is written for wave64 + MFMA; there is no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace lmx {

// ---------------------------------------------------------------------------------------------
// dtypes (mirrors include/llava_mi355x.h: LMX_DTYPE_*)
// ---------------------------------------------------------------------------------------------
enum DType : int { kF32 = 0, kBF16 = 1, kF16 = 2 };

static inline size_t dtype_size(int dt) { return dt == kF32 ? 4 : 2; }

struct bf16_t { uint16_t x; };   // storage-only tags; arithmetic is always fp32
struct f16_t  { uint16_t x; };

typedef __attribute__((ext_vector_type(8))) __bf16   bf16x8_v;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_v;
typedef __attribute__((ext_vector_type(16))) float   f32x16;
typedef __attribute__((ext_vector_type(4)))  float   f32x4;

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float>  { static constexpr int id = kF32;  static constexpr int vec16 = 4; };
template <> struct TypeInfo<bf16_t> { static constexpr int id = kBF16; static constexpr int vec16 = 8; };
template <> struct TypeInfo<f16_t>  { static constexpr int id = kF16;  static constexpr int vec16 = 8; };

// ---- scalar conversions -------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// round-to-nearest-even, NaN stays NaN (same rounding torch uses for .to(bfloat16)).  The native __bf16 cast lowers to the
// gfx950 hardware converter (v_cvt_pk_bf16_f32): branch-free, one instruction per pair.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)b);
    return (float)h;
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;   // RNE
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_bits_to_f32(v.x); }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return f16_bits_to_f32(v.x); }

template <typename T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) { bf16_t r; r.x = (uint16_t)f32_to_bf16_bits(f); return r; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float f) { f16_t r; r.x = (uint16_t)f32_to_f16_bits(f); return r; }

// round an fp32 value through T (used where the reference rounds an intermediate to model dtype)
template <typename T> __device__ __forceinline__ float round_to(float f) { return to_f32<T>(from_f32<T>(f)); }

// One RoPE output with HF's rounding chain, T( T(x c) + T(r s) ) (HF5:models/llama/modeling_llama.py:130-160 on tensors of the model dtype: every
// elementwise op rounds).  Contraction is switched OFF here: for fp16 hipcc narrows T(x c) with half operands to a half multiply (exact) and would then be
// free to fuse it with the add into ONE half fma — skipping the rounding of the product — in one kernel and not in another (seen: rope_kv_kernel and the
// q|k|v epilogue of gemm8p.hip disagreed in 25 % of the fp16 elements by one ulp).  With the pragma every kernel computes the same, HF's, bits.
template <typename T> __device__ __forceinline__ float rope_term(float x, float c, float r, float s) {
#pragma clang fp contract(off)
    const float p = round_to<T>(x * c);
    const float q = round_to<T>(r * s);
    return round_to<T>(p + q);
}

// ---- 16-bit pair pack / unpack ------------------------------------------------------------
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_v __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) {
    const bf16x2_v v = {(__bf16)lo, (__bf16)hi};          // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) {
    const f16x2_v v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
template <typename T> __device__ __forceinline__ float unpack_lo(uint32_t w);
template <typename T> __device__ __forceinline__ float unpack_hi(uint32_t w);
template <> __device__ __forceinline__ float unpack_lo<bf16_t>(uint32_t w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float unpack_hi<bf16_t>(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ float unpack_lo<f16_t>(uint32_t w) { return f16_bits_to_f32(w & 0xffffu); }
template <> __device__ __forceinline__ float unpack_hi<f16_t>(uint32_t w) { return f16_bits_to_f32(w >> 16); }

// Load / store 8 consecutive elements of T as fp32 (16-byte access for 16-bit types, 2x16 B for f32).
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = unpack_lo<bf16_t>(u.x); v[1] = unpack_hi<bf16_t>(u.x);
    v[2] = unpack_lo<bf16_t>(u.y); v[3] = unpack_hi<bf16_t>(u.y);
    v[4] = unpack_lo<bf16_t>(u.z); v[5] = unpack_hi<bf16_t>(u.z);
    v[6] = unpack_lo<bf16_t>(u.w); v[7] = unpack_hi<bf16_t>(u.w);
}
template <> __device__ __forceinline__ void load8<f16_t>(const f16_t* p, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = unpack_lo<f16_t>(u.x); v[1] = unpack_hi<f16_t>(u.x);
    v[2] = unpack_lo<f16_t>(u.y); v[3] = unpack_hi<f16_t>(u.y);
    v[4] = unpack_lo<f16_t>(u.z); v[5] = unpack_hi<f16_t>(u.z);
    v[6] = unpack_lo<f16_t>(u.w); v[7] = unpack_hi<f16_t>(u.w);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint4 u;
    u.x = pack2<bf16_t>(v[0], v[1]); u.y = pack2<bf16_t>(v[2], v[3]);
    u.z = pack2<bf16_t>(v[4], v[5]); u.w = pack2<bf16_t>(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float (&v)[8]) {
    uint4 u;
    u.x = pack2<f16_t>(v[0], v[1]); u.y = pack2<f16_t>(v[2], v[3]);
    u.z = pack2<f16_t>(v[4], v[5]); u.w = pack2<f16_t>(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}

// ---- wave64 / block reductions --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum for blocks of NW waves; `red` is NW floats of LDS. All threads get the result.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();            // protect `red` from a previous use
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- activations (epilogue codes; mirrors include/llava_mi355x.h LMX_ACT_*) -------------------
enum Act : int { kActNone = 0, kActQuickGelu = 1, kActGeluErf = 2, kActSiluMul = 3 };

__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu(float x) { return x / (1.f + __expf(-x)); }

// ---- XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8). Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Logical tile id -> (tile_m, tile_n) of a 256 x 256-tile GEMM.  An XCD runs ~32 CONSECUTIVE logical ids at a time (xcd_remap), each XCD behind its own L2.
// Column-major ids (tile_m fastest) make that window 32 M-tiles of ONE N-panel: the W panel is shared by all of them, but every X panel has a single reader, so X
// is re-read from HBM once per N-panel — harmless at M ~ 1k (five M-tiles: X lives in the L2s), 9 - 12 x the algorithmic traffic at M = 32768 (r6-I: 3.1 GB per
// wgrad launch, 5 TB/s of HBM reads under a "compute-bound" kernel).  From 16 M-tiles on the ids walk groups of GM M-tiles, M fastest inside a group, then N:
// the window is GM x 32 / GM tiles, each X panel read by 32 / GM tiles and each W panel by GM, and a group's X panels stay hot across its whole sweep over N
// (the W panels — the weights, 100 MB at most — come back from the memory-side cache).  Same tiles, same arithmetic: only the order in which they run.
__device__ __forceinline__ void gemm_tile_of(int lid, int mtiles, int ntiles, int& tile_m, int& tile_n) {
    constexpr int GM = 4;               // measured on the 16 x 2048 training step: GM = 16 / 8 / 4 / 2 / 1 -> 1446 / 1359 / 1320 - 1333 / 1323 / 1336 ms
    if (mtiles < 16) { tile_n = lid / mtiles; tile_m = lid - tile_n * mtiles; return; }
    const int gsz = GM * ntiles;
    const int grp = lid / gsz, first = grp * GM;
    const int gm = mtiles - first < GM ? mtiles - first : GM;
    const int in = lid - grp * gsz;
    tile_n = in / gm; tile_m = first + (in - tile_n * gm);
}

// ---------------------------------------------------------------------------------------------
// host-side error plumbing: every C-ABI entry returns int; message is thread-local.
// ---------------------------------------------------------------------------------------------
void set_last_error(const std::string& s);
const char* get_last_error();

struct Error { std::string msg; };

#define LMX_CHECK_HIP(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            throw ::lmx::Error{std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +  \
                               __FILE__ + ":" + std::to_string(__LINE__) + ")"};                  \
        }                                                                                          \
    } while (0)

#define LMX_REQUIRE(cond, what)                                                                    \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            throw ::lmx::Error{std::string("requirement failed: ") + #cond + " — " + (what) +     \
                               " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"};          \
        }                                                                                          \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace lmx
