// Pieces shared by the prefill GEMM kernels (gemm.hip, gemm8p.hip): MFMA wrappers, the LDS tile image / swizzle, and the
// register epilogue (bias / activation / SiLU·mul / residual, 8-byte packed stores).
#pragma once
#include "common.h"
#include "kernels.h"

namespace lmx {

// ---------------------------------------------------------------------------------------------
// MFMA wrappers
// ---------------------------------------------------------------------------------------------
template <typename T> struct Mfma32x32x16;
template <> struct Mfma32x32x16<bf16_t> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32x32x16<f16_t> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a), __builtin_bit_cast(f16x8_v, b), c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------------------
// shared epilogue math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == kActQuickGelu) return act_quick_gelu(v);
    if (act == kActGeluErf) return act_gelu_erf(v);
    return v;
}

template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    uint2 u; u.x = pack2<bf16_t>(v[0], v[1]); u.y = pack2<bf16_t>(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float (&v)[4]) {
    uint2 u; u.x = pack2<f16_t>(v[0], v[1]); u.y = pack2<f16_t>(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = unpack_lo<bf16_t>(u.x); v[1] = unpack_hi<bf16_t>(u.x); v[2] = unpack_lo<bf16_t>(u.y); v[3] = unpack_hi<bf16_t>(u.y);
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float (&v)[4]) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = unpack_lo<f16_t>(u.x); v[1] = unpack_hi<f16_t>(u.x); v[2] = unpack_lo<f16_t>(u.y); v[3] = unpack_hi<f16_t>(u.y);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    float4 u = *reinterpret_cast<const float4*>(p);
    v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}

// ---------------------------------------------------------------------------------------------
// 16-bit MFMA GEMM
// ---------------------------------------------------------------------------------------------
constexpr int GEMM_BK = 64;               // K elements per stage = 128 bytes per row
constexpr int GEMM_ROWB = GEMM_BK * 2;    // row pitch in LDS (bytes)

// LDS byte offset of 16-byte chunk `chunk` (0..7) of tile row `row`: rows 2r,2r+1 share a 256-B bank row, and
// the 8 chunk slots are XORed with (row>>1)&7 so 16 consecutive rows reading the same logical chunk hit 16
// distinct 16-B slots (ds_read_b128 lane groups are 16 lanes wide).
__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
    return row * GEMM_ROWB + (((chunk ^ (row >> 1)) & 7) << 4);
}

// ---------------------------------------------------------------------------------------------
// shared epilogue: acc[i][j][4q+e] = C[m = m_base + j*32 + l31][n = n_base + i*32 + 8q + 4hi + e]
// (weight fragment = MFMA A operand, so a lane owns 4 consecutive n of one row m per accumulator quad)
// ---------------------------------------------------------------------------------------------
template <typename T, int MT, int NTL>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[NTL][MT], int m_base, int n_base, int l31, int hi) {
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const T* __restrict__ bias = reinterpret_cast<const T*>(a.bias);
    const T* R = reinterpret_cast<const T*>(a.R);
    const int act = a.act;

    if (act == kActSiluMul) {
        // fused rows are interleaved [32 gate | 32 up] per 64: tile i even = gate, i+1 = up, same lane/register
        if constexpr (NTL % 2 == 0) {
#pragma unroll
        for (int i = 0; i < NTL; i += 2) {
            const int nfused = n_base + i * 32;            // multiple of 64
            const int nout0 = nfused / 2;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int m = m_base + j * 32 + l31;
                if (m >= a.M) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int no = nout0 + 8 * q + 4 * hi;
                    if (nfused + 8 * q + 4 * hi >= a.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float g = acc[i][j][4 * q + e], u = acc[i + 1][j][4 * q + e];
                        if (bias) {
                            g += to_f32(bias[nfused + 8 * q + 4 * hi + e]);
                            u += to_f32(bias[nfused + 32 + 8 * q + 4 * hi + e]);
                        }
                        v[e] = act_silu(g) * u;
                    }
                    store4<T>(C + (size_t)m * a.ldc + no, v);
                }
            }
        }
        }
        return;
    }

    const int pk_q = a.pk_kc ? a.pk_heads * a.pk_D : 0x7fffffff;      // first column that leaves through the K / V^T pack (CLIP q|k|v)
    // Small tiles (the CLIP-sized launches: <= 4 accumulator blocks per wave) request every bias and residual value BEFORE the first store (round 4): C may
    // alias R and nothing tells the compiler that it does not alias the bias, so in the one-quad-at-a-time loop every quad's loads waited behind the previous
    // quad's store — a chain of dependent L2 round trips at the tail of launches that last 10 - 20 us in all.
    constexpr bool PRELOAD = MT * NTL <= 4;
    uint2 bq[PRELOAD ? NTL : 1][4], rq[PRELOAD ? NTL : 1][PRELOAD ? MT : 1][4];
    if constexpr (PRELOAD) {
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n_base + i * 32 + 8 * q + 4 * hi;
                bq[i][q] = (bias && n < a.N) ? *reinterpret_cast<const uint2*>(bias + n) : uint2{0u, 0u};
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int m = m_base + j * 32 + l31;
                    rq[i][j][q] = (R && m < a.M && n < a.N && n < pk_q) ? *reinterpret_cast<const uint2*>(R + (size_t)m * a.ldr + n) : uint2{0u, 0u};
                }
            }
    }
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m_base + j * 32 + l31;
            if (m >= a.M) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n_base + i * 32 + 8 * q + 4 * hi;
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (bias) {
                    float b[4];
                    if constexpr (PRELOAD) { b[0] = unpack_lo<T>(bq[i][q].x); b[1] = unpack_hi<T>(bq[i][q].x); b[2] = unpack_lo<T>(bq[i][q].y); b[3] = unpack_hi<T>(bq[i][q].y); }
                    else load4<T>(bias + n, b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += b[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
                if (n >= pk_q) {
                    // k column -> K rows [head][pos][D] (8-byte store); v column -> V^T [head][d][pos] (the wave's 32 rows make 64-byte runs per register)
                    const int img = m / a.pk_rows, pos = m - img * a.pk_rows;
                    int c = n - pk_q;
                    const bool isv = c >= pk_q;
                    if (isv) c -= pk_q;
                    const int head = c / a.pk_D, d = c - head * a.pk_D;
                    if (!isv) store4<T>(reinterpret_cast<T*>(a.pk_kc) + img * a.pk_img_stride + ((size_t)head * a.pk_spad + pos) * a.pk_D + d, v);
                    else {
                        T* vt = reinterpret_cast<T*>(a.pk_vt) + img * a.pk_img_stride + ((size_t)head * a.pk_D + d) * a.pk_spad + pos;
#pragma unroll
                        for (int e = 0; e < 4; ++e) vt[(size_t)e * a.pk_spad] = from_f32<T>(v[e]);
                    }
                    continue;
                }
                if (R) {
                    float r[4];
                    if constexpr (PRELOAD) { r[0] = unpack_lo<T>(rq[i][j][q].x); r[1] = unpack_hi<T>(rq[i][j][q].x); r[2] = unpack_lo<T>(rq[i][j][q].y); r[3] = unpack_hi<T>(rq[i][j][q].y); }
                    else load4<T>(R + (size_t)m * a.ldr + n, r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                }
                store4<T>(C + (size_t)m * a.ldc + n, v);
            }
        }
    }
}


}  // namespace lmx
