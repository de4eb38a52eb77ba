// MFMA / LDS-layout helpers shared by the flash forward kernels (attention.hip) and the flash backward kernels (attn_bwd.hip).
#pragma once
#include "common.h"

namespace lmx {

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32<f16_t> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_v, a), __builtin_bit_cast(f16x8_v, b), c, 0, 0, 0);
    }
};

constexpr int FA_QB = 128;     // query rows per workgroup (4 waves x 32)
constexpr int FA_KT = 64;      // keys per tile
constexpr float FA_DEFER = 8.f; // log2 units: skip the O rescale until a row max grows by more than 2^8

// K tile in LDS: [64 keys][D] 16-bit; 16-byte chunks XOR-swizzled so a ds_read_b128 lane group (16 distinct rows,
// same logical chunk) touches 16 distinct 16-byte slots of the 256-byte bank row.
template <int D> __device__ __forceinline__ int k_lds_off(int row, int chunk) {
    if constexpr (D == 128) return row * 256 + (((chunk ^ row) & 15) << 4);
    else return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4);
}
// Vᵀ tile in LDS: [D rows][64 keys] = 128 bytes per row = 8 chunks of 16 bytes (8 keys), chunk c of row r stored at c ^ ((r >> 1) & 7): whole
// 16-byte chunks, so the image can be written by LDS-DMA.
// whole 16-byte chunk (8 keys) of a Vᵀ row: what a P·V fragment reads once the keys of a 32-key block are fed to QKᵀ in the order fa_key_perm — the 16
// lanes of every ds_read_b128 service group then hit 16 distinct 16-byte slots (row & 1 picks the half of the 64 banks, (chunk ^ row >> 1) & 7 the slot)
__device__ __forceinline__ int vt_lds_chunk(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }
// Key order inside a 32-key block.  The 32x32 accumulator of Sᵀ = K Qᵀ holds tile row (r & 3) + 8 (r >> 2) + 4 hi in register r of lane half hi, and the
// P·V step (kb, s2) takes registers 8 s2 .. + 7 as its 8 k-values: with K fed in natural order those are keys {4 hi .. + 3} and {8 + 4 hi .. + 3} (+ 16 s2),
// i.e. the Vᵀ operand needs TWO 8-byte LDS reads, 2-way bank-conflicted (1.46 M conflict cycles per launch in round 2's PMC).  Feeding K row pi(l31) =
// l31 with bits 2 and 3 swapped as tile row l31 makes them the 8 CONSECUTIVE keys 16 s2 + 8 hi .. + 7: one conflict-free 16-byte read.  (pi maps each
// 16-lane service group of ds_read_b128 onto itself, so the K fragment reads stay conflict-free.)
__device__ __forceinline__ int fa_key_perm(int l31) { return (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1); }

}  // namespace lmx
