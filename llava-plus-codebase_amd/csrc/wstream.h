// Hand-counted weight stream for the batch-1 decode linears (gemv2.h: gemv2_body, carried by gemv2_kernel in gemm.hip and decode_kv_attn_kernel in decode_attn.hip)
// and the decode batch's linear (skinny.hip).
//
// Why: hipcc's own s_waitcnt placement never pipelines these loops.  Every wait it emits in gemv_kernel is vmcnt(0) (checked in the ISA: 17 x vmcnt(0),
// no counted wait in the main loop), because the loads sit under lane / trip-count conditions and — in the multi-slot stream of the flow kernel — share
// the vmcnt queue with stores, where loads and stores may retire out of order and the compiler falls back to a full drain.  A wave therefore alternates
// between "P rounds in flight" and "nothing in flight", and only the sheer number of waves keeps the HBM busy.
//
// Here the weight loads are inline asm (buffer_load_dwordx4 ... offen nt: one raw buffer view of W, the lane's chunk offset in ONE VGPR shared by the R
// rows, each row's byte offset wave-uniform in the scalar-offset operand) and the waits are counted by hand: R x P loads are ALWAYS on the wire, a round
// is consumed after `s_waitcnt vmcnt(R (P - 1))`, and its buffer is refilled at once.  Rules (guide §5.7):
//   * hipcc does not know these loads: the destination registers count as written at the asm statement.  Every consumer therefore sits behind a wait
//     statement that names the round's registers "+v" (nothing can be scheduled above it), and the stream ends with a drain that names ALL buffers, so
//     that no register is handed to other code while a load is still landing in it.
//   * Between the first issue and the drain the wave must not execute stores or other loads whose completion order matters: vmcnt counts loads and
//     stores in one queue and only loads retire in order.  (LDS traffic is on lgkmcnt and is free to interleave.)  Loads issued by hipcc BEFORE or
//     BETWEEN are safe in the other direction: its own counted waits only ever become more conservative.
//   * The count must stay exact up to the last round, so rounds past the end of the stream are issued as dummy loads of one hot line (offset 0 of W).
#pragma once
#include "common.h"

namespace lmx {

typedef uint32_t ws_u32x4 __attribute__((ext_vector_type(4)));
typedef int ws_v4i __attribute__((ext_vector_type(4)));

// buffer resource for raw (stride 0) 32-bit-offset addressing: base, num_records (bytes), gfx9-family dword 3
__device__ __forceinline__ ws_v4i ws_make_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    ws_v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

template <typename T> __device__ __forceinline__ void ws_unpack8(const ws_u32x4 v, float (&f)[8]) {
    f[0] = unpack_lo<T>(v.x); f[1] = unpack_hi<T>(v.x); f[2] = unpack_lo<T>(v.y); f[3] = unpack_hi<T>(v.y);
    f[4] = unpack_lo<T>(v.z); f[5] = unpack_hi<T>(v.z); f[6] = unpack_lo<T>(v.w); f[7] = unpack_hi<T>(v.w);
}

// one 16-byte non-temporal load: dst <- W[voff + soff]  (voff: this lane's byte offset inside the row, soff: the row's byte offset, wave-uniform)
__device__ __forceinline__ void ws_load(ws_u32x4& dst, uint32_t voff, const ws_v4i& rs, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// same, default cache policy (KV cache rows, rope table) / agent scope (sc1: an activation row another XCD wrote in this launch)
__device__ __forceinline__ void ws_load_plain(ws_u32x4& dst, uint32_t voff, const ws_v4i& rs, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void ws_load_sc1(ws_u32x4& dst, uint32_t voff, const ws_v4i& rs, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// one 16-bit element (zero-extended) through the same path: the residual value a wave adds at the very end is requested as the FIRST load of its queue.
// (As a plain C++ load under `if (residual)` hipcc puts a vmcnt(0) right behind it — a dependent round trip in front of the whole weight stream.)
__device__ __forceinline__ void ws_load_u16(uint32_t& dst, uint32_t voff, const ws_v4i& rs, uint32_t soff) {
    asm volatile("buffer_load_ushort %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// after the stream's drain: the value passes through, so no use is scheduled above the drain
__device__ __forceinline__ void ws_landed(uint32_t& v) { asm volatile("" : "+v"(v)); }

// wait until at most N of this wave's vector-memory operations are outstanding; the R registers of the round about to be consumed pass through the
// statement, so no consumer can be scheduled above it
template <int N, int R> __device__ __forceinline__ void ws_wait(ws_u32x4 (&b)[R]) {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    static_assert(R == 1 || R == 2 || R == 4 || R == 8, "registers per statement");
    if constexpr (R == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b[0]) : "n"(N) : "memory");
    else if constexpr (R == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
    else if constexpr (R == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "n"(N) : "memory");
}

// same for one register (series of statements when more than 8 registers have to pass: an asm statement takes at most 30 operands)
template <int N> __device__ __forceinline__ void ws_wait1(ws_u32x4& b) {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b) : "n"(N) : "memory");
}

// end of the stream: every load has landed; all buffers pass through, so none of them was free for other code while a load was in flight
template <int P, int R> __device__ __forceinline__ void ws_drain(ws_u32x4 (&b)[P][R]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(b[p][r]));
}

}  // namespace lmx
