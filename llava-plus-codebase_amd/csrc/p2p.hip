// One-shot peer-to-peer all-reduce(sum) for the decode step's [rows, H] partial sums (tensor parallel over xGMI).
//
// A decode step has 2 all-reduces per layer of 8 KiB per sequence: pure latency.  A ring collective pays a launch plus
// 2(n-1) hops; here every rank WRITES its rows straight into a slot of every peer's exchange buffer (mapped through HIP IPC,
// xGMI is point-to-point so all peers are one hop away), raises a flag, and every rank then sums the n slots it RECEIVED
// in rank order — one launch, one hop, bit-identical on every rank, deterministic.
//
//   exchange buffer of a rank (uncached device memory, one allocation, mapped by all peers):
//     data  [2 parities][world][P2P_MAX_ROWS][H]   model dtype
//     flags [2 parities][world][P2P_MAX_ROWS]      uint32: sequence number of the all-reduce whose row this slot holds
//     status                                       uint32: set to the failing sequence number if a wait timed out
//   all-reduce k uses parity k & 1.  Slot reuse is safe without a second handshake: a rank can only start k+2 after it saw
//   every peer's flag for k+1, and a peer raises its k+1 flags after it finished reading all of k (same stream order).
//   One workgroup per row: push the row to all peers (16-byte stores) -> __threadfence_system -> flag stores (system scope)
//   -> spin on the local flags (bounded: 30 s of the 100 MHz realtime counter, then status := k) -> sum slots in rank order.
#include "common.h"
#include "kernels.h"

namespace lmx {

struct P2PArgs {
    void* buf;                    // [rows][H] partial sums in, total out (in place)
    int H, world, rank, rows;
    int last_len;                 // elements of the LAST row (a message need not be a whole number of rows); multiple of 8
    uint32_t seq;
    size_t data_stride_rank;      // bytes between two ranks' slot blocks inside one parity  (= P2P_MAX_ROWS * H * es)
    size_t data_stride_parity;    // bytes between the two parities                           (= world * data_stride_rank)
    size_t flags_off, status_off; // byte offsets inside an exchange buffer
    char* peer[P2P_MAX_WORLD];    // exchange buffer of every rank as mapped in THIS process (peer[rank] = own)
    const void* norm_w; void* x_out; float eps;      // x_out != null: RMSNorm of the summed row as well (P2PLaunch)
};

template <typename T>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs a) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const int par = (int)(a.seq & 1u);
    constexpr int VE = 16 / sizeof(T);
    const int HC = (row == a.rows - 1 ? a.last_len : a.H) / VE;
    T* mine = reinterpret_cast<T*>(a.buf) + (size_t)row * a.H;
    const size_t slot_off = (size_t)par * a.data_stride_parity + (size_t)a.rank * a.data_stride_rank + (size_t)row * a.H * sizeof(T);

    // 1. push my row into slot [par][rank][row] of every rank (own included: the sum below reads all slots uniformly)
    for (int c = tid; c < HC; c += 256) {
        const uint4 v = reinterpret_cast<const uint4*>(mine)[c];
        for (int p = 0; p < a.world; ++p) reinterpret_cast<uint4*>(a.peer[p] + slot_off)[c] = v;
    }
    __threadfence_system();
    __syncthreads();
    // 2. raise my flag for this row on every rank
    if (tid < a.world) {
        uint32_t* f = reinterpret_cast<uint32_t*>(a.peer[tid] + a.flags_off) + ((size_t)par * a.world + a.rank) * P2P_MAX_ROWS + row;
        __hip_atomic_store(f, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 3. wait for every rank's flag on MY buffer (bounded spin)
    if (tid < a.world) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(a.peer[a.rank] + a.flags_off) + ((size_t)par * a.world + tid) * P2P_MAX_ROWS + row;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 3000000000ull) {         // 30 s at 100 MHz: a bound against hangs, far above rank skew
                __hip_atomic_store(reinterpret_cast<uint32_t*>(a.peer[a.rank] + a.status_off), a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __threadfence_system();
    // 4. sum the received slots in rank order (fp32), identical on every rank
    const char* base = a.peer[a.rank] + (size_t)par * a.data_stride_parity + (size_t)row * a.H * sizeof(T);
    for (int c = tid; c < HC; c += 256) {
        float acc[VE > 8 ? VE : 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int p = 0; p < a.world; ++p) {
            const T* src = reinterpret_cast<const T*>(base + (size_t)p * a.data_stride_rank) + c * VE;
            if constexpr (sizeof(T) == 2) {
                // uncached memory: read through a system-scope 16-byte load
                float v[8]; load8<T>(src, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            } else {
                const float4 v = *reinterpret_cast<const float4*>(src);
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            }
        }
        if constexpr (sizeof(T) == 2) store8<T>(mine + c * VE, reinterpret_cast<const float (&)[8]>(acc));
        else *reinterpret_cast<float4*>(mine + c * VE) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    // 5. (x_out) LlamaRMSNorm of the finished row — rmsnorm_kernel's loops on the row this workgroup has just written (the stored, i.e. rounded, values)
    if (a.x_out) {
        __shared__ float red[4];
        __syncthreads();                                   // the row is complete (every thread's stores are visible to the workgroup)
        const T* w = reinterpret_cast<const T*>(a.norm_w);
        T* yr = reinterpret_cast<T*>(a.x_out) + (size_t)row * a.H;
        const int HC8 = a.H >> 3;
        float ss = 0.f;
        for (int c = tid; c < HC8; c += 256) {
            float v[8]; load8<T>(mine + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
        }
        ss = block_sum<4>(ss, red);
        const float inv = rsqrtf(ss / (float)a.H + a.eps);
        for (int c = tid; c < HC8; c += 256) {
            float v[8], g[8]; load8<T>(mine + c * 8, v); load8<T>(w + c * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_to<T>(v[e] * inv) * g[e];
            store8<T>(yr + c * 8, v);
        }
    }
}

void launch_p2p_allreduce(int dtype, const P2PLaunch& l, hipStream_t st) {
    LMX_REQUIRE(l.rows >= 1 && l.rows <= P2P_MAX_ROWS && l.world >= 1 && l.world <= P2P_MAX_WORLD, "p2p all-reduce: bad geometry");
    LMX_REQUIRE(l.H % 8 == 0, "p2p all-reduce: H must be a multiple of 8");
    P2PArgs a{};
    a.buf = l.buf; a.H = l.H; a.world = l.world; a.rank = l.rank; a.rows = l.rows; a.seq = l.seq;
    a.last_len = l.last_len > 0 ? l.last_len : l.H;
    a.norm_w = l.norm_w; a.x_out = l.x_out; a.eps = l.eps;
    LMX_REQUIRE(!l.x_out || (l.norm_w && a.last_len == l.H), "p2p all-reduce: the fused RMSNorm needs its weight and whole rows");
    LMX_REQUIRE(a.last_len % 8 == 0 && a.last_len <= l.H, "p2p all-reduce: the last row must be a multiple of 8 elements");
    const size_t es = dtype_size(dtype);
    a.data_stride_rank = (size_t)P2P_MAX_ROWS * l.H * es;
    a.data_stride_parity = (size_t)l.world * a.data_stride_rank;
    a.flags_off = p2p_flags_offset(l.world, l.H, (int)es);
    a.status_off = a.flags_off + (size_t)2 * l.world * P2P_MAX_ROWS * 4;
    for (int p = 0; p < l.world; ++p) a.peer[p] = static_cast<char*>(l.peer[p]);
#define L(TT) hipLaunchKernelGGL(p2p_allreduce_kernel<TT>, dim3(l.rows), dim3(256), 0, st, a)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two-shot all-reduce (reduce-scatter + all-gather) for prefill-sized messages — see kernels.h.
//   exchange region of a rank:  rs   [2 parities][world sources][chunk_max]   partial sums of MY chunk as pushed by every source rank
//                               ag   [2 parities][world chunks][chunk_max]    finished chunks as pushed by their owners
//                               flags[2 parities][2 phases][world][n_slices_max]  uint32 sequence numbers
// Workgroup g of every rank works on slices g, g + G, ... of every chunk (G workgroups per launch, the same on every rank); "my slices" below:
//   1. push my slices of chunk p of my partial sums into rank p's rs[me] (every p != me), system fence, raise flag (phase 0, me, g) on rank p
//   2. wait for flag (phase 0, q, g) of every q != me on my buffer
//   3. sum my slices of MY chunk over the ranks in rank order (own partial from `buf`, the others from rs[q]), round once, store them into `buf` and push them
//      into ag[me] of every peer, system fence, raise flag (phase 1, me, g) on every peer
//   4. wait for flag (phase 1, q, g) of every q != me, copy my slices of chunk q from ag[q] into `buf`
// Slot reuse across all-reduces: parity = seq & 1, same argument as the one-shot kernel (a rank starts k + 2 only after it saw every peer's flags of
// k + 1, which a peer raises only after its kernel k has finished).
// ---------------------------------------------------------------------------------------------------------------------------
struct P2PBigArgs {
    void* buf; size_t count; size_t chunk;      // chunk: elements per rank chunk of THIS message (multiple of P2P_BIG_SLICE)
    int world, rank; uint32_t seq;
    size_t rs_off, ag_off, flags_off, chunk_max; int n_slices_max;    // parity already applied to the offsets
    size_t status_off;
    char* peer[P2P_MAX_WORLD];
};

template <typename T>
__device__ __forceinline__ void p2p_wait_flags(const P2PBigArgs& a, int phase, int s, int tid) {
    if (tid < a.world && tid != a.rank) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(a.peer[a.rank] + a.flags_off) + ((size_t)phase * a.world + tid) * a.n_slices_max + s;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 3000000000ull) {         // 30 s at 100 MHz
                __hip_atomic_store(reinterpret_cast<uint32_t*>(a.peer[a.rank] + a.status_off), a.seq | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __threadfence_system();
}
template <typename T>
__device__ __forceinline__ void p2p_raise_flags(const P2PBigArgs& a, int phase, int s, int tid) {
    __threadfence_system();
    __syncthreads();
    if (tid < a.world && tid != a.rank) {
        uint32_t* f = reinterpret_cast<uint32_t*>(a.peer[tid] + a.flags_off) + ((size_t)phase * a.world + a.rank) * a.n_slices_max + s;
        __hip_atomic_store(f, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void p2p_allreduce_big_kernel(P2PBigArgs a) {
    constexpr int VE = 16 / sizeof(T);
    // workgroup g walks slices g, g + G, ... of every chunk; its flags are indexed by g: ONE release / acquire pair per phase and workgroup, not per slice —
    // a system-scope fence walks the L2, and the 544 x 4 of them of a 1087 x 4096 message cost 346 us between two processes on one MI355X where the data
    // itself moves in ~10 us (profiles/EXPERIMENTS.md r4-K).  Every rank derives the same G from the message size, so flag g of a peer covers the same slices.
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const int nv = P2P_BIG_SLICE / VE;                                    // 16-byte vectors per slice
    const int n_slices = (int)(a.chunk / P2P_BIG_SLICE);
    T* buf = reinterpret_cast<T*>(a.buf);
    // 1. scatter: my slices of chunk p -> rank p
    for (int s = g; s < n_slices; s += G) {
        const size_t in_chunk = (size_t)s * P2P_BIG_SLICE;                // first element of the slice inside a chunk
        for (int p = 0; p < a.world; ++p) {
            if (p == a.rank) continue;
            const size_t g0 = (size_t)p * a.chunk + in_chunk;
            uint4* dst = reinterpret_cast<uint4*>(a.peer[p] + a.rs_off + ((size_t)a.rank * a.chunk_max + in_chunk) * sizeof(T));
            for (int c = tid; c < nv; c += 256)
                if (g0 + (size_t)c * VE < a.count) dst[c] = reinterpret_cast<const uint4*>(buf + g0)[c];
        }
    }
    p2p_raise_flags<T>(a, 0, g, tid);
    p2p_wait_flags<T>(a, 0, g, tid);
    // 3. reduce my slices of my chunk in rank order, keep them and push them to every peer
    for (int s = g; s < n_slices; s += G) {
        const size_t in_chunk = (size_t)s * P2P_BIG_SLICE;
        const size_t g0 = (size_t)a.rank * a.chunk + in_chunk;
        for (int c = tid; c < nv; c += 256) {
            if (g0 + (size_t)c * VE >= a.count) continue;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int q = 0; q < a.world; ++q) {
                const T* src = q == a.rank ? buf + g0 + (size_t)c * VE
                                           : reinterpret_cast<const T*>(a.peer[a.rank] + a.rs_off + ((size_t)q * a.chunk_max + in_chunk) * sizeof(T)) + (size_t)c * VE;
                if constexpr (sizeof(T) == 2) {
                    float v[8]; load8<T>(src, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += v[e];
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                }
            }
            uint4 out;
            if constexpr (sizeof(T) == 2) {
                store8<T>(buf + g0 + (size_t)c * VE, reinterpret_cast<const float (&)[8]>(acc));
                out = *reinterpret_cast<const uint4*>(buf + g0 + (size_t)c * VE);          // the rounded bits, as every rank will hold them
            } else {
                out = uint4{__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
                *reinterpret_cast<uint4*>(buf + g0 + (size_t)c * VE) = out;
            }
            for (int p = 0; p < a.world; ++p)
                if (p != a.rank) reinterpret_cast<uint4*>(a.peer[p] + a.ag_off + ((size_t)a.rank * a.chunk_max + in_chunk) * sizeof(T))[c] = out;
        }
    }
    p2p_raise_flags<T>(a, 1, g, tid);
    p2p_wait_flags<T>(a, 1, g, tid);
    // 4. gather: my slices of every other chunk from my ag region
    for (int s = g; s < n_slices; s += G) {
        const size_t in_chunk = (size_t)s * P2P_BIG_SLICE;
        for (int q = 0; q < a.world; ++q) {
            if (q == a.rank) continue;
            const size_t g0 = (size_t)q * a.chunk + in_chunk;
            const uint4* src = reinterpret_cast<const uint4*>(a.peer[a.rank] + a.ag_off + ((size_t)q * a.chunk_max + in_chunk) * sizeof(T));
            for (int c = tid; c < nv; c += 256)
                if (g0 + (size_t)c * VE < a.count) reinterpret_cast<uint4*>(buf + g0)[c] = src[c];
        }
    }
}

P2PBigGeom p2p_big_geometry(int world, size_t max_count, int es, size_t base_off) {
    P2PBigGeom g{};
    const size_t per = (max_count + world - 1) / world;
    g.chunk_max = (per + P2P_BIG_SLICE - 1) / P2P_BIG_SLICE * P2P_BIG_SLICE;
    g.n_slices_max = (int)(g.chunk_max / P2P_BIG_SLICE);
    const size_t data = (size_t)world * g.chunk_max * es;                 // one parity of rs (or of ag)
    g.rs_off = (base_off + 255) / 256 * 256; g.rs_par = data;
    g.ag_off = g.rs_off + 2 * data; g.ag_par = data;
    g.flags_par = (size_t)2 * world * g.n_slices_max * 4;
    g.flags_off = g.ag_off + 2 * data;
    g.end = g.flags_off + 2 * g.flags_par + 256;
    return g;
}

void launch_p2p_allreduce_big(int dtype, const P2PBigLaunch& l, hipStream_t st) {
    const size_t es = dtype_size(dtype);
    const int ve = (int)(16 / es);
    LMX_REQUIRE(l.world >= 2 && l.world <= P2P_MAX_WORLD && l.count > 0 && l.count % ve == 0 && (reinterpret_cast<uintptr_t>(l.buf) & 15) == 0,
                "p2p two-shot all-reduce: 2..8 ranks, a 16-byte aligned message of whole 16-byte vectors");
    const size_t per = (l.count + l.world - 1) / l.world;
    const size_t chunk = (per + P2P_BIG_SLICE - 1) / P2P_BIG_SLICE * P2P_BIG_SLICE;
    LMX_REQUIRE(chunk <= l.g.chunk_max, "p2p two-shot all-reduce: message larger than the exchange region");
    P2PBigArgs a{};
    a.buf = l.buf; a.count = l.count; a.chunk = chunk; a.world = l.world; a.rank = l.rank; a.seq = l.seq;
    const int par = (int)(l.seq & 1u);
    a.rs_off = l.g.rs_off + par * l.g.rs_par; a.ag_off = l.g.ag_off + par * l.g.ag_par; a.flags_off = l.g.flags_off + par * l.g.flags_par;
    a.chunk_max = l.g.chunk_max; a.n_slices_max = l.g.n_slices_max; a.status_off = l.status_off;
    for (int p = 0; p < l.world; ++p) a.peer[p] = static_cast<char*>(l.peer[p]);
    // workgroups per launch: every one costs four system-scope fences, so few fat ones (LMX_P2P_BIG_WGS, default 64; the same on every rank)
    constexpr int max_wgs = 64;      // 64 fat workgroups: 349 -> 88 us per 8.9 MB sum (profiles/r04_p2p_big_wgs.txt)
    const int n_slices = (int)(chunk / P2P_BIG_SLICE);
    const int grid = n_slices < max_wgs ? n_slices : max_wgs;
#define L(TT) hipLaunchKernelGGL(p2p_allreduce_big_kernel<TT>, dim3(grid), dim3(256), 0, st, a)
    if (dtype == kBF16) L(bf16_t); else if (dtype == kF16) L(f16_t); else L(float);
#undef L
    LMX_CHECK_HIP(hipGetLastError());
}

size_t p2p_flags_offset(int world, int H, int es) { return ((size_t)2 * world * P2P_MAX_ROWS * H * es + 255) / 256 * 256; }
size_t p2p_buffer_bytes(int world, int H, int es) { return p2p_flags_offset(world, H, es) + (size_t)2 * world * P2P_MAX_ROWS * 4 + 256; }

}  // namespace lmx
