// Weight-gradient GEMM for gfx950 with BOTH operands read in their forward layout:  C[M,N] = sum_r X[r][m] * W[r][n]
// (X = dY [rows][M], W = the layer's input [rows][N], C = dW [M = out features][N = in features]; 16-bit operands, fp32 accumulate).
//
// Replaces the weight half of torch.nn.Linear's backward (grad_weight = grad_output^T @ input) under the reference's training step
// (llava/train/train.py:780-1000 -> transformers Trainer -> torch autograd).  Through round 5 the step transposed both operands
// ([rows][M] -> [M][rows], [rows][N] -> [N][rows]: two activation-sized read + write passes per linear) to feed the forward's NT kernel (gemm8p.hip); the
// transposes were 6.3 % of a 16 x 2048 step (profiles/r06_config5_kernel_stats.csv).  Here the contraction index is the SLOW index of both operands in
// memory and the MFMA fragments (8 consecutive contraction values per lane) come out of LDS through the gfx950 transpose read, ds_read_b64_tr_b16.
//
// Same schedule as gemm8p.hip — one 512-thread workgroup per 256 x 256 tile, 8 waves = 2 (M) x 4 (N) in two groups one barrier apart, a ring of 8 half-tiles
// (16 KiB each) filled by LDS-DMA six phases ahead, one counted `s_waitcnt vmcnt(4)` per K-step, s_setprio around the MFMA segments — so the comments there
// on hazards and counting apply line by line.  What differs:
//   * a half-tile is 64 r (contraction) x 128 columns of the operand.  X half h = tile rows h * 128 .. + 127 (NOT the two 64-row blocks of gemm8p.hip:
//     with the contraction index slow, a half-tile row of memory is 128 contiguous elements = one 256-byte run per r; splitting it per wave row would
//     halve the runs), W half h = tile columns h * 128 .. + 127.  Wave (wm, wn) multiplies X-half rows wm * 64 .. + 63 with W-half columns wn * 32 .. + 31,
//     so its accumulator block acc[i][j] is C[m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32 ..][n0 + i * 128 + wn * 32 ..].
//   * LDS image of a half-tile: [r / 4][column / 16][4 r][16 columns] — 128-byte blocks, 1 KiB per group of 4 r.  One LDS-DMA instruction fills one group
//     (lane l: block l >> 3, r = (l & 7) >> 1, 8 columns (l & 1) * 8: four 256-byte runs of memory per instruction, lane-linear in LDS).
//   * a fragment (32 columns x 8 r per lane half) is two ds_read_b64_tr_b16: the 16 lanes of a group read one [4 r][16 columns] block (lane i: r = i >> 2,
//     columns 4 (i & 3) .. + 3) and receive column i's four r values (tools/probes/tr_b16_probe.hip prints the mapping).  Lanes 0-31 of a read cover two
//     adjacent blocks = 256 contiguous bytes: conflict-free at the rate of a plain ds_read_b64 (same probe: 3.76 cycles per wave-instruction either way; a
//     row-major [r][256 columns] image would be 4-way conflicted, 8.0).
// Shapes: M % 256 == 0, N % 256 == 0, rows % 64 == 0, row strides multiples of 8 elements (the launcher checks; the training step falls back to the transposes
// otherwise).  Results are bit-identical to transposes + gemm8p: same MFMA, same values in the same contraction slots, same K-step order.
#include <mutex>

#include "common.h"
#include "kernels.h"
#include "gemm_common.h"

namespace lmx {

namespace {

constexpr int T8_HALF = 64 * 128 * 2;      // bytes per half-tile buffer
constexpr int T8_LDS = 8 * T8_HALF;        // ring of 8 half-tiles

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;

template <int N> struct IC { static constexpr int value = N; };

__device__ __forceinline__ v4i_t make_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

}  // namespace

template <typename T>
__global__ __launch_bounds__(512) void gemm8t_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const bool group1 = wave >= 4;

    const int mtiles = a.M >> 8;
    const int ntiles = a.N >> 8;
    const int lid = xcd_remap(blockIdx.x, mtiles * ntiles);
    int tile_m, tile_n;
    gemm_tile_of(lid, mtiles, ntiles, tile_m, tile_n);
    const int m0 = tile_m << 8, n0 = tile_n << 8;
    const int nk = a.K >> 6;                          // K-steps of 64 contraction rows

    // ---- LDS-DMA sources: piece i (0, 1) of a half-tile = the group of 4 r number wave + 8 i --------------------------------------------------
    const v4i_t rsX = make_rsrc(a.X, (uint32_t)((size_t)a.K * a.ldx * sizeof(T)));
    const v4i_t rsW = make_rsrc(a.W, (uint32_t)((size_t)a.K * a.ldw * sizeof(T)));
    uint32_t voX[2][2], voW[2][2];                    // [piece][half] byte offsets of this lane's 16 bytes in K-step 0
    {
        const int col = 16 * (lane >> 3) + 8 * (lane & 1);            // column of the half-tile (0 .. 127, multiples of 8)
        const int rr = (lane & 7) >> 1;                               // r inside the group of 4
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = (wave + 8 * i) * 4 + rr;
                voX[i][h] = (uint32_t)(((size_t)r * a.ldx + m0 + h * 128 + col) * sizeof(T));
                voW[i][h] = (uint32_t)(((size_t)r * a.ldw + n0 + h * 128 + col) * sizeof(T));
            }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    const unsigned kstep_x = (unsigned)a.ldx * 64u * (unsigned)sizeof(T), kstep_w = (unsigned)a.ldw * 64u * (unsigned)sizeof(T);

    // half-tile sequence as gemm8p.hip: index 4 kt + j, j = 0: X half 0, 1: W half 0, 2: W half 1, 3: X half 1; ring slot = (kt & 1) * 4 + j
    auto stage = [&](int kt_abs, int j, int slot) {
        const unsigned d0 = lds_wave + slot * T8_HALF, d1 = d0 + 8192;
        const bool isx = (j == 0) || (j == 3);
        const int h = (j == 0 || j == 1) ? 0 : 1;
        const unsigned so = (unsigned)kt_abs * (isx ? kstep_x : kstep_w);
        const uint32_t v0 = isx ? voX[0][h] : voW[0][h], v1 = isx ? voX[1][h] : voW[1][h];
        unsigned keep;
        if (isx)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                         "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(d0), "s"(d1), "v"(v0), "v"(v1), "s"(rsX), "s"(so) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                         "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(d0), "s"(d1), "v"(v0), "v"(v1), "s"(rsW), "s"(so) : "memory");
    };
    auto wait_halves = [&](int halves) {
        if (halves >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- fragment reads ------------------------------------------------------------------------------------------------------------------------
    // 16-lane group g = lane >> 4: column block (g & 1), r block hi = g >> 1 (8 r = two groups of 4 = 2 KiB); lane i of the group reads r = i >> 2, columns 4 (i & 3)
    __attribute__((address_space(3))) char* lds3 = (__attribute__((address_space(3))) char*)smem;
    const int i16 = lane & 15, g = lane >> 4;
    const int lane_off = (g & 1) * 128 + (g >> 1) * 2048 + (i16 >> 2) * 32 + (i16 & 3) * 8;
    const int xb = wm * 512 + lane_off;               // X-half columns wm * 64 + jj * 32 ..: blocks wm * 4 + jj * 2 + (g & 1)
    const int wb = wn * 256 + lane_off;               // W-half columns wn * 32 ..:          blocks wn * 2 + (g & 1)
    auto frag = [&](int off) -> uint4 {               // k-step ks of 16 r: groups of 4 number 4 ks + 2 hi + {0, 1}
        const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4s_t*>(lds3 + off));
        const v4s_t up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4s_t*>(lds3 + off + 1024));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), u2 = __builtin_bit_cast(uint2, up);
        return uint4{l2.x, l2.y, u2.x, u2.y};
    };

    f32x16 acc[2][4];                         // [W half i][2 * X half + jj]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 xa[2][4];                           // X fragments of the current half: [jj][ks]
    uint4 wb0[4], wb1[4];                     // W fragments of half 0 / 1: [ks]

    auto read_x = [&](int slot) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xa[jj][ks] = frag(slot * T8_HALF + xb + jj * 256 + ks * 4096);
    };
    auto read_w = [&](int slot, uint4 (&f)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = frag(slot * T8_HALF + wb + ks * 4096);
    };
    // ---- prologue: six half-tiles in flight, K-step 0 landed ---------------------------------------------------------------------------------
    const int n_half = 4 * nk;
#pragma unroll
    for (int h = 0; h < 6; ++h)
        if (h < n_half) stage(h >> 2, h & 3, ((h >> 2) & 1) * 4 + (h & 3));
    wait_halves(nk >= 2 ? 2 : 0);
    __builtin_amdgcn_s_barrier();
    if (group1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one barrier behind group 0

    auto mma = [&](const uint4 (&f)[4], int nh, int mh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[nh][2 * mh] = Mfma32x32x16<T>::run(f[ks], xa[0][ks], acc[nh][2 * mh]);
            acc[nh][2 * mh + 1] = Mfma32x32x16<T>::run(f[ks], xa[1][ks], acc[nh][2 * mh + 1]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto kstep = [&](auto par_c, int kt) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int SX0 = PAR * 4 + 0, SW0 = PAR * 4 + 1, SW1 = PAR * 4 + 2, SX1 = PAR * 4 + 3;
        // phase 0
        read_w(SW0, wb0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(SX0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(kt + 1, 2, (1 - PAR) * 4 + 2);
        __builtin_amdgcn_s_barrier();
        mma(wb0, 0, 0);
        __builtin_amdgcn_s_barrier();
        // phase 1
        read_w(SW1, wb1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(kt + 1, 3, (1 - PAR) * 4 + 3);
        __builtin_amdgcn_s_barrier();
        mma(wb1, 1, 0);
        __builtin_amdgcn_s_barrier();
        // phase 2
        read_x(SX1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) stage(kt + 2, 0, PAR * 4 + 0);
        __builtin_amdgcn_s_barrier();
        mma(wb1, 1, 1);
        __builtin_amdgcn_s_barrier();
        // phase 3
        if (kt + 2 < nk) stage(kt + 2, 1, PAR * 4 + 1);
        wait_halves(kt + 2 < nk ? 2 : 0);
        __builtin_amdgcn_s_barrier();
        mma(wb0, 0, 1);
        __builtin_amdgcn_s_barrier();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        kstep(IC<0>{}, kt);
        kstep(IC<1>{}, kt + 1);
    }
    if (kt < nk) kstep(IC<0>{}, kt);
    if (!group1) __builtin_amdgcn_s_barrier();        // balance the stagger barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: acc[i][j][4 q + e] = C[m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + l31][n0 + i * 128 + wn * 32 + 8 q + 4 hi + e] -----------------
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + i * 128 + wn * 32 + 8 * q + 4 * hi;
                const float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                store4<T>(C + (size_t)m * a.ldc + n, v);
            }
        }
}

bool gemm_wgrad_direct_ok(int dtype, int M, int N, int rows, int ldx, int ldw, int ldc) {
    return (dtype == kBF16 || dtype == kF16) && M > 0 && N > 0 && rows > 0 && M % 256 == 0 && N % 256 == 0 && rows % 64 == 0 && ldx % 8 == 0 && ldw % 8 == 0 &&
           ldc % 4 == 0 && ldx >= M && ldw >= N && ldc >= N && (size_t)rows * ldx * 2 < ((size_t)1 << 32) && (size_t)rows * ldw * 2 < ((size_t)1 << 32);
}

void launch_gemm_wgrad(int dtype, const void* dy, int lddy, const void* x, int ldx, int rows, int M, int N, void* out, int ldo, hipStream_t st) {
    LMX_REQUIRE(gemm_wgrad_direct_ok(dtype, M, N, rows, lddy, ldx, ldo),
                "gemm_wgrad: 16-bit operands, out features and in features multiples of 256, rows a multiple of 64, 16-byte aligned row strides, operands below 4 GiB");
    LMX_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0, "gemm_wgrad: operands must be 16-byte aligned");
    GemmArgs a{};
    a.X = dy; a.W = x; a.C = out; a.M = M; a.N = N; a.K = rows; a.ldx = lddy; a.ldw = ldx; a.ldc = ldo;
    const int tiles = (M >> 8) * (N >> 8);
    static std::once_flag once;                       // both instantiations decay to the same pointer type: opt both in at once
    std::call_once(once, [] {
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8t_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS));
        LMX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8t_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS));
    });
    auto launch = [&](auto kern) {
        LMX_LAUNCH(kern, dim3(tiles), dim3(512), T8_LDS, st, a);
        LMX_CHECK_HIP(hipGetLastError());
    };
    if (dtype == kBF16) launch(gemm8t_kernel<bf16_t>); else launch(gemm8t_kernel<f16_t>);
}

}  // namespace lmx
