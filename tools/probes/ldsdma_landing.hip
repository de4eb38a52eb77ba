// Probe for the two-phase K-step arm of gemm8p: how long does one LDS-DMA half-tile (128 rows x 64 k of a 16-bit operand = 16 KiB, two
// buffer_load_dwordx4 ... lds per wave of a 512-thread workgroup, the `stage` of csrc/gemm8p.hip) take from issue to landed when the operand is L2 / MALL
// resident and every CU of the chip is doing the same — the situation of a prefill GEMM's main loop.  The arm leaves one half-tile per K-step only two
// phases (~1200 clk ~ 0.63 us) of flight time; the four-phase schedule gives every half-tile ~1.17 us.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_landing.hip -o /tmp/ldsdma_landing && /tmp/ldsdma_landing
//
// Prints, per "depth" (half-tiles kept in flight per wave: 1, 2, 4, 6), min / median / p90 issue->landed time of a half-tile in ns (s_memrealtime, 100 MHz)
// and the chip-wide rate.  Not part of the library; nothing links it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i_t make_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// operand: [rows][K] 16-bit, row pitch K * 2 bytes.  Workgroup b walks K-steps of its own 256-row panel (as a GEMM tile does): half-tile h = rows
// 128 * (h & 1) .. + 127 of K-step h >> 1.  DEPTH half-tiles stay in flight; the time of every DEPTH-th one is recorded by wave 0.
template <int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ op, uint32_t op_bytes, int K, int n_half, uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t rs = make_rsrc(op, op_bytes);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    const int panel = blockIdx.x % 64;                                   // 64 panels of 256 rows: a 16384-row operand, shared by 4 workgroups each (L2 hits)
    const int rb = 8 * wave + (lane >> 3);
    const int chunk = ((lane & 7) ^ (rb >> 1)) & 7;
    uint32_t vo[2][2];
    for (int i = 0; i < 2; ++i)
        for (int h = 0; h < 2; ++h) vo[i][h] = (uint32_t)(((size_t)(panel * 256 + h * 128 + i * 64 + rb) * K + chunk * 8) * 2);
    auto stage = [&](int hidx) {
        const int slot = hidx & 7, kt = hidx >> 1, h = hidx & 1;
        const unsigned d0 = lds_wave + slot * 16384, d1 = d0 + 8192;
        const unsigned so = (unsigned)kt * 128u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(d0), "s"(d1), "v"(vo[0][h]), "v"(vo[1][h]), "s"(rs), "s"(so) : "memory");
    };
    uint64_t t_issue[8];
    int issued = 0, landed = 0;
    for (; issued < DEPTH && issued < n_half; ++issued) { t_issue[issued & 7] = __builtin_amdgcn_s_memrealtime(); stage(issued); }
    while (landed < n_half) {
        // wait until the OLDEST half-tile in flight has landed: at most (in flight - 1) half-tiles = 2 x that many loads may remain
        const int fl = issued - landed;
        if (fl >= 6) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (fl == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (fl == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (fl == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (fl == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
        if (wave == 0 && lane == 0) out[(size_t)blockIdx.x * n_half + landed] = t1 - t_issue[landed & 7];
        ++landed;
        __builtin_amdgcn_s_barrier();                                    // the consumer side of a GEMM phase: all waves' pieces are there
        if (issued < n_half) { t_issue[issued & 7] = __builtin_amdgcn_s_memrealtime(); stage(issued); ++issued; }
    }
}

int main() {
    const int K = 4096, rows = 16384, n_half = 2 * (K / 64);              // one pass over K per workgroup
    const size_t bytes = (size_t)rows * K * 2;                           // 128 MiB: L2 misses on first touch, MALL / L2 hits for the 3 other workgroups of a panel
    char* op; CK(hipMalloc(&op, bytes)); CK(hipMemset(op, 1, bytes));
    const int grid = 256;
    uint64_t* out; CK(hipMalloc(&out, (size_t)grid * n_half * 8));
    std::vector<uint64_t> h((size_t)grid * n_half);
    auto run = [&](auto kern, int depth) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, op, (uint32_t)bytes, K, n_half, out);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<uint64_t> v(h.begin(), h.end()); std::sort(v.begin(), v.end());
        printf("depth %d: half-tile issue->landed min %llu ns, median %llu ns, p90 %llu ns; launch %.1f us = %.2f TB/s L2->LDS over the chip\n", depth,
               (unsigned long long)v.front() * 10, (unsigned long long)v[v.size() / 2] * 10, (unsigned long long)v[v.size() * 9 / 10] * 10, ms * 1e3,
               (double)grid * n_half * 16384 / (ms * 1e-3) / 1e12);
    };
    run(probe<1>, 1); run(probe<2>, 2); run(probe<4>, 4); run(probe<6>, 6);
    return 0;
}
