// Probe: what the exchange buffers of csrc/p2p.hip cost as MEMORY.  The two-shot all-reduce of a 1087 x 4096 bf16 message (8.9 MB) takes 349 us between two
// processes on one MI355X (tests/test_tp_p2p_gpu.py::test_two_shot_allreduce_real_width) where its flag protocol alone is ~30 us: this measures plain copies of
// the same size with the all-reduce's own launch shape (544 workgroups x 256 threads, 16 bytes per lane and step) between ordinary device memory and memory from
// hipExtMallocWithFlags(hipDeviceMallocUncached) / (hipDeviceMallocFinegrained), in both directions, cold and repeated.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/uncached_bw.hip -o /tmp/uncached_bw && /tmp/uncached_bw
//
// Not part of the library; nothing links it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n_vec, int per_block) {
    // block b owns vectors [b * per_block, (b + 1) * per_block): the slice walk of p2p_allreduce_big_kernel
    const size_t v0 = (size_t)blockIdx.x * per_block;
    for (int c = threadIdx.x; c < per_block; c += 256)
        if (v0 + c < n_vec) dst[v0 + c] = src[v0 + c];
}
__global__ __launch_bounds__(256) void sum_kernel(uint4* __restrict__ dst, const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n_vec, int per_block) {
    const size_t v0 = (size_t)blockIdx.x * per_block;
    for (int c = threadIdx.x; c < per_block; c += 256)
        if (v0 + c < n_vec) { uint4 x = a[v0 + c], y = b[v0 + c]; dst[v0 + c] = uint4{x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w}; }
}

static float time_us(hipStream_t st, int reps, const std::function<void()>& f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipStreamSynchronize(st));
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st)); f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const size_t bytes = (size_t)1087 * 4096 * 2;                 // the prefill message
    const size_t n_vec = bytes / 16;
    const int per_block = 4096 * 2 / 16;                          // one 4096-element slice = 512 vectors
    const int grid = (int)((n_vec + per_block - 1) / per_block);  // 1087 workgroups for the whole message (the all-reduce launches 544 per rank, each touching 2 x as much)
    hipStream_t st; CK(hipStreamCreate(&st));
    void *n0, *n1, *uc, *fg;
    CK(hipMalloc(&n0, bytes)); CK(hipMalloc(&n1, bytes));
    if (hipExtMallocWithFlags(&uc, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); uc = nullptr; }
    if (hipExtMallocWithFlags(&fg, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); fg = nullptr; }
    CK(hipMemset(n0, 1, bytes)); CK(hipMemset(n1, 2, bytes));
    if (uc) CK(hipMemset(uc, 3, bytes));
    if (fg) CK(hipMemset(fg, 4, bytes));
    CK(hipDeviceSynchronize());
    struct Case { const char* name; void* dst; void* src; };
    std::vector<Case> cases = {{"device -> device", n1, n0}};
    if (uc) { cases.push_back({"device -> uncached (store side)", uc, n0}); cases.push_back({"uncached -> device (load side)", n1, uc}); }
    if (fg) { cases.push_back({"device -> finegrained (store side)", fg, n0}); cases.push_back({"finegrained -> device (load side)", n1, fg}); }
    printf("message %.2f MB, %d workgroups x 256 threads, 16 B per lane and step\n", bytes / 1e6, grid);
    for (auto& c : cases) {
        const float us = time_us(st, 20, [&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, st, (uint4*)c.dst, (const uint4*)c.src, n_vec, per_block); });
        printf("%-38s %8.1f us  %7.1f GB/s (read + write)\n", c.name, us, 2.0 * bytes / us / 1e3);
    }
    if (uc) {
        const float us = time_us(st, 20, [&] { hipLaunchKernelGGL(sum_kernel, dim3(grid), dim3(256), 0, st, (uint4*)n1, (const uint4*)n0, (const uint4*)uc, n_vec, per_block); });
        printf("%-38s %8.1f us\n", "device + uncached -> device (reduce)", us);
    }
    if (fg) {
        const float us = time_us(st, 20, [&] { hipLaunchKernelGGL(sum_kernel, dim3(grid), dim3(256), 0, st, (uint4*)n1, (const uint4*)n0, (const uint4*)fg, n_vec, per_block); });
        printf("%-38s %8.1f us\n", "device + finegrained -> device (reduce)", us);
    }
    return 0;
}
