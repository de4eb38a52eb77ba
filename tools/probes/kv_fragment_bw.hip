// Probe (round 5): does the SHAPE of the V^T reads — 128-byte / 256-byte runs at a 4 KiB row pitch — cost the decode attention its bandwidth?
// The decode attention kernels read K as contiguous rows (a 64-key piece of a head = 16 KiB in one run) but V^T [d][s_max] as 128 d-rows x (keys x 2 B):
// a 64-key piece is 128 runs of 128 bytes, 4 KiB apart.  Measured: the batch attention streams 3.2 - 4.2 TB/s where the linears stream 6 +, and in the
// single request's launch "K landed 3.4 us, V^T landed 5.2 us" (EXPERIMENTS r3-D).  This reads the SAME bytes of one layer's V cache of 8 sequences
// (8 x 32 heads x 1088 keys x 128 d x 2 B = 71 MB per "layer", 24 layers rotated so that nothing is served by the Infinity Cache) in four shapes:
//   A  piece128  one 8-wave workgroup per (sequence, head); a wave walks 64-key pieces, 16 loads per piece, each load = 8 d-rows x 128 B   (attention_batch.h)
//   D  chunk256  one 4-wave workgroup per (sequence, head, 128-key chunk); 8 loads per lane, each load = 32 d-rows x 128 B, two halves adjacent (flow_attn2)
//   B  rows      one 8-wave workgroup per (sequence, head); a wave owns 16 d-rows and reads each as ONE run of keys x 2 B (1 KiB per load instruction)
//   C  klike     the same byte count from a [keys][d] layout: a 64-key piece = 16 loads of 1 KiB, contiguous (what the K side does today)
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/kv_fragment_bw.hip -o /tmp/kv_fragment_bw && /tmp/kv_fragment_bw
//
// Not part of the library; nothing links it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int D = 128, SMAX = 2048, NH = 32;
constexpr size_t HEAD_ELEMS = (size_t)D * SMAX;                 // one head's cache of one layer (either layout), bf16 elements

__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// A: piece128
__global__ __launch_bounds__(512) void piece128(const uint16_t* __restrict__ base, int ctx, unsigned* sink) {
    const uint16_t* vt = base + (size_t)blockIdx.x * HEAD_ELEMS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = (ctx + 63) / 64;
    unsigned acc = 0;
    for (int p = wave; p < np; p += 8) {
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld16(vt + (size_t)(u * 8 + (lane >> 3)) * SMAX + p * 64 + (lane & 7) * 8);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= fold(v[u]);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// D: chunk256 (grid = heads x chunks)
__global__ __launch_bounds__(256) void chunk256(const uint16_t* __restrict__ base, int ctx, int n_chunk, unsigned* sink) {
    const int hs = blockIdx.x / n_chunk, ch = blockIdx.x % n_chunk;
    const uint16_t* vt = base + (size_t)hs * HEAD_ELEMS;
    const int tid = threadIdx.x, s8 = tid & 7, drow = tid >> 3;
    uint4 v[8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int db = 0; db < 4; ++db) v[kb * 4 + db] = ld16(vt + (size_t)(db * 32 + drow) * SMAX + ch * 128 + kb * 64 + s8 * 8);
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= fold(v[u]);
    if (acc == 0x12345u) sink[0] = acc;
}

// B: rows (a wave owns 16 d-rows; per row ceil(ctx / 512) loads of 1 KiB)
__global__ __launch_bounds__(512) void rows(const uint16_t* __restrict__ base, int ctx, unsigned* sink) {
    const uint16_t* vt = base + (size_t)blockIdx.x * HEAD_ELEMS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nl = (ctx + 511) / 512;
    unsigned acc = 0;
    for (int j = 0; j < nl; ++j) {
        uint4 v[16];
        int k = j * 512 + lane * 8; k = k < ctx ? k : 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld16(vt + (size_t)(wave * 16 + u) * SMAX + k);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= fold(v[u]);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// C: klike ([keys][d]: a 64-key piece = 16 KiB contiguous)
__global__ __launch_bounds__(512) void klike(const uint16_t* __restrict__ base, int ctx, unsigned* sink) {
    const uint16_t* kc = base + (size_t)blockIdx.x * HEAD_ELEMS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = (ctx + 63) / 64;
    unsigned acc = 0;
    for (int p = wave; p < np; p += 8) {
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld16(kc + (size_t)(p * 64 + u * 4 + (lane >> 4)) * D + (lane & 15) * 8);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= fold(v[u]);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// klike with the chunk kernel's geometry (4 waves per 128-key chunk, 8 loads per lane)
__global__ __launch_bounds__(256) void kchunk(const uint16_t* __restrict__ base, int ctx, int n_chunk, unsigned* sink) {
    const int hs = blockIdx.x / n_chunk, ch = blockIdx.x % n_chunk;
    const uint16_t* kc = base + (size_t)hs * HEAD_ELEMS;
    const int tid = threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld16(kc + (size_t)(ch * 128 + u * 16 + (tid >> 4)) * D + (tid & 15) * 8);
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= fold(v[u]);
    if (acc == 0x12345u) sink[0] = acc;
}

int main() {
    const int ctx = 1088, n_chunk = (ctx + 127) / 128, LAYERS = 24;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* sink; CK(hipMalloc(&sink, 256));
    for (int nseq : {1, 8, 32}) {
        const int nhs = nseq * NH;
        const size_t layer_elems = (size_t)nhs * HEAD_ELEMS;
        const int layers = nseq == 32 ? 6 : nseq == 1 ? 64 : LAYERS;
        uint16_t* buf; CK(hipMalloc(&buf, layer_elems * layers * 2)); CK(hipMemset(buf, 1, layer_elems * layers * 2)); CK(hipDeviceSynchronize());
        const double mb = (double)nhs * ctx * D * 2 / 1e6;                   // bytes one launch reads
        auto run = [&](const char* name, const std::function<void(const uint16_t*)>& launch) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int l = 0; l < layers; ++l) launch(buf + l * layer_elems);
            CK(hipStreamSynchronize(st));
            std::vector<float> ts;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, st));
                for (int l = 0; l < layers; ++l) launch(buf + l * layer_elems);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f / layers);
            }
            std::sort(ts.begin(), ts.end());
            printf("{\"sequences\": %d, \"shape\": \"%s\", \"MB_per_launch\": %.1f, \"us_per_launch\": %.2f, \"TBps\": %.2f}\n", nseq, name, mb, ts[2], mb / ts[2]);
            fflush(stdout);
        };
        run("A piece128 (8 waves per head, 128-B runs at 4 KiB pitch)", [&](const uint16_t* b) { hipLaunchKernelGGL(piece128, dim3(nhs), dim3(512), 0, st, b, ctx, sink); });
        run("D chunk256 (4 waves per 128-key chunk, 2 x 128-B runs)", [&](const uint16_t* b) { hipLaunchKernelGGL(chunk256, dim3(nhs * n_chunk), dim3(256), 0, st, b, ctx, n_chunk, sink); });
        run("B rows (a wave reads whole d-rows, 1 KiB per load)", [&](const uint16_t* b) { hipLaunchKernelGGL(rows, dim3(nhs), dim3(512), 0, st, b, ctx, sink); });
        run("C klike (8 waves per head, [keys][d] contiguous)", [&](const uint16_t* b) { hipLaunchKernelGGL(klike, dim3(nhs), dim3(512), 0, st, b, ctx, sink); });
        run("E kchunk (4 waves per 128-key chunk, [keys][d] contiguous)", [&](const uint16_t* b) { hipLaunchKernelGGL(kchunk, dim3(nhs * n_chunk), dim3(256), 0, st, b, ctx, n_chunk, sink); });
        CK(hipFree(buf));
    }
    return 0;
}
