// Probe (round 6): how fast can the training step's operand transposes ([rows][cols] -> [cols][rows], 16-bit) run, and what sets the rate?
// The wgrad of every linear transposes two activation-sized operands (32768 rows at 16 x 2048); profiles/r06_config5_kernel_stats.csv has them at 6.3 % of the
// step.  Variants of train.hip's transpose16_vec_kernel: tile shape, grid order (which tiles run concurrently -> which HBM channels the 256-byte output
// segments at a power-of-two pitch fall on), LDS gather by ds_read_u16 against ds_read_b64_tr_b16.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/transpose_bw.hip -o /tmp/transpose_bw && /tmp/transpose_bw
//
// Not part of the library; nothing links it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ORDER 0: column tiles fastest (the shipping launch), 1: row tiles fastest, 2: row tiles fastest in groups of 8 (XCD round-robin keeps a group's 8 tiles on 8 XCDs)
template <int TR, int TC, int ORDER>
__global__ __launch_bounds__(256) void tr_kernel(const uint16_t* __restrict__ src, int lds_, int rows, int cols, uint16_t* __restrict__ dst, int ldd, int gx, int gy) {
    constexpr int PITCH = TC + 8;
    __shared__ __attribute__((aligned(16))) uint16_t tile[TR * PITCH];
    int bx, by;
    if (ORDER == 0) { bx = blockIdx.x % gx; by = blockIdx.x / gx; }
    else { by = blockIdx.x % gy; bx = blockIdx.x / gy; }
    const int r0 = by * TR, c0 = bx * TC, tid = threadIdx.x;
    constexpr int CPR = TC / 8;                                            // 16-byte chunks per source row
#pragma unroll
    for (int it = 0; it < TR * TC / 8 / 256; ++it) {
        const int i = tid + it * 256, r = i / CPR, ch = i % CPR;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + r) * lds_ + c0 + ch * 8);
        *reinterpret_cast<uint4*>(tile + r * PITCH + ch * 8) = v;
    }
    __syncthreads();
    constexpr int SPR = TR / 8;                                            // 16-byte segments per output row
#pragma unroll
    for (int it = 0; it < TR * TC / 8 / 256; ++it) {
        const int i = tid + it * 256, c = i / SPR, sg = i % SPR;
        if (c0 + c >= cols || r0 + sg * 8 >= rows) continue;
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = tile[(sg * 8 + j) * PITCH + c];
        uint4 v;
        v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16); v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
        *reinterpret_cast<uint4*>(dst + (size_t)(c0 + c) * ldd + r0 + sg * 8) = v;
    }
}

// the same tile with the LDS image swizzled instead of padded: 16-byte chunk q of row r is stored at chunk q ^ ((r >> 3) & 7) (pitch = TC, no padding), so the 8
// lanes of a wave that gather the same column from 8 different row groups (sg = 0 .. 7) hit 8 different chunks = 32 different banks; with the padded image
// (pitch 136 or 72 elements: 8 rows = a multiple of 128 bytes) they all hit the same 4 banks.
template <int TR, int TC, int ORDER>
__global__ __launch_bounds__(256) void tr_swz_kernel(const uint16_t* __restrict__ src, int lds_, int rows, int cols, uint16_t* __restrict__ dst, int ldd, int gx, int gy) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[TR * TC];
    int bx, by;
    if (ORDER == 0) { bx = blockIdx.x % gx; by = blockIdx.x / gx; }
    else { by = blockIdx.x % gy; bx = blockIdx.x / gy; }
    const int r0 = by * TR, c0 = bx * TC, tid = threadIdx.x;
    constexpr int CPR = TC / 8;
#pragma unroll
    for (int it = 0; it < TR * TC / 8 / 256; ++it) {
        const int i = tid + it * 256, r = i / CPR, ch = i % CPR;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + r) * lds_ + c0 + ch * 8);
        *reinterpret_cast<uint4*>(tile + r * TC + ((ch ^ ((r >> 3) & 7)) * 8)) = v;
    }
    __syncthreads();
    constexpr int SPR = TR / 8;
#pragma unroll
    for (int it = 0; it < TR * TC / 8 / 256; ++it) {
        const int i = tid + it * 256, c = i / SPR, sg = i % SPR;
        if (c0 + c >= cols || r0 + sg * 8 >= rows) continue;
        const int col = (((c >> 3) ^ (sg & 7)) << 3) + (c & 7);
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = tile[(sg * 8 + j) * TC + col];
        uint4 v;
        v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16); v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
        *reinterpret_cast<uint4*>(dst + (size_t)(c0 + c) * ldd + r0 + sg * 8) = v;
    }
}

// plain copy with the same access shapes removed: the memory system's rate for this many bytes (upper bound)
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

template <int TR, int TC, int ORDER>
static double run_swz(const uint16_t* src, uint16_t* dst, int rows, int cols, int ldd) {
    const int gx = (cols + TC - 1) / TC, gy = (rows + TR - 1) / TR;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) tr_swz_kernel<TR, TC, ORDER><<<gx * gy, 256>>>(src, cols, rows, cols, dst, ldd, gx, gy);
    CK(hipEventRecord(e0));
    const int reps = 6;
    for (int i = 0; i < reps; ++i) tr_swz_kernel<TR, TC, ORDER><<<gx * gy, 256>>>(src, cols, rows, cols, dst, ldd, gx, gy);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms / reps * 1e3;
}

template <int TR, int TC, int ORDER>
static double run(const uint16_t* src, uint16_t* dst, int rows, int cols, int ldd) {
    const int gx = (cols + TC - 1) / TC, gy = (rows + TR - 1) / TR;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) tr_kernel<TR, TC, ORDER><<<gx * gy, 256>>>(src, cols, rows, cols, dst, ldd, gx, gy);
    CK(hipEventRecord(e0));
    const int reps = 6;
    for (int i = 0; i < reps; ++i) tr_kernel<TR, TC, ORDER><<<gx * gy, 256>>>(src, cols, rows, cols, dst, ldd, gx, gy);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms / reps * 1e3;
}

static bool check(const uint16_t* d_src, const uint16_t* d_dst, int rows, int cols, int ldd) {
    std::vector<uint16_t> s((size_t)rows * cols), d((size_t)cols * ldd);
    CK(hipMemcpy(s.data(), d_src, s.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), d_dst, d.size() * 2, hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; r += 37) for (int c = 0; c < cols; c += 11) if (s[(size_t)r * cols + c] != d[(size_t)c * ldd + r]) return false;
    return true;
}

int main() {
    const int shapes[][2] = {{32768, 4096}, {32768, 11008}, {32768, 12288}, {8192, 4096}, {11008, 4096}, {4096, 11008}};
    const int pads[] = {0, 64};                                            // extra elements on the OUTPUT pitch (breaks the power-of-two pitch)
    uint16_t *src, *dst;
    const size_t cap = (size_t)32768 * 12288 + 12288 * 64;
    CK(hipMalloc(&src, cap * 2)); CK(hipMalloc(&dst, cap * 2));
    std::vector<uint16_t> h(cap);
    uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(x >> 16); }
    CK(hipMemcpy(src, h.data(), cap * 2, hipMemcpyHostToDevice));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t n16 = (size_t)32768 * 12288 / 8;
        copy_kernel<<<256 * 16, 256>>>((const uint4*)src, (uint4*)dst, n16);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) copy_kernel<<<256 * 16, 256>>>((const uint4*)src, (uint4*)dst, n16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"copy\", \"GB\": %.3f, \"us\": %.1f, \"TBps_read_plus_write\": %.2f}\n", n16 * 32 / 1e9, ms / 4 * 1e3, n16 * 32 / (ms / 4 * 1e-3) / 1e12);
    }
    for (auto& sh : shapes)
        for (int pad : pads) {
            const int rows = sh[0], cols = sh[1], ldd = rows + pad;
            const double bytes = 2.0 * rows * cols * 2;
            struct { const char* name; double us; } res[] = {
                {"128x64 cols-fastest (shipping)", run<128, 64, 0>(src, dst, rows, cols, ldd)},
                {"128x64 rows-fastest", run<128, 64, 1>(src, dst, rows, cols, ldd)},
                {"128x128 cols-fastest", run<128, 128, 0>(src, dst, rows, cols, ldd)},
                {"128x128 rows-fastest", run<128, 128, 1>(src, dst, rows, cols, ldd)},
                {"64x128 cols-fastest", run<64, 128, 0>(src, dst, rows, cols, ldd)},
                {"64x128 rows-fastest", run<64, 128, 1>(src, dst, rows, cols, ldd)},
                {"256x64 rows-fastest", run<256, 64, 1>(src, dst, rows, cols, ldd)},
                {"64x256 cols-fastest", run<64, 256, 0>(src, dst, rows, cols, ldd)},
                {"32x256 cols-fastest", run<32, 256, 0>(src, dst, rows, cols, ldd)},
                {"64x128 swizzled cols-fastest", run_swz<64, 128, 0>(src, dst, rows, cols, ldd)},
                {"64x128 swizzled rows-fastest", run_swz<64, 128, 1>(src, dst, rows, cols, ldd)},
                {"128x128 swizzled cols-fastest", run_swz<128, 128, 0>(src, dst, rows, cols, ldd)},
                {"64x256 swizzled cols-fastest", run_swz<64, 256, 0>(src, dst, rows, cols, ldd)},
            };
            const bool ok = check(src, dst, rows, cols, ldd);
            for (auto& r : res)
                printf("{\"rows\": %d, \"cols\": %d, \"out_pitch\": %d, \"form\": \"%s\", \"us\": %.1f, \"TBps\": %.2f, \"last_ok\": %s}\n", rows, cols, ldd, r.name, r.us,
                       bytes / (r.us * 1e-6) / 1e12, ok ? "true" : "false");
        }
    return 0;
}
