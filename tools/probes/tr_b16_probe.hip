// Probe (round 6): what ds_read_b64_tr_b16 delivers, and which LDS layouts it reads without bank conflicts.
//
// Question behind it: a GEMM whose operand is stored [contraction][rows] (rows contiguous: the "TN" / "NN" forms a training step's wgrad / dgrad need) can
// keep the LDS image the LDS-DMA writes (16-byte chunks of 8 rows at one contraction index, placed anywhere) and fetch MFMA fragments with the
// transpose read - IF the image is one the read serves at full rate.  The guide (cdna_hip_programming.md T10) names two conflict-free images and one
// 4-way conflicting one, which a (a/4) mod 64 bank model does not separate; this measures instead of guessing.
//
//   part 1  semantics: LDS holds lds16[i] = i; every lane reads 8 bytes at lane * 8; print what each lane got.
//   part 2  rate: lane (g = lane / 16, i = lane % 16) reads at  G[g] + (i / 4) * R + (i % 4) * 8  for a list of (G, R); 8 waves x 4096 reads per wave,
//           cycles per wave-instruction per CU from s_memtime.  2.0 = the LDS array's full rate for a 64-bit read.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/tr_b16_probe.hip -o /tmp/tr_b16_probe && /tmp/tr_b16_probe
//
// Not part of the library; nothing links it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u2 tr_read(unsigned addr) {
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ u2 plain_read(unsigned addr) {
    u2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

__global__ void semantics(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    u2 v = tr_read((unsigned)(size_t)lds + threadIdx.x * 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}

// per-lane byte offsets come from the host: offs[64]; each wave adds nothing (all waves read the same image: identical addresses across waves do not
// interact, the LDS serves one wave-instruction at a time)
template <bool TR>
__global__ __launch_bounds__(512) void rate(const unsigned* __restrict__ offs, int iters, long long* cycles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(dyn)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)dyn + offs[threadIdx.x & 63];
    unsigned acc = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u2 a, b, c, d;
        if (TR) { a = tr_read(base); b = tr_read(base + 8192); c = tr_read(base + 16384); d = tr_read(base + 24576); }
        else { a = plain_read(base); b = plain_read(base + 8192); c = plain_read(base + 16384); d = plain_read(base + 24576); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc ^= a.x ^ b.y ^ c.x ^ d.y;
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x1234567u) sink[0] = acc;
}

struct Pat { std::string name; unsigned G[4]; unsigned R; unsigned C; };   // C: byte step of (i % 4), normally 8

int main() {
    // ---- part 1 -------------------------------------------------------------------------------------------------------------------------
    unsigned short* d_out; CK(hipMalloc(&d_out, 64 * 4 * 2));
    semantics<<<1, 64>>>(d_out);
    std::vector<unsigned short> h(256);
    CK(hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost));
    printf("semantics: lds16[i] = i, lane reads 8 bytes at lane * 8 (so lane L's own four elements are 4L .. 4L+3)\n");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "   |");
    // ---- part 2 -------------------------------------------------------------------------------------------------------------------------
    std::vector<Pat> pats;
    auto add = [&](const char* n, unsigned g0, unsigned g1, unsigned g2, unsigned g3, unsigned R, unsigned C = 8) { pats.push_back({n, {g0, g1, g2, g3}, R, C}); };
    add("linear lane*8 (R=32, groups 128 apart)", 0, 128, 256, 384, 32);
    add("guide: blocks 512 apart, R=32", 0, 512, 1024, 1536, 32);
    add("blocks 1024 apart, R=32", 0, 1024, 2048, 3072, 32);
    add("blocks 2048 apart, R=32", 0, 2048, 4096, 6144, 32);
    add("blocks 256 apart, R=32", 0, 256, 512, 768, 32);
    add("blocks 384 apart, R=32", 0, 384, 768, 1152, 32);
    add("blocks 640 apart, R=32", 0, 640, 1280, 1920, 32);
    add("32x32x16 on [k/4][n/16][4][16], NT=256: g0 g1 adjacent, g2 g3 +2 k-blocks", 0, 128, 4096, 4224, 32);
    add("row-major [k][256 n] (R=512), 32x32x16 groups: n+16, k+8", 0, 32, 4096, 4128, 512);
    add("row-major [k][256 n] (R=512), 16x16x32 groups: k+8 each", 0, 4096, 8192, 12288, 512);
    add("row pitch 528 (R=528), 16x16x32 groups k+8", 0, 4224, 8448, 12672, 528);
    add("row pitch 544 (R=544), 16x16x32 groups k+8", 0, 4352, 8704, 13056, 544);
    add("row pitch 576 (R=576), 16x16x32 groups k+8", 0, 4608, 9216, 13824, 576);
    add("row pitch 640 (R=640), 16x16x32 groups k+8", 0, 5120, 10240, 15360, 640);
    add("R=64, groups 256 apart", 0, 256, 512, 768, 64);
    add("R=64, groups 32 apart (two n-blocks side by side in a 64-byte row), then +256", 0, 32, 256, 288, 64);
    add("R=128, groups 32 apart in a 128-byte row", 0, 32, 64, 96, 128);
    add("R=128, groups 512 apart", 0, 512, 1024, 1536, 128);
    add("R=256, groups 32 apart", 0, 32, 64, 96, 256);
    add("R=256, groups 64 apart", 0, 64, 128, 192, 256);
    add("R=512, groups 32 apart (row-major, 4 n-blocks, same k)", 0, 32, 64, 96, 512);
    add("R=528, groups 32 apart", 0, 32, 64, 96, 528);
    add("R=544, groups 32 apart", 0, 32, 64, 96, 544);
    add("R=576, groups 32 apart", 0, 32, 64, 96, 576);
    add("R=640, groups 32 apart", 0, 32, 64, 96, 640);
    add("R=1024, groups 32 apart", 0, 32, 64, 96, 1024);
    add("R=32, all groups the same block (broadcast)", 0, 0, 0, 0, 32);
    add("R=8 C=32 (lane's 4 chunks column-wise: i/4 steps 8, i%4 steps 32), groups 128", 0, 128, 256, 384, 8, 32);
    unsigned* d_offs; long long* d_cyc; unsigned* d_sink;
    CK(hipMalloc(&d_offs, 256)); CK(hipMalloc(&d_cyc, 64 * 8)); CK(hipMalloc(&d_sink, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 1024, blocks = 8;
    printf("\nrate: LDS cycles per wave-instruction (8 waves on one CU, 4 reads per iteration, %d iterations); 2.0 = full rate\n", iters);
    for (auto& p : pats) {
        unsigned offs[64];
        for (int l = 0; l < 64; ++l) { const int g = l / 16, i = l % 16; offs[l] = p.G[g] + (i / 4) * p.R + (i % 4) * p.C; }
        double res[2];
        for (int tr = 0; tr < 2; ++tr) {
            CK(hipMemcpy(d_offs, offs, 256, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; ++rep) {
                if (tr) rate<true><<<blocks, 512, 65536>>>(d_offs, iters, d_cyc, d_sink); else rate<false><<<blocks, 512, 65536>>>(d_offs, iters, d_cyc, d_sink);
            }
            CK(hipDeviceSynchronize());
            long long cyc[8]; CK(hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
            long long mn = cyc[0]; for (int b = 1; b < blocks; ++b) mn = cyc[b] < mn ? cyc[b] : mn;
            res[tr] = (double)mn / ((double)iters * 4 * 8);
        }
        printf("  tr %6.2f   plain b64 %6.2f   %s\n", res[1], res[0], p.name.c_str());
    }
    return 0;
}
