// Experiment (not part of the library): how fast can ONE CU be fed from L2 / Infinity Cache / HBM, by direct-to-LDS DMA
// (global_load_lds_dwordx4, the GEMM kernels' staging) and by ordinary 16-byte loads into registers, as a function of the
// bytes in flight?  One 512-thread block per CU (128 KB of LDS requested, like gemm256_kernel), every wave instruction
// fetches 8 rows x 128 bytes (the GEMM's K-tile rows), a wave keeps INF instructions (INF KB) in flight.
//   footprint 2 MB  : all blocks stream the same 2 MB            -> L2 hits
//   footprint 96 MB : each XCD's blocks sweep 96 MB               -> Infinity Cache hits (L2 misses)
//   footprint 2 GB  :                                             -> HBM
// build: hipcc --offload-arch=gfx950 -O3 exp_l2_feed.hip -o exp_l2_feed.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#ifndef ROWBYTES
#define ROWBYTES 2560
#endif
constexpr int kRowBytes = ROWBYTES;  // default: K = 1280 bf16
#ifndef CONTIG
#define CONTIG 0
#endif
constexpr bool kContig = CONTIG;

__device__ __forceinline__ const char* src_of(const char* base, size_t footprint, long long seg, int lane) {
    // segment = 8 rows x 128 B at row stride kRowBytes: 20 segments side by side cover 8 full rows (20 KB)
    if (kContig) return base + ((size_t)seg * 1024 + lane * 16) % footprint;  // 1 KB contiguous per wave instruction
    const long long grp = seg / (kRowBytes / 128), col = seg % (kRowBytes / 128);
    size_t off = (size_t)grp * (8 * kRowBytes) + (size_t)(lane >> 3) * kRowBytes + (size_t)col * 128 + (lane & 7) * 16;
    return base + off % footprint;
}

template <int INF>
__global__ __launch_bounds__(512) void dma_feed(const char* base, size_t footprint, int iters, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* dst = smem + wave * (INF * 1024);
    long long seg = ((long long)blockIdx.x * 8 + wave) * 977;  // blocks start apart
    for (int i = 0; i < INF; ++i)
        __builtin_amdgcn_global_load_lds(src_of(base, footprint, seg + i, lane), (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    for (int it = INF; it < iters; ++it) {
        if constexpr (INF == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (INF == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if constexpr (INF == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (INF == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        __builtin_amdgcn_global_load_lds(src_of(base, footprint, seg + it, lane),
                                         (__attribute__((address_space(3))) void*)(dst + (it % INF) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[threadIdx.x] == 0x5a && out) out[0] = 1;
}

template <int INF>
__global__ __launch_bounds__(512) void reg_feed(const char* base, size_t footprint, int iters, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long seg = ((long long)blockIdx.x * 8 + wave) * 977;
    unsigned acc = 0;
    for (int it = 0; it < iters; it += INF) {
        u32x4_t v[INF];
#pragma unroll
        for (int i = 0; i < INF; ++i) v[i] = *reinterpret_cast<const u32x4_t*>(src_of(base, footprint, seg + it + i, lane));
#pragma unroll
        for (int i = 0; i < INF; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    }
    if (acc == 0x12345678u && out) { out[0] = acc; smem[0] = 1; }
}

template <class K>
void run(const char* name, K k, int blocks, size_t lds, const char* buf, size_t fp, int iters, unsigned* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, 512, lds>>>(buf, fp, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<<<blocks, 512, lds>>>(buf, fp, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double bytes = (double)blocks * 8 * iters * 1024;
    printf(" %s %6.1f", name, bytes / blocks / ms / 1e6);  // GB/s per block (= per CU)
}

int main() {
    unsigned* out;
    hipMalloc(&out, 4);
    const size_t big = (size_t)2 << 30;
    char* buf;
    hipMalloc(&buf, big);
    hipMemset(buf, 1, big);
    const size_t lds = 128 * 1024;
    struct { const char* n; size_t fp; } fps[] = {{"L2 (2 MB)", (size_t)2 << 20}, {"MALL (96 MB)", (size_t)96 << 20}, {"HBM (2 GB)", big}};
    for (int blocks : {256, 64}) {
        for (auto f : fps) {
            const int iters = 4096;
            printf("%3d blocks, %-13s GB/s per CU: DMA in flight/wave", blocks, f.n);
            run("1KB", dma_feed<1>, blocks, lds, buf, f.fp, iters, out);
            run("2KB", dma_feed<2>, blocks, lds, buf, f.fp, iters, out);
            run("4KB", dma_feed<4>, blocks, lds, buf, f.fp, iters, out);
            run("8KB", dma_feed<8>, blocks, lds, buf, f.fp, iters, out);
            run("16KB", dma_feed<16>, blocks, lds, buf, f.fp, iters, out);
            printf(" | regs");
            run("2KB", reg_feed<2>, blocks, lds, buf, f.fp, iters, out);
            run("4KB", reg_feed<4>, blocks, lds, buf, f.fp, iters, out);
            run("8KB", reg_feed<8>, blocks, lds, buf, f.fp, iters, out);
            run("16KB", reg_feed<16>, blocks, lds, buf, f.fp, iters, out);
            printf("\n");
        }
    }
    return 0;
}
