// How well can MI355X overlap matrix-core work with a streaming read?  Three synthetic kernels, each alone and in pairs on two streams:
//   mfma:   256 (or 512) threads per block, N blocks per CU, registers only: a chain of independent 16x16x32 bf16 MFMAs, no memory
//   stream: the decode GEMV's access pattern without its arithmetic: 1024-thread blocks, every wave reads 1-KB contiguous pieces of a
//           2-GB buffer (16 B per lane, 8 loads in flight), one xor-reduce per load
//   gemmio: the tile GEMM's memory behaviour without its MFMAs: 512-thread blocks with 128 KB of LDS (one per CU), global_load ->
//           LDS DMA of 64-KB "K tiles" of the A / W panels the real kernel's tile order assigns (SAM mlp1: 82 % L2 hits), plus
//           128-byte-line stores of an output tile
// build: hipcc --offload-arch=gfx950 -O3 -o exp_overlap.bin exp_overlap.hip ; run: ./exp_overlap.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- MFMA only: `iters` rounds of 16 independent accumulators per wave -----------------------------------------------------
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(float)(threadIdx.x + i);
        b[i] = (__bf16)(float)(blockIdx.x + i);
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

// ---- streaming read: blocks walk the buffer in 1-KB wave pieces ---------------------------------------------------------------
template <int NT>  // 0 plain loads, 1 non-temporal
__global__ __launch_bounds__(1024) void stream_kernel(const uint4* __restrict__ buf, size_t n16, uint32_t* out) {
    const size_t lane = threadIdx.x & 63, wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 16;
    uint4 acc = {0, 0, 0, 0};
    // wave w reads pieces w, w + nwaves, ...: 8 loads in flight
    for (size_t p = wave * 64; p + 64 * 7 * nwaves + 64 <= n16; p += 64 * 8 * nwaves) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u4;
            const u4* q = reinterpret_cast<const u4*>(&buf[p + (size_t)j * 64 * nwaves + lane]);
            u4 t;
            if constexpr (NT == 0) t = *q;
            else if constexpr (NT == 1) t = __builtin_nontemporal_load(q);
            else t = __builtin_nontemporal_load(q);
            v[j] = make_uint4(t[0], t[1], t[2], t[3]);
        }

#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

// ---- the tile GEMM's memory side: DMA of K tiles into LDS + output lines, no MFMAs ----------------------------------------------
template <bool PANEL, bool MFMA>  // PANEL: operands in the K-panel layout (a K tile of a panel is 32 KB contiguous: 1 KB per wave instruction)
__global__ __launch_bounds__(512) void gemmio_kernel(const unsigned char* __restrict__ ops_, size_t ops_bytes, unsigned char* __restrict__ out,
                                                     int ktiles, int tiles_per_block) {  // MFMA: + the 64 MFMAs per wave and K tile of the real kernel (register operands)
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) {
        fa[i] = (__bf16)(float)(threadIdx.x + i);
        fb[i] = (__bf16)(float)(blockIdx.x + i);
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    for (int t = 0; t < tiles_per_block; ++t) {
        // the tile this block would own in launch round t of the real kernel (gemm_tile_origin: 64 x 20 tiles of SAM mlp1, blocks dealt
        // to the 8 XCDs round robin, every XCD a contiguous run of tiles in groups of 8 row tiles): its A panel and its W panel
        const int bid = blockIdx.x + 256 * t, xcd = bid % 8, loc = bid / 8;
        const int lin = xcd * 160 + loc, in_group = lin % 160;
        const int mt = (lin / 160) * 8 + in_group % 8, nt_ = in_group / 8;
        const size_t tile = (size_t)mt * 20 + nt_;
        const size_t a_off = (size_t)mt * 655360, w_off = (size_t)41943040 + (size_t)nt_ * 655360;
        for (int k = 0; k < ktiles; ++k) {
            unsigned char* dst = smem + (k & 1) * 65536;
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // 4 x 8 KB of A rows, 4 x 8 KB of W rows: 128-byte rows a K-panel stride apart
                const size_t row = (size_t)j * 64 + (tid >> 3);
                if (PANEL) {
                    const size_t o = (size_t)k * 32768 + (size_t)j * 8192 + (size_t)tid * 16;
                    __builtin_amdgcn_global_load_lds(ops_ + a_off + o, (__attribute__((address_space(3))) void*)(dst + j * 8192 + (tid >> 6) * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(ops_ + w_off + o, (__attribute__((address_space(3))) void*)(dst + 32768 + j * 8192 + (tid >> 6) * 1024), 16, 0, 0);
                    continue;
                }
                __builtin_amdgcn_global_load_lds(ops_ + a_off + row * 2560 + (size_t)k * 128 + (tid & 7) * 16,
                                                 (__attribute__((address_space(3))) void*)(dst + j * 8192 + (tid >> 6) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(ops_ + w_off + row * 2560 + (size_t)k * 128 + (tid & 7) * 16,
                                                 (__attribute__((address_space(3))) void*)(dst + 32768 + j * 8192 + (tid >> 6) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
            if (MFMA) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // output: 256 x 256 bf16 = 128 KB as whole lines
        uint4 v = *reinterpret_cast<const uint4*>(smem + tid * 16);
        if (MFMA) v.x ^= __float_as_uint(acc[t & 15][0]);
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>(out + tile * 131072 + (size_t)i * 8192 + tid * 16) = v;
    }
}

struct Ctx {
    float* fout; uint32_t* uout;
    uint4* big; size_t big16;              // 2 GB streaming buffer
    unsigned char* ops_; size_t ops_bytes;  // 55 MB operand set
    unsigned char* out;                     // 1280 tiles x 128 KB
    int mfma_blocks_per_cu;
    int nt;
    bool panel, paced;
};

// one "job" of each kind, sized to ~10 ms alone
static void launch(int kind, const Ctx& c, hipStream_t st) {
    if (kind == 0) {  // MFMA only: 2 blocks of 512 threads per CU (4 waves per SIMD) or 1
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mfma_kernel, dim3(256 * c.mfma_blocks_per_cu), dim3(512), 0, st, c.fout, 4000);
    } else if (kind == 1) {  // streaming read: 600 launches of ~100 MB, like the decode GEMVs
        for (int i = 0; i < 600; ++i) {
            const size_t piece16 = (size_t)100 * 1024 * 1024 / 16, off = ((size_t)i * piece16) % (c.big16 - piece16);
#define GO(V) hipLaunchKernelGGL(stream_kernel<V>, dim3(256), dim3(1024), 0, st, (const uint4*)(c.big + off), piece16, c.uout)
            if (c.nt) GO(1); else GO(0);
#undef GO
        }
    } else {  // GEMM memory side: 60 launches x 1280 tiles (5 per block), 20 K tiles each
        for (int i = 0; i < 60; ++i) {
#define GO(P, M) hipLaunchKernelGGL((gemmio_kernel<P, M>), dim3(256), dim3(512), 131072, st, (const unsigned char*)c.ops_, c.ops_bytes, c.out, 20, 5)
            if (c.panel) { if (c.paced) GO(true, true); else GO(true, false); }
            else { if (c.paced) GO(false, true); else GO(false, false); }
#undef GO
        }
    }
}

static void timed(const Ctx& c, hipStream_t sa, hipStream_t sb, int ka, int kb, float* ta, float* tb, float* wall) {
    hipEvent_t a0, a1, b0, b1, w0;
    CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1)); CHECK(hipEventCreate(&w0));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(w0, 0));
    CHECK(hipStreamWaitEvent(sa, w0, 0));
    CHECK(hipStreamWaitEvent(sb, w0, 0));
    if (ka >= 0) { CHECK(hipEventRecord(a0, sa)); launch(ka, c, sa); CHECK(hipEventRecord(a1, sa)); }
    if (kb >= 0) { CHECK(hipEventRecord(b0, sb)); launch(kb, c, sb); CHECK(hipEventRecord(b1, sb)); }
    CHECK(hipDeviceSynchronize());
    *ta = *tb = 0.f;
    float x = 0.f, y = 0.f;
    if (ka >= 0) { CHECK(hipEventElapsedTime(ta, a0, a1)); CHECK(hipEventElapsedTime(&x, w0, a1)); }
    if (kb >= 0) { CHECK(hipEventElapsedTime(tb, b0, b1)); CHECK(hipEventElapsedTime(&y, w0, b1)); }
    *wall = x > y ? x : y;
}

int main() {
    Ctx c;
    c.big16 = (size_t)2 * 1024 * 1024 * 1024 / 16;
    c.ops_bytes = (size_t)55 * 1024 * 1024;
    CHECK(hipMalloc(&c.fout, 64)); CHECK(hipMalloc(&c.uout, 64));
    CHECK(hipMalloc(&c.big, c.big16 * 16)); CHECK(hipMemset(c.big, 1, c.big16 * 16));
    CHECK(hipMalloc(&c.ops_, c.ops_bytes + 4 * 1024 * 1024)); CHECK(hipMemset(c.ops_, 2, c.ops_bytes + 4 * 1024 * 1024));
    CHECK(hipMalloc(&c.out, (size_t)1280 * 131072));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemmio_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemmio_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemmio_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemmio_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
    const char* names[3] = {"mfma", "stream", "gemmio"};
    const char* pol[2] = {"plain", "non-temporal (the decode GEMV kernels' policy)"};
    for (int cfg = 0; cfg < 4; ++cfg) {  // (cfg 4, 5: the MFMA-paced gemmio - its loop serialises DMA latency and MFMAs, 6 us per K tile: not a model of the real kernel)
        const int per_cu = cfg == 0 ? 2 : 1;
        c.mfma_blocks_per_cu = per_cu;
        c.nt = cfg >= 2;
        c.panel = cfg == 3 || cfg == 5;
        c.paced = cfg >= 4;
        printf("streaming loads: %s; GEMM operands: %s; gemmio %s\n", pol[c.nt], c.panel ? "K-panel layout" : "row major",
               c.paced ? "WITH the 64 MFMAs per wave and K tile of the real kernel" : "without MFMAs");
        float alone[3], ta, tb, wall;
        for (int k = 0; k < 3; ++k) {
            timed(c, sa, sb, k, -1, &ta, &tb, &wall);  // warm
            timed(c, sa, sb, k, -1, &ta, &tb, &wall);
            alone[k] = ta;
        }
        printf("MFMA kernel with %d block(s) of 512 threads per CU (%d waves per SIMD): alone  mfma %.2f ms  stream %.2f ms (%.2f TB/s)  gemmio %.2f ms (%.2f TB/s of DMA + %.2f TB/s of stores)\n",
               per_cu, 2 * per_cu, alone[0], alone[1], 600 * 100.0 * 1.048576e6 / alone[1] / 1e9, alone[2], 60 * 1280 * 20 * 65536.0 / alone[2] / 1e9,
               60 * 1280 * 131072.0 / alone[2] / 1e9);
        const int pairs[3][2] = {{0, 1}, {2, 1}, {0, 2}};
        for (auto& pr : pairs) {
            timed(c, sa, sb, pr[0], pr[1], &ta, &tb, &wall);
            printf("  %s + %s: alone %.2f + %.2f = %.2f ms; together %s %.2f ms, %s %.2f ms, wall %.2f ms\n", names[pr[0]], names[pr[1]], alone[pr[0]], alone[pr[1]],
                   alone[pr[0]] + alone[pr[1]], names[pr[0]], ta, names[pr[1]], tb, wall);
        }
    }
    return 0;
}
