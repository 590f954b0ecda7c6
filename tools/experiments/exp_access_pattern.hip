// Experiment (not part of the library): HBM streaming rate of a 100 MB bf16 matrix under the two lane->address maps of
// the decode kernels.
//   rows16 : the skinny MFMA map (gemv_mfma.hip): lane (r = l & 15, kg = l >> 4) reads 16 bytes of row r at chunk 4 s + kg:
//            one wave instruction touches 16 rows x 64 bytes (16 half cache lines);
//   row1   : the GEMV map (gemv.hip): 64 lanes read 1 KB contiguous of one row (8 full lines);
//   rows16_full : 16 rows, but the 4 lanes of a row read chunks {kg, kg + 4, ...}: 2 instructions cover full lines
// build: hipcc --offload-arch=gfx950 -O3 exp_access_pattern.hip -o exp_access_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int U>
__global__ __launch_bounds__(512, 2) void rows16(const u32x4_t* W, int64_t ldw16, int K, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kg = lane >> 4;
    const int nsteps = K / 32, per = nsteps / 8, s0 = wave * per;
    const u32x4_t* wp = W + (int64_t)(blockIdx.x * 16 + r) * ldw16;
    unsigned acc = 0;
    for (int s = s0; s < s0 + per; s += U) {
        u32x4_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + min((s + u) * 4 + kg, K / 8 - 1));
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void row1(const u32x4_t* W, int64_t ldw16, int K, int N, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const u32x4_t* wp = W + (int64_t)row * ldw16;
    const int nch = K / 8;
    unsigned acc = 0;
    for (int c = lane; c < nch; c += 64 * U) {
        u32x4_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + min(c + 64 * u, nch - 1));
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// 16 rows per wave, 4 lanes per row, each lane reads 2 x 16 bytes that are 64 bytes apart -> an instruction still touches 16 lines
// but a wave owns a contiguous [per x 64 B] slice per row; variant: 8 waves interleave at 256-byte granularity
template <int U>
__global__ __launch_bounds__(512, 2) void rows16_interleaved(const u32x4_t* W, int64_t ldw16, int K, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kg = lane >> 4;
    const int nsteps = K / 32;
    const u32x4_t* wp = W + (int64_t)(blockIdx.x * 16 + r) * ldw16;
    unsigned acc = 0;
    for (int s = wave * U; s < nsteps; s += 8 * U) {  // wave w takes steps [w U, w U + U) of every 8 U
        u32x4_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + min((s + u) * 4 + kg, K / 8 - 1));
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
float time_it(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int nbuf = 6;  // rotate buffers so L2 / MALL never hold the matrix
    struct Shape { int N, K; } shapes[] = {{12288, 4096}, {4096, 4096}, {22016, 4096}, {4096, 11008}};
    unsigned* out;
    hipMalloc(&out, 4);
    for (auto sh : shapes) {
        const size_t bytes = (size_t)sh.N * sh.K * 2;
        u32x4_t* W[nbuf];
        for (int i = 0; i < nbuf; ++i) { hipMalloc(&W[i], bytes); hipMemset(W[i], 1, bytes); }
        int it = 0;
        const int64_t ld = sh.K / 8;
        auto rate = [&](float ms) { return bytes / ms / 1e9; };
        float a4 = time_it([&] { rows16<4><<<sh.N / 16, 512>>>(W[it++ % nbuf], ld, sh.K, out); }, 30);
        float a8 = time_it([&] { rows16<8><<<sh.N / 16, 512>>>(W[it++ % nbuf], ld, sh.K, out); }, 30);
        float i4 = time_it([&] { rows16_interleaved<4><<<sh.N / 16, 512>>>(W[it++ % nbuf], ld, sh.K, out); }, 30);
        float b4 = time_it([&] { row1<4><<<sh.N / 4, 256>>>(W[it++ % nbuf], ld, sh.K, sh.N, out); }, 30);
        float b8 = time_it([&] { row1<8><<<sh.N / 4, 256>>>(W[it++ % nbuf], ld, sh.K, sh.N, out); }, 30);
        printf("N %5d K %5d (%.0f MB): rows16 U4 %.2f U8 %.2f  interleaved U4 %.2f | row1 U4 %.2f U8 %.2f TB/s\n", sh.N, sh.K,
               bytes / 1e6, rate(a4), rate(a8), rate(i4), rate(b4), rate(b8));
        for (int i = 0; i < nbuf; ++i) hipFree(W[i]);
    }
    return 0;
}
