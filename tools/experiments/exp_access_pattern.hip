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
#include "ivlm_hip.h"  // the library GEMV under the same conditions (link with -livlm_hip)
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

// row1 + the GEMV's work: fp32 x staged in LDS per block (two float4 planes), exact products, wave reduce, store.
// R rows per wave, one after the other; no software pipeline - occupancy (LDS: 4 K bytes per block) hides the row boundaries.
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;
template <int U, int R>
__global__ __launch_bounds__(256) void row1_dot(const u32x4_t* W, int64_t ldw16, int K, int N, const float* x, float* y) {
    extern __shared__ f32x4v_t xf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = K / 8;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    u32x4_t w[U];
    if (row0 < N) {
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(W + (int64_t)row0 * ldw16 + min(lane + 64 * u, nch - 1));
    }
    for (int c = threadIdx.x; c < nch; c += 256) {
        const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(x) + 2 * c;
        xf[c] = xp[0];
        xf[nch + c] = xp[1];
    }
    __syncthreads();
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= N) return;
        const u32x4_t* wp = W + (int64_t)row * ldw16;
        float acc = 0.0f;
        for (int c = lane; c < nch; c += 64 * U) {
            if (c != lane || r != 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + min(c + 64 * u, nch - 1));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = c + 64 * u;
                if (cc < nch) {
                    const f32x4v_t xa = xf[cc], xb = xf[nch + cc];
                    acc = fmaf(__uint_as_float(w[u][0] << 16), xa[0], acc);
                    acc = fmaf(__uint_as_float(w[u][0] & 0xffff0000u), xa[1], acc);
                    acc = fmaf(__uint_as_float(w[u][1] << 16), xa[2], acc);
                    acc = fmaf(__uint_as_float(w[u][1] & 0xffff0000u), xa[3], acc);
                    acc = fmaf(__uint_as_float(w[u][2] << 16), xb[0], acc);
                    acc = fmaf(__uint_as_float(w[u][2] & 0xffff0000u), xb[1], acc);
                    acc = fmaf(__uint_as_float(w[u][3] << 16), xb[2], acc);
                    acc = fmaf(__uint_as_float(w[u][3] & 0xffff0000u), xb[3], acc);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) y[row] = acc;
    }
}

// v2: WAVES waves per block, each wave takes ROWS rows AT ONCE (x read from LDS once for all of them), single pass, no
// persistence: the hardware dispatcher balances, occupancy hides latency
template <int U, int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rowsN_dot(const u32x4_t* W, int64_t ldw16, int K, int N, const float* x, float* y) {
    extern __shared__ f32x4v_t xf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = K / 8;
    const int row0 = (blockIdx.x * WAVES + wave) * ROWS;
    const u32x4_t* wp[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wp[r] = W + (int64_t)min(row0 + r, N - 1) * ldw16;
    u32x4_t w[ROWS][U];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int u = 0; u < U; ++u) w[r][u] = __builtin_nontemporal_load(wp[r] + min(lane + 64 * u, nch - 1));
    for (int c = threadIdx.x; c < nch; c += 64 * WAVES) {
        const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(x) + 2 * c;
        xf[c] = xp[0];
        xf[nch + c] = xp[1];
    }
    __syncthreads();
    if (row0 >= N) return;
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.0f;
    for (int c = lane; c < nch; c += 64 * U) {
        if (c != lane) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (c + 64 * u < nch) w[r][u] = __builtin_nontemporal_load(wp[r] + c + 64 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + 64 * u;
            if (cc < nch) {
                const f32x4v_t xa = xf[cc], xb = xf[nch + cc];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    float a = acc[r];
                    a = fmaf(__uint_as_float(w[r][u][0] << 16), xa[0], a);
                    a = fmaf(__uint_as_float(w[r][u][0] & 0xffff0000u), xa[1], a);
                    a = fmaf(__uint_as_float(w[r][u][1] << 16), xa[2], a);
                    a = fmaf(__uint_as_float(w[r][u][1] & 0xffff0000u), xa[3], a);
                    a = fmaf(__uint_as_float(w[r][u][2] << 16), xb[0], a);
                    a = fmaf(__uint_as_float(w[r][u][2] & 0xffff0000u), xb[1], a);
                    a = fmaf(__uint_as_float(w[r][u][3] << 16), xb[2], a);
                    a = fmaf(__uint_as_float(w[r][u][3] & 0xffff0000u), xb[3], a);
                    acc[r] = a;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float a = acc[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0 && row0 + r < N) y[row0 + r] = a;
    }
}

template <int U, int ROWS, int WAVES>
float run_rowsN(const u32x4_t* W, int64_t ld, int K, int N, const float* x, float* y, size_t lds) {
    auto k = rowsN_dot<U, ROWS, WAVES>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const int per = ROWS * WAVES;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<<<(N + per - 1) / per, 64 * WAVES, lds>>>(W, ld, K, N, x, y);
    hipDeviceSynchronize();
    return 0.f;
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
        // the product GEMV (fp32 x, fp32 out; RMS prologue on the 12288 / 22016 shapes, SwiGLU on 22016, fp32 residual else)
        float *x, *y, *res; void* gam;
        hipMalloc(&x, sh.K * 4); hipMemset(x, 0, sh.K * 4);
        hipMalloc(&y, sh.N * 4); hipMalloc(&res, sh.N * 4); hipMemset(res, 0, sh.N * 4);
        hipMalloc(&gam, sh.K * 2); hipMemset(gam, 0, sh.K * 2);
        const bool rms = sh.N > 4096, swiglu = sh.N == 22016;
        for (int ks = 1; ks <= 2; ++ks) {
        ivlm_gemv1_tuning(ks);
        float gv = time_it([&] {
            ivlm_gemm_bf16(x, sh.K, W[it++ % nbuf], sh.K, y, swiglu ? sh.N / 2 : sh.N, nullptr, rms ? nullptr : res, sh.N, 0, 1, sh.N,
                           sh.K, swiglu ? 5 : 0, 1, 1, 0, 0, 0, 0, rms ? gam : nullptr, 1e-5f, rms ? 1 : 3, nullptr, nullptr, nullptr);
        }, 30);
        printf("    library GEMV (waves per row %d): %.2f TB/s (%.1f us)\n", ks, rate(gv), gv * 1e3);
        }
        ivlm_gemv1_tuning(0);
        {
            const size_t lds = (size_t)sh.K * 4;
            hipFuncSetAttribute(reinterpret_cast<const void*>(row1_dot<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(row1_dot<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(row1_dot<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(row1_dot<8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(row1_dot<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            float d1 = time_it([&] { row1_dot<8, 1><<<(sh.N + 3) / 4, 256, lds>>>(W[it++ % nbuf], ld, sh.K, sh.N, x, y); }, 30);
            float d2 = time_it([&] { row1_dot<8, 2><<<(sh.N + 7) / 8, 256, lds>>>(W[it++ % nbuf], ld, sh.K, sh.N, x, y); }, 30);
            float d4 = time_it([&] { row1_dot<8, 4><<<(sh.N + 15) / 16, 256, lds>>>(W[it++ % nbuf], ld, sh.K, sh.N, x, y); }, 30);
            float d8 = time_it([&] { row1_dot<8, 8><<<(sh.N + 31) / 32, 256, lds>>>(W[it++ % nbuf], ld, sh.K, sh.N, x, y); }, 30);
            float e4 = time_it([&] { row1_dot<4, 4><<<(sh.N + 15) / 16, 256, lds>>>(W[it++ % nbuf], ld, sh.K, sh.N, x, y); }, 30);
#define RUN(U, R, WV) { auto k = rowsN_dot<U, R, WV>; \
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
            float t = time_it([&] { k<<<(sh.N + R * WV - 1) / (R * WV), 64 * WV, lds>>>(W[it++ % nbuf], ld, sh.K, sh.N, x, y); }, 30); \
            printf(" U%d R%d W%d %.2f", U, R, WV, rate(t)); }
            printf("    rowsN_dot:");
            RUN(8, 1, 4) RUN(8, 1, 8) RUN(8, 1, 16) RUN(4, 2, 4) RUN(4, 2, 8) RUN(4, 2, 16) RUN(8, 2, 8) RUN(8, 2, 16) RUN(4, 4, 8) RUN(4, 4, 16) RUN(2, 4, 16)
            printf(" TB/s\n");
            printf("    row1_dot (x in LDS, dot, reduce) U8: R1 %.2f R2 %.2f R4 %.2f R8 %.2f | U4 R4 %.2f TB/s\n", rate(d1), rate(d2), rate(d4),
                   rate(d8), rate(e4));
        }
        printf("N %5d K %5d (%.0f MB): rows16 U4 %.2f U8 %.2f  interleaved U4 %.2f | row1 U4 %.2f U8 %.2f TB/s\n", sh.N, sh.K,
               bytes / 1e6, rate(a4), rate(a8), rate(i4), rate(b4), rate(b8));
        for (int i = 0; i < nbuf; ++i) hipFree(W[i]);
    }
    return 0;
}
