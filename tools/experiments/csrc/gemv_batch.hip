// Batched decode GEMV for gfx950: out[M, N] = act(x[M, K] . W[N, K]^T + bias) + residual for 2 <= M <= 8 fp32 activation rows
// (B sequences per GPU in one decode step: BASELINE.json configs[2]; model/InteractVLM.py:524-531 with B prompts).
//
// The batch-1 kernel's recipe (gemv.hip, gemv1_kernel: 1024-thread blocks, whole weight rows streamed by one wave with 1-KB
// contiguous wave loads, x in LDS, exact bf16 x fp32 products on the VALU) carried over to M rows:
//   * the skinny MFMA kernel (gemv_mfma.hip) feeds 16 weight rows x 64 B per wave instruction (the MFMA fragment map: 16 half
//     lines; 5.0 instead of 6.4 TB/s even as a pure read), re-reads and hi/lo-splits the activations through the vector memory
//     path in every wave, and reaches 2.9 TB/s at M = 8.  M = 8 rows need 64 FMAs per 16-byte weight chunk: on packed fp32 math
//     (v_pk_fma_f32, two per lane per issue) that is ~47 % of the VALU at the full streaming rate - no matrix core needed;
//   * K is walked in chunks of 2048: the block stages x[m][chunk] (x gamma of the fused RMSNorm) as fp32 in LDS - two 16-byte
//     planes per row, M x 16 KB, double-buffered: the global loads of chunk c + 1 are issued before chunk c is consumed and
//     stored to LDS after it, one barrier per chunk;
//   * a wave owns TWO weight rows (the x fragments read from LDS serve both; SwiGLU over the interleaved gate / up rows
//     finishes inside the wave), 4 + 4 non-temporal 16-byte loads in flight per lane, re-issued for the next chunk as soon as a
//     register is consumed;
//   * results: the products are exact (bf16 weight x fp32 activation in fp32), sums in fp32 - the batched step agrees with
//     the batch-1 GEMV to fp32 summation order.
#include "kernels.h"

namespace ivlm {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;
typedef __attribute__((ext_vector_type(2))) float f32x2v_t;

constexpr int kBWaves = 16, kBThreads = 64 * kBWaves;
constexpr int kKC = 2048, kKCH = kKC / 8;  // K chunk in elements / in 16-byte weight chunks (one lane load each)

__device__ __forceinline__ float act_apply_b(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

template <int M, bool RMS>
__global__ __launch_bounds__(kBThreads, 4) void gemvm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[kBWaves][2];
    // x image: [buffer][row m][plane: elements 0-3 / 4-7 of a chunk][chunk] float4
    f32x4v_t* xf = reinterpret_cast<f32x4v_t*>(smem);
    constexpr int kBuf = M * 2 * kKCH;  // float4 per buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = g.K >> 3;
    const int nck = (g.K + kKC - 1) / kKC;
    const float* X = reinterpret_cast<const float*>(g.A);

    // ---- weights: two rows per wave ----
    const int r0 = (blockIdx.x * kBWaves + wave) * 2;
    const bool live0 = r0 < g.N, live1 = r0 + 1 < g.N;
    const u32x4_t* wp0 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)(live0 ? r0 : g.N - 1) * g.ldw);
    const u32x4_t* wp1 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)(live1 ? r0 + 1 : g.N - 1) * g.ldw);
    u32x4_t w0[4], w1[4];
    auto load_w = [&](int u, int c) {  // 16-byte chunk lane + 64 u of K chunk c (clamped: tail lanes re-read the last chunk)
        const int cc = min(c * kKCH + lane + 64 * u, nchunk - 1);
        w0[u] = __builtin_nontemporal_load(wp0 + cc);
        w1[u] = __builtin_nontemporal_load(wp1 + cc);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) load_w(u, 0);

    // ---- activations: thread (m4 = tid >> 8, ch = tid & 255) stages chunk column ch of rows m4 and m4 + 4 ----
    const int m4 = tid >> 8, ch = tid & 255;
    float ssq0 = 0.0f, ssq1 = 0.0f;
    f32x4v_t sa0, sb0, sa1, sb1;  // staged (not yet stored) values
    auto stage_load = [&](int c) {
        const int col = c * kKCH + ch;  // 16-byte-of-weights chunk index = 8 activations
        const bool ok = col < nchunk;
        const f32x4v_t zero = {0.f, 0.f, 0.f, 0.f};
        u32x4_t gv = {0u, 0u, 0u, 0u};
        if (RMS && ok) gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + col);
        auto one = [&](int m, f32x4v_t& xa, f32x4v_t& xb, float& ssq) {
            xa = zero;
            xb = zero;
            if (ok && m < M) {
                const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(X + (int64_t)m * g.lda) + 2 * col;
                xa = xp[0];
                xb = xp[1];
                if (RMS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ssq += xa[j] * xa[j] + xb[j] * xb[j];
                    xa[0] *= __uint_as_float(gv[0] << 16); xa[1] *= __uint_as_float(gv[0] & 0xffff0000u);
                    xa[2] *= __uint_as_float(gv[1] << 16); xa[3] *= __uint_as_float(gv[1] & 0xffff0000u);
                    xb[0] *= __uint_as_float(gv[2] << 16); xb[1] *= __uint_as_float(gv[2] & 0xffff0000u);
                    xb[2] *= __uint_as_float(gv[3] << 16); xb[3] *= __uint_as_float(gv[3] & 0xffff0000u);
                }
            }
        };
        one(m4, sa0, sb0, ssq0);
        if (M > 4) one(m4 + 4, sa1, sb1, ssq1);
    };
    auto stage_store = [&](int buf) {
        f32x4v_t* b = xf + buf * kBuf;
        if (m4 < M) {
            b[(m4 * 2) * kKCH + ch] = sa0;
            b[(m4 * 2 + 1) * kKCH + ch] = sb0;
        }
        if (M > 4 && m4 + 4 < M) {
            b[((m4 + 4) * 2) * kKCH + ch] = sa1;
            b[((m4 + 4) * 2 + 1) * kKCH + ch] = sb1;
        }
    };

    f32x2v_t acc0[M], acc1[M];  // (even, odd) partial sums of rows r0 / r0 + 1 against activation row m
#pragma unroll
    for (int m = 0; m < M; ++m) acc0[m] = acc1[m] = f32x2v_t{0.f, 0.f};

    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int c = 0; c < nck; ++c) {
        const bool more = c + 1 < nck;
        if (more) stage_load(c + 1);  // global loads now, LDS stores after this chunk's arithmetic
        const f32x4v_t* b = xf + (c & 1) * kBuf;
        const int nch = min(kKCH, nchunk - c * kKCH);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = lane + 64 * u;
            if (cc < nch) {
                f32x2v_t wl0[4], wl1[4];  // weights as fp32 pairs (k, k + 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wl0[j] = f32x2v_t{__uint_as_float(w0[u][j] << 16), __uint_as_float(w0[u][j] & 0xffff0000u)};
                    wl1[j] = f32x2v_t{__uint_as_float(w1[u][j] << 16), __uint_as_float(w1[u][j] & 0xffff0000u)};
                }
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const f32x4v_t xa = b[(m * 2) * kKCH + cc], xb = b[(m * 2 + 1) * kKCH + cc];
                    const f32x2v_t x01 = {xa[0], xa[1]}, x23 = {xa[2], xa[3]}, x45 = {xb[0], xb[1]}, x67 = {xb[2], xb[3]};
                    acc0[m] = __builtin_elementwise_fma(wl0[0], x01, acc0[m]);
                    acc0[m] = __builtin_elementwise_fma(wl0[1], x23, acc0[m]);
                    acc0[m] = __builtin_elementwise_fma(wl0[2], x45, acc0[m]);
                    acc0[m] = __builtin_elementwise_fma(wl0[3], x67, acc0[m]);
                    acc1[m] = __builtin_elementwise_fma(wl1[0], x01, acc1[m]);
                    acc1[m] = __builtin_elementwise_fma(wl1[1], x23, acc1[m]);
                    acc1[m] = __builtin_elementwise_fma(wl1[2], x45, acc1[m]);
                    acc1[m] = __builtin_elementwise_fma(wl1[3], x67, acc1[m]);
                }
            }
            if (more) load_w(u, c + 1);  // the registers are free: next chunk's weights on their way
        }
        if (more) stage_store((c + 1) & 1);
        if (RMS && !more) {  // all of K has been staged: publish this wave's sum(x^2) (rows m4 / m4 + 4)
            ssq0 = wave_sum(ssq0);
            ssq1 = wave_sum(ssq1);
            if (lane == 0) {
                s_red[wave][0] = ssq0;
                s_red[wave][1] = ssq1;
            }
        }
        __syncthreads();
    }

    // ---- reduce, epilogue: lane 0 of the wave writes M x 2 outputs ----
    float v0[M], v1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        v0[m] = wave_sum(acc0[m][0] + acc0[m][1]);
        v1[m] = wave_sum(acc1[m][0] + acc1[m][1]);
    }
    if (lane != 0 || !live0) return;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float a = v0[m], b = v1[m];
        if (RMS) {  // rows m & 3 were summed by waves 4 (m & 3) .. 4 (m & 3) + 3, slot m >> 2
            const int wq = (m & 3) * 4, sl = m >> 2;
            const float q = s_red[wq][sl] + s_red[wq + 1][sl] + s_red[wq + 2][sl] + s_red[wq + 3][sl];
            const float rstd = rsqrtf(q / (float)g.K + g.rms_eps);
            a *= rstd;
            b *= rstd;
        }
        if (g.bias) {
            a += bf16_to_f32(g.bias[r0]);
            if (live1) b += bf16_to_f32(g.bias[r0 + 1]);
        }
        if (g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved
            const float o = (a / (1.0f + __expf(-a))) * b;
            const int64_t idx = (int64_t)m * g.ldc + (r0 >> 1);
            if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
            else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
            continue;
        }
        a = act_apply_b(a, g.act);
        b = act_apply_b(b, g.act);
        if (g.residual) {
            const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
            if (g.res_f32) {
                const float* R = reinterpret_cast<const float*>(g.residual) + rrow * g.ldr;
                a += R[r0];
                if (live1) b += R[r0 + 1];
            } else {
                const bf16_t* R = g.residual + rrow * g.ldr;
                a += bf16_to_f32(R[r0]);
                if (live1) b += bf16_to_f32(R[r0 + 1]);
            }
        }
        if (g.out_f32) {
            float* C = static_cast<float*>(g.C) + (int64_t)m * g.ldc;
            C[r0] = a;
            if (live1) C[r0 + 1] = b;
        } else {
            bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc;
            C[r0] = f32_to_bf16(a);
            if (live1) C[r0 + 1] = f32_to_bf16(b);
        }
    }
}

template <int M>
int launch_gemvm(const GemmArgs& g, hipStream_t st) {
    const size_t lds = (size_t)2 * M * 2 * kKCH * 16;  // two buffers of M rows x 16 KB
    const int blocks = (g.N + 2 * kBWaves - 1) / (2 * kBWaves);
#define IVLM_GO(RMS)                                                                                              \
    do {                                                                                                          \
        auto kfn = gemvm_kernel<M, RMS>;                                                                          \
        static bool set = false;                                                                                  \
        if (!set) {                                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      128 * 1024);                                                                \
            set = true;                                                                                           \
        }                                                                                                         \
        ivlm_launch(kfn, dim3(blocks), dim3(kBThreads), lds, st, g);                                              \
    } while (0)
    if (g.rms_w) IVLM_GO(true); else IVLM_GO(false);
#undef IVLM_GO
    return ivlm_launch_status();
}

}  // namespace

bool gemv_batch_applies(const GemmArgs& g) {
    return g.a_f32 && g.M >= 2 && g.M <= 8 && g.batch == 1 && !(g.K & 7) && !(g.lda & 3) && !(g.ldw & 7) && g.N >= 256 &&
           !(g.act == ACT_SWIGLU && ((g.N & 1) || g.residual)) && !g.out_rows && !g.a_rows;
}

int gemv_batch_f32(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || !gemv_batch_applies(g)) return IVLM_ERR_INVALID_ARG;
    switch (g.M) {
        case 2: return launch_gemvm<2>(g, st);
        case 3: return launch_gemvm<3>(g, st);
        case 4: return launch_gemvm<4>(g, st);
        case 5: return launch_gemvm<5>(g, st);
        case 6: return launch_gemvm<6>(g, st);
        case 7: return launch_gemvm<7>(g, st);
        default: return launch_gemvm<8>(g, st);
    }
}

}  // namespace ivlm
