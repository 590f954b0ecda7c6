// Skinny GEMM on the matrix cores for gfx950: out[M <= 16, N] = act(x[M,K] . W[N,K]^T + bias) + residual.
//
// The batched decode step (B sequences per GPU, BASELINE.json configs[2]; model/InteractVLM.py:524-531 with B prompts)
// is still pure weight streaming - every weight byte is read once per step - but each 16-byte weight chunk now meets
// B activation rows.  The wave-per-row GEMV (gemv.hip) keeps the activations in LDS (B x K x 2 bytes: 176 KB for the
// down projection at B = 8 - more than a CU has) and spends B x 8 dot instructions per chunk; here one
// v_mfma_f32_16x16x32_bf16 consumes 16 weight rows x 32 k against up to 16 activation rows, and nothing is staged:
//
//   * one block = T tiles of 16 weight rows, 8 waves; wave w owns the k-steps [w*per, (w+1)*per) of those rows (split-K
//     inside the block), so N/(16 T) blocks x 8 waves stream the matrix (T = 1, 256 blocks for N = 4096; T = 3 for the
//     fused qkv and the gate-up rows: one to two blocks per CU, each activation fragment loaded and split once for T tiles);
//   * W fragment: lane (n = l & 15, kg = l >> 4) loads 16 bytes of row n at k = 32*step + 8*kg straight from HBM
//     (non-temporal; 4 lanes cover 64 contiguous bytes, consecutive steps continue the row), 8 steps in flight per lane;
//   * x fragment: lane (m = l & 15, kg) loads the same k of activation row m (L2 resident: M x K x 2 bytes), zero for
//     m >= M; the optional RMSNorm prologue of the decode path (x * gamma rounded to bf16, row scale applied to the
//     accumulator) is fused exactly as in gemv.hip;
//   * the 8 partial 16x16 tiles are summed through LDS in a fixed order (deterministic), then bias / activation /
//     SwiGLU over interleaved gate-up rows / residual and the store.
//   * AF32: fp32 activation rows (fp32 residual stream): each 8-element x fragment is split into a bf16 "hi" and a bf16 "lo"
//     part (x = hi + lo to 2^-17) and meets the weight fragment in TWO MFMAs - the kernel is HBM-bound, the second MFMA is
//     free, and the batched decode step then agrees with the batch-1 GEMV (exact fp32 products) to ~1e-5 relative.
#include "gemm_common.h"

namespace ivlm {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int kWaves = 8, kThreads = kWaves * 64;

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return x < 0.0f ? 0.0f : x;  // (torch.relu semantics: a NaN stays a NaN; fmaxf would turn it into 0)
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

// fp32 pair -> packed bf16 (hardware RNE) and the packed bf16 of the remainders: x = hi + lo to 2^-17
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    lo = pack_bf16x2(ra, rb);
}

// T = 16-row weight tiles per block.  Every block re-reads the M x K activation rows (L2 hits, but they share the CU's
// load path with the weight stream: at M = 8 fp32 rows the activations of a one-tile block are as many bytes as its
// weights), so wide matrices give each block several tiles and load / split each activation fragment once for all of them.
template <bool RMS, bool AF32, int T>
__global__ __launch_bounds__(kThreads, 2) void skinny_mfma_kernel(GemmArgs g) {
    __shared__ float s_part[kWaves][16][17];  // [wave][m][n]
    __shared__ float s_ssq[kWaves][16];
    __shared__ float s_fin[16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * (16 * T);
    const int nchunk = g.K >> 3;           // 16-byte chunks per row
    const int nsteps = (nchunk + 3) >> 2;  // MFMA k-steps (32 elements)
    const int per = (nsteps + kWaves - 1) / kWaves;
    const int s0 = wave * per, s1 = min(s0 + per, nsteps);
    const u32x4_t* wp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) wp[t] = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)min(n0 + 16 * t + r, g.N - 1) * g.ldw);
    const bool xrow = r < g.M;
    const u32x4_t* xp = reinterpret_cast<const u32x4_t*>(g.A + (int64_t)(xrow ? r : 0) * g.lda * (AF32 ? 2 : 1));
    const u32x4_t* gp = reinterpret_cast<const u32x4_t*>(g.rms_w);
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    f32x4_t acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    float ssq = 0.0f;
    // k-steps in flight per lane: fp32 rows with one tile keep two blocks per CU resident at 4; several tiles already
    // carry T weight loads per step
    constexpr int kU = T > 2 ? 2 : (AF32 || T > 1) ? 4 : 8;
    for (int s = s0; s < s1; s += kU) {
        u32x4_t w[kU][T], x[kU], x2[kU], gm[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int c = (s + u) * 4 + kg;
            const bool ok = (s + u) < s1 && c < nchunk;
            const int cc = ok ? c : 0;  // clamped (unconditional) weight loads keep the buffers in registers
#pragma unroll
            for (int t = 0; t < T; ++t) w[u][t] = __builtin_nontemporal_load(wp[t] + cc);
            if (AF32) {  // 8 fp32 activations = two 16-byte loads
                x[u] = (ok && xrow) ? xp[2 * cc] : zero;
                x2[u] = (ok && xrow) ? xp[2 * cc + 1] : zero;
            } else {
                x[u] = (ok && xrow) ? xp[cc] : zero;
            }
            if (RMS) gm[u] = gp[cc];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (AF32) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[j] = __uint_as_float(x[u][j]);
                    f[4 + j] = __uint_as_float(x2[u][j]);
                }
                if (RMS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ssq += f[2 * j] * f[2 * j] + f[2 * j + 1] * f[2 * j + 1];
                        f[2 * j] *= __uint_as_float(gm[u][j] << 16);
                        f[2 * j + 1] *= __uint_as_float(gm[u][j] & 0xffff0000u);
                    }
                }
                u32x4_t hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t h, l;
                    split_pair(f[2 * j], f[2 * j + 1], h, l);
                    hi[j] = h;
                    lo[j] = l;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[u][t]),
                                                                     __builtin_bit_cast(bf16x8_t, hi), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[u][t]),
                                                                     __builtin_bit_cast(bf16x8_t, lo), acc[t], 0, 0, 0);
                }
                continue;
            }
            u32x4_t xv = x[u];
            if (RMS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = __uint_as_float(xv[j] << 16), hi = __uint_as_float(xv[j] & 0xffff0000u);
                    ssq += lo * lo + hi * hi;
                    xv[j] = pack_bf16x2(lo * __uint_as_float(gm[u][j] << 16), hi * __uint_as_float(gm[u][j] & 0xffff0000u));
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[u][t]),
                                                                 __builtin_bit_cast(bf16x8_t, xv), acc[t], 0, 0, 0);
        }
    }
    // lane holds C[m = r][n = 4*kg + i] of this wave's K slice, per tile
    if (RMS) {
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);
        if (kg == 0) s_ssq[wave][r] = ssq;
    }
    const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int nt = n0 + 16 * t;
        if (t > 0) __syncthreads();  // the previous tile's s_fin readers are done
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wave][r][kg * 4 + i] = acc[t][i];
        __syncthreads();
        if (threadIdx.x < 256) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) v += s_part[w][m][n];
            if (RMS) {
                float q = 0.0f;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) q += s_ssq[w][m];
                v *= rsqrtf(q / (float)g.K + g.rms_eps);
            }
            const int nn = min(nt + n, g.N - 1);
            s_fin[m][n] = v + (g.bias ? bf16_to_f32(g.bias[nn]) : 0.0f);
        }
        __syncthreads();
        if (threadIdx.x >= 256 || m >= g.M || nt + n >= g.N) continue;
        float v = s_fin[m][n];
        int64_t col = nt + n;
        if (g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved: even n pairs with n + 1
            if (n & 1) continue;
            const float up = s_fin[m][n + 1];
            v = (v / (1.0f + __expf(-v))) * up;
            col >>= 1;
        } else {
            v = act_apply(v, g.act);
            if (g.residual) {
                const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
                v += gemm_residual_at(g, g.residual, rrow * g.ldr + col);
            }
        }
        if (g.out_f32) static_cast<float*>(g.C)[(int64_t)m * g.ldc + col] = v;
        else static_cast<bf16_t*>(g.C)[(int64_t)m * g.ldc + col] = f32_to_bf16(v);
    }
}

template <int T>
void skinny_launch(const GemmArgs& g, int blocks, hipStream_t st) {
    if (g.a_f32) {
        if (g.rms_w) ivlm_launch(skinny_mfma_kernel<true, true, T>, dim3(blocks), dim3(kThreads), 0, st, g);
        else ivlm_launch(skinny_mfma_kernel<false, true, T>, dim3(blocks), dim3(kThreads), 0, st, g);
    } else {
        if (g.rms_w) ivlm_launch(skinny_mfma_kernel<true, false, T>, dim3(blocks), dim3(kThreads), 0, st, g);
        else ivlm_launch(skinny_mfma_kernel<false, false, T>, dim3(blocks), dim3(kThreads), 0, st, g);
    }
}

int g_skinny_tiles = 0;  // 0 = rule below; tools/bench_decode.py sweeps it

// tiles per block: fewest (blocks per CU) x (tiles per block), then the most tiles (least activation re-reading)
int skinny_tiles(int N) {
    if (g_skinny_tiles >= 10) return N > 16384 ? g_skinny_tiles % 10 : (N > 8192 ? g_skinny_tiles / 10 : 1);  // sweep hook: 10 q + g
    if (g_skinny_tiles > 0) return g_skinny_tiles;
    const int tiles = (N + 15) / 16, cus = 256;
    int best = 1, best_cost = 1 << 30;
    for (int t : {1, 2, 3}) {  // (4 and 6 tiles per block measured slower: gate|up 230 blocks of 6 tiles 4.56 ms per step, 459 of 3: 4.44)
        const int blocks = (tiles + t - 1) / t;
        const int cost = ((blocks + cus - 1) / cus) * t;
        if (cost <= best_cost) { best = t; best_cost = cost; }
    }
    return best;
}

}  // namespace

int gemv_mfma_bf16(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.M > 16 || g.N <= 0 || g.K <= 0) return IVLM_ERR_INVALID_ARG;
    if ((g.K & 7) || (g.lda & (g.a_f32 ? 3 : 7)) || (g.ldw & 7) || g.batch != 1) return IVLM_ERR_UNSUPPORTED;
    if (g.act == ACT_SWIGLU && ((g.N & 1) || g.residual)) return IVLM_ERR_UNSUPPORTED;
    const int T = skinny_tiles(g.N);
    const int blocks = (g.N + 16 * T - 1) / (16 * T);
    switch (T) {
        case 1: skinny_launch<1>(g, blocks, st); break;
        case 2: skinny_launch<2>(g, blocks, st); break;
        case 3: skinny_launch<3>(g, blocks, st); break;
        case 4: skinny_launch<4>(g, blocks, st); break;
        default: skinny_launch<6>(g, blocks, st); break;
    }
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" void ivlm_skinny_tuning(int tiles_per_block) { ivlm::g_skinny_tiles = tiles_per_block; }
