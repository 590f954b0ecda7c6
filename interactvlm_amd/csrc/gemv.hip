// Skinny GEMM / GEMV for gfx950: out[M<=8, N] = act(x[M,K] . W[N,K]^T + bias) + residual.
//
// Batch-1 autoregressive decode (HF greedy search driven by InteractVLM.evaluate, model/InteractVLM.py:524-531)
// is pure weight streaming: 13.5 GB of bf16 weights per generated token for LLaMA-7B.  No LDS round trip
// (nothing is shared between waves), no MFMA (M is 1): each wave owns two weight rows at a time, streams
// them with non-temporal 16-byte loads (8 in flight per lane), dots them against the L1-resident
// activation rows and reduces with a wave butterfly.  Two rows per wave also lets the SwiGLU epilogue
// (row-interleaved gate/up weights) complete inside the wave.
#include <stdlib.h>

#include <algorithm>

#include "kernels.h"

namespace ivlm {
namespace {

constexpr int kMaxM = 8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // native vector: nontemporal-loadable

__device__ __forceinline__ float dot8(const u32x4_t& w, const u32x4_t& x) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc += __uint_as_float(w[j] << 16) * __uint_as_float(x[j] << 16);
        acc += __uint_as_float(w[j] & 0xffff0000u) * __uint_as_float(x[j] & 0xffff0000u);
    }
    return acc;
}

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

// ROWS weight rows per wave (2: enables the SwiGLU epilogue and doubles the loads in flight; 1: small N, more
// waves).  XLDS: the activation rows (optionally RMS-normalised: x * gamma, bf16) are staged ONCE per block in LDS
// together with their row scale, instead of every wave re-deriving them from global memory for every weight row.
// Intra-launch producer / consumer hooks (gemv_pair_kernel below): a consumer block issues its first weight loads, then
// waits until *wait_ctr >= wait_target before it reads its input vector (with agent-scope loads); a producer block
// writes its outputs with agent-scope stores and bumps *arrive_ctr once when done.
struct GemvSync {
    const int32_t* wait_ctr = nullptr;
    int wait_target = 0;
    int32_t* arrive_ctr = nullptr;
    int32_t* status = nullptr;
};
constexpr long long kGemvWaitTicks = 100000000LL;  // 1 s of the 100 MHz wall clock

template <int M, int ROWS, bool RMS, bool XLDS, bool COH_X = false, bool COH_OUT = false>
__device__ __forceinline__ void gemv_body(const GemmArgs& g, const int vblock, const int vgrid, const GemvSync sy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[4][M];
    __shared__ float s_rstd[M];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = g.K >> 3;  // 16-byte chunks per row
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);  // [M][nchunk] when XLDS

    constexpr int U = ROWS == 2 ? 4 : 8;  // 8 x 16-byte weight loads in flight per lane per step
    const int wave_global = vblock * 4 + wave;
    const int nwaves = vgrid * 4;
    const int ngroups = (g.N + ROWS - 1) / ROWS;
    const int nbatch = (nchunk + 64 * U - 1) / (64 * U);
    const int my_groups = wave_global < ngroups ? (ngroups - wave_global + nwaves - 1) / nwaves : 0;
    const int nsteps = my_groups * nbatch;

    auto issue = [&](u32x4_t (&wa)[U], u32x4_t (&wb)[U], int step) __attribute__((always_inline)) {
        const int pr = wave_global + (step / nbatch) * nwaves;
        const int n0 = ROWS * pr, n1 = (ROWS == 2 && n0 + 1 < g.N) ? n0 + 1 : n0;
        const u32x4_t* w0 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)n0 * g.ldw);
        const u32x4_t* w1 = reinterpret_cast<const u32x4_t*>(g.W + (int64_t)n1 * g.ldw);
        const int c0 = (step % nbatch) * (64 * U) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int cc = c0 + u * 64;
            cc = cc < nchunk ? cc : nchunk - 1;  // clamped (unconditional) loads keep the buffers in registers
            wa[u] = __builtin_nontemporal_load(w0 + cc);
            if (ROWS == 2) wb[u] = __builtin_nontemporal_load(w1 + cc);
        }
    };

    // ---- block prologue: stage x (x*gamma) in LDS, reduce sum(x^2) -----------------------------------
    auto prologue = [&]() __attribute__((always_inline)) {
        float ssq[M];
#pragma unroll
        for (int m = 0; m < M; ++m) ssq[m] = 0.0f;
        if (COH_X) {
            // x was produced by other blocks of this launch: agent-scope loads.  ALL of this thread's loads are issued
            // before the first is used (a load-use-per-iteration loop costs one memory round trip per iteration: 5-6
            // round trips for the 22 KB vector of down_proj).  XLDS, M == 1, no RMS fusion on this path.
            constexpr int kMaxDw = 48;  // <= 48 KB of x / (256 threads x 4 B); lane-consecutive dwords: fully coalesced
            uint32_t v[kMaxDw];
            const uint32_t* xp = reinterpret_cast<const uint32_t*>(g.A);
            uint32_t* xs32 = reinterpret_cast<uint32_t*>(xs);
            const int ndw = g.K >> 1;
            const int nper = (ndw + 255) >> 8;  // dwords per thread (22 for K = 11008)
#pragma unroll
            for (int i = 0; i < kMaxDw; ++i)
                if (i < nper) {
                    const int d = min((int)threadIdx.x + 256 * i, ndw - 1);
                    v[i] = __hip_atomic_load(const_cast<uint32_t*>(xp) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
            for (int i = 0; i < kMaxDw; ++i)
                if (i < nper) {
                    const int d = threadIdx.x + 256 * i;
                    if (d < ndw) xs32[d] = v[i];
                }
            __syncthreads();
            return;
        }
        for (int c = threadIdx.x; c < nchunk; c += 256) {
            u32x4_t gv;
            if (RMS) gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + c);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                u32x4_t xv = *(reinterpret_cast<const u32x4_t*>(g.A + (int64_t)m * g.lda) + c);
                if (RMS) {
                    ssq[m] += dot8(xv, xv);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xv[j] = pack_bf16x2(__uint_as_float(xv[j] << 16) * __uint_as_float(gv[j] << 16),
                                            __uint_as_float(xv[j] & 0xffff0000u) * __uint_as_float(gv[j] & 0xffff0000u));
                }
                if (XLDS) xs[m * nchunk + c] = xv;
            }
        }
        if (RMS) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(ssq[m]);
                if (lane == 0) s_red[wave][m] = v;
            }
        }
        __syncthreads();
        if (RMS) {
            if (threadIdx.x < M)
                s_rstd[threadIdx.x] = rsqrtf((s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] +
                                              s_red[3][threadIdx.x]) / (float)g.K + g.rms_eps);
            __syncthreads();
        }
        };

    float a0[M], a1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) a0[m] = a1[m] = 0.0f;

    auto consume = [&](const u32x4_t (&wa)[U], const u32x4_t (&wb)[U], int step) __attribute__((always_inline)) {
        const int pr = wave_global + (step / nbatch) * nwaves;
        const int c0 = (step % nbatch) * (64 * U) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c0 + u * 64;
            if (cc < nchunk) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    u32x4_t xv;
                    if (XLDS) {
                        xv = xs[m * nchunk + cc];
                    } else {
                        xv = *(reinterpret_cast<const u32x4_t*>(g.A + (int64_t)m * g.lda) + cc);
                        if (RMS) {
                            const u32x4_t gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + cc);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                xv[j] = pack_bf16x2(
                                    __uint_as_float(xv[j] << 16) * __uint_as_float(gv[j] << 16),
                                    __uint_as_float(xv[j] & 0xffff0000u) * __uint_as_float(gv[j] & 0xffff0000u));
                        }
                    }
                    a0[m] += dot8(wa[u], xv);
                    if (ROWS == 2) a1[m] += dot8(wb[u], xv);
                }
            }
        }
        if ((step % nbatch) != nbatch - 1) return;
        // ---- last batch of this row group: reduce + epilogue ------------------------------------------
        const int n0 = ROWS * pr, n1 = (ROWS == 2 && n0 + 1 < g.N) ? n0 + 1 : n0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            a0[m] = wave_sum(a0[m]);
            if (ROWS == 2) a1[m] = wave_sum(a1[m]);
            if (RMS) {
                a0[m] *= s_rstd[m];
                a1[m] *= s_rstd[m];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v0 = a0[m] + (g.bias ? bf16_to_f32(g.bias[n0]) : 0.0f);
                float v1 = a1[m] + (g.bias ? bf16_to_f32(g.bias[n1]) : 0.0f);
                if (ROWS == 2 && g.act == ACT_SWIGLU) {
                    const float o = (v0 / (1.0f + __expf(-v0))) * v1;
                    const int64_t idx = (int64_t)m * g.ldc + pr;
                    if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
                    else if (COH_OUT) __hip_atomic_store(static_cast<bf16_t*>(g.C) + idx, f32_to_bf16(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
                    continue;
                }
                v0 = act_apply(v0, g.act);
                v1 = act_apply(v1, g.act);
                if (g.residual) {
                    const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
                    v0 += bf16_to_f32(g.residual[rrow * g.ldr + n0]);
                    v1 += bf16_to_f32(g.residual[rrow * g.ldr + n1]);
                }
                if (g.out_f32) {
                    float* C = static_cast<float*>(g.C) + (int64_t)m * g.ldc;
                    C[n0] = v0;
                    if (n1 != n0) C[n1] = v1;
                } else {
                    bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc;
                    C[n0] = f32_to_bf16(v0);
                    if (n1 != n0) C[n1] = f32_to_bf16(v1);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) a0[m] = a1[m] = 0.0f;
    };

    // ---- software pipeline over (row group, chunk batch) steps: the loads of step s+1 are in flight while
    //      step s is consumed, including across row groups and across the block prologue ---------------------
    u32x4_t wa0[U], wb0[U], wa1[U], wb1[U];
    if (nsteps > 0) issue(wa0, wb0, 0);
    if (sy.wait_ctr) {  // consumer: the first weight loads are in flight; now wait for the producers of the input vector
        __shared__ int s_ok;
        if (threadIdx.x == 0) {
            int ok = 1;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(const_cast<int32_t*>(sy.wait_ctr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sy.wait_target) {
                __builtin_amdgcn_s_sleep(100);  // ~2.7 us: up to ~1000 blocks poll this one line; the wait itself is 20+ us
                if (wall_clock64() - t0 > kGemvWaitTicks) {
                    __hip_atomic_store(sy.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;
    }
    if (XLDS || RMS) prologue();
    for (int s2 = 0; s2 < nsteps; s2 += 2) {
        if (s2 + 1 < nsteps) issue(wa1, wb1, s2 + 1);
        consume(wa0, wb0, s2);
        if (s2 + 1 < nsteps) {
            if (s2 + 2 < nsteps) issue(wa0, wb0, s2 + 2);
            consume(wa1, wb1, s2 + 1);
        }
    }
    if (sy.arrive_ctr) {  // producer: outputs performed, one arrival per block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sy.arrive_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int M, int ROWS, bool RMS, bool XLDS>
__global__ __launch_bounds__(256) void gemv_kernel(GemmArgs g) {
    gemv_body<M, ROWS, RMS, XLDS>(g, blockIdx.x, gridDim.x, GemvSync());
}

// (opt-in; measured SLOWER than two launches on MI355X: 80-90 us vs 51 us - each half alone runs at its stand-alone speed
// inside this kernel, 37 + 21 us, but the two streaming patterns sharing one grid cost another 22 us)
// gate|up (RMSNorm + SwiGLU fused) and down (+ residual) of one decoder layer in ONE launch: blocks [0, na) are the gate|up
// GEMV; blocks [na, na + nb) are the down GEMV - they stream their first weight rows immediately and wait on `counter`
// for the na producers before staging h.  Counter monotonic over the tokens of a generation (target = na * (step + 1)).
struct GemvPairArgs {
    GemmArgs ga, gb;
    int na, nb;
    const int32_t* step_dev;
    int32_t* counter;
    int32_t* status;
};
__global__ __launch_bounds__(256) void gemv_pair_kernel(GemvPairArgs p) {
    if ((int)blockIdx.x < p.na) {
        GemvSync sy;
        sy.arrive_ctr = p.counter;
        gemv_body<1, 2, true, true, false, true>(p.ga, blockIdx.x, p.na, sy);
    } else {
        GemvSync sy;
        sy.wait_ctr = p.counter;
        sy.wait_target = p.na * (*p.step_dev + 1);
        sy.status = p.status;
        gemv_body<1, 1, false, true, true, false>(p.gb, (int)blockIdx.x - p.na, p.nb, sy);
    }
}

// first index of the row maximum (torch.argmax tie rule), one 1024-thread block per row, 16-byte loads
// (the lm_head logits row is 128 KB: ~5 us instead of 38 us for the 256-thread scalar version)
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ x, int cols, int32_t* __restrict__ out) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* r = x + (int64_t)blockIdx.x * cols;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    };
    const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 2);  // scalars up to the first 16-byte boundary
    const int nhead = head < cols ? head : cols;
    if ((int)threadIdx.x < nhead) take(r[threadIdx.x], threadIdx.x);
    const int nvec = (cols - nhead) >> 2;
    const float4* rv = reinterpret_cast<const float4*>(r + nhead);
    for (int i = threadIdx.x; i < nvec; i += 1024) {
        const float4 v = rv[i];
        const int c = nhead + 4 * i;
        take(v.x, c);
        take(v.y, c + 1);
        take(v.z, c + 2);
        take(v.w, c + 3);
    }
    for (int i = nhead + 4 * nvec + threadIdx.x; i < cols; i += 1024) take(r[i], i);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = best;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        out[blockIdx.x] = bi;
    }
}

}  // namespace

template <int M>
static int launch_gemv(const GemmArgs& g, hipStream_t st) {
    const size_t xbytes = (size_t)M * g.K * 2;
    // (x staged in LDS also when the GEMV shares the CUs with the encoder's 128-KB-LDS GEMM blocks: never staging it costs
    //  12 ms end to end, staging only K = 4096 vectors 1 ms - LDS room is not what slows the decode next to the encoder)
    const bool xlds = xbytes <= 48 * 1024;
    const bool rows2 = g.act == ACT_SWIGLU || g.N > 8192;  // small N: one row per wave = twice the waves
    const int ngroups = rows2 ? (g.N + 1) / 2 : g.N;
    int blocks = (ngroups + 3) / 4;
    const int cap = 256 * 4;  // 4 resident blocks per CU; waves stride over the remaining rows
    if (blocks > cap) blocks = cap;
    const size_t lds = xlds ? xbytes : 0;
#define IVLM_GEMV_GO(ROWS, RMS, XL) gemv_kernel<M, ROWS, RMS, XL><<<blocks, 256, lds, st>>>(g)
    if (rows2) {
        if (g.rms_w) { if (xlds) IVLM_GEMV_GO(2, true, true); else IVLM_GEMV_GO(2, true, false); }
        else { if (xlds) IVLM_GEMV_GO(2, false, true); else IVLM_GEMV_GO(2, false, false); }
    } else {
        if (g.rms_w) { if (xlds) IVLM_GEMV_GO(1, true, true); else IVLM_GEMV_GO(1, true, false); }
        else { if (xlds) IVLM_GEMV_GO(1, false, true); else IVLM_GEMV_GO(1, false, false); }
    }
#undef IVLM_GEMV_GO
    return ivlm_launch_status();
}

// opt-in: measured slower than the wave-per-row kernel as a stand-alone launch (qkv 20.9 vs 18.7 us, o 18.6 vs 14.6 us):
// one 512-thread block per CU cannot overlap its own prologue / epilogue with streaming the way 4 small blocks per CU do
// h = SwiGLU(W_gu . RMSNorm(x2)), x_out = x2 + W_down . h in one launch (decode, M == 1)
int gemv_gu_down(const bf16_t* x2, const bf16_t* ln_w, float eps, const bf16_t* wgu, const bf16_t* wdown, bf16_t* h_scratch,
                 bf16_t* x_out, int hidden, int inter, const int32_t* step_dev, int32_t* counter, int32_t* status,
                 hipStream_t st) {
    if (!x2 || !ln_w || !wgu || !wdown || !h_scratch || !x_out || !step_dev || !counter || !status) return IVLM_ERR_INVALID_ARG;
    if ((hidden & 7) || (inter & 7) || (size_t)inter * 2 > 48 * 1024 || (size_t)hidden * 2 > 48 * 1024) return IVLM_ERR_UNSUPPORTED;
    GemvPairArgs p;
    p.ga = GemmArgs();
    p.ga.A = x2; p.ga.lda = hidden; p.ga.W = wgu; p.ga.ldw = hidden; p.ga.C = h_scratch; p.ga.ldc = inter;
    p.ga.M = 1; p.ga.N = 2 * inter; p.ga.K = hidden; p.ga.act = ACT_SWIGLU; p.ga.rms_w = ln_w; p.ga.rms_eps = eps;
    p.gb = GemmArgs();
    p.gb.A = h_scratch; p.gb.lda = inter; p.gb.W = wdown; p.gb.ldw = inter; p.gb.C = x_out; p.gb.ldc = hidden;
    p.gb.M = 1; p.gb.N = hidden; p.gb.K = inter; p.gb.act = ACT_NONE; p.gb.residual = x2; p.gb.ldr = hidden;
    auto blocks_of = [](int ngroups) { int b = (ngroups + 3) / 4; return b > 1024 ? 1024 : b; };
    p.na = blocks_of(inter);   // 2 rows (one gate/up pair) per wave
    p.nb = blocks_of(hidden);  // 1 row per wave
    p.step_dev = step_dev; p.counter = counter; p.status = status;
    const size_t lds = (size_t)std::max(hidden, inter) * 2;
    gemv_pair_kernel<<<p.na + p.nb, 256, lds, st>>>(p);
    return ivlm_launch_status();
}

static int g_gemv_slab = 0;
void gemv_set_slab(int on) { g_gemv_slab = on; }

int gemv_bf16(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.M > kMaxM || g.N <= 0 || g.K <= 0) return IVLM_ERR_INVALID_ARG;
    if ((g.K & 7) || (g.lda & 7) || (g.ldw & 7) || g.batch != 1) return IVLM_ERR_UNSUPPORTED;
    if (g.act == ACT_SWIGLU && ((g.N & 1) || g.residual)) return IVLM_ERR_UNSUPPORTED;
    if (g.M == 1 && g_gemv_slab && g.N >= 1024) {  // decode: flat slab streaming (gemv_slab.hip) when the shape qualifies
        const int rc = gemv_slab_bf16(g, st);
        if (rc != IVLM_ERR_UNSUPPORTED) return rc;
    }
    switch (g.M) {
        case 1: return launch_gemv<1>(g, st);
        case 2: return launch_gemv<2>(g, st);
        case 3: return launch_gemv<3>(g, st);
        case 4: return launch_gemv<4>(g, st);
        case 5: return launch_gemv<5>(g, st);
        case 6: return launch_gemv<6>(g, st);
        case 7: return launch_gemv<7>(g, st);
        default: return launch_gemv<8>(g, st);
    }
}

int argmax_f32(const float* x, int rows, int cols, int32_t* out, hipStream_t st) {
    if (!x || !out || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    argmax_kernel<<<rows, 1024, 0, st>>>(x, cols, out);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_argmax_f32(const float* x, int rows, int cols, int32_t* out, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::argmax_f32(x, rows, cols, out, ivlm_stream(stream));
}

extern "C" int ivlm_llama_gateup_down(const void* x2, const void* ln_w, float eps, const void* wgu, const void* wdown,
                                      void* h_scratch, void* x_out, int hidden, int inter, const int32_t* step_dev,
                                      int32_t* counter, int32_t* status, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::gemv_gu_down(static_cast<const bf16_t*>(x2), static_cast<const bf16_t*>(ln_w), eps,
                              static_cast<const bf16_t*>(wgu), static_cast<const bf16_t*>(wdown),
                              static_cast<bf16_t*>(h_scratch), static_cast<bf16_t*>(x_out), hidden, inter, step_dev, counter,
                              status, ivlm_stream(stream));
}
