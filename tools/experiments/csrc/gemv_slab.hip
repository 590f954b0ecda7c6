// Batch-1 decode GEMV as flat slab streaming:  out[1, N] = act((x . rms) . W[N,K]^T + bias) (+ residual)
//
// The linears of one generated token (HF LlamaDecoderLayer under InteractVLM.evaluate's greedy search,
// model/InteractVLM.py:524-531) stream 13.5 GB of bf16 weights; each launch is worth 20-30 us, so what counts beside the
// streaming rate is how evenly the bytes are spread and how little of the launch is not streaming:
//   * N rows are cut into gridDim contiguous slabs (one 512-thread block per CU); a block streams its slab as one flat,
//     perfectly coalesced byte range (lane t owns 16-byte chunks t, t + 512, ...): every lane of every CU carries the same
//     load, whatever N and K are (the wave-per-row kernel of gemv.hip quantises: 6144 row pairs over 4096 waves);
//   * the first 16 loads per lane are issued BEFORE the activation vector is staged (weights do not depend on it), so the
//     RMSNorm statistics / LDS staging hide under the first memory round trip;
//   * 8 bf16 x 8 bf16 -> fp32 on v_dot2c_f32_bf16, wave sums on the DPP network, per-row sums in a fixed order
//     (bit-reproducible), epilogues: fused-RMSNorm scale, bias, activation, SwiGLU over interleaved gate/up rows, residual.
#include "slab_stream.h"

namespace ivlm {
namespace {
using namespace slabk;

__device__ __forceinline__ float act1(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

__global__ __launch_bounds__(kThreads, 2) void gemv_slab_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lds L;
    L.xs = reinterpret_cast<u32x4_t*>(smem);
    L.part = reinterpret_cast<float*>(smem + (size_t)g.K * 2);
    L.rowsum = L.part + kMaxSteps * kWaves * 2;
    L.red = L.rowsum + kMaxRows;
    L.flag = nullptr;
    L.attn = nullptr;
    const int t = threadIdx.x;
    const bool swiglu = g.act == ACT_SWIGLU;
    int r0, r1;
    slab(g.N, swiglu ? 2 : 1, blockIdx.x, gridDim.x, r0, r1);
    u32x4_t buf[kDepth];
    prefetch(buf, g.W, g.K, r0, r1);
    const float rstd = stage_vec(g.A, g.rms_w, g.K, g.rms_eps, g.rms_w != nullptr, false, L);
    stream_slab(buf, g.W, g.K, r0, r1, L);
    const int nrows = r1 - r0;
    if (swiglu) {
        if (t < (nrows >> 1)) {
            const int n = r0 + 2 * t;
            const float v0 = L.rowsum[2 * t] * rstd + (g.bias ? bf16_to_f32(g.bias[n]) : 0.0f);
            const float v1 = L.rowsum[2 * t + 1] * rstd + (g.bias ? bf16_to_f32(g.bias[n + 1]) : 0.0f);
            const float o = (v0 / (1.0f + __expf(-v0))) * v1;
            if (g.out_f32) static_cast<float*>(g.C)[n >> 1] = o;
            else static_cast<bf16_t*>(g.C)[n >> 1] = f32_to_bf16(o);
        }
    } else if (t < nrows) {
        const int n = r0 + t;
        float v = L.rowsum[t] * rstd + (g.bias ? bf16_to_f32(g.bias[n]) : 0.0f);
        v = act1(v, g.act);
        if (g.residual) v += bf16_to_f32(g.residual[n]);  // M == 1: row 0 (res_mod irrelevant)
        if (g.out_f32) static_cast<float*>(g.C)[n] = v;
        else static_cast<bf16_t*>(g.C)[n] = f32_to_bf16(v);
    }
}

}  // namespace

// M == 1 fast path of the decode GEMV; returns IVLM_ERR_UNSUPPORTED when the shape does not qualify (caller falls back)
int gemv_slab_bf16(const GemmArgs& g, hipStream_t st) {
    if (g.M != 1 || g.batch != 1 || g.K < 512 || (g.K & 7) || g.ldw != g.K) return IVLM_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) return IVLM_ERR_UNSUPPORTED;
    if (g.act == ACT_SWIGLU && (g.N & 1)) return IVLM_ERR_UNSUPPORTED;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n_cu = 256;
    }
    const int unit = g.act == ACT_SWIGLU ? 2 : 1;
    const int units = g.N / unit;
    int G = n_cu;  // one resident 512-thread block per CU (256 VGPRs: 16 loads in flight per lane without spills)
    if (G > units) G = units;
    if (G < 1) G = 1;
    // slab tables in LDS: rows per block and lane steps per block
    const int max_rows = ((units + G - 1) / G) * unit + unit;
    const int64_t max_steps = ((int64_t)max_rows * (g.K >> 3) + kThreads - 1) / kThreads;
    if (max_rows > kMaxRows || max_steps + kDepth > kMaxSteps) return IVLM_ERR_UNSUPPORTED;
    const size_t lds = (size_t)g.K * 2 + (size_t)(kMaxSteps * kWaves * 2 + kMaxRows + 2 * kWaves) * 4 + 64;
    if (lds > 64 * 1024) return IVLM_ERR_UNSUPPORTED;  // (two blocks per CU; K <= ~20k)
    gemv_slab_kernel<<<G, kThreads, lds, st>>>(g);
    return ivlm_launch_status();
}

}  // namespace ivlm
