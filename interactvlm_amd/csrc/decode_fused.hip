// Fused decode launches: a consumer GEMV starts in the SAME launch as its producer, streams its weight rows (they do not
// depend on the producer) and only then waits, on a device counter, for the producer's output.
//
//   attn_oproj : blocks [0, H) run single-token attention for one head each (decode_attn.h) and publish their output row
//                with agent-scope stores + one counter increment; the other blocks each own 32 rows of o_proj.weight,
//                put them in flight immediately, wait for the H increments, stage the attention vector and finish
//                x_out = x + W_o . a.  One launch instead of two, and the 33 MB of W_o stream while the attention - which is
//                bounded by one CU's ~30 GB/s per head - runs (HF LlamaAttention, model/InteractVLM.py:524-531 call path).
//
// Protocol (placement-independent): producers are the lowest block ids; the grid (H + rows/32 blocks of 1024 threads, one
// per CU) is far below the resident capacity, so every block is resident whatever the dispatch order; the counter is
// monotonic over the tokens of one generation (target = H * (tokens decoded + 1), the token count lives in device memory
// next to the position, so a captured HIP graph replays unchanged); every spin is bounded by wall clock and reports
// through the status word instead of hanging.
#include "decode_attn.h"

namespace ivlm {
namespace {
using namespace decattn;

constexpr int kRowsPerWave = 2;
constexpr int kRowsPerBlock = (kDecThreads / 64) * kRowsPerWave;  // 32
constexpr long long kSpinTimeoutTicks = 100000000LL;             // 1 s of the 100 MHz wall clock
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

struct AttnOprojArgs {
    const float* qkv;    // [3, H, D] of the new token (fp32: output of the q|k|v GEMV)
    bf16_t* kcache;      // [Tmax, H, D] this layer
    bf16_t* vcache;
    float* attn;         // [H*D] scratch: attention output (exchanged inside the launch), fp32
    const bf16_t* wo;    // [hidden, hidden]
    const float* x;      // [hidden] residual stream (fp32)
    float* x_out;        // [hidden]
    int H, D, tmax;
    float theta, scale;
    const float* cos_tab;
    const float* sin_tab;
    const int32_t* pos_dev;    // position of the new token
    const int32_t* step_dev;   // tokens decoded so far in this generation (0 for the first)
    int32_t* counter;          // this layer's arrival counter (zeroed at the start of the generation)
    int32_t* status;           // [0] != 0: a bounded wait expired
};

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;
// 8 bf16 weights x 8 fp32 activations: exact products, fp32 accumulation
__device__ __forceinline__ float dot8f(const u32x4_t& w, const f32x4v_t& xa, const f32x4v_t& xb, float acc) {
    acc = fmaf(__uint_as_float(w[0] << 16), xa[0], acc);
    acc = fmaf(__uint_as_float(w[0] & 0xffff0000u), xa[1], acc);
    acc = fmaf(__uint_as_float(w[1] << 16), xa[2], acc);
    acc = fmaf(__uint_as_float(w[1] & 0xffff0000u), xa[3], acc);
    acc = fmaf(__uint_as_float(w[2] << 16), xb[0], acc);
    acc = fmaf(__uint_as_float(w[2] & 0xffff0000u), xb[1], acc);
    acc = fmaf(__uint_as_float(w[3] << 16), xb[2], acc);
    acc = fmaf(__uint_as_float(w[3] & 0xffff0000u), xb[3], acc);
    return acc;
}

template <int NB>  // NB = 16-byte chunks per lane per row = hidden / 512 (8 for 4096, 10 for 5120)
__global__ __launch_bounds__(kDecThreads) void attn_oproj_kernel(AttnOprojArgs a) {
    const int hidden = a.H * a.D;
    if ((int)blockIdx.x < a.H) {
        llama_decode_attn_body<true, true>(blockIdx.x, a.qkv, a.kcache, a.vcache, a.attn, a.H, a.D, 0, a.theta, a.scale,
                                           a.cos_tab, a.sin_tab, a.pos_dev, a.tmax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's agent-scope stores are performed
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char xs_raw[8192 * 4];  // hidden <= 8192 fp32 (32 KB), two planes
    __shared__ int s_ok;
    const f32x4v_t* xs = reinterpret_cast<const f32x4v_t*>(xs_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = ((int)blockIdx.x - a.H) * kRowsPerBlock + wave * kRowsPerWave;
    const int nchunk = hidden >> 3;
    // ---- weights first: 2 rows x NB chunks per lane in flight before anything is waited for -------------------------
    u32x4_t w[kRowsPerWave][NB];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int n = min(row0 + r, hidden - 1);
        const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(a.wo + (int64_t)n * hidden);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int cc = min(lane + 64 * c, nchunk - 1);
            w[r][c] = __builtin_nontemporal_load(wr + cc);
        }
    }
    // ---- wait for the H attention blocks of THIS token --------------------------------------------------------------
    if (threadIdx.x == 0) {
        const int target = a.H * (*a.step_dev + 1);
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > kSpinTimeoutTicks) {
                __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    // ---- stage the attention vector (agent-scope loads: written by other CUs inside this launch) ---------------------
    {   // element i = 8c + j -> plane (j >> 2), slot 4c + (j & 3): consecutive lanes then read consecutive 16-byte slots
        float* xs32 = reinterpret_cast<float*>(xs_raw);
        for (int i = threadIdx.x; i < hidden; i += kDecThreads)
            xs32[((i >> 2) & 1) * (hidden >> 1) + ((i >> 3) << 2) + (i & 3)] =
                __hip_atomic_load(a.attn + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int cc = lane + 64 * c;
            if (cc < nchunk) acc = dot8f(w[r][c], xs[cc], xs[nchunk + cc], acc);
        }
        acc = wave_sum(acc);
        const int n = row0 + r;
        if (lane == 0 && n < hidden) a.x_out[n] = acc + a.x[n];
    }
}

}  // namespace

int llama_attn_oproj(const float* qkv, bf16_t* kcache, bf16_t* vcache, int tmax, float* attn_scratch, const bf16_t* wo,
                     const float* x, float* x_out, int H, int D, float theta, float scale, const float* cos_tab,
                     const float* sin_tab, const int32_t* pos_dev, const int32_t* step_dev, int32_t* counter, int32_t* status,
                     hipStream_t st) {
    if (!qkv || !kcache || !vcache || !attn_scratch || !wo || !x || !x_out || !pos_dev || !step_dev || !counter || !status)
        return IVLM_ERR_INVALID_ARG;
    if (H <= 0 || D <= 0 || D > kMaxD || (D & 15)) return IVLM_ERR_INVALID_ARG;
    const int hidden = H * D;
    if (hidden % 512 != 0 || hidden > 8192 || tmax <= 0) return IVLM_ERR_UNSUPPORTED;
    AttnOprojArgs a;
    a.qkv = qkv; a.kcache = kcache; a.vcache = vcache; a.attn = attn_scratch; a.wo = wo; a.x = x; a.x_out = x_out;
    a.H = H; a.D = D; a.tmax = tmax; a.theta = theta; a.scale = scale; a.cos_tab = cos_tab; a.sin_tab = sin_tab;
    a.pos_dev = pos_dev; a.step_dev = step_dev; a.counter = counter; a.status = status;
    const int grid = H + (hidden + kRowsPerBlock - 1) / kRowsPerBlock;
    {  // consumers wait for producers inside the launch: the whole grid must be resident (one block per CU)
        static int cus = 0;  // queried once (never during a stream capture: the graph path warms up first)
        if (cus == 0) {
            int dev = 0;
            IVLM_HIP_TRY(hipGetDevice(&dev));
            IVLM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        }
        if (grid > cus) return IVLM_ERR_UNSUPPORTED;
    }
    switch (hidden / 512) {
        case 8: attn_oproj_kernel<8><<<grid, kDecThreads, 0, st>>>(a); break;    // 4096 (LLaMA-2 7B)
        case 10: attn_oproj_kernel<10><<<grid, kDecThreads, 0, st>>>(a); break;  // 5120 (13B)
        case 1: attn_oproj_kernel<1><<<grid, kDecThreads, 0, st>>>(a); break;    // 512  (tests)
        case 2: attn_oproj_kernel<2><<<grid, kDecThreads, 0, st>>>(a); break;    // 1024 (tests)
        default: return IVLM_ERR_UNSUPPORTED;
    }
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_llama_attn_oproj(const float* qkv, void* kcache, void* vcache, int tmax, float* attn_scratch, const void* wo,
                                     const float* x, float* x_out, int H, int D, float theta, float scale,
                                     const float* cos_tab, const float* sin_tab, const int32_t* pos_dev,
                                     const int32_t* step_dev, int32_t* counter, int32_t* status, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_attn_oproj(qkv, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), tmax, attn_scratch,
                                  static_cast<const bf16_t*>(wo), x, x_out, H, D, theta, scale, cos_tab, sin_tab, pos_dev, step_dev,
                                  counter, status, ivlm_stream(stream));
}
