// All decoder layers of ONE generated token in ONE launch, as a dataflow over role-specialised workgroups.
//
// Replaces, per token, 32 x [q|k|v GEMV, attention, o_proj GEMV, gate|up GEMV, down GEMV] = 160 launches of HF
// LlamaDecoderLayer under InteractVLM.evaluate's greedy search (model/InteractVLM.py:524-531).  Batch-1 decode only
// streams weights (404 MB per 7B layer); as separate launches every GEMV pays a ramp, a tail and a dependency bubble
// (~7 us each, 40 % of the step).  A persistent kernel with device-wide barriers does not fix that (generate.hip: every
// barrier costs ~9 us of skew + store/load latency during which HBM idles).  Here there are NO barriers:
//
//   * the grid is the concatenation, layer by layer, of five kinds of 512-thread blocks (one resident per CU, 256 VGPRs):
//       R0 q|k|v rows -> R1 attention heads (1 per block) -> R2 o_proj rows -> R3 gate|up pairs -> R4 down rows; a GEMV block
//       chains 4-8 work items (16 rows each) through a double-buffered register pipeline and stages its input once;
//   * every block first puts ITS weight rows (R1: its K/V cache rows) in flight - they depend on nothing - and only then
//     waits, on a device counter, for the blocks that produce its input vector; it computes, publishes its outputs with
//     agent-scope stores and bumps its own role's counter;
//   * workgroups are dispatched in index order and a block only ever waits for lower indices, so whatever is resident can
//     make progress; while one role drains (sync latency, RMS statistics, the attention round trips) the blocks of the
//     next roles are already resident and streaming their weights: HBM never idles on a dependency;
//   * counters are monotonic over the tokens of a generation (target = blocks_of_role * (tokens decoded + 1), the token
//     count lives in device memory), so one captured HIP graph replays for every token; every wait is bounded (wall
//     clock) and fails into a status word instead of hanging.
// Arithmetic is that of the per-op path (gemv.hip / decode_attn.h): bf16(x * gamma) staged once per block, fp32 dots on
// v_dot2c_f32_bf16, rstd applied to the fp32 sum, SwiGLU over interleaved gate/up rows, bf16 residual stream.
#include "decode_attn.h"

namespace ivlm {
namespace {
using namespace decattn;

constexpr int kT = 512;      // 8 waves with 256 VGPRs each: two 64-register row sets in flight per lane
constexpr int kW = kT / 64;  // 8 waves
constexpr long long kWaitTicks = 100000000LL;  // 1 s of the 100 MHz wall clock
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef const __attribute__((address_space(1))) u32x4_t* gvec_ptr;  // table pointers are flat to the compiler: force global
__device__ __forceinline__ gvec_ptr as_gvec(const void* p) { return (gvec_ptr)(uintptr_t)p; }

struct LayersArgs {
    const int64_t* layer_ptrs;  // device [L][6]: ln1, q|k|v [3h,h], o [h,h], ln2, gate|up interleaved [2i,h], down [h,i]
    int L, H, D, hidden, inter;
    float eps, theta, scale;
    const float* cos_tab;
    const float* sin_tab;
    bf16_t* kcache;  // [L, Tmax, H, D]
    bf16_t* vcache;
    int64_t cache_layer_stride;
    const bf16_t* x0;   // [hidden] input embedding of the token (written before the launch)
    bf16_t* xbuf;       // [2][hidden] residual stream between layers; the output is xbuf[(L-1) & 1]
    bf16_t* qkv;        // [3*hidden]
    bf16_t* attn;       // [hidden]
    bf16_t* x2;         // [hidden]
    bf16_t* hbuf;       // [inter]
    const int32_t* pos_dev;
    const int32_t* step_dev;
    int32_t* counters;  // [L][5][32] (one 128-byte line per counter), zeroed at the start of a generation
    int32_t* status;    // [0] != 0: a bounded wait expired
    int nb[5];          // blocks per role
    int items[5];       // work items per role (an item = kW waves x rows-per-wave rows)
    int group[5];       // items per block
};

__device__ __forceinline__ float dot8(const u32x4_t& w, const u32x4_t& x) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t wj = w[j], xj = x[j];
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wj), __builtin_bit_cast(bf16x2_t, xj), acc, false);
    }
    return acc;
}
__device__ __forceinline__ uint32_t ld_agent32(const void* p) {
    return __hip_atomic_load(static_cast<uint32_t*>(const_cast<void*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bf16_t ld_agent16(const bf16_t* p) {
    return __hip_atomic_load(const_cast<bf16_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent16(bf16_t* p, bf16_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until *ctr >= target (thread 0 polls, everybody learns the outcome); false = gave up (status set)
__device__ __forceinline__ bool wait_counter(const int32_t* ctr, int target, int32_t* status, int* s_flag) {
    if (threadIdx.x == 0) {
        int ok = 1;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(const_cast<int32_t*>(ctr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > kWaitTicks) {
                __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

// all outputs of this block are performed -> one arrival on the role's counter
__device__ __forceinline__ void arrive(int32_t* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stage a K-vector in LDS (bf16, optionally x * gamma with the RMS statistic); src read coherently unless `plain`
__device__ __forceinline__ float stage(u32x4_t* xs, const bf16_t* src, const bf16_t* gamma, int K, float eps, bool plain,
                                       float* s_red) {
    uint32_t* xs32 = reinterpret_cast<uint32_t*>(xs);
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    const uint32_t* g32 = reinterpret_cast<const uint32_t*>(gamma);
    float ssq = 0.0f;
    for (int i = threadIdx.x; i < (K >> 1); i += kT) {
        uint32_t v = plain ? s32[i] : ld_agent32(s32 + i);
        if (gamma) {
            const float lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
            ssq += lo * lo + hi * hi;
            const uint32_t gv = g32[i];
            v = pack_bf16x2(lo * __uint_as_float(gv << 16), hi * __uint_as_float(gv & 0xffff0000u));
        }
        xs32[i] = v;
    }
    float rstd = 1.0f;
    if (gamma) {
        ssq = wave_sum(ssq);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = ssq;
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < kW; ++w) tot += s_red[w];
        rstd = rsqrtf(tot / (float)K + eps);
    }
    __syncthreads();
    return rstd;
}

// NB 16-byte chunks per lane of one weight row (K = 8 * 64 * NB, the last chunk column may be partial)
template <int NB>
__device__ __forceinline__ void load_row(u32x4_t (&w)[NB], const bf16_t* W, int row, int nrows, int K) {
    const int nchunk = K >> 3;
    gvec_ptr wr = as_gvec(W + (int64_t)min(row, nrows - 1) * K);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < NB; ++c) w[c] = __builtin_nontemporal_load(wr + min(lane + 64 * c, nchunk - 1));
}
template <int NB>
__device__ __forceinline__ float dot_row(const u32x4_t (&w)[NB], const u32x4_t* xs, int K) {
    const int nchunk = K >> 3;
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const int cc = lane + 64 * c;
        if (cc < nchunk) acc += dot8(w[c], xs[cc]);
    }
    return wave_sum(acc);
}

// One GEMV role of one block: `nitems` consecutive items, an item = kW waves x ROWS rows (wave w of item i owns rows
// (first_item + i) * kW * ROWS + w * ROWS ...).  The first two items are in flight before `sync_and_stage()` (wait for the
// producers + stage the input vector ONCE for all items) is called; afterwards the loads of item i+2 go out as soon as item
// i's registers are free (double buffer, static indices).  epi(row0, sums) publishes ROWS results of a wave (lane 0 only).
template <int NB, int ROWS, class SyncStage, class Epi>
__device__ __forceinline__ bool gemv_role(const bf16_t* W, int nrows, int K, int first_item, int nitems, const u32x4_t* xs,
                                          SyncStage sync_and_stage, Epi epi) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4_t wa[ROWS][NB], wb[ROWS][NB];
    auto row_of = [&](int i) { return ((first_item + i) * kW + wave) * ROWS; };
    auto load = [&](u32x4_t (&w)[ROWS][NB], int i) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) load_row<NB>(w[r], W, row_of(i) + r, nrows, K);
    };
    auto compute = [&](const u32x4_t (&w)[ROWS][NB], int i, float scale) {
        float v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) v[r] = dot_row<NB>(w[r], xs, K) * scale;
        if (lane == 0) epi(row_of(i), v);
    };
    load(wa, 0);
    if (nitems > 1) load(wb, 1);
    float scale;
    if (!sync_and_stage(scale)) return false;
    for (int i = 0; i < nitems; i += 2) {
        compute(wa, i, scale);
        if (i + 2 < nitems) load(wa, i + 2);
        if (i + 1 < nitems) {
            compute(wb, i + 1, scale);
            if (i + 3 < nitems) load(wb, i + 3);
        }
    }
    return true;
}

template <int NBH, int NBI>  // chunks per lane per row for K = hidden / K = inter
__global__ __launch_bounds__(kT, 2) void llama_layers_kernel(LayersArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char xs_raw[NBI * 64 * 16 > NBH * 64 * 16 ? NBI * 64 * 16 : NBH * 64 * 16];
    __shared__ float s_red[kW];
    __shared__ int s_flag;
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(xs_raw);
    const int per_layer = a.nb[0] + a.nb[1] + a.nb[2] + a.nb[3] + a.nb[4];
    const int l = blockIdx.x / per_layer;
    int r = blockIdx.x - l * per_layer;
    int role = 0;
    while (r >= a.nb[role]) {
        r -= a.nb[role];
        ++role;
    }
    const int64_t* lp = a.layer_ptrs + (int64_t)l * 6;
    int32_t* ctr = a.counters + ((int64_t)l * 5) * 32;
    const int step1 = *a.step_dev + 1;
    const int hidden = a.hidden, inter = a.inter;
    const bf16_t* x_in = l == 0 ? a.x0 : a.xbuf + (int64_t)((l - 1) & 1) * hidden;  // residual stream entering this layer
    const bool x_plain = l == 0;
    const int first = r * a.group[role];                                     // first item of this block
    const int nitems = min(a.group[role], a.items[role] - first);           // (>= 1 by construction of the grid)

    if (role == 0) {  // ---- q|k|v = W_qkv . RMSNorm(x): items of 2 rows per wave ------------------------------------
        const bool ok = gemv_role<NBH, 2>(
            reinterpret_cast<const bf16_t*>(lp[1]), 3 * hidden, hidden, first, nitems, xs,
            [&](float& scale) {
                if (l > 0 && !wait_counter(ctr - 5 * 32 + 4 * 32, a.nb[4] * step1, a.status, &s_flag)) return false;
                scale = stage(xs, x_in, reinterpret_cast<const bf16_t*>(lp[0]), hidden, a.eps, x_plain, s_red);
                return true;
            },
            [&](int n0, const float (&v)[2]) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (n0 + k < 3 * hidden) st_agent16(a.qkv + n0 + k, f32_to_bf16(v[k]));
            });
        if (!ok) return;
        arrive(ctr + 0 * 32);
    } else if (role == 1) {  // ---- attention of head r ----------------------------------------------------------------
        const int32_t* c0 = ctr + 0 * 32;
        const int target = a.nb[0] * step1;
        int32_t* status = a.status;
        int* flag = &s_flag;
        auto waiter = [=]() { return wait_counter(c0, target, status, flag); };
        llama_decode_attn_body<true, true, decltype(waiter), kT>(r, a.qkv, a.kcache + (int64_t)l * a.cache_layer_stride,
                                                                  a.vcache + (int64_t)l * a.cache_layer_stride, a.attn, a.H,
                                                                  a.D, 0, a.theta, a.scale, a.cos_tab, a.sin_tab, a.pos_dev,
                                                                  waiter);
        if (s_flag == 0) return;
        arrive(ctr + 1 * 32);
    } else if (role == 2) {  // ---- x2 = x + W_o . attn ----------------------------------------------------------------
        const bool ok = gemv_role<NBH, 2>(
            reinterpret_cast<const bf16_t*>(lp[2]), hidden, hidden, first, nitems, xs,
            [&](float& scale) {
                if (!wait_counter(ctr + 1 * 32, a.nb[1] * step1, a.status, &s_flag)) return false;
                scale = stage(xs, a.attn, nullptr, hidden, 0.0f, false, s_red);
                return true;
            },
            [&](int n0, const float (&v)[2]) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (n0 + k < hidden) {
                        const float res = bf16_to_f32(x_plain ? x_in[n0 + k] : ld_agent16(x_in + n0 + k));
                        st_agent16(a.x2 + n0 + k, f32_to_bf16(v[k] + res));
                    }
            });
        if (!ok) return;
        arrive(ctr + 2 * 32);
    } else if (role == 3) {  // ---- h = SiLU(gate) * up, RMSNorm fused: one (gate, up) row pair per wave and item ---------
        const bool ok = gemv_role<NBH, 2>(
            reinterpret_cast<const bf16_t*>(lp[4]), 2 * inter, hidden, first, nitems, xs,
            [&](float& scale) {
                if (!wait_counter(ctr + 2 * 32, a.nb[2] * step1, a.status, &s_flag)) return false;
                scale = stage(xs, a.x2, reinterpret_cast<const bf16_t*>(lp[3]), hidden, a.eps, false, s_red);
                return true;
            },
            [&](int n0, const float (&v)[2]) {
                const int pr = n0 >> 1;
                if (pr < inter) st_agent16(a.hbuf + pr, f32_to_bf16((v[0] / (1.0f + __expf(-v[0]))) * v[1]));
            });
        if (!ok) return;
        arrive(ctr + 3 * 32);
    } else {  // ---- x_next = x2 + W_down . h: one row per wave and item (K = inter) ---------------------------------------
        const bool ok = gemv_role<NBI, 1>(
            reinterpret_cast<const bf16_t*>(lp[5]), hidden, inter, first, nitems, xs,
            [&](float& scale) {
                if (!wait_counter(ctr + 3 * 32, a.nb[3] * step1, a.status, &s_flag)) return false;
                scale = stage(xs, a.hbuf, nullptr, inter, 0.0f, false, s_red);
                return true;
            },
            [&](int n0, const float (&v)[1]) {
                if (n0 < hidden) {
                    const float res = bf16_to_f32(ld_agent16(a.x2 + n0));
                    st_agent16(a.xbuf + (int64_t)(l & 1) * hidden + n0, f32_to_bf16(v[0] + res));
                }
            });
        if (!ok) return;
        arrive(ctr + 4 * 32);
    }
}

}  // namespace

size_t llama_layers_workspace_bytes(int L, int hidden, int inter) {
    return (size_t)L * 5 * 32 * 4 + 256 + (size_t)(2 * hidden + 3 * hidden + hidden + hidden + inter) * 2 + 8 * 256;
}

int llama_decode_layers(const int64_t* layer_ptrs, int L, int H, int D, int hidden, int inter, float eps, float theta,
                        float scale, const float* cos_tab, const float* sin_tab, bf16_t* kcache, bf16_t* vcache,
                        int64_t cache_layer_stride, const bf16_t* x0, bf16_t* x_out, const int32_t* pos_dev,
                        const int32_t* step_dev, void* workspace, size_t ws_bytes, hipStream_t st) {
    if (!layer_ptrs || !cos_tab || !sin_tab || !kcache || !vcache || !x0 || !x_out || !pos_dev || !step_dev || !workspace)
        return IVLM_ERR_INVALID_ARG;
    if (L <= 0 || H <= 0 || D <= 0 || D > kMaxD || (D & 15) || H * D != hidden) return IVLM_ERR_INVALID_ARG;
    if (hidden % 512 != 0 || (inter & 7) || inter < 512) return IVLM_ERR_UNSUPPORTED;
    if (ws_bytes < llama_layers_workspace_bytes(L, hidden, inter) || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return IVLM_ERR_WORKSPACE;
    LayersArgs a;
    a.layer_ptrs = layer_ptrs;
    a.L = L; a.H = H; a.D = D; a.hidden = hidden; a.inter = inter;
    a.eps = eps; a.theta = theta; a.scale = scale;
    a.cos_tab = cos_tab; a.sin_tab = sin_tab;
    a.kcache = kcache; a.vcache = vcache; a.cache_layer_stride = cache_layer_stride;
    a.x0 = x0;
    unsigned char* w = static_cast<unsigned char*>(workspace);
    a.counters = reinterpret_cast<int32_t*>(w);
    size_t off = (size_t)L * 5 * 32 * 4;
    a.status = reinterpret_cast<int32_t*>(w + off);
    off += 256;
    auto take = [&](size_t elems) {
        bf16_t* p = reinterpret_cast<bf16_t*>(w + off);
        off += (elems * 2 + 255) & ~(size_t)255;
        return p;
    };
    a.xbuf = take(2 * (size_t)hidden);
    a.qkv = take(3 * (size_t)hidden);
    a.attn = take(hidden);
    a.x2 = take(hidden);
    a.hbuf = take(inter);
    a.pos_dev = pos_dev;
    a.step_dev = step_dev;
    // items (kW waves x rows) per role and how many of them one block chains through its double-buffered pipeline:
    // more items per block = the input vector staged less often and fewer, fatter blocks; fewer = finer load balance
    a.items[0] = (3 * hidden + 2 * kW - 1) / (2 * kW); a.group[0] = 6;
    a.items[1] = H;                                     a.group[1] = 1;
    a.items[2] = (hidden + 2 * kW - 1) / (2 * kW);      a.group[2] = 4;
    a.items[3] = (inter + kW - 1) / kW;                 a.group[3] = 8;
    a.items[4] = (hidden + kW - 1) / kW;                a.group[4] = 4;
    for (int i = 0; i < 5; ++i) a.nb[i] = (a.items[i] + a.group[i] - 1) / a.group[i];
    const int64_t grid = (int64_t)L * (a.nb[0] + a.nb[1] + a.nb[2] + a.nb[3] + a.nb[4]);
    if (grid > 0x7fffffff) return IVLM_ERR_UNSUPPORTED;
    const int nbh = hidden / 512, nbi = (inter / 8 + 63) / 64;
#define IVLM_GO(NBH, NBI) llama_layers_kernel<NBH, NBI><<<(unsigned)grid, kT, 0, st>>>(a)
    if (nbh == 8 && nbi == 22) IVLM_GO(8, 22);         // LLaMA-2 7B: 4096 / 11008
    else if (nbh == 10 && nbi == 27) IVLM_GO(10, 27);  // 13B: 5120 / 13824
    else if (nbh == 2 && nbi == 3) IVLM_GO(2, 3);      // tests: 1024 / 1376
    else if (nbh == 1 && nbi == 2) IVLM_GO(1, 2);      // tests: 512 / 1024
    else return IVLM_ERR_UNSUPPORTED;
#undef IVLM_GO
    const int rc = ivlm_launch_status();
    if (rc != IVLM_OK) return rc;
    // the layer output sits in the workspace: hand it over (tiny D2D copy on the same stream)
    IVLM_HIP_TRY(hipMemcpyAsync(x_out, a.xbuf + (size_t)((L - 1) & 1) * hidden, (size_t)hidden * 2, hipMemcpyDeviceToDevice, st));
    return IVLM_OK;
}

}  // namespace ivlm

extern "C" size_t ivlm_llama_decode_layers_workspace_bytes(int L, int hidden, int inter) {
    return ivlm::llama_layers_workspace_bytes(L, hidden, inter);
}

extern "C" int ivlm_llama_decode_layers(const int64_t* layer_ptrs, int L, int H, int D, int hidden, int inter, float eps,
                                        float theta, float scale, const float* cos_tab, const float* sin_tab, void* kcache,
                                        void* vcache, int64_t cache_layer_stride, const void* x0, void* x_out,
                                        const int32_t* pos_dev, const int32_t* step_dev, void* workspace,
                                        size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_decode_layers(layer_ptrs, L, H, D, hidden, inter, eps, theta, scale, cos_tab, sin_tab,
                                     static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), cache_layer_stride,
                                     static_cast<const bf16_t*>(x0), static_cast<bf16_t*>(x_out), pos_dev, step_dev, workspace,
                                     workspace_bytes, ivlm_stream(stream));
}
