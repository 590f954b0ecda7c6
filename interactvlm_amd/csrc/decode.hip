// Single-token decode kernels of the LLaMA path (batch-1 greedy search driven by InteractVLM.evaluate,
// model/InteractVLM.py:524-531; arithmetic = HF LlamaAttention with a KV cache).
//
//   llama_decode_attn : RoPE(q,k at position pos) + KV-cache append + softmax(q.K^T/sqrt(d)).V for ONE new token,
//                       one workgroup per head.  Latency-bound (a few hundred KB of cache per layer): the three
//                       launches of the generic path (rope, cache write, attention) collapse into one.
//                       16 lanes share one key row (16-byte loads, 256-B coalesced per row), fp32 softmax.
#include "decode_attn.h"

namespace ivlm {
namespace {

using namespace decattn;

template <bool F32IO, bool LO = false, bool CF16 = false>
__global__ __launch_bounds__(kDecThreads) void llama_decode_attn_kernel(const void* __restrict__ qkv, bf16_t* __restrict__ kcache,
                                                                        bf16_t* __restrict__ vcache, void* __restrict__ o, int H,
                                                                        int D, int pos_arg, float theta, float scale,
                                                                        const float* __restrict__ ct,
                                                                        const float* __restrict__ stab,
                                                                        const int32_t* __restrict__ pos_dev, int tmax,
                                                                        bf16_t* __restrict__ kcache_lo, bf16_t* __restrict__ vcache_lo) {
    // a position past the cache slab is skipped: nothing is appended and the output row is written as ZEROS (a C caller that
    // steps past tmax reads zeros, not stale memory; the batched kernel below does the same)
    const int pos = pos_dev ? __builtin_amdgcn_readfirstlane(*pos_dev) : pos_arg;
    if (pos >= tmax || pos >= kMaxT) {
        if ((int)threadIdx.x < D) {
            const int64_t e = (int64_t)blockIdx.x * D + threadIdx.x;
            if (F32IO) static_cast<float*>(o)[e] = 0.0f;
            else static_cast<bf16_t*>(o)[e] = 0;
        }
        return;
    }
    llama_decode_attn_body<false, F32IO, 1024, LO, CF16>(blockIdx.x, qkv, kcache, vcache, o, H, D, pos_arg, theta, scale, ct, stab, pos_dev,
                                                   tmax, kcache_lo, vcache_lo);
}

// B sequences of one decode step: blockIdx.y picks the sequence; each has its own cache slab, qkv row, output row and
// position (the sequences of a batch sit at different lengths: prompts differ, model/InteractVLM.py:524-531 pads them).
template <bool F32IO, bool LO = false, bool CF16 = false>
__global__ __launch_bounds__(kDecThreads) void llama_decode_attn_batch_kernel(
    const void* __restrict__ qkv, int64_t ldq, bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache, int64_t cache_stride,
    void* __restrict__ o, int64_t ldo, int H, int D, float theta, float scale, const float* __restrict__ ct,
    const float* __restrict__ stab, const int32_t* __restrict__ pos_dev, int tmax, bf16_t* __restrict__ kcache_lo,
    bf16_t* __restrict__ vcache_lo) {
    const int b = blockIdx.y;
    constexpr int esz = F32IO ? 4 : 2;
    // a sequence that has filled its cache slab is skipped: its output row is ZERO (written here, so that the caller does not
    // have to clear the buffer with a launch of its own before every layer of every step)
    const int pos_b = __builtin_amdgcn_readfirstlane(pos_dev[b]);
    if (pos_b >= tmax || pos_b >= kMaxT) {
        if ((int)threadIdx.x < D) {
            const int64_t e = b * ldo + (int64_t)blockIdx.x * D + threadIdx.x;
            if (F32IO) static_cast<float*>(o)[e] = 0.0f;
            else static_cast<bf16_t*>(o)[e] = 0;
        }
        return;
    }
    llama_decode_attn_body<false, F32IO, 1024, LO, CF16>(blockIdx.x, static_cast<const char*>(qkv) + b * ldq * esz,
                                                   kcache + b * cache_stride, vcache + b * cache_stride,
                                                   static_cast<char*>(o) + b * ldo * esz, H, D, 0, theta, scale, ct, stab, pos_dev + b,
                                                   tmax, LO ? kcache_lo + b * cache_stride : nullptr,
                                                   LO ? vcache_lo + b * cache_stride : nullptr);
}

}  // namespace

int llama_decode_attn_batch(const void* qkv, int io_f32, int64_t ldq, bf16_t* kcache, bf16_t* vcache, int64_t cache_stride,
                            int tmax, void* o, int64_t ldo, int B, int H, int D, const int32_t* pos_dev, float theta, float scale,
                            const float* cos_tab, const float* sin_tab, hipStream_t st, bf16_t* kcache_lo, bf16_t* vcache_lo,
                            int cache_f16) {
    if (!qkv || !kcache || !vcache || !o || !pos_dev) return IVLM_ERR_INVALID_ARG;
    if (B <= 0 || B > 65535 || H <= 0 || D <= 0 || D > kMaxD || (D & 15)) return IVLM_ERR_INVALID_ARG;
    if (ldq < 3LL * H * D || ldo < (int64_t)H * D || cache_stride < (int64_t)H * D || ((ldq | ldo | cache_stride) & 7))
        return IVLM_ERR_INVALID_ARG;  // 16-byte rows
    if (tmax <= 0 || (int64_t)tmax * H * D > cache_stride) return IVLM_ERR_INVALID_ARG;
    if ((kcache_lo != nullptr) != (vcache_lo != nullptr) || (kcache_lo && !io_f32)) return IVLM_ERR_INVALID_ARG;
    if (cache_f16 && (!io_f32 || kcache_lo)) return IVLM_ERR_INVALID_ARG;
    if (cache_f16)
        llama_decode_attn_batch_kernel<true, false, true><<<dim3(H, B), kDecThreads, 0, st>>>(
            qkv, ldq, kcache, vcache, cache_stride, o, ldo, H, D, theta, scale, cos_tab, sin_tab, pos_dev, tmax, nullptr, nullptr);
    else if (kcache_lo)
        llama_decode_attn_batch_kernel<true, true><<<dim3(H, B), kDecThreads, 0, st>>>(
            qkv, ldq, kcache, vcache, cache_stride, o, ldo, H, D, theta, scale, cos_tab, sin_tab, pos_dev, tmax, kcache_lo, vcache_lo);
    else if (io_f32)
        llama_decode_attn_batch_kernel<true><<<dim3(H, B), kDecThreads, 0, st>>>(
            qkv, ldq, kcache, vcache, cache_stride, o, ldo, H, D, theta, scale, cos_tab, sin_tab, pos_dev, tmax, nullptr, nullptr);
    else
        llama_decode_attn_batch_kernel<false><<<dim3(H, B), kDecThreads, 0, st>>>(
            qkv, ldq, kcache, vcache, cache_stride, o, ldo, H, D, theta, scale, cos_tab, sin_tab, pos_dev, tmax, nullptr, nullptr);
    return ivlm_launch_status();
}

int llama_decode_attn(const void* qkv, int io_f32, bf16_t* kcache, bf16_t* vcache, int tmax, void* o, int H, int D, int pos,
                      float theta, float scale, hipStream_t st, const float* cos_tab, const float* sin_tab,
                      const int32_t* pos_dev, bf16_t* kcache_lo, bf16_t* vcache_lo, int cache_f16) {
    if (!qkv || !kcache || !vcache || !o || H <= 0 || D <= 0 || D > kMaxD || (D & 15) || tmax <= 0) return IVLM_ERR_INVALID_ARG;
    if (!pos_dev && (pos < 0 || pos >= kMaxT || pos >= tmax)) return IVLM_ERR_INVALID_ARG;
    if ((kcache_lo != nullptr) != (vcache_lo != nullptr) || (kcache_lo && !io_f32)) return IVLM_ERR_INVALID_ARG;
    if (cache_f16 && (!io_f32 || kcache_lo)) return IVLM_ERR_INVALID_ARG;
    if (cache_f16)
        llama_decode_attn_kernel<true, false, true><<<H, kDecThreads, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab,
                                                                               sin_tab, pos_dev, tmax, nullptr, nullptr);
    else if (kcache_lo)
        llama_decode_attn_kernel<true, true><<<H, kDecThreads, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab,
                                                                        sin_tab, pos_dev, tmax, kcache_lo, vcache_lo);
    else if (io_f32)
        llama_decode_attn_kernel<true><<<H, kDecThreads, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab, sin_tab,
                                                                  pos_dev, tmax, nullptr, nullptr);
    else
        llama_decode_attn_kernel<false><<<H, kDecThreads, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab,
                                                                   sin_tab, pos_dev, tmax, nullptr, nullptr);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_llama_decode_attn(const void* qkv, int io_dtype, void* kcache, void* vcache, int tmax, void* o, int H, int D,
                                      int pos, const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                      const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_decode_attn(qkv, io_dtype == IVLM_F32, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), tmax, o,
                                   H, D, pos, theta, scale, ivlm_stream(stream), cos_tab, sin_tab, pos_dev);
}

// "parity" precision: K / V cached as hi + lo bf16 planes (fp32 qkv / o); kcache_lo / vcache_lo as the hi caches
extern "C" int ivlm_llama_decode_attn_split(const void* qkv, void* kcache, void* kcache_lo, void* vcache, void* vcache_lo, int tmax,
                                            void* o, int H, int D, int pos, const int32_t* pos_dev, float theta, float scale,
                                            const float* cos_tab, const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    if (!kcache_lo || !vcache_lo) return IVLM_ERR_INVALID_ARG;
    return ivlm::llama_decode_attn(qkv, 1, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), tmax, o, H, D, pos, theta,
                                   scale, ivlm_stream(stream), cos_tab, sin_tab, pos_dev, static_cast<bf16_t*>(kcache_lo),
                                   static_cast<bf16_t*>(vcache_lo));
}

extern "C" int ivlm_llama_decode_attn_batch_split(const void* qkv, int64_t ldq, void* kcache, void* kcache_lo, void* vcache,
                                                  void* vcache_lo, int64_t cache_stride, int tmax, void* o, int64_t ldo, int B, int H,
                                                  int D, const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                                  const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    if (!kcache_lo || !vcache_lo) return IVLM_ERR_INVALID_ARG;
    return ivlm::llama_decode_attn_batch(qkv, 1, ldq, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), cache_stride, tmax,
                                         o, ldo, B, H, D, pos_dev, theta, scale, cos_tab, sin_tab, ivlm_stream(stream),
                                         static_cast<bf16_t*>(kcache_lo), static_cast<bf16_t*>(vcache_lo));
}

extern "C" int ivlm_llama_decode_attn_batch(const void* qkv, int io_dtype, int64_t ldq, void* kcache, void* vcache,
                                            int64_t cache_stride, int tmax, void* o, int64_t ldo, int B, int H, int D,
                                            const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                            const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_decode_attn_batch(qkv, io_dtype == IVLM_F32, ldq, static_cast<bf16_t*>(kcache),
                                         static_cast<bf16_t*>(vcache), cache_stride, tmax, o, ldo, B, H, D, pos_dev, theta, scale,
                                         cos_tab, sin_tab, ivlm_stream(stream));
}

// fp16 KV cache (the fp16-operand prefill appends IEEE halves): fp32 qkv / o, K / V rows appended as fp16 and read back as fp16
extern "C" int ivlm_llama_decode_attn_f16(const void* qkv, void* kcache, void* vcache, int tmax, void* o, int H, int D, int pos,
                                          const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                          const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_decode_attn(qkv, 1, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), tmax, o, H, D, pos, theta,
                                   scale, ivlm_stream(stream), cos_tab, sin_tab, pos_dev, nullptr, nullptr, 1);
}

extern "C" int ivlm_llama_decode_attn_batch_f16(const void* qkv, int64_t ldq, void* kcache, void* vcache, int64_t cache_stride,
                                                int tmax, void* o, int64_t ldo, int B, int H, int D, const int32_t* pos_dev,
                                                float theta, float scale, const float* cos_tab, const float* sin_tab,
                                                ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_decode_attn_batch(qkv, 1, ldq, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), cache_stride, tmax,
                                         o, ldo, B, H, D, pos_dev, theta, scale, cos_tab, sin_tab, ivlm_stream(stream), nullptr,
                                         nullptr, 1);
}
