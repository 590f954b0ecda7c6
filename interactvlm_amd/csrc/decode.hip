// Single-token decode kernels of the LLaMA path (batch-1 greedy search driven by InteractVLM.evaluate,
// model/InteractVLM.py:524-531; arithmetic = HF LlamaAttention with a KV cache).
//
//   llama_decode_attn : RoPE(q,k at position pos) + KV-cache append + softmax(q.K^T/sqrt(d)).V for ONE new token,
//                       one workgroup per head.  Latency-bound (a few hundred KB of cache per layer): the three
//                       launches of the generic path (rope, cache write, attention) collapse into one.
//                       16 lanes share one key row (16-byte loads, 256-B coalesced per row), fp32 softmax.
#include "kernels.h"

namespace ivlm {
namespace {

constexpr int kMaxD = 128;
constexpr int kMaxT = 4096;  // scores live in LDS (16 KB)

constexpr int kDecThreads = 1024, kDecGroups = kDecThreads / 16;  // 64 key rows per sweep

__global__ __launch_bounds__(kDecThreads) void llama_decode_attn_kernel(const bf16_t* __restrict__ qkv /*[3,H,D]*/,
                                                                bf16_t* __restrict__ kcache /*[Tmax,H,D]*/,
                                                                bf16_t* __restrict__ vcache, bf16_t* __restrict__ o,
                                                                int H, int D, int pos_arg, float theta, float scale,
                                                                const float* __restrict__ ct,
                                                                const float* __restrict__ stab,
                                                                const int32_t* __restrict__ pos_dev) {
    // position from device memory when given: lets one captured HIP graph serve every decode step
    const int pos = pos_dev ? __builtin_amdgcn_readfirstlane(*pos_dev) : pos_arg;
    __shared__ float q_s[kMaxD];
    __shared__ float knew_s[kMaxD];
    __shared__ float vnew_s[kMaxD];
    __shared__ float sc[kMaxT];
    __shared__ float red[2 * kDecThreads / 64];
    __shared__ float part[kDecGroups][kMaxD];
    const int h = blockIdx.x, t = threadIdx.x;
    const int half = D >> 1;
    // ---- RoPE on q and the new k; append k, v to the cache -------------------------------------
    if (t < half) {
        const bf16_t* q = qkv + h * D;
        const bf16_t* k = qkv + (int64_t)H * D + h * D;
        float c, s;
        if (ct) {
            c = ct[pos * half + t];
            s = stab[pos * half + t];
        } else {
            const float ang = (float)pos * powf(theta, -(float)(2 * t) / (float)D);
            c = cosf(ang);
            s = sinf(ang);
        }
        const float q0 = bf16_to_f32(q[t]), q1 = bf16_to_f32(q[t + half]);
        const float k0 = bf16_to_f32(k[t]), k1 = bf16_to_f32(k[t + half]);
        // round q, k to bf16 exactly like the prefill path (rope_kv_kernel) so both paths see the same values
        const bf16_t qa = f32_to_bf16(q0 * c - q1 * s), qb = f32_to_bf16(q1 * c + q0 * s);
        const bf16_t ka = f32_to_bf16(k0 * c - k1 * s), kb = f32_to_bf16(k1 * c + k0 * s);
        q_s[t] = bf16_to_f32(qa);
        q_s[t + half] = bf16_to_f32(qb);
        knew_s[t] = bf16_to_f32(ka);
        knew_s[t + half] = bf16_to_f32(kb);
        bf16_t* kc = kcache + ((int64_t)pos * H + h) * D;
        kc[t] = ka;
        kc[t + half] = kb;
    } else if (t >= 128 && t < 128 + D) {
        const int d = t - 128;
        const bf16_t v = qkv[2 * (int64_t)H * D + h * D + d];
        vnew_s[d] = bf16_to_f32(v);
        vcache[((int64_t)pos * H + h) * D + d] = v;
    }
    __syncthreads();
    // ---- scores: 16 lanes per key row ----------------------------------------------------------
    const int sub = t & 15, grp = t >> 4;  // 64 groups of 16 lanes
    const int nch = D >> 3;                // 16-byte chunks per row (<= 16)
    float qr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] = sub < nch ? q_s[sub * 8 + e] : 0.0f;
    const int nkeys = pos + 1;
    for (int j0 = 0; j0 < nkeys; j0 += kDecGroups) {
        const int j = j0 + grp;
        float d = 0.0f;
        if (j < pos && sub < nch) {
            const uint4 kv = *reinterpret_cast<const uint4*>(kcache + ((int64_t)j * H + h) * D + sub * 8);
            const uint32_t* pk = &kv.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d += __uint_as_float(pk[e] << 16) * qr[2 * e];
                d += __uint_as_float(pk[e] & 0xffff0000u) * qr[2 * e + 1];
            }
        } else if (j == pos && sub < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d += knew_s[sub * 8 + e] * qr[e];
        }
        d += __shfl_xor(d, 8, 64);
        d += __shfl_xor(d, 4, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 1, 64);
        if (sub == 0 && j < nkeys) sc[j] = d * scale;
    }
    __syncthreads();
    // ---- softmax over sc[0..pos] (fp32) ----------------------------------------------------------
    constexpr int NW = kDecThreads / 64;
    float mx = -1.0e30f;
    for (int j = t; j < nkeys; j += kDecThreads) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.0f;
    for (int j = t; j < nkeys; j += kDecThreads) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[NW + (t >> 6)] = sum;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[NW + w];
    const float inv_sum = 1.0f / tot;
    // HF: softmax in fp32, cast to the model dtype, then @ V: round p to bf16 like the MFMA path does
    // ---- O = P.V: group grp owns keys j == grp (mod 64), lane sub owns 8 dims --------------------
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    for (int j = grp; j < nkeys; j += kDecGroups) {
        const float p = bf16_to_f32(f32_to_bf16(sc[j] * inv_sum));
        if (sub < nch) {
            if (j < pos) {
                const uint4 vv = *reinterpret_cast<const uint4*>(vcache + ((int64_t)j * H + h) * D + sub * 8);
                const uint32_t* pv = &vv.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += p * __uint_as_float(pv[e] << 16);
                    acc[2 * e + 1] += p * __uint_as_float(pv[e] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += p * vnew_s[sub * 8 + e];
            }
        }
    }
    if (sub < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[grp][sub * 8 + e] = acc[e];
    }
    __syncthreads();
    if (t < D) {
        float r = 0.0f;
#pragma unroll
        for (int g2 = 0; g2 < kDecGroups; ++g2) r += part[g2][t];
        o[h * D + t] = f32_to_bf16(r);
    }
}

}  // namespace

int llama_decode_attn(const bf16_t* qkv, bf16_t* kcache, bf16_t* vcache, bf16_t* o, int H, int D, int pos, float theta,
                      float scale, hipStream_t st, const float* cos_tab, const float* sin_tab, const int32_t* pos_dev) {
    if (!qkv || !kcache || !vcache || !o || H <= 0 || D <= 0 || D > kMaxD || (D & 15)) return IVLM_ERR_INVALID_ARG;
    if (!pos_dev && (pos < 0 || pos >= kMaxT)) return IVLM_ERR_INVALID_ARG;
    llama_decode_attn_kernel<<<H, kDecThreads, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab, sin_tab,
                                                       pos_dev);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_llama_decode_attn(const void* qkv, void* kcache, void* vcache, void* o, int H, int D, int pos,
                                      float theta, float scale, const float* cos_tab, const float* sin_tab,
                                      ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_decode_attn(static_cast<const bf16_t*>(qkv), static_cast<bf16_t*>(kcache),
                                   static_cast<bf16_t*>(vcache), static_cast<bf16_t*>(o), H, D, pos, theta, scale,
                                   ivlm_stream(stream), cos_tab, sin_tab, nullptr);
}

extern "C" int ivlm_llama_decode_attn_devpos(const void* qkv, void* kcache, void* vcache, void* o, int H, int D,
                                             const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                             const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    if (!pos_dev) return IVLM_ERR_INVALID_ARG;
    return ivlm::llama_decode_attn(static_cast<const bf16_t*>(qkv), static_cast<bf16_t*>(kcache),
                                   static_cast<bf16_t*>(vcache), static_cast<bf16_t*>(o), H, D, 0, theta, scale,
                                   ivlm_stream(stream), cos_tab, sin_tab, pos_dev);
}
