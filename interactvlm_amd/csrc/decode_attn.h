// Single-token LLaMA attention body (RoPE + KV append + softmax.V for one head), shared by decode.hip and decode_fused.hip.
#pragma once
#include "kernels.h"

namespace ivlm {
namespace decattn {

constexpr int kMaxD = 128;
constexpr int kMaxT = 4096;  // scores live in LDS (16 KB)

constexpr int kDecThreads = 1024, kDecGroups = kDecThreads / 16;  // 64 key rows per sweep

// body shared by the stand-alone kernel below and the fused attention + o_proj kernel (decode_fused.hip).
// COHERENT_OUT: the output row is written with agent-scope (sc1) stores, for consumers inside the same launch.
// F32IO: qkv and the output row are fp32 (the decode path keeps fp32 activations between its weight-streaming kernels: q and
// the softmax weights are then NOT rounded to bf16 - only the K/V rows appended to the bf16 cache are); otherwise bf16 in/out
// with the roundings of the MFMA prefill path (q, k after RoPE and P rounded to bf16).
// LO ("parity" precision, with F32IO): the cache holds K / V as hi + lo bf16 planes (kcache_lo / vcache_lo, same layout): the
// appended rows are not rounded to bf16 and the cached ones are read back as hi + lo.
// CF16 (with F32IO): the cache holds IEEE halves (the fp16-operand prefill appends them): appended rows are rounded to fp16, the
// cached ones are read as fp16.
template <bool COHERENT_OUT, bool F32IO = false, int THREADS = 1024, bool LO = false, bool CF16 = false>
__device__ __forceinline__ void llama_decode_attn_body(const int h, const void* __restrict__ qkv_v /*[3,H,D]*/,
                                                                bf16_t* __restrict__ kcache /*[Tmax,H,D]*/,
                                                                bf16_t* __restrict__ vcache, void* __restrict__ o_v,
                                                                int H, int D, int pos_arg, float theta, float scale,
                                                                const float* __restrict__ ct,
                                                                const float* __restrict__ stab,
                                                                const int32_t* __restrict__ pos_dev, int tmax = 0,
                                                                bf16_t* __restrict__ kcache_lo = nullptr,
                                                                bf16_t* __restrict__ vcache_lo = nullptr) {
    constexpr int kDecThreads = THREADS, kDecGroups = THREADS / 16;  // (shadow the namespace defaults)
    // position from device memory when given: lets one captured HIP graph serve every decode step
    const int pos = pos_dev ? __builtin_amdgcn_readfirstlane(*pos_dev) : pos_arg;
    (void)theta;
    // a sequence that has filled its cache slab (batched generation keeps stepping finished sequences) must not append
    if ((tmax > 0 && pos >= tmax) || pos >= kMaxT) return;
    __shared__ float q_s[kMaxD];
    __shared__ float knew_s[kMaxD];
    __shared__ float vnew_s[kMaxD];
    __shared__ float sc[kMaxT];
    __shared__ float red[2 * kDecThreads / 64];
    __shared__ float part[kDecGroups][kMaxD];
    const int t = threadIdx.x;
    const int half = D >> 1;
    // K and V rows of the first kTileKeys keys go in flight before anything else (they do not depend on q): the kernel is a
    // chain of dependent memory round trips otherwise (one per 64 keys).  16 lanes share a key row; group g owns keys
    // g, g + 64, ...
    constexpr int kU = 384 / kDecGroups;  // rows per group and tile: 384 keys per tile (6 K + 6 V chunks per lane at 1024 threads)
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const int sub = t & 15, grp = t >> 4;  // 64 groups of 16 lanes
    const int nch = D >> 3;                // 16-byte chunks per row (<= 16)
    const int csub = sub < nch ? sub : nch - 1;
    const int64_t rstride = (int64_t)H * D;
    const bf16_t* kb = kcache + (int64_t)h * D + csub * 8;
    const bf16_t* vb = vcache + (int64_t)h * D + csub * 8;
    const bf16_t* kbl = LO ? kcache_lo + (int64_t)h * D + csub * 8 : nullptr;
    const bf16_t* vbl = LO ? vcache_lo + (int64_t)h * D + csub * 8 : nullptr;
    u32x4_t kr[kU], vr[kU];
#pragma unroll
    for (int i = 0; i < kU; ++i) {
        int j = grp + kDecGroups * i;
        j = j < pos ? j : (pos > 0 ? pos - 1 : 0);  // clamped, unconditional; masked where used (pos == 0: row 0 is unused)
        kr[i] = *reinterpret_cast<const u32x4_t*>(kb + j * rstride);
        vr[i] = *reinterpret_cast<const u32x4_t*>(vb + j * rstride);
    }
    const bf16_t* qkv = static_cast<const bf16_t*>(qkv_v);
    const float* qkvf = static_cast<const float*>(qkv_v);
    auto ld = [&](int64_t e) -> float { return F32IO ? qkvf[e] : bf16_to_f32(qkv[e]); };
    // ---- RoPE on q and the new k; append k, v to the cache -------------------------------------
    if (t < half) {
        const int64_t q = (int64_t)h * D, k = (int64_t)H * D + h * D;
        float c, s;
        if (ct) {
            c = ct[pos * half + t];
            s = stab[pos * half + t];
        } else {
            const float ang = (float)pos * powf(theta, -(float)(2 * t) / (float)D);
            c = cosf(ang);
            s = sinf(ang);
        }
        const float q0 = ld(q + t), q1 = ld(q + t + half);
        const float k0 = ld(k + t), k1 = ld(k + t + half);
        const float qaf = q0 * c - q1 * s, qbf = q1 * c + q0 * s, kaf = k0 * c - k1 * s, kbf = k1 * c + k0 * s;
        // bf16 I/O: round q, k to bf16 exactly like the prefill path (rope_kv_kernel) so both paths see the same values
        const bf16_t qa = f32_to_bf16(qaf), qb = f32_to_bf16(qbf);
        const bf16_t ka = f32_to_h16<CF16>(kaf), kb = f32_to_h16<CF16>(kbf);
        q_s[t] = F32IO ? qaf : bf16_to_f32(qa);
        q_s[t + half] = F32IO ? qbf : bf16_to_f32(qb);
        knew_s[t] = F32IO ? kaf : bf16_to_f32(ka);
        knew_s[t + half] = F32IO ? kbf : bf16_to_f32(kb);
        static_assert(!CF16 || (F32IO && !LO), "fp16 cache: fp32 qkv / o, no lo planes");
        bf16_t* kc = kcache + ((int64_t)pos * H + h) * D;
        kc[t] = ka;
        kc[t + half] = kb;
        if (LO) {
            bf16_t* kcl = kcache_lo + ((int64_t)pos * H + h) * D;
            kcl[t] = f32_to_bf16(kaf - bf16_to_f32(ka));
            kcl[t + half] = f32_to_bf16(kbf - bf16_to_f32(kb));
        }
    } else if (t >= 128 && t < 128 + D) {
        const int d = t - 128;
        const float v = ld(2 * (int64_t)H * D + h * D + d);
        vnew_s[d] = v;
        const bf16_t vh = f32_to_h16<CF16>(v);
        vcache[((int64_t)pos * H + h) * D + d] = vh;
        if (LO) vcache_lo[((int64_t)pos * H + h) * D + d] = f32_to_bf16(v - bf16_to_f32(vh));
    }
    __syncthreads();
    // ---- scores ------------------------------------------------------------------------------------------------
    float qr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] = sub < nch ? q_s[sub * 8 + e] : 0.0f;
    const int nkeys = pos + 1;
    auto score = [&](const u32x4_t& kv, int j) {
        float d = 0.0f;
        if (sub < nch) {
            if (j < pos) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d += pair_lo_f32<CF16>(kv[e]) * qr[2 * e];
                    d += pair_hi_f32<CF16>(kv[e]) * qr[2 * e + 1];
                }
                if (LO) {
                    const u32x4_t kl = *reinterpret_cast<const u32x4_t*>(kbl + j * rstride);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d += __uint_as_float(kl[e] << 16) * qr[2 * e];
                        d += __uint_as_float(kl[e] & 0xffff0000u) * qr[2 * e + 1];
                    }
                }
            } else if (j == pos) {
#pragma unroll
                for (int e = 0; e < 8; ++e) d += knew_s[sub * 8 + e] * qr[e];
            }
        }
        d += __shfl_xor(d, 8, 64);
        d += __shfl_xor(d, 4, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 1, 64);
        if (sub == 0 && j < nkeys) sc[j] = d * scale;
    };
#pragma unroll
    for (int i = 0; i < kU; ++i) score(kr[i], grp + kDecGroups * i);
    for (int j0 = kDecGroups * kU; j0 < nkeys; j0 += kDecGroups * kU) {  // longer contexts: further tiles
#pragma unroll
        for (int i = 0; i < kU; ++i) {
            int j = j0 + grp + kDecGroups * i;
            j = j < pos ? j : pos - 1;
            kr[i] = *reinterpret_cast<const u32x4_t*>(kb + j * rstride);
        }
#pragma unroll
        for (int i = 0; i < kU; ++i) score(kr[i], j0 + grp + kDecGroups * i);
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
    // ---- O = P.V: group grp owns keys j == grp (mod 64), lane sub owns 8 dims (V rows of tile 0 already loaded) ----
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    auto pv = [&](const u32x4_t& vv, int j) {
        if (j < nkeys && sub < nch) {
            const float p = F32IO ? sc[j] * inv_sum : bf16_to_f32(f32_to_bf16(sc[j] * inv_sum));
            if (j < pos) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += p * pair_lo_f32<CF16>(vv[e]);
                    acc[2 * e + 1] += p * pair_hi_f32<CF16>(vv[e]);
                }
                if (LO) {
                    const u32x4_t vl = *reinterpret_cast<const u32x4_t*>(vbl + j * rstride);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] += p * __uint_as_float(vl[e] << 16);
                        acc[2 * e + 1] += p * __uint_as_float(vl[e] & 0xffff0000u);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += p * vnew_s[sub * 8 + e];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < kU; ++i) pv(vr[i], grp + kDecGroups * i);
    for (int j0 = kDecGroups * kU; j0 < nkeys; j0 += kDecGroups * kU) {
#pragma unroll
        for (int i = 0; i < kU; ++i) {
            int j = j0 + grp + kDecGroups * i;
            j = j < pos ? j : pos - 1;
            vr[i] = *reinterpret_cast<const u32x4_t*>(vb + j * rstride);
        }
#pragma unroll
        for (int i = 0; i < kU; ++i) pv(vr[i], j0 + grp + kDecGroups * i);
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
        if (F32IO) {
            float* o = static_cast<float*>(o_v);
            if (COHERENT_OUT) __hip_atomic_store(o + h * D + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else o[h * D + t] = r;
        } else {
            bf16_t* o = static_cast<bf16_t*>(o_v);
            if (COHERENT_OUT) __hip_atomic_store(o + h * D + t, f32_to_bf16(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else o[h * D + t] = f32_to_bf16(r);
        }
    }
}


}  // namespace decattn
}  // namespace ivlm
