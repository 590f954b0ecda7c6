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

// ---- split-KV single-token attention (fp32 qkv / o; bf16 or fp16 cache) ------------------------------------------------------
// One 1024-thread block per head leaves 224 of the 256 CUs idle and walks its ~650 keys as a chain of dependent round trips
// (K tile, K tile, softmax, V tile): 6-8 us per layer.  Here the keys of a head are cut into S ranges (grid H x S, 256-thread
// blocks: 16 groups of 16 lanes, one key row per group and sweep, 96 keys per tile with K AND V of the tile in flight before
// anything else), every block computes the softmax of its range (local max m, local sum l, unnormalised o = sum p v) and
// publishes (o, m, l); the block that arrives LAST for a head (one agent-scope counter per head, reset by that block: no spin,
// no residency assumption) merges the S partials in range order: o = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s.  The result
// does not depend on which block arrives last.  RoPE of q is recomputed by every block (64 lanes); the new K / V row is appended by
// the block whose range holds the position.
namespace splitkv {
constexpr int kT = 256, kG = kT / 16, kU = 6, kTile = kG * kU;  // 96 keys per tile
constexpr int kMaxSplits = 16;
}  // namespace splitkv

// PARTS: the block only writes its (o, m, l) - rows of D + 4 floats, plain stores - and the CONSUMER merges them (the o_proj GEMV reads the
// S partials of a head while it stages its activation row: ivlm_gemv1_bf12m_parts; the kernel boundary is the synchronisation, no
// counter, no merge round trips).
template <bool CF16, bool PARTS = false>
__global__ __launch_bounds__(splitkv::kT) void llama_decode_attn_splitkv_kernel(
    const float* __restrict__ qkv, bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache, float* __restrict__ o, int H, int D,
    int pos_arg, float theta, float scale, const float* __restrict__ ct, const float* __restrict__ stab,
    const int32_t* __restrict__ pos_dev, int tmax, float* __restrict__ part, int32_t* __restrict__ counters) {
    using namespace splitkv;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const int h = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int t = threadIdx.x;
    const int pos = pos_dev ? __builtin_amdgcn_readfirstlane(*pos_dev) : pos_arg;
    if (pos >= tmax || pos >= kMaxT) {  // past the slab: nothing appended, zeros out (as the one-block kernel)
        if (PARTS) {  // (o = 0, m = 0, l = 1 for range 0 and l = 0 for the others: the merge gives exactly 0)
            float* mine = part + ((int64_t)h * S + sp) * (D + 4);
            if (t < D) mine[t] = 0.0f;
            else if (t == 128) { mine[D] = 0.0f; mine[D + 1] = sp == 0 ? 1.0f : 0.0f; }
        } else if (sp == 0 && t < D) {
            o[(int64_t)h * D + t] = 0.0f;
        }
        return;
    }
    const int nkeys = pos + 1;
    int per = (nkeys + S - 1) / S;
    per = ((per + kG - 1) / kG) * kG;
    const int k0 = sp * per, k1 = min(nkeys, k0 + per);  // this block's keys [k0, k1): empty for the last ranges of a short context
    const int len = k1 > k0 ? k1 - k0 : 0;
    __shared__ float q_s[kMaxD];
    __shared__ float knew_s[kMaxD];
    __shared__ float vnew_s[kMaxD];
    __shared__ float sc[kMaxT];
    __shared__ float red[2 * kT / 64];
    __shared__ float partl[kG][kMaxD];
    __shared__ int s_last;
    const int half = D >> 1;
    const int sub = t & 15, grp = t >> 4;
    const int nch = D >> 3;
    const int csub = sub < nch ? sub : nch - 1;
    const int64_t rstride = (int64_t)H * D;
    const bf16_t* kb = kcache + (int64_t)h * D + csub * 8;
    const bf16_t* vb = vcache + (int64_t)h * D + csub * 8;
    u32x4_t kr[kU], vr[kU];
    const int jmax = pos > 0 ? pos - 1 : 0;  // loads are clamped and unconditional, masked where used
    if (len > 0) {
#pragma unroll
        for (int i = 0; i < kU; ++i) {
            int j = k0 + grp + kG * i;
            j = j < jmax ? j : jmax;
            kr[i] = *reinterpret_cast<const u32x4_t*>(kb + j * rstride);
            vr[i] = *reinterpret_cast<const u32x4_t*>(vb + j * rstride);
        }
    }
    const bool owner = pos >= k0 && pos < k1;
    // ---- RoPE on q (every block) and on the new k (owner); append k, v ----------------------------------------------------------
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
        const float q0 = qkv[q + t], q1 = qkv[q + t + half];
        q_s[t] = q0 * c - q1 * s;
        q_s[t + half] = q1 * c + q0 * s;
        if (owner) {
            const float k0f = qkv[k + t], k1f = qkv[k + t + half];
            const float kaf = k0f * c - k1f * s, kbf = k1f * c + k0f * s;
            knew_s[t] = kaf;
            knew_s[t + half] = kbf;
            bf16_t* kc = kcache + ((int64_t)pos * H + h) * D;
            kc[t] = f32_to_h16<CF16>(kaf);
            kc[t + half] = f32_to_h16<CF16>(kbf);
        }
    } else if (owner && t >= 128 && t < 128 + D) {
        const int d = t - 128;
        const float v = qkv[2 * (int64_t)H * D + h * D + d];
        vnew_s[d] = v;
        vcache[((int64_t)pos * H + h) * D + d] = f32_to_h16<CF16>(v);
    }
    __syncthreads();
    // ---- scores of the range ----------------------------------------------------------------------------------------------------
    float qr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] = sub < nch ? q_s[sub * 8 + e] : 0.0f;
    auto score = [&](const u32x4_t& kv, int j) {
        float d = 0.0f;
        if (sub < nch) {
            if (j < pos) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d += pair_lo_f32<CF16>(kv[e]) * qr[2 * e];
                    d += pair_hi_f32<CF16>(kv[e]) * qr[2 * e + 1];
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
        if (sub == 0 && j < k1) sc[j - k0] = d * scale;
    };
    if (len > 0) {
#pragma unroll
        for (int i = 0; i < kU; ++i) score(kr[i], k0 + grp + kG * i);
        for (int j0 = k0 + kTile; j0 < k1; j0 += kTile) {  // longer ranges: further tiles
#pragma unroll
            for (int i = 0; i < kU; ++i) {
                int j = j0 + grp + kG * i;
                j = j < jmax ? j : jmax;
                kr[i] = *reinterpret_cast<const u32x4_t*>(kb + j * rstride);
            }
#pragma unroll
            for (int i = 0; i < kU; ++i) score(kr[i], j0 + grp + kG * i);
        }
    }
    __syncthreads();
    // ---- softmax of the range: local max, p = e^(s - m), local sum ----------------------------------------------------------------
    constexpr int NW = kT / 64;
    float mx = -1.0e30f;
    for (int j = t; j < len; j += kT) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.0f;
    for (int j = t; j < len; j += kT) {
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
    // ---- o = sum p v over the range (unnormalised) --------------------------------------------------------------------------------
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    auto pv = [&](const u32x4_t& vv, int j) {
        if (j < k1 && sub < nch) {
            const float p = sc[j - k0];
            if (j < pos) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += p * pair_lo_f32<CF16>(vv[e]);
                    acc[2 * e + 1] += p * pair_hi_f32<CF16>(vv[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += p * vnew_s[sub * 8 + e];
            }
        }
    };
    if (len > 0) {
#pragma unroll
        for (int i = 0; i < kU; ++i) pv(vr[i], k0 + grp + kG * i);
        for (int j0 = k0 + kTile; j0 < k1; j0 += kTile) {
#pragma unroll
            for (int i = 0; i < kU; ++i) {
                int j = j0 + grp + kG * i;
                j = j < jmax ? j : jmax;
                vr[i] = *reinterpret_cast<const u32x4_t*>(vb + j * rstride);
            }
#pragma unroll
            for (int i = 0; i < kU; ++i) pv(vr[i], j0 + grp + kG * i);
        }
    }
    if (sub < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) partl[grp][sub * 8 + e] = acc[e];
    }
    __syncthreads();
    // ---- publish (o, m, l); the last block of the head merges --------------------------------------------------------------------
    if (PARTS) {
        float* mine = part + ((int64_t)h * S + sp) * (D + 4);
        if (t < D) {
            float r = 0.0f;
#pragma unroll
            for (int g2 = 0; g2 < kG; ++g2) r += partl[g2][t];
            mine[t] = r;
        } else if (t == 128) {
            mine[D] = mx;
            mine[D + 1] = tot;
        }
        return;
    }
    float* mine = part + ((int64_t)h * S + sp) * (D + 2);
    if (t < D) {
        float r = 0.0f;
#pragma unroll
        for (int g2 = 0; g2 < kG; ++g2) r += partl[g2][t];
        __hip_atomic_store(mine + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (t == 128) {
        __hip_atomic_store(mine + D, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + D + 1, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's agent-scope stores are performed
    __syncthreads();
    if (t == 0) {
        const int old = __hip_atomic_fetch_add(counters + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == S - 1;
        if (last) __hip_atomic_store(counters + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        s_last = last;
    }
    __syncthreads();
    if (!s_last || t >= D) return;
    // all 3 S agent-scope loads in flight at once (a load per partial and round trip was most of the kernel's time)
    const float* base = part + (int64_t)h * S * (D + 2);
    float pm[kMaxSplits], pl[kMaxSplits], po[kMaxSplits];
#pragma unroll
    for (int s2 = 0; s2 < kMaxSplits; ++s2) {
        const float* ps = base + (int64_t)(s2 < S ? s2 : 0) * (D + 2);
        pm[s2] = __hip_atomic_load(ps + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pl[s2] = __hip_atomic_load(ps + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        po[s2] = __hip_atomic_load(ps + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float M = -1.0e30f;
#pragma unroll
    for (int s2 = 0; s2 < kMaxSplits; ++s2)
        if (s2 < S) M = fmaxf(M, pm[s2]);
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < kMaxSplits; ++s2) {
        if (s2 < S && pl[s2] > 0.0f) {  // (an empty range published m = -1e30, l = 0)
            const float w = __expf(pm[s2] - M);
            num += w * po[s2];
            den += w * pl[s2];
        }
    }
    o[(int64_t)h * D + t] = num / den;
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

size_t llama_decode_attn_splitkv_scratch_bytes(int H, int D) {
    if (H <= 0 || D <= 0) return 0;
    return 256 + (((size_t)H * 4 + 255) / 256) * 256 + (size_t)H * splitkv::kMaxSplits * (D + 2) * 4;
}

int g_splitkv_splits = 8;  // A/B hook: ivlm_llama_decode_attn_splits

// the PARTS form: S = 4 ranges (A/B: 2), partials [H][S][D + 4] fp32 for ivlm_gemv1_bf12m_parts
extern int g_decode_parts_S;
int llama_decode_attn_parts(const float* qkv, bf16_t* kcache, bf16_t* vcache, int tmax, float* parts, int H, int D, int pos, float theta,
                            float scale, hipStream_t st, const float* cos_tab, const float* sin_tab, const int32_t* pos_dev,
                            int cache_f16) {
    if (!qkv || !kcache || !vcache || !parts || H <= 0 || D <= 0 || D > kMaxD || (D & 15) || tmax <= 0) return IVLM_ERR_INVALID_ARG;
    if (!pos_dev && (pos < 0 || pos >= kMaxT || pos >= tmax)) return IVLM_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(parts) & 15) return IVLM_ERR_INVALID_ARG;
    const int S = g_decode_parts_S;
    if (cache_f16)
        llama_decode_attn_splitkv_kernel<true, true><<<dim3(H, S), splitkv::kT, 0, st>>>(qkv, kcache, vcache, nullptr, H, D, pos, theta, scale,
                                                                                         cos_tab, sin_tab, pos_dev, tmax, parts, nullptr);
    else
        llama_decode_attn_splitkv_kernel<false, true><<<dim3(H, S), splitkv::kT, 0, st>>>(qkv, kcache, vcache, nullptr, H, D, pos, theta, scale,
                                                                                          cos_tab, sin_tab, pos_dev, tmax, parts, nullptr);
    return ivlm_launch_status();
}

int llama_decode_attn_splitkv(const float* qkv, bf16_t* kcache, bf16_t* vcache, int tmax, float* o, int H, int D, int pos, float theta,
                              float scale, hipStream_t st, const float* cos_tab, const float* sin_tab, const int32_t* pos_dev,
                              int cache_f16, void* scratch, size_t scratch_bytes) {
    if (!qkv || !kcache || !vcache || !o || !scratch || H <= 0 || D <= 0 || D > kMaxD || (D & 15) || tmax <= 0) return IVLM_ERR_INVALID_ARG;
    if (!pos_dev && (pos < 0 || pos >= kMaxT || pos >= tmax)) return IVLM_ERR_INVALID_ARG;
    if (scratch_bytes < llama_decode_attn_splitkv_scratch_bytes(H, D) || (reinterpret_cast<uintptr_t>(scratch) & 15)) return IVLM_ERR_WORKSPACE;
    int32_t* counters = static_cast<int32_t*>(scratch);
    float* part = reinterpret_cast<float*>(static_cast<char*>(scratch) + (((size_t)H * 4 + 255) / 256) * 256);
    const int S = g_splitkv_splits;
    if (cache_f16)
        llama_decode_attn_splitkv_kernel<true><<<dim3(H, S), splitkv::kT, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab,
                                                                                   sin_tab, pos_dev, tmax, part, counters);
    else
        llama_decode_attn_splitkv_kernel<false><<<dim3(H, S), splitkv::kT, 0, st>>>(qkv, kcache, vcache, o, H, D, pos, theta, scale, cos_tab,
                                                                                    sin_tab, pos_dev, tmax, part, counters);
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

// split-KV variant of ivlm_llama_decode_attn (fp32 qkv / o; cache_dtype IVLM_BF16 or IVLM_F16): H x S blocks, partials merged by the
// last block of each head.  scratch: ivlm_llama_decode_attn_splitkv_scratch_bytes(H, D) bytes, ZEROED once by the caller (it holds
// the per-head arrival counters, which the kernel leaves at zero), not shared by launches that may run concurrently.
extern "C" size_t ivlm_llama_decode_attn_splitkv_scratch_bytes(int H, int D) { return ivlm::llama_decode_attn_splitkv_scratch_bytes(H, D); }

extern "C" int ivlm_llama_decode_attn_splitkv(const float* qkv, int cache_dtype, void* kcache, void* vcache, int tmax, float* o, int H,
                                              int D, int pos, const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                              const float* sin_tab, void* scratch, size_t scratch_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    if (cache_dtype != IVLM_BF16 && cache_dtype != IVLM_F16) return IVLM_ERR_INVALID_ARG;
    return ivlm::llama_decode_attn_splitkv(qkv, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), tmax, o, H, D, pos, theta,
                                           scale, ivlm_stream(stream), cos_tab, sin_tab, pos_dev, cache_dtype == IVLM_F16, scratch,
                                           scratch_bytes);
}

extern "C" int ivlm_llama_decode_attn_splits(int splits) {
    if (splits < 1 || splits > ivlm::splitkv::kMaxSplits) return IVLM_ERR_INVALID_ARG;
    ivlm::g_splitkv_splits = splits;
    return IVLM_OK;
}

// Split-KV attention WITHOUT the merge: four key ranges per head, partials parts[H][4][D + 4] fp32 (o unnormalised | max | sum | pad) for
// the o_proj GEMV that merges them while it stages its activation row (ivlm_gemv1_bf12m_parts).  parts: 16-byte aligned,
// H * 4 * (D + 4) floats; no other state.
extern "C" int ivlm_llama_decode_attn_parts(const float* qkv, int cache_dtype, void* kcache, void* vcache, int tmax, float* parts, int H,
                                            int D, int pos, const int32_t* pos_dev, float theta, float scale, const float* cos_tab,
                                            const float* sin_tab, ivlm_stream_t stream) {
    ivlm_enter();
    if (cache_dtype != IVLM_BF16 && cache_dtype != IVLM_F16) return IVLM_ERR_INVALID_ARG;
    return ivlm::llama_decode_attn_parts(qkv, static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), tmax, parts, H, D, pos, theta,
                                         scale, ivlm_stream(stream), cos_tab, sin_tab, pos_dev, cache_dtype == IVLM_F16);
}
