// Persistent greedy-decode kernel for the LLaMA path: ALL layers of ALL new tokens in ONE launch.
//
// What it replaces: HF greedy search under InteractVLM.evaluate (model/InteractVLM.py:524-531) -> per token
// LlamaModel.forward (32 x [RMSNorm, q|k|v, RoPE, attention, o_proj, RMSNorm, gate|up, SwiGLU, down]) + final
// RMSNorm + lm_head + argmax + embed_tokens of the chosen id.  Batch-1 decode is pure weight streaming (13.5 GB of
// bf16 per token for 7B); as separate launches each GEMV pays a ramp, a tail and a row-quantisation loss that add up
// to ~40 % of the step.  Here one workgroup per CU stays resident and walks the phases of every layer:
//
//   * every weight matrix is cut into G contiguous row slabs (G = resident workgroups); a block streams its slab as
//     ONE flat, perfectly coalesced byte range - lane t owns 16-byte chunks t, t+512, ... - so all lanes of all CUs
//     carry the same load (no wave-per-row quantisation), 16 x 16 B in flight per lane;
//   * the activation vector (x * gamma for the RMS-fused phases) is staged once per phase in LDS; per 64-lane step
//     the partial dot is wave-reduced (a step touches at most two rows) into an LDS slot, and the rows are summed
//     in a fixed order afterwards: bit-reproducible;
//   * phases are separated by a device-wide barrier (one atomic counter).  The first 16 loads per lane of the NEXT
//     phase's slab are issued BEFORE arriving at the barrier, so HBM keeps streaming through it (32 MB in flight);
//   * activations crossing the barrier (<= 27 KB) are written/read with agent-scope (sc1) accesses, so no bulk L2
//     write-back / invalidate is needed; the KV cache of a head is only ever touched by the block that owns the head;
//   * lm_head + argmax (first-index tie rule) + the EOS / forced-token logic run on the device: the host reads the
//     generated ids once, after the single launch.
//
// Barrier waits are bounded (2 s of wall clock) and report through `status`: a scheduling problem can never hang the GPU.
#include <stdlib.h>

#include <algorithm>

#include "slab_stream.h"

namespace ivlm {
namespace {

using namespace slabk;

struct GenArgs {
    const int64_t* layer_ptrs;  // device [L][6] addresses: ln1, qkv[3*hidden,hidden], o, ln2, gu[2*inter,hidden] (gate/up rows interleaved), down
    int L, H, D, hidden, inter, vocab;
    float eps, scale;
    const float* cos_tab;  // [max_len, D/2]
    const float* sin_tab;
    bf16_t* kcache;  // [L, max_len, H, D]
    bf16_t* vcache;
    int64_t cache_layer_stride;
    const bf16_t* embed;       // [vocab, hidden]
    const bf16_t* final_norm;  // [hidden]
    const bf16_t* lm_head;     // [vocab, hidden]
    bf16_t* hidden_out;        // [>= pos0 + n_max - 1, hidden]; row pos0-1 is the input (last prefill hidden)
    int pos0, n_max, eos;
    const int32_t* forced;  // [n_max] or null
    int32_t* new_ids;       // [n_max]
    int32_t* argmax_ids;    // [n_max]
    int32_t* status;        // [0] tokens generated, [1] error flag (1 = barrier timeout)
    unsigned* barrier;      // zeroed before launch
    bf16_t *qkv_buf, *attn_buf, *x2_buf, *xa_buf, *xb_buf, *h_buf;
    float* cand_val;    // [G]
    int32_t* cand_idx;  // [G]
    long long* trace;   // optional (IVLM_GEN_TRACE=1): block 0 timestamps, [0] = count, then (tag, wall_clock64) pairs
};

constexpr int kTraceMax = 1000;
__device__ __forceinline__ void trace_mark(const GenArgs& a, int tag, int& n) {
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0 && n < kTraceMax) {  // stores only: no round trip
        a.trace[1 + 2 * n] = tag;
        a.trace[2 + 2 * n] = wall_clock64();
        a.trace[0] = ++n;
    }
}

// Device-wide barrier.  Callers drain their stores (drain_stores) BEFORE issuing the next phase's prefetch and arriving.
// Two levels, so that no address sees more than G/8 (or 8) read-modify-writes per barrier: blocks arrive on the counter
// of their group (blockIdx % 8 - the XCD under round-robin dispatch; correctness does not depend on that), the last
// arriver of a group arrives on the top counter, the last of those publishes the epoch to one flag line per group,
// and everybody polls (plain agent-scope loads with back-off) only its own group's flag.
//   words: [0..7]*32 group counters, [8*32] top counter, [(9+g)*32] group flags   (128-byte lines)
// Split in two so that the round trip hides behind useful work: arrive() right after this block's outputs are performed,
// then the caller issues the next slab's prefetch (which stalls on the memory queues for microseconds anyway), then
// wait() - by which time the other blocks have normally arrived as well.
__device__ __forceinline__ void barrier_arrive(const GenArgs& a, unsigned& epoch, int G) {
    block_sync_lds();  // every storing wave of this block has drained its stores
    epoch += 1u;
    if (threadIdx.x == 0) {
        const int grp = blockIdx.x & 7;
        const unsigned gsize = (unsigned)((G - grp + 7) >> 3);
        unsigned* bar = a.barrier;
        const unsigned prev = __hip_atomic_fetch_add(bar + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == epoch * gsize) {
            const unsigned ngrp = (unsigned)(G < 8 ? G : 8);
            const unsigned ptop = __hip_atomic_fetch_add(bar + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ptop + 1u == epoch * ngrp) {
                for (unsigned g2 = 0; g2 < ngrp; ++g2)
                    __hip_atomic_store(bar + (9 + g2) * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__device__ __forceinline__ bool barrier_wait(const GenArgs& a, unsigned epoch, Lds& L) {
    if (threadIdx.x == 0) {
        const int grp = blockIdx.x & 7;
        unsigned* bar = a.barrier;
        const long long t0 = wall_clock64();
        int dead = 0;
        while (__hip_atomic_load(bar + (9 + grp) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > kTimeoutTicks) {
                __hip_atomic_store(a.status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = 1;
                break;
            }
        }
        if (!dead && __hip_atomic_load(a.status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) dead = 1;
        L.flag[0] = dead;
    }
    block_sync_lds();
    return L.flag[0] == 0;
}

__device__ __forceinline__ bool grid_barrier(const GenArgs& a, unsigned& epoch, int G, Lds& L) {
    barrier_arrive(a, epoch, G);
    return barrier_wait(a, epoch, L);
}

// final LlamaRMSNorm exactly as norm.hip: y = bf16(bf16(x * rstd) * w); staged as the lm_head input and (block 0)
// stored as the hidden state of this position.
__device__ __forceinline__ void stage_final_norm(const bf16_t* x, const GenArgs& a, bf16_t* hidden_row, Lds& L) {
    uint32_t* xs32 = reinterpret_cast<uint32_t*>(L.xs);
    const int nd = a.hidden >> 1;
    float ssq = 0.0f;
    for (int i = threadIdx.x; i < nd; i += kThreads) {
        const uint32_t v = ld_agent(reinterpret_cast<const uint32_t*>(x) + i);
        xs32[i] = v;
        const float lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
        ssq += lo * lo + hi * hi;
    }
    ssq = wave_sum(ssq);
    if ((threadIdx.x & 63) == 0) L.red[threadIdx.x >> 6] = ssq;
    block_sync_lds();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) tot += L.red[w];
    const float rstd = rsqrtf(tot / (float)a.hidden + a.eps);
    for (int i = threadIdx.x; i < nd; i += kThreads) {
        const uint32_t v = xs32[i];
        const uint32_t gv = as_gword(a.final_norm)[i];
        const float lo = bf16_to_f32(f32_to_bf16(__uint_as_float(v << 16) * rstd)) * __uint_as_float(gv << 16);
        const float hi = bf16_to_f32(f32_to_bf16(__uint_as_float(v & 0xffff0000u) * rstd)) * __uint_as_float(gv & 0xffff0000u);
        const uint32_t y = pack_bf16x2(lo, hi);
        xs32[i] = y;
        if (hidden_row && blockIdx.x == 0) reinterpret_cast<uint32_t*>(hidden_row)[i] = y;
    }
    block_sync_lds();
}

// One head of single-token attention (same arithmetic as llama_decode_attn_kernel in decode.hip): RoPE(q, k) with
// bf16 rounding, KV append, fp32 softmax, probabilities rounded to bf16 before P.V.
// Latency-bound (~180 KB of cache per head): 16 lanes share a key row, a group of 16 lanes owns keys grp, grp+32, ...;
// the K rows AND the V rows of a tile of kTile keys per group are all put in flight before anything waits on them
// (the K loads do not depend on q): one memory round trip per 384 keys instead of one per 32.
constexpr int kTile = 12;  // keys per 16-lane group and tile: 12 K + 12 V chunks (96 VGPRs) in flight per lane

struct AttnArgs1 {
    bf16_t* kcache;  // this layer
    bf16_t* vcache;
    const bf16_t* qkv;
    bf16_t* out;
    const float* cos_tab;
    const float* sin_tab;
    int H, D;
    float scale;
};

// Deliberately NOT inlined: the K/V tiles need half the register file, and as a separate function the allocator deals
// with them on their own instead of fighting the streaming state of the caller (nothing of which is live here).
typedef __attribute__((address_space(3))) float lds_f32;  // explicit LDS pointers: a generic pointer cannot cross the call

__device__ __attribute__((noinline)) void attention_head(AttnArgs1 a, int h, int pos, lds_f32* lds) {
    constexpr int kGroups = kThreads / 16;
    lds_f32* q_s = lds;
    lds_f32* knew_s = q_s + kMaxHeadDim;
    lds_f32* vnew_s = knew_s + kMaxHeadDim;
    lds_f32* red = vnew_s + kMaxHeadDim;        // [2*kWaves]
    lds_f32* sc = red + 2 * kWaves;             // [kMaxPos]
    lds_f32* part = sc + kMaxPos;               // [kGroups][kMaxHeadDim]
    const int t = threadIdx.x, H = a.H, D = a.D, half = D >> 1;
    bf16_t* kcache = a.kcache;
    bf16_t* vcache = a.vcache;
    const bf16_t* qkv = a.qkv;
    const int sub = t & 15, grp = t >> 4;
    const int nch = D >> 3;  // 16-byte chunks per row (<= 16)
    const int csub = sub < nch ? sub : nch - 1;
    const int nkeys = pos + 1;
    const int64_t rstride = (int64_t)H * D;
    const bf16_t* kbase = kcache + (int64_t)h * D + csub * 8;
    const bf16_t* vbase = vcache + (int64_t)h * D + csub * 8;

    // ---- tile 0: K and V rows in flight first -----------------------------------------------------------------
    u32x4_t kr[kTile], vr[kTile];
#pragma unroll
    for (int i = 0; i < kTile; ++i) {
        int j = grp + kGroups * i;
        j = j < pos ? j : (pos > 0 ? pos - 1 : 0);  // clamped (unconditional) loads; masked when used
        kr[i] = *reinterpret_cast<const u32x4_t*>(kbase + j * rstride);
        vr[i] = *reinterpret_cast<const u32x4_t*>(vbase + j * rstride);
    }
    // ---- RoPE on q and the new k; append k, v to the cache -------------------------------------------------------
    if (t < half) {
        const bf16_t* q = qkv + h * D;
        const bf16_t* k = qkv + (int64_t)H * D + h * D;
        const float c = a.cos_tab[pos * half + t], s = a.sin_tab[pos * half + t];
        const float q0 = bf16_to_f32(ld_agent16(q + t)), q1 = bf16_to_f32(ld_agent16(q + t + half));
        const float k0 = bf16_to_f32(ld_agent16(k + t)), k1 = bf16_to_f32(ld_agent16(k + t + half));
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
        const bf16_t v = ld_agent16(qkv + 2 * (int64_t)H * D + h * D + d);
        vnew_s[d] = bf16_to_f32(v);
        vcache[((int64_t)pos * H + h) * D + d] = v;
    }
    block_sync_lds();
    float qr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] = sub < nch ? q_s[sub * 8 + e] : 0.0f;

    // ---- scores ---------------------------------------------------------------------------------------------------
    auto score = [&](const u32x4_t& kv, int j) {
        float d = 0.0f;
        if (sub < nch) {
            if (j < pos) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d += __uint_as_float(kv[e] << 16) * qr[2 * e];
                    d += __uint_as_float(kv[e] & 0xffff0000u) * qr[2 * e + 1];
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
        if (sub == 0 && j < nkeys) sc[j] = d * a.scale;
    };
#pragma unroll
    for (int i = 0; i < kTile; ++i) score(kr[i], grp + kGroups * i);
    for (int j0 = kGroups * kTile; j0 < nkeys; j0 += kGroups * kTile) {  // contexts longer than one tile
#pragma unroll
        for (int i = 0; i < kTile; ++i) {
            int j = j0 + grp + kGroups * i;
            j = j < pos ? j : pos - 1;
            kr[i] = *reinterpret_cast<const u32x4_t*>(kbase + j * rstride);
        }
#pragma unroll
        for (int i = 0; i < kTile; ++i) score(kr[i], j0 + grp + kGroups * i);
    }
    block_sync_lds();
    // ---- softmax over sc[0..pos] (fp32) ---------------------------------------------------------------------------
    float mx = -1.0e30f;
    for (int j = t; j < nkeys; j += kThreads) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    block_sync_lds();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.0f;
    for (int j = t; j < nkeys; j += kThreads) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[kWaves + (t >> 6)] = sum;
    block_sync_lds();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) tot += red[kWaves + w];
    const float inv_sum = 1.0f / tot;
    // ---- O = P.V (HF: softmax in fp32, cast to bf16, then @ V) ------------------------------------------------------
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    auto pv = [&](const u32x4_t& vv, int j) {
        if (j < nkeys && sub < nch) {
            const float p = bf16_to_f32(f32_to_bf16(sc[j] * inv_sum));
            if (j < pos) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += p * __uint_as_float(vv[e] << 16);
                    acc[2 * e + 1] += p * __uint_as_float(vv[e] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += p * vnew_s[sub * 8 + e];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < kTile; ++i) pv(vr[i], grp + kGroups * i);
    for (int j0 = kGroups * kTile; j0 < nkeys; j0 += kGroups * kTile) {
#pragma unroll
        for (int i = 0; i < kTile; ++i) {
            int j = j0 + grp + kGroups * i;
            j = j < pos ? j : pos - 1;
            vr[i] = *reinterpret_cast<const u32x4_t*>(vbase + j * rstride);
        }
#pragma unroll
        for (int i = 0; i < kTile; ++i) pv(vr[i], j0 + grp + kGroups * i);
    }
    if (sub < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[grp * kMaxHeadDim + sub * 8 + e] = acc[e];
    }
    block_sync_lds();
    if (t < D) {
        float r = 0.0f;
#pragma unroll 8
        for (int g2 = 0; g2 < kGroups; ++g2) r += part[g2 * kMaxHeadDim + t];
        st_agent16(a.out + h * D + t, f32_to_bf16(r));
    }
    block_sync_lds();
}

__global__ __launch_bounds__(kThreads, 1) void llama_generate_kernel(GenArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int G = gridDim.x, b = blockIdx.x;
    const int kmax = a.hidden > a.inter ? a.hidden : a.inter;
    Lds L;
    L.xs = reinterpret_cast<u32x4_t*>(smem);
    L.part = reinterpret_cast<float*>(smem + (size_t)kmax * 2);
    L.rowsum = L.part + kMaxSteps * kWaves * 2;
    L.red = L.rowsum + kMaxRows;
    L.attn = smem;  // overlay
    // the flag words must not be clobbered by the attention overlay: they sit behind both regions
    {
        const size_t attn_bytes = (size_t)(3 * kMaxHeadDim + 2 * kWaves + kMaxPos + (kThreads / 16) * kMaxHeadDim) * 4;
        const size_t gemv_bytes = (size_t)kmax * 2 + (size_t)(kMaxSteps * kWaves * 2 + kMaxRows + 2 * kWaves) * 4;
        const size_t off = attn_bytes > gemv_bytes ? attn_bytes : gemv_bytes;
        L.flag = reinterpret_cast<int*>(smem + ((off + 15) & ~(size_t)15));
    }
    const int t = threadIdx.x;
    unsigned epoch = 0;
    int ntrace = 0;
    u32x4_t buf[kDepth];

    // One state machine, ONE instance of the streaming code.  Phase p of a token: 0 = lm_head (+argmax),
    // 1 + 4*l + {0: q|k|v, 1: o_proj, 2: gate|up, 3: down} for layer l.
    int r0, r1;
    slab(a.vocab, 1, b, G, r0, r1);
    prefetch(buf, a.lm_head, a.hidden, r0, r1);
    // input of step 0: the (already final-normed) hidden state of the last prefill position
    stage_vec(a.hidden_out + (int64_t)(a.pos0 - 1) * a.hidden, nullptr, a.hidden, 0.0f, false, false, L);

    // Blocks 0..H-1 run the attention of one head each.  When there are plenty of other blocks they get no o_proj slab:
    // their registers are then free for the K/V rows during attention (nothing prefetched is live across it), and the
    // o_proj rows (3 % of the bytes) are spread over the remaining blocks.
    const int attn_skip = G >= 4 * a.H ? a.H : 0;
    int pos = a.pos0, step = 0, p = 0;
    const int nphase = 1 + 4 * a.L;
    const bf16_t* x_cur = nullptr;  // residual stream entering the current layer
    bool x_coherent = false;
    while (true) {
        const int l = p > 0 ? (p - 1) >> 2 : 0, ph = p > 0 ? (p - 1) & 3 : -1;
        const int64_t* lp = a.layer_ptrs + (int64_t)l * 6;
        bf16_t* x_next = (l & 1) ? a.xb_buf : a.xa_buf;
        // ---- describe + stage ---------------------------------------------------------------------------
        const bf16_t* W;
        int K = a.hidden;
        float rstd = 1.0f;
        if (ph < 0) {
            W = a.lm_head;  // xs already holds the final-normed hidden state
        } else if (ph == 0) {
            W = reinterpret_cast<const bf16_t*>(lp[1]);
            rstd = stage_vec(x_cur, reinterpret_cast<const bf16_t*>(lp[0]), a.hidden, a.eps, true, x_coherent, L);
        } else if (ph == 1) {
            W = reinterpret_cast<const bf16_t*>(lp[2]);
            stage_vec(a.attn_buf, nullptr, a.hidden, 0.0f, false, true, L);
        } else if (ph == 2) {
            W = reinterpret_cast<const bf16_t*>(lp[4]);
            rstd = stage_vec(a.x2_buf, reinterpret_cast<const bf16_t*>(lp[3]), a.hidden, a.eps, true, true, L);
        } else {
            W = reinterpret_cast<const bf16_t*>(lp[5]);
            K = a.inter;
            stage_vec(a.h_buf, nullptr, a.inter, 0.0f, false, true, L);
        }
        // residual element of this thread's output row: fetched now, off the critical path of the epilogue
        float res = 0.0f;
        if (ph == 1 && t < r1 - r0) res = bf16_to_f32(x_coherent ? ld_agent16(x_cur + r0 + t) : x_cur[r0 + t]);
        if (ph == 3 && t < r1 - r0) res = bf16_to_f32(ld_agent16(a.x2_buf + r0 + t));
        trace_mark(a, 10 + ph, ntrace);  // staged
        // ---- stream this block's slab ---------------------------------------------------------------------
        stream_slab(buf, W, K, r0, r1, L);
        trace_mark(a, 20 + ph, ntrace);  // streamed
        // ---- epilogue ---------------------------------------------------------------------------------------
        const int nrows = r1 - r0;
        if (ph < 0) {  // local argmax candidate (first index wins ties)
            if (t < 64) {
                float best = -INFINITY;
                int bi = 0x7fffffff;
                for (int i = t; i < nrows; i += 64) {
                    const float v = L.rowsum[i];
                    if (v > best || (v == best && r0 + i < bi)) {
                        best = v;
                        bi = r0 + i;
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > best || (ov == best && oi < bi)) {
                        best = ov;
                        bi = oi;
                    }
                }
                if (t == 0) {
                    st_agent_f(a.cand_val + b, best);
                    st_agent_i(a.cand_idx + b, bi);
                }
            }
        } else if (ph == 0) {
            if (t < nrows) st_agent16(a.qkv_buf + r0 + t, f32_to_bf16(L.rowsum[t] * rstd));
        } else if (ph == 1) {
            if (t < nrows) st_agent16(a.x2_buf + r0 + t, f32_to_bf16(L.rowsum[t] + res));
        } else if (ph == 2) {  // SiLU(gate) * up over row-interleaved (gate_j, up_j) weights
            if (t < (nrows >> 1)) {
                const float v0 = L.rowsum[2 * t] * rstd, v1 = L.rowsum[2 * t + 1] * rstd;
                st_agent16(a.h_buf + (r0 >> 1) + t, f32_to_bf16((v0 / (1.0f + __expf(-v0))) * v1));
            }
        } else {
            if (t < nrows) st_agent16(x_next + r0 + t, f32_to_bf16(L.rowsum[t] + res));
        }
        // only the waves that stored wait for their stores to be performed; the others go straight on to the prefetch
        if ((t & ~63) < nrows) drain_stores();
        // ---- arrive, prefetch the next phase's slab while the barrier completes, then wait -----------------------
        barrier_arrive(a, epoch, G);
        const int pn = p + 1 < nphase ? p + 1 : 0;
        {
            const int ln = pn > 0 ? (pn - 1) >> 2 : 0, phn = pn > 0 ? (pn - 1) & 3 : -1;
            const int64_t* lpn = a.layer_ptrs + (int64_t)ln * 6;
            const bf16_t* Wn;
            int Kn = a.hidden;
            if (phn < 0) { Wn = a.lm_head; slab(a.vocab, 1, b, G, r0, r1); }
            else if (phn == 0) { Wn = reinterpret_cast<const bf16_t*>(lpn[1]); slab(3 * a.hidden, 1, b, G, r0, r1); }
            else if (phn == 1) { Wn = reinterpret_cast<const bf16_t*>(lpn[2]); slab(a.hidden, 1, b, G, r0, r1, attn_skip); }
            else if (phn == 2) { Wn = reinterpret_cast<const bf16_t*>(lpn[4]); slab(2 * a.inter, 2, b, G, r0, r1); }
            else { Wn = reinterpret_cast<const bf16_t*>(lpn[5]); Kn = a.inter; slab(a.hidden, 1, b, G, r0, r1); }
            if (!(phn == 1 && b < a.H)) prefetch(buf, Wn, Kn, r0, r1);  // attention blocks: after the attention
        }
        trace_mark(a, 30 + ph, ntrace);  // epilogue stored, arrived, next slab prefetched
        if (!barrier_wait(a, epoch, L)) return;
        trace_mark(a, 40 + ph, ntrace);  // barrier passed
        // ---- after the barrier --------------------------------------------------------------------------------
        if (ph < 0) {  // global argmax (every block, identically), EOS / forced-token logic
            if (t < 64) {
                float best = -INFINITY;
                int bi = 0x7fffffff;
                for (int i = t; i < G; i += 64) {
                    const float v = ld_agent_f(a.cand_val + i);
                    const int idx = (int)ld_agent(a.cand_idx + i);
                    if (v > best || (v == best && idx < bi)) {
                        best = v;
                        bi = idx;
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > best || (ov == best && oi < bi)) {
                        best = ov;
                        bi = oi;
                    }
                }
                if (t == 0) {
                    const int tk = a.forced ? a.forced[step] : bi;
                    L.flag[1] = tk;
                    if (b == 0) {
                        a.argmax_ids[step] = bi;
                        a.new_ids[step] = tk;
                    }
                }
            }
            block_sync_lds();
            const int tok = L.flag[1];
            if (tok == a.eos || step == a.n_max - 1) {
                if (b == 0 && t == 0) a.status[0] = step + 1;
                return;
            }
            x_cur = a.embed + (int64_t)tok * a.hidden;  // embed_tokens row: read-only, no coherence needed
            x_coherent = false;
        } else if (ph == 0) {  // attention, one block per head, then a second barrier
            if (b < a.H) {
                AttnArgs1 aa;
                aa.kcache = a.kcache + (int64_t)l * a.cache_layer_stride;
                aa.vcache = a.vcache + (int64_t)l * a.cache_layer_stride;
                aa.qkv = a.qkv_buf;
                aa.out = a.attn_buf;
                aa.cos_tab = a.cos_tab;
                aa.sin_tab = a.sin_tab;
                aa.H = a.H;
                aa.D = a.D;
                aa.scale = a.scale;
                for (int h = b; h < a.H; h += G) attention_head(aa, h, pos, (lds_f32*)smem);
                // nothing prefetched is live across the attention (its K/V tiles need the registers); this block's
                // o_proj slab is empty when attn_skip is active (every lane then just reads the zero chunk)
                prefetch(buf, reinterpret_cast<const bf16_t*>(lp[2]), a.hidden, r0, r1);
            }
            drain_stores();  // (also lands the W_o prefetch, which the next phase consumes first anyway)
            trace_mark(a, 50, ntrace);  // attention done
            if (!grid_barrier(a, epoch, G, L)) return;
            trace_mark(a, 51, ntrace);
        } else if (ph == 3) {
            x_cur = x_next;
            x_coherent = true;
            if (l == a.L - 1) {  // final RMSNorm -> hidden state of this position = next lm_head input
                stage_final_norm(x_cur, a, a.hidden_out + (int64_t)pos * a.hidden, L);
                ++pos;
                ++step;
            }
        }
        p = pn;
    }
}

}  // namespace

size_t llama_generate_workspace_bytes(int hidden, int inter) {
    // status[4] + barrier[4] + cand (2 x 1024) + qkv/attn/x2/xa/xb/h vectors, 256-byte aligned pieces
    return 4096 + 2 * 4096 + (size_t)(3 * hidden + 4 * hidden + inter) * 2 + 8 * 256 + (size_t)(2 * kTraceMax + 2) * 8;
}

int llama_generate(const int64_t* layer_ptrs, int L, int H, int D, int hidden, int inter, int vocab, float eps, float scale,
                   const float* cos_tab, const float* sin_tab, bf16_t* kcache, bf16_t* vcache, int64_t cache_layer_stride,
                   int max_len, const bf16_t* embed, const bf16_t* final_norm, const bf16_t* lm_head, bf16_t* hidden_out,
                   int pos0, int n_max, int eos, const int32_t* forced, int32_t* new_ids, int32_t* argmax_ids,
                   void* workspace, size_t ws_bytes, hipStream_t st) {
    if (!layer_ptrs || !cos_tab || !sin_tab || !kcache || !vcache || !embed || !final_norm || !lm_head || !hidden_out ||
        !new_ids || !argmax_ids || !workspace)
        return IVLM_ERR_INVALID_ARG;
    if (L <= 0 || H <= 0 || D <= 0 || D > kMaxHeadDim || (D & 15) || H * D != hidden || n_max <= 0 || pos0 < 1)
        return IVLM_ERR_INVALID_ARG;
    if (hidden < 512 || inter < 512 || (hidden & 7) || (inter & 7)) return IVLM_ERR_UNSUPPORTED;  // >= 64 chunks per row
    if (pos0 + n_max - 1 > max_len || pos0 + n_max - 1 > kMaxPos) return IVLM_ERR_INVALID_ARG;
    if (ws_bytes < llama_generate_workspace_bytes(hidden, inter)) return IVLM_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return IVLM_ERR_INVALID_ARG;
    int dev = 0, cus = 0;
    IVLM_HIP_TRY(hipGetDevice(&dev));
    IVLM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int G = cus;
    if (G > 1024) G = 1024;
    if (H > G) return IVLM_ERR_UNSUPPORTED;
    // slab sizes must fit the LDS row / step tables
    auto rows_of = [&](int N, int unit) { return ((N / unit + G - 1) / G) * unit + unit; };
    auto steps_of = [&](int N, int unit, int K) { return (int)(((int64_t)rows_of(N, unit) * (K >> 3) + kThreads - 1) / kThreads); };
    const int max_rows = std::max(std::max(rows_of(vocab, 1), rows_of(3 * hidden, 1)), rows_of(2 * inter, 2));
    const int max_steps = std::max(std::max(steps_of(vocab, 1, hidden), steps_of(3 * hidden, 1, hidden)),
                                   std::max(steps_of(2 * inter, 2, hidden), steps_of(hidden, 1, inter)));
    if (max_rows > kMaxRows || max_steps + kDepth > kMaxSteps) return IVLM_ERR_UNSUPPORTED;

    GenArgs a;
    a.layer_ptrs = layer_ptrs;
    a.L = L; a.H = H; a.D = D; a.hidden = hidden; a.inter = inter; a.vocab = vocab;
    a.eps = eps; a.scale = scale;
    a.cos_tab = cos_tab; a.sin_tab = sin_tab;
    a.kcache = kcache; a.vcache = vcache; a.cache_layer_stride = cache_layer_stride;
    a.embed = embed; a.final_norm = final_norm; a.lm_head = lm_head;
    a.hidden_out = hidden_out;
    a.pos0 = pos0; a.n_max = n_max; a.eos = eos;
    a.forced = forced; a.new_ids = new_ids; a.argmax_ids = argmax_ids;
    unsigned char* w = static_cast<unsigned char*>(workspace);
    a.status = reinterpret_cast<int32_t*>(w);
    a.barrier = reinterpret_cast<unsigned*>(w + 256);  // 17 lines of 128 B: 256 .. 2432
    a.cand_val = reinterpret_cast<float*>(w + 4096);
    a.cand_idx = reinterpret_cast<int32_t*>(w + 4096 + 4096);
    size_t off = 4096 + 2 * 4096;
    auto take = [&](size_t elems) {
        bf16_t* p = reinterpret_cast<bf16_t*>(w + off);
        off += (elems * 2 + 255) & ~(size_t)255;
        return p;
    };
    a.qkv_buf = take(3 * (size_t)hidden);
    a.attn_buf = take(hidden);
    a.x2_buf = take(hidden);
    a.xa_buf = take(hidden);
    a.xb_buf = take(hidden);
    a.h_buf = take(inter);
    a.trace = nullptr;
    if (getenv("IVLM_GEN_TRACE")) {
        a.trace = reinterpret_cast<long long*>(w + off);
        IVLM_HIP_TRY(hipMemsetAsync(a.trace, 0, 8, st));
    }
    IVLM_HIP_TRY(hipMemsetAsync(w, 0, 4096, st));

    const int kmax = std::max(hidden, inter);
    const size_t attn_bytes = (size_t)(3 * kMaxHeadDim + 2 * kWaves + kMaxPos + (kThreads / 16) * kMaxHeadDim) * 4;
    const size_t gemv_bytes = (size_t)kmax * 2 + (size_t)(kMaxSteps * kWaves * 2 + kMaxRows + 2 * kWaves) * 4;
    size_t lds = ((std::max(attn_bytes, gemv_bytes) + 15) & ~(size_t)15) + 64;
    if (lds < 96 * 1024) lds = 96 * 1024;  // > half of the 160 KB LDS: at most one block per CU => all CUs get one
    if (lds > 160 * 1024) return IVLM_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        IVLM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(llama_generate_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    llama_generate_kernel<<<G, kThreads, lds, st>>>(a);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" size_t ivlm_llama_generate_workspace_bytes(int hidden, int inter) {
    return ivlm::llama_generate_workspace_bytes(hidden, inter);
}

extern "C" int ivlm_llama_generate(const int64_t* layer_ptrs, int L, int H, int D, int hidden, int inter, int vocab,
                                   float eps, float scale, const float* cos_tab, const float* sin_tab, void* kcache,
                                   void* vcache, int64_t cache_layer_stride, int max_len, const void* embed,
                                   const void* final_norm, const void* lm_head, void* hidden_out, int pos0, int n_max,
                                   int eos, const int32_t* forced, int32_t* new_ids, int32_t* argmax_ids, void* workspace,
                                   size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::llama_generate(layer_ptrs, L, H, D, hidden, inter, vocab, eps, scale, cos_tab, sin_tab,
                                static_cast<bf16_t*>(kcache), static_cast<bf16_t*>(vcache), cache_layer_stride, max_len,
                                static_cast<const bf16_t*>(embed), static_cast<const bf16_t*>(final_norm),
                                static_cast<const bf16_t*>(lm_head), static_cast<bf16_t*>(hidden_out), pos0, n_max, eos,
                                forced, new_ids, argmax_ids, workspace, workspace_bytes, ivlm_stream(stream));
}
