// Flat "slab" weight streaming shared by the decode GEMV kernels (gemv_slab.hip) and the persistent generation kernel
// (generate.hip): a block streams a contiguous range of weight rows as ONE coalesced byte range, lane t owning 16-byte
// chunks t, t + 512, ..., kDepth loads in flight per lane; the activation vector is staged once in LDS; every 64-lane
// step is wave-reduced on the DPP network into an LDS slot and the rows are summed in a fixed order afterwards.
#pragma once
#include "kernels.h"

namespace ivlm {
namespace slabk {

constexpr int kThreads = 512;   // 8 waves = 2 per SIMD: 256 VGPRs each, room for 16 x 16-byte loads in flight per lane
constexpr int kWaves = kThreads / 64;
constexpr int kDepth = 16;       // 16-byte weight loads in flight per lane (128 KB per CU)
constexpr int kMaxSteps = 352;   // lane steps per phase (slab chunks / 512, rounded up to groups of kDepth)
constexpr int kMaxHeadDim = 128;
constexpr int kMaxPos = 4096;    // attention scores in LDS
constexpr int kMaxRows = 256;    // rows of a slab
constexpr long long kTimeoutTicks = 200000000LL;  // wall_clock64 runs at 100 MHz

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
// Weight pointers come out of a device-side table, so the compiler only knows them as flat pointers; flat loads also
// count on lgkmcnt (every LDS sync would wait for the weight stream).  Re-type them as global (address space 1).
typedef const __attribute__((address_space(1))) u32x4_t* gvec_ptr;
typedef const __attribute__((address_space(1))) uint32_t* gword_ptr;
__device__ __forceinline__ gvec_ptr as_gvec(const void* p) { return (gvec_ptr)(uintptr_t)p; }
__device__ __forceinline__ gword_ptr as_gword(const void* p) { return (gword_ptr)(uintptr_t)p; }

// ---- agent-scope (cross-XCD coherent) accesses for the activations that cross a barrier --------------------
__device__ __forceinline__ uint32_t ld_agent(const void* p) {
    return __hip_atomic_load(static_cast<uint32_t*>(const_cast<void*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent_f(const float* p) {
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bf16_t ld_agent16(const bf16_t* p) {
    return __hip_atomic_load(const_cast<bf16_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent16(bf16_t* p, bf16_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_f(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_i(int32_t* p, int32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS-only workgroup sync: __syncthreads() carries a workgroup release fence = s_waitcnt vmcnt(0), which would make every
// wave wait for its prefetched weight loads at each sync.  Cross-thread traffic inside a block goes through LDS only.
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// all of this wave's global stores performed (agent-scope stores: visible device-wide) before anything that follows
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// 8 bf16 x 8 bf16 -> fp32 on v_dot2c_f32_bf16 (gfx950): no unpacking, 4 VALU ops per 16-byte chunk
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float dot8(const u32x4_t& w, const u32x4_t& x) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // scalar copies first: __builtin_bit_cast applied directly to an ext-vector element expression reads lane 0
        const uint32_t wj = w[j], xj = x[j];
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wj), __builtin_bit_cast(bf16x2_t, xj), acc, false);
    }
    return acc;
}

struct Lds {
    u32x4_t* xs;     // staged activation vector, 16-byte chunks
    float* part;     // [kMaxSteps][kWaves][2] wave partials
    float* rowsum;   // [kMaxRows]
    float* red;      // [2 * kWaves]
    int* flag;       // [4]: dead, token, ...
    unsigned char* attn;  // attention scratch (overlays xs/part: they are idle during the attention phase)
};

// rows [r0, r1) of an N-row matrix owned by block b (units of `unit` rows: 2 keeps SwiGLU gate/up pairs together)
// `skip` leading blocks get nothing (the o_proj slabs leave out the blocks that run attention, see the kernel).
__device__ __forceinline__ void slab(int N, int unit, int b, int G, int& r0, int& r1, int skip = 0) {
    const unsigned nu = (unsigned)(N / unit), g = (unsigned)(G - skip);  // nu * G < 2^32 (vocab x 1024)
    if (b < skip) {
        r0 = r1 = 0;
        return;
    }
    const unsigned bb = (unsigned)(b - skip);
    r0 = (int)(nu * bb / g) * unit;
    r1 = (int)(nu * (bb + 1) / g) * unit;
    if (bb == g - 1) r1 = N;
}

// 16 zero bytes: the load target of every lane step that lies beyond the end of a slab.  Keeps the streaming code
// free of branches (so the compiler can count loads in flight) and makes such steps contribute exactly 0.
__device__ __attribute__((aligned(16))) const uint32_t kZeroChunk[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ u32x4_t load_chunk(gvec_ptr base, int f, int total) {
    gvec_ptr p = f < total ? base + f : as_gvec(kZeroChunk);
    return __builtin_nontemporal_load(p);
}

// issue the first kDepth loads per lane of a slab (non-temporal: weights are read exactly once per token)
__device__ __forceinline__ void prefetch(u32x4_t (&buf)[kDepth], const bf16_t* W, int K, int r0, int r1) {
    const int total = (r1 - r0) * (K >> 3);  // 16-byte chunks of the slab (< 2^31: slabs are a few MB)
    gvec_ptr base = as_gvec(W + (int64_t)r0 * K);
#pragma unroll
    for (int d = 0; d < kDepth; ++d) buf[d] = load_chunk(base, (int)threadIdx.x + kThreads * d, total);
}

// wave64 sum on the DPP network (no LDS crossbar traffic): quad swaps, row rotations, then the gfx9 row broadcasts;
// the total lands in lane 63 and is returned wave-uniform.  Fixed order => deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_total(float v) {
    v = dpp_add<0xb1, 0xf>(v);   // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);   // quad_perm:[2,3,0,1]
    v = dpp_add<0x124, 0xf>(v);  // row_ror:4
    v = dpp_add<0x128, 0xf>(v);  // row_ror:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Stream the slab against the staged vector; leaves rowsum[i] = dot(W[r0+i,:], xs) for i < r1-r0.
// `buf` must hold the prefetched first kDepth steps.  STRADDLE: rows are not a whole number of waves long, a 64-lane
// step may end one row and begin the next (never more: K >= 512).
struct StreamState {
    int f, row, c;
    float* slot;
};

template <bool STRADDLE, bool ISSUE>
__device__ __forceinline__ void stream_step(u32x4_t& w, gvec_ptr base, int total, int nchunk, int dr, int dc,
                                            StreamState& st, const Lds& L) {
    const float p = dot8(w, L.xs[st.c]);  // steps beyond the slab read the zero chunk: p == 0
    if (ISSUE) w = load_chunk(base, st.f + kThreads * kDepth, total);
    float s0, s1 = 0.0f;
    if (STRADDLE) {
        const int row_first = __builtin_amdgcn_readfirstlane(st.row);
        s0 = wave_total(st.row == row_first ? p : 0.0f);
        s1 = wave_total(st.row == row_first ? 0.0f : p);
    } else {
        s0 = wave_total(p);
    }
    if ((threadIdx.x & 63) == 0) {
        st.slot[0] = s0;
        st.slot[1] = s1;
    }
    st.slot += kWaves * 2;
    st.f += kThreads;
    st.row += dr;
    st.c += dc;
    if (st.c >= nchunk) {
        st.c -= nchunk;
        ++st.row;
    }
}

// Groups of kDepth steps, all straight-line (steady state waits with vmcnt(kDepth-1)); the last group does not refill.
// The step count is rounded up to whole groups: the padding steps stream the zero chunk.
template <bool STRADDLE>
__device__ __forceinline__ void stream_loop(u32x4_t (&buf)[kDepth], gvec_ptr base, int total, int nsteps, int nchunk,
                                            const Lds& L) {
    const int dr = kThreads / nchunk, dc = kThreads % nchunk;
    StreamState st;
    st.f = threadIdx.x;
    st.row = threadIdx.x / nchunk;
    st.c = threadIdx.x % nchunk;
    st.slot = L.part + ((threadIdx.x >> 6) << 1);
    const int ngroups = (nsteps + kDepth - 1) / kDepth;
    for (int g = 0; g + 1 < ngroups; ++g) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) stream_step<STRADDLE, true>(buf[d], base, total, nchunk, dr, dc, st, L);
    }
    if (ngroups > 0) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) stream_step<STRADDLE, false>(buf[d], base, total, nchunk, dr, dc, st, L);
    }
}

__device__ __forceinline__ void stream_slab(u32x4_t (&buf)[kDepth], const bf16_t* W, int K, int r0, int r1, Lds& L) {
    const int nchunk = K >> 3;
    const int nrows = r1 - r0;
    const int total = nrows * nchunk;
    const int nsteps = (total + kThreads - 1) / kThreads;
    gvec_ptr base = as_gvec(W + (int64_t)r0 * K);
    if (nchunk & 63) stream_loop<true>(buf, base, total, nsteps, nchunk, L);
    else stream_loop<false>(buf, base, total, nsteps, nchunk, L);
    block_sync_lds();
    if ((int)threadIdx.x < nrows) {
        const int t = threadIdx.x;
        const int ws_lo = (t * nchunk) >> 6;
        const int ws_hi = ((t + 1) * nchunk - 1) >> 6;
        float s = 0.0f;
        for (int ws = ws_lo; ws <= ws_hi; ++ws) {
            const int first_row = (ws << 6) / nchunk;
            s += L.part[(ws << 1) + (first_row == t ? 0 : 1)];
        }
        L.rowsum[t] = s;
    }
    block_sync_lds();
}

// stage a K-vector in LDS.  rms: xs = bf16(x * gamma) and returns rsqrt(mean(x^2)+eps) (the HF rounding variant used
// by the fused GEMV path: the scale is applied to the fp32 dot).  src is read with agent-scope loads when `coherent`.
__device__ __forceinline__ float stage_vec(const bf16_t* src, const bf16_t* gamma, int K, float eps, bool rms, bool coherent,
                                           Lds& L) {
    uint32_t* xs32 = reinterpret_cast<uint32_t*>(L.xs);
    const int nd = K >> 1;
    float ssq = 0.0f;
    for (int i = threadIdx.x; i < nd; i += kThreads) {
        uint32_t v = coherent ? ld_agent(reinterpret_cast<const uint32_t*>(src) + i)
                              : as_gword(src)[i];
        if (rms) {
            const float lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
            ssq += lo * lo + hi * hi;
            const uint32_t gv = as_gword(gamma)[i];
            v = pack_bf16x2(lo * __uint_as_float(gv << 16), hi * __uint_as_float(gv & 0xffff0000u));
        }
        xs32[i] = v;
    }
    float rstd = 1.0f;
    if (rms) {
        ssq = wave_sum(ssq);
        if ((threadIdx.x & 63) == 0) L.red[threadIdx.x >> 6] = ssq;
        block_sync_lds();
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += L.red[w];
        rstd = rsqrtf(tot / (float)K + eps);
    }
    block_sync_lds();
    return rstd;
}

}  // namespace slabk
}  // namespace ivlm
