// Pieces shared by the GEMM tile kernels: activation, LDS-DMA helper, XCD-aware tile raster, fused epilogue.
// (A 4-phase / counted-vmcnt schedule of the 256x256 tile was built and measured in round 1: correct but 5-8 % slower
// than the one-barrier-per-K-tile loop of gemm.hip, so it is not shipped)
#pragma once
#include "kernels.h"

namespace ivlm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// one MX fp8 step: the two 16-byte fragments a lane holds for the two bf16 k-steps of a 128-byte K tile are, as bytes, 32 e4m3
// values of ONE 16x16x128 step (both operands use the same lane -> k-subset map, so the sum over k is the same sum)
__device__ __forceinline__ f32x4_t mfma_fp8_128(const bf16x8_t& a0, const bf16x8_t& a1, const bf16x8_t& b0, const bf16x8_t& b1,
                                                const f32x4_t& c) {
    struct P { bf16x8_t lo, hi; };
    const P pa{a0, a1}, pb{b0, b1};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_bit_cast(i32x8_t, pa), __builtin_bit_cast(i32x8_t, pb), c,
                                                            0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);  // e4m3 x e4m3, scales 2^0
}

// exact-GELU x * Phi(x) with Phi through erfc (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7, no cancellation for x < 0):
// ~14 VALU ops instead of ~45 for ocml erff - the GELU epilogue of the 16384x5120x1280 SAM MLP GEMM was 30 % of its time
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    // (every multiply-add is an explicit fma: the result must not depend on which contractions the compiler picks in the
    //  caller's context - the direct and the LDS-staged epilogues of the same GEMM are compared bit for bit)
    const float a = 0.5f * p * t, e = __expf(-z * z);
    const float q = a * e;  // 0.5 * erfc(z)
    return x * (x < 0.0f ? q : fmaf(-a, e, 1.0f));
}

// two elements at a time: the multiplies and fmas become v_pk_mul_f32 / v_pk_fma_f32 (two lanes' worth per issue slot), only rcp
// and exp stay scalar - the GELU of the 16384 x 5120 SAM mlp1 tile is VALU-bound (335 M erfs per 65536-row GEMM: ~225 of its
// 250 us of epilogue).  Same operations in the same order as gelu_erf_fast: identical values.
typedef float __attribute__((ext_vector_type(2))) f32x2_t;
__device__ __forceinline__ f32x2_t gelu_erf_fast2(f32x2_t x) {
    const f32x2_t ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2_t z = ax * 0.70710678118654752f;
    const f32x2_t d = __builtin_elementwise_fma(f32x2_t{0.3275911f, 0.3275911f}, z, f32x2_t{1.0f, 1.0f});
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2_t p = __builtin_elementwise_fma(f32x2_t{1.061405429f, 1.061405429f}, t, f32x2_t{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{0.254829592f, 0.254829592f});
    const f32x2_t a = (p * 0.5f) * t;
    const f32x2_t nzz = -z * z;
    const f32x2_t e = {__expf(nzz.x), __expf(nzz.y)};
    const f32x2_t q = a * e;
    const f32x2_t omq = __builtin_elementwise_fma(-a, e, f32x2_t{1.0f, 1.0f});
    return f32x2_t{x.x * (x.x < 0.0f ? q.x : omq.x), x.y * (x.y < 0.0f ? q.y : omq.y)};
}

__device__ __forceinline__ float gemm_act(float x, int act) {
    switch (act) {
        case ACT_GELU: return gelu_erf_fast(x);
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return x < 0.0f ? 0.0f : x;  // (torch.relu semantics: a NaN stays a NaN; fmaxf would turn it into 0)
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

// 16 zero bytes: DMA source for the K tail (K % 64 != 0) so that partial tiles contribute nothing
__device__ __attribute__((aligned(16))) const uint32_t kGemmZeroChunk[4] = {0u, 0u, 0u, 0u};

// ---- ablation switches of the tile GEMMs (per-shape ceiling table, profiles/r05_gemm_ceilings.txt): NEVER defined in the product
// build; tools/experiments/build_variant.sh compiles gemm.hip / gemm256.hip / gemm320.hip with -DIVLM_ABL_... into a side library.
//   IVLM_ABL_NODMA   the global -> LDS copies are not issued (the counted waits fall through)
//   IVLM_ABL_NOLDS   fragments are lane constants instead of LDS reads
//   IVLM_ABL_NOMFMA  the matrix instructions are replaced by a register use of their operands (keeps the reads alive)
//   IVLM_ABL_NOEPI   the epilogue stores one dummy value per lane instead of the tile
// (results of such a library are garbage by construction: timing only)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
#ifndef IVLM_ABL_NODMA
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}
// one fragment read (ds_read_b128); ablated: a lane-dependent constant the compiler cannot fold into the MFMA chain
__device__ __forceinline__ bf16x8_t gemm_frag_read(const unsigned char* p) {
#ifdef IVLM_ABL_NOLDS
    uint32_t v = (uint32_t)(uintptr_t)p * 0x9e3779b1u | 0x3c003c00u;
    asm volatile("" : "+v"(v));
    typedef __attribute__((ext_vector_type(4))) uint32_t u4_t;
    return __builtin_bit_cast(bf16x8_t, u4_t{v, v, v, v});
#else
    return *reinterpret_cast<const bf16x8_t*>(p);
#endif
}
// ablated MFMA: the operands are "used" (so that their reads stay), the accumulator is left alone
#ifdef IVLM_ABL_NOMFMA
#define IVLM_ABL_MFMA_USE(a, b)                                                          \
    do {                                                                                 \
        auto a_ = (a);                                                                   \
        auto b_ = (b);                                                                   \
        asm volatile("" ::"v"(a_), "v"(b_));                                             \
    } while (0)
#endif

// XCD-aware, grouped tile raster (see gemm.hip): block id -> tile origin
__device__ __forceinline__ void gemm_tile_origin(const GemmArgs& g, int BM, int BN, int& m0, int& n0) {
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    constexpr int kXcd = 8, kGroupM = 8;
    const int bid = blockIdx.x;
    const int xcd = bid % kXcd, loc = bid / kXcd;
    const int q = nwg / kXcd, r = nwg % kXcd;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per_group = kGroupM * tiles_n;
    const int first_m = (lin / per_group) * kGroupM;
    const int gsz = tiles_m - first_m < kGroupM ? tiles_m - first_m : kGroupM;
    const int in_group = lin % per_group;
    m0 = (first_m + in_group % gsz) * BM;
    n0 = (in_group / gsz) * BN;
}

// one residual element (bf16 or fp32 stream)
__device__ __forceinline__ float gemm_residual_at(const GemmArgs& g, const bf16_t* R, int64_t idx) {
    return g.res_f32 ? reinterpret_cast<const float*>(R)[idx] : bf16_to_f32(R[idx]);
}

// split-K: columns n .. n + 3 of row m - the slices summed in slice order, then bias / activation / residual / output conversion of
// `g` (the GEMM's REAL arguments).  One function for the reduction launch and for the fused fixup: the same values either way.
template <bool OUT_F32>
__device__ __forceinline__ void splitk_finish4(const GemmArgs& g, const float* __restrict__ part, int splits, int m, int n) {
    const int64_t slice = (int64_t)g.M * g.N;
    const float* p = part + (int64_t)m * g.N + n;
    float4 acc = *reinterpret_cast<const float4*>(p);
    for (int sidx = 1; sidx < splits; ++sidx) {
        const float4 t = *reinterpret_cast<const float4*>(p + sidx * slice);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (g.bias) v[j] += bf16_to_f32(g.bias[n + j]);
        v[j] = gemm_act(v[j], g.act);
        if (g.residual) {
            const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
            v[j] += gemm_residual_at(g, g.residual, rrow * g.ldr + n + j);
        }
    }
    const int64_t o = (int64_t)m * g.ldc + n;
    if (OUT_F32 && g.out_split) {
        uint32_t h0, l0, h1, l1;
        split_16x2(v[0], v[1], h0, l0, g.out_f16);
        split_16x2(v[2], v[3], h1, l1, g.out_f16);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(g.C) + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(g.C) + o + g.c_lo) = make_uint2(l0, l1);
    } else if (OUT_F32) *reinterpret_cast<float4*>(static_cast<float*>(g.C) + o) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<uint2*>(static_cast<bf16_t*>(g.C) + o) = make_uint2(pack_16x2(v[0], v[1], g.out_f16), pack_16x2(v[2], v[3], g.out_f16));
}

// Fused split-K fixup (SplitKFused): called by EVERY thread of a block after its partial tile has been stored.  The block counts its
// arrival; the last one of the tile reduces it.  Release / acquire at agent scope (the slices of a tile run on different XCDs, whose
// L2s are not coherent with each other): fence -> barrier -> one atomic per block; the last block fences again before it reads.
template <int BM, int BN, int THREADS>
__device__ __forceinline__ void splitk_fixup(const GemmArgs& g, int m0, int n0) {
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(g.sk.count + blockIdx.x, 1);
        const int last = old == g.batch - 1;
        if (last) g.sk.count[blockIdx.x] = 0;  // every arrival is in: leave the counter at zero for the next launch
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    GemmArgs f = g;  // the real epilogue (uniform values)
    f.C = g.sk.C; f.bias = g.sk.bias; f.residual = g.sk.residual;
    f.ldc = g.sk.ldc; f.ldr = g.sk.ldr; f.c_lo = g.sk.c_lo;
    f.res_mod = g.sk.res_mod; f.act = g.sk.act; f.out_f32 = g.sk.out_f32; f.out_f16 = g.sk.out_f16; f.out_split = g.sk.out_split;
    f.res_f32 = g.sk.res_f32;
    const float* part = static_cast<const float*>(g.C);
    constexpr int kQ = BN / 4;
    for (int i = threadIdx.x; i < BM * kQ; i += THREADS) {
        const int m = m0 + i / kQ, n = n0 + (i % kQ) * 4;
        if (m >= g.M || n >= g.N) continue;  // (N % 4 == 0)
        if (f.out_f32) splitk_finish4<true>(f, part, g.batch, m, n);
        else splitk_finish4<false>(f, part, g.batch, m, n);
    }
}

// bias + activation of one accumulator fragment (the lane owns columns n .. n+3; N % 4 == 0, n < N): the value part of the
// epilogue, for kernels that stage the tile through LDS and store whole lines (gemm256.hip)
template <int ACT>
__device__ __forceinline__ void gemm_value4(const GemmArgs& g, int n, const f32x4_t& acc, float (&v)[4]) {
    v[0] = acc[0]; v[1] = acc[1]; v[2] = acc[2]; v[3] = acc[3];
    if (g.fp8) {
        const float alpha = (*g.scale_a) * (*g.scale_w);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= alpha;
    }
    if (g.bias) {
        const uint2 b2 = *reinterpret_cast<const uint2*>(g.bias + n);
        v[0] += bf16_to_f32((bf16_t)(b2.x & 0xffff));
        v[1] += bf16_to_f32((bf16_t)(b2.x >> 16));
        v[2] += bf16_to_f32((bf16_t)(b2.y & 0xffff));
        v[3] += bf16_to_f32((bf16_t)(b2.y >> 16));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = gemm_act(v[j], ACT);
}

// ---- whole-line epilogue ------------------------------------------------------------------------------------------
// A lane holds C[m][n..n+3] (m = lane & 15, n = (lane >> 4) * 4) of each 16 x 16 fragment: stored directly, one wave instruction
// touches 16 rows x 32 (bf16) / 64 (fp32) bytes - 16 PARTIAL cache lines - and the 32 fragments of a 128 x 64 wave tile cost
// ~15 us per 256 x 256 block tile, a quarter to a third of the SAM GEMMs (ablation of gemm256_kernel at M = 65536: qkv 806 ->
// 589 us, mlp1 1204 -> 680 us without the epilogue; tools/experiments/README.md).  After the K loop the tile buffers are free:
// each wave transposes its sub-tile through its own slice of LDS (chunk swizzle: conflict-free both ways) and then writes - and
// reads the residual as - WHOLE lines: 8 rows x 128 B or 4 rows x 256 B per wave instruction.  Same values, same order of
// operations per element as gemm_epilogue4 (bit-identical results).
template <bool OUT_F32>
__host__ __device__ __forceinline__ bool gemm_whole_lines_ok(const GemmArgs& g, int act) {
    return act != ACT_SWIGLU && !g.out_fp8 && g.batch == 1 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
           (OUT_F32 ? ((g.N & 3) == 0 && (g.ldc & 3) == 0 && (!g.residual || (g.ldr & 3) == 0) && (!g.out_split || (g.c_lo & 3) == 0))
                    : ((g.N & 7) == 0 && !g.residual && (g.c_panel ? (g.c_panel & 7) == 0 : (g.ldc & 7) == 0)));
}

// wbuf: this wave's LDS slice (>= PASS_MI * 16 rows x NI * 16 elements); (mw, nw): origin of the wave's sub-tile
// (NI_ALL / NI0: the accumulator array may be wider than the NI fragments stored by this call - they start at fragment NI0)
template <int ACT, bool OUT_F32, int MI, int NI, int PASS_MI, int NI_ALL = NI, int NI0 = 0>
__device__ __forceinline__ void gemm_store_lines(const GemmArgs& g, unsigned char* wbuf, int mw, int nw, int lane,
                                                 const f32x4_t (&acc)[NI_ALL][MI]) {
    constexpr int ES = OUT_F32 ? 4 : 2;
    constexpr int RB = NI * 16 * ES;  // bytes per row of the sub-tile
    constexpr int CH = RB / 16;       // 16-byte chunks per row
    static_assert(CH == 8 || CH == 16, "rows of 128 or 256 bytes");
    static_assert(MI % PASS_MI == 0, "whole passes");
    constexpr int RPI = 64 / CH;      // rows per wave instruction
    constexpr int ROWS = PASS_MI * 16;
    const int cc = lane % CH, n = nw + cc * (16 / ES);
#pragma unroll
    for (int p = 0; p < MI / PASS_MI; ++p) {
#pragma unroll
        for (int mi = 0; mi < PASS_MI; ++mi) {
            const int r = mi * 16 + (lane & 15);
            const int sw = CH == 8 ? ((r >> 1) & 7) : (r & 15);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int nl = ni * 16 + (lane >> 4) * 4;  // local column
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (nw + nl < g.N) gemm_value4<ACT>(g, nw + nl, acc[NI0 + ni][p * PASS_MI + mi], v);
                unsigned char* dst = wbuf + r * RB + ((((nl * ES) >> 4) ^ sw) << 4) + ((nl * ES) & 15);
                if (OUT_F32) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_16x2(v[0], v[1], g.out_f16), pack_16x2(v[2], v[3], g.out_f16));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < ROWS / RPI; ++it) {
            const int rr = it * RPI + lane / CH;
            const int sw = CH == 8 ? ((rr >> 1) & 7) : (rr & 15);
            const unsigned char* src = wbuf + rr * RB + ((cc ^ sw) << 4);
            int m = mw + p * ROWS + rr;
            if (m >= g.M || n >= g.N) continue;
            if (g.out_rows) {  // scatter epilogue: destination (and residual) row from the map; negative = dropped
                m = g.out_rows[m];
                if (m < 0) continue;
            }
            if (OUT_F32) {
                float4 q = *reinterpret_cast<const float4*>(src);
                if (g.residual) {
                    const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
                    if (g.res_f32) {
                        const float4 r4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.residual) + rrow * g.ldr + n);
                        q.x += r4.x; q.y += r4.y; q.z += r4.z; q.w += r4.w;
                    } else {
                        const uint2 r2 = *reinterpret_cast<const uint2*>(g.residual + rrow * g.ldr + n);
                        q.x += bf16_to_f32((bf16_t)(r2.x & 0xffff));
                        q.y += bf16_to_f32((bf16_t)(r2.x >> 16));
                        q.z += bf16_to_f32((bf16_t)(r2.y & 0xffff));
                        q.w += bf16_to_f32((bf16_t)(r2.y >> 16));
                    }
                }
                if (g.out_split) {  // [hi | lo] bf16 halves of the fp32 value (16 lanes x 8 B = one 128-byte line each)
                    uint32_t h0, l0, h1, l1;
                    split_16x2(q.x, q.y, h0, l0, g.out_f16);
                    split_16x2(q.z, q.w, h1, l1, g.out_f16);
                    bf16_t* cb = static_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc + n;
                    *reinterpret_cast<uint2*>(cb) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(cb + g.c_lo) = make_uint2(l0, l1);
                } else {
                    *reinterpret_cast<float4*>(static_cast<float*>(g.C) + (int64_t)m * g.ldc + n) = q;
                }
            } else {
                const int64_t o = g.c_panel ? (int64_t)(n >> 6) * g.c_panel + (int64_t)m * 64 + (n & 63) : (int64_t)m * g.ldc + n;
                *reinterpret_cast<uint4*>(static_cast<bf16_t*>(g.C) + o) = *reinterpret_cast<const uint4*>(src);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the same through a tile SHARED by several waves (gemm320.hip: the fifth fragment column of the four waves of a wave row forms
// one 64-column strip).  Stage: a wave writes one 16-column fragment column (MI fragments = MI * 16 rows) at local column `cl` of a
// [rows x 64] tile; after a block barrier, rows: a wave stores `nrows` rows of the tile as whole lines.  Same swizzle, same values
// and order of operations per element as gemm_store_lines / gemm_epilogue4.
template <int ACT, bool OUT_F32, int MI, int NI_ALL>
__device__ __forceinline__ void gemm_stage_strip(const GemmArgs& g, unsigned char* tile, int cl, int n_global, int lane,
                                                 const f32x4_t (&acc)[NI_ALL][MI], int ni_src) {
    constexpr int ES = OUT_F32 ? 4 : 2;
    constexpr int RB = 64 * ES, CH = RB / 16;
    const int nl = cl + (lane >> 4) * 4;  // local column of the lane's four values
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = mi * 16 + (lane & 15);
        const int sw = CH == 8 ? ((r >> 1) & 7) : (r & 15);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (n_global + (lane >> 4) * 4 < g.N) gemm_value4<ACT>(g, n_global + (lane >> 4) * 4, acc[ni_src][mi], v);
        unsigned char* dst = tile + r * RB + ((((nl * ES) >> 4) ^ sw) << 4) + ((nl * ES) & 15);
        if (OUT_F32) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_16x2(v[0], v[1], g.out_f16), pack_16x2(v[2], v[3], g.out_f16));
    }
}

template <bool OUT_F32>
__device__ __forceinline__ void gemm_store_strip_rows(const GemmArgs& g, const unsigned char* tile, int row0, int nrows, int m_base, int n_base,
                                                      int lane) {
    constexpr int ES = OUT_F32 ? 4 : 2;
    constexpr int RB = 64 * ES, CH = RB / 16, RPI = 64 / CH;
    const int cc = lane % CH, n = n_base + cc * (16 / ES);
    for (int it = 0; it < nrows / RPI; ++it) {
        const int rr = row0 + it * RPI + lane / CH;
        const int sw = CH == 8 ? ((rr >> 1) & 7) : (rr & 15);
        const unsigned char* src = tile + rr * RB + ((cc ^ sw) << 4);
        int m = m_base + rr;
        if (m >= g.M || n >= g.N) continue;
        if (g.out_rows) {
            m = g.out_rows[m];
            if (m < 0) continue;
        }
        if (OUT_F32) {
            float4 q = *reinterpret_cast<const float4*>(src);
            if (g.residual) {
                const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
                if (g.res_f32) {
                    const float4 r4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.residual) + rrow * g.ldr + n);
                    q.x += r4.x; q.y += r4.y; q.z += r4.z; q.w += r4.w;
                } else {
                    const uint2 r2 = *reinterpret_cast<const uint2*>(g.residual + rrow * g.ldr + n);
                    q.x += bf16_to_f32((bf16_t)(r2.x & 0xffff));
                    q.y += bf16_to_f32((bf16_t)(r2.x >> 16));
                    q.z += bf16_to_f32((bf16_t)(r2.y & 0xffff));
                    q.w += bf16_to_f32((bf16_t)(r2.y >> 16));
                }
            }
            if (g.out_split) {
                uint32_t h0, l0, h1, l1;
                split_16x2(q.x, q.y, h0, l0, g.out_f16);
                split_16x2(q.z, q.w, h1, l1, g.out_f16);
                bf16_t* cb = static_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc + n;
                *reinterpret_cast<uint2*>(cb) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(cb + g.c_lo) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<float4*>(static_cast<float*>(g.C) + (int64_t)m * g.ldc + n) = q;
            }
        } else {
            *reinterpret_cast<uint4*>(static_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc + n) = *reinterpret_cast<const uint4*>(src);
        }
    }
}

// Epilogue for one accumulator fragment: the lane owns C[m][n .. n+3] (operands were swapped so that the four
// registers are consecutive N).  bias -> activation (or SwiGLU) -> residual -> store.
template <int ACT, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue4(const GemmArgs& g, int bz, int m, int n, const f32x4_t& acc) {
    if (m >= g.M || n >= g.N) return;
    if (g.out_rows) {  // scatter epilogue: destination (and residual) row from the map; negative = dropped
        m = g.out_rows[m];
        if (m < 0) return;
    }
    const bf16_t* __restrict__ bias = g.bias;
    const bf16_t* __restrict__ R = g.residual ? g.residual + (int64_t)bz * g.strideR * (g.res_f32 ? 2 : 1) : nullptr;
    const int64_t rrow = g.res_mod > 0 ? (m % g.res_mod) : m;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (g.fp8) {  // per-tensor dequantisation scales (device scalars)
        const float alpha = (*g.scale_a) * (*g.scale_w);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= alpha;
    }
    if ((g.N & 3) == 0) {
        if (bias) {
            const uint2 b2 = *reinterpret_cast<const uint2*>(bias + n);
            v[0] += bf16_to_f32((bf16_t)(b2.x & 0xffff));
            v[1] += bf16_to_f32((bf16_t)(b2.x >> 16));
            v[2] += bf16_to_f32((bf16_t)(b2.y & 0xffff));
            v[3] += bf16_to_f32((bf16_t)(b2.y >> 16));
        }
        if (ACT == ACT_SWIGLU) {
            // rows interleaved (gate_j, up_j): out[j] = silu(gate_j) * up_j, two outputs per lane
            const float o0 = (v[0] / (1.0f + __expf(-v[0]))) * v[1];
            const float o1 = (v[2] / (1.0f + __expf(-v[2]))) * v[3];
            const int64_t o = (int64_t)m * g.ldc + (n >> 1);
            if (g.out_fp8) {  // e4m3 pair with the consumer's calibrated per-tensor scale (LLaMA gate|up -> down)
                const float inv = 1.0f / (*g.scale_out);
                const uint32_t wq = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o0 * inv, -448.0f), 448.0f),
                                                                    fminf(fmaxf(o1 * inv, -448.0f), 448.0f), 0u, false);
                *reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(g.C) + (int64_t)bz * g.strideC + o) = (uint16_t)(wq & 0xffffu);
                return;
            }
            if (OUT_F32 && g.out_split) {
                bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
                uint32_t hi, lo;
                split_16x2(o0, o1, hi, lo, g.out_f16);
                *reinterpret_cast<uint32_t*>(C + o) = hi;
                *reinterpret_cast<uint32_t*>(C + o + g.c_lo) = lo;
            } else if (OUT_F32) {
                float* C = static_cast<float*>(g.C) + (int64_t)bz * g.strideC;
                *reinterpret_cast<float2*>(C + o) = make_float2(o0, o1);
            } else {
                bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
                *reinterpret_cast<uint32_t*>(C + o) = pack_16x2(o0, o1, g.out_f16);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gemm_act(v[j], ACT);
        if (R) {
            if (g.res_f32) {  // fp32 residual stream
                const float4 r4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(R) + rrow * g.ldr + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            } else {
                const uint2 r2 = *reinterpret_cast<const uint2*>(R + rrow * g.ldr + n);
                v[0] += bf16_to_f32((bf16_t)(r2.x & 0xffff));
                v[1] += bf16_to_f32((bf16_t)(r2.x >> 16));
                v[2] += bf16_to_f32((bf16_t)(r2.y & 0xffff));
                v[3] += bf16_to_f32((bf16_t)(r2.y >> 16));
            }
        }
        const int64_t o = g.c_panel ? (int64_t)(n >> 6) * g.c_panel + (int64_t)m * 64 + (n & 63) : (int64_t)m * g.ldc + n;
        if (g.out_fp8) {  // e4m3 output with the consumer's calibrated per-tensor scale (mlp1 -> mlp2)
            // (values beyond the calibrated range saturate at +-448: e4m3 has no infinity, an unclamped overflow converts to NaN)
            const float inv = 1.0f / (*g.scale_out);
            float c4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c4[j] = fminf(fmaxf(v[j] * inv, -448.0f), 448.0f);
            uint32_t w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[0], c4[1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(c4[2], c4[3], w, true);
            *reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(g.C) + (int64_t)bz * g.strideC + o) = w;
        } else if (OUT_F32 && g.out_split) {
            bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
            uint32_t h0, l0, h1, l1;
            split_16x2(v[0], v[1], h0, l0, g.out_f16);
            split_16x2(v[2], v[3], h1, l1, g.out_f16);
            *reinterpret_cast<uint2*>(C + o) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(C + o + g.c_lo) = make_uint2(l0, l1);
        } else if (OUT_F32) {
            float* C = static_cast<float*>(g.C) + (int64_t)bz * g.strideC;
            *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
            *reinterpret_cast<uint2*>(C + o) = make_uint2(pack_16x2(v[0], v[1], g.out_f16), pack_16x2(v[2], v[3], g.out_f16));
        }
    } else {  // ragged N: scalar tail (never on the hot shapes)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= g.N) break;
            float x = v[j] + (bias ? bf16_to_f32(bias[n + j]) : 0.0f);
            x = gemm_act(x, ACT);
            if (R) x += gemm_residual_at(g, R, rrow * g.ldr + n + j);
            const int64_t o = (int64_t)m * g.ldc + n + j;
            if (OUT_F32 && g.out_split) {
                bf16_t* C = static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC;
                uint32_t h2, l2;
                split_16x2(x, 0.0f, h2, l2, g.out_f16);
                C[o] = (bf16_t)(h2 & 0xffffu);
                C[o + g.c_lo] = (bf16_t)(l2 & 0xffffu);
            } else if (OUT_F32)
                (static_cast<float*>(g.C) + (int64_t)bz * g.strideC)[o] = x;
            else
                (static_cast<bf16_t*>(g.C) + (int64_t)bz * g.strideC)[o] = g.out_f16 ? f32_to_h16<true>(x) : f32_to_bf16(x);
        }
    }
}

}  // namespace ivlm
