// Fused (flash-style) multi-head attention forward for gfx950, bf16 in/out, fp32 softmax.
//
// One kernel family serves every attention on the hot path (reference call sites):
//   SAM ViT-H windowed (196 tok) / global (4096 tok), head dim 80, decomposed rel-pos bias
//       model/segment_anything/modeling/image_encoder.py:235-260, 354-392
//   CLIP ViT-L/14 (257 tok, head dim 64) and LLaMA (causal, head dim 128, fp32 softmax)   HF transformers
//   SAM two-way decoder attention (head dim 32 / 16)     modeling/transformer.py:220-242
//
// Design (wave64, MFMA 16x16x32 bf16):
//   * block = 4 waves x 32 queries; K/V tiles of 64 keys are staged once per block through
//     registers into LDS (issue-early / write-late, so HBM latency hides under the MFMAs).
//   * both products are computed TRANSPOSED:  S^T = K.Q^T  and  O^T = V^T.P^T.  A lane then owns ONE
//     query (column l&15) in every accumulator: the running max / sum / rescale are lane-local
//     (two xor-shuffles across the four 16-lane groups), and P never leaves registers: the four
//     scores a lane holds per 16-key tile are exactly the K-slots it must feed to the next MFMA
//     once the key order inside each 32-key step is permuted consistently for P and V.
//   * LDS images are built for the hardware's lane groups (a ds_read_b128 is served 16 lanes at a time: {0-3,12-15,
//     20-27}, {4-11,16-19,28-31}, ...; bank = dword address mod 64):
//       K  [k-step][key][32 elements]: 64-byte rows, 16-byte chunk g of key r stored at chunk g ^ ((r >> 3 & 1) << 1) -
//          every lane group then covers 16 distinct 16-byte bank slots (a padded row-major image is 2-way conflicted);
//       V  [16-wide d tile][key][16 elements], row-major as it arrives from HBM (16-byte stores, no scatter): the
//          V^T fragments of the second MFMA come from ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group, lane i
//          receives column i), two reads per fragment in the permuted key order.
//     Both are double-buffered: one barrier per key tile.  Head dim 80 pads to 96 for QK^T (zero chunks).
//   * the S x S score matrix is never materialised (the reference materialises [B*16,4096,4096]).
#include <algorithm>

#include "kernels.h"

namespace ivlm {


namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// native vector (NOT HIP's uint4 struct): arrays of it are promoted to registers across the K/V loop
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// Operand kind of the 16-bit tiles (F16 template flag): bf16 (default) or IEEE fp16 - same tiles, LDS images and fragment layouts,
// the f16 matrix instruction; an activation rounded to fp16 (11 significant bits) carries an eighth of the bf16 rounding error.
template <bool F16>
__device__ __forceinline__ f32x4_t mma16(const bf16x8_t a, const bf16x8_t b, const f32x4_t c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// two fp32 -> one packed 16-bit pair (RNE; fp16 overflows to inf), and back
template <bool F16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (F16) return pack_f16x2(lo, hi);
    else return pack_bf16x2(lo, hi);
}
template <bool F16>
__device__ __forceinline__ float lo16(uint32_t u) {
    if constexpr (F16) return (float)__builtin_bit_cast(ivlm_f16x2_t, u)[0];
    else return __uint_as_float(u << 16);
}
template <bool F16>
__device__ __forceinline__ float hi16(uint32_t u) {
    if constexpr (F16) return (float)__builtin_bit_cast(ivlm_f16x2_t, u)[1];
    else return __uint_as_float(u & 0xffff0000u);
}

constexpr int kQPerWave = 32, kWaves = 4, kQPerBlock = kQPerWave * kWaves, kKV = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -1.0e30f;

// max / sum over the four 16-lane groups (lanes l, l^16, l^32, l^48) with gfx950's row-swap permutes: no LDS round trip
// (ds_bpermute costs ~100 cycles of latency per step, eight dependent steps per key tile)
__device__ __forceinline__ float groups_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float groups_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Block barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait at EVERY key tile for the
// global loads of the next tiles that were issued precisely so that they stay in flight across the barrier.
__device__ __forceinline__ void sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// REL: 0 no bias; 1 rel-pos folded into the QK^T MFMA (KH + KW <= 32: one extra k-step whose Q' operand is
// [rel_h | rel_w] and whose K' operand is the one-hot (kh, KH + kw) code of the key: exact, no VALU);
// 2 rel_kw == 64 == key-tile: rel_w is tile-invariant (held as packed bf16 in registers), rel_h is one
// value per (query, tile); 3 any other grid: per-score table lookups (slow, non-SAM-H sizes only).
// PP (ping-pong): 8 waves = two groups of four that share the K/V tiles of a 256-query block and run half a tile apart -
// while one group issues its MFMAs (P.V of the previous tile, then Q.K^T of the next), the other does its softmax on the
// VALU, and they swap at every barrier.  Waves w and w + 4 share a SIMD, so the matrix unit and the VALU are both busy
// instead of both groups queueing for the same unit (without PP the two co-resident blocks drift into the same phase:
// measured time = MFMA time + VALU time).
// SPLIT ("parity" precision): q / k / v arrive as hi + lo bf16 planes (x = hi + lo to 2^-17: the [hi | lo] rows the GEMM's split
// epilogue writes; the lo tensors have the strides of the hi ones) and both products run as three MFMAs per fragment,
//     S^T = Kh.Qh^T + Kh.Ql^T + Kl.Qh^T,      O^T = Vh^T.Ph^T + Vh^T.Pl^T + Vl^T.Ph^T      (the lo.lo terms are below 2^-17),
// i.e. fp32-operand attention on the bf16 matrix cores at 3x the MFMA work; the output is written as hi + lo planes again.
// The LDS tiles double (dynamic LDS, one block per CU), one register set stages the next tile.
// QLO (with F16; the "exact q" path of the fp16 mode): q arrives as hi + lo IEEE halves (a.q_lo: the [hi | lo] rows of the q projection
// on split rows) - the rel-pos table product and Q.K^T take both halves (two MFMAs per fragment: q enters the scores exactly), and the
// softmax weights are split the same way for P.V (two MFMAs per fragment).  k, v and the output stay single fp16.
// QLV 1: the lo half of q enters the in-kernel rel-pos table products only (REL 4 / REL 5), see win_attn_kernel.
// REL 5 = REL 2 (SAM's 64 x 64 global grid, key tile = one grid row) with the decomposed rel-pos terms computed HERE from the
// [rel_pos_h (127 rows) ; rel_pos_w (127 rows)] table instead of read from [B*H, S, 64] arrays: no batched G = q . T^T GEMM
// (268 MB of fp32 G per global block), no gather pass (2 x 67 MB written, read back by this kernel).
template <int DQK, int DV, bool CAUSAL, int REL, bool PP, bool SPLIT = false, bool F16 = false, int QLV = 0>
__global__ __launch_bounds__((PP || SPLIT) ? 512 : 256, (PP || SPLIT) ? 1 : 2) void attn_kernel(AttnArgs a) {
    static_assert(!(SPLIT && F16), "fp16 operands are single-pass");
    static_assert(!QLV || (F16 && !SPLIT && !PP), "QLV: fp16 operands");
    static_assert(REL != 5 || (!SPLIT && !PP && DV == 80), "REL 5: SAM's global grid, the 4-wave block");
    constexpr bool QLO = QLV == 2;
    constexpr bool R2 = REL == 2 || REL == 5;  // rel_w seeds the accumulators, rel_h is one value per (query, key tile)
    constexpr uint32_t kOne16 = F16 ? 0x3C00u : 0x3F80u;  // 1.0 as a 16-bit operand
    // SPLIT: 8 waves of ONE 16-query tile each (the same 128 queries per block): two waves per SIMD to cover each other's LDS
    // and MFMA latencies - with two query tiles per wave the split kernel needs > 256 registers, i.e. one wave per SIMD (measured:
    // SAM global attention 1618 us, windows 302 us that way)
    constexpr int QT = SPLIT ? 1 : 2;          // 16-query tiles per wave
    constexpr int kQPerWave = 16 * QT;         // (shadows the namespace constant)
    constexpr int NT = (PP || SPLIT) ? 512 : 256;  // threads per block
    constexpr int kQBlk = PP ? 2 * kQPerBlock : kQPerBlock;
    constexpr int KS = DQK / 32;        // MFMA k-steps over the head dim
    constexpr int DT = DV / 16;         // 16-wide output tiles over the head dim
    constexpr int DCH = DV / 8;         // 16-byte chunks per K/V row actually present in memory
    constexpr int NCH = kKV * DCH;      // chunks per K (or V) tile
    constexpr int CPT = (NCH + NT - 1) / NT;
    // plane strides carry one extra row: the staging stores (8 consecutive lanes of a ds_write_b128 = 8 consecutive 16-byte
    // chunks of a key) then walk 8 distinct 16-byte bank slots instead of hitting one slot 2x (K) / 4x (V); the fragment
    // reads of one plane are only rotated by it
    constexpr int KPL = kKV * 32 + 32;   // elements per k-step plane  [key][32] (+ 64 B)
    constexpr int VPL = kKV * 16 + 16;   // elements per d-tile plane  [key][16] (+ 32 B)
    constexpr int KBUF = KS * KPL;
    constexpr int VBUF = DT * VPL;
    // (SPLIT: the tiles live in dynamic LDS - [hi planes | lo planes] per buffer, 2 x the bytes)
    __shared__ __attribute__((aligned(16))) bf16_t Ks_[2][SPLIT ? 8 : KBUF];
    __shared__ __attribute__((aligned(16))) bf16_t Vs_[2][SPLIT ? 8 : VBUF];
    // REL 2: rel_h[query][key row] of the block's queries, one value per (query, key tile).  Read from global at its point of use it is a dependent load in every tile, and the s_waitcnt the
    // compiler puts in front of it also drains the K/V prefetches of the following tiles.
    __shared__ __attribute__((aligned(16))) float Rh_[(R2 && !SPLIT) ? kQBlk * kKV : 4];
    // REL 4: per-wave scratch of the rel-pos table product G^T = T . Q^T (64 table rows x 16 queries, row stride 65 floats)
    constexpr int kGW = 16 * 65;
    __shared__ float Gs_[(REL == 4 && !SPLIT) ? (NT / 64) * kGW : 4];
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_dyn[];
    constexpr int KB2 = SPLIT ? 2 * KBUF : KBUF, VB2 = SPLIT ? 2 * VBUF : VBUF;  // elements per buffer
    bf16_t* const Ks0 = SPLIT ? reinterpret_cast<bf16_t*>(attn_dyn) : &Ks_[0][0];
    bf16_t* const Vs0 = SPLIT ? Ks0 + 2 * KB2 : &Vs_[0][0];
    float* const Rh = SPLIT ? reinterpret_cast<float*>(Vs0 + 2 * VB2) : &Rh_[0];
    float* const Gs = SPLIT ? Rh : &Gs_[0];  // (SPLIT: the dynamic region behind the tiles; REL 2 and REL 4 never coincide)
    auto Ks = [&](int buf) __attribute__((always_inline)) { return Ks0 + buf * KB2; };
    auto Vs = [&](int buf) __attribute__((always_inline)) { return Vs0 + buf * VB2; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int b = blockIdx.z, h = blockIdx.y, qblk = blockIdx.x;
    if (REL == 5) {
        // 1-D launch, XCD-aware: consecutive block ids go to the 8 XCDs round-robin, and the 32 query blocks of a (view, head) read
        // the SAME 1.3 MB of K / V - all of them on ONE XCD, so that its L2 fetches them once instead of every XCD fetching them
        // (PMC: 800 MB of fabric traffic per launch of which 670 MB were those eight-fold K / V fetches)
        const int nq = a.Sq / kQBlk, npairs = a.B * a.H;
        const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
        const int full = (npairs / 8) * 8;
        int pair = (j / nq) * 8 + xcd;
        qblk = j % nq;
        if (!a.xcd_map) {  // (A/B hook ivlm_attention_xcd_map(0): plain order, a pair's query blocks spread over all XCDs)
            pair = i / nq;
            qblk = i % nq;
        } else if (pair >= full) {  // tail pairs (B * H not a multiple of 8): plain order
            const int t = i - full * nq;
            pair = full + t / nq;
            qblk = t % nq;
        }
        b = pair / a.H;
        h = pair - b * a.H;
    }
    const int q_blk0 = qblk * kQBlk;
    const int q0 = q_blk0 + wave * kQPerWave;
    const int bkv = b / a.kv_batch_div;
    const bf16_t* __restrict__ Q = a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* __restrict__ K = a.k + bkv * a.k_bs + h * a.k_hs;
    const bf16_t* __restrict__ V = a.v + bkv * a.v_bs + h * a.v_hs;
    const bf16_t* __restrict__ Ql = (SPLIT || QLV) ? a.q_lo + b * a.q_bs + h * a.q_hs : nullptr;
    const bf16_t* __restrict__ Kl = SPLIT ? a.k_lo + bkv * a.k_bs + h * a.k_hs : nullptr;
    const bf16_t* __restrict__ Vl = SPLIT ? a.v_lo + bkv * a.v_bs + h * a.v_hs : nullptr;

    // ---- zero the pad chunks of K once (head dim < DQK), both buffers ------------------------------
    auto zero_pads = [&]() __attribute__((always_inline)) {
        if (DCH < KS * 4) {
            constexpr int NPAD = KS * 4 - DCH;
            for (int i = tid; i < 2 * kKV * NPAD; i += NT) {
                const int buf = i / (kKV * NPAD), r = i % (kKV * NPAD);
                const int key = r / NPAD, ch = DCH + r % NPAD;
                const int phys = (ch & 3) ^ (((key >> 3) & 1) << 1);
                *reinterpret_cast<u32x4_t*>(&Ks(buf)[(ch >> 2) * KPL + key * 32 + phys * 8]) = u32x4_t{0u, 0u, 0u, 0u};
                if (SPLIT) *reinterpret_cast<u32x4_t*>(&Ks(buf)[KBUF + (ch >> 2) * KPL + key * 32 + phys * 8]) = u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
    if (REL != 5) zero_pads();  // (REL 5 first uses the K buffers as the scratch of its table products)

    if (REL == 2) {  // rel_kh == 64 key rows (dispatch); rows of queries past Sq are clamped like the Q loads
        const int64_t bh = (int64_t)b * a.H + h;
        for (int i = tid; i < kQBlk * (kKV / 4); i += NT) {
            const int ql = i / (kKV / 4), c4 = i % (kKV / 4);
            int qi = q_blk0 + ql;
            qi = qi < a.Sq ? qi : a.Sq - 1;
            const float4 v4 = *reinterpret_cast<const float4*>(a.rel_h + (bh * a.Sq + qi) * a.rel_kh + c4 * 4);
            *reinterpret_cast<float4*>(&Rh[ql * kKV + c4 * 4]) = v4;
        }
    }

    // ---- Q fragments (B operand): lane holds Q[q0 + qt*16 + l15][(s*4+g)*8 .. +8] ---------------
    bf16x8_t qf[QT][KS];
    constexpr bool QL = SPLIT || QLV != 0;
    bf16x8_t qfl[QT][QL ? KS : 1];  // SPLIT / QLV: the lo halves
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int qi = q0 + qt * 16 + l15;
        qi = qi < a.Sq ? qi : a.Sq - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d0 = (s * 4 + g) * 8;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (d0 < DV) u = *reinterpret_cast<const uint4*>(Q + (int64_t)qi * a.q_rs + d0);
            if (QL) {
                u32x4_t ul = u32x4_t{0u, 0u, 0u, 0u};
                if (d0 < DV) ul = *reinterpret_cast<const u32x4_t*>(Ql + (int64_t)qi * a.q_rs + d0);
                qfl[qt][QL ? s : 0] = __builtin_bit_cast(bf16x8_t, ul);
            }
            qf[qt][s] = *reinterpret_cast<bf16x8_t*>(&u);
        }
    }
    // q * scale before the dot product (SAM, HF-CLIP).  Default precision: rounded to bf16 like the reference's bf16 model does;
    // SPLIT: in fp32 on the hi + lo value, split again.  (REL 4 first needs the UNSCALED q for the rel-pos terms: applied below.)
    auto prescale = [&]() __attribute__((always_inline)) {
        if (!a.prescale_q) return;
        const float sc = a.scale;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                u32x4_t uh = __builtin_bit_cast(u32x4_t, qf[qt][s_]);
                if (SPLIT) {
                    u32x4_t ul = __builtin_bit_cast(u32x4_t, qfl[qt][SPLIT ? s_ : 0]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = (__uint_as_float(uh[e] << 16) + __uint_as_float(ul[e] << 16)) * sc;
                        const float x1 = (__uint_as_float(uh[e] & 0xffff0000u) + __uint_as_float(ul[e] & 0xffff0000u)) * sc;
                        uint32_t hh, ll;
                        split_bf16x2(x0, x1, hh, ll);
                        uh[e] = hh;
                        ul[e] = ll;
                    }
                    qfl[qt][SPLIT ? s_ : 0] = __builtin_bit_cast(bf16x8_t, ul);
                } else if (QLO) {
                    u32x4_t ul = __builtin_bit_cast(u32x4_t, qfl[qt][QLO ? s_ : 0]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t hh, ll;
                        split_16x2((lo16<true>(uh[e]) + lo16<true>(ul[e])) * sc, (hi16<true>(uh[e]) + hi16<true>(ul[e])) * sc, hh, ll, 1);
                        uh[e] = hh;
                        ul[e] = ll;
                    }
                    qfl[qt][QLO ? s_ : 0] = __builtin_bit_cast(bf16x8_t, ul);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) uh[e] = pack2<F16>(lo16<F16>(uh[e]) * sc, hi16<F16>(uh[e]) * sc);
                }
                qf[qt][s_] = __builtin_bit_cast(bf16x8_t, uh);
            }
    };
    if (REL != 4 && REL != 5) prescale();

    f32x4_t o[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = kNegBig;
        l_run[qt] = 0.f;
    }

    int ntiles = (a.Sk + kKV - 1) / kKV;
    if (CAUSAL) {
        int last_q = q_blk0 + kQBlk - 1;
        last_q = last_q < a.Sq ? last_q : a.Sq - 1;
        const int last_key = last_q + a.q_pos0;  // inclusive
        const int lim = last_key / kKV + 1;
        ntiles = ntiles < lim ? ntiles : lim;
    }

    // ---- register staging of one K/V tile ------------------------------------------------------
    int ld_key[CPT], ld_d[CPT];  // this thread's chunks of a tile: key inside the tile, element offset inside the row
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int c = tid + i * NT;
        c = c < NCH ? c : NCH - 1;  // unconditional (clamped) loads keep the staging registers in registers across the loop
        ld_key[i] = c / DCH;
        ld_d[i] = (c % DCH) * 8;
    }
    const int last_key = a.Sk - 1;
    auto gload_k = [&](u32x4_t (&kreg)[CPT], int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            int key = t * kKV + ld_key[i];
            key = key < last_key ? key : last_key;
            kreg[i] = *reinterpret_cast<const u32x4_t*>(K + (int64_t)key * a.k_rs + ld_d[i]);
        }
    };
    auto gload_v = [&](u32x4_t (&vreg)[CPT], int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            int key = t * kKV + ld_key[i];
            key = key < last_key ? key : last_key;
            vreg[i] = *reinterpret_cast<const u32x4_t*>(V + (int64_t)key * a.v_rs + ld_d[i]);
        }
    };
    auto store_k = [&](const u32x4_t (&kreg)[CPT], int buf, int plane_off = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            if (c < NCH) {
                const int key = c / DCH, dch = c % DCH;
                const int phys = (dch & 3) ^ (((key >> 3) & 1) << 1);
                *reinterpret_cast<u32x4_t*>(&Ks(buf)[plane_off + (dch >> 2) * KPL + key * 32 + phys * 8]) = kreg[i];
            }
        }
    };
    auto store_v = [&](const u32x4_t (&vreg)[CPT], int buf, int plane_off = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            if (c < NCH) {
                const int key = c / DCH, dch = c % DCH;
                *reinterpret_cast<u32x4_t*>(&Vs(buf)[plane_off + (dch >> 1) * VPL + key * 16 + (dch & 1) * 8]) = vreg[i];
            }
        }
    };
    auto gload_lo = [&](u32x4_t (&reg)[CPT], const bf16_t* base, int64_t rs, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            int key = t * kKV + ld_key[i];
            key = key < last_key ? key : last_key;
            reg[i] = *reinterpret_cast<const u32x4_t*>(base + (int64_t)key * rs + ld_d[i]);
        }
    };
    const int kswz = (g ^ ((l15 >> 3) << 1)) * 8;                       // this lane's K chunk inside a 64-byte row
    const int voff = (g * 4 + (l15 >> 2)) * 16 + (l15 & 3) * 4;         // tr-read address: key g*4 + i/4, d-quad i%4

    // ---- rel-pos operands of this lane's two queries -------------------------------------------
    bf16x8_t qrel[QT];       // REL 1: [rel_h(KH) | rel_w(KW) | 0] features g*8 .. g*8+7
    bf16x8_t qrel_lo[QT];    //        (SPLIT: the fp32 bias values as hi + lo + lo2 operands - three bf16 terms carry all 24 bits:
    bf16x8_t qrel_lo2[QT];   //         the one-hot MFMAs add them exactly; a bias of +-20 split in two would be off by 1e-4)
    f32x4_t rwf[QT][4];      // REL 2: rel_w[q][kt*16 + g*4 + r]: tile-invariant, SEEDS the score accumulators (no add later)
    const float* rhp[QT] = {};
    const float* rwg[QT] = {};  // REL 3
    // packs 8 fp32 bias values (features g*8 .. g*8+7 of this lane's query) into the one-hot MFMA operand(s)
    auto pack_qrel = [&](const float (&f)[8], int qt) __attribute__((always_inline)) {
        uint4 u = make_uint4(pack2<F16>(f[0], f[1]), pack2<F16>(f[2], f[3]), pack2<F16>(f[4], f[5]), pack2<F16>(f[6], f[7]));
        if (F16) {  // the terms ADD to the scores: an fp16 rounding of a bias of +-8 would be an absolute 2e-3 on the score, so
                    // they travel as hi + lo fp16 operands (22 bits) of two one-hot MFMAs
            const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
            u32x4_t ul;
#pragma unroll
            for (int e = 0; e < 4; ++e) ul[e] = pack2<F16>(f[2 * e] - lo16<F16>(uu[e]), f[2 * e + 1] - hi16<F16>(uu[e]));
            qrel_lo[qt] = __builtin_bit_cast(bf16x8_t, ul);
        }
        if (SPLIT) {
            u32x4_t uh, ul, ul2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t hh, ll;
                split_bf16x2(f[2 * e], f[2 * e + 1], hh, ll);
                const float r0 = (f[2 * e] - __uint_as_float(hh << 16)) - __uint_as_float(ll << 16);
                const float r1 = (f[2 * e + 1] - __uint_as_float(hh & 0xffff0000u)) - __uint_as_float(ll & 0xffff0000u);
                uh[e] = hh;
                ul[e] = ll;
                ul2[e] = pack_bf16x2(r0, r1);
            }
            u = make_uint4(uh[0], uh[1], uh[2], uh[3]);
            qrel_lo[qt] = __builtin_bit_cast(bf16x8_t, ul);
            qrel_lo2[qt] = __builtin_bit_cast(bf16x8_t, ul2);
        }
        qrel[qt] = *reinterpret_cast<bf16x8_t*>(&u);
    };
    if (REL == 5) {
        // T = a.rel_h: 16-bit [>= 254, DV] = [rel_pos_h (rows 0 .. 126) ; rel_pos_w (rows 127 .. 253)] of the 64 x 64 grid.  A wave's 32
        // queries lie in ONE grid row qh (32 | 64), at columns qw0 + 0 .. 31 (qw0 = 0 or 32):
        //   rel_h[q][kh] = Q[q] . T_h[qh - kh + 63]: the 64 rows qh .. qh + 63 for all of them -> 4 row tiles, straight into the Rh block;
        //   rel_w[q][kw] = Q[q] . T_w[qw - kw + 63]: rows qw .. qw + 63, per 16-query tile the 79 rows from its first column on ->
        //   5 row tiles, through a per-wave scratch (overlaid on the K buffers, which are not yet in use) for the Toeplitz pick of
        //   each lane's 16 key columns.  39 MFMAs per query tile (x 2 with a lo half of q) against 2816 of the tile loop.
        constexpr int G = kKV;  // grid side = key tile = 64 (dispatch)
        const bf16_t* tab = reinterpret_cast<const bf16_t*>(a.rel_h);
        const int qh = q0 / G, qw0 = q0 - qh * G;
        constexpr int kSW = 16 * 81;  // floats per wave: 16 queries x (80 rows + 1)
        static_assert(REL != 5 || 4 * kSW * 4 <= 2 * KBUF * 2, "REL 5 scratch must fit the K buffers");
        float* Sw = reinterpret_cast<float*>(Ks0) + wave * kSW;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4_t gh[4], gw[5];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) gh[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) gw[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int wbase = qw0 + qt * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d0 = (ks * 4 + g) * 8;
#pragma unroll
                for (int rt = 0; rt < 9; ++rt) {
                    int row = rt < 4 ? qh + rt * 16 + l15 : (2 * G - 1) + wbase + (rt - 4) * 16 + l15;
                    row = row < 2 * (2 * G - 1) ? row : 2 * (2 * G - 1) - 1;  // (rows past the table are never picked)
                    u32x4_t t4 = u32x4_t{0u, 0u, 0u, 0u};
                    if (d0 < DV) t4 = *reinterpret_cast<const u32x4_t*>(tab + row * DV + d0);
                    const bf16x8_t tf = __builtin_bit_cast(bf16x8_t, t4);
                    f32x4_t& acc = rt < 4 ? gh[rt < 4 ? rt : 0] : gw[rt < 4 ? 0 : rt - 4];
                    acc = mma16<F16>(tf, qf[qt][ks], acc);
                    if (QL) acc = mma16<F16>(tf, qfl[qt][QL ? ks : 0], acc);
                }
            }
            // (bf16 operands: the terms rounded to bf16, as the bf16 reference materialises them; fp16 operands: fp32 terms)
            auto rnd = [](float x) __attribute__((always_inline)) { return F16 ? x : bf16_to_f32(f32_to_bf16(x)); };
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Rh[(wave * kQPerWave + qt * 16 + l15) * kKV + (G - 1) - (rt * 16 + g * 4 + r)] = rnd(gh[rt][r]);
#pragma unroll
            for (int rt = 0; rt < 5; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sw[l15 * 81 + rt * 16 + g * 4 + r] = gw[rt][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rwf[qt][kt][r] = rnd(Sw[l15 * 81 + l15 + (G - 1) - (kt * 16 + g * 4 + r)]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();  // the scratch is reused by the next query tile
        }
        prescale();
        __syncthreads();  // every wave is done with its scratch: the K buffers may be staged now
        zero_pads();
    } else if (REL == 4) {
        // The decomposed rel-pos terms computed HERE instead of by a kernel of their own (54 us and 120 MB of traffic per
        // windowed block): G^T = T . Q^T on the matrix cores - T = [rel_pos_h ; rel_pos_w ; 0] (64 rows, a.rel_h), Q the UNSCALED
        // query fragments already in registers - lands in C layout (lane = query l15, rows g*4+r of each 16-row tile), goes through
        // a per-wave LDS scratch and comes back as the 2 * side Toeplitz picks of this lane's query:
        //   rel_h[q][kh] = G[q][qh - kh + side - 1],   rel_w[q][kw] = G[q][(2 side - 1) + qw - kw + side - 1].
        const bf16_t* tab = reinterpret_cast<const bf16_t*>(a.rel_h);
        const int side = a.rel_kh;
        float* Gw = Gs + wave * kGW;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4_t gacc[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) gacc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d0 = (ks * 4 + g) * 8;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    u32x4_t t4 = u32x4_t{0u, 0u, 0u, 0u};
                    if (d0 < DV) t4 = *reinterpret_cast<const u32x4_t*>(tab + (rt * 16 + l15) * DV + d0);
                    const bf16x8_t tf = __builtin_bit_cast(bf16x8_t, t4);
                    gacc[rt] = mma16<F16>(tf, qf[qt][ks], gacc[rt]);
                    if (QL) gacc[rt] = mma16<F16>(tf, qfl[qt][QL ? ks : 0], gacc[rt]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gw[l15 * 65 + rt * 16 + g * 4 + r] = gacc[rt][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            int qi = q0 + qt * 16 + l15;
            qi = qi < a.Sq ? qi : a.Sq - 1;
            const int qh = qi / side, qw = qi - qh * side;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int fi = g * 8 + e;
                float v = 0.0f;
                if (fi < side) v = Gw[l15 * 65 + qh - fi + side - 1];
                else if (fi < 2 * side) v = Gw[l15 * 65 + (2 * side - 1) + qw - (fi - side) + side - 1];
                // default precision: the bf16 reference materialises the terms in bf16 (pack_qrel rounds); SPLIT keeps fp32
                f[e] = v;
            }
            pack_qrel(f, qt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();  // the scratch is reused by the next query tile
        }
        prescale();
    } else if (REL != 0) {
        const int64_t bh = (int64_t)b * a.H + h;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            int qi = q0 + qt * 16 + l15;
            qi = qi < a.Sq ? qi : a.Sq - 1;
            const float* rh = a.rel_h + (bh * a.Sq + qi) * a.rel_kh;
            const float* rw = a.rel_w + (bh * a.Sq + qi) * a.rel_kw;
            if (REL == 1) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int fi = g * 8 + e;
                    f[e] = fi < a.rel_kh ? rh[fi] : (fi < a.rel_kh + a.rel_kw ? rw[fi - a.rel_kh] : 0.0f);
                }
                pack_qrel(f, qt);
            } else if (REL == 3) {
                rhp[qt] = rh;
                rwg[qt] = rw;
            } else {
                rhp[qt] = rh;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float4 w4 = *reinterpret_cast<const float4*>(rw + kt * 16 + g * 4);
                    rwf[qt][kt] = f32x4_t{w4.x, w4.y, w4.z, w4.w};
                }
            }
        }
    }
    const float sc2 = a.prescale_q ? kLog2e : a.scale * kLog2e;

    // ---- the three phases of a key tile (state: s = scores then probabilities, pf = packed P^T fragments) ------------
    f32x4_t s[QT][4];
    bf16x8_t pf[QT][2];
    bf16x8_t pfl[QT][QL ? 2 : 1];  // SPLIT / QLO: lo halves of the probabilities
    auto nkt_of = [&](int t) __attribute__((always_inline)) {
        int nkt = (a.Sk - t * kKV + 15) >> 4;  // 16-key sub-tiles that hold real keys (wave-uniform)
        return nkt < 4 ? nkt : 4;
    };
    auto phase_qk = [&](const int t) __attribute__((always_inline)) {
        const bf16_t* Kb = Ks(t & 1);
        const int nkt = nkt_of(t);
        // ---- S^T = K . Q^T : s[qt][kt] holds keys kt*16 + g*4 + r of query l15 ------------------
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[qt][kt] = R2 ? rwf[qt][kt] : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (!R2 && kt >= nkt) continue;  // keys past Sk: their scores are masked below (REL 2 / 5: Sk % 64 == 0)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf =
                    *reinterpret_cast<const bf16x8_t*>(&Kb[ks * KPL + (kt * 16 + l15) * 32 + kswz]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s[qt][kt] = mma16<F16>(kf, qf[qt][ks], s[qt][kt]);
                    if (QLO) s[qt][kt] = mma16<F16>(kf, qfl[qt][QLO ? ks : 0], s[qt][kt]);
                }
                if (SPLIT) {
                    const bf16x8_t kfl =
                        *reinterpret_cast<const bf16x8_t*>(&Kb[KBUF + ks * KPL + (kt * 16 + l15) * 32 + kswz]);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        s[qt][kt] = mma16<F16>(kf, qfl[qt][ks], s[qt][kt]);
                        s[qt][kt] = mma16<F16>(kfl, qf[qt][ks], s[qt][kt]);
                    }
                }
            }
            if (REL == 1 || REL == 4) {
                const int key = t * kKV + kt * 16 + l15;
                const int kh = key / a.rel_kw;
                const int f1 = kh - g * 8, f2 = a.rel_kh + (key - kh * a.rel_kw) - g * 8;  // hot slots e
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lo = (f1 == 2 * j || f2 == 2 * j) ? kOne16 : 0u;
                    const uint32_t hi = (f1 == 2 * j + 1 || f2 == 2 * j + 1) ? (kOne16 << 16) : 0u;
                    w[j] = lo | hi;
                }
                uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
                const bf16x8_t hot = *reinterpret_cast<bf16x8_t*>(&u);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    s[qt][kt] = mma16<F16>(hot, qrel[qt], s[qt][kt]);
                    if (F16) s[qt][kt] = mma16<F16>(hot, qrel_lo[qt], s[qt][kt]);
                    if (SPLIT) {
                        s[qt][kt] = mma16<F16>(hot, qrel_lo[qt], s[qt][kt]);
                        s[qt][kt] = mma16<F16>(hot, qrel_lo2[qt], s[qt][kt]);
                    }
                }
            }
        }

    };
    auto phase_softmax = [&](const int t) __attribute__((always_inline)) {
        // ---- scale, bias, mask, online softmax (log2 domain) -----------------------------------
        // The softmax, not the MFMAs, bounds this kernel (32 scores per lane per tile on the VALU): ~8 VALU slots per score
        // (bias add, one fma into the log2 domain, max, subtract, raw v_exp_f32, sum; bf16 packing by v_cvt_pk_bf16_f32
        // below), and the accumulators are rescaled only when some running maximum moved.  Edge tiles (keys past Sk, the
        // causal diagonal) first overwrite their invalid raw scores with -1e30 in a pre-pass under one wave-uniform
        // branch; exp2 of those is exactly 0 because every query has a valid key in its first tile (key 0).
        const int kv0 = t * kKV;
        bool edge = !R2 && kv0 + kKV > a.Sk;  // (REL 2 / 5 are dispatched only when Sk is a whole number of tiles)
        if (CAUSAL) edge = edge || (kv0 + kKV - 1 > q0 + a.q_pos0);
        if (REL == 3 || edge) {
            asm volatile("" ::: "memory");  // keep this a real branch (if-converted it costs 2 VALU per score on EVERY tile)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int qi = q0 + qt * 16 + l15;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int kb = kv0 + kt * 16 + g * 4;
                    int gkh = 0, gkw = 0;
                    if (REL == 3) {
                        gkh = kb / a.rel_kw;
                        gkw = kb - gkh * a.rel_kw;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb + r;
                        bool ok = key < a.Sk;
                        if (REL == 3) {
                            if (ok) s[qt][kt][r] += rhp[qt][gkh] + rwg[qt][gkw];
                            if (++gkw == a.rel_kw) {
                                gkw = 0;
                                ++gkh;
                            }
                        }
                        if (CAUSAL) ok = ok && (key <= qi + a.q_pos0);
                        s[qt][kt][r] = ok ? s[qt][kt][r] : kNegBig;
                    }
                }
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // log2-domain score y = x * sc2 + bias2 (monotonic in x): the maximum is taken on the raw scores, and the
            // subtraction of the running maximum rides in the same fma as the scaling: p = exp2(x * sc2 + (bias2 - m))
            float bias2 = 0.0f;
            if (R2) bias2 = Rh[(wave * kQPerWave + qt * 16 + l15) * kKV + t] * sc2;
            float mx = kNegBig;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][kt][r]);
            mx = __builtin_fmaf(groups_max(mx), sc2, bias2);
            const float m_new = fmaxf(m_run[qt], mx);
            if (__any(m_new > m_run[qt])) {  // wave-uniform: some query's running maximum moved
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    o[qt][dt][0] *= alpha;
                    o[qt][dt][1] *= alpha;
                    o[qt][dt][2] *= alpha;
                    o[qt][dt][3] *= alpha;
                }
                m_run[qt] = m_new;
            }
            const float c2 = bias2 - m_new;
            float rs = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][kt][r], sc2, c2));
                    s[qt][kt][r] = p;
                    rs += p;
                }
            l_run[qt] += groups_sum(rs);
        }

        // ---- P^T fragments straight from the score registers (key order permuted per 32-step) --
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;  // round-to-nearest-even pairs in one instruction each (probabilities: finite, no NaN handling needed)
                if constexpr (F16) {  // (probabilities are in [0, 1]: no saturation needed)
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u.x) : "v"(s[qt][2 * s2][0]), "v"(s[qt][2 * s2][1]));
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u.y) : "v"(s[qt][2 * s2][2]), "v"(s[qt][2 * s2][3]));
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u.z) : "v"(s[qt][2 * s2 + 1][0]), "v"(s[qt][2 * s2 + 1][1]));
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u.w) : "v"(s[qt][2 * s2 + 1][2]), "v"(s[qt][2 * s2 + 1][3]));
                } else {
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.x) : "v"(s[qt][2 * s2][0]), "v"(s[qt][2 * s2][1]));
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.y) : "v"(s[qt][2 * s2][2]), "v"(s[qt][2 * s2][3]));
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.z) : "v"(s[qt][2 * s2 + 1][0]), "v"(s[qt][2 * s2 + 1][1]));
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u.w) : "v"(s[qt][2 * s2 + 1][2]), "v"(s[qt][2 * s2 + 1][3]));
                }
                pf[qt][s2] = *reinterpret_cast<bf16x8_t*>(&u);
                if (QLO) {  // lo halves: p - fp16(p)
                    uint4 ul;
                    const f32x4_t& sa = s[qt][2 * s2];
                    const f32x4_t& sb = s[qt][2 * s2 + 1];
                    ul.x = pack2<true>(sa[0] - lo16<true>(u.x), sa[1] - hi16<true>(u.x));
                    ul.y = pack2<true>(sa[2] - lo16<true>(u.y), sa[3] - hi16<true>(u.y));
                    ul.z = pack2<true>(sb[0] - lo16<true>(u.z), sb[1] - hi16<true>(u.z));
                    ul.w = pack2<true>(sb[2] - lo16<true>(u.w), sb[3] - hi16<true>(u.w));
                    pfl[qt][QLO ? s2 : 0] = *reinterpret_cast<bf16x8_t*>(&ul);
                }
                if (SPLIT) {  // lo halves: p - bf16(p), exactly representable differences rounded once
                    uint4 ul;
                    const f32x4_t& sa = s[qt][2 * s2];
                    const f32x4_t& sb = s[qt][2 * s2 + 1];
                    ul.x = pack_bf16x2(sa[0] - __uint_as_float(u.x << 16), sa[1] - __uint_as_float(u.x & 0xffff0000u));
                    ul.y = pack_bf16x2(sa[2] - __uint_as_float(u.y << 16), sa[3] - __uint_as_float(u.y & 0xffff0000u));
                    ul.z = pack_bf16x2(sb[0] - __uint_as_float(u.z << 16), sb[1] - __uint_as_float(u.z & 0xffff0000u));
                    ul.w = pack_bf16x2(sb[2] - __uint_as_float(u.w << 16), sb[3] - __uint_as_float(u.w & 0xffff0000u));
                    pfl[qt][s2] = *reinterpret_cast<bf16x8_t*>(&ul);
                }
            }

    };
    auto phase_pv = [&](const int t) __attribute__((always_inline)) {
        const bf16_t* Vb = Vs(t & 1);
        const int nkt = nkt_of(t);
        // ---- O^T += V^T . P^T  (key step outer: one uniform skip test per 32 keys, DT independent accumulators inner) ---
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (!R2 && 2 * s2 >= nkt) continue;  // all 32 keys of this step are past Sk (their P is 0)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                typedef __attribute__((ext_vector_type(4))) short s16x4_t;
                typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
                const bf16_t* vp = Vb + dt * VPL + (2 * s2) * 16 * 16 + voff;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vp));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vp + 16 * 16));
                typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                const s16x8_t v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, v8);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    o[qt][dt] = mma16<F16>(vf, pf[qt][s2], o[qt][dt]);
                    if (QLO) o[qt][dt] = mma16<F16>(vf, pfl[qt][QLO ? s2 : 0], o[qt][dt]);
                }
                if (SPLIT) {
                    const bf16_t* vpl = vp + VBUF;
                    const s16x4_t lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vpl));
                    const s16x4_t hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vpl + 16 * 16));
                    const s16x8_t v8l = __builtin_shufflevector(lo2, hi2, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8_t vfl = __builtin_bit_cast(bf16x8_t, v8l);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        o[qt][dt] = mma16<F16>(vf, pfl[qt][s2], o[qt][dt]);
                        o[qt][dt] = mma16<F16>(vfl, pf[qt][s2], o[qt][dt]);
                    }
                }
            }
        }
    };

    if (SPLIT) {
        // one register set (hi + lo of K and V): tile t + 1 is loaded while tile t is computed and written to the other buffer
        // at the end of the step; one barrier per tile (three MFMAs per fragment give the loads three times the cover)
        u32x4_t kh[CPT], kl[CPT], vh[CPT], vl[CPT];
        gload_k(kh, 0);
        gload_lo(kl, Kl, a.k_rs, 0);
        gload_v(vh, 0);
        gload_lo(vl, Vl, a.v_rs, 0);
        store_k(kh, 0);
        store_k(kl, 0, KBUF);
        store_v(vh, 0);
        store_v(vl, 0, VBUF);
        for (int t = 0; t < ntiles; ++t) {
            sync_lds();
            if (t + 1 < ntiles) {
                gload_k(kh, t + 1);
                gload_lo(kl, Kl, a.k_rs, t + 1);
                gload_v(vh, t + 1);
                gload_lo(vl, Vl, a.v_rs, t + 1);
            }
            phase_qk(t);
            phase_softmax(t);
            phase_pv(t);
            if (t + 1 < ntiles) {
                store_k(kh, (t + 1) & 1);
                store_k(kl, (t + 1) & 1, KBUF);
                store_v(vh, (t + 1) & 1);
                store_v(vl, (t + 1) & 1, VBUF);
            }
        }
    } else if (!PP) {
        // two register sets: the loads of tile t + 2 are issued while tile t is computed, tile t + 1 - loaded one step
        // earlier - is written to the free LDS buffer at the end of the step; one barrier per tile
        u32x4_t kregA[CPT], vregA[CPT], kregB[CPT], vregB[CPT];
        auto tile = [&](const int t, u32x4_t (&kld)[CPT], u32x4_t (&vld)[CPT], const u32x4_t (&kst)[CPT],
                        const u32x4_t (&vst)[CPT]) __attribute__((always_inline)) {
            sync_lds();  // tile t is in buffer t & 1; every wave is done with tile t - 1 (the other buffer)
            if (t + 2 < ntiles) {
                gload_k(kld, t + 2);
                gload_v(vld, t + 2);
            }
            phase_qk(t);
            phase_softmax(t);
            phase_pv(t);
            if (t + 1 < ntiles) {  // the other buffer: last read in step t - 1, before this barrier round
                store_k(kst, (t + 1) & 1);
                store_v(vst, (t + 1) & 1);
            }
        };
        gload_k(kregA, 0);
        gload_v(vregA, 0);
        if (ntiles > 1) {
            gload_k(kregB, 1);
            gload_v(vregB, 1);
        }
        store_k(kregA, 0);
        store_v(vregA, 0);
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, kregA, vregA, kregB, vregB);                           // loads tile t + 2 -> A, stores tile t + 1 <- B
            if (t + 1 < ntiles) tile(t + 1, kregB, vregB, kregA, vregA);   // loads tile t + 3 -> B, stores tile t + 2 <- A
        }
    } else {
        // Ping-pong.  Per wave the order is M(0) X(0) M(1) X(1) ... M(n) with M(t) = P.V(t-1) then Q.K^T(t) [matrix unit] and
        // X(t) = softmax(t) [VALU]; group 1 runs one barrier interval behind group 0.  Interval 2t: group 0 in M(t), group 1
        // in X(t-1); interval 2t+1: group 0 in X(t), group 1 in M(t).  Both intervals of a pair read K(t) and V(t-1), so
        // K(t+1) and V(t) are loaded at the start of interval 2t and written to the other buffers at the end of interval
        // 2t+1 (their previous contents, K(t-1) / V(t-2), were last read in interval 2t-1).
        const int grp = wave >> 2;
        u32x4_t kreg[CPT], vreg[CPT];
        gload_k(kreg, 0);
        store_k(kreg, 0);
        for (int t = 0; t <= ntiles; ++t) {
            sync_lds();  // ---- interval 2t
            if (t + 1 < ntiles) gload_k(kreg, t + 1);
            if (t < ntiles) gload_v(vreg, t);
            if (grp == 0) {
                if (t > 0) phase_pv(t - 1);
                if (t < ntiles) phase_qk(t);
            } else if (t > 0) {
                phase_softmax(t - 1);
            }
            sync_lds();  // ---- interval 2t + 1
            if (grp == 0) {
                if (t < ntiles) phase_softmax(t);
            } else {
                if (t > 0) phase_pv(t - 1);
                if (t < ntiles) phase_qk(t);
            }
            if (t + 1 < ntiles) store_k(kreg, (t + 1) & 1);
            if (t < ntiles) store_v(vreg, t & 1);
        }
    }

    // ---- normalise and store: lane holds O[query l15][d = dt*16 + g*4 + r] ---------------------
    bf16_t* __restrict__ O = a.o + b * a.o_bs + h * a.o_hs;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = q0 + qt * 16 + l15;
        if (qi >= a.Sq) continue;
        const float inv = l_run[qt] > 0.0f ? 1.0f / l_run[qt] : 0.0f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (SPLIT) {
                uint2 wh, wl;
                split_bf16x2(o[qt][dt][0] * inv, o[qt][dt][1] * inv, wh.x, wl.x);
                split_bf16x2(o[qt][dt][2] * inv, o[qt][dt][3] * inv, wh.y, wl.y);
                *reinterpret_cast<uint2*>(O + (int64_t)qi * a.o_rs + dt * 16 + g * 4) = wh;
                *reinterpret_cast<uint2*>(a.o_lo + b * a.o_bs + h * a.o_hs + (int64_t)qi * a.o_rs + dt * 16 + g * 4) = wl;
                continue;
            }
            const uint2 w = make_uint2(pack2<F16>(o[qt][dt][0] * inv, o[qt][dt][1] * inv),
                                       pack2<F16>(o[qt][dt][2] * inv, o[qt][dt][3] * inv));
            *reinterpret_cast<uint2*>(O + (int64_t)qi * a.o_rs + dt * 16 + g * 4) = w;
        }
    }
}

// ======================================================================================================================
// Window attention of the SAM ViT (14 x 14 windows: S = 196 keys, head dim 80, decomposed rel-pos bias), second generation.
// The generic flash kernel above tiles keys by 64 and queries by 128: a 196-token window then runs a fourth key tile holding 4
// keys, a second query block that is 53 % full, and re-stages K / V for it - with only four tiles per block the load pipeline
// never fills (129 us per windowed block of 4 views; 28 of the encoder's 32 blocks are windowed).  Here ONE block owns one
// (window, head): the whole K and V of the window sit in LDS (one barrier in total), 8 waves take the 13 query tiles of 16, and a
// query tile is a single pass - all 208 (padded) scores in registers, no online-softmax rescaling:
//     table product G^T = T . Q^T (rel-pos terms, see REL 4 above)  ->  S^T = K . Q^T + one-hot . rel   ->  softmax  ->  O^T = V^T . P^T.
// SPLIT: hi + lo planes of q / k / v, three MFMAs per fragment, fp32 rel-pos terms, hi + lo output ("parity" precision); the hi + lo
// K / V planes leave no LDS for the table-product scratch, so SPLIT reads the rel-pos terms from the fp32 arrays of
// ivlm_relpos_bias_split (array mode) instead of computing them here (table mode).
// QLV (with F16): q = hi + lo IEEE halves (a.q_lo).  1: the lo half enters the rel-pos table product only - the terms are what
// amplifies q's rounding (a term is q . R with |R| ~ 6 x the |k| * scale of the score product: tools/emulate_f16_sites.py), Q.K^T and
// P.V stay single fp16: 12 extra MFMAs per query tile; 2: the lo half also enters Q.K^T and the softmax weights are split for P.V
// (3 % closer, +74 MFMAs per tile).
template <bool SPLIT, bool F16 = false, int QLV = 0>
__global__ __launch_bounds__(SPLIT ? 512 : 1024, 1) void win_attn_kernel(AttnArgs a) {
    static_assert(!(SPLIT && F16), "fp16 operands are single-pass");
    static_assert(!QLV || F16, "QLV: q as hi + lo IEEE halves");
    constexpr bool QLO = QLV == 2;          // lo half in Q.K^T, split softmax weights
    constexpr bool QL = SPLIT || QLV != 0;  // a lo half of q is loaded
    constexpr uint32_t kOne16 = F16 ? 0x3C00u : 0x3F80u;  // 1.0 as a 16-bit operand
    constexpr bool TAB = !SPLIT;
    // SPLIT: 8 waves (two per SIMD, 240 registers: held under 256 with scheduling barriers in the fragment loops), the 13 query tiles
    // in two rounds.  Default precision: 126 registers allow 16 waves (four per SIMD) - every query tile has a wave of its own (ONE
    // round; the kernel is latency-bound: ~12 % MFMA utilisation), the other three waves only help staging.
    constexpr int NT = SPLIT ? 512 : 1024, NWV = NT / 64;
    constexpr int NGW = SPLIT ? 8 : 13;  // waves that own a table-product scratch slice (a query tile)
    constexpr int DV = 80, KS = 3, DT = 5, DCH = 10;
    constexpr int KT = 13;            // 16-key tiles (208 padded keys) of the score pass
    constexpr int VS = 7;             // 32-key steps (224 padded keys) of the P.V pass
    constexpr int SPK = KT * 16, SPV = VS * 32;
    constexpr int KPL = SPK * 32 + 32, VPL = SPV * 16 + 16;   // plane strides (elements), padded like the tiles of attn_kernel
    constexpr int KBUF = KS * KPL, VBUF = DT * VPL;
    constexpr int kGW = 16 * 65;
    extern __shared__ __attribute__((aligned(16))) unsigned char win_dyn[];
    bf16_t* const Kh = reinterpret_cast<bf16_t*>(win_dyn);
    bf16_t* const Kl = Kh + KBUF;                              // (SPLIT only)
    bf16_t* const Vh = Kh + (SPLIT ? 2 : 1) * KBUF;
    bf16_t* const Vl = Vh + VBUF;                              // (SPLIT only)
    u32x4_t* const Hot = reinterpret_cast<u32x4_t*>(Vh + (SPLIT ? 2 : 1) * VBUF);  // one-hot rel-pos operands [KT][64 lanes]
    float* const Gs = reinterpret_cast<float*>(Hot + KT * 64);                    // (table mode only)
    bf16_t* const Tb = reinterpret_cast<bf16_t*>(Gs + NGW * kGW);                   // (table mode) the rel-pos table, K-plane layout
    constexpr int TPL = 64 * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-aware block -> (window, head) map: consecutive block ids go to the 8 XCDs round-robin, and the 16 heads of a window read
    // the SAME q|k|v rows (a head is a 160-byte segment of every 7680-byte row), so all heads of a window are given to ONE XCD,
    // consecutively: its L2 then fetches the window's rows once instead of every XCD fetching them for its two heads
    int b, h;
    {
        const int nb = a.B, H = a.H;
        const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
        const int full = (nb / 8) * 8;             // windows covered by whole rounds of 8
        const int w = (j / H) * 8 + xcd;
        if (w < full) {
            b = w;
            h = j % H;
        } else {                                   // tail windows (nb % 8): plain order
            const int t = i - full * H;
            b = full + t / H;
            h = t % H;
        }
    }
    const int S = a.Sq, side = a.rel_kh;
    const bf16_t* __restrict__ Q = a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* __restrict__ K = a.k + b * a.k_bs + h * a.k_hs;
    const bf16_t* __restrict__ V = a.v + b * a.v_bs + h * a.v_hs;
    const bf16_t* __restrict__ Ql = QL ? a.q_lo + b * a.q_bs + h * a.q_hs : nullptr;
    const bf16_t* __restrict__ Kl_g = SPLIT ? a.k_lo + b * a.k_bs + h * a.k_hs : nullptr;
    const bf16_t* __restrict__ Vl_g = SPLIT ? a.v_lo + b * a.v_bs + h * a.v_hs : nullptr;

    // ---- stage the whole window: K planes [k-step][key][32] (chunk swizzle of attn_kernel), V planes [d tile][key][16];
    //      padded keys and the two pad chunks of K's third plane are zero -----------------------------------------------
    // (all global loads of the block are issued before the first LDS store: a load -> store loop serialises ~5 memory round trips)
    constexpr int NKC = (SPK * 12 + NT - 1) / NT, NVC = (SPV * DCH + NT - 1) / NT;
    const u32x4_t zero4 = u32x4_t{0u, 0u, 0u, 0u};
    {
        u32x4_t kr[NKC], krl[SPLIT ? NKC : 1];
        u32x4_t vr[SPLIT ? 1 : NVC];  // (default precision: K and V in flight together; SPLIT: K, then V - register budget)
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = tid + i * NT;
            const int key = c / 12, dch = c % 12;
            const bool ok = key < S && dch < DCH;
            const int64_t off = (int64_t)(ok ? key : 0) * a.k_rs + (ok ? dch : 0) * 8;
            kr[i] = *reinterpret_cast<const u32x4_t*>(K + off);
            if (SPLIT) krl[SPLIT ? i : 0] = *reinterpret_cast<const u32x4_t*>(Kl_g + off);
        }
        if (!SPLIT) {
#pragma unroll
            for (int i = 0; i < NVC; ++i) {
                const int c = tid + i * NT;
                const int key = c / DCH, dch = c % DCH;
                vr[SPLIT ? 0 : i] = *reinterpret_cast<const u32x4_t*>(V + (int64_t)(key < S ? key : 0) * a.v_rs + dch * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = tid + i * NT;
            if (c < SPK * 12) {
                const int key = c / 12, dch = c % 12;
                const bool ok = key < S && dch < DCH;
                const int phys = (dch & 3) ^ (((key >> 3) & 1) << 1);
                const int off = (dch >> 2) * KPL + key * 32 + phys * 8;
                *reinterpret_cast<u32x4_t*>(&Kh[off]) = ok ? kr[i] : zero4;
                if (SPLIT) *reinterpret_cast<u32x4_t*>(&Kl[off]) = ok ? krl[SPLIT ? i : 0] : zero4;
            }
        }
        if (!SPLIT) {
#pragma unroll
            for (int i = 0; i < NVC; ++i) {
                const int c = tid + i * NT;
                if (c < SPV * DCH) {
                    const int key = c / DCH, dch = c % DCH;
                    *reinterpret_cast<u32x4_t*>(&Vh[(dch >> 1) * VPL + key * 16 + (dch & 1) * 8]) = key < S ? vr[SPLIT ? 0 : i] : zero4;
                }
            }
        }
    }
    if (SPLIT) {
        u32x4_t vr[NVC], vrl[NVC];
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = tid + i * NT;
            const int key = c / DCH, dch = c % DCH;
            const int64_t off = (int64_t)(key < S ? key : 0) * a.v_rs + dch * 8;
            vr[i] = *reinterpret_cast<const u32x4_t*>(V + off);
            vrl[i] = *reinterpret_cast<const u32x4_t*>(Vl_g + off);
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = tid + i * NT;
            if (c < SPV * DCH) {
                const int key = c / DCH, dch = c % DCH;
                const int off = (dch >> 1) * VPL + key * 16 + (dch & 1) * 8;
                *reinterpret_cast<u32x4_t*>(&Vh[off]) = key < S ? vr[i] : zero4;
                *reinterpret_cast<u32x4_t*>(&Vl[off]) = key < S ? vrl[i] : zero4;
            }
        }
    }
    if (TAB) {  // [rel_pos_h ; rel_pos_w ; 0] (64 rows x 80) as three k-step planes [row][32], chunk swizzle as K: 12 KB
        const bf16_t* tabg = reinterpret_cast<const bf16_t*>(a.rel_h);
        for (int c = tid; c < 64 * 12; c += NT) {
            const int row = c / 12, dch = c % 12;
            u32x4_t v4 = u32x4_t{0u, 0u, 0u, 0u};
            if (dch < DCH) v4 = *reinterpret_cast<const u32x4_t*>(tabg + row * DV + dch * 8);
            const int phys = (dch & 3) ^ (((row >> 3) & 1) << 1);
            *reinterpret_cast<u32x4_t*>(&Tb[(dch >> 2) * TPL + row * 32 + phys * 8]) = v4;
        }
    }
    // the one-hot (kh, side + kw) operand of every 16-key tile depends on the key alone: built once per block (13 KB), read back
    // as one 16-byte fragment per tile by every query tile
    for (int c = tid; c < (TAB ? KT * 64 : 0); c += NT) {  // (SPLIT: no LDS left - built per tile in registers)
        const int kt = c >> 6, ln = c & 63;
        const int key = kt * 16 + (ln & 15), gg = ln >> 4;
        const int kh = key / a.rel_kh;
        const int f1 = kh - gg * 8, f2 = a.rel_kh + (key - kh * a.rel_kh) - gg * 8;
        u32x4_t w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = (f1 == 2 * j || f2 == 2 * j) ? kOne16 : 0u;
            const uint32_t hi = (f1 == 2 * j + 1 || f2 == 2 * j + 1) ? (kOne16 << 16) : 0u;
            w[j] = lo | hi;
        }
        Hot[c] = w;
    }
    __syncthreads();

    const int kswz = (g ^ ((l15 >> 3) << 1)) * 8;
    const int voff = (g * 4 + (l15 >> 2)) * 16 + (l15 & 3) * 4;
    float* Gw = Gs + (wave < NGW ? wave : 0) * kGW;  // (waves without a query tile never touch it)
    const float sc = a.scale;
    const int nqt = (S + 15) >> 4;

    // the (unscaled) query fragments of a wave's NEXT tile are fetched while it works on the current one
    bf16x8_t qn[KS], qnl[QL ? KS : 1];
    auto fetch_q = [&](int qtile) __attribute__((always_inline)) {
        int qi = qtile * 16 + l15;
        qi = qi < S ? qi : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = (ks * 4 + g) * 8;
            u32x4_t u = u32x4_t{0u, 0u, 0u, 0u}, ul = u;
            if (d0 < DV) {
                u = *reinterpret_cast<const u32x4_t*>(Q + (int64_t)qi * a.q_rs + d0);
                if (QL) ul = *reinterpret_cast<const u32x4_t*>(Ql + (int64_t)qi * a.q_rs + d0);
            }
            qn[ks] = __builtin_bit_cast(bf16x8_t, u);
            if (QL) qnl[QL ? ks : 0] = __builtin_bit_cast(bf16x8_t, ul);
        }
    };
    if (wave < nqt) fetch_q(wave);
    for (int qtile = wave; qtile < nqt; qtile += NWV) {
        int qi = qtile * 16 + l15;
        const bool q_ok = qi < S;
        qi = q_ok ? qi : S - 1;
        // ---- query fragments (unscaled) and the rel-pos table product --------------------------------------------------
        bf16x8_t qf[KS], qfl[QL ? KS : 1];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = qn[ks];
            if (QL) qfl[QL ? ks : 0] = qnl[QL ? ks : 0];
        }
        // (16 waves: S <= 208 is at most 13 query tiles, a wave never has a second one - no prefetch registers held across the tile)
        constexpr bool ONE = NWV >= 13;
        if (!ONE && qtile + NWV < nqt) fetch_q(qtile + NWV);
        float f[8];
        if (TAB) {
            f32x4_t gacc[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) gacc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const bf16x8_t tf = *reinterpret_cast<const bf16x8_t*>(&Tb[ks * TPL + (rt * 16 + l15) * 32 + kswz]);
                    gacc[rt] = mma16<F16>(tf, qf[ks], gacc[rt]);
                    if (QLV) gacc[rt] = mma16<F16>(tf, qfl[QLV ? ks : 0], gacc[rt]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gw[l15 * 65 + rt * 16 + g * 4 + r] = gacc[rt][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int qh = qi / side, qw = qi - qh * side;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int fi = g * 8 + e;
                float v = 0.0f;
                if (fi < side) v = Gw[l15 * 65 + qh - fi + side - 1];
                else if (fi < 2 * side) v = Gw[l15 * 65 + (2 * side - 1) + qw - (fi - side) + side - 1];
                f[e] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {  // array mode: rel_h / rel_w f32 [B*H, S, side]
            const int64_t bq = ((int64_t)b * a.H + h) * S + qi;
            const float* rh = a.rel_h + bq * side;
            const float* rw = a.rel_w + bq * side;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int fi = g * 8 + e;
                f[e] = fi < side ? rh[fi] : (fi < 2 * side ? rw[fi - side] : 0.0f);
            }
        }
        bf16x8_t qrel, qrel_lo, qrel_lo2;
        {
            u32x4_t uh, ul, ul2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t hh, ll = 0;
                if constexpr (F16) {  // hi + lo fp16 operands (22 bits) of two one-hot MFMAs: see attn_kernel
                    hh = pack_f16x2(f[2 * e], f[2 * e + 1]);
                    ul[e] = pack_f16x2(f[2 * e] - lo16<true>(hh), f[2 * e + 1] - hi16<true>(hh));
                } else {
                    split_bf16x2(f[2 * e], f[2 * e + 1], hh, ll);
                }
                uh[e] = hh;
                if (SPLIT) {
                    const float r0 = (f[2 * e] - __uint_as_float(hh << 16)) - __uint_as_float(ll << 16);
                    const float r1 = (f[2 * e + 1] - __uint_as_float(hh & 0xffff0000u)) - __uint_as_float(ll & 0xffff0000u);
                    ul[e] = ll;
                    ul2[e] = pack_bf16x2(r0, r1);
                }
            }
            qrel = __builtin_bit_cast(bf16x8_t, uh);  // default precision: the terms rounded to bf16, as the bf16 reference holds them
            if (F16) qrel_lo = __builtin_bit_cast(bf16x8_t, ul);
            if (SPLIT) {
                qrel_lo = __builtin_bit_cast(bf16x8_t, ul);
                qrel_lo2 = __builtin_bit_cast(bf16x8_t, ul2);
            }
        }
        // ---- q * scale (bf16-rounded in default precision like SAM's bf16 model; fp32 on hi + lo in SPLIT) --------------
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4_t uh = __builtin_bit_cast(u32x4_t, qf[ks]);
            if (SPLIT) {
                u32x4_t ul = __builtin_bit_cast(u32x4_t, qfl[SPLIT ? ks : 0]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = (__uint_as_float(uh[e] << 16) + __uint_as_float(ul[e] << 16)) * sc;
                    const float x1 = (__uint_as_float(uh[e] & 0xffff0000u) + __uint_as_float(ul[e] & 0xffff0000u)) * sc;
                    uint32_t hh, ll;
                    split_bf16x2(x0, x1, hh, ll);
                    uh[e] = hh;
                    ul[e] = ll;
                }
                qfl[SPLIT ? ks : 0] = __builtin_bit_cast(bf16x8_t, ul);
            } else if (QLO) {
                u32x4_t ul = __builtin_bit_cast(u32x4_t, qfl[QLO ? ks : 0]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t hh, ll;
                    split_16x2((lo16<true>(uh[e]) + lo16<true>(ul[e])) * sc, (hi16<true>(uh[e]) + hi16<true>(ul[e])) * sc, hh, ll, 1);
                    uh[e] = hh;
                    ul[e] = ll;
                }
                qfl[QLO ? ks : 0] = __builtin_bit_cast(bf16x8_t, ul);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    uh[e] = pack2<F16>(lo16<F16>(uh[e]) * sc, hi16<F16>(uh[e]) * sc);
            }
            qf[ks] = __builtin_bit_cast(bf16x8_t, uh);
        }
        // ---- S^T = K . Q^T + bias: s[kt] holds keys kt*16 + g*4 + r of query l15 -----------------------------------------
        f32x4_t s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&Kh[ks * KPL + (kt * 16 + l15) * 32 + kswz]);
                s[kt] = mma16<F16>(kf, qf[ks], s[kt]);
                if (QLO) s[kt] = mma16<F16>(kf, qfl[QLO ? ks : 0], s[kt]);
                if (SPLIT) {
                    const bf16x8_t kfl = *reinterpret_cast<const bf16x8_t*>(&Kl[ks * KPL + (kt * 16 + l15) * 32 + kswz]);
                    s[kt] = mma16<F16>(kf, qfl[SPLIT ? ks : 0], s[kt]);
                    s[kt] = mma16<F16>(kfl, qf[ks], s[kt]);
                }
            }
            bf16x8_t hot;
            if (TAB) {
                hot = __builtin_bit_cast(bf16x8_t, Hot[kt * 64 + lane]);
            } else {
                const int key = kt * 16 + l15;
                const int kh = key / side;
                const int f1 = kh - g * 8, f2 = side + (key - kh * side) - g * 8;
                u32x4_t w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lo = (f1 == 2 * j || f2 == 2 * j) ? kOne16 : 0u;
                    const uint32_t hi = (f1 == 2 * j + 1 || f2 == 2 * j + 1) ? (kOne16 << 16) : 0u;
                    w[j] = lo | hi;
                }
                hot = __builtin_bit_cast(bf16x8_t, w);
            }
            s[kt] = mma16<F16>(hot, qrel, s[kt]);
            if (F16) s[kt] = mma16<F16>(hot, qrel_lo, s[kt]);
            if (SPLIT) {
                s[kt] = mma16<F16>(hot, qrel_lo, s[kt]);
                s[kt] = mma16<F16>(hot, qrel_lo2, s[kt]);
                if (kt & 1) __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting all 78 fragment reads: spills)
            }
        }
        // ---- one-pass softmax over the S valid keys (log2 domain; padded keys -> probability exactly 0) -------------------
        constexpr int ktl = KT - 1;  // the only tile that can hold padded keys
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ktl * 16 + g * 4 + r >= S) s[ktl][r] = kNegBig;
        float mx = kNegBig;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = groups_max(mx) * kLog2e;
        float rs = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], kLog2e, -mx));
                s[kt][r] = p;
                rs += p;
            }
        const float inv = 1.0f / groups_sum(rs);
        // ---- O^T = V^T . P^T over 7 steps of 32 keys (the 14th 16-key tile does not exist: zeros) -------------------------
        f32x4_t o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < VS; ++s2) {
            const f32x4_t sa = s[2 * s2];
            const f32x4_t sb = (2 * s2 + 1 < KT) ? s[(2 * s2 + 1 < KT) ? 2 * s2 + 1 : 0] : f32x4_t{0.f, 0.f, 0.f, 0.f};
            u32x4_t u, ul;
            u[0] = pack2<F16>(sa[0], sa[1]);
            u[1] = pack2<F16>(sa[2], sa[3]);
            u[2] = pack2<F16>(sb[0], sb[1]);
            u[3] = pack2<F16>(sb[2], sb[3]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, u);
            bf16x8_t pfl;
            if (QLO) {
                ul[0] = pack2<true>(sa[0] - lo16<true>(u[0]), sa[1] - hi16<true>(u[0]));
                ul[1] = pack2<true>(sa[2] - lo16<true>(u[1]), sa[3] - hi16<true>(u[1]));
                ul[2] = pack2<true>(sb[0] - lo16<true>(u[2]), sb[1] - hi16<true>(u[2]));
                ul[3] = pack2<true>(sb[2] - lo16<true>(u[3]), sb[3] - hi16<true>(u[3]));
                pfl = __builtin_bit_cast(bf16x8_t, ul);
            }
            if (SPLIT) {
                ul[0] = pack_bf16x2(sa[0] - __uint_as_float(u[0] << 16), sa[1] - __uint_as_float(u[0] & 0xffff0000u));
                ul[1] = pack_bf16x2(sa[2] - __uint_as_float(u[1] << 16), sa[3] - __uint_as_float(u[1] & 0xffff0000u));
                ul[2] = pack_bf16x2(sb[0] - __uint_as_float(u[2] << 16), sb[1] - __uint_as_float(u[2] & 0xffff0000u));
                ul[3] = pack_bf16x2(sb[2] - __uint_as_float(u[3] << 16), sb[3] - __uint_as_float(u[3] & 0xffff0000u));
                pfl = __builtin_bit_cast(bf16x8_t, ul);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                typedef __attribute__((ext_vector_type(4))) short s16x4_t;
                typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
                typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                const bf16_t* vp = Vh + dt * VPL + (2 * s2) * 16 * 16 + voff;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vp));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vp + 16 * 16));
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                o[dt] = mma16<F16>(vf, pf, o[dt]);
                if (QLO) o[dt] = mma16<F16>(vf, pfl, o[dt]);
                if (SPLIT) {
                    const bf16_t* vpl = Vl + dt * VPL + (2 * s2) * 16 * 16 + voff;
                    const s16x4_t lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vpl));
                    const s16x4_t hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vpl + 16 * 16));
                    const bf16x8_t vfl = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo2, hi2, 0, 1, 2, 3, 4, 5, 6, 7));
                    o[dt] = mma16<F16>(vf, pfl, o[dt]);
                    o[dt] = mma16<F16>(vfl, pf, o[dt]);
                }
            }
            if (SPLIT) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- normalise and store: lane holds O[query l15][d = dt*16 + g*4 + r] ---------------------------------------------
        if (q_ok) {
            bf16_t* __restrict__ O = a.o + b * a.o_bs + h * a.o_hs + (int64_t)qi * a.o_rs;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (SPLIT) {
                    uint2 wh, wl;
                    split_bf16x2(o[dt][0] * inv, o[dt][1] * inv, wh.x, wl.x);
                    split_bf16x2(o[dt][2] * inv, o[dt][3] * inv, wh.y, wl.y);
                    *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = wh;
                    *reinterpret_cast<uint2*>(a.o_lo + b * a.o_bs + h * a.o_hs + (int64_t)qi * a.o_rs + dt * 16 + g * 4) = wl;
                } else {
                    *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) =
                        make_uint2(pack2<F16>(o[dt][0] * inv, o[dt][1] * inv), pack2<F16>(o[dt][2] * inv, o[dt][3] * inv));
                }
            }
        }
        if (ONE) break;
    }
}

static int g_xcd_map = 1;  // REL 5: XCD-aware block map (A/B hook: ivlm_attention_xcd_map)
static int g_win_v2 = 1;  // 0: the generic flash kernel for windows too (A/B hook: ivlm_attention_window_kernel)

template <bool SPLIT, bool F16 = false, int QLV = 0>
static int launch_win(const AttnArgs& a, hipStream_t st) {
    constexpr int KS = 3, DT = 5, KPL = 13 * 16 * 32 + 32, VPL = 7 * 32 * 16 + 16;
    constexpr size_t lds = (size_t)(SPLIT ? 2 : 1) * (KS * KPL + DT * VPL) * 2 +
                           (SPLIT ? 0 : (size_t)13 * 64 * 16 + 13 * 16 * 65 * 4 + 3 * 64 * 32 * 2);
    static_assert(lds <= 160 * 1024, "window tiles must fit the LDS");
    auto kfn = win_attn_kernel<SPLIT, F16, QLV>;
    static ivlm_dev_mask_t attr_set{0};
    if (ivlm_dev_pending(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ivlm_dev_done(attr_set);
    }
    kfn<<<dim3(a.H * a.B), SPLIT ? 512 : 1024, lds, st>>>(a);
    return ivlm_launch_status();
}

static int g_attn_pp = -1;  // -1 automatic; 0 / 1 force the 4-wave / 8-wave ping-pong kernel (benchmark hook)
void attn_set_pingpong(int mode) { g_attn_pp = mode; }

// SPLIT kernels (dynamic LDS: K and V tiles as hi + lo planes, double-buffered, + the rel_h block of REL 2)
template <int DQK, int DV, bool CAUSAL, int REL>
int launch_split_k(const AttnArgs& a, hipStream_t st) {
    constexpr int KS = DQK / 32, DT = DV / 16;
    constexpr size_t lds = (size_t)(2 * 2 * KS * (kKV * 32 + 32) + 2 * 2 * DT * (kKV * 16 + 16)) * 2 +
                           (REL == 2 ? (size_t)kQPerBlock * kKV * 4 : (REL == 4 ? (size_t)8 * 16 * 65 * 4 : 0));
    auto kfn = attn_kernel<DQK, DV, CAUSAL, REL, false, true>;
    static ivlm_dev_mask_t attr_set{0};  // per instantiation
    if (ivlm_dev_pending(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ivlm_dev_done(attr_set);
    }
    dim3 grid((a.Sq + kQPerBlock - 1) / kQPerBlock, a.H, a.B);
    kfn<<<grid, 512, lds, st>>>(a);
    return ivlm_launch_status();
}

// the shapes of the path only (keeps the build small): CLIP (64, plain), SAM ViT (80, rel-pos), LLaMA prefill (128, causal)
template <int DQK, int DV>
int launch_split(const AttnArgs& a, hipStream_t st) {
    const bool rel = a.rel_h != nullptr;
    if constexpr (DV == 128) {
        if (a.causal && !rel) return launch_split_k<DQK, DV, true, 0>(a, st);
    } else if constexpr (DV == 64) {
        if (!a.causal && !rel) return launch_split_k<DQK, DV, false, 0>(a, st);
    } else if constexpr (DV == 80) {
        if (!a.causal && rel && a.prescale_q) {
            if (!a.rel_w) return launch_split_k<DQK, DV, false, 4>(a, st);  // rel-pos terms from the table, in the kernel
            if (a.rel_kw == kKV && a.Sk == a.rel_kh * a.rel_kw) return launch_split_k<DQK, DV, false, 2>(a, st);
            if (a.rel_kh + a.rel_kw <= 32) return launch_split_k<DQK, DV, false, 1>(a, st);
            return launch_split_k<DQK, DV, false, 3>(a, st);
        }
    }
    return IVLM_ERR_UNSUPPORTED;
}

// fp16 operands (a.f16): the shapes of the path, like the split kernels
template <int DQK, int DV>
int launch_f16(const AttnArgs& a, hipStream_t st) {
    const bool rel = a.rel_h != nullptr;
    dim3 grid((a.Sq + kQPerBlock - 1) / kQPerBlock, a.H, a.B);
    if constexpr (DV == 80) {  // SAM's 64 x 64 grid in table mode (REL 5): the terms computed in the kernel, q_lo (level 1) in them
        if (!a.causal && rel && a.prescale_q && !a.rel_w && a.rel_kh == kKV && a.rel_kw == kKV && a.Sq == kKV * kKV && a.Sk == a.Sq) {
            if (a.q_lo && a.q_lo_level >= 2) return IVLM_ERR_UNSUPPORTED;
            const dim3 g1((a.Sq / kQPerBlock) * a.H * a.B);  // (REL 5: 1-D, XCD-aware block map in the kernel)
            if (a.q_lo) attn_kernel<DQK, DV, false, 5, false, false, true, 1><<<g1, 256, 0, st>>>(a);
            else attn_kernel<DQK, DV, false, 5, false, false, true, 0><<<g1, 256, 0, st>>>(a);
            return ivlm_launch_status();
        }
    }
    if (a.q_lo && a.q_lo_level < 2 && a.rel_w) {
        // level 1 with the terms as arrays: the lo half of q is already in them (ivlm_relpos_* on hi + lo) - the plain fp16 kernel
    } else if (a.q_lo) {  // level 2 (QLO): SAM's global grid with the rel-pos terms as arrays (windows take the whole-window kernel)
        if constexpr (DV == 80) {
            if (!a.causal && rel && a.prescale_q && a.rel_w && a.rel_kw == kKV && a.Sk == a.rel_kh * a.rel_kw) {
                attn_kernel<DQK, DV, false, 2, false, false, true, 2><<<grid, 256, 0, st>>>(a);
                return ivlm_launch_status();
            }
        }
        return IVLM_ERR_UNSUPPORTED;
    }
    if (!rel) {  // plain / causal attention at every head dim (CLIP and LLaMA towers of any configuration)
        if (a.causal) attn_kernel<DQK, DV, true, 0, false, false, true><<<grid, 256, 0, st>>>(a);
        else attn_kernel<DQK, DV, false, 0, false, false, true><<<grid, 256, 0, st>>>(a);
        return ivlm_launch_status();
    }
    if constexpr (DV == 80) {
        if (!a.causal && rel && a.prescale_q) {
            if (!a.rel_w) attn_kernel<DQK, DV, false, 4, false, false, true><<<grid, 256, 0, st>>>(a);
            else if (a.rel_kw == kKV && a.Sk == a.rel_kh * a.rel_kw) attn_kernel<DQK, DV, false, 2, false, false, true><<<grid, 256, 0, st>>>(a);
            else if (a.rel_kh + a.rel_kw <= 32) attn_kernel<DQK, DV, false, 1, false, false, true><<<grid, 256, 0, st>>>(a);
            else attn_kernel<DQK, DV, false, 3, false, false, true><<<grid, 256, 0, st>>>(a);
            return ivlm_launch_status();
        }
    }
    return IVLM_ERR_UNSUPPORTED;
}

template <int DQK, int DV, bool PP>
int launch_dp(const AttnArgs& a, hipStream_t st) {
    constexpr int NT = PP ? 512 : 256, QB = PP ? 2 * kQPerBlock : kQPerBlock;
    dim3 grid((a.Sq + QB - 1) / QB, a.H, a.B);
    const bool rel = a.rel_h != nullptr;
    if (a.causal) {
        if (rel) return IVLM_ERR_UNSUPPORTED;
        attn_kernel<DQK, DV, true, 0, PP><<<grid, NT, 0, st>>>(a);
    } else if (rel) {
        if (DV != 80) return IVLM_ERR_UNSUPPORTED;  // only SAM's ViT uses rel-pos; keeps the build small
        if (!a.prescale_q) return IVLM_ERR_UNSUPPORTED;
        if (!a.rel_w && a.rel_kh == kKV && a.rel_kw == kKV && a.Sq == kKV * kKV && a.Sk == a.Sq) {  // the 64 x 64 grid in table mode
            if constexpr (DV == 80 && !PP) attn_kernel<DQK, DV, false, 5, false><<<dim3((a.Sq / kQPerBlock) * a.H * a.B), 256, 0, st>>>(a);
            else return IVLM_ERR_UNSUPPORTED;
        } else if (!a.rel_w)  // rel_h is the bf16 table [64, D]: the rel-pos terms are computed in the kernel (windows: 2 * side <= 32)
            attn_kernel<DQK, DV, false, DV == 80 ? 4 : 0, false><<<dim3((a.Sq + kQPerBlock - 1) / kQPerBlock, a.H, a.B), 256, 0, st>>>(a);
        else if (a.rel_kw == kKV && a.Sk == a.rel_kh * a.rel_kw)
            attn_kernel<DQK, DV, false, DV == 80 ? 2 : 0, PP><<<grid, NT, 0, st>>>(a);
        else if (a.rel_kh + a.rel_kw <= 32)
            attn_kernel<DQK, DV, false, DV == 80 ? 1 : 0, PP><<<grid, NT, 0, st>>>(a);
        else
            attn_kernel<DQK, DV, false, DV == 80 ? 3 : 0, false><<<dim3((a.Sq + kQPerBlock - 1) / kQPerBlock, a.H, a.B), 256, 0, st>>>(a);
    } else {
        attn_kernel<DQK, DV, false, 0, PP><<<grid, NT, 0, st>>>(a);
    }
    return ivlm_launch_status();
}

template <int DQK, int DV>
int launch_d(const AttnArgs& a, hipStream_t st) {
    // the 8-wave ping-pong block (256 queries) is opt-in: measured 5-15 % SLOWER than two independent 4-wave blocks per CU on
    // every shape of the path (SAM global 634 vs 604 us, windows 151 vs 132 us) - the loop is bound by the issue latency of
    // the dependent softmax chain, not by the two waves of a SIMD contending for the same unit
    if constexpr (DV == 80) {  // SAM's windows: the whole-window kernel (default precision: table mode; SPLIT: array mode)
        const bool win = g_win_v2 && a.rel_h && a.Sq <= 208 && a.Sq == a.Sk && !a.causal && a.prescale_q && a.H <= 65535 &&
                         a.B <= 65535 && a.rel_kh == a.rel_kw && 2 * a.rel_kh <= 32 && a.Sq == a.rel_kh * a.rel_kw;
        if (win && !a.q_lo && !a.rel_w) return a.f16 ? launch_win<false, true>(a, st) : launch_win<false>(a, st);
        if (win && a.q_lo && !a.rel_w && a.f16) return a.q_lo_level >= 2 ? launch_win<false, true, 2>(a, st) : launch_win<false, true, 1>(a, st);
        if (win && a.q_lo && a.rel_w && !a.f16) return launch_win<true>(a, st);
    }
    if (a.f16) return launch_f16<DQK, DV>(a, st);
    if (a.q_lo) return launch_split<DQK, DV>(a, st);
    const bool pp = g_attn_pp > 0;
    return pp ? launch_dp<DQK, DV, true>(a, st) : launch_dp<DQK, DV, false>(a, st);
}

// rel_h[bh,q,kh] = q_vec . rel_pos_h[qh - kh + KH - 1],  rel_w[bh,q,kw] = q_vec . rel_pos_w[qw - kw + KW - 1]
// (image_encoder.py:321-392 with q_size == k_size).  One thread per output element.
__global__ __launch_bounds__(256) void relpos_kernel(const bf16_t* __restrict__ q, int64_t q_bs, int64_t q_hs,
                                                     int64_t q_rs, const bf16_t* __restrict__ tab_h,
                                                     const bf16_t* __restrict__ tab_w, int B, int H, int SH, int SW,
                                                     int D, float* __restrict__ rel_h, float* __restrict__ rel_w) {
    const int S = SH * SW;
    const int per_q = SH + SW;
    const int64_t total = (int64_t)B * H * S * per_q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % per_q);
        const int64_t bq = i / per_q;
        const int qi = (int)(bq % S);
        const int64_t bh = bq / S;
        const int b = (int)(bh / H), h = (int)(bh % H);
        const int qh = qi / SW, qw = qi - qh * SW;
        const bf16_t* qv = q + b * q_bs + h * q_hs + (int64_t)qi * q_rs;
        const bf16_t* tv;
        if (j < SH) tv = tab_h + (int64_t)(qh - j + SH - 1) * D;
        else tv = tab_w + (int64_t)(qw - (j - SH) + SW - 1) * D;
        float acc = 0.0f;
        for (int c = 0; c < D; c += 8) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(qv + c);
            const uint4 t4 = *reinterpret_cast<const uint4*>(tv + c);
            const bf16_t* ae = reinterpret_cast<const bf16_t*>(&a4);
            const bf16_t* te = reinterpret_cast<const bf16_t*>(&t4);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += bf16_to_f32(ae[e]) * bf16_to_f32(te[e]);
        }
        // the reference materialises rel_h/rel_w in the model dtype (bf16): round like it does
        acc = bf16_to_f32(f32_to_bf16(acc));
        if (j < SH) rel_h[bq * SH + j] = acc;
        else rel_w[bq * SW + (j - SH)] = acc;
    }
}

// Same operator, restructured for throughput (the thread-per-output kernel above spent 5.6 ms per image): one thread per
// QUERY keeps its q vector packed in registers and walks its SH + SW outputs; both rel-pos tables sit in LDS (row stride
// D + 8 elements: the 16 lanes of a ds_read_b128 pass that read 16 consecutive rows hit distinct banks; lanes that share a
// row broadcast); 8 x 8 bf16 products per v_dot2c_f32_bf16 quad; outputs leave as 16-byte stores, 64 contiguous bytes per
// lane per 16 outputs (whole sectors).  NCH = D / 8.
typedef __attribute__((ext_vector_type(2))) __bf16 rp_bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int rp_u32x4_t;

// SPLIT ("parity" precision): q = hi + lo planes (q_lo, same strides), fp32 results NOT rounded to bf16 (the fp32 reference
// keeps them in fp32; only a bf16 model materialises them in bf16).
template <int NCH, bool SPLIT = false, bool ROUND = !SPLIT>
__global__ __launch_bounds__(256) void relpos_rows_kernel(const bf16_t* __restrict__ q, int64_t q_bs, int64_t q_hs,
                                                          int64_t q_rs, const bf16_t* __restrict__ tab_h,
                                                          const bf16_t* __restrict__ tab_w, int H, int SH, int SW,
                                                          float* __restrict__ rel_h, float* __restrict__ rel_w,
                                                          const bf16_t* __restrict__ q_lo = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rp_smem[];
    constexpr int D = NCH * 8, kStride = D + 8;  // elements
    rp_u32x4_t* th = reinterpret_cast<rp_u32x4_t*>(rp_smem);                        // [2SH-1][kStride/8] chunks
    rp_u32x4_t* tw = th + (2 * SH - 1) * (kStride / 8);
    const int nh = (2 * SH - 1) * NCH, nw = (2 * SW - 1) * NCH;
    for (int i = threadIdx.x; i < nh + nw; i += 256) {
        const bool isw = i >= nh;
        const int k = isw ? i - nh : i;
        const int row = k / NCH, c = k - row * NCH;
        const rp_u32x4_t v = *reinterpret_cast<const rp_u32x4_t*>((isw ? tab_w : tab_h) + (int64_t)row * D + c * 8);
        (isw ? tw : th)[row * (kStride / 8) + c] = v;
    }
    __syncthreads();
    const int S = SH * SW;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= S) return;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qh = qi / SW, qw = qi - qh * SW;
    const bf16_t* qv = q + b * q_bs + h * q_hs + (int64_t)qi * q_rs;
    rp_u32x4_t qr[NCH], ql[SPLIT ? NCH : 1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) qr[c] = *reinterpret_cast<const rp_u32x4_t*>(qv + c * 8);
    if (SPLIT) {
        const bf16_t* qlv = q_lo + b * q_bs + h * q_hs + (int64_t)qi * q_rs;
#pragma unroll
        for (int c = 0; c < NCH; ++c) ql[SPLIT ? c : 0] = *reinterpret_cast<const rp_u32x4_t*>(qlv + c * 8);
    }
    const int64_t bq = ((int64_t)b * H + h) * S + qi;
    auto run = [&](const rp_u32x4_t* tab, int base_row, int n, float* out) {
        // out[j] = q . tab[base_row - j], j = 0..n-1, in groups of 4 (16-byte stores; n % 4 tail scalar)
        int j = 0;
        for (; j + 4 <= n; j += 4) {
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const rp_u32x4_t* row = tab + (base_row - (j + u)) * (kStride / 8);
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const rp_u32x4_t t4 = row[c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t a = qr[c][e], w = t4[e];
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2_t, a),
                                                               __builtin_bit_cast(rp_bf16x2_t, w), acc, false);
                        if (SPLIT)
                            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2_t, (uint32_t)ql[SPLIT ? c : 0][e]),
                                                                   __builtin_bit_cast(rp_bf16x2_t, w), acc, false);
                    }
                }
                r[u] = ROUND ? bf16_to_f32(f32_to_bf16(acc)) : acc;  // the (bf16) reference materialises rel_h/rel_w in bf16
            }
            if ((((uintptr_t)(out + j)) & 15) == 0) {
                *reinterpret_cast<float4*>(out + j) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) out[j + u] = r[u];
            }
        }
        for (; j < n; ++j) {
            const rp_u32x4_t* row = tab + (base_row - j) * (kStride / 8);
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const rp_u32x4_t t4 = row[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t a = qr[c][e], w = t4[e];
                    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2_t, a),
                                                           __builtin_bit_cast(rp_bf16x2_t, w), acc, false);
                    if (SPLIT)
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2_t, (uint32_t)ql[SPLIT ? c : 0][e]),
                                                               __builtin_bit_cast(rp_bf16x2_t, w), acc, false);
                }
            }
            out[j] = ROUND ? bf16_to_f32(f32_to_bf16(acc)) : acc;
        }
    };
    run(th, qh + SH - 1, SH, rel_h + bq * SH);
    run(tw, qw + SW - 1, SW, rel_w + bq * SW);
}

// rel-pos operands from ONE batched MFMA GEMM: G[h][b*S + s][:] = q[b,h,s,:] . [rel_pos_h ; rel_pos_w]^T (bf16, the model
// dtype the reference's einsum produces), then the Toeplitz shift is a pure gather:
//   rel_h[bh,s,kh] = G[..][qh - kh + SH - 1],   rel_w[bh,s,kw] = G[..][(2 SH - 1) + qw - kw + SW - 1].
// The dot products (2 x (2S-1) x D MACs per query: 5.6 GFLOP per global block) leave the VALU / LDS for the matrix cores;
// this kernel only moves data: one thread per 4 consecutive outputs (16-byte stores).
template <int GK>  // element type of G: 0 bf16, 1 fp16, 2 fp32
__global__ __launch_bounds__(256) void relpos_gather_kernel(const bf16_t* __restrict__ G, int64_t g_hs /*head stride*/, int npad,
                                                            int H, int SH, int SW, float* __restrict__ rel_h,
                                                            float* __restrict__ rel_w) {
    // threadIdx.x = output slot of a query (kh for x < SH, then kw), threadIdx.y = query inside the block: the loads of a query
    // are two reversed contiguous runs of its G row, the stores two contiguous runs - no integer division per output
    const int S = SH * SW;
    const int s_ = blockIdx.x * blockDim.y + threadIdx.y;
    const int j = threadIdx.x;
    if (s_ >= S || j >= SH + SW) return;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qh = s_ / SW, qw = s_ - qh * SW;
    const bf16_t* row = G + h * g_hs + ((int64_t)b * S + s_) * npad;
    const int64_t bq = ((int64_t)b * H + h) * S + s_;
    const int col = j < SH ? qh - j + SH - 1 : (2 * SH - 1) + qw - (j - SH) + SW - 1;
    float g;
    if constexpr (GK == 2) g = (reinterpret_cast<const float*>(G) + h * g_hs + ((int64_t)b * S + s_) * npad)[col];
    else g = GK == 1 ? (float)__builtin_bit_cast(_Float16, row[col]) : bf16_to_f32(row[col]);
    if (j < SH) rel_h[bq * SH + j] = g;
    else rel_w[bq * SW + (j - SH)] = g;
}

}  // namespace

int relpos_gather(const bf16_t* G, int64_t g_hs, int npad, int B, int H, int SH, int SW, float* rel_h, float* rel_w,
                  hipStream_t st, int f16 = 0 /* G: 0 bf16, 1 fp16, 2 fp32 */) {
    if (!G || !rel_h || !rel_w || B <= 0 || H <= 0 || SH <= 0 || SW <= 0 || npad < 2 * SH - 1 + 2 * SW - 1) return IVLM_ERR_INVALID_ARG;
    if (SH + SW > 256 || H > 65535 || B > 65535) return IVLM_ERR_UNSUPPORTED;
    int bx = 32;
    while (bx < SH + SW) bx <<= 1;
    const int by = 256 / bx;
    const dim3 grid((SH * SW + by - 1) / by, H, B), blk(bx, by);
    if (f16 == 2) relpos_gather_kernel<2><<<grid, blk, 0, st>>>(G, g_hs, npad, H, SH, SW, rel_h, rel_w);
    else if (f16) relpos_gather_kernel<1><<<grid, blk, 0, st>>>(G, g_hs, npad, H, SH, SW, rel_h, rel_w);
    else relpos_gather_kernel<0><<<grid, blk, 0, st>>>(G, g_hs, npad, H, SH, SW, rel_h, rel_w);
    return ivlm_launch_status();
}

int attention_bf16(const AttnArgs& a_in, hipStream_t st) {
    AttnArgs a = a_in;
    a.xcd_map = g_xcd_map;
    if (!a.q || !a.k || !a.v || !a.o || a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Sk <= 0) return IVLM_ERR_INVALID_ARG;
    if (a.H > 65535 || a.B > 65535 || a.kv_batch_div <= 0) return IVLM_ERR_INVALID_ARG;
    if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.q_hs | a.k_hs | a.v_hs | a.q_bs | a.k_bs | a.v_bs) & 7)
        return IVLM_ERR_UNSUPPORTED;  // 16-byte row chunks
    if ((a.o_rs | a.o_hs | a.o_bs) & 3) return IVLM_ERR_UNSUPPORTED;
    if (a.rel_h && (a.rel_kh <= 0 || a.rel_kw <= 0)) return IVLM_ERR_INVALID_ARG;
    if (a.rel_h && !a.rel_w) {  // table mode: square windows whose 2 * side terms fit one MFMA k-step, or the 64 x 64 grid (REL 5)
        if (a.rel_kh != a.rel_kw || (2 * a.rel_kh > 32 && a.rel_kh != kKV) || a.Sq != a.rel_kh * a.rel_kw || a.Sk != a.Sq || a.D != 80 ||
            (reinterpret_cast<uintptr_t>(a.rel_h) & 15))
            return IVLM_ERR_UNSUPPORTED;
        if (a.rel_kh == kKV && a.q_lo && !a.f16) return IVLM_ERR_UNSUPPORTED;  // (the split kernels take the terms as arrays)
    }
    if (a.f16) {  // fp16 operands: at most a lo plane of q ("exact q")
        if (a.k_lo || a.v_lo || a.o_lo) return IVLM_ERR_INVALID_ARG;
    } else if ((a.q_lo || a.k_lo || a.v_lo || a.o_lo) && !(a.q_lo && a.k_lo && a.v_lo && a.o_lo)) {
        return IVLM_ERR_INVALID_ARG;
    }
    switch (a.D) {
        case 16: return launch_d<32, 16>(a, st);
        case 32: return launch_d<32, 32>(a, st);
        case 64: return launch_d<64, 64>(a, st);
        case 80: return launch_d<96, 80>(a, st);
        case 128: return launch_d<128, 128>(a, st);
        default: return IVLM_ERR_UNSUPPORTED;
    }
}

int relpos_bias(const bf16_t* q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const bf16_t* tab_h, const bf16_t* tab_w,
                int B, int H, int SH, int SW, int D, float* rel_h, float* rel_w, hipStream_t st, const bf16_t* q_lo) {
    if (!q || !tab_h || !tab_w || !rel_h || !rel_w || (D & 7)) return IVLM_ERR_INVALID_ARG;
    if (q_lo == q) {  // (q_lo == q: no lo plane - the bf16 q of the default path, but results kept in fp32, not rounded to bf16)
        if (D != 80 || SH > 128 || SW > 128 || H > 65535 || B > 65535 || ((q_bs | q_hs | q_rs) & 7) ||
            (reinterpret_cast<uintptr_t>(q) & 15))
            return IVLM_ERR_UNSUPPORTED;
        const size_t lds = (size_t)(2 * SH - 1 + 2 * SW - 1) * (D + 8) * 2;
        relpos_rows_kernel<10, false, false><<<dim3((SH * SW + 255) / 256, H, B), 256, lds, st>>>(q, q_bs, q_hs, q_rs, tab_h, tab_w,
                                                                                                H, SH, SW, rel_h, rel_w, nullptr);
        return ivlm_launch_status();
    }
    if (q_lo) {  // "parity" precision: q as hi + lo planes, unrounded fp32 results (SAM head dim only)
        if (D != 80 || SH > 128 || SW > 128 || H > 65535 || B > 65535 || ((q_bs | q_hs | q_rs) & 7) ||
            ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(q_lo)) & 15))
            return IVLM_ERR_UNSUPPORTED;
        const size_t lds = (size_t)(2 * SH - 1 + 2 * SW - 1) * (D + 8) * 2;
        relpos_rows_kernel<10, true><<<dim3((SH * SW + 255) / 256, H, B), 256, lds, st>>>(q, q_bs, q_hs, q_rs, tab_h, tab_w, H, SH,
                                                                                         SW, rel_h, rel_w, q_lo);
        return ivlm_launch_status();
    }
    if ((D == 80 || D == 64) && SH <= 128 && SW <= 128 && H <= 65535 && B <= 65535 &&
        ((q_bs | q_hs | q_rs) & 7) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
        const size_t lds = (size_t)(2 * SH - 1 + 2 * SW - 1) * (D + 8) * 2;
        dim3 grid((SH * SW + 255) / 256, H, B);
        if (D == 80) relpos_rows_kernel<10><<<grid, 256, lds, st>>>(q, q_bs, q_hs, q_rs, tab_h, tab_w, H, SH, SW, rel_h, rel_w);
        else relpos_rows_kernel<8><<<grid, 256, lds, st>>>(q, q_bs, q_hs, q_rs, tab_h, tab_w, H, SH, SW, rel_h, rel_w);
        return ivlm_launch_status();
    }
    const int64_t total = (int64_t)B * H * SH * SW * (SH + SW);
    const int grid = (int)((total + 255) / 256 < 65535 * 4 ? (total + 255) / 256 : 65535 * 4);
    relpos_kernel<<<grid, 256, 0, st>>>(q, q_bs, q_hs, q_rs, tab_h, tab_w, B, H, SH, SW, D, rel_h, rel_w);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" {

int ivlm_relpos_gather(const void* G, int64_t g_head_stride, int npad, int B, int H, int SH, int SW, float* rel_h, float* rel_w,
                       ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::relpos_gather(static_cast<const bf16_t*>(G), g_head_stride, npad, B, H, SH, SW, rel_h, rel_w,
                               ivlm_stream(stream));
}

int ivlm_relpos_gather_f32(const void* G, int64_t g_head_stride, int npad, int B, int H, int SH, int SW, float* rel_h, float* rel_w,
                           ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::relpos_gather(static_cast<const bf16_t*>(G), g_head_stride, npad, B, H, SH, SW, rel_h, rel_w,
                               ivlm_stream(stream), 2);
}

// fp16 q (optionally hi + lo halves): the terms through ONE batched GEMM over the heads against the fp16 [rel_pos_h ; rel_pos_w] table
// (fp32 G: the terms add to the scores and are not rounded), then the Toeplitz gather
int ivlm_relpos_bias_f16(const void* q, const void* q_lo, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* cat16, int npad, int B,
                         int H, int SH, int SW, int D, float* G_ws, size_t g_bytes, float* rel_h, float* rel_w, ivlm_stream_t stream) {
    ivlm_enter();
    const int S = SH * SW;
    if (!q || !cat16 || !G_ws || !rel_h || !rel_w || B <= 0 || H <= 0 || SH <= 0 || SW <= 0 || D <= 0 || (D & 7)) return IVLM_ERR_INVALID_ARG;
    if (npad < 2 * SH - 1 + 2 * SW - 1 || (npad & 3)) return IVLM_ERR_INVALID_ARG;
    if (q_bs != (int64_t)S * q_rs || ((q_rs | q_hs) & 7) || (reinterpret_cast<uintptr_t>(q) & 15)) return IVLM_ERR_UNSUPPORTED;  // rows of all (b, s) uniformly strided
    const int64_t M = (int64_t)B * S;
    if (M > 0x7fffffff || g_bytes < (size_t)H * M * npad * 4) return IVLM_ERR_WORKSPACE;
    ivlm::GemmArgs g;
    g.A = static_cast<const bf16_t*>(q); g.lda = q_rs;
    g.W = static_cast<const bf16_t*>(cat16); g.ldw = D;
    g.C = G_ws; g.ldc = npad;
    g.M = (int)M; g.N = npad; g.K = D;
    g.batch = H; g.strideA = q_hs; g.strideW = 0; g.strideC = M * npad;
    g.out_f32 = 1;
    g.f16 = 1;
    if (q_lo) {
        const int64_t off = static_cast<const bf16_t*>(q_lo) - static_cast<const bf16_t*>(q);
        if (off < D || (off & 7)) return IVLM_ERR_INVALID_ARG;  // the lo plane lies behind the hi plane of the same rows
        g.a_split = 1;
        g.a_lo = off;
    }
    const int rc = ivlm::gemm_bf16(g, ivlm_stream(stream));
    if (rc != IVLM_OK) return rc;
    return ivlm::relpos_gather(reinterpret_cast<const bf16_t*>(G_ws), M * npad, npad, B, H, SH, SW, rel_h, rel_w, ivlm_stream(stream), 2);
}

int ivlm_attention_xcd_map(int on) {  // benchmark hook: XCD-aware block map of the 64 x 64 grid's table-mode kernel (default 1)
    const int prev = ivlm::g_xcd_map;
    ivlm::g_xcd_map = on ? 1 : 0;
    return prev;
}

int ivlm_attention_window_kernel(int v2) {  // benchmark/test hook: 1 (default) whole-window kernel, 0 generic flash kernel
    ivlm::g_win_v2 = v2;
    return 0;
}

int ivlm_attention_pingpong(int mode) {  // benchmark/test hook: -1 automatic, 0 four-wave kernel, 1 eight-wave ping-pong
    ivlm::attn_set_pingpong(mode);
    return 0;
}

static int attention_16(const void* q, const void* k, const void* v, void* o, const int64_t* strides /*[12]*/, int B,
                        int H, int Sq, int Sk, int D, float scale, int causal, int q_pos0, const float* rel_h,
                        const float* rel_w, int rel_kh, int rel_kw, int kv_batch_div, int prescale_q, int f16, ivlm_stream_t stream,
                        const void* q_lo = nullptr, int q_lo_level = 0) {
    ivlm_enter();
    if (!strides) return IVLM_ERR_INVALID_ARG;
    ivlm::AttnArgs a;
    a.f16 = f16;
    a.q_lo = static_cast<const bf16_t*>(q_lo);
    a.q_lo_level = q_lo_level;
    if (reinterpret_cast<uintptr_t>(q_lo) & 15) return IVLM_ERR_INVALID_ARG;
    a.q = static_cast<const bf16_t*>(q);
    a.k = static_cast<const bf16_t*>(k);
    a.v = static_cast<const bf16_t*>(v);
    a.o = static_cast<bf16_t*>(o);
    a.q_bs = strides[0]; a.q_hs = strides[1]; a.q_rs = strides[2];
    a.k_bs = strides[3]; a.k_hs = strides[4]; a.k_rs = strides[5];
    a.v_bs = strides[6]; a.v_hs = strides[7]; a.v_rs = strides[8];
    a.o_bs = strides[9]; a.o_hs = strides[10]; a.o_rs = strides[11];
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.D = D;
    a.scale = scale;
    a.causal = causal;
    a.q_pos0 = q_pos0;
    a.rel_h = rel_h;
    a.rel_w = rel_w;
    a.rel_kh = rel_kh;
    a.rel_kw = rel_kw;
    a.kv_batch_div = kv_batch_div < 1 ? 1 : kv_batch_div;
    a.prescale_q = prescale_q;
    return ivlm::attention_bf16(a, ivlm_stream(stream));
}

int ivlm_attention_bf16(const void* q, const void* k, const void* v, void* o, const int64_t* strides /*[12]*/, int B,
                        int H, int Sq, int Sk, int D, float scale, int causal, int q_pos0, const float* rel_h,
                        const float* rel_w, int rel_kh, int rel_kw, int kv_batch_div, int prescale_q, ivlm_stream_t stream) {
    return attention_16(q, k, v, o, strides, B, H, Sq, Sk, D, scale, causal, q_pos0, rel_h, rel_w, rel_kh, rel_kw, kv_batch_div,
                        prescale_q, 0, stream);
}

// the same operator on IEEE fp16 q / k / v / o (and, in table mode, an fp16 rel-pos table): see ivlm_hip.h
int ivlm_attention_f16(const void* q, const void* k, const void* v, void* o, const int64_t* strides /*[12]*/, int B,
                       int H, int Sq, int Sk, int D, float scale, int causal, int q_pos0, const float* rel_h,
                       const float* rel_w, int rel_kh, int rel_kw, int kv_batch_div, int prescale_q, ivlm_stream_t stream) {
    return attention_16(q, k, v, o, strides, B, H, Sq, Sk, D, scale, causal, q_pos0, rel_h, rel_w, rel_kh, rel_kw, kv_batch_div,
                        prescale_q, 1, stream);
}

// ... with q as hi + lo IEEE halves (q_lo: the strides of q): the "exact q" attention of the fp16 mode (SAM shapes), see ivlm_hip.h
int ivlm_attention_f16_qsplit(const void* q, const void* q_lo, const void* k, const void* v, void* o, const int64_t* strides /*[12]*/,
                              int B, int H, int Sq, int Sk, int D, float scale, const float* rel_h, const float* rel_w, int rel_kh,
                              int rel_kw, int level, ivlm_stream_t stream) {
    if (!q_lo || (level != 1 && level != 2)) return IVLM_ERR_INVALID_ARG;
    return attention_16(q, k, v, o, strides, B, H, Sq, Sk, D, scale, 0, 0, rel_h, rel_w, rel_kh, rel_kw, 1, 1, 1, stream, q_lo, level);
}

int ivlm_attention_bf16_split(const void* q, const void* q_lo, const void* k, const void* k_lo, const void* v, const void* v_lo,
                              void* o, void* o_lo, const int64_t* strides /*[12]*/, int B, int H, int Sq, int Sk, int D, float scale,
                              int causal, int q_pos0, const float* rel_h, const float* rel_w, int rel_kh, int rel_kw,
                              int kv_batch_div, int prescale_q, ivlm_stream_t stream) {
    ivlm_enter();
    if (!strides || !q_lo || !k_lo || !v_lo || !o_lo) return IVLM_ERR_INVALID_ARG;
    ivlm::AttnArgs a;
    a.q = static_cast<const bf16_t*>(q);
    a.k = static_cast<const bf16_t*>(k);
    a.v = static_cast<const bf16_t*>(v);
    a.o = static_cast<bf16_t*>(o);
    a.q_lo = static_cast<const bf16_t*>(q_lo);
    a.k_lo = static_cast<const bf16_t*>(k_lo);
    a.v_lo = static_cast<const bf16_t*>(v_lo);
    a.o_lo = static_cast<bf16_t*>(o_lo);
    a.q_bs = strides[0]; a.q_hs = strides[1]; a.q_rs = strides[2];
    a.k_bs = strides[3]; a.k_hs = strides[4]; a.k_rs = strides[5];
    a.v_bs = strides[6]; a.v_hs = strides[7]; a.v_rs = strides[8];
    a.o_bs = strides[9]; a.o_hs = strides[10]; a.o_rs = strides[11];
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.D = D;
    a.scale = scale;
    a.causal = causal;
    a.q_pos0 = q_pos0;
    a.rel_h = rel_h;
    a.rel_w = rel_w;
    a.rel_kh = rel_kh;
    a.rel_kw = rel_kw;
    a.kv_batch_div = kv_batch_div < 1 ? 1 : kv_batch_div;
    a.prescale_q = prescale_q;
    if ((reinterpret_cast<uintptr_t>(q_lo) | reinterpret_cast<uintptr_t>(k_lo) | reinterpret_cast<uintptr_t>(v_lo)) & 15)
        return IVLM_ERR_INVALID_ARG;
    return ivlm::attention_bf16(a, ivlm_stream(stream));
}

int ivlm_relpos_bias(const void* q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* tab_h, const void* tab_w,
                     int B, int H, int SH, int SW, int D, float* rel_h, float* rel_w, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::relpos_bias(static_cast<const bf16_t*>(q), q_bs, q_hs, q_rs, static_cast<const bf16_t*>(tab_h),
                             static_cast<const bf16_t*>(tab_w), B, H, SH, SW, D, rel_h, rel_w, ivlm_stream(stream), nullptr);
}

int ivlm_relpos_bias_split(const void* q, const void* q_lo, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void* tab_h,
                           const void* tab_w, int B, int H, int SH, int SW, int D, float* rel_h, float* rel_w,
                           ivlm_stream_t stream) {
    ivlm_enter();
    if (!q_lo) return IVLM_ERR_INVALID_ARG;
    return ivlm::relpos_bias(static_cast<const bf16_t*>(q), q_bs, q_hs, q_rs, static_cast<const bf16_t*>(tab_h),
                             static_cast<const bf16_t*>(tab_w), B, H, SH, SW, D, rel_h, rel_w, ivlm_stream(stream),
                             static_cast<const bf16_t*>(q_lo));
}

}  // extern "C"
