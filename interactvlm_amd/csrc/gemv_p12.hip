// Batch-1 decode GEMV over LOSSLESSLY packed bf16 weights ("bf12": 12 bits per weight) for gfx950.
//
// The decode step of the greedy search under InteractVLM.evaluate (model/InteractVLM.py:524-531) is pure weight streaming: 13.5 GB of
// bf16 weights per generated token for LLaMA-7B, 60 % of an image's time.  The 8 exponent bits of a weight matrix carry ~2.7 bits
// of information: inside one row nearly every weight lies within 15 binades of the row's largest (a Gaussian row: all but 1e-4).
// So a row is stored as
//     P  [K]   bytes   sign << 7 | mantissa (7 bits)
//     E  [K/2] bytes   two 4-bit codes: exponent field - ebase[row] in 1 .. 15; 0 = the weight is zero or lies outside the window
//     ebase    int32   per row (row maximum of the exponent field - 15, clamped at 0)
//     patches  CSR     (column, bf16 value) of the nonzero weights outside the window (subnormals, the far tail)
// = 1.5 bytes per weight instead of 2, and EVERY weight is reconstructed bit for bit (ivlm_unpack_bf12 is the proof: tests).
// The kernel never rebuilds the bf16 value: a lane turns (P byte, code) into the fp32 number 1.m x 2^(code - 127) with four integer
// operations (byte permute, mask, bit-field extract, shift-add; code 0 gives exactly 0.0), multiplies it with x * 2^64 (x is staged
// once per block in LDS, pre-scaled by an exact power of two so that the tiny products stay normal fp32 numbers) and the row sum is
// scaled back by 2^(ebase - 100) - exact power-of-two scalings, so the arithmetic is that of gemv1_kernel (bf16 weight x fp32
// activation products, exact; fp32 accumulation) up to the summation order.  Same shape as gemv1_kernel: 1024-thread blocks, one row
// per wave, non-temporal loads (16 B of P + 8 B of E per lane and step = 16 weights), fused RMSNorm prologue, SwiGLU / residual
// epilogues.
#include <algorithm>

#include "kernels.h"

namespace ivlm {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

constexpr int kWaves = 16;
// x is staged as x * 2^64: an exact power-of-two scaling, chosen in the MIDDLE of fp32's exponent range (2^100 turned any
// |x| >= 2^28 ~ 2.7e8 into inf and the row into NaN, a range gemv1_kernel's plain fp32 x does not have).  A rebuilt weight is
// 1.m x 2^(code - 127) with code 1 .. 15, i.e. 2^-126 .. 2^-111, so the products x * 2^64 * w are normal fp32 numbers for
// 2^-64 <~ |x| < 2^64 (5e-20 .. 1.8e19) and every result inside that range is independent of the scale (the hi / lo / lo2 split and
// the fp32 accumulation are scale-invariant): bit-identical to the 2^100 staging on every activation the model produces.
constexpr int kXScaleExp = 64;

struct P12 {
    const uint8_t* P;   // [N][ldp]
    const uint8_t* E;   // [N][lde]
    int64_t ldp, lde;
    const int32_t* ebase;      // [N]
    const int32_t* patch_ptr;  // [N+1]
    const int32_t* patch_col;
    const bf16_t* patch_val;
};

// one weight: [b_k, b_k, 0, 0] puts the byte's sign on bit 31 and its mantissa on bits 22 .. 16 (bits 30 .. 23 are masked away); the code
// becomes the exponent field: 1.m x 2^(code - 127); code 0 (with P = 0) is exactly 0.0.  Four integer operations + the FMA.
// (The packed fp32 FMA - two products per instruction - was measured SLOWER: 3.07 vs 2.56 ms per token; register-pair moves.)
template <int K>
__device__ __forceinline__ float w_p12(uint32_t pw, uint32_t e16) {
    const uint32_t t = __builtin_amdgcn_perm(0u, pw, (uint32_t)((K << 24) | (K << 16) | 0x0c0cu));
    const uint32_t f0 = t & 0x807f0000u;
    const uint32_t n = __builtin_amdgcn_ubfe(e16, 4 * K, 4);
    return __uint_as_float((n << 23) + f0);
}

// four weights of one 32-bit word of P against four x values; e16: their four codes in bits 0 .. 15
__device__ __forceinline__ float dot4_p12(uint32_t pw, uint32_t e16, const f32x4v_t& x, float acc) {
    acc = fmaf(w_p12<0>(pw, e16), x[0], acc);
    acc = fmaf(w_p12<1>(pw, e16), x[1], acc);
    acc = fmaf(w_p12<2>(pw, e16), x[2], acc);
    acc = fmaf(w_p12<3>(pw, e16), x[3], acc);
    return acc;
}

__device__ __forceinline__ float act1(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));
        case ACT_RELU: return x < 0.0f ? 0.0f : x;  // (torch.relu semantics: a NaN stays a NaN; fmaxf would turn it into 0)
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-x));
        default: return x;
    }
}

template <bool RMS>
__global__ __launch_bounds__(64 * kWaves, 8) void gemv1_p12_kernel(GemmArgs g, P12 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_red[kWaves];
    __shared__ float s_val[kWaves];
    constexpr int U = 4;  // 4 x (16 + 8) bytes in flight per lane: one pass over a 4096-weight row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = g.K >> 3;  // fp32 x chunks of 8 elements (two 16-byte planes)
    const int nw16 = g.K >> 4;    // 16-weight chunks per row
    f32x4v_t* xf = reinterpret_cast<f32x4v_t*>(smem);
    const int row = blockIdx.x * kWaves + wave;
    const bool live = row < g.N;
    const int rr = live ? row : g.N - 1;
    const u32x4_t* pp = reinterpret_cast<const u32x4_t*>(p.P + (int64_t)rr * p.ldp);
    const u32x2_t* ep = reinterpret_cast<const u32x2_t*>(p.E + (int64_t)rr * p.lde);
    u32x4_t w[U];
    u32x2_t e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = min(lane + 64 * u, nw16 - 1);
        w[u] = __builtin_nontemporal_load(pp + c);
        e[u] = __builtin_nontemporal_load(ep + c);
    }
    const int eb = p.ebase[rr];
    const int p0 = p.patch_ptr[rr], p1 = p.patch_ptr[rr + 1];
    // ---- stage x * 2^64 (x * gamma * 2^64) in LDS, sum(x^2) of the unscaled row ----
    const float xs = __builtin_ldexpf(1.0f, kXScaleExp);
    float ssq = 0.0f;
    for (int c = threadIdx.x; c < nchunk; c += 64 * kWaves) {
        const f32x4v_t* xp = reinterpret_cast<const f32x4v_t*>(g.A) + 2 * c;
        f32x4v_t xa = xp[0], xb = xp[1];
        if (RMS) {
            const u32x4_t gv = *(reinterpret_cast<const u32x4_t*>(g.rms_w) + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) ssq += xa[j] * xa[j] + xb[j] * xb[j];
            xa[0] *= __uint_as_float(gv[0] << 16); xa[1] *= __uint_as_float(gv[0] & 0xffff0000u);
            xa[2] *= __uint_as_float(gv[1] << 16); xa[3] *= __uint_as_float(gv[1] & 0xffff0000u);
            xb[0] *= __uint_as_float(gv[2] << 16); xb[1] *= __uint_as_float(gv[2] & 0xffff0000u);
            xb[2] *= __uint_as_float(gv[3] << 16); xb[3] *= __uint_as_float(gv[3] & 0xffff0000u);
        }
        xf[c] = xa * xs;
        xf[nchunk + c] = xb * xs;
    }
    if (RMS) {
        ssq = wave_sum(ssq);
        if (lane == 0) s_red[wave] = ssq;
    }
    __syncthreads();
    float acc = 0.0f;
    for (int c = lane; c < nw16; c += 64 * U) {
        if (c != lane) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c + 64 * u < nw16) {
                    w[u] = __builtin_nontemporal_load(pp + c + 64 * u);
                    e[u] = __builtin_nontemporal_load(ep + c + 64 * u);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + 64 * u;
            if (cc < nw16) {  // weights 16 cc .. 16 cc + 15 against x chunks 2 cc (planes a | b) and 2 cc + 1
                acc = dot4_p12(w[u][0], e[u][0], xf[2 * cc], acc);
                acc = dot4_p12(w[u][1], e[u][0] >> 16, xf[nchunk + 2 * cc], acc);
                acc = dot4_p12(w[u][2], e[u][1], xf[2 * cc + 1], acc);
                acc = dot4_p12(w[u][3], e[u][1] >> 16, xf[nchunk + 2 * cc + 1], acc);
            }
        }
    }
    // the nonzero weights outside the row's exponent window (exact bf16 values; usually none, on average < 1 per row)
    float pacc = 0.0f;
    for (int i = p0 + lane; i < p1; i += 64) {
        const int col = p.patch_col[i];
        const int ch = col >> 3, wi = col & 7;
        const float xv = reinterpret_cast<const float*>(xf)[((wi < 4 ? ch : nchunk + ch) << 2) + (wi & 3)];
        pacc = fmaf(bf16_to_f32(p.patch_val[i]), xv, pacc);
    }
    acc = __builtin_ldexpf(wave_sum(acc), eb - kXScaleExp) + __builtin_ldexpf(wave_sum(pacc), -kXScaleExp);
    if (RMS) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < kWaves; ++i) q += s_red[i];
        acc *= rsqrtf(q / (float)g.K + g.rms_eps);
    }
    float v = acc + ((g.bias && live) ? bf16_to_f32(g.bias[row]) : 0.0f);
    if (g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved: the even wave finishes the pair
        if (lane == 0) s_val[wave] = v;
        __syncthreads();
        if (lane != 0 || (wave & 1) || !live) return;
        const float o = (v / (1.0f + __expf(-v))) * s_val[wave + 1];
        const int64_t idx = row >> 1;
        if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
        else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
        return;
    }
    if (lane != 0 || !live) return;
    v = act1(v, g.act);
    if (g.residual) v += g.res_f32 ? reinterpret_cast<const float*>(g.residual)[row] : bf16_to_f32(g.residual[row]);
    if (g.out_f32) static_cast<float*>(g.C)[row] = v;
    else static_cast<bf16_t*>(g.C)[row] = f32_to_bf16(v);
}

// =====================================================================================================================================
// MFMA variant ("fragment layout"): the same 12-bit weights, dots on the matrix cores.
//
// gemv1_p12_kernel spends five VALU operations per weight (four to rebuild the fp32 number + the FMA): it is issue-bound, not HBM-bound
// (2.56 instead of 0.75 x 2.67 ms per token).  Here a lane rebuilds bf16 PAIRS (byte permute, and-or, multiply, and-or: two operations
// per weight) and v_mfma_f32_16x16x32_bf16 does the products: B operand = 16 weight rows x 32 k, A operand = x as THREE bf16 rows
// (hi + lo + lo2 of x * 2^64: 24 significant bits, i.e. the fp32 activation exactly), so lane j < 16 finds sum_k w[j][k] (hi + lo + lo2)[k]
// in its own accumulator registers d[0] + d[1] + d[2]: exact bf16 x bf16 products, fp32 accumulation - the arithmetic of gemv1_kernel
// up to the summation order.  Rows 3 .. 15 of A repeat lo2 and are ignored.
//
// A block = 16 weight rows x 8 waves; a wave owns a contiguous range of 64-weight "step pairs" (two MFMA steps) of those rows and the
// 16 partial sums of every wave meet in LDS in wave order.  The planes are stored in the order the lanes consume them, so every wave
// instruction fetches 1 KB (P) / 512 B (E) contiguous:
//     P [N/16][K/64][64 lanes][16 B]   lane = q * 16 + r (r = row in the block, q = k-quarter): bytes h * 8 + i = weight
//                                      k = sp * 64 + h * 32 + q * 8 + i of row rb * 16 + r            (h: MFMA step of the pair)
//     E [N/16][K/64][64 lanes][ 8 B]   the same weights' codes, two per byte (low nibble = even k)
// (ops.PackedBf12 builds both layouts from the same P / E bytes; code 0 <-> bf16 zero, patches and ebase as above.)
struct P12M {
    const u32x4_t* P;
    const u32x2_t* E;
    const int32_t* ebase;
    const int32_t* patch_ptr;
    const int32_t* patch_col;
    const bf16_t* patch_val;
    const float* parts;  // PARTS form: x = the merge of the split-KV attention partials [K / pD heads][pS][pD + 4] (decode.hip)
    int pD, pS;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8m_t;
typedef __attribute__((ext_vector_type(4))) float f32x4m_t;

// two weights -> one dword of two bf16 numbers 1.m x 2^(code - 127) (code 0, P 0 -> 0): pd = [b1 b1 b0 b0] (each P byte twice: the sign
// lands on bit 15 / 31, the mantissa on bits 6..0 / 22..16), e = the byte with the two codes (-> exponent-field bits 10..7 / 26..23)
__device__ __forceinline__ uint32_t pair_p12m(uint32_t pd, uint32_t e) {
    return (pd & 0x807f807fu) | ((e * 0x00080080u) & 0x07800780u);
}
// the eight weights of a lane for one MFMA step: P dwords x0 (k 0..3), x1 (k 4..7), code bytes in e
__device__ __forceinline__ bf16x8m_t frag_p12m(uint32_t x0, uint32_t x1, uint32_t e) {
    u32x4_t d;
    d[0] = pair_p12m(__builtin_amdgcn_perm(0u, x0, 0x01010000u), e & 0xffu);
    d[1] = pair_p12m(__builtin_amdgcn_perm(0u, x0, 0x03030202u), (e >> 8) & 0xffu);
    d[2] = pair_p12m(__builtin_amdgcn_perm(0u, x1, 0x01010000u), (e >> 16) & 0xffu);
    d[3] = pair_p12m(__builtin_amdgcn_perm(0u, x1, 0x03030202u), e >> 24);
    return __builtin_bit_cast(bf16x8m_t, d);
}

// (4-wave blocks, six per CU: 2.53 instead of 2.39 ms per token.)
// kWavesM = 8: 512-thread blocks, three fit a CU at <= 80 registers (1024-thread blocks: two only at <= 64, which this loop does not fit
// without spills), so one block's prologue / tail runs under the others' streaming - the matrices with more row blocks than CUs.
// kWavesM = 16: when there is at most one block per CU anyway (N <= 4096: o_proj, down_proj) all 16 wave slots of its SIMDs' share
// belong to that block: twice the loads in flight.
// U: step pairs in flight per lane, U x (16 + 8) bytes (8 with two 8-wave blocks per CU was measured slower than 4 with three: 2.66 vs
// 2.44 ms per token)
// PARTS: the activation row is not read from g.A but merged, while it is staged, from the four (o, max, sum) partials per head that
// llama_decode_attn_parts wrote: x[h][d] = sum_s e^(m_s - M) o_s[d] / sum_s e^(m_s - M) l_s (the o_proj of the decode step).
template <bool RMS, int kWavesM, int U, bool PARTS = false>
__global__ __launch_bounds__(64 * kWavesM, kWavesM == 16 ? 4 : 6) void gemv1_p12m_kernel(GemmArgs g, P12M p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // x * 2^64 as three bf16 planes [3][K]
    __shared__ float s_red[kWavesM];
    __shared__ float s_part[kWavesM][16];
    __shared__ float s_patch[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = g.K, nsp = K >> 6;
    const int per = (nsp + kWavesM - 1) / kWavesM;
    const int s0 = wave * per, s1 = min(nsp, s0 + per);
    const int64_t base = (int64_t)blockIdx.x * nsp * 64 + lane;
    const u32x4_t* pp = p.P + base;
    const u32x2_t* ep = p.E + base;
    u32x4_t w[U];
    u32x2_t e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int sp = min(s0 + u, nsp - 1);
        w[u] = __builtin_nontemporal_load(pp + (int64_t)sp * 64);
        e[u] = __builtin_nontemporal_load(ep + (int64_t)sp * 64);
    }
    // The tail's operands.  kWavesM == 16 (one block per CU: nothing else would hide a round trip after the last MFMA): fetched NOW,
    // behind the weights - raw bits only, a conversion here would be a use and make the wave wait, in order, for every load issued
    // so far.  kWavesM == 8 (three blocks per CU, 80 registers): fetched in the tail, under the other blocks' streaming.
    constexpr bool EARLY = kWavesM == 16;  // (for the 8-wave form too: measured no faster)
    const bool fin = wave == 0 && lane < 16;
    const int row = blockIdx.x * 16 + (lane & 15);  // (N % 16 == 0: every row is live)
    int eb = 0;
    uint32_t bias_raw = 0, res_raw = 0;  // raw bits (16 or 32 of them); shifted / converted in the tail
    auto load_fin = [&]() {
        if (fin) {
            eb = p.ebase[row];
            if (g.bias) bias_raw = g.bias[row];
            if (g.residual) res_raw = g.res_f32 ? reinterpret_cast<const uint32_t*>(g.residual)[row] : (uint32_t)g.residual[row];
        }
    };
    int e_pp0 = 0, e_pp1 = 0, e_col = 0;
    uint32_t e_val = 0;
    if (EARLY) {
        e_pp0 = p.patch_ptr[blockIdx.x * 16 + wave];
        e_pp1 = p.patch_ptr[blockIdx.x * 16 + wave + 1];
        if (e_pp0 + lane < e_pp1) {
            e_col = p.patch_col[e_pp0 + lane];
            e_val = p.patch_val[e_pp0 + lane];
        }
        load_fin();
    }
    // ---- stage x * 2^64 (x * gamma * 2^64) as hi + lo + lo2 bf16 planes, sum(x^2) of the unscaled row ----
    const float xs = __builtin_ldexpf(1.0f, kXScaleExp);
    float ssq = 0.0f;
    uint32_t* xw = reinterpret_cast<uint32_t*>(smem);
#ifdef IVLM_ABL_XSTAGE  // (ablation, never in the product build: the upper bound of what producer-side x staging could save -
                        //  no x / gamma loads, no hi + lo + lo2 split, no sum of squares; the planes get a constant)
    for (int c = threadIdx.x; c < (K >> 2) && !PARTS; c += 64 * kWavesM) {
        *reinterpret_cast<u32x2_t*>(xw + 2 * c) = u32x2_t{0x5f805f80u, 0x5f805f80u};
        *reinterpret_cast<u32x2_t*>(xw + (K >> 1) + 2 * c) = u32x2_t{0u, 0u};
        *reinterpret_cast<u32x2_t*>(xw + K + 2 * c) = u32x2_t{0u, 0u};
        ssq = 1.0f;
    }
    for (int c = threadIdx.x; c < (PARTS ? (K >> 2) : 0); c += 64 * kWavesM) {
#else
    for (int c = threadIdx.x; c < (K >> 2); c += 64 * kWavesM) {
#endif
        f32x4v_t xv4;
        if (PARTS) {
            const int k = c << 2, hd = k / p.pD, d0 = k - hd * p.pD, ps = p.pD + 4;
            const float* ph = p.parts + (int64_t)hd * p.pS * ps;
            float m[4], l[4];
            f32x4v_t o4[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int sc = s2 < p.pS ? s2 : 0;  // (pS = 2: ranges 2, 3 repeat range 0 with weight 0)
                m[s2] = ph[sc * ps + p.pD];
                l[s2] = ph[sc * ps + p.pD + 1];
                o4[s2] = *reinterpret_cast<const f32x4v_t*>(ph + sc * ps + d0);
            }
            const float M = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
            float den = 0.0f;
            xv4 = f32x4v_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float wgt = s2 < p.pS ? __expf(m[s2] - M) : 0.0f;
                den = fmaf(wgt, l[s2], den);
                xv4 += o4[s2] * wgt;
            }
            xv4 = xv4 * (1.0f / den);
        } else {
            xv4 = reinterpret_cast<const f32x4v_t*>(g.A)[c];
        }
        float v[4] = {xv4[0], xv4[1], xv4[2], xv4[3]};
        if (RMS) {
            const u32x2_t gv = *(reinterpret_cast<const u32x2_t*>(g.rms_w) + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) ssq += v[j] * v[j];
            v[0] *= __uint_as_float(gv[0] << 16);
            v[1] *= __uint_as_float(gv[0] & 0xffff0000u);
            v[2] *= __uint_as_float(gv[1] << 16);
            v[3] *= __uint_as_float(gv[1] & 0xffff0000u);
        }
        uint32_t hi[2], lo[2], l2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float a = v[2 * j] * xs, b = v[2 * j + 1] * xs;
            const float ah = bf16_to_f32(f32_to_bf16(a)), bh = bf16_to_f32(f32_to_bf16(b));
            hi[j] = pack_bf16x2(ah, bh);
            a -= ah;
            b -= bh;
            const float al = bf16_to_f32(f32_to_bf16(a)), bl = bf16_to_f32(f32_to_bf16(b));
            lo[j] = pack_bf16x2(al, bl);
            l2[j] = pack_bf16x2(a - al, b - bl);
        }
        *reinterpret_cast<u32x2_t*>(xw + 2 * c) = u32x2_t{hi[0], hi[1]};
        *reinterpret_cast<u32x2_t*>(xw + (K >> 1) + 2 * c) = u32x2_t{lo[0], lo[1]};
        *reinterpret_cast<u32x2_t*>(xw + K + 2 * c) = u32x2_t{l2[0], l2[1]};
    }
    if (RMS) {
        ssq = wave_sum(ssq);
        if (lane == 0) s_red[wave] = ssq;
    }
    __syncthreads();
    // A fragments: lane (i = lane & 15: plane min(i, 2); q = lane >> 4) reads k = step * 32 + q * 8 .. + 7 of its plane
    const int plane = min(lane & 15, 2);
    const u32x4_t* xa = reinterpret_cast<const u32x4_t*>(smem) + ((plane * K) >> 3) + (lane >> 4);
    f32x4m_t d = {0.0f, 0.0f, 0.0f, 0.0f};
    auto step_pair = [&](const u32x4_t& pw, const u32x2_t& ew, int sp) {
        const bf16x8m_t a0 = __builtin_bit_cast(bf16x8m_t, xa[sp * 8]);
        const bf16x8m_t a1 = __builtin_bit_cast(bf16x8m_t, xa[sp * 8 + 4]);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, frag_p12m(pw[0], pw[1], ew[0]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, frag_p12m(pw[2], pw[3], ew[1]), d, 0, 0, 0);
    };
    constexpr int UH = U / 2;
    for (int sp = s0; sp < s1; sp += U) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int u = hf * UH; u < (hf + 1) * UH; ++u)
                if (sp + u < s1) step_pair(w[u], e[u], sp + u);
#pragma unroll
            for (int u = hf * UH; u < (hf + 1) * UH; ++u) {
                const int nx = sp + U + u;
                if (nx < s1) {
                    w[u] = __builtin_nontemporal_load(pp + (int64_t)nx * 64);
                    e[u] = __builtin_nontemporal_load(ep + (int64_t)nx * 64);
                }
            }
        }
    }
    if (lane < 16) s_part[wave][lane] = d[0] + d[1] + d[2];
    // ---- the tail: wave w owns the patches of row w (and w + 8 in the 8-wave form) of the block (the nonzero weights outside the row's exponent window:
    //      exact bf16 values, on average < 1 per row) against (hi + lo + lo2)[col] = the fp32 activation * 2^64 (the three parts add up
    //      exactly); the 16 finishing lanes of wave 0 fetch their row's exponent base / bias / residual meanwhile ----
    if (!EARLY) load_fin();
    if (EARLY) {  // pin the first USE of the early loads here (the compiler hoists a shift / an address computation to the load, and
                  // with it the in-order wait for everything issued before)
        asm volatile("" : "+v"(e_col), "+v"(e_val), "+v"(eb), "+v"(bias_raw), "+v"(res_raw));
    }
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(smem);
    auto xat = [&](int col) { return (bf16_to_f32(xb[col]) + bf16_to_f32(xb[K + col])) + bf16_to_f32(xb[2 * K + col]); };
#pragma unroll
    for (int h = 0; h < 16 / kWavesM; ++h) {
        const int r = wave + kWavesM * h;
        const int pp0 = EARLY ? e_pp0 : p.patch_ptr[blockIdx.x * 16 + r];  // (wave-uniform)
        const int pp1 = EARLY ? e_pp1 : p.patch_ptr[blockIdx.x * 16 + r + 1];
        float pacc = 0.0f;
        if (pp0 < pp1) {
            int i = pp0 + lane;
            if (EARLY) {
                if (i < pp1) pacc = __uint_as_float(e_val << 16) * xat(e_col);
                i += 64;
            }
            for (; i < pp1; i += 64) pacc = fmaf(bf16_to_f32(p.patch_val[i]), xat(p.patch_col[i]), pacc);
            pacc = wave_sum(pacc);
        }
        if (lane == 0) s_patch[r] = pacc;
    }
    __syncthreads();
    if (!fin) return;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < kWavesM; ++i) acc += s_part[i][lane];
    acc = __builtin_ldexpf(acc, eb - kXScaleExp) + __builtin_ldexpf(s_patch[lane], -kXScaleExp);
    if (RMS) {
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < kWavesM; ++i) q += s_red[i];
        acc *= rsqrtf(q / (float)K + g.rms_eps);
    }
    float v = acc + __uint_as_float(bias_raw << 16);
    if (g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved: the even lane finishes the pair
        const float up = __shfl_down(v, 1, 64);
        if (lane & 1) return;
        const float o = (v / (1.0f + __expf(-v))) * up;
        const int64_t idx = row >> 1;
        if (g.out_f32) static_cast<float*>(g.C)[idx] = o;
        else static_cast<bf16_t*>(g.C)[idx] = f32_to_bf16(o);
        return;
    }
    v = act1(v, g.act) + __uint_as_float(g.res_f32 ? res_raw : res_raw << 16);
    if (g.out_f32) static_cast<float*>(g.C)[row] = v;
    else static_cast<bf16_t*>(g.C)[row] = f32_to_bf16(v);
}

// =====================================================================================================================================
// M <= 16 activation rows (the batched decode step, BASELINE configs[2]: one token of each of B sequences) on the same packed weights:
// the structure of skinny_mfma_kernel (gemv_mfma.hip) - block = T tiles of 16 weight rows x 8 waves over K, activation fragments loaded
// straight from L2 (fp32 rows, split into hi + lo bf16 operands in registers: x = hi + lo to 2^-17, two MFMAs per weight fragment), the 8
// partial 16 x 16 tiles summed through LDS in wave order - with the weight fragments rebuilt from the fragment-layout planes (1 KB
// contiguous per wave instruction instead of 16 rows x 64 B, 0.75 x the bytes; two VALU operations per weight, shared by all M rows).
// A operand = the 16 weight rows, B operand = the activation rows: lane holds C[weight row 4 kg + i][activation row r].
__device__ __forceinline__ void split_pair_p12(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

template <bool RMS, int T>
__global__ __launch_bounds__(512, 2) void skinny_p12m_kernel(GemmArgs g, P12M p) {
    constexpr int kW = 8, U = 2;  // waves per block; step pairs in flight per lane (4 MFMA k-steps)
    __shared__ float s_part[kW][16][17];  // [wave][m][n]
    __shared__ float s_ssq[kW][16];
    __shared__ float s_fin[16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int K = g.K, nsp = K >> 6;
    const int per = (nsp + kW - 1) / kW;
    const int s0 = wave * per, s1 = min(nsp, s0 + per);
    const int ntiles = g.N >> 4;
    const u32x4_t* pp[T];
    const u32x2_t* ep[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t rb = min((int)blockIdx.x * T + t, ntiles - 1);
        pp[t] = p.P + rb * nsp * 64 + lane;
        ep[t] = p.E + rb * nsp * 64 + lane;
    }
    const bool xrow = r < g.M;
    const u32x4_t* xp = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const float*>(g.A) + (int64_t)(xrow ? r : 0) * g.lda);
    const u32x4_t* gp = reinterpret_cast<const u32x4_t*>(g.rms_w);
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    const float xs = __builtin_ldexpf(1.0f, kXScaleExp);
    f32x4m_t acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4m_t{0.0f, 0.0f, 0.0f, 0.0f};
    float ssq = 0.0f;
    for (int sp = s0; sp < s1; sp += U) {
        u32x4_t pw[U][T], xa[U][2], xb[U][2], gm[U][2];
        u32x2_t ew[U][T];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = sp + u < s1;
            const int spc = ok ? sp + u : s0;  // clamped (unconditional) weight loads keep the buffers in registers
#pragma unroll
            for (int t = 0; t < T; ++t) {
                pw[u][t] = __builtin_nontemporal_load(pp[t] + (int64_t)spc * 64);
                ew[u][t] = __builtin_nontemporal_load(ep[t] + (int64_t)spc * 64);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // the lane's 8 fp32 activations of MFMA step 2 spc + h: k = spc * 64 + h * 32 + kg * 8 ..
                const int c = spc * 8 + h * 4 + kg;  // 8-element chunk index
                xa[u][h] = (ok && xrow) ? xp[2 * c] : zero;
                xb[u][h] = (ok && xrow) ? xp[2 * c + 1] : zero;
                if (RMS) gm[u][h] = gp[c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[j] = __uint_as_float(xa[u][h][j]);
                    f[4 + j] = __uint_as_float(xb[u][h][j]);
                }
                if (RMS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ssq += f[2 * j] * f[2 * j] + f[2 * j + 1] * f[2 * j + 1];
                        f[2 * j] *= __uint_as_float(gm[u][h][j] << 16);
                        f[2 * j + 1] *= __uint_as_float(gm[u][h][j] & 0xffff0000u);
                    }
                }
                u32x4_t hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t h2, l2;
                    split_pair_p12(f[2 * j] * xs, f[2 * j + 1] * xs, h2, l2);
                    hi[j] = h2;
                    lo[j] = l2;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bf16x8m_t wf = frag_p12m(pw[u][t][2 * h], pw[u][t][2 * h + 1], ew[u][t][h]);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8m_t, hi), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8m_t, lo), acc[t], 0, 0, 0);
                }
            }
        }
    }
    // lane holds C[m = r][n = 4 * kg + i] of this wave's K slice, per tile
    if (RMS) {
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);
        if (kg == 0) s_ssq[wave][r] = ssq;
    }
    const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
    const float* xg = reinterpret_cast<const float*>(g.A);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int nt = ((int)blockIdx.x * T + t) * 16;
        if (t > 0) __syncthreads();  // the previous tile's s_fin readers are done
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wave][r][kg * 4 + i] = acc[t][i];
        __syncthreads();
        if (threadIdx.x < 256 && nt < g.N) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < kW; ++w) v += s_part[w][m][n];
            const int nn = nt + n;  // (N % 16 == 0)
            v = __builtin_ldexpf(v, p.ebase[nn] - kXScaleExp);
            if (m < g.M) {  // the nonzero weights outside the row's exponent window: exact bf16 values x fp32 activations
                for (int i = p.patch_ptr[nn]; i < p.patch_ptr[nn + 1]; ++i) {
                    const int col = p.patch_col[i];
                    float xv = xg[(int64_t)m * g.lda + col];
                    if (RMS) xv *= bf16_to_f32(g.rms_w[col]);
                    v = fmaf(bf16_to_f32(p.patch_val[i]), xv, v);
                }
            }
            if (RMS) {
                float q = 0.0f;
#pragma unroll
                for (int w = 0; w < kW; ++w) q += s_ssq[w][m];
                v *= rsqrtf(q / (float)K + g.rms_eps);
            }
            s_fin[m][n] = v + (g.bias ? bf16_to_f32(g.bias[nn]) : 0.0f);
        }
        __syncthreads();
        if (threadIdx.x >= 256 || m >= g.M || nt >= g.N) continue;
        float v = s_fin[m][n];
        int64_t col = nt + n;
        if (g.act == ACT_SWIGLU) {  // rows (gate_j, up_j) interleaved: even n pairs with n + 1
            if (n & 1) continue;
            const float up = s_fin[m][n + 1];
            v = (v / (1.0f + __expf(-v))) * up;
            col >>= 1;
        } else {
            v = act1(v, g.act);
            if (g.residual) {
                const int64_t idx = (int64_t)m * g.ldr + col;
                v += g.res_f32 ? reinterpret_cast<const float*>(g.residual)[idx] : bf16_to_f32(g.residual[idx]);
            }
        }
        if (g.out_f32) static_cast<float*>(g.C)[(int64_t)m * g.ldc + col] = v;
        else static_cast<bf16_t*>(g.C)[(int64_t)m * g.ldc + col] = f32_to_bf16(v);
    }
}

// exact reconstruction of the bf16 matrix (the losslessness check of the tests; not on the path): one thread per weight
__global__ __launch_bounds__(256) void unpack_p12_kernel(P12 p, int N, int K, bf16_t* __restrict__ out) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / K), c = (int)(i - (int64_t)r * K);
        const uint32_t b = p.P[(int64_t)r * p.ldp + c];
        const uint32_t code = (p.E[(int64_t)r * p.lde + (c >> 1)] >> ((c & 1) * 4)) & 0xfu;
        out[i] = code ? (bf16_t)(((b & 0x80u) << 8) | ((uint32_t)(p.ebase[r] + (int)code) << 7) | (b & 0x7fu)) : (bf16_t)0;
    }
}
__global__ __launch_bounds__(256) void unpack_p12_patch_kernel(P12 p, int N, int K, bf16_t* __restrict__ out) {
    const int r = blockIdx.x;
    for (int i = p.patch_ptr[r] + threadIdx.x; i < p.patch_ptr[r + 1]; i += 256) out[(int64_t)r * K + p.patch_col[i]] = p.patch_val[i];
}

}  // namespace
}  // namespace ivlm

using namespace ivlm;

static bool p12_ok(const void* P, int64_t ldp, const void* E, int64_t lde, const int32_t* ebase, const int32_t* patch_ptr,
                   const int32_t* patch_col, const void* patch_val, int N, int K) {
    return P && E && ebase && patch_ptr && patch_col && patch_val && N > 0 && K > 0 && !(K & 15) && ldp >= K && lde >= K / 2 &&
           !(ldp & 15) && !(lde & 7) && !(reinterpret_cast<uintptr_t>(P) & 15) && !(reinterpret_cast<uintptr_t>(E) & 7);
}

extern "C" int ivlm_gemv1_bf12(const float* x, const void* P, int64_t ldp, const void* E, int64_t lde, const int32_t* ebase,
                               const int32_t* patch_ptr, const int32_t* patch_col, const void* patch_val, void* C, const void* bias,
                               const void* residual, int N, int K, int act, int out_f32, const void* rms_w, float rms_eps, int flags,
                               ivlm_stream_t stream) {
    ivlm_enter();
    if (!x || !C || !p12_ok(P, ldp, E, lde, ebase, patch_ptr, patch_col, patch_val, N, K)) return IVLM_ERR_INVALID_ARG;
    if ((size_t)K * 4 > 60 * 1024) return IVLM_ERR_UNSUPPORTED;  // the fp32 x image lives in LDS
    if (act == ACT_SWIGLU && ((N & 1) || residual)) return IVLM_ERR_UNSUPPORTED;
    GemmArgs g;
    g.A = reinterpret_cast<const bf16_t*>(x);
    g.a_f32 = 1;
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    g.M = 1; g.N = N; g.K = K;
    g.lda = K;
    g.act = act;
    g.out_f32 = out_f32;
    g.rms_w = static_cast<const bf16_t*>(rms_w);
    g.rms_eps = rms_eps;
    P12 p{static_cast<const uint8_t*>(P), static_cast<const uint8_t*>(E), ldp, lde, ebase, patch_ptr, patch_col,
          static_cast<const bf16_t*>(patch_val)};
    hipStream_t st = ivlm_stream(stream);
    static ivlm_dev_mask_t set0{0}, set1{0};
    const dim3 grid((N + kWaves - 1) / kWaves), block(64 * kWaves);
    if (rms_w) {
        auto kfn = gemv1_p12_kernel<true>;
        if (ivlm_dev_pending(set1)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            ivlm_dev_done(set1);
        }
        ivlm_launch(kfn, grid, block, (size_t)K * 4, st, g, p);
    } else {
        auto kfn = gemv1_p12_kernel<false>;
        if (ivlm_dev_pending(set0)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            ivlm_dev_done(set0);
        }
        ivlm_launch(kfn, grid, block, (size_t)K * 4, st, g, p);
    }
    return ivlm_launch_status();
}

extern "C" int ivlm_unpack_bf12(const void* P, int64_t ldp, const void* E, int64_t lde, const int32_t* ebase, const int32_t* patch_ptr,
                                const int32_t* patch_col, const void* patch_val, int N, int K, void* w_out, ivlm_stream_t stream) {
    ivlm_enter();
    if (!w_out || !p12_ok(P, ldp, E, lde, ebase, patch_ptr, patch_col, patch_val, N, K)) return IVLM_ERR_INVALID_ARG;
    P12 p{static_cast<const uint8_t*>(P), static_cast<const uint8_t*>(E), ldp, lde, ebase, patch_ptr, patch_col,
          static_cast<const bf16_t*>(patch_val)};
    hipStream_t st = ivlm_stream(stream);
    const int64_t total = (int64_t)N * K;
    unpack_p12_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 8), 256, 0, st>>>(p, N, K, static_cast<bf16_t*>(w_out));
    unpack_p12_patch_kernel<<<N, 256, 0, st>>>(p, N, K, static_cast<bf16_t*>(w_out));
    return ivlm_launch_status();
}

int g_skinny_p12m_tiles = 0;       // 0 = rule in ivlm_gemv16_bf12m (A/B hook: ivlm_gemv16_bf12m_tuning)
int g_p12m_wide_max_blocks = 256;  // A/B hooks: ivlm_gemv1_bf12m_tuning
int g_p12m_deep = 1;
namespace ivlm {
int g_decode_parts_S = 4;  // key ranges per head of the split-KV attention / o_proj pair (A/B hook: ivlm_decode_parts_tuning; 2 or 4)
}

// MFMA variant on the fragment layout (see gemv1_p12m_kernel): Pf / Ef = the P / E bytes of ivlm_gemv1_bf12 re-ordered as
// [N/16][K/64][64 lanes][16 | 8 bytes]; N % 16 == 0, K % 64 == 0, 6 K bytes of LDS.  Same contract otherwise.
static int gemv1_bf12m(const float* x, const float* parts, int pD, const void* Pf, const void* Ef, const int32_t* ebase,
                      const int32_t* patch_ptr, const int32_t* patch_col, const void* patch_val, void* C, const void* bias,
                      const void* residual, int N, int K, int act, int out_f32, const void* rms_w, float rms_eps, int flags,
                      ivlm_stream_t stream) {
    ivlm_enter();
    if ((!x && !parts) || !C || !Pf || !Ef || !ebase || !patch_ptr || !patch_col || !patch_val || N <= 0 || K <= 0)
        return IVLM_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(Pf) & 15) || (reinterpret_cast<uintptr_t>(Ef) & 7) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(parts) & 15))
        return IVLM_ERR_INVALID_ARG;
    if (parts && (pD <= 0 || (pD & 3) || K % pD != 0 || rms_w)) return IVLM_ERR_INVALID_ARG;
    if ((N & 15) || (K & 63) || (size_t)K * 6 > 100 * 1024) return IVLM_ERR_UNSUPPORTED;
    if (act == ACT_SWIGLU && residual) return IVLM_ERR_UNSUPPORTED;
    GemmArgs g;
    g.A = reinterpret_cast<const bf16_t*>(x);
    g.a_f32 = 1;
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    g.M = 1; g.N = N; g.K = K;
    g.lda = K;
    g.act = act;
    g.out_f32 = out_f32;
    g.rms_w = static_cast<const bf16_t*>(rms_w);
    g.rms_eps = rms_eps;
    P12M p{static_cast<const u32x4_t*>(Pf), static_cast<const u32x2_t*>(Ef), ebase, patch_ptr, patch_col,
           static_cast<const bf16_t*>(patch_val), parts, pD, g_decode_parts_S};
    hipStream_t st = ivlm_stream(stream);
    const dim3 grid(N / 16);
    const bool wide = grid.x <= (unsigned)g_p12m_wide_max_blocks;  // at most one block per CU: 16 waves per block
    auto go = [&](auto kfn, int waves, ivlm_dev_mask_t& set) {
        if (ivlm_dev_pending(set)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            ivlm_dev_done(set);
        }
        ivlm_launch(kfn, grid, dim3(64 * waves), (size_t)K * 6, st, g, p);
    };
    static ivlm_dev_mask_t set[8];  // (zero-initialised; one mask per instantiation, one bit per device)
    if (parts) {  // (the o_proj of a decode step: K = hidden, never the 8-deep form)
        if (wide) go(gemv1_p12m_kernel<false, 16, 4, true>, 16, set[6]);
        else go(gemv1_p12m_kernel<false, 8, 4, true>, 8, set[7]);
        return ivlm_launch_status();
    }
    // (one block per CU and a long row - down_proj: 8 step pairs in flight per lane, so that most of the block's 264 KB is requested
    //  BEFORE the x staging, which otherwise runs with only the first 96 KB on their way)
    const bool deep = wide && (K >> 6) > 4 * 16 && g_p12m_deep;
    if (rms_w) {
        if (deep) go(gemv1_p12m_kernel<true, 16, 8>, 16, set[4]);
        else if (wide) go(gemv1_p12m_kernel<true, 16, 4>, 16, set[0]);
        else go(gemv1_p12m_kernel<true, 8, 4>, 8, set[1]);
    } else {
        if (deep) go(gemv1_p12m_kernel<false, 16, 8>, 16, set[5]);
        else if (wide) go(gemv1_p12m_kernel<false, 16, 4>, 16, set[2]);
        else go(gemv1_p12m_kernel<false, 8, 4>, 8, set[3]);
    }
    return ivlm_launch_status();
}

extern "C" int ivlm_gemv1_bf12m(const float* x, const void* Pf, const void* Ef, const int32_t* ebase, const int32_t* patch_ptr,
                                const int32_t* patch_col, const void* patch_val, void* C, const void* bias, const void* residual, int N,
                                int K, int act, int out_f32, const void* rms_w, float rms_eps, int flags, ivlm_stream_t stream) {
    if (!x) return IVLM_ERR_INVALID_ARG;
    return gemv1_bf12m(x, nullptr, 0, Pf, Ef, ebase, patch_ptr, patch_col, patch_val, C, bias, residual, N, K, act, out_f32, rms_w, rms_eps,
                       flags, stream);
}

// ... with the activation row merged from the split-KV attention partials parts[K / D][4][D + 4] of ivlm_llama_decode_attn_parts (the
// o_proj of a decode step: no RMSNorm prologue)
extern "C" int ivlm_gemv1_bf12m_parts(const float* parts, int D, const void* Pf, const void* Ef, const int32_t* ebase,
                                      const int32_t* patch_ptr, const int32_t* patch_col, const void* patch_val, void* C, const void* bias,
                                      const void* residual, int N, int K, int act, int out_f32, int flags, ivlm_stream_t stream) {
    if (!parts) return IVLM_ERR_INVALID_ARG;
    return gemv1_bf12m(nullptr, parts, D, Pf, Ef, ebase, patch_ptr, patch_col, patch_val, C, bias, residual, N, K, act, out_f32, nullptr,
                       0.0f, flags, stream);
}

extern "C" void ivlm_gemv1_bf12m_tuning(int wide_max_blocks) {
    g_p12m_deep = wide_max_blocks >= 0;  // (negative: |value| as the limit, without the 8-deep form)
    g_p12m_wide_max_blocks = wide_max_blocks < 0 ? -wide_max_blocks : wide_max_blocks;
}

extern "C" int ivlm_decode_parts_tuning(int ranges) {
    if (ranges != 2 && ranges != 4) return IVLM_ERR_INVALID_ARG;
    g_decode_parts_S = ranges;
    return IVLM_OK;
}

// The batched decode step's linears (M <= 16 fp32 activation rows) on the fragment-layout planes: out[M, N] = act(x . W^T + bias) +
// residual; x = hi + lo bf16 operands (2^-17), fp32 accumulation.  N % 16 == 0, K % 64 == 0; lda / ldc / ldr in elements.
extern "C" int ivlm_gemv16_bf12m(const float* x, int64_t lda, int M, const void* Pf, const void* Ef, const int32_t* ebase,
                                 const int32_t* patch_ptr, const int32_t* patch_col, const void* patch_val, void* C, int64_t ldc,
                                 const void* bias, const void* residual, int64_t ldr, int N, int K, int act, int out_f32,
                                 const void* rms_w, float rms_eps, int flags, ivlm_stream_t stream) {
    ivlm_enter();
    if (!x || !C || !Pf || !Ef || !ebase || !patch_ptr || !patch_col || !patch_val || M <= 0 || M > 16 || N <= 0 || K <= 0)
        return IVLM_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(Pf) & 15) || (reinterpret_cast<uintptr_t>(Ef) & 7) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (lda & 3) || lda < K)
        return IVLM_ERR_INVALID_ARG;
    if ((N & 15) || (K & 63)) return IVLM_ERR_UNSUPPORTED;
    if (act == ACT_SWIGLU && residual) return IVLM_ERR_UNSUPPORTED;
    //  the row strides of the output and of the residual must hold a whole row
    if (ldc < (act == ACT_SWIGLU ? N / 2 : N) || (residual && ldr < N)) return IVLM_ERR_INVALID_ARG;
    GemmArgs g;
    g.A = reinterpret_cast<const bf16_t*>(x);
    g.a_f32 = 1;
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldc = ldc; g.ldr = ldr;
    g.act = act;
    g.out_f32 = out_f32;
    g.rms_w = static_cast<const bf16_t*>(rms_w);
    g.rms_eps = rms_eps;
    P12M p{static_cast<const u32x4_t*>(Pf), static_cast<const u32x2_t*>(Ef), ebase, patch_ptr, patch_col,
           static_cast<const bf16_t*>(patch_val), nullptr, 0, 0};
    hipStream_t st = ivlm_stream(stream);
    const int tiles = N / 16;
    int T = g_skinny_p12m_tiles;
    if (T <= 0) {  // fewest (blocks per CU) x (tiles per block), then the most tiles (least activation re-reading): as skinny_tiles()
        int best_cost = 1 << 30;
        for (int t : {1, 2, 3}) {
            const int blocks = (tiles + t - 1) / t, cost = ((blocks + 255) / 256) * t;
            if (cost <= best_cost) { T = t; best_cost = cost; }
        }
    }
    const dim3 grid((tiles + T - 1) / T), block(512);
    if (rms_w) {
        if (T == 1) ivlm_launch(skinny_p12m_kernel<true, 1>, grid, block, 0, st, g, p);
        else if (T == 2) ivlm_launch(skinny_p12m_kernel<true, 2>, grid, block, 0, st, g, p);
        else ivlm_launch(skinny_p12m_kernel<true, 3>, grid, block, 0, st, g, p);
    } else {
        if (T == 1) ivlm_launch(skinny_p12m_kernel<false, 1>, grid, block, 0, st, g, p);
        else if (T == 2) ivlm_launch(skinny_p12m_kernel<false, 2>, grid, block, 0, st, g, p);
        else ivlm_launch(skinny_p12m_kernel<false, 3>, grid, block, 0, st, g, p);
    }
    return ivlm_launch_status();
}

extern "C" void ivlm_gemv16_bf12m_tuning(int tiles_per_block) { g_skinny_p12m_tiles = tiles_per_block > 3 ? 3 : tiles_per_block; }

// =====================================================================================================================================
// The packer (weight preparation, once per matrix): bf16 [N, K] -> fragment-layout planes + per-row exponent bases + CSR patches, so that
// a C caller needs nothing but this library to build what ivlm_gemv1_bf12m / ivlm_llama_decode_step_bf12 read.  Two calls around one
// host read of the patch total: ivlm_pack_bf12m_count (ebase, patch_ptr), ivlm_pack_bf12m_fill (planes, patch_col / patch_val in column
// order).  Rows beyond N_valid (padding up to a multiple of 16) pack as zeros.
namespace ivlm {
namespace {

__global__ __launch_bounds__(256) void p12_rowstats_kernel(const bf16_t* __restrict__ w, int n_valid, int K, int32_t* __restrict__ ebase,
                                                           int32_t* __restrict__ cnt) {
    __shared__ int s_red[4];
    const int row = blockIdx.x;
    int mx = 0;
    if (row < n_valid)
        for (int c = threadIdx.x; c < K; c += 256) mx = max(mx, (int)((w[(int64_t)row * K + c] >> 7) & 0xff));
    for (int o = 32; o; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    const int eb = max(mx - 15, 0);
    int n = 0;
    if (row < n_valid)
        for (int c = threadIdx.x; c < K; c += 256) {
            const uint32_t b = w[(int64_t)row * K + c];
            n += ((int)((b >> 7) & 0xff) - eb < 1) && (b & 0x7fffu);  // nonzero and outside the window
        }
    for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        ebase[row] = eb;
        cnt[row + 1] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        if (row == 0) cnt[0] = 0;
    }
}

// in-place inclusive scan of cnt[1 .. N] (one block; N is at most a few 10^4 rows)
__global__ __launch_bounds__(1024) void p12_scan_kernel(int32_t* __restrict__ ptr, int N) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 1; base <= N; base += 1024) {
        const int i = base + threadIdx.x;
        int v = i <= N ? ptr[i] : 0, x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        int off = s_carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) off += s_w[k];
        if (i <= N) ptr[i] = x + off;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = x + off;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void p12_fill_kernel(const bf16_t* __restrict__ w, int n_valid, int K, const int32_t* __restrict__ ebase,
                                                       const int32_t* __restrict__ patch_ptr, uint8_t* __restrict__ Pf,
                                                       uint8_t* __restrict__ Ef, int32_t* __restrict__ patch_col,
                                                       bf16_t* __restrict__ patch_val) {
    __shared__ int s_cnt[4];
    const int row = blockIdx.x, rb = row >> 4, r = row & 15, nsp = K >> 6;
    const int eb = ebase[row];
    int out = patch_ptr[row];
    for (int c0 = 0; c0 < K; c0 += 512) {  // a thread packs the weight PAIR (c, c + 1): one P half-word, one E byte
        const int c = c0 + 2 * threadIdx.x;
        uint32_t b[2] = {0u, 0u};
        if (row < n_valid && c < K) {
            const uint32_t two = *reinterpret_cast<const uint32_t*>(w + (int64_t)row * K + c);
            b[0] = two & 0xffffu;
            b[1] = two >> 16;
        }
        uint32_t pbyte[2], code[2];
        bool esc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cd = (int)((b[j] >> 7) & 0xff) - eb;
            const bool in = cd >= 1;
            esc[j] = !in && (b[j] & 0x7fffu);
            pbyte[j] = in ? (((b[j] >> 8) & 0x80u) | (b[j] & 0x7fu)) : 0u;
            code[j] = in ? (uint32_t)cd : 0u;
        }
        if (c < K) {
            const int sp = c >> 6, h = (c >> 5) & 1, q = (c >> 3) & 3, i = c & 7;
            const int64_t lane_rec = ((((int64_t)rb * nsp + sp) * 4 + q) * 16 + r);
            *reinterpret_cast<uint16_t*>(Pf + lane_rec * 16 + h * 8 + i) = (uint16_t)(pbyte[0] | (pbyte[1] << 8));
            Ef[lane_rec * 8 + h * 4 + (i >> 1)] = (uint8_t)(code[0] | (code[1] << 4));
        }
        // patches in column order: an exclusive scan of the escape flags over the 512 columns of this trip
        const int mine = (int)esc[0] + (int)esc[1];
        int x = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_cnt[threadIdx.x >> 6] = x;
        __syncthreads();
        int off = out + x - mine;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) off += s_cnt[k];
        if (esc[0]) { patch_col[off] = c; patch_val[off] = (bf16_t)b[0]; ++off; }
        if (esc[1]) { patch_col[off] = c + 1; patch_val[off] = (bf16_t)b[1]; }
        out += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
}

}  // namespace
}  // namespace ivlm

// n_rows = the rows of the planes (a multiple of 16 >= N_valid); ebase [n_rows], patch_ptr [n_rows + 1] (the caller reads
// patch_ptr[n_rows] = the number of patches, allocates patch_col / patch_val of at least max(1, that) entries, and calls _fill)
extern "C" int ivlm_pack_bf12m_count(const void* w, int N_valid, int n_rows, int K, int32_t* ebase, int32_t* patch_ptr,
                                     ivlm_stream_t stream) {
    ivlm_enter();
    if (!w || !ebase || !patch_ptr || N_valid <= 0 || n_rows < N_valid || (n_rows & 15) || K <= 0 || (K & 63)) return IVLM_ERR_INVALID_ARG;
    hipStream_t st = ivlm_stream(stream);
    p12_rowstats_kernel<<<n_rows, 256, 0, st>>>(static_cast<const bf16_t*>(w), N_valid, K, ebase, patch_ptr);
    p12_scan_kernel<<<1, 1024, 0, st>>>(patch_ptr, n_rows);
    return ivlm_launch_status();
}

extern "C" int ivlm_pack_bf12m_fill(const void* w, int N_valid, int n_rows, int K, const int32_t* ebase, const int32_t* patch_ptr, void* Pf,
                                    void* Ef, int32_t* patch_col, void* patch_val, ivlm_stream_t stream) {
    ivlm_enter();
    if (!w || !ebase || !patch_ptr || !Pf || !Ef || !patch_col || !patch_val || N_valid <= 0 || n_rows < N_valid || (n_rows & 15) ||
        K <= 0 || (K & 63) || (reinterpret_cast<uintptr_t>(w) & 3))
        return IVLM_ERR_INVALID_ARG;
    p12_fill_kernel<<<n_rows, 256, 0, ivlm_stream(stream)>>>(static_cast<const bf16_t*>(w), N_valid, K, ebase, patch_ptr,
                                                             static_cast<uint8_t*>(Pf), static_cast<uint8_t*>(Ef), patch_col,
                                                             static_cast<bf16_t*>(patch_val));
    return ivlm_launch_status();
}
