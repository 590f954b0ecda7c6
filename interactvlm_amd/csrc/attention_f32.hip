// fp32 attention for the SAM mask decoder's tiny attentions (transformer.py:220-242: 8 heads of 16 or 32 channels;
// 9 tokens x 4096 image positions or the reverse): o = softmax((q.k^T) / sqrt(d)) . v with fp32 operands end to end.
//
// These attentions are ~75 MFLOP per image - nothing next to the 24 TFLOP of the encoder - but they sit at the very end of
// the pipeline, where every rounding goes straight into the mask logits.  The decoder therefore keeps fp32 activations
// (DESIGN.md "precision policy") and its attention runs on the VALU in fp32 instead of rounding q, k, v and P to bf16 for
// the MFMA kernel of attention.hip.  Two shapes of work:
//   few keys  (Sk <= 64: image -> tokens, token self-attention): one THREAD per query, K / V of the (batch, head) in LDS;
//   many keys (tokens -> image):                                  one WAVE per query, lanes stride the keys with an online
//                                                                 softmax, fixed-order butterfly merge (deterministic).
#include "kernels.h"

namespace ivlm {
namespace {

struct AttnF32Args {
    const float *q, *k, *v;
    float* o;
    int64_t q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
    int B, H, Sq, Sk, D, kv_div;
    float scale;
};

template <int D4>  // D = 4 * D4
__global__ __launch_bounds__(256) void attn_f32_fewkeys_kernel(AttnF32Args a) {
    constexpr int D = 4 * D4;
    __shared__ float ks[64 * D], vs[64 * D];
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int bk = b / a.kv_div;
    const float* kp = a.k + bk * a.k_bs + h * a.k_hs;
    const float* vp = a.v + bk * a.v_bs + h * a.v_hs;
    for (int i = threadIdx.x; i < a.Sk * D4; i += 256) {
        const int j = i / D4, c = i % D4;
        reinterpret_cast<float4*>(ks)[i] = *reinterpret_cast<const float4*>(kp + j * a.k_rs + 4 * c);
        reinterpret_cast<float4*>(vs)[i] = *reinterpret_cast<const float4*>(vp + j * a.v_rs + 4 * c);
    }
    __syncthreads();
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= a.Sq) return;
    float q[D], acc[D];
    const float* qp = a.q + b * a.q_bs + h * a.q_hs + (int64_t)qi * a.q_rs;
#pragma unroll
    for (int c = 0; c < D4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * c);
        q[4 * c] = t.x; q[4 * c + 1] = t.y; q[4 * c + 2] = t.z; q[4 * c + 3] = t.w;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    for (int j = 0; j < a.Sk; ++j) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(q[d], ks[j * D + d], s);
        s *= a.scale;
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);  // (first key: exp(-inf) = 0)
        l = l * corr + p;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vs[j * D + d], acc[d] * corr);
        m = mn;
    }
    const float inv = 1.0f / l;
    float* op = a.o + b * a.o_bs + h * a.o_hs + (int64_t)qi * a.o_rs;
#pragma unroll
    for (int c = 0; c < D4; ++c)
        *reinterpret_cast<float4*>(op + 4 * c) =
            make_float4(acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv);
}

template <int D4>
__global__ __launch_bounds__(256) void attn_f32_manykeys_kernel(AttnF32Args a) {
    constexpr int D = 4 * D4;
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per (b, h, query)
    if (w >= (int64_t)a.B * a.H * a.Sq) return;
    const int qi = (int)(w % a.Sq), h = (int)((w / a.Sq) % a.H), b = (int)(w / ((int64_t)a.Sq * a.H));
    const int bk = b / a.kv_div;
    const float* kp = a.k + bk * a.k_bs + h * a.k_hs;
    const float* vp = a.v + bk * a.v_bs + h * a.v_hs;
    const float* qp = a.q + b * a.q_bs + h * a.q_hs + (int64_t)qi * a.q_rs;
    float q[D], acc[D];
#pragma unroll
    for (int c = 0; c < D4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * c);
        q[4 * c] = t.x; q[4 * c + 1] = t.y; q[4 * c + 2] = t.z; q[4 * c + 3] = t.w;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    for (int j = lane; j < a.Sk; j += 64) {
        float kr[D], vr[D];
#pragma unroll
        for (int c = 0; c < D4; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(kp + (int64_t)j * a.k_rs + 4 * c);
            const float4 u = *reinterpret_cast<const float4*>(vp + (int64_t)j * a.v_rs + 4 * c);
            kr[4 * c] = t.x; kr[4 * c + 1] = t.y; kr[4 * c + 2] = t.z; kr[4 * c + 3] = t.w;
            vr[4 * c] = u.x; vr[4 * c + 1] = u.y; vr[4 * c + 2] = u.z; vr[4 * c + 3] = u.w;
        }
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(q[d], kr[d], s);
        s *= a.scale;
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);
        l = l * corr + p;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vr[d], acc[d] * corr);
        m = mn;
    }
    // merge the 64 partial softmaxes: common maximum, rescale, fixed butterfly sums
    const float mw = wave_max(m);
    const float sc = (m == -INFINITY) ? 0.0f : __expf(m - mw);  // lanes without a key contribute nothing
    l = wave_sum(l * sc);
    const float inv = 1.0f / l;
    float* op = a.o + b * a.o_bs + h * a.o_hs + (int64_t)qi * a.o_rs;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float r = wave_sum(acc[d] * sc);
        if (lane == 0) op[d] = r * inv;
    }
}

// any head width up to 256 (AttentionSplitter, components.py:155-193: one head of 128 channels over V = 4 keys): one wave
// per query, lane l owns channels 4l..4l+3, one butterfly sum per key
__global__ __launch_bounds__(256) void attn_f32_wide_kernel(AttnF32Args a) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)a.B * a.H * a.Sq) return;
    const int qi = (int)(w % a.Sq), h = (int)((w / a.Sq) % a.H), b = (int)(w / ((int64_t)a.Sq * a.H));
    const int bk = b / a.kv_div;
    const bool on = 4 * lane < a.D;
    const float* kp = a.k + bk * a.k_bs + h * a.k_hs + 4 * lane;
    const float* vp = a.v + bk * a.v_bs + h * a.v_hs + 4 * lane;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 q = on ? *reinterpret_cast<const float4*>(a.q + b * a.q_bs + h * a.q_hs + (int64_t)qi * a.q_rs + 4 * lane) : zero;
    float4 acc = zero;
    float m = -INFINITY, l = 0.0f;
    for (int j = 0; j < a.Sk; ++j) {
        const float4 kr = on ? *reinterpret_cast<const float4*>(kp + (int64_t)j * a.k_rs) : zero;
        const float4 vr = on ? *reinterpret_cast<const float4*>(vp + (int64_t)j * a.v_rs) : zero;
        const float s = wave_sum(q.x * kr.x + q.y * kr.y + q.z * kr.z + q.w * kr.w) * a.scale;
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);
        l = l * corr + p;
        acc.x = fmaf(p, vr.x, acc.x * corr); acc.y = fmaf(p, vr.y, acc.y * corr);
        acc.z = fmaf(p, vr.z, acc.z * corr); acc.w = fmaf(p, vr.w, acc.w * corr);
        m = mn;
    }
    const float inv = 1.0f / l;
    if (on)
        *reinterpret_cast<float4*>(a.o + b * a.o_bs + h * a.o_hs + (int64_t)qi * a.o_rs + 4 * lane) =
            make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

}  // namespace

int attention_f32(const float* q, const float* k, const float* v, float* o, const int64_t* st12, int B, int H, int Sq, int Sk,
                  int D, float scale, int kv_div, hipStream_t st) {
    if (!q || !k || !v || !o || !st12 || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || kv_div <= 0 || B % kv_div) return IVLM_ERR_INVALID_ARG;
    if (D <= 0 || D > 256 || (D & 3)) return IVLM_ERR_UNSUPPORTED;
    for (int i = 0; i < 12; ++i)
        if (st12[i] & 3) return IVLM_ERR_INVALID_ARG;  // 16-byte rows
    AttnF32Args a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.q_bs = st12[0]; a.q_hs = st12[1]; a.q_rs = st12[2];
    a.k_bs = st12[3]; a.k_hs = st12[4]; a.k_rs = st12[5];
    a.v_bs = st12[6]; a.v_hs = st12[7]; a.v_rs = st12[8];
    a.o_bs = st12[9]; a.o_hs = st12[10]; a.o_rs = st12[11];
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.D = D; a.kv_div = kv_div; a.scale = scale;
    if (D != 16 && D != 32) {
        const int64_t waves = (int64_t)B * H * Sq;
        attn_f32_wide_kernel<<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(a);
    } else if (Sk <= 64) {
        dim3 grid((Sq + 255) / 256, B * H);
        if (D == 16) attn_f32_fewkeys_kernel<4><<<grid, 256, 0, st>>>(a);
        else attn_f32_fewkeys_kernel<8><<<grid, 256, 0, st>>>(a);
    } else {
        const int64_t waves = (int64_t)B * H * Sq;
        const unsigned grid = (unsigned)((waves + 3) / 4);
        if (D == 16) attn_f32_manykeys_kernel<4><<<grid, 256, 0, st>>>(a);
        else attn_f32_manykeys_kernel<8><<<grid, 256, 0, st>>>(a);
    }
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" int ivlm_attention_f32(const float* q, const float* k, const float* v, float* o, const int64_t* strides_host, int B,
                                  int H, int Sq, int Sk, int D, float scale, int kv_batch_div, ivlm_stream_t stream) {
    ivlm_enter();
    return ivlm::attention_f32(q, k, v, o, strides_host, B, H, Sq, Sk, D, scale, kv_batch_div, ivlm_stream(stream));
}
