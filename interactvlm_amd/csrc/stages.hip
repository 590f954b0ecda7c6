// Stage-level entry points of the language path (ivlm_llama_prefill / ivlm_llama_decode): thin C++ sequencers
// over the op launchers of this library, so that a non-Python caller can run a stage without re-implementing llava.py.
//
//   ivlm_llama_prefill      HF LlamaModel.forward over T new positions with a KV cache (model/llava/model/language_model/
//                           llava_llama.py:93-102 -> transformers LlamaModel): RMSNorm -> q|k|v GEMM -> RoPE + cache append ->
//                           causal flash attention -> o_proj (+ fp32 residual) -> RMSNorm -> gate|up GEMM with the SwiGLU
//                           epilogue -> down_proj (+ residual), final RMSNorm.
//   ivlm_llama_decode_step  the same for ONE new position on the weight-streaming kernels (fp32 activations, exact products):
//                           RMSNorm fused into the q|k|v and gate|up GEMVs, attention + o_proj in one launch when the grid fits
//                           the CUs, SwiGLU / residual adds in the GEMV epilogues.
//
// Weights arrive as a table of DEVICE pointers (one ivlm_llama_layer per decoder layer, bf16, the layouts of
// interactvlm_amd/llava.py: q|k|v rows concatenated, gate/up rows interleaved); all scratch lives in a caller workspace; nothing
// is allocated, nothing synchronises.  Same kernels and the same launch order as interactvlm_amd/llava.py: the results are
// bit-identical to the Python-sequenced path (tests/test_stages_gpu.py).
#include <algorithm>

#include "kernels.h"

namespace ivlm {

// the split-K rule of the small-M tile GEMMs (also exported: interactvlm_amd/ops.py asks this function, one source of truth)
int gemm_splitk_choice(int M, int N, int K, int act, int has_rms) {
    if (M <= 8 || M > 1024 || act == ACT_SWIGLU || has_rms || (N & 3) || (K & 63)) return 1;
    if (M <= 16 && N >= 1024 && K >= 1024) return 1;  // the skinny MFMA kernel takes these
    const long tiles = (long)((M + 127) / 128) * ((N + 63) / 64);
    if (tiles >= 256) return 1;
    int best = 1;
    const int k64 = K / 64;
    for (int sp = 2; sp <= std::min<long>(8, 1024 / tiles); ++sp)
        if (k64 % sp == 0 && K / sp >= 512) best = sp;
    return best;
}

namespace {

inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

struct Carver {
    char* p;
    size_t left;
    bool ok = true;
    void* take(size_t bytes) {
        bytes = al(bytes);
        if (bytes > left) {
            ok = false;
            return nullptr;
        }
        void* r = p;
        p += bytes;
        left -= bytes;
        return r;
    }
};

int lin(const void* A, int a_f32, int64_t lda, const void* W, int64_t ldw, void* C, int out_f32, int64_t ldc, const void* res,
        int res_f32, int M, int N, int K, int act, const void* rms_w, float eps, float* splitk_ws, size_t splitk_bytes,
        hipStream_t st) {
    GemmArgs g;
    g.A = static_cast<const bf16_t*>(A);
    g.a_f32 = a_f32;
    g.W = static_cast<const bf16_t*>(W);
    g.C = C;
    g.residual = static_cast<const bf16_t*>(res);
    g.res_f32 = res_f32;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = N;
    g.M = M; g.N = N; g.K = K;
    g.act = act;
    g.out_f32 = out_f32;
    g.rms_w = static_cast<const bf16_t*>(rms_w);
    g.rms_eps = eps;
    const int sp = a_f32 ? 1 : gemm_splitk_choice(M, N, K, act, rms_w != nullptr);
    if (sp > 1 && (ldc & 3) == 0) return gemm_bf16_splitk(g, sp, splitk_ws, splitk_bytes, st);
    return linear_bf16(g, st);
}

__global__ void bump_kernel(int32_t* a, int32_t* b) {
    if (a) *a += 1;
    if (b) *b += 1;
}

}  // namespace
}  // namespace ivlm

using namespace ivlm;

extern "C" int ivlm_gemm_splitk_choice(int M, int N, int K, int act, int has_rms) {
    return gemm_splitk_choice(M, N, K, act, has_rms);
}

static bool cfg_ok(const ivlm_llama_cfg* c) {
    return c && c->layers > 0 && c->hidden > 0 && c->heads > 0 && c->inter > 0 && c->hidden % c->heads == 0 && c->max_len > 0 &&
           (c->hidden & 7) == 0 && (c->inter & 7) == 0;
}

extern "C" size_t ivlm_llama_prefill_workspace_bytes(const ivlm_llama_cfg* c, int T) {
    if (!cfg_ok(c) || T <= 0) return 0;
    const size_t h = c->hidden, in = c->inter, t = T;
    size_t b = al(t * h * 2) + al(t * 3 * h * 2) + al(t * h * 2) + al(t * in * 2) + 2 * al(t * h * 4);
    b += al(8 * t * std::max(h, in) * 4);  // split-K partials (<= 8 slices of [T, N <= hidden])
    return b + 256;
}

extern "C" int ivlm_llama_prefill(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm,
                                  void* kcache, void* vcache, const float* cos_tab, const float* sin_tab, const float* x_in, int T,
                                  int pos0, float* hidden_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    if (!cfg_ok(c) || !layers_host || !final_norm || !kcache || !vcache || !x_in || !hidden_out || !workspace || T <= 0 || pos0 < 0 ||
        pos0 + T > c->max_len)
        return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_llama_prefill_workspace_bytes(c, T)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    const int Hd = c->hidden, H = c->heads, D = Hd / H, I = c->inter;
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    bf16_t* y = static_cast<bf16_t*>(cv.take((size_t)T * Hd * 2));
    bf16_t* qkv = static_cast<bf16_t*>(cv.take((size_t)T * 3 * Hd * 2));
    bf16_t* att = static_cast<bf16_t*>(cv.take((size_t)T * Hd * 2));
    bf16_t* hh = static_cast<bf16_t*>(cv.take((size_t)T * I * 2));
    float* xa = static_cast<float*>(cv.take((size_t)T * Hd * 4));
    float* xb = static_cast<float*>(cv.take((size_t)T * Hd * 4));
    const size_t skb = (size_t)8 * T * std::max(Hd, I) * 4;
    float* sk = static_cast<float*>(cv.take(skb));
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    const int64_t cache_layer = (int64_t)c->max_len * Hd;
    const float* x = x_in;
    int rc;
    for (int l = 0; l < c->layers; ++l) {
        const ivlm_llama_layer& L = layers_host[l];
        bf16_t* kc = static_cast<bf16_t*>(kcache) + l * cache_layer;
        bf16_t* vc = static_cast<bf16_t*>(vcache) + l * cache_layer;
        if ((rc = rmsnorm(x, 1, static_cast<const bf16_t*>(L.ln1), y, 0, T, Hd, c->eps, st))) return rc;
        if ((rc = lin(y, 0, Hd, L.qkv, Hd, qkv, 0, 3 * Hd, nullptr, 0, T, 3 * Hd, Hd, ACT_NONE, nullptr, 0.f, sk, skb, st))) return rc;
        if ((rc = rope_kv(qkv, 3 * Hd, T, H, D, pos0, c->theta, kc, vc, st, cos_tab, sin_tab))) return rc;
        AttnArgs a{};
        a.q = qkv; a.k = kc; a.v = vc; a.o = att;
        a.q_bs = 0; a.q_hs = D; a.q_rs = 3 * Hd;
        a.k_bs = 0; a.k_hs = D; a.k_rs = Hd;
        a.v_bs = 0; a.v_hs = D; a.v_rs = Hd;
        a.o_bs = 0; a.o_hs = D; a.o_rs = Hd;
        a.B = 1; a.H = H; a.Sq = T; a.Sk = pos0 + T; a.D = D;
        a.scale = 1.0f / sqrtf((float)D);
        a.causal = 1; a.q_pos0 = pos0;
        a.kv_batch_div = 1; a.prescale_q = 0;
        if ((rc = attention_bf16(a, st))) return rc;
        float* x1 = (x == xa) ? xb : xa;
        if ((rc = lin(att, 0, Hd, L.o, Hd, x1, 1, Hd, x, 1, T, Hd, Hd, ACT_NONE, nullptr, 0.f, sk, skb, st))) return rc;
        if ((rc = rmsnorm(x1, 1, static_cast<const bf16_t*>(L.ln2), y, 0, T, Hd, c->eps, st))) return rc;
        if ((rc = lin(y, 0, Hd, L.gu, Hd, hh, 0, I, nullptr, 0, T, 2 * I, Hd, ACT_SWIGLU, nullptr, 0.f, sk, skb, st))) return rc;
        float* x2 = (x1 == xa) ? xb : xa;
        if ((rc = lin(hh, 0, I, L.down, I, x2, 1, Hd, x1, 1, T, Hd, I, ACT_NONE, nullptr, 0.f, sk, skb, st))) return rc;
        x = x2;
    }
    return rmsnorm(x, 1, static_cast<const bf16_t*>(final_norm), hidden_out, 1, T, Hd, c->eps, st);
}

extern "C" size_t ivlm_llama_decode_workspace_bytes(const ivlm_llama_cfg* c) {
    if (!cfg_ok(c)) return 0;
    const size_t h = c->hidden, in = c->inter, L = c->layers;
    // activations | fused-launch state: per-layer arrival counters (128 B apart), status + step words, per-layer attention rows
    return al(3 * h * 4) + al(in * 4) + 2 * al(h * 4) + al(h * 4) + al(L * 32 * 4) + 256 + al(L * h * 4) + 256;
}

extern "C" int ivlm_llama_decode_step(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm,
                                      void* kcache, void* vcache, const float* cos_tab, const float* sin_tab, const float* x_in,
                                      int32_t* pos_dev, int advance, float* hidden_out, void* workspace, size_t workspace_bytes,
                                      ivlm_stream_t stream) {
    ivlm_enter();
    if (!cfg_ok(c) || !layers_host || !final_norm || !kcache || !vcache || !cos_tab || !sin_tab || !x_in || !pos_dev || !hidden_out ||
        !workspace)
        return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_llama_decode_workspace_bytes(c)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    const int Hd = c->hidden, H = c->heads, D = Hd / H, I = c->inter;
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    float* qkv = static_cast<float*>(cv.take((size_t)3 * Hd * 4));
    float* hh = static_cast<float*>(cv.take((size_t)I * 4));
    float* xa = static_cast<float*>(cv.take((size_t)Hd * 4));
    float* xb = static_cast<float*>(cv.take((size_t)Hd * 4));
    float* att = static_cast<float*>(cv.take((size_t)Hd * 4));
    int32_t* counters = static_cast<int32_t*>(cv.take((size_t)c->layers * 32 * 4));
    int32_t* words = static_cast<int32_t*>(cv.take(256));  // [0] status, [1] tokens decoded so far (the caller zeroes the
    float* scratch = static_cast<float*>(cv.take((size_t)c->layers * Hd * 4));  // workspace at the start of a generation)
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        IVLM_HIP_TRY(hipGetDevice(&dev));
        IVLM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const bool fuse = (Hd == 512 || Hd == 1024 || Hd == 4096 || Hd == 5120) && H + Hd / 32 <= cus;
    const int64_t cache_layer = (int64_t)c->max_len * Hd;
    const float scale = 1.0f / sqrtf((float)D);
    const float* x = x_in;
    int rc;
    for (int l = 0; l < c->layers; ++l) {
        const ivlm_llama_layer& L = layers_host[l];
        bf16_t* kc = static_cast<bf16_t*>(kcache) + l * cache_layer;
        bf16_t* vc = static_cast<bf16_t*>(vcache) + l * cache_layer;
        if ((rc = lin(x, 1, Hd, L.qkv, Hd, qkv, 1, 3 * Hd, nullptr, 0, 1, 3 * Hd, Hd, ACT_NONE, L.ln1, c->eps, nullptr, 0, st))) return rc;
        float* x1 = (x == xa) ? xb : xa;
        if (fuse) {
            rc = llama_attn_oproj(qkv, kc, vc, c->max_len, scratch + (size_t)l * Hd, static_cast<const bf16_t*>(L.o), x, x1, H, D,
                                  c->theta, scale, cos_tab, sin_tab, pos_dev, words + 1, counters + l * 32, words, st);
            if (rc) return rc;
        } else {
            if ((rc = llama_decode_attn(qkv, 1, kc, vc, c->max_len, att, H, D, 0, c->theta, scale, st, cos_tab, sin_tab, pos_dev))) return rc;
            if ((rc = lin(att, 1, Hd, L.o, Hd, x1, 1, Hd, x, 1, 1, Hd, Hd, ACT_NONE, nullptr, 0.f, nullptr, 0, st))) return rc;
        }
        if ((rc = lin(x1, 1, Hd, L.gu, Hd, hh, 1, I, nullptr, 0, 1, 2 * I, Hd, ACT_SWIGLU, L.ln2, c->eps, nullptr, 0, st))) return rc;
        float* x2 = (x1 == xa) ? xb : xa;
        if ((rc = lin(hh, 1, I, L.down, I, x2, 1, Hd, x1, 1, 1, Hd, I, ACT_NONE, nullptr, 0.f, nullptr, 0, st))) return rc;
        x = x2;
    }
    if ((rc = rmsnorm(x, 1, static_cast<const bf16_t*>(final_norm), hidden_out, 1, 1, Hd, c->eps, st))) return rc;
    bump_kernel<<<1, 1, 0, st>>>(fuse ? words + 1 : nullptr, advance ? pos_dev : nullptr);  // tokens decoded += 1 (position += 1)
    return ivlm_launch_status();
}
