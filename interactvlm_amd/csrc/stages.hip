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
#include <cstdlib>

#include "kernels.h"

namespace ivlm {

// the split-K rule of the small-M tile GEMMs (also exported: interactvlm_amd/ops.py asks this function, one source of truth)
int gemm_splitk_choice(int M, int N, int K, int act, int has_rms) {
    if (M <= 8 || M > 1024 || act == ACT_SWIGLU || has_rms || (N & 3) || (K & 63)) return 1;
    if (M <= 16 && N >= 1024 && K >= 1024) return 1;  // the skinny MFMA kernel takes these
    if (M > 128 && M <= 352 && N >= 8192) {  // the row-stationary 176 x 128 tiles (gemm.hip): two K slices when they leave CUs idle
        const long t176 = (long)((M + 175) / 176) * ((N + 127) / 128);
        return (t176 < 256 && (K / 64) % 2 == 0 && K / 2 >= 512) ? 2 : 1;
    }
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

// The fused split-K reduction (SplitKFused, kernels.h) is OPT-IN for the stage sequencers, as it is for the host model
// (ops.SPLITK_FUSED): on MI355X it measured 2 - 3 x SLOWER than the two-launch form (down_proj 64 -> 192 us, prefill 12.0 -> 26.3 ms;
// tools/experiments/README.md) - each block's agent-scope release / acquire around its arrival costs more than the launch it saves.
// Switch: ivlm_stages_splitk_fused(1), or IVLM_SPLITK_FUSED=1 in the environment at first use.  Off (default): no counters, no
// memset, the two-launch form everywhere.
// When it is on, the arrival counters live in the LAST 16 KB of a sequencer's split-K region: every stage call zeroes them once
// (sk_counters_zero, right after the region is carved) and the GEMMs leave them at zero; a product whose partials would reach into
// them takes the two-launch form on the region BELOW the counters (sk_partial_bytes), so the partials can never overwrite them.
int g_sk_fused = -1;  // -1: not read yet
bool sk_fused_on() {
    if (g_sk_fused < 0) {
        const char* e = getenv("IVLM_SPLITK_FUSED");
        g_sk_fused = (e && e[0] == '1' && e[1] == 0) ? 1 : 0;
    }
    return g_sk_fused == 1;
}
constexpr size_t kSkCounterBytes = (size_t)kSplitKCounters * 4;
int32_t* sk_counters(float* sk, size_t skb, size_t partial_bytes) {
    if (!sk_fused_on() || !sk || skb < partial_bytes + kSkCounterBytes + 16) return nullptr;
    return reinterpret_cast<int32_t*>(reinterpret_cast<char*>(sk) + ((skb - kSkCounterBytes) & ~(size_t)15));
}
// bytes of the region the PARTIALS may use: everything when the fused form is off, the part below the counters when it is on
size_t sk_partial_bytes(size_t skb) {
    if (!sk_fused_on() || skb < kSkCounterBytes + 16) return skb;
    return (skb - kSkCounterBytes) & ~(size_t)15;
}
int sk_counters_zero(float* sk, size_t skb, hipStream_t st);
// one split-K product of a sequencer: fused when the switch is on and the partials stay below the counters, else the two-launch
// form - on the region below the counters, or (partials larger than that) on the whole region with the counters re-zeroed after
int sk_gemm(const GemmArgs& g, int sp, float* sk, size_t skb, hipStream_t st) {
    const size_t pb = (size_t)sp * g.M * g.N * 4;
    int32_t* c = sk_counters(sk, skb, pb);
    if (c) return gemm_bf16_splitk(g, sp, sk, skb, st, c);
    if (pb <= sk_partial_bytes(skb)) return gemm_bf16_splitk(g, sp, sk, sk_partial_bytes(skb), st, nullptr);
    if (int rc = gemm_bf16_splitk(g, sp, sk, skb, st, nullptr)) return rc;
    return sk_counters_zero(sk, skb, st);  // (only reachable with the switch on: the partials ran over the counter words)
}
int sk_counters_zero(float* sk, size_t skb, hipStream_t st) {
    int32_t* c = sk_counters(sk, skb, 0);
    if (c) IVLM_HIP_TRY(hipMemsetAsync(c, 0, kSkCounterBytes, st));
    return IVLM_OK;
}

int lin(const void* A, int a_f32, int64_t lda, const void* W, int64_t ldw, void* C, int out_f32, int64_t ldc, const void* res,
        int res_f32, int M, int N, int K, int act, const void* rms_w, float eps, float* splitk_ws, size_t splitk_bytes,
        hipStream_t st, int f16 = 0, int out_f16 = 0) {
    GemmArgs g;
    g.f16 = f16;          // A and W are IEEE halves (the fp16-operand prefill)
    g.out_f16 = out_f16;  // a 16-bit output is written as IEEE halves
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
    if (sp > 1 && (ldc & 3) == 0)
        return sk_gemm(g, sp, splitk_ws, splitk_bytes, st);
    return linear_bf16(g, st);
}

__global__ void bump_kernel(int32_t* a, int32_t* b) {
    if (a) *a += 1;
    if (b) *b += 1;
}

}  // namespace
}  // namespace ivlm

using namespace ivlm;

extern "C" int ivlm_stages_splitk_fused(int on) {
    const int prev = sk_fused_on() ? 1 : 0;
    if (on == 0 || on == 1) g_sk_fused = on;
    return prev;
}

extern "C" int ivlm_gemm_splitk_choice(int M, int N, int K, int act, int has_rms) {
    return gemm_splitk_choice(M, N, K, act, has_rms);
}

static bool cfg_ok(const ivlm_llama_cfg* c) {
    // (max_len: the decode attention keeps one score per cached position in LDS - decattn::kMaxT = 4096 positions)
    return c && c->layers > 0 && c->hidden > 0 && c->heads > 0 && c->inter > 0 && c->hidden % c->heads == 0 && c->max_len > 0 &&
           c->max_len <= 4096 && (c->hidden & 7) == 0 && (c->inter & 7) == 0;
}

extern "C" size_t ivlm_llama_prefill_workspace_bytes(const ivlm_llama_cfg* c, int T) {
    if (!cfg_ok(c) || T <= 0) return 0;
    const size_t h = c->hidden, in = c->inter, t = T;
    size_t b = al(t * h * 2) + al(t * 3 * h * 2) + al(t * h * 2) + al(t * in * 2) + 2 * al(t * h * 4);
    b += al(8 * t * std::max(h, in) * 4);  // split-K partials (<= 8 slices of [T, N <= hidden])
    return b + 256;
}

static int llama_prefill(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm, void* kcache,
                         void* vcache, const float* cos_tab, const float* sin_tab, const float* x_in, int T, int pos0, float* hidden_out,
                         void* workspace, size_t workspace_bytes, ivlm_stream_t stream, const int f16) {
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
    if (int rc0 = sk_counters_zero(sk, skb, st)) return rc0;
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    const int64_t cache_layer = (int64_t)c->max_len * Hd;
    const float* x = x_in;
    int rc;
    for (int l = 0; l < c->layers; ++l) {
        const ivlm_llama_layer& L = layers_host[l];
        bf16_t* kc = static_cast<bf16_t*>(kcache) + l * cache_layer;
        bf16_t* vc = static_cast<bf16_t*>(vcache) + l * cache_layer;
        if ((rc = rmsnorm(x, 1, static_cast<const bf16_t*>(L.ln1), y, f16 ? 4 : 0, T, Hd, c->eps, st))) return rc;
        if ((rc = lin(y, 0, Hd, L.qkv, Hd, qkv, 0, 3 * Hd, nullptr, 0, T, 3 * Hd, Hd, ACT_NONE, nullptr, 0.f, sk, skb, st, f16, f16))) return rc;
        if ((rc = rope_kv(qkv, 3 * Hd, T, H, D, pos0, c->theta, kc, vc, st, cos_tab, sin_tab, f16))) return rc;
        AttnArgs a{};
        a.f16 = f16;
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
        if ((rc = lin(att, 0, Hd, L.o, Hd, x1, 1, Hd, x, 1, T, Hd, Hd, ACT_NONE, nullptr, 0.f, sk, skb, st, f16, 0))) return rc;
        if ((rc = rmsnorm(x1, 1, static_cast<const bf16_t*>(L.ln2), y, f16 ? 4 : 0, T, Hd, c->eps, st))) return rc;
        if ((rc = lin(y, 0, Hd, L.gu, Hd, hh, 0, I, nullptr, 0, T, 2 * I, Hd, ACT_SWIGLU, nullptr, 0.f, sk, skb, st, f16, f16))) return rc;
        float* x2 = (x1 == xa) ? xb : xa;
        if ((rc = lin(hh, 0, I, L.down, I, x2, 1, Hd, x1, 1, T, Hd, I, ACT_NONE, nullptr, 0.f, sk, skb, st, f16, 0))) return rc;
        x = x2;
    }
    return rmsnorm(x, 1, static_cast<const bf16_t*>(final_norm), hidden_out, 1, T, Hd, c->eps, st);
}

extern "C" int ivlm_llama_prefill(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm,
                                  void* kcache, void* vcache, const float* cos_tab, const float* sin_tab, const float* x_in, int T,
                                  int pos0, float* hidden_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    return llama_prefill(c, layers_host, final_norm, kcache, vcache, cos_tab, sin_tab, x_in, T, pos0, hidden_out, workspace,
                         workspace_bytes, stream, 0);
}

// The default precision of the host model (interactvlm_amd/llava.py Llama._layer_f16): IEEE fp16 MFMA operands in one pass - the
// qkv / o / gu / down pointers of layers16_host are fp16 copies of the bf16 weights (ivlm_bf16_to_f16: exact inside the fp16
// range; ln1 / ln2 stay the bf16 norm weights), RMSNorm / q|k|v / SwiGLU outputs and the KV cache are fp16.  Needs T > 16.
extern "C" int ivlm_llama_prefill_f16(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers16_host, const void* final_norm,
                                      void* kcache16, void* vcache16, const float* cos_tab, const float* sin_tab, const float* x_in,
                                      int T, int pos0, float* hidden_out, void* workspace, size_t workspace_bytes,
                                      ivlm_stream_t stream) {
    if (T <= 16) return IVLM_ERR_UNSUPPORTED;  // (the fp16 tile GEMMs; short chunks go through ivlm_llama_decode_step_f16kv)
    return llama_prefill(c, layers16_host, final_norm, kcache16, vcache16, cos_tab, sin_tab, x_in, T, pos0, hidden_out, workspace,
                         workspace_bytes, stream, 1);
}

extern "C" size_t ivlm_llama_decode_workspace_bytes(const ivlm_llama_cfg* c) {
    if (!cfg_ok(c)) return 0;
    const size_t h = c->hidden, in = c->inter, L = c->layers;
    // activations | fused-launch state: per-layer arrival counters (128 B apart), status + step words, per-layer attention rows
    // + the split-KV attention partials of the packed step [heads][4][head_dim + 4] fp32
    return al(3 * h * 4) + al(in * 4) + 2 * al(h * 4) + al(h * 4) + al(L * 32 * 4) + 256 + al(L * h * 4) + 256 +
           al((size_t)c->heads * 4 * (h / c->heads + 4) * 4);
}

static int llama_decode_step(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm, void* kcache,
                             void* vcache, const float* cos_tab, const float* sin_tab, const float* x_in, int32_t* pos_dev, int advance,
                             float* hidden_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream, const int cache_f16) {
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
    const bool fuse = !cache_f16 && c->fuse_attn_oproj && (Hd == 512 || Hd == 1024 || Hd == 4096 || Hd == 5120) && H + Hd / 32 <= cus;
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
            if ((rc = llama_decode_attn(qkv, 1, kc, vc, c->max_len, att, H, D, 0, c->theta, scale, st, cos_tab, sin_tab, pos_dev, nullptr,
                                        nullptr, cache_f16)))
                return rc;
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

extern "C" int ivlm_llama_decode_step(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm,
                                      void* kcache, void* vcache, const float* cos_tab, const float* sin_tab, const float* x_in,
                                      int32_t* pos_dev, int advance, float* hidden_out, void* workspace, size_t workspace_bytes,
                                      ivlm_stream_t stream) {
    return llama_decode_step(c, layers_host, final_norm, kcache, vcache, cos_tab, sin_tab, x_in, pos_dev, advance, hidden_out, workspace,
                             workspace_bytes, stream, 0);
}

// ... against the fp16 KV cache ivlm_llama_prefill_f16 fills: the bf16 weights of layers_host (fp32 activations, exact products, as
// in every mode), K / V rows appended and read as IEEE halves (separate attention / o_proj launches).
extern "C" int ivlm_llama_decode_step_f16kv(const ivlm_llama_cfg* c, const ivlm_llama_layer* layers_host, const void* final_norm,
                                            void* kcache16, void* vcache16, const float* cos_tab, const float* sin_tab,
                                            const float* x_in, int32_t* pos_dev, int advance, float* hidden_out, void* workspace,
                                            size_t workspace_bytes, ivlm_stream_t stream) {
    return llama_decode_step(c, layers_host, final_norm, kcache16, vcache16, cos_tab, sin_tab, x_in, pos_dev, advance, hidden_out,
                             workspace, workspace_bytes, stream, 1);
}

// The same step with the four linears of a layer on losslessly packed weights (ivlm_gemv1_bf12m): the default decode path of the host
// model.  Separate attention / o_proj launches; fp16 or bf16 KV cache.
extern "C" int ivlm_llama_decode_step_bf12(const ivlm_llama_cfg* c, const ivlm_llama_layer_bf12* layers_host, const void* final_norm,
                                           void* kcache, void* vcache, int cache_dtype, const float* cos_tab, const float* sin_tab,
                                           const float* x_in, int32_t* pos_dev, int advance, float* hidden_out, void* workspace,
                                           size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    if (!cfg_ok(c) || !layers_host || !final_norm || !kcache || !vcache || !cos_tab || !sin_tab || !x_in || !pos_dev || !hidden_out ||
        !workspace || (cache_dtype != IVLM_BF16 && cache_dtype != IVLM_F16))
        return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_llama_decode_workspace_bytes(c)) return IVLM_ERR_WORKSPACE;
    const int Hd = c->hidden, H = c->heads, D = Hd / H, I = c->inter;
    if ((Hd & 63) || (I & 63)) return IVLM_ERR_UNSUPPORTED;
    hipStream_t st = ivlm_stream(stream);
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    float* qkv = static_cast<float*>(cv.take((size_t)3 * Hd * 4));
    float* hh = static_cast<float*>(cv.take((size_t)I * 4));
    float* xa = static_cast<float*>(cv.take((size_t)Hd * 4));
    float* xb = static_cast<float*>(cv.take((size_t)Hd * 4));
    float* att = static_cast<float*>(cv.take((size_t)Hd * 4));
    cv.take((size_t)c->layers * 32 * 4);  // (the regions of the other forms of the step: same workspace, same offsets)
    cv.take(256);
    cv.take((size_t)c->layers * Hd * 4);
    float* parts = static_cast<float*>(cv.take((size_t)H * 4 * (D + 4) * 4));
    const bool split = !c->fuse_attn_oproj;  // split-KV attention merged by the o_proj prologue (the host model's default);
                                             // fuse_attn_oproj != 0 selects the one-block attention + separate o_proj instead
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    auto gemv = [&](const float* x, const ivlm_bf12m& m, float* out, const float* res, int N, int K, int act, const void* rms) {
        return ivlm_gemv1_bf12m(x, m.Pf, m.Ef, m.ebase, m.patch_ptr, m.patch_col, m.patch_val, out, nullptr, res, N, K, act, 1, rms,
                                rms ? c->eps : 0.0f, res ? IVLM_GEMM_RES_F32 : 0, stream);
    };
    const int64_t cache_layer = (int64_t)c->max_len * Hd;
    const float scale = 1.0f / sqrtf((float)D);
    const float* x = x_in;
    int rc;
    for (int l = 0; l < c->layers; ++l) {
        const ivlm_llama_layer_bf12& L = layers_host[l];
        bf16_t* kc = static_cast<bf16_t*>(kcache) + l * cache_layer;
        bf16_t* vc = static_cast<bf16_t*>(vcache) + l * cache_layer;
        if ((rc = gemv(x, L.qkv, qkv, nullptr, 3 * Hd, Hd, ACT_NONE, L.ln1))) return rc;
        float* x1 = (x == xa) ? xb : xa;
        if (split) {
            if ((rc = llama_decode_attn_parts(qkv, kc, vc, c->max_len, parts, H, D, 0, c->theta, scale, st, cos_tab, sin_tab, pos_dev,
                                              cache_dtype == IVLM_F16)))
                return rc;
            if ((rc = ivlm_gemv1_bf12m_parts(parts, D, L.o.Pf, L.o.Ef, L.o.ebase, L.o.patch_ptr, L.o.patch_col, L.o.patch_val, x1, nullptr,
                                             x, Hd, Hd, ACT_NONE, 1, IVLM_GEMM_RES_F32, stream)))
                return rc;
        } else {
            if ((rc = llama_decode_attn(qkv, 1, kc, vc, c->max_len, att, H, D, 0, c->theta, scale, st, cos_tab, sin_tab, pos_dev, nullptr,
                                        nullptr, cache_dtype == IVLM_F16)))
                return rc;
            if ((rc = gemv(att, L.o, x1, x, Hd, Hd, ACT_NONE, nullptr))) return rc;
        }
        if ((rc = gemv(x1, L.gu, hh, nullptr, 2 * I, Hd, ACT_SWIGLU, L.ln2))) return rc;
        float* x2 = (x1 == xa) ? xb : xa;
        if ((rc = gemv(hh, L.down, x2, x1, Hd, I, ACT_NONE, nullptr))) return rc;
        x = x2;
    }
    if ((rc = rmsnorm(x, 1, static_cast<const bf16_t*>(final_norm), hidden_out, 1, 1, Hd, c->eps, st))) return rc;
    bump_kernel<<<1, 1, 0, st>>>(nullptr, advance ? pos_dev : nullptr);
    return ivlm_launch_status();
}

// =====================================================================================================================
// Vision stages: ivlm_clip_encode (CLIPVisionTower.forward + feature_select, clip_encoder.py:31-60) and ivlm_sam_encode
// (ImageEncoderViT.forward, image_encoder.py:110-125) as C++ sequencers - the launch order of interactvlm_amd/llava.py
// ClipTower._forward and interactvlm_amd/sam.py SamImageEncoder._forward (bf16 operands, fp32 residual stream).
// =====================================================================================================================
namespace ivlm {
namespace {

// window_partition / window_unpartition row maps of SAM's ViT on the device (image_encoder.py:263-318): part[w] = image row
// of window position w (-1 = zero padding), unpart[r] = window position of image row r
__global__ void sam_window_maps_kernel(int V, int g, int ws, int nw, int32_t* __restrict__ part, int32_t* __restrict__ unpart) {
    const int S = ws * ws;
    const int64_t total = (int64_t)V * nw * nw * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % ws), iy = (int)((i / ws) % ws);
        const int wx = (int)((i / S) % nw), wy = (int)((i / ((int64_t)S * nw)) % nw), v = (int)(i / ((int64_t)S * nw * nw));
        const int y = wy * ws + iy, x = wx * ws + ix;
        const bool ok = y < g && x < g;
        const int src = (v * g + y) * g + x;
        part[i] = ok ? src : -1;
        if (ok) unpart[src] = (int32_t)i;
    }
}

// dst[r] = row wherever part[r] < 0 (the zero-padded window positions: their q|k|v rows are the bias)
__global__ void fill_pad_rows_kernel(bf16_t* __restrict__ dst, int64_t ldd, const int32_t* __restrict__ part, int64_t rows,
                                     const bf16_t* __restrict__ row, int cols) {
    const int c8n = cols >> 3;
    const int64_t total = rows * c8n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c8n;
        if (part[r] >= 0) continue;
        const int c8 = (int)(i % c8n);
        *reinterpret_cast<uint4*>(dst + r * ldd + c8 * 8) = *reinterpret_cast<const uint4*>(row + c8 * 8);
    }
}

// generic tile-GEMM call of the sequencers (bias, activation, bf16 / fp32 residual with row modulo, row maps, split-K rule)
int gemm(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int out_f32, int64_t ldc, const void* bias,
         const void* res, int res_f32, int64_t ldr, int res_mod, int M, int N, int K, int act, const int32_t* out_rows,
         const int32_t* a_rows, float* sk, size_t skb, hipStream_t st, int a_split = 0, int out_split = 0, int f16 = 0,
         int out_f16 = 0) {
    GemmArgs g;
    g.f16 = f16;  // A and W are IEEE halves
    g.out_f16 = out_f16;  // a 16-bit output is written as IEEE halves
    if (a_split) {  // "parity" precision: A rows are [hi(K) | lo(K)] bf16
        g.a_split = 1;
        g.a_lo = K;
    }
    if (out_split) {  // the fp32 result written as [hi(N) | lo(N)] bf16 rows
        g.out_split = 1;
        g.c_lo = N;
        out_f32 = 1;
    }
    g.A = static_cast<const bf16_t*>(A);
    g.W = static_cast<const bf16_t*>(W);
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(res);
    g.res_f32 = res_f32;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.res_mod = res_mod;
    g.M = M; g.N = N; g.K = K;
    g.act = act;
    g.out_f32 = out_f32;
    g.out_rows = out_rows;
    g.a_rows = a_rows;
    const int sp = (out_rows || a_rows) ? 1 : gemm_splitk_choice(M, N, K, act, 0);
    if (sp > 1 && (ldc & 3) == 0 && sk && skb >= (size_t)sp * M * N * 4)
        return sk_gemm(g, sp, sk, skb, st);
    return linear_bf16(g, st);
}

}  // namespace
}  // namespace ivlm

extern "C" size_t ivlm_clip_encode_workspace_bytes(const ivlm_clip_cfg* c, int B) {
    if (!c || B <= 0) return 0;
    const size_t T = c->tokens, h = c->hidden, rows = (size_t)B * T;
    return al((size_t)B * (T - 1) * c->kpad * 2) + 2 * al(rows * h * 4) + al(rows * h * 2) + al(rows * 3 * h * 2) + al(rows * h * 2) +
           al(rows * c->inter * 2) + al(8 * rows * std::max<size_t>(3 * h, c->inter) * 4) + al(rows * 4) + 512;
}

static int clip_encode(const ivlm_clip_cfg* c, const ivlm_clip_head* hd, const ivlm_clip_layer* layers_host, const void* images, int B,
                       void* features_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream, const int f16) {
    ivlm_enter();
    if (!c || !hd || !layers_host || !images || !features_out || !workspace || B <= 0) return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_clip_encode_workspace_bytes(c, B)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    const int T = c->tokens, Hd = c->hidden, H = c->heads, D = Hd / H, I = c->inter, R = B * T;
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    bf16_t* cols = static_cast<bf16_t*>(cv.take((size_t)B * (T - 1) * c->kpad * 2));
    float* xa = static_cast<float*>(cv.take((size_t)R * Hd * 4));
    float* xb = static_cast<float*>(cv.take((size_t)R * Hd * 4));
    bf16_t* y = static_cast<bf16_t*>(cv.take((size_t)R * Hd * 2));
    bf16_t* qkv = static_cast<bf16_t*>(cv.take((size_t)R * 3 * Hd * 2));
    bf16_t* att = static_cast<bf16_t*>(cv.take((size_t)R * Hd * 2));
    bf16_t* hh = static_cast<bf16_t*>(cv.take((size_t)R * I * 2));
    const size_t skb = (size_t)8 * R * std::max(3 * Hd, I) * 4;
    float* sk = static_cast<float*>(cv.take(skb));
    if (int rc0 = sk_counters_zero(sk, skb, st)) return rc0;
    int32_t* prow = static_cast<int32_t*>(cv.take((size_t)R * 4));
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    int rc;
    if ((rc = im2col_nchw(static_cast<const bf16_t*>(images), cols, B, 3, c->image_size, c->image_size, c->patch, c->patch, c->kpad, st))) return rc;
    for (int b = 0; b < B; ++b) {  // patch GEMM writes rows 1..T-1 (+ their position embeddings); row 0 = class + position 0
        float* xrow = xa + (size_t)b * T * Hd;
        if ((rc = gemm(cols + (size_t)b * (T - 1) * c->kpad, c->kpad, hd->patch_w, c->kpad, xrow + Hd, 1, Hd, nullptr,
                       static_cast<const bf16_t*>(hd->pos) + Hd, 0, Hd, 0, T - 1, Hd, c->kpad, ACT_NONE, nullptr, nullptr, sk, skb, st)))
            return rc;
        if ((rc = gather_rows(xrow, 1, Hd, hd->cls_row, 1, Hd, nullptr, nullptr, 0, 0, 1, Hd, st))) return rc;
    }
    if ((rc = layernorm(xa, 1, static_cast<const bf16_t*>(hd->pre_ln_w), static_cast<const bf16_t*>(hd->pre_ln_b), xb, 1, R, Hd, c->eps, st))) return rc;
    float* x = xb;
    const int64_t strides[12] = {(int64_t)T * 3 * Hd, D, 3 * Hd, (int64_t)T * 3 * Hd, D, 3 * Hd, (int64_t)T * 3 * Hd, D, 3 * Hd,
                                 (int64_t)T * Hd, D, Hd};
    for (int l = 0; l < c->layers_run; ++l) {
        const ivlm_clip_layer& L = layers_host[l];
        const int k16 = f16 ? 4 : 0;  // (LayerNorm output kind: IEEE halves | bf16)
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(L.ln1_w), static_cast<const bf16_t*>(L.ln1_b), y, k16, R, Hd, c->eps, st))) return rc;
        if ((rc = gemm(y, Hd, L.qkv_w, Hd, qkv, 0, 3 * Hd, L.qkv_b, nullptr, 0, 0, 0, R, 3 * Hd, Hd, ACT_NONE, nullptr, nullptr, sk, skb, st, 0, 0, f16, f16))) return rc;
        if ((rc = (f16 ? ivlm_attention_f16 : ivlm_attention_bf16)(qkv, qkv + Hd, qkv + 2 * Hd, att, strides, B, H, T, T, D,
                                                                  1.0f / sqrtf((float)D), 0, 0, nullptr, nullptr, 0, 0, 1, 1, stream)))
            return rc;
        float* x1 = (x == xa) ? xb : xa;
        if ((rc = gemm(att, Hd, L.out_w, Hd, x1, 1, Hd, L.out_b, x, 1, Hd, 0, R, Hd, Hd, ACT_NONE, nullptr, nullptr, sk, skb, st, 0, 0, f16, 0))) return rc;
        if ((rc = layernorm(x1, 1, static_cast<const bf16_t*>(L.ln2_w), static_cast<const bf16_t*>(L.ln2_b), y, k16, R, Hd, c->eps, st))) return rc;
        if ((rc = gemm(y, Hd, L.fc1_w, Hd, hh, 0, I, L.fc1_b, nullptr, 0, 0, 0, R, I, Hd, ACT_QUICK_GELU, nullptr, nullptr, sk, skb, st, 0, 0, f16, f16))) return rc;
        float* x2 = (x1 == xa) ? xb : xa;
        if ((rc = gemm(hh, I, L.fc2_w, I, x2, 1, Hd, L.fc2_b, x1, 1, Hd, 0, R, Hd, I, ACT_NONE, nullptr, nullptr, sk, skb, st, 0, 0, f16, 0))) return rc;
        x = x2;
    }
    // drop the CLS row of every image, fp32 stream -> bf16 features (the mm_projector's operand); fp16 mode: [hi | lo] bf16 rows
    // of width 2 * hidden (the projector takes split rows there: 0.03 % of the image's FLOPs)
    const int fw = f16 ? 2 * Hd : Hd;
    for (int b = 0; b < B; ++b)
        if ((rc = gather_rows(static_cast<bf16_t*>(features_out) + (size_t)b * (T - 1) * fw, f16 ? 2 : 0, fw, x + ((size_t)b * T + 1) * Hd, 1, Hd,
                              nullptr, nullptr, 0, 0, T - 1, Hd, st)))
            return rc;
    (void)prow;
    return IVLM_OK;
}

extern "C" int ivlm_clip_encode(const ivlm_clip_cfg* c, const ivlm_clip_head* hd, const ivlm_clip_layer* layers_host,
                                const void* images, int B, void* features_out, void* workspace, size_t workspace_bytes,
                                ivlm_stream_t stream) {
    return clip_encode(c, hd, layers_host, images, B, features_out, workspace, workspace_bytes, stream, 0);
}

// The default precision of the host model (interactvlm_amd/llava.py ClipTower, precision "f16"): IEEE fp16 MFMA operands - the
// qkv_w / out_w / fc1_w / fc2_w pointers of layers16_host are fp16 copies of the bf16 weights (biases and LayerNorm weights stay
// bf16); features_out: bf16 [B, tokens-1, 2*hidden] = [hi | lo] rows of the fp32 features (the mm_projector's split operand).
extern "C" int ivlm_clip_encode_f16(const ivlm_clip_cfg* c, const ivlm_clip_head* hd, const ivlm_clip_layer* layers16_host,
                                    const void* images, int B, void* features_split_out, void* workspace, size_t workspace_bytes,
                                    ivlm_stream_t stream) {
    return clip_encode(c, hd, layers16_host, images, B, features_split_out, workspace, workspace_bytes, stream, 1);
}

extern "C" size_t ivlm_sam_encode_workspace_bytes(const ivlm_sam_cfg* c, int V) {
    if (!c || V <= 0) return 0;
    const size_t g2 = (size_t)c->grid * c->grid, rows = (size_t)V * g2, D = c->embed_dim;
    const int nw = (c->grid + c->window - 1) / c->window;
    const size_t wrows = (size_t)V * nw * nw * c->window * c->window, qrows = std::max(rows, wrows);
    const size_t npad = (((size_t)2 * (2 * c->grid - 1)) + 7) / 8 * 8;
    size_t b = al(rows * 3 * c->patch * c->patch * 2) + al(rows * D * 4) + al(rows * D * 2) + al(qrows * 3 * D * 2) + al(qrows * D * 2) +
               al(rows * c->mlp_dim * 2);
    b += 2 * al((size_t)V * c->heads * g2 * c->grid * 4);          // rel_h / rel_w of a global block (the larger case)
    b += 2 * al(wrows * c->heads * c->window * 4);                  // ... of a windowed block
    b += al((size_t)c->heads * rows * npad * 2);                    // G of the rel-pos GEMM
    b += al(wrows * 4) + al(rows * 4);                              // part / unpart maps
    b += al(rows * c->out_chans * 2) * 2 + al(rows * 9 * c->out_chans * 2) + al(rows * D * 2);
    return b + 1024;
}

extern "C" int ivlm_sam_encode(const ivlm_sam_cfg* c, const ivlm_sam_head* hd, const ivlm_sam_block* blocks_host, const void* images,
                               int V, float* embeddings_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    if (!c || !hd || !blocks_host || !images || !embeddings_out || !workspace || V <= 0) return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_sam_encode_workspace_bytes(c, V)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    const int g = c->grid, D = c->embed_dim, H = c->heads, hdim = D / H, wsz = c->window, OC = c->out_chans;
    const int nw = (g + wsz - 1) / wsz, g2 = g * g, R = V * g2, nwin = V * nw * nw, WS = wsz * wsz, WR = nwin * WS;
    const int Kp = 3 * c->patch * c->patch;
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    bf16_t* cols = static_cast<bf16_t*>(cv.take((size_t)R * Kp * 2));
    float* x = static_cast<float*>(cv.take((size_t)R * D * 4));
    bf16_t* xn = static_cast<bf16_t*>(cv.take((size_t)R * D * 2));
    const size_t qrows = std::max(R, WR);
    bf16_t* qkv = static_cast<bf16_t*>(cv.take(qrows * 3 * D * 2));
    bf16_t* att = static_cast<bf16_t*>(cv.take(qrows * D * 2));
    bf16_t* hh = static_cast<bf16_t*>(cv.take((size_t)R * c->mlp_dim * 2));
    float* relh_g = static_cast<float*>(cv.take((size_t)V * H * g2 * g * 4));
    float* relw_g = static_cast<float*>(cv.take((size_t)V * H * g2 * g * 4));
    float* relh_w = static_cast<float*>(cv.take((size_t)WR * H * wsz * 4));
    float* relw_w = static_cast<float*>(cv.take((size_t)WR * H * wsz * 4));
    const int npad = (2 * (2 * g - 1) + 7) / 8 * 8;
    bf16_t* G = static_cast<bf16_t*>(cv.take((size_t)H * R * npad * 2));
    int32_t* part = static_cast<int32_t*>(cv.take((size_t)WR * 4));
    int32_t* unpart = static_cast<int32_t*>(cv.take((size_t)R * 4));
    bf16_t* n0 = static_cast<bf16_t*>(cv.take((size_t)R * OC * 2));
    bf16_t* n1 = static_cast<bf16_t*>(cv.take((size_t)R * OC * 2));
    bf16_t* c3 = static_cast<bf16_t*>(cv.take((size_t)R * 9 * OC * 2));
    bf16_t* xb16 = static_cast<bf16_t*>(cv.take((size_t)R * D * 2));
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    int rc;
    sam_window_maps_kernel<<<256, 256, 0, st>>>(V, g, wsz, nw, part, unpart);
    if ((rc = ivlm_launch_status())) return rc;
    if ((rc = im2col_nchw(static_cast<const bf16_t*>(images), cols, V, 3, c->img_size, c->img_size, c->patch, c->patch, Kp, st))) return rc;
    if ((rc = gemm(cols, Kp, hd->patch_w, Kp, x, 1, D, hd->patch_b, hd->pos_embed, 0, D, g2, R, D, Kp, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
    const float scale = 1.0f / sqrtf((float)hdim);
    for (int l = 0; l < c->depth; ++l) {
        const ivlm_sam_block& Bk = blocks_host[l];
        // rel_cat is READ by the attention kernels' table mode (windows: 64 rows; the 64 x 64 grid: 254 rows; zero-padded to a
        // multiple of 64 rows - ABI version 4): a NULL table would silently drop the rel-pos bias
        if (!Bk.rel_cat || !Bk.rel_h || !Bk.rel_w) return IVLM_ERR_INVALID_ARG;
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm1_w), static_cast<const bf16_t*>(Bk.norm1_b), xn, 0, R, D, 1e-6f, st))) return rc;
        const int side = Bk.global_attn ? g : wsz, S = side * side, nb = Bk.global_attn ? V : nwin;
        if (Bk.global_attn) {
            if ((rc = gemm(xn, D, Bk.qkv_w, D, qkv, 0, 3 * D, Bk.qkv_b, nullptr, 0, 0, 0, R, 3 * D, D, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
        } else {  // real rows only, scattered to their window positions; the padded positions get the bias
            if ((rc = gemm(xn, D, Bk.qkv_w, D, qkv, 0, 3 * D, Bk.qkv_b, nullptr, 0, 0, 0, R, 3 * D, D, ACT_NONE, unpart, nullptr, nullptr, 0, st))) return rc;
            fill_pad_rows_kernel<<<2048, 256, 0, st>>>(qkv, 3 * D, part, WR, static_cast<const bf16_t*>(Bk.qkv_b), 3 * D);
            if ((rc = ivlm_launch_status())) return rc;
        }
        float *rh, *rw;
        if (side == 64 && hdim == 80 && 2 * (2 * side - 1) <= npad) {
            // the 64 x 64 grid: TABLE MODE too (REL 5 of attn_kernel) - every 128-query block computes its terms from rel_cat
            // ([rel_pos_h (127 rows) ; rel_pos_w (127 rows)]) before its tile loop: no G GEMM, no gather
            rh = reinterpret_cast<float*>(const_cast<void*>(Bk.rel_cat));
            rw = nullptr;
        } else if (side >= 32) {  // rel-pos operands through one batched GEMM over the heads + Toeplitz gather
            rh = relh_g; rw = relw_g;
            const int M = nb * S;
            GemmArgs gg;
            gg.A = qkv; gg.lda = 3 * D; gg.W = static_cast<const bf16_t*>(Bk.rel_cat); gg.ldw = hdim; gg.C = G; gg.ldc = npad;
            gg.M = M; gg.N = npad; gg.K = hdim; gg.batch = H; gg.strideA = hdim; gg.strideW = 0; gg.strideC = (int64_t)M * npad;
            if ((rc = linear_bf16(gg, st))) return rc;
            if ((rc = ivlm_relpos_gather(G, (int64_t)M * npad, npad, nb, H, side, side, rh, rw, stream))) return rc;
        } else if (2 * side <= 32 && hdim == 80 && 2 * (2 * side - 1) <= 64) {
            // windows: TABLE MODE of the attention kernel - it computes the decomposed rel-pos terms itself from rel_cat
            // ([rel_pos_h ; rel_pos_w] zero-padded to 64 rows): no relpos pass, no [nb*H, S, 2 side] arrays
            rh = reinterpret_cast<float*>(const_cast<void*>(Bk.rel_cat));
            rw = nullptr;
        } else {
            rh = relh_w; rw = relw_w;
            if ((rc = relpos_bias(qkv, (int64_t)S * 3 * D, hdim, 3 * D, static_cast<const bf16_t*>(Bk.rel_h), static_cast<const bf16_t*>(Bk.rel_w),
                                  nb, H, side, side, hdim, rh, rw, st)))
                return rc;
        }
        const int64_t strides[12] = {(int64_t)S * 3 * D, hdim, 3 * D, (int64_t)S * 3 * D, hdim, 3 * D, (int64_t)S * 3 * D, hdim, 3 * D,
                                     (int64_t)S * D, hdim, D};
        if ((rc = ivlm_attention_bf16(qkv, qkv + D, qkv + 2 * D, att, strides, nb, H, S, S, hdim, scale, 0, 0, rh, rw, side, side, 1, 1, stream)))
            return rc;
        // proj (+ window_unpartition via the gather prologue) + shortcut, in place on the fp32 stream
        if ((rc = gemm(att, D, Bk.proj_w, D, x, 1, D, Bk.proj_b, x, 1, D, 0, R, D, D, ACT_NONE, nullptr, Bk.global_attn ? nullptr : unpart, nullptr, 0, st)))
            return rc;
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm2_w), static_cast<const bf16_t*>(Bk.norm2_b), xn, 0, R, D, 1e-6f, st))) return rc;
        if ((rc = gemm(xn, D, Bk.lin1_w, D, hh, 0, c->mlp_dim, Bk.lin1_b, nullptr, 0, 0, 0, R, c->mlp_dim, D, ACT_GELU, nullptr, nullptr, nullptr, 0, st))) return rc;
        if ((rc = gemm(hh, c->mlp_dim, Bk.lin2_w, c->mlp_dim, x, 1, D, Bk.lin2_b, x, 1, D, 0, R, D, c->mlp_dim, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
    }
    if ((rc = gather_rows(xb16, 0, D, x, 1, D, nullptr, nullptr, 0, 0, R, D, st))) return rc;
    if ((rc = gemm(xb16, D, hd->neck0_w, D, n0, 0, OC, nullptr, nullptr, 0, 0, 0, R, OC, D, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
    if ((rc = layernorm(n0, 0, static_cast<const bf16_t*>(hd->neck1_w), static_cast<const bf16_t*>(hd->neck1_b), n1, 0, R, OC, 1e-6f, st))) return rc;
    if ((rc = im2col3x3_nhwc(n1, c3, V, g, g, OC, st))) return rc;
    if ((rc = gemm(c3, 9 * OC, hd->neck2_w, 9 * OC, n0, 0, OC, nullptr, nullptr, 0, 0, 0, R, OC, 9 * OC, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
    return layernorm(n0, 0, static_cast<const bf16_t*>(hd->neck3_w), static_cast<const bf16_t*>(hd->neck3_b), embeddings_out, 1, R, OC, 1e-6f, st);
}

// =====================================================================================================================
// ivlm_sam_encode_parity: the same stage in "parity" precision - every activation that feeds an MFMA travels as [hi | lo] bf16
// rows (split LayerNorm outputs, split-operand / split-output GEMMs, split-operand attention with fp32 rel-pos terms, split
// neck) - the launch order of interactvlm_amd/sam.py SamImageEncoder._forward_parity (bit-identical to it).
// =====================================================================================================================
extern "C" size_t ivlm_sam_encode_parity_workspace_bytes(const ivlm_sam_cfg* c, int V) {
    if (!c || V <= 0) return 0;
    const size_t g2 = (size_t)c->grid * c->grid, rows = (size_t)V * g2, D = c->embed_dim;
    const int nw = (c->grid + c->window - 1) / c->window;
    const size_t wrows = (size_t)V * nw * nw * c->window * c->window, qrows = std::max(rows, wrows);
    size_t b = al(rows * 3 * c->patch * c->patch * 2) + al(rows * D * 4) + al(rows * 2 * D * 2) + al(qrows * 6 * D * 2) +
               al(qrows * 2 * D * 2) + al(rows * 2 * c->mlp_dim * 2);
    b += 2 * al((size_t)V * c->heads * g2 * c->grid * 4);          // rel_h / rel_w of a global block
    b += 2 * al(wrows * c->heads * c->window * 4);                  // ... of a windowed block
    b += al(wrows * 4) + al(rows * 4) + al(6 * D * 2);             // part / unpart maps, the [bias | 0] row of the padded positions
    b += al(rows * c->out_chans * 4) + al(rows * 2 * c->out_chans * 2) + al(rows * 18 * c->out_chans * 2);
    return b + 1024;
}

namespace ivlm {
namespace {
int sam_encode_parity(const ivlm_sam_cfg* c, const ivlm_sam_head* hd, const ivlm_sam_block* blocks_host, const ivlm_sam_mlp_f16* mlp16,
                      const void* images, int V, float* embeddings_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    if (!c || !hd || !blocks_host || !images || !embeddings_out || !workspace || V <= 0) return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_sam_encode_parity_workspace_bytes(c, V)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    const int g = c->grid, D = c->embed_dim, H = c->heads, hdim = D / H, wsz = c->window, OC = c->out_chans, MD = c->mlp_dim;
    const int nw = (g + wsz - 1) / wsz, g2 = g * g, R = V * g2, nwin = V * nw * nw, WS = wsz * wsz, WR = nwin * WS;
    const int Kp = 3 * c->patch * c->patch;
    if (hdim != 80) return IVLM_ERR_UNSUPPORTED;  // (the split attention / rel-pos kernels are built for SAM's head dim)
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    bf16_t* cols = static_cast<bf16_t*>(cv.take((size_t)R * Kp * 2));
    float* x = static_cast<float*>(cv.take((size_t)R * D * 4));
    bf16_t* xn = static_cast<bf16_t*>(cv.take((size_t)R * 2 * D * 2));
    const size_t qrows = std::max(R, WR);
    bf16_t* qkv = static_cast<bf16_t*>(cv.take(qrows * 6 * D * 2));
    bf16_t* att = static_cast<bf16_t*>(cv.take(qrows * 2 * D * 2));
    bf16_t* hh = static_cast<bf16_t*>(cv.take((size_t)R * 2 * MD * 2));
    float* relh_g = static_cast<float*>(cv.take((size_t)V * H * g2 * g * 4));
    float* relw_g = static_cast<float*>(cv.take((size_t)V * H * g2 * g * 4));
    float* relh_w = static_cast<float*>(cv.take((size_t)WR * H * wsz * 4));
    float* relw_w = static_cast<float*>(cv.take((size_t)WR * H * wsz * 4));
    int32_t* part = static_cast<int32_t*>(cv.take((size_t)WR * 4));
    int32_t* unpart = static_cast<int32_t*>(cv.take((size_t)R * 4));
    bf16_t* brow = static_cast<bf16_t*>(cv.take((size_t)6 * D * 2));
    float* n0 = static_cast<float*>(cv.take((size_t)R * OC * 4));
    bf16_t* n1 = static_cast<bf16_t*>(cv.take((size_t)R * 2 * OC * 2));
    bf16_t* c3 = static_cast<bf16_t*>(cv.take((size_t)R * 18 * OC * 2));
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    int rc;
    sam_window_maps_kernel<<<256, 256, 0, st>>>(V, g, wsz, nw, part, unpart);
    if ((rc = ivlm_launch_status())) return rc;
    if ((rc = im2col_nchw(static_cast<const bf16_t*>(images), cols, V, 3, c->img_size, c->img_size, c->patch, c->patch, Kp, st))) return rc;
    if ((rc = gemm(cols, Kp, hd->patch_w, Kp, x, 1, D, hd->patch_b, hd->pos_embed, 0, D, g2, R, D, Kp, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
    const float scale = 1.0f / sqrtf((float)hdim);
    for (int l = 0; l < c->depth; ++l) {
        const ivlm_sam_block& Bk = blocks_host[l];
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm1_w), static_cast<const bf16_t*>(Bk.norm1_b), xn, 2, R, D, 1e-6f, st))) return rc;
        const int side = Bk.global_attn ? g : wsz, S = side * side, nb = Bk.global_attn ? V : nwin;
        if (Bk.global_attn) {
            if ((rc = gemm(xn, 2 * D, Bk.qkv_w, D, qkv, 0, 6 * D, Bk.qkv_b, nullptr, 0, 0, 0, R, 3 * D, D, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 1, 1))) return rc;
        } else {  // real rows scattered to their window positions; the padded positions get [bias | 0]
            if ((rc = gemm(xn, 2 * D, Bk.qkv_w, D, qkv, 0, 6 * D, Bk.qkv_b, nullptr, 0, 0, 0, R, 3 * D, D, ACT_NONE, unpart, nullptr, nullptr, 0, st, 1, 1))) return rc;
            IVLM_HIP_TRY(hipMemsetAsync(brow, 0, (size_t)6 * D * 2, st));
            IVLM_HIP_TRY(hipMemcpyAsync(brow, Bk.qkv_b, (size_t)3 * D * 2, hipMemcpyDeviceToDevice, st));
            fill_pad_rows_kernel<<<2048, 256, 0, st>>>(qkv, 6 * D, part, WR, brow, 6 * D);
            if ((rc = ivlm_launch_status())) return rc;
        }
        float* rh = Bk.global_attn ? relh_g : relh_w;
        float* rw = Bk.global_attn ? relw_g : relw_w;
        // fp32 rel-pos terms from q = hi + lo (rows: [q k v hi | q k v lo], row stride 6 D)
        if ((rc = relpos_bias(qkv, (int64_t)S * 6 * D, hdim, 6 * D, static_cast<const bf16_t*>(Bk.rel_h), static_cast<const bf16_t*>(Bk.rel_w),
                              nb, H, side, side, hdim, rh, rw, st, qkv + 3 * D)))
            return rc;
        AttnArgs a{};
        a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D;
        a.q_lo = qkv + 3 * D; a.k_lo = qkv + 4 * D; a.v_lo = qkv + 5 * D;
        a.o = att; a.o_lo = att + D;
        a.q_bs = a.k_bs = a.v_bs = (int64_t)S * 6 * D;
        a.q_hs = a.k_hs = a.v_hs = hdim;
        a.q_rs = a.k_rs = a.v_rs = 6 * D;
        a.o_bs = (int64_t)S * 2 * D; a.o_hs = hdim; a.o_rs = 2 * D;
        a.B = nb; a.H = H; a.Sq = S; a.Sk = S; a.D = hdim;
        a.scale = scale; a.causal = 0; a.q_pos0 = 0;
        a.rel_h = rh; a.rel_w = rw; a.rel_kh = side; a.rel_kw = side;
        a.kv_batch_div = 1; a.prescale_q = 1;
        if ((rc = attention_bf16(a, st))) return rc;
        if ((rc = gemm(att, 2 * D, Bk.proj_w, D, x, 1, D, Bk.proj_b, x, 1, D, 0, R, D, D, ACT_NONE, nullptr, Bk.global_attn ? nullptr : unpart, nullptr, 0, st, 1, 0)))
            return rc;
        if (mlp16) {  // the MLP on fp16 operands: norm2 and the GELU epilogue write IEEE halves, one MFMA pass
            if (!mlp16[l].lin1_w16 || !mlp16[l].lin2_w16) return IVLM_ERR_INVALID_ARG;
            if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm2_w), static_cast<const bf16_t*>(Bk.norm2_b), xn, 4, R, D, 1e-6f, st))) return rc;
            if ((rc = gemm(xn, D, mlp16[l].lin1_w16, D, hh, 0, MD, Bk.lin1_b, nullptr, 0, 0, 0, R, MD, D, ACT_GELU, nullptr, nullptr, nullptr, 0, st, 0, 0, 1, 1))) return rc;
            if ((rc = gemm(hh, MD, mlp16[l].lin2_w16, MD, x, 1, D, Bk.lin2_b, x, 1, D, 0, R, D, MD, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 0, 0, 1, 0))) return rc;
            continue;
        }
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm2_w), static_cast<const bf16_t*>(Bk.norm2_b), xn, 2, R, D, 1e-6f, st))) return rc;
        if ((rc = gemm(xn, 2 * D, Bk.lin1_w, D, hh, 0, 2 * MD, Bk.lin1_b, nullptr, 0, 0, 0, R, MD, D, ACT_GELU, nullptr, nullptr, nullptr, 0, st, 1, 1))) return rc;
        if ((rc = gemm(hh, 2 * MD, Bk.lin2_w, MD, x, 1, D, Bk.lin2_b, x, 1, D, 0, R, D, MD, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 1, 0))) return rc;
    }
    // neck: 1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d on split operands
    if ((rc = gather_rows(xn, 2, 2 * D, x, 1, D, nullptr, nullptr, 0, 0, R, D, st))) return rc;
    if ((rc = gemm(xn, 2 * D, hd->neck0_w, D, n0, 1, OC, nullptr, nullptr, 0, 0, 0, R, OC, D, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 1, 0))) return rc;
    if ((rc = layernorm(n0, 1, static_cast<const bf16_t*>(hd->neck1_w), static_cast<const bf16_t*>(hd->neck1_b), n1, 2, R, OC, 1e-6f, st))) return rc;
    if ((rc = im2col3x3_nhwc(n1, c3, V, g, g, OC, st, 2 * OC, 18 * OC))) return rc;
    if ((rc = im2col3x3_nhwc(n1 + OC, c3 + 9 * OC, V, g, g, OC, st, 2 * OC, 18 * OC))) return rc;
    if ((rc = gemm(c3, 18 * OC, hd->neck2_w, 9 * OC, n0, 1, OC, nullptr, nullptr, 0, 0, 0, R, OC, 9 * OC, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 1, 0))) return rc;
    return layernorm(n0, 1, static_cast<const bf16_t*>(hd->neck3_w), static_cast<const bf16_t*>(hd->neck3_b), embeddings_out, 1, R, OC, 1e-6f, st);
}
}  // namespace
}  // namespace ivlm

extern "C" int ivlm_sam_encode_parity(const ivlm_sam_cfg* c, const ivlm_sam_head* hd, const ivlm_sam_block* blocks_host,
                                      const void* images, int V, float* embeddings_out, void* workspace, size_t workspace_bytes,
                                      ivlm_stream_t stream) {
    ivlm_enter();
    return sam_encode_parity(c, hd, blocks_host, nullptr, images, V, embeddings_out, workspace, workspace_bytes, stream);
}

// ... with the MLP of every block on fp16 operands (the "parity-encoder" mode of interactvlm_amd/model.py; bit-identical to
// SamImageEncoder._forward_parity with sites n1, attn, proj, f16mlp)
extern "C" int ivlm_sam_encode_parity_f16mlp(const ivlm_sam_cfg* c, const ivlm_sam_head* hd, const ivlm_sam_block* blocks_host,
                                             const ivlm_sam_mlp_f16* mlp16_host, const void* images, int V, float* embeddings_out,
                                             void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    if (!mlp16_host) return IVLM_ERR_INVALID_ARG;
    return sam_encode_parity(c, hd, blocks_host, mlp16_host, images, V, embeddings_out, workspace, workspace_bytes, stream);
}

// =====================================================================================================================
// ivlm_sam_encode_f16: the stage in the DEFAULT precision of the host model (interactvlm_amd/sam.py SamImageEncoder._forward_parity
// with SITES_F16Q, bit-identical to it): IEEE fp16 MFMA operands in one pass with the q path exact -
//   norm1 -> [hi | lo] halves; q = W_q . (hi + lo) as its own GEMM, written as [hi | lo] halves; k | v: one GEMM on the hi half;
//   attention on fp16 q / k / v with the lo half of q in the rel-pos table products (windows: whole-window kernel; the 64 x 64
//   grid: REL 5), fp32 rel-pos terms; proj / mlp1 / mlp2 on fp16 operands; the neck on hi + lo bf16 operands.
// blocks16_host[l]: fp16 copies of block l's four GEMM weights, of its q|k|v bias and of rel_cat (ivlm_bf16_to_f16).
// =====================================================================================================================
extern "C" size_t ivlm_sam_encode_f16_workspace_bytes(const ivlm_sam_cfg* c, int V) {
    if (!c || V <= 0) return 0;
    const size_t g2 = (size_t)c->grid * c->grid, rows = (size_t)V * g2, D = c->embed_dim;
    const int nw = (c->grid + c->window - 1) / c->window;
    const size_t wrows = (size_t)V * nw * nw * c->window * c->window, qrows = std::max(rows, wrows);
    size_t b = al(rows * 3 * c->patch * c->patch * 2) + al(rows * D * 4) + al(rows * 2 * D * 2) + 2 * al(qrows * 2 * D * 2) +
               al(qrows * D * 2) + al(rows * c->mlp_dim * 2);
    b += al(wrows * 4) + al(rows * 4) + al(2 * D * 2);            // part / unpart maps, the [q bias | 0] row of the padded positions
    b += al(rows * c->out_chans * 4) + al(rows * 2 * c->out_chans * 2) + al(rows * 18 * c->out_chans * 2);
    return b + 1024;
}

extern "C" int ivlm_sam_encode_f16(const ivlm_sam_cfg* c, const ivlm_sam_head* hd, const ivlm_sam_block* blocks_host,
                                   const ivlm_sam_block_f16* blocks16_host, const void* images, int V, float* embeddings_out,
                                   void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    ivlm_enter();
    if (!c || !hd || !blocks_host || !blocks16_host || !images || !embeddings_out || !workspace || V <= 0) return IVLM_ERR_INVALID_ARG;
    if (workspace_bytes < ivlm_sam_encode_f16_workspace_bytes(c, V)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    const int g = c->grid, D = c->embed_dim, H = c->heads, hdim = D / H, wsz = c->window, OC = c->out_chans, MD = c->mlp_dim;
    const int nw = (g + wsz - 1) / wsz, g2 = g * g, R = V * g2, nwin = V * nw * nw, WS = wsz * wsz, WR = nwin * WS;
    const int Kp = 3 * c->patch * c->patch;
    if (hdim != 80 || g != 64 || 2 * wsz > 32) return IVLM_ERR_UNSUPPORTED;  // (table-mode attention: SAM's head dim, grid and windows)
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    bf16_t* cols = static_cast<bf16_t*>(cv.take((size_t)R * Kp * 2));
    float* x = static_cast<float*>(cv.take((size_t)R * D * 4));
    bf16_t* xn = static_cast<bf16_t*>(cv.take((size_t)R * 2 * D * 2));
    const size_t qrows = std::max(R, WR);
    bf16_t* q2 = static_cast<bf16_t*>(cv.take(qrows * 2 * D * 2));   // [q hi | q lo]
    bf16_t* kv = static_cast<bf16_t*>(cv.take(qrows * 2 * D * 2));   // [k | v]
    bf16_t* att = static_cast<bf16_t*>(cv.take(qrows * D * 2));
    bf16_t* hh = static_cast<bf16_t*>(cv.take((size_t)R * MD * 2));
    int32_t* part = static_cast<int32_t*>(cv.take((size_t)WR * 4));
    int32_t* unpart = static_cast<int32_t*>(cv.take((size_t)R * 4));
    bf16_t* brow = static_cast<bf16_t*>(cv.take((size_t)2 * D * 2));
    float* n0 = static_cast<float*>(cv.take((size_t)R * OC * 4));
    bf16_t* n1 = static_cast<bf16_t*>(cv.take((size_t)R * 2 * OC * 2));
    bf16_t* c3 = static_cast<bf16_t*>(cv.take((size_t)R * 18 * OC * 2));
    if (!cv.ok) return IVLM_ERR_WORKSPACE;
    int rc;
    sam_window_maps_kernel<<<256, 256, 0, st>>>(V, g, wsz, nw, part, unpart);
    if ((rc = ivlm_launch_status())) return rc;
    if ((rc = im2col_nchw(static_cast<const bf16_t*>(images), cols, V, 3, c->img_size, c->img_size, c->patch, c->patch, Kp, st))) return rc;
    if ((rc = gemm(cols, Kp, hd->patch_w, Kp, x, 1, D, hd->patch_b, hd->pos_embed, 0, D, g2, R, D, Kp, ACT_NONE, nullptr, nullptr, nullptr, 0, st))) return rc;
    const float scale = 1.0f / sqrtf((float)hdim);
    for (int l = 0; l < c->depth; ++l) {
        const ivlm_sam_block& Bk = blocks_host[l];
        const ivlm_sam_block_f16& B16 = blocks16_host[l];
        if (!B16.qkv_w16 || !B16.proj_w16 || !B16.lin1_w16 || !B16.lin2_w16 || !B16.qkv_b16 || !B16.rel_cat16) return IVLM_ERR_INVALID_ARG;
        const bf16_t* wq = static_cast<const bf16_t*>(B16.qkv_w16);
        const bf16_t* bq = static_cast<const bf16_t*>(Bk.qkv_b);
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm1_w), static_cast<const bf16_t*>(Bk.norm1_b), xn, 5, R, D, 1e-6f, st))) return rc;
        const int side = Bk.global_attn ? g : wsz, S = side * side, nb = Bk.global_attn ? V : nwin;
        const int32_t* omap = Bk.global_attn ? nullptr : unpart;
        // q = W_q . (hi + lo) -> [hi | lo] halves;  k | v = W_kv . hi  (real rows only; windows: scattered to their window positions)
        if ((rc = gemm(xn, 2 * D, wq, D, q2, 0, 2 * D, bq, nullptr, 0, 0, 0, R, D, D, ACT_NONE, omap, nullptr, nullptr, 0, st, 1, 1, 1, 1))) return rc;
        if ((rc = gemm(xn, 2 * D, wq + (size_t)D * D, D, kv, 0, 2 * D, bq + D, nullptr, 0, 0, 0, R, 2 * D, D, ACT_NONE, omap, nullptr, nullptr, 0, st, 0, 0, 1, 1))) return rc;
        if (!Bk.global_attn) {  // the padded window positions: q = [bias | 0], k | v = bias
            IVLM_HIP_TRY(hipMemsetAsync(brow, 0, (size_t)2 * D * 2, st));
            IVLM_HIP_TRY(hipMemcpyAsync(brow, B16.qkv_b16, (size_t)D * 2, hipMemcpyDeviceToDevice, st));
            fill_pad_rows_kernel<<<2048, 256, 0, st>>>(q2, 2 * D, part, WR, brow, 2 * D);
            fill_pad_rows_kernel<<<2048, 256, 0, st>>>(kv, 2 * D, part, WR, static_cast<const bf16_t*>(B16.qkv_b16) + D, 2 * D);
            if ((rc = ivlm_launch_status())) return rc;
        }
        AttnArgs a{};
        a.f16 = 1;
        a.q = q2; a.q_lo = q2 + D; a.q_lo_level = 1;
        a.k = kv; a.v = kv + D; a.o = att;
        a.q_bs = a.k_bs = a.v_bs = (int64_t)S * 2 * D;
        a.q_hs = a.k_hs = a.v_hs = hdim;
        a.q_rs = a.k_rs = a.v_rs = 2 * D;
        a.o_bs = (int64_t)S * D; a.o_hs = hdim; a.o_rs = D;
        a.B = nb; a.H = H; a.Sq = S; a.Sk = S; a.D = hdim;
        a.scale = scale; a.causal = 0; a.q_pos0 = 0;
        a.rel_h = reinterpret_cast<const float*>(B16.rel_cat16); a.rel_w = nullptr; a.rel_kh = side; a.rel_kw = side;  // table mode
        a.kv_batch_div = 1; a.prescale_q = 1;
        if ((rc = attention_bf16(a, st))) return rc;
        if ((rc = gemm(att, D, B16.proj_w16, D, x, 1, D, Bk.proj_b, x, 1, D, 0, R, D, D, ACT_NONE, nullptr, omap, nullptr, 0, st, 0, 0, 1, 0))) return rc;
        if ((rc = layernorm(x, 1, static_cast<const bf16_t*>(Bk.norm2_w), static_cast<const bf16_t*>(Bk.norm2_b), xn, 4, R, D, 1e-6f, st))) return rc;
        if ((rc = gemm(xn, D, B16.lin1_w16, D, hh, 0, MD, Bk.lin1_b, nullptr, 0, 0, 0, R, MD, D, ACT_GELU, nullptr, nullptr, nullptr, 0, st, 0, 0, 1, 1))) return rc;
        if ((rc = gemm(hh, MD, B16.lin2_w16, MD, x, 1, D, Bk.lin2_b, x, 1, D, 0, R, D, MD, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 0, 0, 1, 0))) return rc;
    }
    // neck: 1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d on hi + lo bf16 operands (as the parity stage)
    if ((rc = gather_rows(xn, 2, 2 * D, x, 1, D, nullptr, nullptr, 0, 0, R, D, st))) return rc;
    if ((rc = gemm(xn, 2 * D, hd->neck0_w, D, n0, 1, OC, nullptr, nullptr, 0, 0, 0, R, OC, D, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 1, 0))) return rc;
    if ((rc = layernorm(n0, 1, static_cast<const bf16_t*>(hd->neck1_w), static_cast<const bf16_t*>(hd->neck1_b), n1, 2, R, OC, 1e-6f, st))) return rc;
    if ((rc = im2col3x3_nhwc(n1, c3, V, g, g, OC, st, 2 * OC, 18 * OC))) return rc;
    if ((rc = im2col3x3_nhwc(n1 + OC, c3 + 9 * OC, V, g, g, OC, st, 2 * OC, 18 * OC))) return rc;
    if ((rc = gemm(c3, 18 * OC, hd->neck2_w, 9 * OC, n0, 1, OC, nullptr, nullptr, 0, 0, 0, R, OC, 9 * OC, ACT_NONE, nullptr, nullptr, nullptr, 0, st, 1, 0))) return rc;
    return layernorm(n0, 1, static_cast<const bf16_t*>(hd->neck3_w), static_cast<const bf16_t*>(hd->neck3_b), embeddings_out, 1, R, OC, 1e-6f, st);
}

// =====================================================================================================================
// ivlm_sam_decode: PromptEncoder.forward(text_embeds) + MaskDecoder.forward(multimask_output=False) (prompt_encoder.py:140-186,
// mask_decoder.py:75-164, transformer.py:62-242) with fp32 activations end to end - the launch order of
// interactvlm_amd/sam.py SamMaskDecoder._forward.  Every linear takes [hi | lo] bf16 rows against [W | W] weights.
// =====================================================================================================================
namespace ivlm {
namespace {

struct DecCtx {
    hipStream_t st;
    ivlm_stream_t stream;
    Carver* cv;
    float* sk;
    size_t skb;
    int rc = 0;
    void* take(size_t b) {
        void* p = cv->take(b);
        if (!p && !rc) rc = IVLM_ERR_WORKSPACE;
        return p;
    }
    // fp32 rows -> [hi | lo] bf16 rows
    bf16_t* split(const float* x, int64_t rows, int cols) {
        bf16_t* o = static_cast<bf16_t*>(take((size_t)rows * 2 * cols * 2));
        if (o && !rc) rc = gather_rows(o, 2, 2 * cols, x, 1, cols, nullptr, nullptr, 0, 0, rows, cols, st);
        return o;
    }
    // (a + b[r % b_rows]) as fp32 rows or as split rows
    void* add(const float* a, const float* b, int64_t rows, int cols, int64_t b_rows, bool as_split) {
        void* o = take((size_t)rows * cols * 4);  // (split rows: 2 * cols bf16 = the same bytes)
        if (o && !rc) rc = add_rows(o, as_split ? 2 : 1, a, 1, b, 1, rows, cols, b_rows, st, 0);
        return o;
    }
    // act(x_split . [W|W]^T + bias) + residual -> fp32 [M, N]
    float* lin(const bf16_t* xs, const ivlm_lin& L, int M, int act, const float* res, float* dst = nullptr) {
        float* o = dst ? dst : static_cast<float*>(take((size_t)M * L.n * 4));
        if (o && !rc) rc = gemm(xs, 2 * L.k, L.w2, 2 * L.k, o, 1, L.n, L.b, res, 1, L.n, 0, M, L.n, 2 * L.k, act, nullptr, nullptr, sk, skb, st);
        return o;
    }
    float* norm(const float* x, const void* w, const void* b, int64_t rows, int cols, float eps, int gelu = 0) {
        float* o = static_cast<float*>(take((size_t)rows * cols * 4));
        if (o && !rc) rc = layernorm(x, 1, static_cast<const bf16_t*>(w), static_cast<const bf16_t*>(b), o, 1, rows, cols, eps, st, gelu);
        return o;
    }
    // Attention.forward (transformer.py:220-242) up to, not including, out_proj; returns the split rows of its output
    bf16_t* attn(const ivlm_dec_attn& a, const bf16_t* qs, const bf16_t* ks, const bf16_t* vs, int B, int Sq, int Sk, int heads) {
        float* q = lin(qs, a.q, B * Sq, ACT_NONE, nullptr);
        float* k = lin(ks, a.k, B * Sk, ACT_NONE, nullptr);
        float* v = lin(vs, a.v, B * Sk, ACT_NONE, nullptr);
        const int inner = a.q.n, d = inner / heads;
        float* o = static_cast<float*>(take((size_t)B * Sq * inner * 4));
        if (rc) return nullptr;
        const int64_t s12[12] = {(int64_t)Sq * inner, d, inner, (int64_t)Sk * inner, d, inner, (int64_t)Sk * inner, d, inner,
                                 (int64_t)Sq * inner, d, inner};
        rc = attention_f32(q, k, v, o, s12, B, heads, Sq, Sk, d, 1.0f / sqrtf((float)d), 1, st);
        return split(o, (int64_t)B * Sq, inner);
    }
};

}  // namespace
}  // namespace ivlm

extern "C" size_t ivlm_sam_decode_workspace_bytes(int V, int grid, int C, int n_text, int mlp_dim) {
    if (V <= 0 || grid <= 0 || C <= 0 || n_text <= 0) return 0;
    const size_t HW = (size_t)grid * grid, rows = (size_t)V * HW;
    // every intermediate has its own slot (one call = one carve, nothing is reused): ~40 image-sized fp32 / split buffers of
    // [V*HW, C], the upscaler's [V*HW*4, 128] pair, and the token-sized ones
    return 48 * al(rows * C * 4) + 3 * al(rows * 4 * 128 * 4) + 64 * al((size_t)V * (5 + n_text) * std::max(mlp_dim, 2 * C) * 4) +
           al((size_t)8 * V * (5 + n_text) * std::max(mlp_dim, C) * 4) + (1 << 20);
}

extern "C" int ivlm_sam_decode(const ivlm_sam_dec* w, int V, int grid, int n_text, const float* image_embeddings,
                               const float* text_embeds, float* low_res_out, float* iou_out, void* workspace, size_t workspace_bytes,
                               ivlm_stream_t stream) {
    ivlm_enter();
    if (!w || !image_embeddings || !text_embeds || !low_res_out || !iou_out || !workspace || V <= 0 || grid <= 0 || n_text <= 0 ||
        w->depth <= 0 || w->depth > 4)
        return IVLM_ERR_INVALID_ARG;
    const int C = w->C, HW = grid * grid, Nt = 5 + n_text, heads = w->heads;
    if (workspace_bytes < ivlm_sam_decode_workspace_bytes(V, grid, C, n_text, w->layers[0].lin1.n)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    DecCtx x;
    x.st = st; x.stream = stream; x.cv = &cv;
    x.skb = (size_t)8 * V * Nt * std::max(w->layers[0].lin1.n, C) * 4;
    x.sk = static_cast<float*>(x.take(x.skb));
    if (x.sk) {
        if (int rc0 = sk_counters_zero(x.sk, x.skb, st)) return rc0;
    }
    // tokens = [iou token ; mask tokens ; text embeds], the same set for every view
    float* tokens = static_cast<float*>(x.take((size_t)Nt * C * 4));
    float* query_pe = static_cast<float*>(x.take((size_t)V * Nt * C * 4));
    if (x.rc) return x.rc;
    int rc;
    if ((rc = gather_rows(tokens, 1, C, w->out_tokens, 1, C, nullptr, nullptr, 0, 0, 5, C, st))) return rc;
    if ((rc = gather_rows(tokens + 5 * C, 1, C, text_embeds, 1, C, nullptr, nullptr, 0, 0, n_text, C, st))) return rc;
    for (int v = 0; v < V; ++v)
        if ((rc = gather_rows(query_pe + (size_t)v * Nt * C, 1, C, tokens, 1, C, nullptr, nullptr, 0, 0, Nt, C, st))) return rc;
    const float* queries = query_pe;
    const float* keys = static_cast<const float*>(x.add(image_embeddings, static_cast<const float*>(w->no_mask), (int64_t)V * HW, C, 1, false));
    const float* key_pe = static_cast<const float*>(w->key_pe);
    const bf16_t* k_split = nullptr;
    for (int li = 0; li < w->depth && !x.rc; ++li) {
        const ivlm_dec_layer& L = w->layers[li];
        if (li == 0) {  // skip_first_layer_pe: queries = self_attn(q = k = v = queries), no residual
            const bf16_t* qs = x.split(queries, (int64_t)V * Nt, C);
            queries = x.lin(x.attn(L.self_attn, qs, qs, qs, V, Nt, Nt, heads), L.self_attn.o, V * Nt, ACT_NONE, nullptr);
        } else {
            const bf16_t* q = static_cast<const bf16_t*>(x.add(queries, query_pe, (int64_t)V * Nt, C, (int64_t)V * Nt, true));
            const bf16_t* sa = x.attn(L.self_attn, q, q, x.split(queries, (int64_t)V * Nt, C), V, Nt, Nt, heads);
            queries = x.lin(sa, L.self_attn.o, V * Nt, ACT_NONE, queries);
        }
        queries = x.norm(queries, L.norm1_w, L.norm1_b, (int64_t)V * Nt, C, 1e-5f);
        const bf16_t* q = static_cast<const bf16_t*>(x.add(queries, query_pe, (int64_t)V * Nt, C, (int64_t)V * Nt, true));
        k_split = static_cast<const bf16_t*>(x.add(keys, key_pe, (int64_t)V * HW, C, HW, true));
        const bf16_t* ca = x.attn(L.t2i, q, k_split, x.split(keys, (int64_t)V * HW, C), V, Nt, HW, heads);
        queries = x.norm(x.lin(ca, L.t2i.o, V * Nt, ACT_NONE, queries), L.norm2_w, L.norm2_b, (int64_t)V * Nt, C, 1e-5f);
        const float* h1 = x.lin(x.split(queries, (int64_t)V * Nt, C), L.lin1, V * Nt, ACT_RELU, nullptr);
        const float* mlp = x.lin(x.split(h1, (int64_t)V * Nt, L.lin1.n), L.lin2, V * Nt, ACT_NONE, queries);
        queries = x.norm(mlp, L.norm3_w, L.norm3_b, (int64_t)V * Nt, C, 1e-5f);
        q = static_cast<const bf16_t*>(x.add(queries, query_pe, (int64_t)V * Nt, C, (int64_t)V * Nt, true));
        const bf16_t* ia = x.attn(L.i2t, k_split, q, x.split(queries, (int64_t)V * Nt, C), V, HW, Nt, heads);  // image attends to tokens
        keys = x.norm(x.lin(ia, L.i2t.o, V * HW, ACT_NONE, keys), L.norm4_w, L.norm4_b, (int64_t)V * HW, C, 1e-5f);
    }
    if (x.rc) return x.rc;
    const bf16_t* q = static_cast<const bf16_t*>(x.add(queries, query_pe, (int64_t)V * Nt, C, (int64_t)V * Nt, true));
    const bf16_t* k = static_cast<const bf16_t*>(x.add(keys, key_pe, (int64_t)V * HW, C, HW, true));
    const bf16_t* fa = x.attn(w->final_attn, q, k, x.split(keys, (int64_t)V * HW, C), V, Nt, HW, heads);
    const float* hs = x.norm(x.lin(fa, w->final_attn.o, V * Nt, ACT_NONE, queries), w->norm_final_w, w->norm_final_b, (int64_t)V * Nt, C, 1e-5f);
    float* iou_tok = static_cast<float*>(x.take((size_t)V * C * 4));
    float* mask_tok = static_cast<float*>(x.take((size_t)V * C * 4));
    if (x.rc) return x.rc;
    if ((rc = gather_rows(iou_tok, 1, C, hs, 1, (int64_t)Nt * C, nullptr, nullptr, 0, 0, V, C, st))) return rc;       // hs[:, 0, :]
    if ((rc = gather_rows(mask_tok, 1, C, hs + C, 1, (int64_t)Nt * C, nullptr, nullptr, 0, 0, V, C, st))) return rc;  // hs[:, 1, :]
    // output_upscaling: ConvT(C -> C/4) -> LayerNorm2d -> GELU -> ConvT(C/4 -> C/8) -> GELU, as GEMMs on pixels
    const float* u = x.lin(x.split(keys, (int64_t)V * HW, C), w->up0, V * HW, ACT_NONE, nullptr);  // [V*HW, (dy,dx,C/4)]
    const int cm = w->up0.n / 4;
    const float* un = x.norm(u, w->up_ln_w, w->up_ln_b, (int64_t)V * HW * 4, cm, 1e-6f, 1);
    const float* u2 = x.lin(x.split(un, (int64_t)V * HW * 4, cm), w->up1, V * HW * 4, ACT_GELU, nullptr);  // [V*HW*4, (dy2,dx2,C/8)]
    const float* h0 = x.lin(x.split(mask_tok, V, C), w->hyper[0], V, ACT_RELU, nullptr);
    const float* h1 = x.lin(x.split(h0, V, w->hyper[0].n), w->hyper[1], V, ACT_RELU, nullptr);
    const float* h2 = x.lin(x.split(h1, V, w->hyper[1].n), w->hyper[2], V, ACT_NONE, nullptr);  // [V, C/8]
    if (x.rc) return x.rc;
    if ((rc = mask_dot(u2, h2, 1, low_res_out, V, grid, grid, w->hyper[2].n, st))) return rc;
    const float* i0 = x.lin(x.split(iou_tok, V, C), w->iou[0], V, ACT_RELU, nullptr);
    const float* i1 = x.lin(x.split(i0, V, w->iou[0].n), w->iou[1], V, ACT_RELU, nullptr);
    x.lin(x.split(i1, V, w->iou[1].n), w->iou[2], V, ACT_NONE, nullptr, iou_out);  // [V, n_mask]: column 0 is the kept mask's IoU
    return x.rc;
}
