// Internal C++ launch API shared by the C-ABI wrappers and the stage runners (not installed).
#pragma once
#include "ivlm_common.h"

namespace ivlm {

// ---- epilogue activation codes (also exposed through the C ABI) --------------------------------
enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_QUICK_GELU = 2, ACT_RELU = 3, ACT_SILU = 4, ACT_SWIGLU = 5, ACT_SIGMOID = 6 };

// Fused split-K (gemm_bf16_splitk with a counter array): blockIdx.z = K slice.  Every block stores its fp32 partial tile to the
// workspace, then counts its arrival on the tile's counter; the block that arrives LAST sums the slices of that tile in slice order
// (the arithmetic of splitk_reduce_kernel: bit-identical) and applies this epilogue - no reduction launch, no second pass over the
// whole product.  `count` != null switches it on; the counters are zero on entry and left at zero.
struct SplitKFused {
    int32_t* count = nullptr;  // [tiles of one slice]
    void* C = nullptr;         // the GEMM's real output and epilogue (the kernel's own C / bias / ... describe the partials)
    const bf16_t* bias = nullptr;
    const bf16_t* residual = nullptr;
    int64_t ldc = 0, ldr = 0, c_lo = 0;
    int res_mod = 0, act = 0, out_f32 = 0, out_f16 = 0, out_split = 0, res_f32 = 0;
};
constexpr int kSplitKCounters = 4096;  // counters a fused split-K launch may use (tiles per K slice)

struct GemmArgs {
    const bf16_t* A = nullptr;  // activations [M,K], row stride lda (elements); float when a_f32 (skinny paths only)
    const bf16_t* W = nullptr;  // weights     [N,K], row stride ldw  (nn.Linear layout)
    void* C = nullptr;          // output      [M,N] (or [M,N/2] for ACT_SWIGLU), row stride ldc
    const bf16_t* bias = nullptr;      // [N] or null
    const bf16_t* residual = nullptr;  // [*,N] added after the activation, row stride ldr, or null; float when res_f32
    int64_t lda = 0, ldw = 0, ldc = 0, ldr = 0;
    int res_mod = 0;   // >0: residual row = m % res_mod (broadcast tables such as pos_embed)
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    int out_f32 = 0;   // 1: C is float
    // scatter epilogue (tile GEMM only): row m of the product is written to row out_rows[m] of C (and takes its residual from
    // that row); rows with out_rows[m] < 0 are dropped.  SAM's window_unpartition + shortcut without a separate pass.
    const int32_t* out_rows = nullptr;
    // gather prologue (tile GEMM only): row m of the product reads row a_rows[m] of A (all entries valid).  SAM's proj GEMM of
    // a windowed block runs on the 16384 real rows only, reading them from their window positions.
    const int32_t* a_rows = nullptr;
    // K-panel operand layouts (tile GEMM only, bf16, K % 64 == 0): the matrix is stored as K/64 panels of [rows][64] elements,
    // element (r, k) at (k / 64) * kstep + r * 64 + (k % 64), kstep >= rows * 64.  One wave DMA instruction (8 rows x 128 B of a
    // K tile) then reads 1 KB CONTIGUOUS instead of eight 128-byte lines one row stride apart - measured 78 instead of 52
    // GB/s of L2-hit feed per CU (tools/experiments/exp_l2_feed.hip), the ceiling of the 256 x 256 kernel.  0 = row-major.
    int64_t a_kstep = 0, w_kstep = 0;
    // c_panel != 0: bf16 C written in the same K-panel layout (for the GEMM that consumes it): (m, n) at
    // (n / 64) * c_panel + m * 64 + (n % 64); ldc is ignored.
    int64_t c_panel = 0;
    // fp8 (OCP e4m3) operands on the MX matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales): A and W are
    // BYTE matrices [M,K] / [N,K]; K, lda, ldw are then given in 2-byte units (K_fp8 / 2) so that tiles, DMA and LDS images
    // are byte-identical to the bf16 path.  The fp32 accumulator is multiplied by *scale_a * *scale_w (per-tensor scales in
    // device memory) before the epilogue.  out_fp8: C is e4m3, written as act(...) / *scale_out.
    int fp8 = 0, out_fp8 = 0;
    const float *scale_a = nullptr, *scale_w = nullptr, *scale_out = nullptr;
    int a_f32 = 0;     // 1: A is float (M <= 16 weight-streaming paths: the products are exact, no operand rounding)
    // fp32-activation GEMM on the bf16 matrix cores ("parity" precision; tile GEMM path): A rows are [hi(K) | lo(K)] bf16 with
    // x = hi + lo to 2^-17 (ivlm_gather_rows / norm / GEMM outputs of kind IVLM_BF16_SPLIT), the lo half a_lo elements after the
    // hi half; K counts the columns of ONE half (= the columns of W).  The K loop visits every W tile twice in a row (against the
    // hi tile, then the lo tile of A): 2x the MFMA work, the second W read hits L2.
    int a_split = 0;
    int64_t a_lo = 0;
    // out_split: C is bf16, the fp32 result v of column n is written as hi = bf16(v) at column n and lo = bf16(v - hi) at column
    // n + c_lo (the next GEMM's / attention's split operand).  Set together with out_f32 = 1 (fp32 epilogue arithmetic).
    int out_split = 0;
    int64_t c_lo = 0;
    // fp16 operands (tile GEMM path): A and W hold IEEE half values (11 significant bits: an activation rounded to fp16 carries an
    // eighth of the bf16 rounding error at the same MFMA rate; bf16 weights convert exactly when they are inside the fp16 range).
    // out_f16: a 16-bit output (out_f32 = 0) is written as fp16 instead of bf16.
    int f16 = 0, out_f16 = 0;
    int res_f32 = 0;   // 1: residual is float (fp32 residual stream)
    // batched (strided) variant: blockIdx.z = batch
    int batch = 1;
    int64_t strideA = 0, strideW = 0, strideC = 0, strideR = 0;
    int tile = 0;  // 0: choose, 128 / 256: force the block tile (benchmarks, tests)
    // GEMV path only: fuse the preceding RMSNorm, out = W . (x * rsqrt(mean(x^2)+eps) * rms_w)
    const bf16_t* rms_w = nullptr;
    float rms_eps = 0.0f;
    SplitKFused sk;  // (set by gemm_bf16_splitk only)
};

// bf16 x bf16 -> fp32-accumulate MFMA GEMM with fused bias/activation/residual epilogue.
int gemm_bf16(const GemmArgs& g, hipStream_t st);
// 256x256 tile with the 8-phase ping-pong K loop (gemm256.hip); same contract as gemm_bf16
int gemm_bf16_256p(const GemmArgs& g, hipStream_t st);
int gemm_bf16_320p(const GemmArgs& g, hipStream_t st);  // 256 x 320 tiles (gemm320.hip)
// split-K variant for small M (fp32 partials in `workspace`, >= splits*M*N*4 bytes); act != SWIGLU.  counters != null
// (kSplitKCounters zeroed int32, left at zero): the reduction is fused into the GEMM launch (SplitKFused), else a second launch
int gemm_bf16_splitk(const GemmArgs& g, int splits, float* workspace, size_t ws_bytes, hipStream_t st, int32_t* counters = nullptr);
// nn.Linear dispatch: GEMV (M <= 8) or the MFMA tile kernel
int linear_bf16(const GemmArgs& g, hipStream_t st);

// ---- normalisation ---------------------------------------------------------------------------
// y = (x-mean)/sqrt(var+eps)*w+b over the last dim (rows x cols); x / y bf16 or fp32, fp32 statistics.
// out_rows != null: row r is written to row out_rows[r] of y (SAM's window_partition folded into norm1)
int layernorm(const void* x, int x_f32, const bf16_t* w, const bf16_t* b, void* y, int y_f32, int64_t rows, int cols, float eps,
              hipStream_t st, int gelu = 0, const int32_t* out_rows = nullptr, const float* fp8_scale = nullptr);
// y = x * rsqrt(mean(x^2)+eps) * w  (LLaMA RMSNorm; fp32 statistics; a bf16 input is cast back before the weight multiply as HF does)
int rmsnorm(const void* x, int x_f32, const bf16_t* w, void* y, int y_f32, int64_t rows, int cols, float eps, hipStream_t st,
            const float* fp8_scale = nullptr);

// ---- attention ---------------------------------------------------------------------------------
struct AttnArgs {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int64_t q_bs, q_hs, q_rs;  // batch / head / row strides in elements
    int64_t k_bs, k_hs, k_rs;
    int64_t v_bs, v_hs, v_rs;
    int64_t o_bs, o_hs, o_rs;
    int B, H, Sq, Sk, D;
    float scale;
    int causal;    // key j visible to query i iff j <= i + q_pos0
    int q_pos0;
    const float* rel_h;  // [B*H, Sq, rel_kh] or null
    const float* rel_w;  // [B*H, Sq, rel_kw]
    int rel_kh, rel_kw;
    int kv_batch_div;    // key/value batch index = b / kv_batch_div (broadcast K/V over query batches)
    int prescale_q;      // 1: scores = bf16(q*scale).k (SAM, HF-CLIP); 0: scores = (q.k)*scale (HF-LLaMA)
    // "parity" precision: lo planes of q / k / v / o (x = hi + lo; same strides as the hi tensors).  All four set or none.
    const bf16_t *q_lo = nullptr, *k_lo = nullptr, *v_lo = nullptr;
    bf16_t* o_lo = nullptr;
    int xcd_map = 1;     // (set by attention_bf16 from the A/B hook)
    int q_lo_level = 0;  // fp16 with q_lo: 1 = the lo half of q enters the rel-pos terms only, 2 = also Q.K^T (+ split softmax weights)
    int f16 = 0;  // 1: q / k / v / o (and the table-mode rel-pos table) are IEEE fp16: same tiles, the f16 matrix instruction
};

// softmax(scale * Q.K^T (+ rel-pos bias) (+ causal mask)) . V ; bf16 in/out, fp32 softmax. D in {16,32,64,80,128}.
int attention_bf16(const AttnArgs& a, hipStream_t st);
// fp32 operands end to end (SAM mask decoder): D in {16, 32}; strides[12] as the C ABI documents
int attention_f32(const float* q, const float* k, const float* v, float* o, const int64_t* st12, int B, int H, int Sq, int Sk,
                  int D, float scale, int kv_div, hipStream_t st);
// decomposed relative-position bias terms of SAM's ViT (image_encoder.py:354-392) as fp32 tables
int relpos_bias(const bf16_t* q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const bf16_t* tab_h, const bf16_t* tab_w,
                int B, int H, int SH, int SW, int D, float* rel_h, float* rel_w, hipStream_t st, const bf16_t* q_lo = nullptr);

// ---- data movement / elementwise (elementwise.hip) -----------------------------------------------
int im2col_nchw(const bf16_t* x, bf16_t* out, int B, int C, int H, int W, int ks, int stride, int Kpad, hipStream_t st);
int im2col3x3_nhwc(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, hipStream_t st, int64_t ldx = 0, int64_t ldo = 0);
int rope_kv_split(bf16_t* qkv, int64_t ld, int T, int H, int D, int pos0, bf16_t* kcache, bf16_t* kcache_lo, bf16_t* vcache,
                  bf16_t* vcache_lo, const float* cos_tab, const float* sin_tab, hipStream_t st);
// kinds: 0 bf16, 1 fp32, 2 (outputs only) split [hi | lo] bf16 rows of width 2*cols
// (kind 3, outputs only: e4m3 bytes of x / *scale)
int gather_rows(void* dst, int dst_kind, int64_t ldd, const void* src, int src_kind, int64_t lds_, const int32_t* idx,
                const void* add, int add_kind, int64_t lda, int64_t rows, int cols, hipStream_t st, const float* scale = nullptr);
int amax(const void* x, int kind, int64_t n, float* out, hipStream_t st);
int add_rows(void* out, int out_kind, const void* a, int a_kind, const void* b, int b_kind, int64_t rows, int cols,
             int64_t b_rows, hipStream_t st, int op = 0);  // op 0: a + b, 1: a * b
int fill_rows(bf16_t* dst, int64_t ldd, const int32_t* idx, int64_t n_idx, const bf16_t* row, int cols, hipStream_t st);
int dense_pe(const float* gauss, void* pe, int pe_f32, int h, int w, int F, hipStream_t st);
int rope_kv(bf16_t* qkv, int64_t ld, int T, int H, int D, int pos0, float theta, bf16_t* kcache, bf16_t* vcache,
            hipStream_t st, const float* cos_tab = nullptr, const float* sin_tab = nullptr, int f16 = 0);
int rope_table(float* cos_tab, float* sin_tab, int T, int D, float theta, hipStream_t st);
int normalize_pad_u8(const uint8_t* src, int H, int W, int y0, int x0, int ch, int cw, const float* mean3,
                     const float* std3, void* out, int out_bf16, int OH, int OW, hipStream_t st);
int mask_dot(const void* up, const void* hyper, int kind, float* low, int B, int gh, int gw, int C, hipStream_t st);

// ---- metrics / SMPL-X transfer (metrics.hip) ------------------------------------------------------------
int contact_prf(const float* gt, const float* pred, int B, int n, float thr, float* out, hipStream_t st);
int spmv_csr(const int32_t* row_ptr, const int32_t* col, const float* val, const float* x, int B, int rows, int cols,
             float* y, hipStream_t st);

// ---- rasterisation (raster.hip) -------------------------------------------------------------------
size_t raster_workspace_bytes(int n_prims_verts, int H, int W);
int rasterize_mesh(const float* verts, int nv, const int32_t* faces, int nf, const float* cam12_host, float fov_deg,
                   int H, int W, int32_t* p2v, float* bary, int32_t* pix_to_face, void* ws, size_t ws_bytes,
                   hipStream_t st);
int rasterize_points(const float* pts, int np, const float* cam12_host, float fov_deg, float radius, int H, int W,
                     int32_t* map, void* ws, size_t ws_bytes, hipStream_t st);

// ---- shaded colour renders (shade.hip) --------------------------------------------------------------
int phong_shade(const int32_t* p2v, const float* bary, const float* verts, const float* normals, const float* colors, int npix,
                const float* light3_host, const float* cam3_host, float ambient, float diffuse, float specular, float shininess,
                const float* bg3_host, uint8_t* out, hipStream_t st);

// ---- single-token decode (decode.hip) ---------------------------------------------------------------
// qkv / o bf16 or fp32 (io_f32); tmax = rows of the cache slab (a position >= tmax is skipped, never appended)
int llama_decode_attn(const void* qkv, int io_f32, bf16_t* kcache, bf16_t* vcache, int tmax, void* o, int H, int D, int pos,
                      float theta, float scale, hipStream_t st, const float* cos_tab = nullptr, const float* sin_tab = nullptr,
                      const int32_t* pos_dev = nullptr, bf16_t* kcache_lo = nullptr, bf16_t* vcache_lo = nullptr, int cache_f16 = 0);
int llama_decode_attn_batch(const void* qkv, int io_f32, int64_t ldq, bf16_t* kcache, bf16_t* vcache, int64_t cache_stride,
                            int tmax, void* o, int64_t ldo, int B, int H, int D, const int32_t* pos_dev, float theta, float scale,
                            const float* cos_tab, const float* sin_tab, hipStream_t st, bf16_t* kcache_lo = nullptr,
                            bf16_t* vcache_lo = nullptr, int cache_f16 = 0);

// split-KV variant (fp32 qkv / o, no lo planes): grid H x S, scratch zeroed once by the caller (decode.hip)
size_t llama_decode_attn_splitkv_scratch_bytes(int H, int D);
int llama_decode_attn_splitkv(const float* qkv, bf16_t* kcache, bf16_t* vcache, int tmax, float* o, int H, int D, int pos, float theta,
                              float scale, hipStream_t st, const float* cos_tab, const float* sin_tab, const int32_t* pos_dev,
                              int cache_f16, void* scratch, size_t scratch_bytes);

int llama_decode_attn_parts(const float* qkv, bf16_t* kcache, bf16_t* vcache, int tmax, float* parts, int H, int D, int pos, float theta,
                            float scale, hipStream_t st, const float* cos_tab, const float* sin_tab, const int32_t* pos_dev,
                            int cache_f16);

// fused decode attention + o_proj (decode_fused.hip)
int llama_attn_oproj(const float* qkv, bf16_t* kcache, bf16_t* vcache, int tmax, float* attn_scratch, const bf16_t* wo,
                     const float* x, float* x_out, int H, int D, float theta, float scale, const float* cos_tab,
                     const float* sin_tab, const int32_t* pos_dev, const int32_t* step_dev, int32_t* counter, int32_t* status,
                     hipStream_t st);
int gemm_splitk_choice(int M, int N, int K, int act, int has_rms);

// ---- skinny GEMM (gemv.hip): M <= 8 rows of activations against streamed weights -------------------
int gemv_bf16(const GemmArgs& g, hipStream_t st);
// batch-1 GEMV with e4m3 weight bytes (per-tensor scale g.scale_w), fp32 x (gemv.hip)
int gemv1_fp8w(const GemmArgs& g, hipStream_t st);
// skinny GEMM on MFMA (gemv_mfma.hip): M <= 16 activation rows, split-K inside the block, RMSNorm prologue optional
int gemv_mfma_bf16(const GemmArgs& g, hipStream_t st);
int argmax_f32(const float* x, int rows, int cols, int32_t* out, hipStream_t st, int32_t* bump = nullptr);

}  // namespace ivlm
