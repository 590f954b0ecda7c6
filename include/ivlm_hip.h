/* ivlm_hip.h — C ABI of libivlm_hip.so: the MI355X (gfx950) implementation of InteractVLM's
 * contact-inference hot path.
 *
 * The reference (saidwivedi/InteractVLM) is 100 % Python and has no FFI of its own; each entry
 * point below names the reference operator it replaces (file:line in /root/reference) and is
 * what a ctypes/cffi binding on the reference side would bind.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; every pointer is a DEVICE pointer unless the
 *     name ends in _host; all buffers (outputs, workspaces) are owned and allocated by the caller
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work
 *   - return value: 0 = IVLM_OK, negative = error (ivlm_error_string()); nothing throws
 *   - row-major contiguous tensors; shapes in comments use the reference's names:
 *       B images, V views, HW = H*W pixels, Nv mesh vertices, Np points
 *   - dtype codes: IVLM_F32 = 0, IVLM_BF16 = 1
 */
#ifndef IVLM_HIP_H
#define IVLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVLM_OK 0
#define IVLM_ERR_INVALID_ARG (-1)
#define IVLM_ERR_WORKSPACE (-2)
#define IVLM_ERR_LAUNCH (-3)
#define IVLM_ERR_UNSUPPORTED (-4)

#define IVLM_F32 0
#define IVLM_BF16 1

typedef void *ivlm_stream_t;

/* library identity --------------------------------------------------------------------------- */
int ivlm_abi_version(void);
const char *ivlm_error_string(int code);
const char *ivlm_build_arch(void); /* "gfx950" */
/* detail of the most recent IVLM_ERR_LAUNCH on this thread: HIP error text + source location */
const char *ivlm_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Render-Localize-Lift: 2D multi-view masks -> per-vertex / per-point contact
 * ------------------------------------------------------------------------------------------- */

/* One-time inversion of constant pixel->vertex tables into a vertex-major CSR ("lift plan").
 * Replaces the per-call table handling of HumanContact3DPredictor.__init__/_process_view
 * (model/components.py:203-218, 253-262): ids outside [0,Nv) in ANY slot drop the whole pixel.
 *   vid   i32 [V,HW,3]   (the reference's int64 table narrowed once by the host)
 *   bary  f32 [V,HW,3]
 *   row_ptr  i32 [V*Nv+1]      out: CSR row starts, row r = v*Nv + vertex
 *   ent_pix  i32 [cap]         out: pixel index (within the view) of each entry
 *   ent_w    f32 [cap]         out: barycentric weight of each entry
 *   cap >= 3*V*HW is always sufficient; *nnz_out (device i32) receives the entry count.
 * Entries of a row are ordered by (slot k, pixel) — the reference's summation order.
 */
size_t ivlm_lift_plan_workspace_bytes(int V, int64_t HW, int Nv);
int ivlm_lift_plan_build(const int32_t *vid, const float *bary, int V, int64_t HW, int Nv,
                         int32_t *row_ptr, int32_t *ent_pix, float *ent_w, int64_t cap, int32_t *nnz_out,
                         void *workspace, size_t workspace_bytes, ivlm_stream_t stream);

/* HumanContact3DPredictor.forward (model/components.py:220-277), deterministic vertex-major
 * gather over a lift plan:  m = sigmoid(clamp(logit,+-clampv)); per view votes/cnt; mean over
 * views with cnt>0; clamp [0,1].
 *   logits f32 [B,V,HW]  ->  out f32 [B,Nv];  nviews f32 [B,Nv] (may be NULL)
 * mode 0 = soft (above); mode 1 = thresholded object-mesh rule (components.py:445-489):
 *   p = sigmoid(logit) (no clamp), only pixels with p > param vote, no final clamp.
 */
int ivlm_lift_mesh_plan(const float *logits, const int32_t *row_ptr, const int32_t *ent_pix,
                        const float *ent_w, int B, int V, int64_t HW, int Nv, int mode, float param,
                        float *out, float *nviews, ivlm_stream_t stream);

/* Fused postprocess + lift: identical result to ivlm_postprocess_masks followed by ivlm_lift_mesh_plan,
 * but the logit of each table entry is evaluated on the fly from the low-res masks (low f32|bf16 [B,V,lh,lw]) with
 * the arithmetic of Sam.postprocess_masks; the plan must have been built for (oh, ow) pixel maps. */
int ivlm_lift_mesh_plan_lowres(const void *low, int dtype, int lh, int lw, int img, int in_h, int in_w, int oh, int ow,
                               const int32_t *row_ptr, const int32_t *ent_pix, const float *ent_w, int B, int V,
                               int Nv, int mode, float param, float *out, float *nviews, ivlm_stream_t stream);

/* Same operators, streaming directly over the dense tables (single-use tables, e.g. a fresh
 * lift2d_dict.pkl: ObjectMeshContact3DPredictor.forward_inference, components.py:392-424).
 * Accumulates with LDS/L2 atomics, so float summation order is not fixed (<= ~1e-6 abs). */
size_t ivlm_lift_mesh_dense_workspace_bytes(int B, int V, int Nv);
int ivlm_lift_mesh_dense(const float *logits, const int32_t *vid, const float *bary, int B, int V,
                         int64_t HW, int Nv, int mode, float param, float *out, float *nviews,
                         void *workspace, size_t workspace_bytes, ivlm_stream_t stream);

/* ObjectPCAfford3DPredictor.forward (model/components.py:289-347; NumPy twin
 * preprocess_data/utils_obj_pc.py:47-86): per-view mean of the values of the pixels mapped to a
 * point, then mean over the views that saw it.  No sigmoid, no clamp.
 *   probs f32 [B,V,HW]; pid i32 [B,V,HW] (or [V,HW] shared by the batch when pid_batched == 0),
 *   -1 = no point;  out f32 [B,Np];  nviews f32 [B,Np] (may be NULL) */
size_t ivlm_lift_points_workspace_bytes(int B, int V, int Np);
int ivlm_lift_points(const float *probs, const int32_t *pid, int pid_batched, int B, int V, int64_t HW,
                     int Np, float *out, float *nviews, void *workspace, size_t workspace_bytes,
                     ivlm_stream_t stream);

/* Sam.postprocess_masks (model/segment_anything/modeling/sam.py:137-172): bilinear
 * (align_corners=False) h x w -> img x img, crop to (in_h,in_w), bilinear -> (oh,ow); fp32 out
 * whatever the input dtype (sam.py:161).  apply_sigmoid != 0 additionally applies the in-place
 * sigmoid of InteractVLM.py:452-456 (oafford + 'HM' views).
 *   low  f32|bf16 [n,h,w]  ->  out f32 [n,oh,ow] */
int ivlm_postprocess_masks(const void *low, int dtype, int n, int h, int w, int img, int in_h, int in_w,
                           int oh, int ow, int apply_sigmoid, float *out, ivlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense building blocks (bf16 storage, fp32 accumulation) used by the stage runners below and
 * exposed for unit testing.  Activation codes: 0 none, 1 GELU(erf), 2 quick-GELU, 3 ReLU, 4 SiLU,
 * 5 SwiGLU over row-interleaved (gate_j, up_j) weights (output has N/2 columns), 6 sigmoid.
 * ------------------------------------------------------------------------------------------- */

/* nn.Linear: C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[m (% res_mod), N]
 * (HF LlamaModel / CLIPVisionModel linears; SAM image_encoder.py:222-260, common.py:13-27,
 * transformer.py:185-242; InteractVLM.py:100-112 text_hidden_fcs; llava_arch.py:35 mm_projector).
 * bf16 A/W/bias/residual, K % 8 == 0, lda/ldw % 8 == 0; C bf16 or f32 (out_f32).  batch > 1 runs a
 * strided batch (strides in elements).  M <= 8 takes the weight-streaming GEMV path (batch-1 decode:
 * HF greedy search under InteractVLM.evaluate, model/InteractVLM.py:524-531), K % 8 == 0 suffices there.
 * rms_w != NULL (M <= 8 only) fuses the preceding HF LlamaRMSNorm: C = act((A * rsqrt(mean(A^2)+rms_eps) * rms_w) . W^T). */
int ivlm_gemm_bf16(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                   const void *bias, const void *residual, int64_t ldr, int res_mod, int M, int N, int K,
                   int act, int out_f32, int batch, int64_t strideA, int64_t strideW, int64_t strideC,
                   int64_t strideR, const void *rms_w, float rms_eps, ivlm_stream_t stream);

/* Split-K variant for small-M GEMMs (LLaMA prefill, CLIP: too few output tiles for 256 CUs): same result contract as
 * ivlm_gemm_bf16 (batch 1, act != SwiGLU, no RMS fusion); K % (8*splits) == 0, N % 4 == 0.  fp32 partial sums of the
 * `splits` K-slices go to the caller's workspace (ivlm_gemm_splitk_workspace_bytes) and are summed in slice order. */
size_t ivlm_gemm_splitk_workspace_bytes(int M, int N, int splits);
int ivlm_gemm_bf16_splitk(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                          const void *bias, const void *residual, int64_t ldr, int res_mod, int M, int N, int K,
                          int act, int out_f32, int splits, void *workspace, size_t workspace_bytes,
                          ivlm_stream_t stream);
/* Tail-split variant for large GEMMs whose 256 x 256 tile count is just over a multiple of the 256 CUs (SAM mlp2:
 * 16384 x 1280 -> 320 tiles = one full round + a quarter-filled one): the full rounds run as usual, the last tiles % 256
 * tiles run as `splits` K slices on the otherwise idle CUs (tail * splits <= 256, K % (64*splits) == 0; fp32 partials in
 * workspace [splits, M, N]) and are reduced with the bias / activation / residual epilogue by a second small kernel.
 * Same result contract as ivlm_gemm_bf16 (batch 1, act != SwiGLU); IVLM_ERR_UNSUPPORTED when the shape does not qualify. */
int ivlm_gemm_bf16_tailsplit(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                             const void *bias, const void *residual, int64_t ldr, int res_mod, int M, int N, int K,
                             int act, int out_f32, int splits, void *workspace, size_t workspace_bytes,
                             ivlm_stream_t stream);

/* Benchmark/test hook: M == 1 GEMVs use the wave-per-row kernel (0, default: faster as a stand-alone launch) or the flat
 * slab-streaming kernel (1; the streaming code of ivlm_llama_generate). */
int ivlm_gemv_slab_enable(int on);
/* Benchmark/test hook for the skinny-M dispatch: rows M in [min_m, 16] against matrices with K, N >= 1024 go to the
 * split-K MFMA kernel (csrc/gemv_mfma.hip) instead of the wave-per-row GEMV / tile GEMM; 0 restores the automatic choice. */
int ivlm_gemv_mfma_min_m(int min_m);

/* Benchmark/test hook: force the GEMM block tile (64 = 128x64, 128, 256; 0 = automatic choice). Returns the previous value. */
int ivlm_gemm_tile_override(int tile);

/* nn.LayerNorm over the last dim (also SAM LayerNorm2d with NHWC activations, common.py:32-42);
 * bf16 in/out, fp32 statistics, cols % 8 == 0, cols <= 8192.  gelu_after != 0 fuses the exact-erf GELU that
 * follows LayerNorm2d in the mask decoder's upscaler (mask_decoder.py:53-63). */
int ivlm_layernorm_bf16(const void *x, const void *w, const void *b, void *y, int64_t rows, int cols,
                        float eps, int gelu_after, ivlm_stream_t stream);
/* HF LlamaRMSNorm: y = w * bf16(x * rsqrt(mean(x^2) + eps)). */
int ivlm_rmsnorm_bf16(const void *x, const void *w, void *y, int64_t rows, int cols, float eps,
                      ivlm_stream_t stream);

/* Fused multi-head attention: o = softmax(scale * q.k^T + bias + mask) . v, never materialising
 * the score matrix (SAM image_encoder.py:235-260, transformer.py:220-242; HF CLIP / LLaMA attention).
 *   q [B,H,Sq,D], k/v [B/kv_batch_div,H,Sk,D], o [B,H,Sq,D] addressed through element strides
 *   strides[12] = {q_b,q_h,q_row, k_b,k_h,k_row, v_b,v_h,v_row, o_b,o_h,o_row} (multiples of 8; o: of 4)
 *   D in {16,32,64,80,128};  causal: key j visible to query i iff j <= i + q_pos0 (KV cache offset)
 *   prescale_q: 1 = scores are bf16(q*scale).k (SAM image_encoder.py:244, HF CLIP); 0 = (q.k)*scale (HF LLaMA)
 *   rel_h f32 [B*H,Sq,rel_kh], rel_w f32 [B*H,Sq,rel_kw] (or NULL): bias[q,k] = rel_h[q,k/rel_kw] + rel_w[q,k%rel_kw] */
int ivlm_attention_bf16(const void *q, const void *k, const void *v, void *o, const int64_t *strides_host, int B,
                        int H, int Sq, int Sk, int D, float scale, int causal, int q_pos0, const float *rel_h,
                        const float *rel_w, int rel_kh, int rel_kw, int kv_batch_div, int prescale_q,
                        ivlm_stream_t stream);
/* Benchmark/test hook: -1 (default) picks per shape; 0 forces the 4-wave / 128-query block, 1 the 8-wave ping-pong block
 * (256 queries; one wave group on the matrix unit while the other does its softmax on the VALU). */
int ivlm_attention_pingpong(int mode);

/* add_decomposed_rel_pos operands (image_encoder.py:354-392), q_size == k_size == (SH,SW):
 *   rel_h[bh,q,kh] = q . rel_pos_h[qh-kh+SH-1],  rel_w[bh,q,kw] = q . rel_pos_w[qw-kw+SW-1]  (rounded to bf16
 *   like the reference's model-dtype einsum, stored f32).  tab_h bf16 [2*SH-1,D], tab_w bf16 [2*SW-1,D]. */
int ivlm_relpos_bias(const void *q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void *tab_h, const void *tab_w,
                     int B, int H, int SH, int SW, int D, float *rel_h, float *rel_w, ivlm_stream_t stream);
/* Second half of the GEMM formulation of the same operands: G bf16 [H][B*S][npad] = q . [rel_pos_h ; rel_pos_w]^T (one
 * batched ivlm_gemm_bf16 over the heads, K = head dim), g_head_stride = elements between heads; this gathers the Toeplitz
 * shifts rel_h[bh,s,kh] = G[qh-kh+SH-1], rel_w[bh,s,kw] = G[(2SH-1) + qw-kw+SW-1] into f32 [B*H,S,SH] / [B*H,S,SW]. */
int ivlm_relpos_gather(const void *G, int64_t g_head_stride, int npad, int B, int H, int SH, int SW, float *rel_h,
                       float *rel_w, ivlm_stream_t stream);

/* Whole greedy generation after the prefill, in ONE persistent launch (HF GenerationMixin greedy search driven by
 * InteractVLM.evaluate, model/InteractVLM.py:524-531; per token: LlamaModel.forward of one position with the KV
 * cache, final RMSNorm, lm_head, argmax, embed_tokens of the chosen id).  One resident workgroup per CU streams
 * contiguous row slabs of every weight matrix; phases are separated by a device-wide barrier; argmax, the EOS test
 * and the optional forced ids run on the device.
 *   layer_ptrs  device int64 [L][6]: addresses of input_layernorm.weight, q|k|v weight [3*hidden, hidden] (q, k, v
 *               rows concatenated), o_proj.weight, post_attention_layernorm.weight, gate|up weight [2*inter, hidden]
 *               (rows interleaved gate_0, up_0, gate_1, ...), down_proj.weight [hidden, inter]; all bf16, dense.
 *   kcache/vcache bf16 [L, max_len, H, D] (cache_layer_stride elements per layer), holding positions < pos0.
 *   hidden_out  bf16 [>= pos0 + n_max - 1, hidden]: row pos0-1 is the INPUT (final-normed hidden state of the last
 *               prompt position); rows pos0.. receive the final-normed hidden state of each generated position.
 *   forced      int32 [n_max] or NULL: ids fed back instead of the argmax (teacher forcing; argmax still computed).
 *   new_ids / argmax_ids int32 [n_max]; generation stops after an id == eos or n_max ids.
 *   workspace   >= ivlm_llama_generate_workspace_bytes, 256-byte aligned; after the stream has drained, its first two
 *               int32 are {number of ids generated, error flag (1: a barrier wait exceeded 2 s - results invalid)}.
 * hidden, inter >= 512 and % 8 == 0, H * D == hidden, D <= 128, H <= number of CUs. */
size_t ivlm_llama_generate_workspace_bytes(int hidden, int inter);
int ivlm_llama_generate(const int64_t *layer_ptrs, int L, int H, int D, int hidden, int inter, int vocab, float eps,
                        float scale, const float *cos_tab, const float *sin_tab, void *kcache, void *vcache,
                        int64_t cache_layer_stride, int max_len, const void *embed, const void *final_norm,
                        const void *lm_head, void *hidden_out, int pos0, int n_max, int eos, const int32_t *forced,
                        int32_t *new_ids, int32_t *argmax_ids, void *workspace, size_t workspace_bytes,
                        ivlm_stream_t stream);

/* One decode step of HF LlamaAttention with a KV cache, for the newest token only: rotate-half RoPE of q,k at
 * position pos, append k,v to kcache/vcache [Tmax,H,D], o = softmax(q.K[0..pos]^T * scale).V[0..pos].
 * qkv bf16 [3,H,D] (output of the fused q|k|v projection), o bf16 [H,D]; D <= 128, pos < 4096. */
int ivlm_llama_decode_attn(const void *qkv, void *kcache, void *vcache, void *o, int H, int D, int pos,
                           float theta, float scale, const float *cos_tab, const float *sin_tab,
                           ivlm_stream_t stream);
/* Same, with the position read from device memory (int32 *pos_dev; the caller keeps *pos_dev < Tmax and < 4096): one
 * captured HIP graph of a decode step can then be replayed for every generated token. */
int ivlm_llama_decode_attn_devpos(const void *qkv, void *kcache, void *vcache, void *o, int H, int D,
                                  const int32_t *pos_dev, float theta, float scale, const float *cos_tab,
                                  const float *sin_tab, ivlm_stream_t stream);

/* B sequences in one launch (grid H x B): sequence b reads qkv + b*ldq, appends to kcache/vcache + b*cache_stride
 * ([Tmax,H,D] each), writes o + b*ldo and sits at position pos_dev[b] (strides in elements, multiples of 8).  The
 * batched counterpart of the reference's padded-batch generate (model/InteractVLM.py:524-531 with B > 1 prompts). */
int ivlm_llama_decode_attn_batch(const void *qkv, int64_t ldq, void *kcache, void *vcache, int64_t cache_stride,
                                 void *o, int64_t ldo, int B, int H, int D, const int32_t *pos_dev, float theta,
                                 float scale, const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);

/* torch.argmax(logits, -1) of HF greedy search (first maximal index); x f32 [rows, cols] -> out i32 [rows]. */
int ivlm_argmax_f32(const float *x, int rows, int cols, int32_t *out, ivlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data movement on the path (all bf16, 16-byte granules: cols/strides % 8 == 0)
 * ------------------------------------------------------------------------------------------- */
/* Conv2d(kernel = stride = ks) as GEMM operand: out[(b,gy,gx), (c,ky,kx)] zero-padded to Kpad columns
 * (SAM PatchEmbed image_encoder.py:404-426; HF CLIP patch_embedding). x [B,C,H,W] */
int ivlm_im2col_nchw(const void *x, void *out, int B, int C, int H, int W, int ks, int stride, int Kpad,
                     ivlm_stream_t stream);
/* 3x3 / pad 1 conv operand from channels-last x [B,H,W,C] -> [(b,y,x), (ky,kx,c)] (SAM neck, image_encoder.py:92-108) */
int ivlm_im2col3x3_nhwc(const void *x, void *out, int B, int H, int W, int C, ivlm_stream_t stream);
/* dst[r] = (idx[r] >= 0 ? src[idx[r]] : 0) + (add ? add[r] : 0): window_partition / window_unpartition + shortcut
 * (image_encoder.py:263-318, 177-193), embed_tokens gather (llava_arch.py:185-208). */
int ivlm_gather_rows(void *dst, int64_t ldd, const void *src, int64_t lds, const int32_t *idx, const void *add,
                     int64_t lda, int64_t rows, int cols, ivlm_stream_t stream);
/* out[r] = a[r] (op 0: +, op 1: *) b[r % b_rows]  (queries + query_pe, keys + key_pe: transformer.py:160-176;
 * [SEG] embedding * view encoding: InteractVLM.py:275-282) */
int ivlm_add_rows(void *out, const void *a, const void *b, int64_t rows, int cols, int64_t b_rows, int op,
                  ivlm_stream_t stream);
/* PositionEmbeddingRandom.forward (prompt_encoder.py:219-229): gauss f32 [2,F] -> pe bf16 [h*w, 2F]
 * (the table is a constant of the weights: computed once at load, in fp32) */
int ivlm_dense_pe(const void *gauss, void *pe, int h, int w, int F, ivlm_stream_t stream);
/* HF LlamaAttention rotary (rotate-half, base theta) applied in place to q,k of qkv [T,3,H,D] (row stride ld) at
 * positions pos0+t, and KV-cache append (kcache/vcache [Tmax,H,D], may be NULL). */
int ivlm_rope_kv(void *qkv, int64_t ld, int T, int H, int D, int pos0, float theta, void *kcache, void *vcache,
                 const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);
/* Decode attention (as ivlm_llama_decode_attn_devpos) and the o_proj GEMV + residual of the same layer in ONE launch:
 * x_out[hidden] = x + W_o . attention(qkv).  The o_proj blocks stream their weight rows while the attention blocks run and
 * wait for them on `counter` (int32, zeroed by the caller at the start of a generation; `step_dev` = tokens decoded so far,
 * incremented by the caller after each token; target = H * (step + 1)).  attn_scratch: hidden bf16.  status[0] != 0 after the
 * stream drained = a bounded wait expired (results invalid).  hidden = H*D in {512, 1024, 4096, 5120}. */
int ivlm_llama_attn_oproj(const void *qkv, void *kcache, void *vcache, void *attn_scratch, const void *wo, const void *x,
                          void *x_out, int H, int D, float theta, float scale, const float *cos_tab, const float *sin_tab,
                          const int32_t *pos_dev, const int32_t *step_dev, int32_t *counter, int32_t *status,
                          ivlm_stream_t stream);
/* ALL decoder layers of one generated token in ONE launch, as a dataflow of role-specialised workgroups (q|k|v rows ->
 * attention heads -> o_proj rows -> gate|up pairs -> down rows, layer after layer): every block streams its weight rows first
 * and then waits on a device counter for the blocks that produce its input, so no launch / barrier bubble stalls HBM.
 *   layer_ptrs as for ivlm_llama_generate; kcache/vcache bf16 [L, Tmax, H, D]; x0 bf16 [hidden] = embedding of the token;
 *   x_out bf16 [hidden] = residual stream after the last layer (before the final RMSNorm);
 *   pos_dev / step_dev: int32 in device memory (position of the token; tokens decoded so far in this generation);
 *   workspace: ivlm_llama_decode_layers_workspace_bytes, 256-byte aligned; its first L*5*32 int32 (counters) and the
 *   following int32 (status) must be zeroed by the caller at the start of a generation; status != 0 afterwards = a bounded
 *   wait expired, results invalid.  (hidden, inter) in {(4096,11008), (5120,13824), (1024,1376), (512,1024)}. */
size_t ivlm_llama_decode_layers_workspace_bytes(int L, int hidden, int inter);
int ivlm_llama_decode_layers(const int64_t *layer_ptrs, int L, int H, int D, int hidden, int inter, float eps, float theta,
                             float scale, const float *cos_tab, const float *sin_tab, void *kcache, void *vcache,
                             int64_t cache_layer_stride, const void *x0, void *x_out, const int32_t *pos_dev,
                             const int32_t *step_dev, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* The MLP of a decode layer (HF LlamaMLP + post_attention_layernorm + residual) in ONE launch:
 * x_out = x2 + W_down . (SiLU(g) * u), (g_j, u_j) = rows (2j, 2j+1) of wgu . RMSNorm(x2).  The down_proj blocks stream their
 * first weight rows while the gate|up blocks run and wait for them on `counter` (zeroed at the start of a generation;
 * target = number of gate|up blocks * (*step_dev + 1)); h_scratch bf16 [inter]; status as for ivlm_llama_attn_oproj. */
int ivlm_llama_gateup_down(const void *x2, const void *ln_w, float eps, const void *wgu, const void *wdown, void *h_scratch,
                           void *x_out, int hidden, int inter, const int32_t *step_dev, int32_t *counter, int32_t *status,
                           ivlm_stream_t stream);
/* fp32 rotary tables cos/sin [T, D/2] (optional inputs of ivlm_rope_kv / ivlm_llama_decode_attn; NULL = compute) */
int ivlm_rope_table(float *cos_tab, float *sin_tab, int T, int D, float theta, ivlm_stream_t stream);
/* Caller-side image preprocessing (run_demo.py:65-79 `preprocess`: (x - mean)/std then zero-pad to the square model
 * input; HF CLIPImageProcessor: centre crop, 1/255 rescale, normalise): src u8 [H,W,3] RGB on the device, crop
 * (y0,x0,ch,cw) -> out bf16|f32 [3,OH,OW].  mean3/std3 are HOST pointers in 0..255 units. */
int ivlm_normalize_pad_u8(const uint8_t *src, int H, int W, int y0, int x0, int ch, int cw, const float *mean3_host,
                          const float *std3_host, void *out, int out_bf16, int OH, int OW, ivlm_stream_t stream);
/* masks = hyper_in @ upscaled_embedding (mask_decoder.py:150-153) for one mask token: up bf16
 * [B,gh,gw,2,2,2,2,C] (output of the two k2s2 transposed convs, channels last), hyper bf16 [B,C]
 * -> low f32 [B,4gh,4gw] */
int ivlm_mask_dot(const void *up, const void *hyper, float *low, int B, int gh, int gw, int C,
                  ivlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Right after the path
 * ------------------------------------------------------------------------------------------- */
/* get_h_contact_metrics (utils/eval_utils.py:63-94): per sample F1 / precision / recall of (pred >= thr) against
 * (gt > 0); gt, pred f32 [B,n] -> out f32 [B,3] = (f1, precision, recall). */
/* get_h_geo_metric (utils/eval_utils.py:129-151): dist f32 [n,n] geodesic matrix, pred / gt f32 [B,n] -> out f32 [B,2] =
 * (false-positive distance, false-negative distance) per sample: rows = {pred >= 0.5} (all rows if empty), columns =
 * {gt == 1} (all if empty), fp = mean over rows of the min over columns, fn = mean over columns of the min over rows.
 * n <= 8192.  The caller averages over the batch like the reference does. */
size_t ivlm_h_geo_workspace_bytes(int n);
int ivlm_h_geo_metric(const float *dist, const float *pred, const float *gt, int B, int n, float *out, void *workspace,
                      size_t workspace_bytes, ivlm_stream_t stream);
int ivlm_contact_prf(const float *gt, const float *pred, int B, int n, float thr, float *out, ivlm_stream_t stream);
/* convert_contacts (utils/utils.py:428-443): y[b] = M . x[b] with the SMPL->SMPL-X matrix M [rows,cols] held in CSR
 * (row_ptr i32 [rows+1], col i32 [nnz], val f32 [nnz]) instead of the reference's dense 10475x6890 bmm. */
int ivlm_spmv_csr(const int32_t *row_ptr, const int32_t *col, const float *val, const float *x, int B, int rows,
                  int cols, float *y, ivlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * "Render" of Render-Localize-Lift: rasterise a mesh / point cloud into the lift tables.
 * pytorch3d semantics (preprocess_data/render_mesh_utils.py:115-174, utils_obj_pc.py:28-42,88-113,
 * utils/demo_utils.py:171-257): FoV-perspective camera, X_view = X_world.R + T, NDC +X left / +Y up,
 * faces_per_pixel = 1, blur_radius = 0, perspective-correct barycentrics, nearest z wins.
 *   cam12_host: HOST pointer to 12 floats = R (row-major 3x3) then T, from look_at_view_transform(d,e,a) with the
 *   x/y translation already added to T.  fov_deg = 60 in the reference.
 * ------------------------------------------------------------------------------------------- */
size_t ivlm_raster_workspace_bytes(int n_verts_or_points, int H, int W);
/* verts f32 [Nv,3], faces i32 [Nf,3] -> p2v i32 [H,W,3] (-1 background), bary f32 [H,W,3] (-1 background),
 * pix_to_face i32 [H,W] (may be NULL) */
int ivlm_rasterize_mesh(const float *verts, int nv, const int32_t *faces, int nf, const float *cam12_host,
                        float fov_deg, int H, int W, int32_t *p2v, float *bary, int32_t *pix_to_face,
                        void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* pts f32 [Np,3], disc radius in NDC units -> map i32 [H,W] = index of the nearest covering point, -1 none */
int ivlm_rasterize_points(const float *pts, int np, const float *cam12_host, float fov_deg, float radius, int H,
                          int W, int32_t *map, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* Hard Phong shading of a rasterised mesh = the colour renders fed to SAM for object meshes (utils/demo_utils.py:146-168:
 * MeshRenderer + HardPhongShader + one PointLights over TexturesVertex colours; same in render_mesh_utils.py:177-198).
 * p2v i32 [npix,3] / bary f32 [npix,3] from ivlm_rasterize_mesh; verts / normals / colors f32 [Nv,3] (world space, unit
 * vertex normals, vertex colours); light3 / cam3 / bg3: HOST pointers to 3 floats (light location, camera centre, background
 * colour); ambient / diffuse / specular: the light's grey levels (0.5 / 0.3 / 0.2 in the reference), shininess 64.
 * out_rgb u8 [npix,3] = trunc(colour * 255). */
int ivlm_phong_shade(const int32_t *p2v, const float *bary, const float *verts, const float *normals, const float *colors,
                     int npix, const float *light3_host, const float *cam3_host, float ambient, float diffuse,
                     float specular, float shininess, const float *bg3_host, uint8_t *out_rgb, ivlm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IVLM_HIP_H */
