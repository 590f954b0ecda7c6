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
 *   - dtype codes: IVLM_F32 = 0, IVLM_BF16 = 1; IVLM_BF16_SPLIT = 2 (outputs of the row kernels only): a row of
 *     `cols` fp32 values written as [hi(cols) | lo(cols)] bf16 with x = hi + lo to 2^-17 - the A operand of an
 *     fp32-activation GEMM on the bf16 matrix cores (K' = 2K against [W | W])
 *   - precision policy: weights bf16 (the checkpoint's own dtype); residual streams fp32; MFMA operands
 *     bf16; the weight-streaming decode kernels (M <= 16) take fp32 activations with exact products.
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
#define IVLM_BF16_SPLIT 2
#define IVLM_F16 4 /* IEEE fp16 (outputs of ivlm_layernorm; operands of ivlm_gemm_bf16 with IVLM_GEMM_F16) */
#define IVLM_F16_SPLIT 5 /* a row [hi(cols) | lo(cols)] of IEEE halves with x = hi + lo (outputs of ivlm_layernorm / ivlm_rmsnorm; the
                          * A operand of ivlm_gemm_bf16 with IVLM_GEMM_F16 | IVLM_GEMM_A_SPLIT; its hi half alone is the IVLM_F16 row) */
#define IVLM_FP8 3 /* OCP e4m3 bytes (BASELINE configs[4]: fp8 operands for the big GEMMs); always with a per-tensor scale */

/* flags of ivlm_gemm_bf16 / ivlm_gemm_bf16_splitk */
#define IVLM_GEMM_A_F32 1   /* A is fp32 [M,K] (M <= 16 weight-streaming paths only; lda % 4 == 0) */
#define IVLM_GEMM_RES_F32 2 /* residual is fp32 (fp32 residual stream) */
#define IVLM_GEMM_A_SPLIT 4 /* tile GEMM (M > 16): A rows are [hi(K) | lo(K)] bf16 (IVLM_BF16_SPLIT, lda >= 2K): an fp32-activation
                               GEMM on the bf16 matrix cores against the plain [N,K] weight (each W tile is used twice) */
#define IVLM_GEMM_OUT_SPLIT 8 /* tile GEMM: C is bf16 [M, >= 2 n_out], the fp32 result written as [hi(n_out) | lo(n_out)] */
#define IVLM_GEMM_F16 16      /* tile GEMM: A and W hold IEEE fp16 values (an fp16 activation carries 1/8 of the bf16 rounding error
                                 at the same MFMA rate; bf16 weights inside the fp16 range convert exactly) */
#define IVLM_GEMM_OUT_F16 32  /* tile GEMM: the 16-bit output (out_f32 = 0) is written as fp16; with IVLM_GEMM_OUT_SPLIT: the [hi | lo]
                                 halves are IEEE halves (IVLM_F16_SPLIT rows).  IVLM_GEMM_F16 | IVLM_GEMM_A_SPLIT: A rows are
                                 IVLM_F16_SPLIT (the "exact q" projection of the fp16 mode: q = W_q . (hi + lo)) */

#define IVLM_GEMM_W_PANEL 64  /* tile GEMM: W is stored as K/64 panels of [N][64] (element (n, k) at (k/64) * 64 N + n * 64 + k % 64;
                                 K % 64 == 0, ldw ignored): a wave's DMA instruction then reads 1 KB contiguous instead of eight
                                 128-byte lines a row stride apart (78 vs 52 - 64 GB/s of L2-hit feed per CU).  Static weights are
                                 panelised once at load (the LLaMA prefill of the host model); same results as the row-major layout */

typedef void *ivlm_stream_t;

/* library identity --------------------------------------------------------------------------- */
int ivlm_abi_version(void);
const char *ivlm_error_string(int code);
const char *ivlm_build_arch(void); /* "gfx950" */
/* detail of the most recent IVLM_ERR_LAUNCH on this thread: HIP error text + source location */
const char *ivlm_last_hip_error(void);

/* Measurement hook (bench.py's roofline legs): arm kernel-attached timing for the GEMM / GEMV / lift launches issued by the
 * following calls on THIS thread - start_event (hipEvent_t) is recorded by the command processor when the first of them starts,
 * stop_event when the last one ends (hipExtLaunchKernelGGL); pass NULL, NULL to disarm.  Returns the number of launches that
 * were instrumented since the previous call.  Not for use during stream capture. */
int ivlm_profile_launches(void *start_event, void *stop_event);

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
/* The same inversion for a pixel -> point map of ObjectPCAfford3DPredictor (components.py:318-347; pid i32 [V,HW], -1 = no point):
 * a point-major CSR with weights 1, rows ordered by pixel; cap >= V*HW.  Evaluated by ivlm_lift_mesh_plan with mode 2 (the map's
 * own values are averaged: per view over the pixels of a point, then over the views that see it; no sigmoid, no clip): no
 * atomics, no workspace, bit-reproducible - the p2pmap files are cached per path by the predictor, so a map that comes back is
 * inverted once. */
int ivlm_lift_points_plan_build(const int32_t *pid, int V, int64_t HW, int Np, int32_t *row_ptr, int32_t *ent_pix, float *ent_w,
                                int64_t cap, int32_t *nnz_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* Benchmark hook: blocks per CU of the streaming lift kernels (ivlm_lift_mesh_dense / ivlm_lift_points); returns the previous value. */
int ivlm_lift_stream_blocks_per_cu(int bpc);

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
/* Optional heads of ModifiedSAM (InteractVLM.py:20-44; off in every released configuration).
 * UncertaintyModule.forward (components.py:55-78): embeddings fp32 [rows, 256] (channels last: rows = views * 64 * 64), bf16
 * weights linear1 [64,256] / linear2 [16,64] / linear3 [1,16] + biases -> out fp32 [rows] holding the bf16 values of the bf16
 * module (input, every linear output and the softplus rounded to bf16 as the module running inside the bf16 model rounds them). */
int ivlm_uncertainty_mlp(const float *embeddings, int64_t rows, const void *w1, const void *b1, const void *w2, const void *b2,
                         const void *w3, const void *b3, float *out, ivlm_stream_t stream);
/* F.interpolate(src, size=(oh, ow), mode="bilinear", align_corners=False) (InteractVLM.py:446-447, 615-616): src fp32 [n,h,w] ->
 * dst [n,oh,ow] fp32 or bf16 (fp32 taps and weights, one rounding). */
int ivlm_resize_bilinear(const float *src, int n, int h, int w, void *dst, int dst_dtype, int oh, int ow, ivlm_stream_t stream);

/* The same with the sigmoid of InteractVLM.py:452-456 applied only where the ground-truth mask gt f32 [n,oh,ow] differs from
 * ignore_label ('oafford' samples with 'HM' object views; raw logits elsewhere). */
int ivlm_postprocess_masks_valid(const void *low, int dtype, int n, int h, int w, int img, int in_h, int in_w, int oh, int ow,
                                 const float *gt, float ignore_label, float *out, ivlm_stream_t stream);

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
 * rms_w != NULL (M <= 16 only) fuses the preceding HF LlamaRMSNorm: C = act((A * rsqrt(mean(A^2)+rms_eps) * rms_w) . W^T).
 * flags: IVLM_GEMM_A_F32 (M <= 16: fp32 activations, exact bf16 x fp32 products in the GEMV, hi + lo bf16 operand split on the
 * skinny MFMA kernel), IVLM_GEMM_RES_F32 (fp32 residual).
 * out_rows != NULL (tile GEMM path, M > 16): scatter epilogue - row m of the product is written to row out_rows[m] of C and
 * takes its residual from that row; rows with out_rows[m] < 0 are dropped (SAM window_unpartition + shortcut,
 * image_encoder.py:186-190, folded into the proj GEMM; C may alias the residual).
 * a_rows != NULL (tile GEMM path): gather prologue - row m of the product reads row a_rows[m] of A (all entries valid). */
int ivlm_gemm_bf16(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                   const void *bias, const void *residual, int64_t ldr, int res_mod, int M, int N, int K,
                   int act, int out_f32, int batch, int64_t strideA, int64_t strideW, int64_t strideC,
                   int64_t strideR, const void *rms_w, float rms_eps, int flags, const int32_t *out_rows,
                   const int32_t *a_rows, ivlm_stream_t stream);

/* The tile GEMM (M > 16, bf16, batch 1, K % 64 == 0) with operands and / or output in the K-PANEL layout: a matrix [rows, K] stored
 * as K/64 panels of [rows][64] elements, element (r, k) at (k/64) * kstep + r * 64 + (k % 64) (kstep >= 64 * rows, in elements).
 * One wave DMA instruction of the kernels (8 rows x 128 B of a K tile) then reads 1 KB contiguous instead of eight lines a row
 * stride apart: 78 instead of 52 GB/s of L2-hit feed per CU, the ceiling of the 256 x 256 kernel.  a_kstep / w_kstep
 * = 0: that operand is row-major (lda / ldw).  c_panel != 0: bf16 C written in the panel layout (N % 64 == 0; ldc ignored) for
 * the GEMM that consumes it; out_rows / a_rows only with row-major C / A.  The SAM encoder's weights are panelised once at
 * load, norm1 / norm2 and the mlp lin1 epilogue write panels (image_encoder.py:150-260). */
int ivlm_gemm_bf16_panel(const void *A, int64_t lda, int64_t a_kstep, const void *W, int64_t ldw, int64_t w_kstep, void *C,
                         int64_t ldc, int64_t c_panel, const void *bias, const void *residual, int64_t ldr, int M, int N, int K,
                         int act, int out_f32, int flags, const int32_t *out_rows, const int32_t *a_rows, ivlm_stream_t stream);

/* fp8 (OCP e4m3) operands for the big GEMMs (BASELINE.json configs[4]): C = act((A8 . W8^T) * *scale_a *
 * *scale_w + bias) + residual on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, fp32 accumulation, the tile kernels
 * of ivlm_gemm_bf16 (same tiles, DMA and epilogues; half the K tiles).  A [M,K] / W [N,K] are byte matrices (K, lda, ldw in
 * bytes, multiples of 16; K % 128 != 0 tails are zero-filled); per-tensor scales are device scalars (x = q * scale).
 * out_kind: IVLM_BF16, IVLM_F32, or 2 = e4m3 output act(...) / *scale_out (mlp1 -> mlp2).  act: none | GELU.  M > 16.
 * The reference has no counterpart (it is a bf16 model): an opt-in variant whose error is reported against the bf16 path. */
int ivlm_gemm_fp8(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc, const void *bias,
                  const void *residual, int64_t ldr, int M, int N, int K, int act, int out_kind, const float *scale_a,
                  const float *scale_w, const float *scale_out, int flags, const int32_t *out_rows, const int32_t *a_rows,
                  ivlm_stream_t stream);

/* Batch-1 decode linear with e4m3 WEIGHTS (BASELINE.json configs[4]; opt-in variant): C[1,N] = act((x . Wq^T) * *scale_w + bias) +
 * residual, x fp32 [K] (optionally RMS-normalised: rms_w / rms_eps as in ivlm_gemm_bf16), Wq e4m3 bytes [N,K] (row stride ldw
 * bytes; K, ldw multiples of 16; K <= 15360), one per-tensor scale in device memory (the tensor ivlm_gemm_fp8 uses for the
 * prefill).  The decode step streams half the bytes per token; products are exact, the only error is the weight quantisation.
 * act: none | SwiGLU (row-interleaved gate / up) | ...; flags: IVLM_GEMM_RES_F32. */
int ivlm_gemv_fp8w(const float *x, const void *Wq, int64_t ldw, const float *scale_w, void *C, const void *bias,
                   const void *residual, int N, int K, int act, int out_f32, const void *rms_w, float rms_eps, int flags,
                   ivlm_stream_t stream);

/* Split-K variant for small-M GEMMs (LLaMA prefill, CLIP: too few output tiles for 256 CUs): same result contract as
 * ivlm_gemm_bf16 (batch 1, act != SwiGLU, no RMS fusion); K % (8*splits) == 0, N % 4 == 0.  fp32 partial sums of the
 * `splits` K-slices go to the caller's workspace (ivlm_gemm_splitk_workspace_bytes) and are summed in slice order. */
size_t ivlm_gemm_splitk_workspace_bytes(int M, int N, int splits);
int ivlm_gemm_bf16_splitk(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                          const void *bias, const void *residual, int64_t ldr, int res_mod, int M, int N, int K,
                          int act, int out_f32, int splits, void *workspace, size_t workspace_bytes, int flags,
                          ivlm_stream_t stream);
/* ... with the reduction FUSED into the GEMM launch (no second launch, no second pass over the product): every block stores its fp32
 * partial tile and counts its arrival on the tile's counter; the block that arrives last sums the tile's slices in slice order and
 * applies the epilogue - the same values as ivlm_gemm_bf16_splitk, bit for bit.  `counters`: IVLM_SPLITK_COUNTERS int32 words
 * that the caller zeroes ONCE (hipMemset); every call leaves them at zero.  One array per stream: launches that may run
 * concurrently must not share it.  (M = 330 LLaMA prefill / M = 257 CLIP: 199 launches fewer per image.)
 * Tiling: only the tile kernels of gemm.hip carry the fix-up, so a chosen or FORCED (ivlm_gemm_tile_override) 256 / 320 / 512 tile
 * is remapped to 128 x 128 here - the partial GEMMs of this entry point can run another tiling than ivlm_gemm_bf16_splitk's
 * (same values); a tile-override benchmark of the fused form measures the remapped tiling. */
#define IVLM_SPLITK_COUNTERS 4096
int ivlm_gemm_bf16_splitk_fused(const void *A, int64_t lda, const void *W, int64_t ldw, void *C, int64_t ldc,
                                const void *bias, const void *residual, int64_t ldr, int res_mod, int M, int N, int K,
                                int act, int out_f32, int splits, void *workspace, size_t workspace_bytes, int32_t *counters,
                                int flags, ivlm_stream_t stream);
/* Benchmark hook of the GEMV grid shaping: resident blocks per CU assumed (default 4) and the N above which a wave takes two
 * weight rows per step (default 8192); 0 = default.  max_blocks_per_cu < 0: the persistent kernel also serves M = 1 fp32 rows
 * (which otherwise take gemv1_kernel: one row per wave, 1024-thread blocks, no persistence). */
int ivlm_gemv_tuning(int max_blocks_per_cu, int rows2_min_n);
/* Benchmark hook of the batch-1 decode GEMV (gemv1_kernel): waves per weight row (1 or 2); 0 = default (1). */
void ivlm_gemv1_tuning(int ksplit);
/* A/B hook: dynamic LDS (bytes, <= 128 KB) requested by batch-1 GEMV grids of at most one block per CU (the 4096-row matrices): above
   80 KB only one block fits a CU, so the dispatcher has to give every CU exactly one.  0 = off. */
void ivlm_gemv1_lds_floor(int bytes);
/* Benchmark/test hook for the skinny-M dispatch: rows M in [min_m, 16] against matrices with K, N >= 1024 go to the
 * split-K MFMA kernel (csrc/gemv_mfma.hip) instead of the wave-per-row GEMV / tile GEMM; 0 restores the automatic choice. */
int ivlm_gemv_mfma_min_m(int min_m);
/* Benchmark hook of the skinny MFMA kernel: 16-row weight tiles per block (1, 2, 3, 4 or 6); 0 = the automatic choice
 * (fewest blocks-per-CU x tiles among 1 / 2 / 3, then the most tiles: 3 for the 12288 fused qkv rows and the 22016 gate-up rows);
 * 10 q + g: q tiles for 8192 < N <= 16384, g tiles above, 1 below (sweeps). */
void ivlm_skinny_tuning(int tiles_per_block);

/* Benchmark hook: column split of GEMMs whose 256 x 256 tile count under-fills its last round (default 1 = on). */
int ivlm_gemm_nsplit(int on);
/* Benchmark/test hook: force the GEMM block tile (64 = 128x64, 128, 256, 512 = the 8-phase 256x256 kernel, 320 = 256x320, 176;
 * 0 = automatic choice). Returns the previous value. */
int ivlm_gemm_tile_override(int tile);
/* Benchmark/test hook: automatic choice of the 256 x 320 tile for N % 320 == 0 (SAM ViT-H widths; default on). Returns the previous value. */
int ivlm_gemm_tile320(int on);

/* nn.LayerNorm over the last dim (also SAM LayerNorm2d with NHWC activations, common.py:32-42);
 * x / y bf16 or fp32 (dtype codes), fp32 statistics, cols % 8 == 0, cols <= 8192.  gelu_after != 0 fuses the exact-erf GELU
 * that follows LayerNorm2d in the mask decoder's upscaler (mask_decoder.py:53-63).  out_rows != NULL: row r is written to
 * row out_rows[r] of y (SAM window_partition, image_encoder.py:263-288, folded into norm1; the caller keeps the padded rows
 * of y zero).  y_dtype IVLM_FP8 (cols <= 2048, no GELU): y = e4m3(norm / *fp8_scale), the operand of ivlm_gemm_fp8. */
int ivlm_layernorm(const void *x, int x_dtype, const void *w, const void *b, void *y, int y_dtype, int64_t rows, int cols,
                   float eps, int gelu_after, const int32_t *out_rows, const float *fp8_scale, ivlm_stream_t stream);
/* HF LlamaRMSNorm: y = w * (x * rsqrt(mean(x^2) + eps)); a bf16 x is cast back to bf16 before the weight multiply as HF
 * does, an fp32 x (fp32 residual stream) is not. */
int ivlm_rmsnorm(const void *x, int x_dtype, const void *w, void *y, int y_dtype, int64_t rows, int cols, float eps,
                 ivlm_stream_t stream);

/* The same with an e4m3 output: y = e4m3(norm(x) / *fp8_scale) (bytes [rows, cols]): the operand of ivlm_gemm_fp8 (fp8 variant of
 * the LLaMA prefill, BASELINE.json configs[4]). */
int ivlm_rmsnorm_fp8(const void *x, int x_dtype, const void *w, void *y, int64_t rows, int cols, float eps, const float *fp8_scale,
                     ivlm_stream_t stream);

/* Fused multi-head attention: o = softmax(scale * q.k^T + bias + mask) . v, never materialising
 * the score matrix (SAM image_encoder.py:235-260, transformer.py:220-242; HF CLIP / LLaMA attention).
 *   q [B,H,Sq,D], k/v [B/kv_batch_div,H,Sk,D], o [B,H,Sq,D] addressed through element strides
 *   strides[12] = {q_b,q_h,q_row, k_b,k_h,k_row, v_b,v_h,v_row, o_b,o_h,o_row} (multiples of 8; o: of 4)
 *   D in {16,32,64,80,128};  causal: key j visible to query i iff j <= i + q_pos0 (KV cache offset)
 *   prescale_q: 1 = scores are bf16(q*scale).k (SAM image_encoder.py:244, HF CLIP); 0 = (q.k)*scale (HF LLaMA)
 *   rel_h f32 [B*H,Sq,rel_kh], rel_w f32 [B*H,Sq,rel_kw] (or NULL): bias[q,k] = rel_h[q,k/rel_kw] + rel_w[q,k%rel_kw]
 *   TABLE MODE (rel_w == NULL, rel_h != NULL; D = 80, rel_kh == rel_kw == side, 2 * side <= 32, Sq == Sk == side^2: SAM's 14 x 14
 *   windows): rel_h points to the bf16 table [64, D] = [rel_pos_h (2 side - 1 rows) ; rel_pos_w (2 side - 1 rows) ; zeros] and the
 *   kernel computes the terms itself (one small MFMA product per query tile) - no ivlm_relpos_bias pass, no [B*H,Sq,2 side] arrays.
 *   The 64 x 64 grid (SAM's global blocks: rel_kh == rel_kw == 64, Sq == Sk == 4096) has a table mode too: rel_h points to the
 *   [>= 254, D] table [rel_pos_h (127 rows) ; rel_pos_w (127 rows)] and every 128-query block computes its rel_h / rel_w terms
 *   before its tile loop (39 MFMAs per query tile against 2816) - no G = q . T^T GEMM, no gather pass. */
int ivlm_attention_bf16(const void *q, const void *k, const void *v, void *o, const int64_t *strides_host, int B,
                        int H, int Sq, int Sk, int D, float scale, int causal, int q_pos0, const float *rel_h,
                        const float *rel_w, int rel_kh, int rel_kw, int kv_batch_div, int prescale_q,
                        ivlm_stream_t stream);
/* The same operator on IEEE fp16 tensors (q / k / v / o, and in table mode the fp16 [64, D] table): identical tiles and layouts on
 * v_mfma_f32_16x16x32_f16; q * scale, the softmax weights and the output are rounded to fp16 (11 significant bits: an eighth of
 * the bf16 rounding error at the same matrix-core rate); a value past +-65504 becomes inf (not clamped: the caller can see it).  The default precision of the three towers
 * (image_encoder.py:235-260; HF CLIP / LLaMA attention).  Shapes of the path: D = 64 (plain), D = 80 with rel_h (prescale_q = 1),
 * D = 128 causal. */
int ivlm_attention_f16(const void *q, const void *k, const void *v, void *o, const int64_t *strides_host, int B,
                       int H, int Sq, int Sk, int D, float scale, int causal, int q_pos0, const float *rel_h,
                       const float *rel_w, int rel_kh, int rel_kw, int kv_batch_div, int prescale_q,
                       ivlm_stream_t stream);
/* ... with q as hi + lo IEEE halves (q_lo: same strides as q; the IVLM_F16_SPLIT rows of the q projection) - q is the operand whose
 * rounding SAM's decomposed rel-pos terms amplify.  level 1: the lo half enters the rel-pos terms only (table mode: the kernel's
 * table product takes both halves; array mode: the caller computed rel_h / rel_w from hi + lo and this is ivlm_attention_f16);
 * level 2: Q.K^T takes both halves too and the softmax weights are split the same way for P.V (two MFMAs per fragment each).
 * k / v / o stay single fp16.  SAM's shapes: D = 80, prescale_q, no mask; windows in table mode (rel_w == NULL) or the 64 x 64
 * grid with the terms as arrays. */
int ivlm_attention_f16_qsplit(const void *q, const void *q_lo, const void *k, const void *v, void *o, const int64_t *strides_host,
                              int B, int H, int Sq, int Sk, int D, float scale, const float *rel_h, const float *rel_w, int rel_kh,
                              int rel_kw, int level, ivlm_stream_t stream);
/* "Parity" precision of the same operator (fp32-operand attention on the bf16 matrix cores): q / k / v / o are given as hi + lo
 * bf16 planes (x = hi + lo to 2^-17: the [hi | lo] halves of IVLM_BF16_SPLIT rows; the *_lo tensors use the strides of the hi
 * ones), both products run as three MFMAs per fragment (hi.hi + hi.lo + lo.hi), q * scale and the rel-pos bias stay fp32.
 * Shapes of the path only: D = 64 (plain), D = 80 with rel_h / rel_w (prescale_q = 1), D = 128 causal. */
int ivlm_attention_bf16_split(const void *q, const void *q_lo, const void *k, const void *k_lo, const void *v, const void *v_lo,
                              void *o, void *o_lo, const int64_t *strides_host, int B, int H, int Sq, int Sk, int D, float scale,
                              int causal, int q_pos0, const float *rel_h, const float *rel_w, int rel_kh, int rel_kw,
                              int kv_batch_div, int prescale_q, ivlm_stream_t stream);
/* The same operator with fp32 q / k / v / o and no operand rounding, for the SAM mask decoder's small attentions
 * (transformer.py:220-242: head dim 16 or 32; 9 tokens x 4096 image positions or the reverse) and the AttentionSplitter
 * (components.py:155-193: one head of 128 over V keys): scores = (q.k) * scale, fp32 softmax, fp32 P.V on the VALU.
 * D % 4 == 0, D <= 256 (16 / 32 take the fast kernels).  Strides as above but multiples of 4. */
int ivlm_attention_f32(const float *q, const float *k, const float *v, float *o, const int64_t *strides_host, int B, int H,
                       int Sq, int Sk, int D, float scale, int kv_batch_div, ivlm_stream_t stream);
/* Benchmark/test hook: -1 (default) picks per shape; 0 forces the 4-wave / 128-query block, 1 the 8-wave ping-pong block
 * (256 queries; one wave group on the matrix unit while the other does its softmax on the VALU). */
int ivlm_attention_pingpong(int mode);
/* Benchmark hook: 1 (default) = the table-mode kernel of the 64 x 64 grid maps all query blocks of a (view, head) to one XCD (its L2
 * fetches that head's K / V once); 0 = plain block order.  Returns the previous value. */
int ivlm_attention_xcd_map(int on);
/* Benchmark/test hook: 1 (default) = SAM's windows in table mode run on the whole-window kernel (one block per (window, head), K / V
 * of the window resident in LDS, one-pass softmax); 0 = the generic flash kernel for them too. */
int ivlm_attention_window_kernel(int v2);

/* add_decomposed_rel_pos operands (image_encoder.py:354-392), q_size == k_size == (SH,SW):
 *   rel_h[bh,q,kh] = q . rel_pos_h[qh-kh+SH-1],  rel_w[bh,q,kw] = q . rel_pos_w[qw-kw+SW-1]  (rounded to bf16
 *   like the reference's model-dtype einsum, stored f32).  tab_h bf16 [2*SH-1,D], tab_w bf16 [2*SW-1,D]. */
int ivlm_relpos_bias(const void *q, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void *tab_h, const void *tab_w,
                     int B, int H, int SH, int SW, int D, float *rel_h, float *rel_w, ivlm_stream_t stream);
/* "Parity" precision of the same operands: q as hi + lo bf16 planes (q_lo: same strides), results unrounded fp32 (D = 80).
 * q_lo == q: there is no lo plane (bf16 q), only the rounding of the results to bf16 is dropped. */
int ivlm_relpos_bias_split(const void *q, const void *q_lo, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void *tab_h,
                           const void *tab_w, int B, int H, int SH, int SW, int D, float *rel_h, float *rel_w,
                           ivlm_stream_t stream);
/* Second half of the GEMM formulation of the same operands: G bf16 [H][B*S][npad] = q . [rel_pos_h ; rel_pos_w]^T (one
 * batched ivlm_gemm_bf16 over the heads, K = head dim), g_head_stride = elements between heads; this gathers the Toeplitz
 * shifts rel_h[bh,s,kh] = G[qh-kh+SH-1], rel_w[bh,s,kw] = G[(2SH-1) + qw-kw+SW-1] into f32 [B*H,S,SH] / [B*H,S,SW]. */
int ivlm_relpos_gather(const void *G, int64_t g_head_stride, int npad, int B, int H, int SH, int SW, float *rel_h,
                       float *rel_w, ivlm_stream_t stream);
/* The same gather from an fp32 G (the product of fp16 q with the fp16 table, ivlm_gemm_bf16 with IVLM_GEMM_F16 and out_f32: the terms
 * add to the scores, so the fp16-operand path keeps them unrounded); g_head_stride in fp32 elements. */
int ivlm_relpos_gather_f32(const void *G, int64_t g_head_stride, int npad, int B, int H, int SH, int SW, float *rel_h,
                           float *rel_w, ivlm_stream_t stream);

/* The operands for fp16 q in ONE call: G = q . [rel_pos_h ; rel_pos_w]^T as a batched fp16 GEMM over the heads into the fp32 workspace
 * G_ws (>= H * B * S * npad * 4 bytes; the terms add to the scores and are not rounded), then the gather.  q_lo != NULL: q = hi + lo
 * IEEE halves (q_lo behind q in the same rows, e.g. the [hi | lo] halves of an IVLM_F16_SPLIT row: same strides) and the product
 * takes both.  cat16: fp16 [npad, D] = [rel_pos_h ; rel_pos_w ; zeros]; q rows of all (b, s) uniformly strided (q_bs == S * q_rs). */
int ivlm_relpos_bias_f16(const void *q, const void *q_lo, int64_t q_bs, int64_t q_hs, int64_t q_rs, const void *cat16, int npad, int B,
                         int H, int SH, int SW, int D, float *G_ws, size_t g_bytes, float *rel_h, float *rel_w, ivlm_stream_t stream);

/* One decode step of HF LlamaAttention with a KV cache, for the newest token only: rotate-half RoPE of q,k at
 * position pos, append k,v (rounded to bf16) to kcache/vcache [tmax,H,D], o = softmax(q.K[0..pos]^T * scale).V[0..pos].
 * qkv [3,H,D] (output of the fused q|k|v projection) and o [H,D] are bf16 or fp32 (io_dtype): with fp32 I/O q and the softmax
 * weights are not rounded (the decode path keeps fp32 activations), with bf16 they are rounded like the MFMA prefill path.
 * pos_dev != NULL: the position is read from device memory (one captured HIP graph then serves every decode step).
 * A position >= tmax (or >= 4096) is skipped: nothing is appended and o is written as zeros.  D <= 128. */
int ivlm_llama_decode_attn(const void *qkv, int io_dtype, void *kcache, void *vcache, int tmax, void *o, int H, int D, int pos,
                           const int32_t *pos_dev, float theta, float scale, const float *cos_tab, const float *sin_tab,
                           ivlm_stream_t stream);
/* The same against an IEEE fp16 KV cache (written by ivlm_rope_kv_f16): fp32 qkv / o, the new K / V rows appended as fp16. */
int ivlm_llama_decode_attn_f16(const void *qkv, void *kcache, void *vcache, int tmax, void *o, int H, int D, int pos,
                               const int32_t *pos_dev, float theta, float scale, const float *cos_tab, const float *sin_tab,
                               ivlm_stream_t stream);
/* Split-KV variant (fp32 qkv / o; cache_dtype IVLM_BF16 or IVLM_F16): the keys of a head are cut into S ranges (grid H x S, default
 * S = 8, A/B hook ivlm_llama_decode_attn_splits), each block publishes (o, max, sum) of its range and the block that arrives last for
 * a head merges them in range order - no block ever waits for another.  Same result as ivlm_llama_decode_attn[_f16] up to the fp32
 * summation order.  scratch: ivlm_llama_decode_attn_splitkv_scratch_bytes(H, D) bytes, 16-byte aligned, ZEROED once by the caller
 * (per-head arrival counters, left at zero by every launch), never shared by launches that can run concurrently. */
size_t ivlm_llama_decode_attn_splitkv_scratch_bytes(int H, int D);
int ivlm_llama_decode_attn_splitkv(const float *qkv, int cache_dtype, void *kcache, void *vcache, int tmax, float *o, int H, int D,
                                   int pos, const int32_t *pos_dev, float theta, float scale, const float *cos_tab,
                                   const float *sin_tab, void *scratch, size_t scratch_bytes, ivlm_stream_t stream);
int ivlm_llama_decode_attn_splits(int splits); /* 1..16 */
int ivlm_llama_decode_attn_batch_f16(const void *qkv, int64_t ldq, void *kcache, void *vcache, int64_t cache_stride, int tmax,
                                     void *o, int64_t ldo, int B, int H, int D, const int32_t *pos_dev, float theta, float scale,
                                     const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);

/* B sequences in one launch (grid H x B): sequence b reads qkv + b*ldq, appends to kcache/vcache + b*cache_stride
 * ([tmax,H,D] each), writes o + b*ldo and sits at position pos_dev[b] (strides in elements, multiples of 8); a sequence with
 * pos_dev[b] >= tmax is skipped (nothing appended) and its output row written as zeros.  The batched counterpart of the reference's padded-batch generate (model/InteractVLM.py:524-531 with B > 1 prompts). */
int ivlm_llama_decode_attn_batch(const void *qkv, int io_dtype, int64_t ldq, void *kcache, void *vcache, int64_t cache_stride,
                                 int tmax, void *o, int64_t ldo, int B, int H, int D, const int32_t *pos_dev, float theta,
                                 float scale, const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);

/* "Parity" precision of the two kernels above: fp32 qkv / o, and the cache holds K / V as hi + lo bf16 planes (kcache_lo /
 * vcache_lo: the layout of the hi caches) - the appended rows are not rounded to bf16, the cached ones are read as hi + lo. */
int ivlm_llama_decode_attn_split(const void *qkv, void *kcache, void *kcache_lo, void *vcache, void *vcache_lo, int tmax, void *o,
                                 int H, int D, int pos, const int32_t *pos_dev, float theta, float scale, const float *cos_tab,
                                 const float *sin_tab, ivlm_stream_t stream);
int ivlm_llama_decode_attn_batch_split(const void *qkv, int64_t ldq, void *kcache, void *kcache_lo, void *vcache, void *vcache_lo,
                                       int64_t cache_stride, int tmax, void *o, int64_t ldo, int B, int H, int D,
                                       const int32_t *pos_dev, float theta, float scale, const float *cos_tab,
                                       const float *sin_tab, ivlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stage-level entry points of the language path: C++ sequencers over the ops above, the same kernels in the same
 * order as interactvlm_amd/llava.py (bit-identical results).  Weights: a HOST array of per-layer DEVICE pointers, bf16, in the
 * layouts of the loader (qkv = q|k|v rows concatenated [3*hidden, hidden]; gu = gate/up rows interleaved [2*inter, hidden]).
 * kcache / vcache bf16 [layers, max_len, heads, head_dim]; cos / sin fp32 [max_len, head_dim/2] (ivlm_rope_table).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int layers, hidden, heads, inter, max_len;
    float eps, theta;
    int fuse_attn_oproj; /* decode step: attention + o_proj in ONE launch (o_proj blocks wait on device counters) when the grid
                          * fits the chip; 0 (default of the Python host) = separate launches - as fast since gemv1_kernel */
} ivlm_llama_cfg;
typedef struct {
    const void *ln1, *qkv, *o, *ln2, *gu, *down;
} ivlm_llama_layer;
/* HF LlamaModel.forward over T new positions pos0..pos0+T-1 with the KV cache (llava_llama.py:93-102): x_in fp32
 * [T, hidden] input embeddings -> hidden_out fp32 [T, hidden] after the final RMSNorm; appends K/V.  fp32 residual stream,
 * bf16 MFMA operands. */
size_t ivlm_llama_prefill_workspace_bytes(const ivlm_llama_cfg *cfg, int T);
int ivlm_llama_prefill(const ivlm_llama_cfg *cfg, const ivlm_llama_layer *layers_host, const void *final_norm, void *kcache,
                       void *vcache, const float *cos_tab, const float *sin_tab, const float *x_in, int T, int pos0,
                       float *hidden_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* One decode step (HF greedy search under InteractVLM.evaluate, model/InteractVLM.py:524-531): x_in fp32 [hidden] = embedding of
 * the newest token at position *pos_dev (device int32) -> hidden_out fp32 [hidden]; fp32 activations, exact products.
 * workspace: ivlm_llama_decode_workspace_bytes, ZEROED by the caller at the start of every generation (arrival counters and
 * the tokens-decoded word of the fused attention + o_proj launch; its int32 word [activations..] status stays 0 unless a bounded
 * device-side wait expired).  advance != 0: *pos_dev += 1 at the end of the step (graph-replay friendly). */
/* The prefill in the DEFAULT precision of the host model: IEEE fp16 MFMA operands; layers16_host = the same struct with qkv / o / gu /
 * down pointing to fp16 copies of the bf16 weights (ln1 / ln2 unchanged), kcache16 / vcache16 hold IEEE halves.  T > 16 (shorter
 * chunks: ivlm_llama_decode_step_f16kv token by token).  Workspace: ivlm_llama_prefill_workspace_bytes. */
int ivlm_llama_prefill_f16(const ivlm_llama_cfg *cfg, const ivlm_llama_layer *layers16_host, const void *final_norm, void *kcache16,
                           void *vcache16, const float *cos_tab, const float *sin_tab, const float *x_in, int T, int pos0,
                           float *hidden_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
size_t ivlm_llama_decode_workspace_bytes(const ivlm_llama_cfg *cfg);
int ivlm_llama_decode_step(const ivlm_llama_cfg *cfg, const ivlm_llama_layer *layers_host, const void *final_norm, void *kcache,
                           void *vcache, const float *cos_tab, const float *sin_tab, const float *x_in, int32_t *pos_dev,
                           int advance, float *hidden_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* The decode step against the fp16 KV cache of ivlm_llama_prefill_f16: bf16 weights (layers_host), fp32 activations, K / V rows appended
 * and read as IEEE halves; attention and o_proj as separate launches. */
int ivlm_llama_decode_step_f16kv(const ivlm_llama_cfg *cfg, const ivlm_llama_layer *layers_host, const void *final_norm, void *kcache16,
                                 void *vcache16, const float *cos_tab, const float *sin_tab, const float *x_in, int32_t *pos_dev,
                                 int advance, float *hidden_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* The decode step of the host model's default configuration: the four linears of every layer stream LOSSLESSLY packed weights
 * (ivlm_gemv1_bf12m: 1.5 bytes per weight, dots on the matrix cores; each matrix as built by the host: fragment-layout planes + per-row
 * exponent bases + CSR patches), fp32 activations, KV cache of IEEE halves (cache_dtype IVLM_F16: after ivlm_llama_prefill_f16) or
 * bf16 (IVLM_BF16: after ivlm_llama_prefill).  hidden % 64 == 0, inter % 64 == 0 (every matrix must take the fragment layout), else
 * IVLM_ERR_UNSUPPORTED.  Workspace: ivlm_llama_decode_workspace_bytes.  Same arithmetic as ivlm_llama_decode_step[_f16kv] up to the fp32
 * summation order. */
typedef struct {
    const void *Pf, *Ef;          /* fragment-layout planes (ivlm_gemv1_bf12m) */
    const int32_t *ebase;         /* [rows] */
    const int32_t *patch_ptr;     /* [rows + 1] */
    const int32_t *patch_col;
    const void *patch_val;        /* bf16 */
} ivlm_bf12m;
typedef struct {
    const void *ln1, *ln2;        /* bf16 [hidden] */
    ivlm_bf12m qkv, o, gu, down;  /* rows: 3 hidden | hidden | 2 inter (gate / up interleaved) | hidden */
} ivlm_llama_layer_bf12;
int ivlm_llama_decode_step_bf12(const ivlm_llama_cfg *cfg, const ivlm_llama_layer_bf12 *layers_host, const void *final_norm,
                                void *kcache, void *vcache, int cache_dtype, const float *cos_tab, const float *sin_tab,
                                const float *x_in, int32_t *pos_dev, int advance, float *hidden_out, void *workspace,
                                size_t workspace_bytes, ivlm_stream_t stream);
/* CLIPVisionTower.forward + feature_select('patch', layer -2) (clip_encoder.py:31-60): images bf16 [B,3,S,S] -> bf16
 * [B, tokens-1, hidden] (the mm_projector's operand).  layers_run = the encoder layers actually needed (23 of 24 for layer -2);
 * patch_w bf16 [hidden, kpad] (conv weight as GEMM rows, K zero-padded to kpad % 64 == 0), pos bf16 [tokens, hidden], cls_row
 * fp32 [hidden] = class_embedding + position_embedding[0].  qkv_w / qkv_b: q|k|v rows concatenated. */
typedef struct {
    int layers_run, hidden, heads, inter, image_size, patch, kpad, tokens;
    float eps;
} ivlm_clip_cfg;
typedef struct {
    const void *patch_w, *pos, *cls_row, *pre_ln_w, *pre_ln_b;
} ivlm_clip_head;
typedef struct {
    const void *ln1_w, *ln1_b, *qkv_w, *qkv_b, *out_w, *out_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} ivlm_clip_layer;
size_t ivlm_clip_encode_workspace_bytes(const ivlm_clip_cfg *cfg, int B);
int ivlm_clip_encode(const ivlm_clip_cfg *cfg, const ivlm_clip_head *head, const ivlm_clip_layer *layers_host, const void *images,
                     int B, void *features_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* The same stage in the DEFAULT precision of the host model: IEEE fp16 MFMA operands (fp16 LayerNorm / q|k|v / quick-GELU outputs,
 * fp16 attention).  layers16_host: the same struct with qkv_w / out_w / fc1_w / fc2_w pointing to fp16 copies (ivlm_bf16_to_f16) of
 * the bf16 weights, biases and LayerNorm weights unchanged.  features_split_out: bf16 [B, tokens-1, 2*hidden] = [hi | lo] rows of the
 * fp32 features (the mm_projector takes them with IVLM_GEMM_A_SPLIT).  Workspace: ivlm_clip_encode_workspace_bytes. */
int ivlm_clip_encode_f16(const ivlm_clip_cfg *cfg, const ivlm_clip_head *head, const ivlm_clip_layer *layers16_host, const void *images,
                         int B, void *features_split_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);

/* ImageEncoderViT.forward (image_encoder.py:110-125; Block :177-193, Attention :235-260, window partition :263-318, decomposed
 * rel-pos :354-392, neck :92-108): images bf16 [V,3,img,img] -> embeddings fp32 [V, grid*grid, out_chans] (channels last).
 * Conv weights in GEMM layout: patch_w [D, 3*p*p], neck0_w [OC, D], neck2_w [OC, (ky,kx,OC)]; rel_cat = [rel_pos_h ; rel_pos_w]
 * zero-padded to a multiple of 64 rows (the rel-pos GEMM of the global blocks; the attention kernel's table mode for the windows). */
typedef struct {
    int embed_dim, depth, heads, grid, window, patch, img_size, out_chans, mlp_dim;
} ivlm_sam_cfg;
typedef struct {
    const void *patch_w, *patch_b, *pos_embed, *neck0_w, *neck1_w, *neck1_b, *neck2_w, *neck3_w, *neck3_b;
} ivlm_sam_head;
typedef struct {
    const void *norm1_w, *norm1_b, *qkv_w, *qkv_b, *rel_h, *rel_w, *rel_cat, *proj_w, *proj_b, *norm2_w, *norm2_b, *lin1_w, *lin1_b,
        *lin2_w, *lin2_b;
    int global_attn;
} ivlm_sam_block;
size_t ivlm_sam_encode_workspace_bytes(const ivlm_sam_cfg *cfg, int V);
int ivlm_sam_encode(const ivlm_sam_cfg *cfg, const ivlm_sam_head *head, const ivlm_sam_block *blocks_host, const void *images, int V,
                    float *embeddings_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* The same stage in "parity" precision (the mode that holds 1e-3 against the reference's fp32 path at the real depth): no
 * activation is rounded to bf16 - split LayerNorm outputs, split-operand / split-output GEMMs, split-operand attention with fp32
 * rel-pos terms.  Same structs, its own workspace size (head dim 80). */
size_t ivlm_sam_encode_parity_workspace_bytes(const ivlm_sam_cfg *cfg, int V);
int ivlm_sam_encode_parity(const ivlm_sam_cfg *cfg, const ivlm_sam_head *head, const ivlm_sam_block *blocks_host, const void *images,
                           int V, float *embeddings_out, void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* ... with the two MLP GEMMs of every block on IEEE fp16 operands (norm2 and the GELU epilogue write halves: 11 significant bits, an
 * eighth of the bf16 rounding error, ONE MFMA pass instead of two): the encoder of the "parity-encoder" mode - measured 4.6e-4
 * end to end at depth 32 (the all-split stage: 4.0e-4) for 12 ms less per 4 views.  mlp16_host[l] = fp16 copies of block l's
 * lin1_w / lin2_w (ivlm_bf16_to_f16; exact for |w| >= 2^-14, within 2^-25 below).  Workspace: the parity size. */
typedef struct {
    const void *lin1_w16, *lin2_w16;
} ivlm_sam_mlp_f16;
int ivlm_sam_encode_parity_f16mlp(const ivlm_sam_cfg *cfg, const ivlm_sam_head *head, const ivlm_sam_block *blocks_host,
                                  const ivlm_sam_mlp_f16 *mlp16_host, const void *images, int V, float *embeddings_out,
                                  void *workspace, size_t workspace_bytes, ivlm_stream_t stream);
/* The stage in the DEFAULT precision of the host model - IEEE fp16 MFMA operands in one pass (an eighth of the bf16 operand rounding at
 * the same matrix-core rate) with the q path exact: norm1 as [hi | lo] halves, q = W_q . (hi + lo) from its own GEMM as [hi | lo]
 * halves, its lo half in the rel-pos table products of the attention kernels (fp32 rel-pos terms), k | v / proj / mlp on single fp16
 * operands, the neck on hi + lo bf16 operands.  4 - 6e-4 end to end against the fp32 oracle at depth 32 (bf16 operands: 6e-3 .. 1e-2)
 * for ~10 % more encoder time.  blocks16_host[l] = fp16 copies (ivlm_bf16_to_f16) of block l's qkv_w / proj_w / lin1_w / lin2_w, of
 * its q|k|v bias and of rel_cat.  SAM ViT shapes (head dim 80, 64 x 64 grid, windows of 2 * side <= 32). */
typedef struct {
    const void *qkv_w16, *proj_w16, *lin1_w16, *lin2_w16, *qkv_b16, *rel_cat16;
} ivlm_sam_block_f16;
size_t ivlm_sam_encode_f16_workspace_bytes(const ivlm_sam_cfg *cfg, int V);
int ivlm_sam_encode_f16(const ivlm_sam_cfg *cfg, const ivlm_sam_head *head, const ivlm_sam_block *blocks_host,
                        const ivlm_sam_block_f16 *blocks16_host, const void *images, int V, float *embeddings_out, void *workspace,
                        size_t workspace_bytes, ivlm_stream_t stream);
/* bf16 -> IEEE fp16 (round to nearest even; |x| > 65504 -> inf), n elements: weight copies for IVLM_GEMM_F16 / the *_f16 stages */
int ivlm_bf16_to_f16(const void *src_bf16, void *dst_f16, int64_t n, ivlm_stream_t stream);

/* PromptEncoder.forward(text_embeds) + MaskDecoder.forward(multimask_output=False) (prompt_encoder.py:140-186,
 * mask_decoder.py:75-164, transformer.py:62-242) with fp32 activations end to end: image_embeddings fp32 [V, grid*grid, C]
 * (ivlm_sam_encode's output), text_embeds fp32 [n_text, C] (the view-conditioned [SEG] embeddings, one token per view) ->
 * low_res fp32 [V, 4*grid, 4*grid] (mask token 0), iou fp32 [V, n_mask] (column 0 belongs to that mask).
 * Every linear is an ivlm_lin: w2 = [W | W] bf16 [n, 2k] (the fp32 input arrives as hi | lo bf16 rows), b bf16 [n].
 * Transposed convs as GEMMs: up0 = [(dy,dx,co), ci | same], bias tiled over (dy,dx).  no_mask fp32 [C], key_pe fp32
 * [grid*grid, C] (ivlm_dense_pe), out_tokens fp32 [5, C] = [iou_token ; mask_tokens]. */
typedef struct {
    const void *w2, *b;
    int n, k;
} ivlm_lin;
typedef struct {
    ivlm_lin q, k, v, o;
} ivlm_dec_attn;
typedef struct {
    ivlm_dec_attn self_attn, t2i, i2t;
    const void *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b, *norm4_w, *norm4_b;
    ivlm_lin lin1, lin2;
} ivlm_dec_layer;
typedef struct {
    int C, heads, depth;
    const void *no_mask, *key_pe, *out_tokens;
    ivlm_dec_layer layers[4];
    ivlm_dec_attn final_attn;
    const void *norm_final_w, *norm_final_b, *up_ln_w, *up_ln_b;
    ivlm_lin up0, up1, hyper[3], iou[3];
} ivlm_sam_dec;
size_t ivlm_sam_decode_workspace_bytes(int V, int grid, int C, int n_text, int mlp_dim);
int ivlm_sam_decode(const ivlm_sam_dec *weights, int V, int grid, int n_text, const float *image_embeddings,
                    const float *text_embeds, float *low_res_out, float *iou_out, void *workspace, size_t workspace_bytes,
                    ivlm_stream_t stream);

/* The split-K rule of the small-M tile GEMMs (number of K slices, 1 = none): shared by the sequencers and the Python host. */
int ivlm_gemm_splitk_choice(int M, int N, int K, int act, int has_rms);
/* The split-K products of the stage sequencers (ivlm_clip_encode, ivlm_llama_prefill, ivlm_sam_decode and their _f16 forms) take the
 * two-launch form (partials + ordered reduction) by default.  on = 1 selects the reduction FUSED into the GEMM launch
 * (ivlm_gemm_bf16_splitk_fused: same values bit for bit, one launch less, measured 2 - 3 x slower on MI355X; the arrival counters then
 * live in the last 16 KB of the sequencer's split-K region and every stage call zeroes them once); on = 0 switches back; any other
 * value only queries.  Default: the environment's IVLM_SPLITK_FUSED=1 at first use, else off.  Returns the previous setting. */
int ivlm_stages_splitk_fused(int on);

/* Batch-1 decode linear over LOSSLESSLY packed bf16 weights ("bf12", 1.5 bytes per weight instead of 2; the decode step of
 * InteractVLM.evaluate's greedy search, model/InteractVLM.py:524-531, is pure weight streaming).  A row of W [N, K] is stored as
 *   P  u8 [N, ldp >= K]      sign << 7 | mantissa (7 bits)
 *   E  u8 [N, lde >= K/2]    two 4-bit codes per byte (low nibble = even column): exponent field - ebase[row] in 1 .. 15;
 *                            0 = the weight is zero or outside the row's window (then P = 0)
 *   ebase i32 [N]            row maximum of the exponent field - 15, clamped at 0
 *   patch_ptr i32 [N+1], patch_col i32, patch_val bf16: CSR of the nonzero weights outside the window, with their exact values
 * - every weight is reconstructed bit for bit (ivlm_unpack_bf12).  y = act(W . x (* rsqrt(mean(x^2) + eps) with rms_w: the fused
 * RMSNorm of ivlm_gemm_bf16's M = 1 path) + bias) + residual; x fp32 [K], exact bf16 x fp32 products, fp32 accumulation; act as
 * ivlm_gemm_bf16 (SWIGLU: rows interleaved (gate_j, up_j), out [N/2]); flags: IVLM_GEMM_RES_F32.  K % 16 == 0, K <= 15360. */
int ivlm_gemv1_bf12(const float *x, const void *P, int64_t ldp, const void *E, int64_t lde, const int32_t *ebase,
                    const int32_t *patch_ptr, const int32_t *patch_col, const void *patch_val, void *C, const void *bias,
                    const void *residual, int N, int K, int act, int out_f32, const void *rms_w, float rms_eps, int flags,
                    ivlm_stream_t stream);
/* The same product with the dots on the matrix cores: Pf / Ef hold the bytes of P / E in the order the lanes of
 * v_mfma_f32_16x16x32_bf16 consume them - [N/16][K/64][64 lanes][16 | 8 bytes], lane = q*16 + r, byte h*8 + i of a lane = weight
 * k = sp*64 + h*32 + q*8 + i of row rb*16 + r (E: two codes per byte, low nibble = even k) - the weights are the B operand, x enters
 * as three bf16 rows hi + lo + lo2 (24 significant bits: the fp32 activation exactly), fp32 accumulation.  Two VALU operations per
 * weight instead of five.  N % 16 == 0, K % 64 == 0, K <= 17066 (6 K bytes of LDS); x 16-byte aligned. */
int ivlm_gemv1_bf12m(const float *x, const void *Pf, const void *Ef, const int32_t *ebase, const int32_t *patch_ptr,
                     const int32_t *patch_col, const void *patch_val, void *C, const void *bias, const void *residual, int N, int K,
                     int act, int out_f32, const void *rms_w, float rms_eps, int flags, ivlm_stream_t stream);
/* The o_proj of a decode step with the split-KV attention's merge in its prologue: the activation row x[h][d] = sum_s e^(m_s - M)
 * o_s[d] / sum_s e^(m_s - M) l_s is computed, while it is staged, from parts[K / D heads][4 ranges][D + 4] fp32 (o unnormalised | max |
 * sum | pad) written by ivlm_llama_decode_attn_parts - 128 attention blocks instead of 32 (a head's K / V rows are otherwise streamed by
 * ONE CU at its ~25 GB/s), no merge launch, no counters: the kernel boundary is the synchronisation. */
int ivlm_llama_decode_attn_parts(const float *qkv, int cache_dtype, void *kcache, void *vcache, int tmax, float *parts, int H, int D,
                                 int pos, const int32_t *pos_dev, float theta, float scale, const float *cos_tab, const float *sin_tab,
                                 ivlm_stream_t stream);
int ivlm_gemv1_bf12m_parts(const float *parts, int D, const void *Pf, const void *Ef, const int32_t *ebase, const int32_t *patch_ptr,
                           const int32_t *patch_col, const void *patch_val, void *C, const void *bias, const void *residual, int N, int K,
                           int act, int out_f32, int flags, ivlm_stream_t stream);
/* The packer (weight preparation, once per matrix): bf16 w [N_valid, K] -> what ivlm_gemv1_bf12m / ivlm_gemv16_bf12m /
 * ivlm_llama_decode_step_bf12 read.  n_rows = N_valid rounded up to a multiple of 16 (the extra rows pack as zeros); K % 64 == 0.
 *   1. ivlm_pack_bf12m_count: ebase [n_rows], patch_ptr [n_rows + 1] (row r's patches are entries patch_ptr[r] .. patch_ptr[r + 1]);
 *   2. the caller reads patch_ptr[n_rows] (the number of nonzero weights outside their row's 15-binade window: ~1e-4 of a trained
 *      matrix), allocates patch_col (int32) / patch_val (bf16) of max(1, that) entries and the planes Pf (n_rows * K bytes, 16-byte
 *      aligned) / Ef (n_rows * K / 2 bytes);
 *   3. ivlm_pack_bf12m_fill writes the planes in the fragment layout and the patches in column order.
 * Lossless: ivlm_unpack_bf12 on the row-layout view of the same bytes returns every bf16 value bit for bit (-0.0 packs as +0.0). */
int ivlm_pack_bf12m_count(const void *w, int N_valid, int n_rows, int K, int32_t *ebase, int32_t *patch_ptr, ivlm_stream_t stream);
int ivlm_pack_bf12m_fill(const void *w, int N_valid, int n_rows, int K, const int32_t *ebase, const int32_t *patch_ptr, void *Pf, void *Ef,
                         int32_t *patch_col, void *patch_val, ivlm_stream_t stream);
/* The linears of the BATCHED decode step (M <= 16 fp32 activation rows: one token of each sequence) on the same planes: the weight
 * fragments are rebuilt once and meet all M rows (x = hi + lo bf16 operands to 2^-17, two MFMAs per fragment, fp32 accumulation);
 * blocks of 1 - 3 tiles of 16 weight rows x 8 waves over K.  out[M, N] = act(x . W^T + bias) + residual; N % 16 == 0, K % 64 == 0. */
int ivlm_gemv16_bf12m(const float *x, int64_t lda, int M, const void *Pf, const void *Ef, const int32_t *ebase,
                      const int32_t *patch_ptr, const int32_t *patch_col, const void *patch_val, void *C, int64_t ldc, const void *bias,
                      const void *residual, int64_t ldr, int N, int K, int act, int out_f32, const void *rms_w, float rms_eps, int flags,
                      ivlm_stream_t stream);
void ivlm_gemv16_bf12m_tuning(int tiles_per_block); /* A/B hook: 1 - 3, 0 = automatic */
int ivlm_decode_parts_tuning(int ranges); /* A/B hook: 2 or 4 (default) key ranges per head for the pair above */
/* A/B hook: grids of at most this many 16-row blocks run 16 waves per block (default 256 = one block per CU), larger ones 8;
   a negative value: the same limit without the 8-deep prefetch of long rows. */
void ivlm_gemv1_bf12m_tuning(int wide_max_blocks);
/* The packed matrix (row layout) back as bf16 [N, K] (the losslessness check; not on the path). */
int ivlm_unpack_bf12(const void *P, int64_t ldp, const void *E, int64_t lde, const int32_t *ebase, const int32_t *patch_ptr,
                     const int32_t *patch_col, const void *patch_val, int N, int K, void *w_out, ivlm_stream_t stream);

/* torch.argmax(logits, -1) of HF greedy search (first maximal index); x f32 [rows, cols] -> out i32 [rows]. */
int ivlm_argmax_f32(const float *x, int rows, int cols, int32_t *out, ivlm_stream_t stream);
/* ... and bump[row] += 1 in the same launch (bump i32 [rows], may be NULL): the device-side positions of a decode graph. */
int ivlm_argmax_f32_bump(const float *x, int rows, int cols, int32_t *out, int32_t *bump, ivlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data movement on the path (all bf16, 16-byte granules: cols/strides % 8 == 0)
 * ------------------------------------------------------------------------------------------- */
/* Conv2d(kernel = stride = ks) as GEMM operand: out[(b,gy,gx), (c,ky,kx)] zero-padded to Kpad columns
 * (SAM PatchEmbed image_encoder.py:404-426; HF CLIP patch_embedding). x [B,C,H,W] */
int ivlm_im2col_nchw(const void *x, void *out, int B, int C, int H, int W, int ks, int stride, int Kpad,
                     ivlm_stream_t stream);
/* 3x3 / pad 1 conv operand from channels-last x [B,H,W,C] -> [(b,y,x), (ky,kx,c)] (SAM neck, image_encoder.py:92-108) */
int ivlm_im2col3x3_nhwc(const void *x, void *out, int B, int H, int W, int C, ivlm_stream_t stream);
/* the same with pixel stride ldx >= C of x and row stride ldo >= 9C of out (elements, multiples of 8): the hi and the lo half of
 * IVLM_BF16_SPLIT rows are unfolded separately into the two halves of a [rows, 2 * 9C] split operand */
int ivlm_im2col3x3_nhwc_strided(const void *x, int64_t ldx, void *out, int64_t ldo, int B, int H, int W, int C,
                                ivlm_stream_t stream);
/* dst[r] = (idx ? (idx[r] >= 0 ? src[idx[r]] : 0) : src[r]) + (add ? add[r] : 0): window_partition / window_unpartition +
 * shortcut (image_encoder.py:263-318, 177-193), embed_tokens gather (llava_arch.py:185-208), dtype conversion.  src / add
 * bf16 or fp32, dst bf16, fp32, IVLM_BF16_SPLIT (row stride ldd >= 2*cols) or IVLM_FP8 (bytes of x / *fp8_scale, clamped to
 * +-448; ldd in bytes). */
int ivlm_gather_rows(void *dst, int dst_kind, int64_t ldd, const void *src, int src_dtype, int64_t lds, const int32_t *idx,
                     const void *add, int add_dtype, int64_t lda, int64_t rows, int cols, const float *fp8_scale,
                     ivlm_stream_t stream);
/* *out = max(*out, max |x|) over n elements (n % 8 == 0; *out >= 0 set by the caller): calibration of per-tensor fp8 scales */
int ivlm_amax(const void *x, int dtype, int64_t n, float *out, ivlm_stream_t stream);
/* out[r] = a[r] (op 0: +, op 1: *) b[r % b_rows]  (queries + query_pe, keys + key_pe: transformer.py:160-176;
 * [SEG] embedding * view encoding: InteractVLM.py:275-282); a / b bf16 or fp32, out bf16, fp32 or IVLM_BF16_SPLIT
 * (dense rows of 2*cols). */
int ivlm_add_rows(void *out, int out_kind, const void *a, int a_dtype, const void *b, int b_dtype, int64_t rows, int cols,
                  int64_t b_rows, int op, ivlm_stream_t stream);
/* dst[idx[r]] = row for r < n_idx (bf16, cols % 8 == 0): the q|k|v rows of SAM's zero-padded window positions are the bias
 * alone (image_encoder.py:179-183, 222-243), so the qkv GEMM computes the real rows only and these are filled. */
int ivlm_fill_rows(void *dst, int64_t ldd, const int32_t *idx, int64_t n_idx, const void *row, int cols,
                   ivlm_stream_t stream);
/* PositionEmbeddingRandom.forward (prompt_encoder.py:219-229): gauss f32 [2,F] -> pe bf16 | fp32 [h*w, 2F]
 * (the table is a constant of the weights: computed once at load, in fp32) */
int ivlm_dense_pe(const void *gauss, void *pe, int pe_dtype, int h, int w, int F, ivlm_stream_t stream);
/* HF LlamaAttention rotary (rotate-half, base theta) applied in place to q,k of qkv [T,3,H,D] (row stride ld) at
 * positions pos0+t, and KV-cache append (kcache/vcache [Tmax,H,D], may be NULL). */
int ivlm_rope_kv(void *qkv, int64_t ld, int T, int H, int D, int pos0, float theta, void *kcache, void *vcache,
                 const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);
/* The same on IEEE fp16 qkv rows and fp16 caches (the fp16-operand prefill: q | k | v from ivlm_gemm_bf16 with IVLM_GEMM_OUT_F16). */
int ivlm_rope_kv_f16(void *qkv, int64_t ld, int T, int H, int D, int pos0, float theta, void *kcache, void *vcache,
                     const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);
/* "Parity" precision of ivlm_rope_kv: qkv rows are IVLM_BF16_SPLIT ([hi(3HD) | lo(3HD)], ld >= 6HD), rotated in fp32 on hi + lo
 * and written back as hi + lo; K / V are appended to hi + lo cache planes ([Tmax,H,D] each; all four or none).  cos / sin tables
 * are required. */
int ivlm_rope_kv_split(void *qkv, int64_t ld, int T, int H, int D, int pos0, void *kcache, void *kcache_lo, void *vcache,
                       void *vcache_lo, const float *cos_tab, const float *sin_tab, ivlm_stream_t stream);
/* Decode attention (as ivlm_llama_decode_attn with fp32 I/O and a device position) and the o_proj GEMV + residual of the same
 * layer in ONE launch: x_out[hidden] = x + W_o . attention(qkv), qkv / x / x_out fp32 (fp32 residual stream), W_o bf16.  The
 * o_proj blocks stream their weight rows while the attention blocks run and wait for them on `counter` (int32, zeroed by the
 * caller at the start of a generation; `step_dev` = tokens decoded so far, incremented by the caller after each token;
 * target = H * (step + 1)).  attn_scratch: hidden fp32.  status[0] != 0 after the stream drained = a bounded wait expired
 * (results invalid).  hidden = H*D in {512, 1024, 4096, 5120}; the grid (H + hidden/32 blocks) must fit the CUs. */
int ivlm_llama_attn_oproj(const float *qkv, void *kcache, void *vcache, int tmax, float *attn_scratch, const void *wo,
                          const float *x, float *x_out, int H, int D, float theta, float scale, const float *cos_tab,
                          const float *sin_tab, const int32_t *pos_dev, const int32_t *step_dev, int32_t *counter,
                          int32_t *status, ivlm_stream_t stream);
/* fp32 rotary tables cos/sin [T, D/2] (optional inputs of ivlm_rope_kv / ivlm_llama_decode_attn; NULL = compute) */
int ivlm_rope_table(float *cos_tab, float *sin_tab, int T, int D, float theta, ivlm_stream_t stream);
/* Caller-side image preprocessing (run_demo.py:65-79 `preprocess`: (x - mean)/std then zero-pad to the square model
 * input; HF CLIPImageProcessor: centre crop, 1/255 rescale, normalise): src u8 [H,W,3] RGB on the device, crop
 * (y0,x0,ch,cw) -> out bf16|f32 [3,OH,OW].  mean3/std3 are HOST pointers in 0..255 units. */
int ivlm_normalize_pad_u8(const uint8_t *src, int H, int W, int y0, int x0, int ch, int cw, const float *mean3_host,
                          const float *std3_host, void *out, int out_bf16, int OH, int OW, ivlm_stream_t stream);
/* masks = hyper_in @ upscaled_embedding (mask_decoder.py:150-153) for one mask token: up
 * [B,gh,gw,2,2,2,2,C] (output of the two k2s2 transposed convs, channels last), hyper [B,C], both bf16 or both fp32 (dtype)
 * -> low f32 [B,4gh,4gw] */
int ivlm_mask_dot(const void *up, const void *hyper, int dtype, float *low, int B, int gh, int gw, int C,
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
