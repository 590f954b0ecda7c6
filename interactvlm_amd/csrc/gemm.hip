// bf16 MFMA GEMM for gfx950:  C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual)
//
// This one kernel carries every dense contraction of the hot path (reference call sites):
//   SAM ViT-H qkv/proj/mlp      model/segment_anything/modeling/image_encoder.py:222-260, common.py:13-27
//   CLIP ViT-L/14 (HF)          model/llava/model/multimodal_encoder/clip_encoder.py:41-57
//   LLaMA q/k/v/o/gate/up/down  model/llava/model/language_model/llava_llama.py:93-102 (HF LlamaModel)
//   mm_projector, text_hidden_fcs, SAM decoder linears (transformer.py:185-242, mask_decoder.py:169-191)
//
// Design (CDNA4, wave64):
//   * 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of
//     v_mfma_f32_16x16x32_bf16, fp32 accumulators in registers.
//   * both operands are K-contiguous ([rows][K]), so each MFMA fragment is one ds_read_b128.
//   * global -> LDS by direct DMA (global_load_lds_dwordx4, 1 KiB per wave instruction), double
//     buffered; the LDS image is lane-linear, so the bank-conflict-free XOR swizzle is applied to
//     the per-lane SOURCE address and again on the fragment read (same involution).
//       16-B chunk c of row r lives at chunk  c ^ ((r >> 1) & 7)   of the 128-B LDS row.
//   * operands are swapped (D = W_tile . A_tile^T) so that a lane's 4 accumulator registers are 4
//     consecutive N of one output row: bias/activation/residual and an 8-byte (bf16x4) store.
//   * epilogues: bias, GELU(erf) / quick-GELU / ReLU / SiLU, SwiGLU over interleaved gate/up rows,
//     residual add (optionally row-modulo for broadcast tables), bf16 or fp32 output.
#include "gemm_common.h"

namespace ivlm {
namespace {

constexpr int BK = 64;

// Tile configuration: WM x WN waves, each owning (MI*16) x (NI*16) of the output.
//   <2,2,4,4>: 128x128 block, 256 threads, 64 KiB LDS, 2 blocks/CU  (small / skinny problems)
//   <2,4,8,4>: 256x256 block, 512 threads, 128 KiB LDS, 1 block/CU  (per-wave 128x64: 32 MFMAs per 12 fragment
//              reads instead of 16 per 8, and half the global->LDS bytes per FLOP)
template <int WM_, int WN_, int MI_, int NI_, int MINB_ = 2>
struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_;
    static constexpr int kMinBlocks = MINB_;  // launch bound: resident blocks per CU the register budget is sized for
    static constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    static constexpr int kWaves = WM * WN, kThreads = kWaves * 64;
    static constexpr int kTileBytesA = BM * BK * 2, kTileBytesW = BN * BK * 2;
    static constexpr int kStageBytes = kTileBytesA + kTileBytesW;
    static constexpr int kLdsBytes = 2 * kStageBytes;
    // 1-KiB DMA pieces (8 rows) per wave per tile; a tile whose piece count is not a multiple of the wave count (176 rows = 22
    // pieces over 4 waves) gives its last waves one piece less (guarded where the pieces are addressed and issued)
    static constexpr int PA = (BM / 8 + kWaves - 1) / kWaves, PW = (BN / 8 + kWaves - 1) / kWaves;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "whole 8-row pieces");
};

// OPK: operand kind - 0 bf16, 1 e4m3 (MX instruction), 2 IEEE fp16 (a compile-time choice: a wave-uniform runtime branch around the
// MFMA groups was measured to halve the speed of EVERY GEMM - the accumulators no longer stay put across it)
template <int ACT, bool OUT_F32, typename CFG, int OPK = 0>
__global__ __launch_bounds__(CFG::kThreads, CFG::kMinBlocks) void gemm_bf16_kernel(GemmArgs g) {
    constexpr bool FP8 = OPK == 1;
    constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI, NI = CFG::NI, PA = CFG::PA, PW = CFG::PW;
    constexpr int kTileBytesA = CFG::kTileBytesA, kStageBytes = CFG::kStageBytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware, grouped tile raster: every XCD (private 4 MiB L2) gets one contiguous range of the tile sequence,
    // ordered in groups of 8 tile-rows with the row fastest, so the ~64 tiles it runs concurrently form an ~8x8
    // patch sharing 8 A + 8 W panels per K step (measured: fabric traffic halved, profiles/r01_pmc_traffic*.txt).
    int m0, n0;
    gemm_tile_origin(g, BM, BN, m0, n0);
    const int bz = blockIdx.z;
    const bf16_t* __restrict__ A = g.A + (int64_t)bz * g.strideA;
    const bf16_t* __restrict__ W = g.W + (int64_t)bz * g.strideW;

    // ---- staging addresses: wave w copies pieces w*4 .. w*4+3 (8 rows each) of both tiles ----
    const bf16_t* srcA[PA];
    const bf16_t* srcW[PW];
    int kcolA[PA], kcolW[PW];  // first K index of the chunk this lane copies (per piece)
    // row stride / K-tile step of each operand: (lda, 64) row-major, (64, kstep) in the K-panel layout
    const int64_t a_rs = g.a_kstep ? 64 : g.lda, a_ks = g.a_kstep ? g.a_kstep : BK;
    const int64_t w_rs = g.w_kstep ? 64 : g.ldw, w_ks = g.w_kstep ? g.w_kstep : BK;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 8 + (lane >> 3);  // (rows past BM belong to no piece: never issued, see stage())
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source chunk that lands in LDS chunk lane&7
        kcolA[i] = chunk * 8;
        int ra = m0 + row;
        ra = ra < g.M ? ra : g.M - 1;
        if (g.a_rows) ra = g.a_rows[ra];
        srcA[i] = A + (int64_t)ra * a_rs + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = (wave * PW + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        kcolW[i] = chunk * 8;
        int rw = n0 + row;
        rw = rw < g.N ? rw : g.N - 1;
        srcW[i] = W + (int64_t)rw * w_rs + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* baseA = smem + buf * kStageBytes + wave * PA * 1024;
        unsigned char* baseW = smem + buf * kStageBytes + kTileBytesA + wave * PW * 1024;
        // split A ([hi | lo] halves): K tile kt pairs W tile kt / 2 with the hi (even kt) or the lo (odd kt) tile of A
        const int kw = g.a_split ? kt >> 1 : kt;
        const int64_t alo = (g.a_split && (kt & 1)) ? g.a_lo : 0;
        const int koff = kw * BK;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(kGemmZeroChunk);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if ((BM / 8) % CFG::kWaves == 0 || (wave * PA + i) * 8 < BM)
                glds16(kcolA[i] + koff < g.K ? srcA[i] + kw * a_ks + alo : zero, baseA + i * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i)
            if ((BN / 8) % CFG::kWaves == 0 || (wave * PW + i) * 8 < BN)
                glds16(kcolW[i] + koff < g.K ? srcW[i] + kw * w_ks : zero, baseW + i * 1024);
    };

    // ---- fragment read offsets (bytes within a tile), constant over the K loop ---------------
    const int wm = wave % CFG::WM, wn = wave / CFG::WM;
    int offA[MI], offW[NI];  // for k-step 0; k-step 1 flips chunk bit 2 (chunk ^ 4)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wm * (MI * 16) + i * 16 + (lane & 15);
        offA[i] = r * 128 + ((((lane >> 4)) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rn = wn * (NI * 16) + i * 16 + (lane & 15);
        offW[i] = rn * 128 + ((((lane >> 4)) ^ ((rn >> 1) & 7)) << 4);
    }

    f32x4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = ((g.K + BK - 1) / BK) << (g.a_split ? 1 : 0);
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();  // tile t landed (vmcnt(0) + barrier); everyone is done with tile t-1
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const unsigned char* ta = smem + (t & 1) * kStageBytes;
        const unsigned char* tw = ta + kTileBytesA;
        if (FP8) {  // one e4m3 16x16x128 step per K tile: the two bf16 k-step fragments of a lane, taken as 32 bytes
            bf16x8_t fa[MI][2], fw[NI][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < NI; ++i) fw[i][kk] = *reinterpret_cast<const bf16x8_t*>(tw + (offW[i] ^ (kk << 6)));
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][kk] = *reinterpret_cast<const bf16x8_t*>(ta + (offA[i] ^ (kk << 6)));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = mfma_fp8_128(fw[ni][0], fw[ni][1], fa[mi][0], fa[mi][1], acc[ni][mi]);
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[MI], fw[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = gemm_frag_read(tw + (offW[i] ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = gemm_frag_read(ta + (offA[i] ^ (kk << 6)));
#ifdef IVLM_ABL_NOMFMA
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) IVLM_ABL_MFMA_USE(fw[ni], fw[ni]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) IVLM_ABL_MFMA_USE(fa[mi], fa[mi]);
            continue;
#endif
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if constexpr (OPK == 2)  // IEEE-half operands: same tiles and layouts, the f16 instruction
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fw[ni]),
                                                                             __builtin_bit_cast(f16x8_t, fa[mi]), acc[ni][mi], 0, 0, 0);
                    else
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
                }
        }
    }

#ifdef IVLM_ABL_NOEPI
    if (g.M > 0) {  // (one dummy store per lane keeps the accumulators live)
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sacc == 123.456f) static_cast<float*>(g.C)[tid] = sacc;
        return;
    }
#endif
    // ---- epilogue: whole lines through LDS where the wave's rows are 128 / 256 bytes (gemm_common.h) ----------------
    bool stored = false;
    {
        constexpr int kRowBytes = NI * 16 * (OUT_F32 ? 4 : 2);
        if constexpr (kRowBytes == 128 || kRowBytes == 256) {
            constexpr int kBudget = CFG::kLdsBytes / CFG::kWaves;  // this wave's slice of the (now free) tile buffers
            constexpr int kFit = kBudget / (16 * kRowBytes);
            constexpr int kPass = kFit >= MI ? MI : (kFit >= 4 ? 4 : (kFit >= 2 ? 2 : (kFit >= 1 ? 1 : 0)));
            if constexpr (kPass > 0 && MI % kPass == 0) {
                if (gemm_whole_lines_ok<OUT_F32>(g, ACT)) {
                    __syncthreads();  // every wave is done with the last K tile
                    gemm_store_lines<ACT, OUT_F32, MI, NI, kPass>(g, smem + wave * kBudget, m0 + wm * (MI * 16), n0 + wn * (NI * 16),
                                                                  lane, acc);
                    stored = true;
                }
            }
        }
    }
    if (!stored) {  // direct fragment stores: lane holds C[m][n..n+3], m = l&15, n = (l>>4)*4
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * (MI * 16) + mi * 16 + (lane & 15);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = n0 + wn * (NI * 16) + ni * 16 + (lane >> 4) * 4;
                gemm_epilogue4<ACT, OUT_F32>(g, bz, m, n, acc[ni][mi]);
            }
        }
    }
    // fused split-K (the fp32 partial-tile instantiations only): count this block's arrival, the tile's last block reduces it
    if constexpr (ACT == ACT_NONE && OUT_F32 && OPK != 1) {
        if (g.sk.count) splitk_fixup<BM, BN, CFG::kThreads>(g, m0, n0);
    }
}

using Cfg128 = TileCfg<2, 2, 4, 4>;
using Cfg256 = TileCfg<2, 4, 8, 4>;

template <int ACT, bool OUT_F32, typename CFG, int OPK = 0>
int launch_cfg(const GemmArgs& g, hipStream_t st) {
    const int tiles = ((g.M + CFG::BM - 1) / CFG::BM) * ((g.N + CFG::BN - 1) / CFG::BN);
    dim3 grid(tiles, 1, g.batch);
    auto kfn = gemm_bf16_kernel<ACT, OUT_F32, CFG, OPK>;
    if (CFG::kLdsBytes > 64 * 1024) {
        static ivlm_dev_mask_t attr_set{0};  // per instantiation, one bit per device
        if (ivlm_dev_pending(attr_set)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      CFG::kLdsBytes);
            ivlm_dev_done(attr_set);
        }
    }
    ivlm_launch(kfn, grid, dim3(CFG::kThreads), CFG::kLdsBytes, st, g);
    return ivlm_launch_status();
}

using Cfg128x64 = TileCfg<2, 2, 4, 2>;  // 128 x 64 block: more tiles for skinny problems (LLM prefill o/down)
// Row-stationary tile for the LLaMA prefill (M = 330 = two row tiles of 176): a wave owns ALL rows of a 32-column strip, so each A
// fragment read feeds 2 MFMAs and each W fragment 11 (0.59 LDS reads per MFMA against 0.75 for the 128 x 64 tile), the tile's
// arithmetic intensity is 74 FLOP per DMA byte against 43, and 2 x 172 tiles of gate|up fit the 512 block slots in ONE round
// (3 x 172 tiles of 128 x 128 need a second round for 4 tiles).  Cold weights (tools/bench_prefill_gemm.py): gate|up 115 -> 86 us,
// q|k|v 73 -> 60 us (with 2 K slices); o / down (N = 4096) do not gain.
using Cfg176x128 = TileCfg<1, 4, 11, 2>;     // 176 x 128, 2 blocks per CU (78 KB of LDS each)
// (352 x 128 - all 330 rows in one block, one wave per SIMD with its accumulators in AGPRs - was built and measured: 134 us for
//  q|k|v and 136 us for gate|up against 60 / 86 us for 176 x 128: a single wave per SIMD cannot hide its own LDS latency.)
using Cfg128x96 = TileCfg<2, 2, 4, 3>;  // 128 x 96 block: M ~ 330 prefill against wide N (3 tile rows: see choose_tile)

// Pick the block tile from the tile counts (measured with tools/bench_gemm.py):
//   * fewer than ~400 tiles of 128^2 (under one resident wave at 2 blocks/CU): halve the tile (128x64) to
//     put more CUs to work (LLM prefill o/down, CLIP: +15-30 %);
//   * 256^2 (+10 % on SAM qkv / mlp1) when it fills whole waves of the 256 CUs (quantisation efficiency >= 0.85);
//   * 128^2 otherwise.
// 256 x 320 tiles (gemm320.hip): N a multiple of 320 (SAM ViT-H: 1280 / 3840 / 5120) and enough row tiles that whole rounds of
// the 256 CUs are filled - proj / mlp2 of the 4-view call are exactly ONE round instead of 1.25 rounds of 256^2 tiles
inline bool tile320_fits(const GemmArgs& g) {
    if (g.fp8 || g.out_fp8 || g.batch != 1 || g.a_kstep || g.w_kstep || g.c_panel || g.rms_w) return false;
    if ((g.act != ACT_NONE && g.act != ACT_GELU) || g.N % 320 || g.M < 2048) return false;
    if (!(g.out_f32 ? gemm_whole_lines_ok<true>(g, g.act) : gemm_whole_lines_ok<false>(g, g.act))) return false;  // (its only epilogue)
    // fp32 outputs only (SAM proj / mlp2: 1.25 rounds of 256^2 tiles -> one round).  The 16-bit-output GEMMs (qkv, mlp1) are also
    // faster on this tile ALONE (qkv 163 -> 155 us, encoder 29.64 -> 29.56 ms) but evaluate() is SLOWER with them on it (101.2 vs
    // 99.8 ms, alternating runs on one box; tile off everywhere: 100.5): a 256 x 320 tile holds its CU 25 % longer than a 256^2
    // one, and the decode / prefill kernels of the other stream - the critical path - wait for CUs at tile granularity.
    // (fp16 "exact q" projections: k|v, N = 2560, 16-bit output, and q on split rows with a split output, N = 1280 - exactly two /
    //  one whole rounds of this tile where 256^2 tiles leave a quarter-filled round or fall back to 128^2)
    const long tm = (g.M + 255) / 256, tiles = tm * (g.N / 320);
    if ((!g.out_f32 || g.out_split) && !(g.f16 && tiles % 256 == 0 && tiles <= 512)) return false;
    const double q = (double)tiles / (double)(((tiles + 255) / 256) * 256);
    return tiles >= 256 && q >= 0.85 && (double)(tm * 256) / (double)g.M < 1.1;
}
static int g_tile320 = 1;  // benchmark hook (ivlm_gemm_tile320): 0 = never pick the 256 x 320 tile automatically

inline int choose_tile(const GemmArgs& g) {
    if (g.tile == 128 || g.tile == 256 || g.tile == 64 || g.tile == 96 || g.tile == 512 || g.tile == 176 || g.tile == 320) return g.tile;
    if (g_tile320 && tile320_fits(g)) return 320;
    if (g.M > 128 && g.M <= 352 && g.N >= 8192 && g.batch <= 2 && (g.act == ACT_NONE || g.act == ACT_SWIGLU)) return 176;  // LLaMA prefill, wide N
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch;
    // (128x96 tiles for the LLaMA prefill's wide projections, M ~ 330 = 3 tile rows: faster in the warm micro-benchmark -
    //  qkv 58.6 -> 54.6 us, gate|up 105.5 -> 93.2 us - but SLOWER in the pipeline, where the 100-180 MB of weights come cold
    //  from HBM and the encoder shares the CUs: 67.5 -> 73.4 us and 114 -> 133 us by rocprofv3; available as tile 96 only)
    if (t128 < 400) return 64;
    const long t256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * g.batch;
    const double q = (double)t256 / (double)(((t256 + 255) / 256) * 256);
    const double edge = (double)(((g.M + 255) / 256) * 256) * (((g.N + 255) / 256) * 256) / ((double)g.M * g.N);
    if (t256 >= 512 && q >= 0.85 && edge < 1.1) return 512;  // 256^2 tile, 8-phase ping-pong pipeline (gemm256.hip)
    const int keff = g.a_split ? 2 * g.K : g.K;  // (a split A operand walks every K tile twice)
    if (t256 >= 256 && keff >= 2048 && edge < 1.1) return 512;     // long K amortises its prologue/epilogue even at 1-2 waves of tiles (SAM mlp2: 911 vs 837 TF)
    return 128;
}

template <int ACT>
int launch(const GemmArgs& g, hipStream_t st) {
    if (g.fp8) {  // fp8 operands: the 8-phase 256^2 kernel, or 128x128 / 128x64 tiles for what it does not fit
        const int t = choose_tile(g);
        if (t == 512 || t == 256) return gemm_bf16_256p(g, st);
        if (t == 64) return g.out_f32 ? launch_cfg<ACT, true, Cfg128x64, 1>(g, st) : launch_cfg<ACT, false, Cfg128x64, 1>(g, st);
        return g.out_f32 ? launch_cfg<ACT, true, Cfg128, 1>(g, st) : launch_cfg<ACT, false, Cfg128, 1>(g, st);
    }
    if (g.f16) {  // fp16 operands: the tilings of the bf16 path (the epilogues the three towers use: keeps the build small)
        if constexpr (ACT == ACT_NONE || ACT == ACT_GELU || ACT == ACT_QUICK_GELU || ACT == ACT_SWIGLU) {
            const int t = choose_tile(g);
            if (t == 320) {
                const int rc = gemm_bf16_320p(g, st);
                if (rc != IVLM_ERR_UNSUPPORTED) return rc;
            }
            if (t == 512 || t == 256) return gemm_bf16_256p(g, st);
            if (t == 176) {
                if constexpr (ACT == ACT_NONE || ACT == ACT_SWIGLU)
                    return g.out_f32 ? launch_cfg<ACT, true, Cfg176x128, 2>(g, st) : launch_cfg<ACT, false, Cfg176x128, 2>(g, st);
            }
            if (t == 64 || t == 176 || t == 96)
                return g.out_f32 ? launch_cfg<ACT, true, Cfg128x64, 2>(g, st) : launch_cfg<ACT, false, Cfg128x64, 2>(g, st);
            return g.out_f32 ? launch_cfg<ACT, true, Cfg128, 2>(g, st) : launch_cfg<ACT, false, Cfg128, 2>(g, st);
        } else {
            return IVLM_ERR_UNSUPPORTED;
        }
    }
    switch (choose_tile(g)) {
        case 320: {  // 256 x 320 (gemm320.hip); a FORCED 320 on a problem outside its epilogue falls through to 128^2
            const int rc = gemm_bf16_320p(g, st);
            if (rc != IVLM_ERR_UNSUPPORTED) return rc;
            return g.out_f32 ? launch_cfg<ACT, true, Cfg128>(g, st) : launch_cfg<ACT, false, Cfg128>(g, st);
        }
        case 512: return gemm_bf16_256p(g, st);  // 256^2, 8-phase ping-pong pipeline (gemm256.hip)
        case 256: return g.out_f32 ? launch_cfg<ACT, true, Cfg256>(g, st) : launch_cfg<ACT, false, Cfg256>(g, st);
        case 64: return g.out_f32 ? launch_cfg<ACT, true, Cfg128x64>(g, st) : launch_cfg<ACT, false, Cfg128x64>(g, st);
        case 96: return g.out_f32 ? launch_cfg<ACT, true, Cfg128x96>(g, st) : launch_cfg<ACT, false, Cfg128x96>(g, st);
        case 176:  // (instantiated for the epilogues the prefill uses only: keeps the build small)
            if constexpr (ACT == ACT_NONE || ACT == ACT_SWIGLU)
                return g.out_f32 ? launch_cfg<ACT, true, Cfg176x128>(g, st) : launch_cfg<ACT, false, Cfg176x128>(g, st);
            else
                return g.out_f32 ? launch_cfg<ACT, true, Cfg128x64>(g, st) : launch_cfg<ACT, false, Cfg128x64>(g, st);
        default: return g.out_f32 ? launch_cfg<ACT, true, Cfg128>(g, st) : launch_cfg<ACT, false, Cfg128>(g, st);
    }
}

}  // namespace

static int g_nsplit = 1;  // benchmark hook: 0 = never split N
void gemm_set_nsplit(int on) { g_nsplit = on; }

// Column split for the 256 x 256 path: a GEMM whose tile count is just over a whole number of rounds on the 256 CUs (SAM
// proj / mlp2: 16384 x 1280 -> 64 x 5 = 320 tiles = one full round + a quarter-filled one, 62 % efficiency) runs as the
// columns that DO fill whole rounds (4 x 256 = 1024 -> 256 tiles, 8-phase kernel) plus the remaining strip on 128 x 64 tiles
// (512 small blocks, 2-3 per CU).  Two launches of the existing kernels; no partial sums, the epilogues are per column.
static int nsplit_cols(const GemmArgs& g) {
    if (!g_nsplit || g.tile != 0 || g.batch != 1 || g.act == ACT_SWIGLU || g.M < 4096 || (g.N & 255)) return 0;
    if (g_tile320 && tile320_fits(g)) return 0;  // (whole rounds of 256 x 320 tiles: nothing to split off)
    const long tm = (g.M + 255) / 256, tn = g.N / 256;
    const long tiles = tm * tn;
    const double q = (double)tiles / (double)(((tiles + 255) / 256) * 256);
    if (q >= 0.8 || tn < 2 || (g.a_split ? 2 * g.K : g.K) < 2048) return 0;  // (K = 1280: the split loses - 76.6 vs 70.4 us on SAM proj)
    for (long c = tn - 1; c >= tn / 2; --c)
        if ((tm * c) % 256 == 0 || (double)(tm * c) / (double)(((tm * c + 255) / 256) * 256) >= 0.97) return (int)(c * 256);
    return 0;
}

int gemm_bf16(const GemmArgs& g, hipStream_t st) {
    if (!g.A || !g.W || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0 || g.batch <= 0) return IVLM_ERR_INVALID_ARG;
    if (const int n1 = nsplit_cols(g)) {
        GemmArgs a = g, b = g;
        a.N = n1;
        a.tile = 512;
        int rc = gemm_bf16(a, st);
        if (rc != IVLM_OK) return rc;
        b.N = g.N - n1;
        b.tile = 64;
        b.W = g.W + (int64_t)n1 * (g.w_kstep ? 64 : g.ldw);
        b.C = g.c_panel ? static_cast<char*>(g.C) + (size_t)(n1 / 64) * g.c_panel * 2  // (n1 % 256 == 0: whole panels)
                        : static_cast<char*>(g.C) + (size_t)n1 * (g.out_fp8 ? 1 : ((g.out_f32 && !g.out_split) ? 4 : 2));
        if (g.bias) b.bias = g.bias + n1;
        if (g.residual) b.residual = g.residual + (g.res_f32 ? 2 * n1 : n1);
        return gemm_bf16(b, st);
    }
    if (g.K % 8 != 0) return IVLM_ERR_UNSUPPORTED;  // 16-byte K granules; a K % 64 tail is zero-filled in LDS
    if ((g.lda & 7) || (g.ldw & 7)) return IVLM_ERR_UNSUPPORTED;  // 16-byte DMA granules
    if (g.act == ACT_SWIGLU && ((g.N & 3) || g.residual)) return IVLM_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) return IVLM_ERR_INVALID_ARG;
    if (g.a_f32) return IVLM_ERR_UNSUPPORTED;
    if (g.f16 || g.out_f16) {  // IEEE-half operands / output: plain tile kernels
        // (out_f16 with out_split: the fp32 value leaves as [hi | lo] IEEE halves; a_split with f16: [hi | lo] fp16 rows)
        if (g.fp8 || g.out_fp8 || g.a_f32 || (g.out_f16 && g.out_f32 && !g.out_split) || (g.a_split && !g.f16)) return IVLM_ERR_UNSUPPORTED;
    }
    if (g.a_split || g.out_split) {  // fp32-activation ("parity") operands / outputs: bf16 tile kernels only
        if (g.fp8 || g.out_fp8 || g.a_kstep || g.w_kstep || g.c_panel) return IVLM_ERR_UNSUPPORTED;
        if (g.a_split && ((g.a_lo & 7) || g.a_lo < g.K)) return IVLM_ERR_INVALID_ARG;
        if (g.out_split && (!g.out_f32 || (g.c_lo & 1) || g.c_lo <= 0)) return IVLM_ERR_INVALID_ARG;
    }
    if (g.a_kstep || g.w_kstep || g.c_panel) {  // K-panel layouts: bf16, whole 64-wide panels, plain epilogues
        // (A / C panels: one problem, plain epilogues; W panels alone also serve the split-K slices and the SwiGLU epilogue)
        if (g.fp8 || g.out_fp8 || (g.K & 63)) return IVLM_ERR_UNSUPPORTED;
        if ((g.a_kstep || g.c_panel) && (g.batch != 1 || g.act == ACT_SWIGLU)) return IVLM_ERR_UNSUPPORTED;
        if ((g.a_kstep && (g.a_kstep < (int64_t)64 * g.M || (g.a_kstep & 7) || g.a_rows)) || (g.w_kstep && (g.w_kstep < (int64_t)64 * g.N || (g.w_kstep & 7))))
            return IVLM_ERR_INVALID_ARG;
        if (g.c_panel && (g.out_f32 || (g.N & 63) || g.c_panel < (int64_t)64 * g.M || g.out_rows)) return IVLM_ERR_UNSUPPORTED;
    }
    if (g.fp8) {
        if (!g.scale_a || !g.scale_w || (g.out_fp8 && (!g.scale_out || g.out_f32 || (g.ldc & 3)))) return IVLM_ERR_INVALID_ARG;
        // (only the epilogues the fp8 paths use are instantiated: SAM encoder, CLIP, LLaMA prefill)
        if (g.act == ACT_NONE) return launch<ACT_NONE>(g, st);
        if (g.act == ACT_GELU) return launch<ACT_GELU>(g, st);
        if (g.act == ACT_QUICK_GELU) return launch<ACT_QUICK_GELU>(g, st);
        if (g.act == ACT_SWIGLU) return (g.N & 3) || g.residual ? IVLM_ERR_UNSUPPORTED : launch<ACT_SWIGLU>(g, st);
        return IVLM_ERR_UNSUPPORTED;
    }
    if (g.out_fp8) return IVLM_ERR_UNSUPPORTED;
    switch (g.act) {
        case ACT_NONE: return launch<ACT_NONE>(g, st);
        case ACT_GELU: return launch<ACT_GELU>(g, st);
        case ACT_QUICK_GELU: return launch<ACT_QUICK_GELU>(g, st);
        case ACT_RELU: return launch<ACT_RELU>(g, st);
        case ACT_SILU: return launch<ACT_SILU>(g, st);
        case ACT_SWIGLU: return launch<ACT_SWIGLU>(g, st);
        case ACT_SIGMOID: return launch<ACT_SIGMOID>(g, st);
        default: return IVLM_ERR_INVALID_ARG;
    }
}

// ---- split-K ---------------------------------------------------------------------------------------------------
// Small-M GEMMs (LLaMA prefill M ~ 330, CLIP M = 257) produce too few output tiles to fill 256 CUs and each tile walks
// the whole K serially.  K is cut into `splits` slices that run as the batch dimension of the same MFMA kernel
// (fp32 partials in a caller-owned workspace); one pass sums the slices in slice order (deterministic) and applies
// bias / activation / residual.
namespace {
template <bool OUT_F32>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, GemmArgs g) {
    const int n4 = g.N >> 2;
    const int64_t total = (int64_t)g.M * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
        splitk_finish4<OUT_F32>(g, part, splits, m, n);
    }
}
}  // namespace

int gemm_bf16_splitk(const GemmArgs& g, int splits, float* workspace, size_t ws_bytes, hipStream_t st, int32_t* counters) {
    if (splits < 2 || g.batch != 1 || g.act == ACT_SWIGLU || g.rms_w || g.out_rows || g.a_rows || !workspace) return IVLM_ERR_INVALID_ARG;
    if (g.K % (splits * 8) != 0 || (g.N & 3) || (g.ldc & 3)) return IVLM_ERR_UNSUPPORTED;
    if (ws_bytes < (size_t)splits * g.M * g.N * sizeof(float)) return IVLM_ERR_WORKSPACE;
    GemmArgs p = g;
    p.K = g.K / splits;
    p.batch = splits;
    p.strideA = p.K;
    p.strideW = g.w_kstep ? (int64_t)(p.K / 64) * g.w_kstep : p.K;  // (K-panel W: slice z starts z * (K / splits / 64) panels in)
    p.strideC = (int64_t)g.M * g.N;
    p.strideR = 0;
    p.C = workspace;
    p.ldc = g.N;
    p.out_f32 = 1;
    p.out_f16 = 0;    // (a 16-bit / split fp16 output is written by the reduce pass)
    p.out_split = 0;  // (the partials are fp32; the reduce pass applies the epilogue, split output included; a_split / a_lo stay)
    p.bias = nullptr;
    p.residual = nullptr;
    p.act = ACT_NONE;
    if (counters && !g.fp8 && (g.N & 3) == 0) {
        // fused reduction: the tile kernels of gemm.hip only (they carry the fixup), and a tile count the counter array covers
        int t = choose_tile(p);
        if (t == 512 || t == 256 || t == 320) t = 128;
        const long tiles = (long)((g.M + 127) / 128) * ((g.N + 63) / 64);  // (the smallest tile any of them uses: an upper bound)
        if (tiles <= kSplitKCounters) {
            p.tile = t;
            p.sk.count = counters;
            p.sk.C = g.C; p.sk.bias = g.bias; p.sk.residual = g.residual;
            p.sk.ldc = g.ldc; p.sk.ldr = g.ldr; p.sk.c_lo = g.c_lo;
            p.sk.res_mod = g.res_mod; p.sk.act = g.act; p.sk.out_f32 = g.out_f32; p.sk.out_f16 = g.out_f16; p.sk.out_split = g.out_split;
            p.sk.res_f32 = g.res_f32;
            return gemm_bf16(p, st);
        }
    }
    const int rc = gemm_bf16(p, st);
    if (rc != IVLM_OK) return rc;
    const int64_t total = (int64_t)g.M * (g.N >> 2);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
    if (g.out_f32) ivlm_launch(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, splits, g);
    else ivlm_launch(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, splits, g);
    return ivlm_launch_status();
}

// dispatch: weight-streaming GEMV for M <= 8 (wave per row while the activation rows fit LDS, the split-K MFMA skinny
// kernel of gemv_mfma.hip for the batched decode step: M up to 16 rows against big matrices), MFMA tile kernel otherwise
static int g_skinny_min_m = 0;  // 0: automatic; > 0: every M in [min_m, 16] with K, N >= 1024 (benchmark hook)
void gemv_mfma_set_min_m(int m) { g_skinny_min_m = m; }

int linear_bf16(const GemmArgs& g, hipStream_t st) {
    if (g.batch == 1 && g.M <= 16 && g.K >= 1024 && g.N >= 1024 && !(g.K & 7) && !(g.lda & 7) && !(g.ldw & 7) &&
        !(g.act == ACT_SWIGLU && ((g.N & 1) || g.residual))) {
        const bool skinny = g_skinny_min_m > 0 ? g.M >= g_skinny_min_m
                                               : (g.M >= 5 || (g.M >= 3 && (size_t)g.M * g.K * 2 > 48 * 1024));
        // (measured on the LLaMA-7B shapes, tools/bench_skinny.py: the wave-per-row GEMV wins for M <= 4 while its
        //  activation rows fit LDS - 5.0-5.7 TB/s at M = 1 - and falls to 0.9-2.5 TB/s at M = 8; this kernel holds
        //  3.4-4.4 TB/s for every M <= 8 and 2.9-3.4 TB/s at M = 16, where the 128x64 tile GEMM reaches 0.7-1.8 TB/s)
        // fp32 activations: exact in the GEMV (fp32 x in LDS) for one row, hi + lo bf16 operand split on the MFMA for more
        if (!g.out_rows && !g.a_rows && (skinny || (g.a_f32 && g.M >= 2))) return gemv_mfma_bf16(g, st);
    }
    if (g.M <= 8 && g.batch == 1 && !g.out_rows && !g.a_rows) return gemv_bf16(g, st);
    if (g.a_f32 && g.M <= 16 && g.batch == 1 && !g.out_rows && !g.a_rows) {
        // 9..16 fp32 rows against a matrix the skinny MFMA kernel does not take (K or N < 1024: cam-pose encoders, AttentionSplitter,
        // text_hidden_fcs[1], small-width decode batches): two passes of the weight-streaming GEMV over row chunks of <= 8
        for (int m0 = 0; m0 < g.M; m0 += 8) {
            GemmArgs c = g;
            c.M = g.M - m0 < 8 ? g.M - m0 : 8;
            c.A = reinterpret_cast<const bf16_t*>(reinterpret_cast<const float*>(g.A) + (int64_t)m0 * g.lda);
            c.C = static_cast<char*>(g.C) + (size_t)m0 * g.ldc * (g.out_f32 ? 4 : 2);
            if (g.residual && g.res_mod <= 0) c.residual = g.residual + (int64_t)m0 * g.ldr * (g.res_f32 ? 2 : 1);
            if (g.residual && g.res_mod > 0) return IVLM_ERR_UNSUPPORTED;
            const int rc = gemv_bf16(c, st);
            if (rc != IVLM_OK) return rc;
        }
        return IVLM_OK;
    }
    if (g.a_f32) return IVLM_ERR_UNSUPPORTED;  // the tile kernels DMA bf16 operands (use ivlm_gather_rows split + IVLM_GEMM_A_SPLIT)
    if (g.rms_w) return IVLM_ERR_UNSUPPORTED;  // the RMSNorm fusion exists on the decode (GEMV) path only
    return gemm_bf16(g, st);
}

}  // namespace ivlm

static int g_tile_override = 0;

extern "C" int ivlm_gemv_mfma_min_m(int min_m) {  // benchmark/test hook: 0 = automatic choice
    ivlm::gemv_mfma_set_min_m(min_m);
    return 0;
}

extern "C" int ivlm_gemm_nsplit(int on) {  // benchmark hook: column split of under-filled 256 x 256 rounds (default on)
    ivlm::gemm_set_nsplit(on);
    return 0;
}

extern "C" int ivlm_gemm_tile320(int on) {  // benchmark hook: automatic choice of the 256 x 320 tile (default on); returns the previous value
    const int prev = ivlm::g_tile320;
    ivlm::g_tile320 = on ? 1 : 0;
    return prev;
}

extern "C" int ivlm_gemm_tile_override(int tile) {
    const int prev = g_tile_override;
    if (tile == 0 || tile == 64 || tile == 96 || tile == 128 || tile == 256 || tile == 512 || tile == 176 || tile == 320) g_tile_override = tile;
    return prev;
}

extern "C" int ivlm_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                              const void* bias, const void* residual, int64_t ldr, int res_mod, int M, int N, int K,
                              int act, int out_f32, int batch, int64_t strideA, int64_t strideW, int64_t strideC,
                              int64_t strideR, const void* rms_w, float rms_eps, int flags, const int32_t* out_rows,
                              const int32_t* a_rows, ivlm_stream_t stream) {
    ivlm_enter();
    ivlm::GemmArgs g;
    g.out_rows = out_rows;
    g.a_rows = a_rows;
    g.a_f32 = (flags & IVLM_GEMM_A_F32) ? 1 : 0;
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    if (flags & IVLM_GEMM_A_SPLIT) { g.a_split = 1; g.a_lo = K; }
    if (flags & IVLM_GEMM_OUT_SPLIT) { g.out_split = 1; g.c_lo = act == ivlm::ACT_SWIGLU ? N / 2 : N; out_f32 = 1; }
    if (flags & IVLM_GEMM_F16) g.f16 = 1;
    if (flags & IVLM_GEMM_OUT_F16) g.out_f16 = 1;
    if (flags & IVLM_GEMM_W_PANEL) {
        if (M <= 16 || (K & 63) || batch > 1) return IVLM_ERR_UNSUPPORTED;
        g.w_kstep = (int64_t)64 * N;
        ldw = 64;
    }
    if ((g.a_split || g.out_split || g.f16 || g.out_f16) && M <= 16) return IVLM_ERR_UNSUPPORTED;  // tile GEMM path only
    g.rms_w = static_cast<const bf16_t*>(rms_w);
    g.rms_eps = rms_eps;
    g.tile = g_tile_override;
    g.A = static_cast<const bf16_t*>(A);
    g.W = static_cast<const bf16_t*>(W);
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.res_mod = res_mod;
    g.M = M; g.N = N; g.K = K;
    g.act = act;
    g.out_f32 = out_f32;
    g.batch = batch < 1 ? 1 : batch;
    g.strideA = strideA; g.strideW = strideW; g.strideC = strideC; g.strideR = strideR;
    return ivlm::linear_bf16(g, ivlm_stream(stream));
}

extern "C" int ivlm_gemm_bf16_panel(const void* A, int64_t lda, int64_t a_kstep, const void* W, int64_t ldw, int64_t w_kstep, void* C,
                                    int64_t ldc, int64_t c_panel, const void* bias, const void* residual, int64_t ldr, int M, int N,
                                    int K, int act, int out_f32, int flags, const int32_t* out_rows, const int32_t* a_rows,
                                    ivlm_stream_t stream) {
    ivlm_enter();
    if (M <= 16) return IVLM_ERR_UNSUPPORTED;  // tile GEMM only
    ivlm::GemmArgs g;
    g.out_rows = out_rows;
    g.a_rows = a_rows;
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    if (flags & IVLM_GEMM_A_F32) return IVLM_ERR_UNSUPPORTED;
    g.tile = g_tile_override;
    g.A = static_cast<const bf16_t*>(A);
    g.W = static_cast<const bf16_t*>(W);
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.lda = a_kstep ? 64 : lda; g.ldw = w_kstep ? 64 : ldw; g.ldc = ldc; g.ldr = ldr;
    g.a_kstep = a_kstep; g.w_kstep = w_kstep; g.c_panel = c_panel;
    g.M = M; g.N = N; g.K = K;
    g.act = act;
    g.out_f32 = out_f32;
    return ivlm::gemm_bf16(g, ivlm_stream(stream));
}

extern "C" int ivlm_gemm_fp8(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias,
                             const void* residual, int64_t ldr, int M, int N, int K, int act, int out_kind, const float* scale_a,
                             const float* scale_w, const float* scale_out, int flags, const int32_t* out_rows,
                             const int32_t* a_rows, ivlm_stream_t stream) {
    ivlm_enter();
    if ((K & 15) || (lda & 15) || (ldw & 15) || M <= 16) return IVLM_ERR_UNSUPPORTED;  // 16-byte granules; tile GEMM only
    ivlm::GemmArgs g;
    g.fp8 = 1;
    g.out_fp8 = out_kind == 2 ? 1 : 0;
    g.out_f32 = out_kind == IVLM_F32 ? 1 : 0;
    g.scale_a = scale_a; g.scale_w = scale_w; g.scale_out = scale_out;
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    g.out_rows = out_rows;
    g.a_rows = a_rows;
    g.A = static_cast<const bf16_t*>(A);
    g.W = static_cast<const bf16_t*>(W);
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.lda = lda / 2; g.ldw = ldw / 2; g.ldc = ldc; g.ldr = ldr;  // byte matrices seen as 2-byte matrices of half the width
    g.M = M; g.N = N; g.K = K / 2;
    g.act = act;
    g.tile = g_tile_override;
    return ivlm::gemm_bf16(g, ivlm_stream(stream));
}

extern "C" size_t ivlm_gemm_splitk_workspace_bytes(int M, int N, int splits) {
    return (size_t)(splits < 1 ? 1 : splits) * (size_t)M * (size_t)N * sizeof(float);
}

static int splitk_entry(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                     const void* bias, const void* residual, int64_t ldr, int res_mod, int M, int N,
                                     int K, int act, int out_f32, int splits, void* workspace, size_t workspace_bytes,
                                     int flags, ivlm_stream_t stream, int32_t* counters) {
    ivlm_enter();
    if (flags & IVLM_GEMM_A_F32) return IVLM_ERR_UNSUPPORTED;
    ivlm::GemmArgs g;
    g.res_f32 = (flags & IVLM_GEMM_RES_F32) ? 1 : 0;
    if (flags & IVLM_GEMM_A_SPLIT) { g.a_split = 1; g.a_lo = K; }
    if (flags & IVLM_GEMM_OUT_SPLIT) { g.out_split = 1; g.c_lo = N; out_f32 = 1; }
    if (flags & IVLM_GEMM_F16) g.f16 = 1;
    if (flags & IVLM_GEMM_OUT_F16) {
        if (out_f32 && !g.out_split) return IVLM_ERR_INVALID_ARG;
        g.out_f16 = 1;
    }
    if (flags & IVLM_GEMM_W_PANEL) {
        if (K % (splits * 64) != 0) return IVLM_ERR_UNSUPPORTED;  // (whole panels per K slice)
        g.w_kstep = (int64_t)64 * N;
        ldw = 64;
    }
    g.tile = g_tile_override;
    g.A = static_cast<const bf16_t*>(A);
    g.W = static_cast<const bf16_t*>(W);
    g.C = C;
    g.bias = static_cast<const bf16_t*>(bias);
    g.residual = static_cast<const bf16_t*>(residual);
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.res_mod = res_mod;
    g.M = M; g.N = N; g.K = K;
    g.act = act;
    g.out_f32 = out_f32;
    return ivlm::gemm_bf16_splitk(g, splits, static_cast<float*>(workspace), workspace_bytes, ivlm_stream(stream), counters);
}

extern "C" int ivlm_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                     const void* bias, const void* residual, int64_t ldr, int res_mod, int M, int N,
                                     int K, int act, int out_f32, int splits, void* workspace, size_t workspace_bytes,
                                     int flags, ivlm_stream_t stream) {
    return splitk_entry(A, lda, W, ldw, C, ldc, bias, residual, ldr, res_mod, M, N, K, act, out_f32, splits, workspace, workspace_bytes,
                        flags, stream, nullptr);
}

// ... with the reduction fused into the GEMM launch: `counters` = IVLM_SPLITK_COUNTERS int32 words, zeroed ONCE by the caller (every
// call leaves them at zero), not shared by launches that may run concurrently (one array per stream).  Same results, bit for bit.
extern "C" int ivlm_gemm_bf16_splitk_fused(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                           const void* bias, const void* residual, int64_t ldr, int res_mod, int M, int N,
                                           int K, int act, int out_f32, int splits, void* workspace, size_t workspace_bytes,
                                           int32_t* counters, int flags, ivlm_stream_t stream) {
    if (!counters) return IVLM_ERR_INVALID_ARG;
    return splitk_entry(A, lda, W, ldw, C, ldc, bias, residual, ldr, res_mod, M, N, K, act, out_f32, splits, workspace, workspace_bytes,
                        flags, stream, counters);
}
