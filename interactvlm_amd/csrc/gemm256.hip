// 256x256x64 bf16 MFMA GEMM for gfx950 with an 8-phase, ping-pong K loop (the large-M GEMMs of the SAM ViT-H encoder:
// qkv / proj / mlp of model/segment_anything/modeling/image_encoder.py:222-260, common.py:13-27).
//
// Same contract and epilogues as gemm_bf16_kernel (gemm.hip); different pipeline:
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the output = 8 x 4 MFMA fragments (128 acc regs).
//   * A K tile (BK = 64) is held as four 16-KiB HALF-TILES, cut to match how the waves consume it:
//         A0 / A1 = the 64 rows that output quadrant mq = 0 / 1 of BOTH wave rows reads,
//         B0 / B1 = the 32 columns that quadrant nq = 0 / 1 of ALL FOUR wave columns reads;
//     two K tiles of buffering = 128 KiB of LDS, one block per CU, 2 waves per SIMD.
//   * The K tile is computed in four phases, one output quadrant (16 MFMAs) each:
//         phase 1: read A0 + B0 -> (mq0,nq0)   phase 2: read B1 -> (mq0,nq1)
//         phase 3: read A1      -> (mq1,nq1)   phase 4: (no reads) -> (mq1,nq0)
//     and every phase issues the direct-to-LDS DMA of ONE half-tile, five half-tiles ahead of its first use:
//     half-tile s (issue order A0,B0,B1,A1 of tile 0, then tile 1, ...) is issued in global phase s-5 and first read in
//     phase >= s.  `s_waitcnt vmcnt(8)` once per phase (never 0) therefore leaves four half-tiles in flight across the
//     barriers, and a slot is only restaged >= 2 phases after its last read (reads retire right after the phase's barrier).
//   * ONE barrier per phase.  It publishes the half-tile whose DMA every wave has just waited for (vmcnt(8): half-tiles up to
//     s+1 have landed in phase s) and, because a slot is restaged at least two phases after its last read, also orders the restaging
//     DMA behind everybody's reads.  (Rounds 1-2 ran two barriers per phase with the two wave rows staggered by half a phase,
//     one row computing while the other loads: same speed to the noise - 102.2 vs 102.4 ms end to end, 20.48 vs 20.52 img/s on
//     the batched job - so the simpler schedule stays.  Ablation of this loop at M = 65536, SAM qkv: MFMAs alone 380 us (1.7
//     PFLOP/s: the practical matrix ceiling at the sustained clock), DMA alone 233 us (82 GB/s per CU), barriers alone 25 us,
//     everything 563 us + 31 us of epilogue: what is lost is overlap between a wave's LDS reads and its own MFMAs, with only
//     two waves per SIMD to cover for each other.)
//   * LDS image and swizzle as in gemm.hip: DMA destination is lane-linear, so 16-byte chunk c of local row r is fetched
//     from source chunk c ^ ((r >> 1) & 7) and read back with the same XOR.
#include "gemm_common.h"

namespace ivlm {
namespace {

constexpr int kBK = 64;
constexpr int kHalfBytes = 128 * kBK * 2;        // 16 KiB
constexpr int kTileLds = 4 * kHalfBytes;         // 64 KiB per K tile
constexpr int kLds256 = 2 * kTileLds;            // 128 KiB
// slot order inside a tile buffer == kind index: 0 A0, 1 B0, 2 B1, 3 A1 (also the issue order)

template <int ACT, bool OUT_F32, int OPK>  // OPK: 0 bf16, 1 e4m3, 2 IEEE fp16 operands (compile time: see gemm.hip)
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs g) {
    constexpr bool FP8 = OPK == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int m0, n0;
    gemm_tile_origin(g, 256, 256, m0, n0);
    const int bz = blockIdx.z;
    const bf16_t* __restrict__ A = g.A + (int64_t)bz * g.strideA;
    const bf16_t* __restrict__ W = g.W + (int64_t)bz * g.strideW;

    // ---- DMA sources: this lane copies local rows lr0 and lr0 + 64 of every half-tile ----------------------------------
    const int lr0 = wave * 8 + (lane >> 3);                 // 0..63
    const int chunk = (lane & 7) ^ ((lr0 >> 1) & 7);        // source chunk landing in LDS chunk lane & 7 (same for lr0+64)
    const int kcol = chunk * 8;
    const bf16_t* src[4][2];  // [kind][piece]
    // row stride / K-tile step of each operand: (lda, 64) row-major, (64, kstep) in the K-panel layout (kernels.h)
    const int64_t a_rs = g.a_kstep ? 64 : g.lda, a_ks = g.a_kstep ? g.a_kstep : kBK;
    const int64_t w_rs = g.w_kstep ? 64 : g.ldw, w_ks = g.w_kstep ? g.w_kstep : kBK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int ra = m0 + i * 128 + q * 64 + lr0;            // A half q: local row i*64 + lr0 = wave row i, quadrant row lr0
            ra = ra < g.M ? ra : g.M - 1;
            if (g.a_rows) ra = g.a_rows[ra];
            src[q == 0 ? 0 : 3][i] = A + (int64_t)ra * a_rs + kcol;
            int rn = n0 + (2 * i + (lr0 >> 5)) * 64 + q * 32 + (lr0 & 31);  // B half q: local row i*64+lr0 = wave col, col
            rn = rn < g.N ? rn : g.N - 1;
            src[q == 0 ? 1 : 2][i] = W + (int64_t)rn * w_rs + kcol;
        }
    }
    const int nt = ((g.K + kBK - 1) / kBK) << (g.a_split ? 1 : 0);  // split A: every W tile twice (hi then lo tile of A)
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(kGemmZeroChunk);
    unsigned char* dma_base = smem + wave * 1024;  // piece i of a slot: + i * 8192

    // half-tile `kind` of K tile `tile` -> its slot (tiles past the end stream the zero chunk: keeps vmcnt uniform)
    auto stage = [&](int kind, int tile) {
        unsigned char* dst = dma_base + (tile & 1) * kTileLds + kind * kHalfBytes;
        const int kw = g.a_split ? tile >> 1 : tile;
        const bool ok = tile < nt && kcol + kw * kBK < g.K;
        const bool isa = kind == 0 || kind == 3;
        const int64_t koff = kw * (isa ? a_ks : w_ks) + ((isa && g.a_split && (tile & 1)) ? g.a_lo : 0);
        glds16(ok ? src[kind][0] + koff : zero, dst);
        glds16(ok ? src[kind][1] + koff : zero, dst + 8192);
    };

    // ---- fragment read offsets (bytes inside a half-tile) --------------------------------------------------------------
    const int sw = (((lane & 15) >> 1) & 7);
    const int offA = (wr * 64 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);  // + i * 2048, ^ (kk << 6)
    const int offB = (wc * 32 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);  // + j * 2048, ^ (kk << 6)

    f32x4_t acc[4][8];  // [n fragment][m fragment]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[4][2];      // A quadrant in use: [m frag][k step]
    bf16x8_t fb[2][2][2];   // both B quadrants:  [nq][n frag][k step]

    auto read_a = [&](const unsigned char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fa[i][kk] = gemm_frag_read(slot + ((offA + i * 2048) ^ (kk << 6)));
    };
    auto read_b = [&](int nq, const unsigned char* slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fb[nq][j][kk] = gemm_frag_read(slot + ((offB + j * 2048) ^ (kk << 6)));
    };
    auto mfma_quadrant = [&](int mq, int nq) {
#ifdef IVLM_ABL_NOMFMA
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) IVLM_ABL_MFMA_USE(fb[nq][j][kk], fb[nq][j][kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) IVLM_ABL_MFMA_USE(fa[i][kk], fa[i][kk]);
        }
        return;
#endif
        if (FP8) {  // the same fragments as bytes: one 16x16x128 e4m3 step instead of two 16x16x32 bf16 steps
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[nq * 2 + j][mq * 4 + i] =
                        mfma_fp8_128(fb[nq][j][0], fb[nq][j][1], fa[i][0], fa[i][1], acc[nq * 2 + j][mq * 4 + i]);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (OPK == 2)  // IEEE-half operands: same fragments, the f16 instruction
                        acc[nq * 2 + j][mq * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            __builtin_bit_cast(f16x8_t, fb[nq][j][kk]), __builtin_bit_cast(f16x8_t, fa[i][kk]), acc[nq * 2 + j][mq * 4 + i], 0, 0, 0);
                    else
                        acc[nq * 2 + j][mq * 4 + i] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nq][j][kk], fa[i][kk], acc[nq * 2 + j][mq * 4 + i], 0, 0, 0);
                }
    };
    // one phase: [LDS reads][DMA of one half-tile][counted wait] | barrier | retire reads, 16 MFMAs
#define IVLM_PHASE(READS, KIND, TILE, MQ, NQ)                    \
    do {                                                         \
        READS;                                                   \
        stage(KIND, TILE);                                       \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_setprio(1);                           \
        mfma_quadrant(MQ, NQ);                                   \
        __builtin_amdgcn_s_setprio(0);                           \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define IVLM_KTILE(T, BUF)                                                                                        \
    do {                                                                                                          \
        const unsigned char* tb = smem + (BUF) * kTileLds;                                                        \
        IVLM_PHASE({ read_b(0, tb + 1 * kHalfBytes); read_a(tb + 0 * kHalfBytes); }, 2, (T) + 1, 0, 0);           \
        IVLM_PHASE({ read_b(1, tb + 2 * kHalfBytes); }, 3, (T) + 1, 0, 1);                                         \
        IVLM_PHASE({ read_a(tb + 3 * kHalfBytes); }, 0, (T) + 2, 1, 1);                                            \
        IVLM_PHASE({}, 1, (T) + 2, 1, 0);                                                                          \
    } while (0)

    // ---- prologue: half-tiles 0..5 (tile 0 complete, A0/B0 of tile 1) --------------------------------------------------
    stage(0, 0);
    stage(1, 0);
    stage(2, 0);
    stage(3, 0);
    stage(0, 1);
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A0, B0 of tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();

    int t = 0;
    for (; t + 1 < nt; t += 2) {
        IVLM_KTILE(t, 0);
        IVLM_KTILE(t + 1, 1);
    }
    if (t < nt) IVLM_KTILE(t, 0);
#undef IVLM_KTILE
#undef IVLM_PHASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-chunk DMAs must not outlive the block's LDS

#ifdef IVLM_ABL_NOEPI
    if (g.M > 0) {  // (one dummy store per lane keeps the accumulators live)
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sacc == 123.456f) static_cast<float*>(g.C)[tid] = sacc;
        return;
    }
#endif
    // ---- epilogue: whole lines through LDS (gemm_common.h), direct fragment stores for what that does not cover --------------
    if (gemm_whole_lines_ok<OUT_F32>(g, ACT)) {
        __syncthreads();  // every wave is done with the K tiles
        // acc is [n fragment][m fragment]; 16 KB of LDS per wave: the bf16 sub-tile in one pass, the fp32 one in two halves
        gemm_store_lines<ACT, OUT_F32, 8, 4, OUT_F32 ? 4 : 8>(g, smem + wave * 16384, m0 + wr * 128, n0 + wc * 64, lane, acc);
        return;
    }
    // direct stores (SwiGLU, fp8 output, ragged N, bf16 residual with bf16 output)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + wr * 128 + mi * 16 + (lane & 15);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + (lane >> 4) * 4;
            gemm_epilogue4<ACT, OUT_F32>(g, bz, m, n, acc[ni][mi]);
        }
    }
}

template <int ACT>
int launch256(const GemmArgs& g, hipStream_t st) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    dim3 grid(tiles, 1, g.batch);
#define IVLM_GO(F32, F8)                                                                                         \
    do {                                                                                                         \
        auto kfn = gemm256_kernel<ACT, F32, F8>;                                                                 \
        static ivlm_dev_mask_t attr_set{0};                                                                      \
        if (ivlm_dev_pending(attr_set)) {                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      kLds256);                                                                  \
            ivlm_dev_done(attr_set);                                                                             \
        }                                                                                                        \
        ivlm_launch(kfn, grid, dim3(512), kLds256, st, g);                                                       \
    } while (0)
    if (g.fp8) {
        if (g.out_f32) IVLM_GO(true, 1); else IVLM_GO(false, 1);
    } else if (g.f16) {  // (instantiated for the epilogues of the three towers only)
        if constexpr (ACT == ACT_NONE || ACT == ACT_GELU || ACT == ACT_QUICK_GELU || ACT == ACT_SWIGLU) {
            if (g.out_f32) IVLM_GO(true, 2); else IVLM_GO(false, 2);
        } else {
            return IVLM_ERR_UNSUPPORTED;
        }
    } else {
        if (g.out_f32) IVLM_GO(true, 0); else IVLM_GO(false, 0);
    }
#undef IVLM_GO
    return ivlm_launch_status();
}

}  // namespace

// tile-256 8-phase variant (arguments already validated by gemm_bf16)
int gemm_bf16_256p(const GemmArgs& g, hipStream_t st) {
    switch (g.act) {
        case ACT_NONE: return launch256<ACT_NONE>(g, st);
        case ACT_GELU: return launch256<ACT_GELU>(g, st);
        case ACT_QUICK_GELU: return launch256<ACT_QUICK_GELU>(g, st);
        case ACT_RELU: return launch256<ACT_RELU>(g, st);
        case ACT_SILU: return launch256<ACT_SILU>(g, st);
        case ACT_SWIGLU: return launch256<ACT_SWIGLU>(g, st);
        case ACT_SIGMOID: return launch256<ACT_SIGMOID>(g, st);
        default: return IVLM_ERR_INVALID_ARG;
    }
}

}  // namespace ivlm
