// 256 x 320 x 64 MFMA GEMM for gfx950: the block tile that fits SAM ViT-H's widths (1280 = 4 x 320, 3840 = 12 x 320, 5120 = 16 x 320;
// image_encoder.py:222-260, common.py:13-27) so that 64 row tiles of the 4-view call fill WHOLE rounds of the 256 CUs:
//     proj / mlp2 (N = 1280): 64 x 4 = 256 tiles = exactly one round        (256^2 tiles: 320 = 1.25 rounds, run as two launches)
//     qkv (N = 3840):         64 x 12 = 768 = 3 rounds of 1.25 tile units   (256^2: 960 tiles = 3.75 -> 4 rounds)
//     mlp1 (N = 5120):        64 x 16 = 1024 = 4 rounds                     (256^2: 1280 = 5 rounds of 1.0: the same)
// Same contract and epilogues as gemm256_kernel; the differences:
//   * 8 waves as 2 (M) x 4 (N), a wave owns 128 x 80 of the output = 8 x 5 fragments (160 accumulator registers): columns
//     wc*64 .. +63 of the tile's first 256 and columns 256 + wc*16 .. +15 of its last 64 (so that the epilogue writes whole,
//     line-aligned rows: the four fifth fragment columns of a wave row are stored together).  Fragments are read ONE k-step (32)
//     at a time: 4 + 5 fragment register sets live instead of both k-steps of a quadrant.  13 LDS fragment reads per 40 MFMAs
//     (256^2: 12 per 32).
//   * a K tile (BK = 64) is a 32-KiB A image + a 40-KiB W image; two K tiles = 144 KiB of LDS (of 160), one block per CU.
//   * four phases per K tile, 20 MFMAs each: (k-step 0, rows 0-63) | (k-step 0, rows 64-127) | (k-step 1, ...) | (k-step 1, ...);
//     W fragments of a k-step are read in its first phase and reused by the second.
//   * DMA (global_load ... lds, 16 B per lane, 9 instructions per thread per K tile): tile T+1's W image and the A rows of the
//     first row-phase ("part 0", 7 instructions) are issued in phase 1 of tile T (their slots were last read in phase 3 of tile
//     T-1, behind phase 4's barrier), the A rows of the second row-phase ("part 1") in phase 2.  Counted waits, never 0 inside the
//     loop: vmcnt(7) in phase 1 (part 1 of T has landed: phase 2 reads it), vmcnt(2) in phase 4 (part 0 of T+1 has landed) -
//     every piece is issued three phases (60 MFMAs per wave, two waves per SIMD: ~1900 cycles) before the wait that needs it.
//   * TWO barriers per K tile (phases 1 and 4: they publish what the waits covered and order the restaging behind everyone's
//     reads) against four in gemm256_kernel.
#include "gemm_common.h"

namespace ivlm {
namespace {

constexpr int kBK = 64;
constexpr int kABytes = 256 * kBK * 2;        // 32 KiB
constexpr int kWBytes = 320 * kBK * 2;        // 40 KiB
constexpr int kTileLds = kABytes + kWBytes;   // 72 KiB per K tile
constexpr int kLds320 = 2 * kTileLds;         // 144 KiB

template <int ACT, bool OUT_F32, int OPK>  // OPK: 0 bf16, 2 IEEE fp16 operands
__global__ __launch_bounds__(512, 2) void gemm320_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int m0, n0;
    gemm_tile_origin(g, 256, 320, m0, n0);
    const int bz = blockIdx.z;
    const bf16_t* __restrict__ A = g.A + (int64_t)bz * g.strideA;
    const bf16_t* __restrict__ W = g.W + (int64_t)bz * g.strideW;

    // ---- DMA sources: piece p of an image = its rows p*64 .. p*64+63; this lane copies row p*64 + lr0 of every piece -------------
    const int lr0 = wave * 8 + (lane >> 3);                 // 0..63
    const int chunk = (lane & 7) ^ ((lr0 >> 1) & 7);        // source chunk landing in LDS chunk lane & 7 (p*64 does not change it)
    const int kcol = chunk * 8;
    const bf16_t* srcA[4];
    const bf16_t* srcW[5];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int ra = m0 + p * 64 + lr0;
        ra = ra < g.M ? ra : g.M - 1;
        if (g.a_rows) ra = g.a_rows[ra];
        srcA[p] = A + (int64_t)ra * g.lda + kcol;
    }
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        int rn = n0 + p * 64 + lr0;
        rn = rn < g.N ? rn : g.N - 1;
        srcW[p] = W + (int64_t)rn * g.ldw + kcol;
    }
    const int nt = ((g.K + kBK - 1) / kBK) << (g.a_split ? 1 : 0);  // split A: every W tile twice (hi then lo tile of A)
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(kGemmZeroChunk);
    unsigned char* dma_base = smem + wave * 1024;  // piece p of an image: + p * 8192

    // DMA of K tile `tile`: part 0 = the W image + A pieces 0 and 2 (rows 0-63 of both wave rows), part 1 = A pieces 1 and 3.
    // Tiles past the end stream the zero chunk (keeps vmcnt uniform).
    auto stage = [&](int part, int tile) {
        unsigned char* dst = dma_base + (tile & 1) * kTileLds;
        const int kw = g.a_split ? tile >> 1 : tile;
        const bool ok = tile < nt && kcol + kw * kBK < g.K;
        const int64_t koffw = (int64_t)kw * kBK;
        const int64_t koffa = koffw + ((g.a_split && (tile & 1)) ? g.a_lo : 0);
        if (part == 0) {
            glds16(ok ? srcA[0] + koffa : zero, dst);
            glds16(ok ? srcA[2] + koffa : zero, dst + 2 * 8192);
#pragma unroll
            for (int p = 0; p < 5; ++p) glds16(ok ? srcW[p] + koffw : zero, dst + kABytes + p * 8192);
        } else {
            glds16(ok ? srcA[1] + koffa : zero, dst + 1 * 8192);
            glds16(ok ? srcA[3] + koffa : zero, dst + 3 * 8192);
        }
    };

    // ---- fragment read offsets (bytes inside an image) ------------------------------------------------------------------------
    const int sw = (((lane & 15) >> 1) & 7);
    const int offA = (wr * 128 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);  // + (mq * 4 + i) * 2048, ^ (kk << 6)
    // a wave's columns: fragments 0-3 = columns wc*64 .. wc*64+63 of the first 256 (whole, line-aligned rows in the epilogue),
    // fragment 4 = columns 256 + wc*16 .. +15: the fifth fragments of the four waves of a wave row form one 64-column strip
    const int offW = (wc * 64 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);   // + j * 2048 (j < 4),  ^ (kk << 6)
    const int offW4 = (256 + wc * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);

    f32x4_t acc[5][8];  // [n fragment][m fragment]
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[4];  // A fragments of the row-phase in use (one k-step)
    bf16x8_t fw[5];  // W fragments of the k-step in use

    auto read_a = [&](const unsigned char* tb, int mq, int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = gemm_frag_read(tb + ((offA + (mq * 4 + i) * 2048) ^ (kk << 6)));
    };
    auto read_w = [&](const unsigned char* tb, int kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = gemm_frag_read(tb + kABytes + ((offW + j * 2048) ^ (kk << 6)));
        fw[4] = gemm_frag_read(tb + kABytes + (offW4 ^ (kk << 6)));
    };
    auto mfma_rows = [&](int mq) {
#ifdef IVLM_ABL_NOMFMA
#pragma unroll
        for (int j = 0; j < 5; ++j) IVLM_ABL_MFMA_USE(fw[j], fw[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) IVLM_ABL_MFMA_USE(fa[i], fa[i]);
        return;
#endif
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (OPK == 2)
                    acc[j][mq * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fw[j]), __builtin_bit_cast(f16x8_t, fa[i]),
                                                                                acc[j][mq * 4 + i], 0, 0, 0);
                else
                    acc[j][mq * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[j][mq * 4 + i], 0, 0, 0);
            }
    };
    // one phase: [LDS reads of data published by an EARLIER phase's wait + barrier][DMA][counted wait + barrier for a later phase] |
    // retire reads, 20 MFMAs
#define IVLM_PHASE(READS, DMA, SYNC, MQ)                         \
    do {                                                         \
        READS;                                                   \
        DMA;                                                     \
        SYNC;                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_setprio(1);                           \
        mfma_rows(MQ);                                           \
        __builtin_amdgcn_s_setprio(0);                           \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define IVLM_WAIT_BAR(N)                                         \
    do {                                                         \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");    \
        __builtin_amdgcn_s_barrier();                            \
    } while (0)
    // phase 1: part 0 of T+1 issued (7); what must have landed for phase 2 is part 1 of T (2 instructions, issued before them)
    // phase 4: outstanding are part 0 (7) and part 1 (2) of T+1: part 0 must have landed for phase 1 of T+1.  The two barriers
    // also order the restaging: part 0 of T+1 overwrites what phase 3 of T-1 read last (behind phase 4's barrier of T-1), part 1
    // what phase 4 of T-1 read last (behind phase 1's barrier of T).
#define IVLM_KTILE(T)                                                                                                  \
    do {                                                                                                               \
        const unsigned char* tb = smem + ((T) & 1) * kTileLds;                                                         \
        IVLM_PHASE({ read_w(tb, 0); read_a(tb, 0, 0); }, stage(0, (T) + 1), IVLM_WAIT_BAR(7), 0);                      \
        IVLM_PHASE({ read_a(tb, 1, 0); }, stage(1, (T) + 1), {}, 1);                                                   \
        IVLM_PHASE({ read_w(tb, 1); read_a(tb, 0, 1); }, {}, {}, 0);                                                   \
        IVLM_PHASE({ read_a(tb, 1, 1); }, {}, IVLM_WAIT_BAR(2), 1);                                                    \
    } while (0)

    // ---- prologue: all of tile 0; part 0 (issued first) must have landed for phase 1 -------------------------------------------
    stage(0, 0);
    stage(1, 0);
    IVLM_WAIT_BAR(2);
    for (int t = 0; t < nt; ++t) IVLM_KTILE(t);
#undef IVLM_KTILE
#undef IVLM_WAIT_BAR
#undef IVLM_PHASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-chunk DMAs must not outlive the block's LDS

    // ---- epilogue (whole lines through LDS, gemm_common.h): a wave's first four fragment columns from its own 16-KB slice; then the
    // fifth fragment columns of the four waves of a wave row as ONE shared 128 x 64 strip (bf16 16 KB / fp32 32 KB per wave row), of
    // which every wave stores 32 rows.  (The dispatcher only sends problems for which gemm_whole_lines_ok holds: no per-fragment
    //  fallback here - a second, not fully unrolled loop over the accumulators would index them dynamically and push all 160
    //  registers through scratch.)
#ifdef IVLM_ABL_NOEPI
    if (g.M > 0) {  // (one dummy store per lane keeps the accumulators live)
        float sacc = 0.0f;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) sacc += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
        if (sacc == 123.456f) static_cast<float*>(g.C)[tid] = sacc;
        return;
    }
#endif
    const int mw = m0 + wr * 128;
    __syncthreads();  // every wave is done with the K tiles
    gemm_store_lines<ACT, OUT_F32, 8, 4, OUT_F32 ? 4 : 8, 5, 0>(g, smem + wave * 16384, mw, n0 + wc * 64, lane, acc);
    __syncthreads();  // the strips overlay the slices
    unsigned char* strip = smem + wr * (OUT_F32 ? 32768 : 16384);
    gemm_stage_strip<ACT, OUT_F32, 8, 5>(g, strip, wc * 16, n0 + 256 + wc * 16, lane, acc, 4);
    __syncthreads();
    gemm_store_strip_rows<OUT_F32>(g, strip, wc * 32, 32, mw, n0 + 256, lane);
}

template <int ACT>
int launch320(const GemmArgs& g, hipStream_t st) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + 319) / 320);
    dim3 grid(tiles, 1, g.batch);
#define IVLM_GO(F32, OPK)                                                                                        \
    do {                                                                                                         \
        auto kfn = gemm320_kernel<ACT, F32, OPK>;                                                                \
        static ivlm_dev_mask_t attr_set{0};                                                                      \
        if (ivlm_dev_pending(attr_set)) {                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      kLds320);                                                                  \
            ivlm_dev_done(attr_set);                                                                             \
        }                                                                                                        \
        ivlm_launch(kfn, grid, dim3(512), kLds320, st, g);                                                       \
    } while (0)
    if (g.f16) {
        if (g.out_f32) IVLM_GO(true, 2); else IVLM_GO(false, 2);
    } else {
        if (g.out_f32) IVLM_GO(true, 0); else IVLM_GO(false, 0);
    }
#undef IVLM_GO
    return ivlm_launch_status();
}

}  // namespace

// 256 x 320 tile variant (arguments validated by gemm_bf16; bf16 / fp16 operands, row-major, the epilogues of the SAM encoder)
int gemm_bf16_320p(const GemmArgs& g, hipStream_t st) {
    if (g.fp8 || g.out_fp8 || g.a_kstep || g.w_kstep || g.c_panel) return IVLM_ERR_UNSUPPORTED;
    if (!(g.out_f32 ? gemm_whole_lines_ok<true>(g, g.act) : gemm_whole_lines_ok<false>(g, g.act))) return IVLM_ERR_UNSUPPORTED;
    switch (g.act) {
        case ACT_NONE: return launch320<ACT_NONE>(g, st);
        case ACT_GELU: return launch320<ACT_GELU>(g, st);
        default: return IVLM_ERR_UNSUPPORTED;
    }
}

}  // namespace ivlm
