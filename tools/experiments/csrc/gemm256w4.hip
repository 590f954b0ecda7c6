// 256x256x64 bf16 MFMA GEMM for gfx950, FOUR waves of 128 x 128 (the large-M GEMMs of the SAM ViT-H encoder:
// qkv / proj / mlp of model/segment_anything/modeling/image_encoder.py:222-260, common.py:13-27).
//
// Same contract, operand layouts, LDS image and epilogues as gemm256_kernel (gemm256.hip); different division of labour.  The
// ablation of the 8-wave kernel (tools/experiments/README.md) shows its K loop losing a third of its time whenever the LDS
// fragment reads and the LDS-DMA writes are both on: 8 waves of 128 x 64 read (128 + 64) x 128 B = 24 KB each per K tile, 192
// KB per block, next to 64 KB of DMA writes.  Here
//   * 4 waves (2 x 2), one per SIMD, each owning 128 x 128 of the output = 8 x 8 MFMA fragments (256 accumulator registers):
//     (128 + 128) x 128 B = 32 KB of fragment reads per wave per K tile, 128 KB per block - a third less LDS traffic;
//   * the wave software-pipelines itself: the fragments of k-step kk + 1 are read (into the other register set) while the
//     64 MFMAs of k-step kk issue - the matrix pipe of a SIMD is fed by one instruction stream that never waits for its own
//     reads, instead of two waves taking turns around block barriers;
//   * ONE barrier per K tile: after a wave has waited for its own DMA pieces of tile t + 1 (vmcnt(0): they were issued a whole
//     K tile earlier) and for its reads of tile t, the barrier publishes tile t + 1 and frees tile t's buffer, into which every
//     wave then issues its 16 DMA instructions of tile t + 2 (two K tiles of buffering = 128 KB of LDS, one block per CU).
#include "gemm_common.h"

namespace ivlm {
namespace {

constexpr int kBKw = 64;
constexpr int kStageW = 2 * 256 * kBKw * 2;  // A tile + W tile: 64 KiB
constexpr int kLdsW4 = 2 * kStageW;          // 128 KiB

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int m0, n0;
    gemm_tile_origin(g, 256, 256, m0, n0);
    const bf16_t* __restrict__ A = g.A;
    const bf16_t* __restrict__ W = g.W;

    // ---- DMA sources: wave w copies rows [64 w, 64 w + 64) of the A tile and of the W tile, 8 rows per instruction ---------
    const int64_t a_rs = g.a_kstep ? 64 : g.lda, a_ks = g.a_kstep ? g.a_kstep : kBKw;
    const int64_t w_rs = g.w_kstep ? 64 : g.ldw, w_ks = g.w_kstep ? g.w_kstep : kBKw;
    const int lr = lane >> 3;                                   // row inside an 8-row piece
    const int row0 = wave * 64 + lr;                            // + 8 i
    // source chunk landing in LDS chunk lane & 7 of local row r: (lane & 7) ^ ((r >> 1) & 7); r = row0 + 8 i -> (r >> 1) & 7 =
    // ((row0 >> 1) + 4 i) & 7: two values, by the parity of i
    const int ch0 = (lane & 7) ^ ((row0 >> 1) & 7), ch1 = (lane & 7) ^ (((row0 >> 1) + 4) & 7);
    int offA[8], offW[8];  // element offsets of (row, chunk) at K tile 0 (the host checks that they fit 31 bits)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = (i & 1) ? ch1 : ch0;
        int ra = m0 + row0 + 8 * i;
        ra = ra < g.M ? ra : g.M - 1;
        if (g.a_rows) ra = g.a_rows[ra];
        offA[i] = (int)(ra * a_rs + ch * 8);
        int rn = n0 + row0 + 8 * i;
        rn = rn < g.N ? rn : g.N - 1;
        offW[i] = (int)(rn * w_rs + ch * 8);
    }
    const int nt = (g.K + kBKw - 1) / kBKw;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(kGemmZeroChunk);
    auto stage = [&](int tile) {  // all 16 pieces of K tile `tile` (tiles past the end stream the zero chunk: uniform vmcnt)
        unsigned char* dstA = smem + (tile & 1) * kStageW + wave * 8192;
        unsigned char* dstW = dstA + 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = (i & 1) ? ch1 : ch0;
            const bool ok = tile < nt && ch * 8 + tile * kBKw < g.K;
            glds16(ok ? A + offA[i] + tile * a_ks : zero, dstA + i * 1024);
            glds16(ok ? W + offW[i] + tile * w_ks : zero, dstW + i * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a tile image: [row][128 B], chunk c of row r at c ^ ((r >> 1) & 7)) -----------
    const int sw = ((lane & 15) >> 1) & 7;
    const int rdA = (wr * 128 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);          // + i * 2048, ^ (kk << 6)
    const int rdW = 32768 + (wc * 128 + (lane & 15)) * 128 + (((lane >> 4) ^ sw) << 4);  // + j * 2048, ^ (kk << 6)

    f32x4_t acc[8][8];  // [n fragment][m fragment]
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment registers: A of a k-step in one of two sets (double-buffered across k-steps), W in two HALF sets of four n
    // fragments (double-buffered across the halves of a k-step): 96 registers; a full second copy of both (128) spills
    bf16x8_t fa[2][8], fb[2][4];

    auto read_a = [&](int set, const unsigned char* tb, int kk) {
        const unsigned char* pa = tb + (rdA ^ (kk << 6));  // (the k-step flips chunk bit 2; the fragment index is an immediate)
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[set][i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 2048);
    };
    auto read_w = [&](int set, const unsigned char* tb, int kk, int half) {
        const unsigned char* pw = tb + (rdW ^ (kk << 6)) + half * 8192;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const bf16x8_t*>(pw + j * 2048);
    };
    auto mfmas = [&](int aset, int wset, int half) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[half * 4 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[wset][j], fa[aset][i], acc[half * 4 + j][i], 0, 0, 0);
    };
    // one pipeline stage = half a k-step (32 MFMAs): wait for the fragments it uses (read one stage ago: landed long since),
    // send out the reads of the next stage, then issue the MFMAs
#define IVLM_STAGE(READS, ASET, WSET, HALF)             \
    do {                                                \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ \
        __builtin_amdgcn_sched_barrier(0);              \
        READS;                                          \
        __builtin_amdgcn_sched_barrier(0);              \
        mfmas(ASET, WSET, HALF);                        \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)

    // ---- prologue: tiles 0 and 1 in flight, fragments of (tile 0, k-step 0, half 0) in registers ----------------------------
    stage(0);
    stage(1);
    __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16): this wave's pieces of tile 0 (the builtin keeps the compiler's own
                                         // counter tracking alive; after an inline-asm wait it re-waits for everything)
    __builtin_amdgcn_s_barrier();
    read_a(0, smem, 0);
    read_w(0, smem, 0, 0);

    for (int t = 0; t < nt; ++t) {
        const unsigned char* tb = smem + (t & 1) * kStageW;
        const unsigned char* tn = smem + ((t + 1) & 1) * kStageW;
        IVLM_STAGE({ read_w(1, tb, 0, 1); }, 0, 0, 0);                       // k-step 0, n fragments 0-3
        IVLM_STAGE({ read_a(1, tb, 1); read_w(0, tb, 1, 0); }, 0, 1, 1);     // k-step 0, n fragments 4-7
        IVLM_STAGE({ read_w(1, tb, 1, 1); }, 1, 0, 0);                       // k-step 1, n fragments 0-3
        // tile t is read (its last fragments have landed), this wave's pieces of tile t + 1 have landed: publish / release
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage(t + 2);                        // into tile t's buffer
        IVLM_STAGE({ if (t + 1 < nt) { read_a(0, tn, 0); read_w(0, tn, 0, 0); } }, 1, 1, 1);  // k-step 1, n fragments 4-7
    }
#undef IVLM_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-chunk DMAs must not outlive the block's LDS

    // ---- epilogue: whole lines through LDS (gemm_common.h): 32 KB per wave --------------------------------------------------
    const int mw = m0 + wr * 128, nw = n0 + wc * 128;
    // (the host only picks this kernel when gemm_whole_lines_ok holds)
    __syncthreads();  // every wave is done with the K tiles
    unsigned char* wbuf = smem + wave * 32768;
    if constexpr (OUT_F32) {  // 512-byte rows: the two 64-column halves one after the other (256-byte rows each)
        gemm_store_lines<ACT, true, 8, 4, 8, 8, 0>(g, wbuf, mw, nw, lane, acc);
        gemm_store_lines<ACT, true, 8, 4, 8, 8, 4>(g, wbuf, mw, nw + 64, lane, acc);
    } else {
        gemm_store_lines<ACT, false, 8, 8, 8>(g, wbuf, mw, nw, lane, acc);
    }
}

template <int ACT>
int launch_w4(const GemmArgs& g, hipStream_t st) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
#define IVLM_GO(F32)                                                                                              \
    do {                                                                                                          \
        auto kfn = gemm256w4_kernel<ACT, F32>;                                                                    \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsW4); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        ivlm_launch(kfn, dim3(tiles), dim3(256), kLdsW4, st, g);                                                  \
    } while (0)
    if (g.out_f32) IVLM_GO(true); else IVLM_GO(false);
#undef IVLM_GO
    return ivlm_launch_status();
}

}  // namespace

// four-wave 256^2 variant (bf16 operands, batch 1; arguments already validated by gemm_bf16)
bool gemm_256w4_applies(const GemmArgs& g) {
    if (g.fp8 || g.out_fp8 || g.batch != 1 || (g.act != ACT_NONE && g.act != ACT_GELU)) return false;
    const int64_t a_rs = g.a_kstep ? 64 : g.lda, w_rs = g.w_kstep ? 64 : g.ldw;
    if ((int64_t)g.M * a_rs >= (1ll << 31) || (int64_t)g.N * w_rs >= (1ll << 31)) return false;  // 32-bit element offsets
    if (g.a_rows) return false;  // (gathered rows may point anywhere in a larger buffer)
    return (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
           (g.out_f32 ? ((g.N & 3) == 0 && (g.ldc & 3) == 0 && (!g.residual || (g.ldr & 3) == 0))
                      : ((g.N & 7) == 0 && !g.residual && (g.c_panel ? (g.c_panel & 7) == 0 : (g.ldc & 7) == 0)));
}

int gemm_bf16_256w4(const GemmArgs& g, hipStream_t st) {
    if (!gemm_256w4_applies(g)) return gemm_bf16_256p(g, st);
    switch (g.act) {
        case ACT_NONE: return launch_w4<ACT_NONE>(g, st);
        case ACT_GELU: return launch_w4<ACT_GELU>(g, st);
        default: return IVLM_ERR_UNSUPPORTED;
    }
}

}  // namespace ivlm
