// Render-Localize-Lift kernels for gfx950: multi-view 2D masks -> per-vertex / per-point contact.
//
// Replaces (citations into /root/reference):
//   HumanContact3DPredictor        model/components.py:195-277
//   ObjectMeshContact3DPredictor   model/components.py:350-489
//   ObjectPCAfford3DPredictor      model/components.py:279-347
//
// Two formulations of the bary-weighted vote:
//   * "plan" (vertex-major CSR gather): the constant pixel->vertex tables are inverted ONCE
//     (ivlm_lift_plan_build); at run time one wave owns one vertex, streams its (pixel, weight)
//     entries with coalesced 4-byte loads, gathers the logits, and reduces with a fixed wave
//     butterfly.  No atomics, no workspace, bit-reproducible, single launch.
//   * "dense" (pixel-major streaming): for single-use tables. 16-byte coalesced loads of
//     logits/ids/weights, votes privatised in LDS (2*Nv floats <= 160 KB), one flush per block.
// Both are HBM/L2-bound byte shuffles: no MFMA here by design.
#include "bilinear.h"

namespace {

constexpr int kBlock = 256;

// =============================================================================================
// plan build
// =============================================================================================
__device__ __forceinline__ bool triple_ok(int a, int b, int c, int nv) {
    return ((unsigned)a < (unsigned)nv) & ((unsigned)b < (unsigned)nv) & ((unsigned)c < (unsigned)nv);
}

__global__ __launch_bounds__(kBlock) void plan_count_kernel(const int32_t* __restrict__ vid, int V, int64_t HW,
                                                            int nv, int32_t* __restrict__ row_cnt) {
    const int64_t n = (int64_t)V * HW;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const int a = vid[3 * p], b = vid[3 * p + 1], c = vid[3 * p + 2];
        if (!triple_ok(a, b, c, nv)) continue;
        const int64_t base = (p / HW) * nv;
        atomicAdd(&row_cnt[base + a], 1);
        atomicAdd(&row_cnt[base + b], 1);
        atomicAdd(&row_cnt[base + c], 1);
    }
}

// single-block exclusive scan of row_cnt[R] -> row_ptr[R+1]; zeroes row_cnt (re-used as cursor)
__global__ __launch_bounds__(1024) void plan_scan_kernel(int32_t* __restrict__ row_cnt, int R,
                                                         int32_t* __restrict__ row_ptr, int32_t* __restrict__ nnz) {
    __shared__ int32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (R + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, R);
    int32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += row_cnt[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        int32_t v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int32_t run = tid ? s_part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        int32_t c = row_cnt[i];
        row_ptr[i] = run;
        row_cnt[i] = 0;
        run += c;
    }
    if (tid == 1023) {
        row_ptr[R] = s_part[1023];
        if (nnz) *nnz = s_part[1023];
    }
}

// key = slot k (2 bits) << 30 | pixel  (HW < 2^30)
__global__ __launch_bounds__(kBlock) void plan_fill_kernel(const int32_t* __restrict__ vid,
                                                           const float* __restrict__ bary, int V, int64_t HW, int nv,
                                                           const int32_t* __restrict__ row_ptr,
                                                           int32_t* __restrict__ cursor, uint32_t* __restrict__ ent_key,
                                                           float* __restrict__ ent_w, int64_t cap) {
    const int64_t n = (int64_t)V * HW;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        int id[3] = {vid[3 * p], vid[3 * p + 1], vid[3 * p + 2]};
        if (!triple_ok(id[0], id[1], id[2], nv)) continue;
        const int64_t v = p / HW;
        const uint32_t pix = (uint32_t)(p - v * HW);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t row = v * nv + id[k];
            const int64_t pos = (int64_t)row_ptr[row] + atomicAdd(&cursor[row], 1);
            if (pos < cap) {
                ent_key[pos] = ((uint32_t)k << 30) | pix;
                ent_w[pos] = bary[3 * p + k];
            }
        }
    }
}

// Sort each row by key so that the per-vertex summation order is (slot, pixel) — the reference's
// three scatter passes in pixel order — and independent of the atomic fill order above.
// One block per row, bitonic network over the row padded (virtually) to a power of two.
// standard bitonic network over npow2 LDS-resident (key, weight) pairs; pad keys = 0xffffffff
__device__ void bitonic_row(uint32_t* key, float* w, int npow2) {
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const uint32_t ki = key[i], kl = key[l];
                    if ((ki > kl) == up) {
                        key[i] = kl;
                        key[l] = ki;
                        const float t = w[i];
                        w[i] = w[l];
                        w[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
}

constexpr int kSortLdsMax = 8192;  // 64 KB of LDS (key + weight)

__global__ __launch_bounds__(kBlock) void plan_sort_kernel(const int32_t* __restrict__ row_ptr, int R,
                                                           uint32_t* __restrict__ ent_key, float* __restrict__ ent_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const int s = row_ptr[row], n = row_ptr[row + 1] - s;
        if (n <= 0) continue;
        if (n > 1) {
            int npow2 = 1;
            while (npow2 < n) npow2 <<= 1;
            // the bitonic network needs real +inf padding, so it runs on an LDS copy when the padded
            // row fits; longer rows fall back to an in-place odd-even transposition.
            if (npow2 <= kSortLdsMax) {
                uint32_t* sk = reinterpret_cast<uint32_t*>(smem);
                float* sw = reinterpret_cast<float*>(smem + sizeof(uint32_t) * kSortLdsMax);
                for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                    sk[i] = i < n ? ent_key[s + i] : 0xffffffffu;
                    sw[i] = i < n ? ent_w[s + i] : 0.0f;
                }
                __syncthreads();
                bitonic_row(sk, sw, npow2);
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    ent_key[s + i] = sk[i];
                    ent_w[s + i] = sw[i];
                }
                __syncthreads();
            } else {
                // long rows (low-poly objects filling the frame): odd-even transposition in global
                // memory, O(n^2/threads) but one-time and rare.
                uint32_t* gk = ent_key + s;
                float* gw = ent_w + s;
                for (int pass = 0; pass < n; ++pass) {
                    for (int i = 2 * threadIdx.x + (pass & 1); i + 1 < n; i += 2 * blockDim.x) {
                        const uint32_t a = gk[i], b = gk[i + 1];
                        if (a > b) {
                            gk[i] = b;
                            gk[i + 1] = a;
                            const float t = gw[i];
                            gw[i] = gw[i + 1];
                            gw[i + 1] = t;
                        }
                    }
                    __syncthreads();
                }
            }
        }
        // strip the slot bits: run-time kernels only need the pixel
        for (int i = threadIdx.x; i < n; i += blockDim.x) ent_key[s + i] &= 0x3fffffffu;
        __syncthreads();
    }
}

// =============================================================================================
// plan gather: one block per (image, vertex), its waves stride the views
// =============================================================================================
// The rows of a vertex are short (~180 entries per view for SMPL in 1024^2 renders), so the kernel is bound by the
// dependent-load chain row_ptr -> entry -> logit, not by bytes.  One wave per (vertex, view) keeps that chain at three
// round trips, and the 4-deep predicated unroll puts every entry load of a typical row in flight at once.
// Summation order is fixed: lane-strided partial sums in index order, butterfly wave_sum, views added in view order
// by one thread - bit-reproducible run to run.
template <int MODE, typename Fetch>
__device__ __forceinline__ void lift_plan_body(Fetch fetch, const int32_t* __restrict__ row_ptr,
                                               const int32_t* __restrict__ ent_pix, const float* __restrict__ ent_w,
                                               int V, int nv, float param, float* __restrict__ out,
                                               float* __restrict__ nviews) {
    constexpr int kWaves = kBlock / 64;
    __shared__ float s_ratio[kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of vertex ids, i.e. (for a mesh
    // numbered coherently, like SMPL) one region of each view, so its L2 fetches ~1/8 of the masks instead of all
    const int chunk = gridDim.x >> 3;
    const int vert = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3), b = blockIdx.y;
    if (vert >= nv) return;  // whole block
    float pred = 0.0f, seen_views = 0.0f;  // thread 0 only
    for (int v0 = 0; v0 < V; v0 += kWaves) {
        const int v = v0 + wave;
        float ratio = -1.0f;  // "vertex not seen in this view"
        if (v < V) {
            const int row = v * nv + vert;
            const int s = row_ptr[row], e = row_ptr[row + 1];
            float votes = 0.0f, cnt = 0.0f;
            for (int i = s + lane; i < e; i += 256) {
                int p[4];
                float w[4], x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = min(i + 64 * k, e - 1);  // clamped: the load is always legal, the add is predicated
                    p[k] = ent_pix[j];
                    w[k] = ent_w[j];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = fetch(v, p[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (MODE == 0) x[k] = fminf(fmaxf(x[k], -param), param);
                    const float m = sigmoid_f32(x[k]);
                    if (i + 64 * k < e && (MODE == 0 || m > param)) {
                        votes += w[k] * m;
                        cnt += w[k];
                    }
                }
            }
            votes = wave_sum(votes);
            cnt = wave_sum(cnt);
            if (cnt > 0.0f) ratio = votes / cnt;  // components.py:273-277
        }
        if (lane == 0) s_ratio[wave] = ratio;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 0; k < kWaves && v0 + k < V; ++k)
                if (s_ratio[k] >= 0.0f) {
                    pred += s_ratio[k];
                    seen_views += 1.0f;
                }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (seen_views > 0.0f) pred = pred / seen_views;           // components.py:240-241
        if (MODE == 0) pred = fminf(fmaxf(pred, 0.0f), 1.0f);      // components.py:242 (soft only)
        out[(int64_t)b * nv + vert] = pred;
        if (nviews) nviews[(int64_t)b * nv + vert] = seen_views;
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void lift_plan_kernel(const float* __restrict__ logits,
                                                           const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ ent_pix,
                                                           const float* __restrict__ ent_w, int V, int64_t HW, int nv,
                                                           float param, float* __restrict__ out,
                                                           float* __restrict__ nviews) {
    const float* __restrict__ lg = logits + (int64_t)blockIdx.y * V * HW;
    lift_plan_body<MODE>([&](int v, int p) { return lg[(int64_t)v * HW + p]; }, row_ptr, ent_pix, ent_w, V, nv, param,
                         out, nviews);
}

// Same gather, but the logit of a pixel is evaluated on the fly from the 256x256 low-res mask with the exact
// arithmetic of Sam.postprocess_masks (bilinear.h): the full-resolution masks are never read back.
// Measured (profiles/r01_lift_microbench.json): four L1/L2 gathers + address math per entry cost more than the one
// gather from the freshly written (Infinity-Cache resident) full-res mask, so the two-step path stays the default.
template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void lift_plan_lowres_kernel(const T* __restrict__ low, int lh, int lw, int img,
                                                                  int in_h, int in_w, int oh, int ow,
                                                                  const int32_t* __restrict__ row_ptr,
                                                                  const int32_t* __restrict__ ent_pix,
                                                                  const float* __restrict__ ent_w, int V, int nv,
                                                                  float param, float* __restrict__ out,
                                                                  float* __restrict__ nviews) {
    const T* __restrict__ lo = low + (int64_t)blockIdx.y * V * lh * lw;
    lift_plan_body<MODE>(
        [&](int v, int p) {
            const int y = p / ow, x = p - y * ow;
            return ivlm_bilinear::postprocess_at(lo + (int64_t)v * lh * lw, lh, lw, img, in_h, in_w, oh, ow, y, x);
        },
        row_ptr, ent_pix, ent_w, V, nv, param, out, nviews);
}

// =============================================================================================
// dense streaming variant
// =============================================================================================
// grid (chunks, V, B); each thread walks groups of 4 consecutive pixels (16-byte loads).
template <int MODE, bool USE_LDS>
__global__ __launch_bounds__(kBlock) void lift_dense_kernel(const float* __restrict__ logits,
                                                            const int32_t* __restrict__ vid,
                                                            const float* __restrict__ bary, int V, int64_t HW, int nv,
                                                            float param, float* __restrict__ ws /*[B,V,2,nv]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_votes = reinterpret_cast<float*>(smem);
    float* s_cnt = s_votes + nv;
    const int v = blockIdx.y, b = blockIdx.z;
    float* g_votes = ws + (((int64_t)b * V + v) * 2) * nv;
    float* g_cnt = g_votes + nv;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < 2 * nv; i += kBlock) s_votes[i] = 0.0f;
        __syncthreads();
    }
    float* votes = USE_LDS ? s_votes : g_votes;
    float* cnt = USE_LDS ? s_cnt : g_cnt;

    const float* lg = logits + ((int64_t)b * V + v) * HW;
    const int32_t* vd = vid + (int64_t)v * HW * 3;
    const float* br = bary + (int64_t)v * HW * 3;
    const int64_t ngroups = HW >> 2;  // HW % 4 == 0 checked by the launcher
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * kBlock) {
        const float4 x4 = reinterpret_cast<const float4*>(lg)[g];
        const int4* ip = reinterpret_cast<const int4*>(vd + 12 * g);
        const float4* wp = reinterpret_cast<const float4*>(br + 12 * g);
        const int4 i0 = ip[0], i1 = ip[1], i2 = ip[2];
        const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
        const int ids[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
        const float ws_[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = ids[3 * j], bb = ids[3 * j + 1], c = ids[3 * j + 2];
            if (!triple_ok(a, bb, c, nv)) continue;
            float x = xs[j];
            if (MODE == 0) x = fminf(fmaxf(x, -param), param);
            const float m = sigmoid_f32(x);
            if (MODE == 1 && !(m > param)) continue;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float w = ws_[3 * j + k];
                atomicAdd(&votes[ids[3 * j + k]], w * m);
                atomicAdd(&cnt[ids[3 * j + k]], w);
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < nv; i += kBlock) {
            const float vv = s_votes[i], cc = s_cnt[i];
            if (vv != 0.0f) atomicAdd(&g_votes[i], vv);
            if (cc != 0.0f) atomicAdd(&g_cnt[i], cc);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void lift_finalize_kernel(const float* __restrict__ ws, int V, int n,
                                                               float* __restrict__ out, float* __restrict__ nviews) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    float pred = 0.0f, seen = 0.0f;
    for (int v = 0; v < V; ++v) {
        const float* base = ws + (((int64_t)b * V + v) * 2) * n;
        const float votes = base[i], cnt = base[n + i];
        if (cnt > 0.0f) {
            pred += votes / cnt;
            seen += 1.0f;
        }
    }
    if (seen > 0.0f) pred /= seen;
    if (MODE == 0) pred = fminf(fmaxf(pred, 0.0f), 1.0f);
    out[(int64_t)b * n + i] = pred;
    if (nviews) nviews[(int64_t)b * n + i] = seen;
}

// point-cloud lift: votes += value, cnt += 1 per mapped pixel
template <bool USE_LDS>
__global__ __launch_bounds__(kBlock) void lift_points_kernel(const float* __restrict__ probs,
                                                             const int32_t* __restrict__ pid, int pid_batched, int V,
                                                             int64_t HW, int np, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_votes = reinterpret_cast<float*>(smem);
    float* s_cnt = s_votes + np;
    const int v = blockIdx.y, b = blockIdx.z;
    float* g_votes = ws + (((int64_t)b * V + v) * 2) * np;
    float* g_cnt = g_votes + np;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < 2 * np; i += kBlock) s_votes[i] = 0.0f;
        __syncthreads();
    }
    float* votes = USE_LDS ? s_votes : g_votes;
    float* cnt = USE_LDS ? s_cnt : g_cnt;
    const float* pr = probs + ((int64_t)b * V + v) * HW;
    const int32_t* mp = pid + ((int64_t)(pid_batched ? b : 0) * V + v) * HW;
    const int64_t ngroups = HW >> 2;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * kBlock) {
        const float4 x4 = reinterpret_cast<const float4*>(pr)[g];
        const int4 i4 = reinterpret_cast<const int4*>(mp)[g];
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
        const int ids[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((unsigned)ids[j] < (unsigned)np) {  // -1 (and anything out of range) = no point
                atomicAdd(&votes[ids[j]], xs[j]);
                atomicAdd(&cnt[ids[j]], 1.0f);
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += kBlock) {
            const float cc = s_cnt[i];
            if (cc != 0.0f) {
                atomicAdd(&g_votes[i], s_votes[i]);
                atomicAdd(&g_cnt[i], cc);
            }
        }
    }
}

constexpr size_t kLdsBudget = 160 * 1024;

inline int dense_chunks(int B, int V, int64_t HW) {
    // ~2 blocks per CU over the whole launch, at least 1, at most one block per 1024 pixels
    int64_t want = (2 * 256 + (int64_t)B * V - 1) / ((int64_t)B * V);
    int64_t maxc = (HW / 4 + kBlock - 1) / kBlock;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

size_t ivlm_lift_plan_workspace_bytes(int V, int64_t HW, int Nv) {
    (void)HW;
    return sizeof(int32_t) * ((size_t)V * Nv + 16);
}

int ivlm_lift_plan_build(const int32_t* vid, const float* bary, int V, int64_t HW, int Nv, int32_t* row_ptr,
                         int32_t* ent_pix, float* ent_w, int64_t cap, int32_t* nnz_out, void* workspace,
                         size_t workspace_bytes, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(vid && bary && row_ptr && ent_pix && ent_w && workspace);
    IVLM_CHECK_ARG(V > 0 && HW > 0 && Nv > 0 && HW < (1ll << 30) && cap > 0);
    IVLM_CHECK_ARG((int64_t)V * Nv < (1ll << 31) - 2 && (int64_t)V * HW * 3 < (1ll << 31));
    if (workspace_bytes < ivlm_lift_plan_workspace_bytes(V, HW, Nv)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    const int R = V * Nv;
    int32_t* row_cnt = static_cast<int32_t*>(workspace);
    IVLM_HIP_TRY(hipMemsetAsync(row_cnt, 0, sizeof(int32_t) * (size_t)R, st));
    const int64_t n = (int64_t)V * HW;
    const int grid = (int)((n + kBlock - 1) / kBlock < 4096 ? (n + kBlock - 1) / kBlock : 4096);
    plan_count_kernel<<<grid, kBlock, 0, st>>>(vid, V, HW, Nv, row_cnt);
    plan_scan_kernel<<<1, 1024, 0, st>>>(row_cnt, R, row_ptr, nnz_out);
    plan_fill_kernel<<<grid, kBlock, 0, st>>>(vid, bary, V, HW, Nv, row_ptr, row_cnt,
                                              reinterpret_cast<uint32_t*>(ent_pix), ent_w, cap);
    const size_t lds = (sizeof(uint32_t) + sizeof(float)) * kSortLdsMax;
    plan_sort_kernel<<<R < 8192 ? R : 8192, kBlock, lds, st>>>(row_ptr, R, reinterpret_cast<uint32_t*>(ent_pix),
                                                              ent_w);
    return ivlm_launch_status();
}

int ivlm_lift_mesh_plan(const float* logits, const int32_t* row_ptr, const int32_t* ent_pix, const float* ent_w,
                        int B, int V, int64_t HW, int Nv, int mode, float param, float* out, float* nviews,
                        ivlm_stream_t stream) {
    IVLM_CHECK_ARG(logits && row_ptr && ent_pix && ent_w && out);
    IVLM_CHECK_ARG(B > 0 && V > 0 && HW > 0 && Nv > 0 && B <= 65535 && (mode == 0 || mode == 1));
    dim3 grid(8 * ((Nv + 7) / 8), B);
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    if (mode == 0)
        ivlm_launch(lift_plan_kernel<0>, dim3(grid), dim3(kBlock), 0, st, logits, row_ptr, ent_pix, ent_w, V, HW, Nv, param, out, nviews);
    else
        ivlm_launch(lift_plan_kernel<1>, dim3(grid), dim3(kBlock), 0, st, logits, row_ptr, ent_pix, ent_w, V, HW, Nv, param, out, nviews);
    return ivlm_launch_status();
}

int ivlm_lift_mesh_plan_lowres(const void* low, int dtype, int lh, int lw, int img, int in_h, int in_w, int oh, int ow,
                               const int32_t* row_ptr, const int32_t* ent_pix, const float* ent_w, int B, int V, int Nv,
                               int mode, float param, float* out, float* nviews, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(low && row_ptr && ent_pix && ent_w && out);
    IVLM_CHECK_ARG(B > 0 && V > 0 && Nv > 0 && B <= 65535 && (mode == 0 || mode == 1));
    IVLM_CHECK_ARG(lh > 0 && lw > 0 && img > 0 && in_h > 0 && in_w > 0 && in_h <= img && in_w <= img && oh > 0 && ow > 0);
    dim3 grid(8 * ((Nv + 7) / 8), B);
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
#define IVLM_LL(MODE, T)                                                                                          \
    lift_plan_lowres_kernel<MODE, T><<<grid, kBlock, 0, st>>>(static_cast<const T*>(low), lh, lw, img, in_h, in_w, oh, \
                                                              ow, row_ptr, ent_pix, ent_w, V, Nv, param, out, nviews)
    if (dtype == IVLM_F32) {
        if (mode == 0) IVLM_LL(0, float); else IVLM_LL(1, float);
    } else if (dtype == IVLM_BF16) {
        if (mode == 0) IVLM_LL(0, bf16_t); else IVLM_LL(1, bf16_t);
    } else {
        return IVLM_ERR_UNSUPPORTED;
    }
#undef IVLM_LL
    return ivlm_launch_status();
}

size_t ivlm_lift_mesh_dense_workspace_bytes(int B, int V, int Nv) {
    return sizeof(float) * 2 * (size_t)B * V * Nv;
}

int ivlm_lift_mesh_dense(const float* logits, const int32_t* vid, const float* bary, int B, int V, int64_t HW,
                         int Nv, int mode, float param, float* out, float* nviews, void* workspace,
                         size_t workspace_bytes, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(logits && vid && bary && out && workspace);
    IVLM_CHECK_ARG(B > 0 && V > 0 && HW > 0 && Nv > 0 && (mode == 0 || mode == 1));
    IVLM_CHECK_ARG(HW % 4 == 0 && B <= 65535 && V <= 65535);
    const size_t need = ivlm_lift_mesh_dense_workspace_bytes(B, V, Nv);
    if (workspace_bytes < need) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    float* ws = static_cast<float*>(workspace);
    IVLM_HIP_TRY(hipMemsetAsync(ws, 0, need, st));
    const size_t lds = sizeof(float) * 2 * (size_t)Nv;
    const bool use_lds = lds <= kLdsBudget - 1024;
    dim3 grid(dense_chunks(B, V, HW), V, B);
#define IVLM_LAUNCH_DENSE(MODE, LDS)                                                                        \
    do {                                                                                                    \
        auto kfn = lift_dense_kernel<MODE, LDS>;                                                            \
        if (LDS && lds > 64 * 1024)                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        kfn<<<grid, kBlock, LDS ? lds : 0, st>>>(logits, vid, bary, V, HW, Nv, param, ws);                   \
    } while (0)
    if (mode == 0) {
        if (use_lds) IVLM_LAUNCH_DENSE(0, true); else IVLM_LAUNCH_DENSE(0, false);
    } else {
        if (use_lds) IVLM_LAUNCH_DENSE(1, true); else IVLM_LAUNCH_DENSE(1, false);
    }
#undef IVLM_LAUNCH_DENSE
    dim3 fgrid((Nv + kBlock - 1) / kBlock, B);
    if (mode == 0)
        lift_finalize_kernel<0><<<fgrid, kBlock, 0, st>>>(ws, V, Nv, out, nviews);
    else
        lift_finalize_kernel<1><<<fgrid, kBlock, 0, st>>>(ws, V, Nv, out, nviews);
    return ivlm_launch_status();
}

size_t ivlm_lift_points_workspace_bytes(int B, int V, int Np) { return sizeof(float) * 2 * (size_t)B * V * Np; }

int ivlm_lift_points(const float* probs, const int32_t* pid, int pid_batched, int B, int V, int64_t HW, int Np,
                     float* out, float* nviews, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(probs && pid && out && workspace);
    IVLM_CHECK_ARG(B > 0 && V > 0 && HW > 0 && Np > 0 && HW % 4 == 0 && B <= 65535 && V <= 65535);
    const size_t need = ivlm_lift_points_workspace_bytes(B, V, Np);
    if (workspace_bytes < need) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    float* ws = static_cast<float*>(workspace);
    IVLM_HIP_TRY(hipMemsetAsync(ws, 0, need, st));
    const size_t lds = sizeof(float) * 2 * (size_t)Np;
    const bool use_lds = lds <= kLdsBudget - 1024;
    dim3 grid(dense_chunks(B, V, HW), V, B);
    if (use_lds) {
        auto kfn = lift_points_kernel<true>;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        kfn<<<grid, kBlock, lds, st>>>(probs, pid, pid_batched, V, HW, Np, ws);
    } else {
        lift_points_kernel<false><<<grid, kBlock, 0, st>>>(probs, pid, pid_batched, V, HW, Np, ws);
    }
    dim3 fgrid((Np + kBlock - 1) / kBlock, B);
    lift_finalize_kernel<1><<<fgrid, kBlock, 0, st>>>(ws, V, Np, out, nviews);  // MODE 1: no clamp
    return ivlm_launch_status();
}

}  // extern "C"
