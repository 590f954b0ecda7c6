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
constexpr int kStream = 1024;  // block of the streaming (pixel-major) kernels: 16 waves share one LDS vote table - at one block per CU
                               // that is 4 waves per SIMD with two 16-byte groups in flight per lane (256-thread blocks: 1 wave per SIMD)

// =============================================================================================
// plan build
// =============================================================================================
// torch.clamp / np.clip semantics: a NaN stays a NaN (fminf / fmaxf alone would turn it into a bound, and a NaN mask logit into a
// plausible contact value)
__device__ __forceinline__ float clamp_nan(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ bool triple_ok(int a, int b, int c, int nv) {
    return ((unsigned)a < (unsigned)nv) & ((unsigned)b < (unsigned)nv) & ((unsigned)c < (unsigned)nv);
}

// SLOTS = 3: mesh tables (vid [V,HW,3] + barycentric weights); SLOTS = 1: pixel -> point maps (pid [V,HW], weight 1: the rows of
// the point-major plan of ObjectPCAfford3DPredictor's cached p2pmap files)
template <int SLOTS>
__global__ __launch_bounds__(kBlock) void plan_count_kernel(const int32_t* __restrict__ vid, int V, int64_t HW,
                                                            int nv, int32_t* __restrict__ row_cnt) {
    const int64_t n = (int64_t)V * HW;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const int64_t base = (p / HW) * nv;
        if (SLOTS == 1) {
            const int a = vid[p];
            if ((unsigned)a < (unsigned)nv) atomicAdd(&row_cnt[base + a], 1);
            continue;
        }
        const int a = vid[3 * p], b = vid[3 * p + 1], c = vid[3 * p + 2];
        if (!triple_ok(a, b, c, nv)) continue;
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
template <int SLOTS>
__global__ __launch_bounds__(kBlock) void plan_fill_kernel(const int32_t* __restrict__ vid,
                                                           const float* __restrict__ bary, int V, int64_t HW, int nv,
                                                           const int32_t* __restrict__ row_ptr,
                                                           int32_t* __restrict__ cursor, uint32_t* __restrict__ ent_key,
                                                           float* __restrict__ ent_w, int64_t cap) {
    const int64_t n = (int64_t)V * HW;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const int64_t v = p / HW;
        const uint32_t pix = (uint32_t)(p - v * HW);
        if (SLOTS == 1) {
            const int a = vid[p];
            if ((unsigned)a >= (unsigned)nv) continue;
            const int64_t row = v * nv + a;
            const int64_t pos = (int64_t)row_ptr[row] + atomicAdd(&cursor[row], 1);
            if (pos < cap) {
                ent_key[pos] = pix;
                ent_w[pos] = 1.0f;
            }
            continue;
        }
        int id[3] = {vid[3 * p], vid[3 * p + 1], vid[3 * p + 2]};
        if (!triple_ok(id[0], id[1], id[2], nv)) continue;
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
// MODE 0: soft (clamped logits -> sigmoid, weighted mean, clipped), 1: thresholded sigmoid, 2: plain mean of the map's own values
// (point clouds: ObjectPCAfford3DPredictor, components.py:318-347 - weights 1, no sigmoid, no clip).
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
                    if (MODE == 0) x[k] = clamp_nan(x[k], -param, param);
                    const float m = MODE == 2 ? x[k] : sigmoid_f32(x[k]);  // (MODE 2: the map holds the values to average)
                    if (i + 64 * k < e && (MODE != 1 || m > param)) {
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
                if (!(s_ratio[k] < 0.0f)) {  // (seen; a NaN ratio - NaN logits - counts and makes the vertex NaN, as in the reference)
                    pred += s_ratio[k];
                    seen_views += 1.0f;
                }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (seen_views > 0.0f) pred = pred / seen_views;           // components.py:240-241
        if (MODE == 0) pred = clamp_nan(pred, 0.0f, 1.0f);      // components.py:242 (soft only)
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
__global__ __launch_bounds__(kStream) void lift_dense_kernel(const float* __restrict__ logits,
                                                            const int32_t* __restrict__ vid,
                                                            const float* __restrict__ bary, int V, int64_t HW, int nv,
                                                            float param, float* __restrict__ ws /*[B,V,2,nv]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_votes = reinterpret_cast<float*>(smem);
    float* s_cnt = s_votes + nv;
    const int v = blockIdx.y, b = blockIdx.z;
    // USE_LDS: every block owns a slab [2][nv] of the workspace ([B,V,chunks,2,nv]) and writes its LDS table there with plain
    // stores; the finalize kernel sums the slabs in chunk order.  (Device-scope float atomics execute on the memory side of the 8
    // XCDs: flushing through them cost ~4 us per block-per-CU - tools/sweep_lift_blocks.py - and needed a memset launch.)
    // !USE_LDS (2 nv floats do not fit the LDS): one zeroed slab per (b, v), global atomics.
    float* g_votes = ws + ((((int64_t)b * V + v) * (USE_LDS ? gridDim.x : 1) + (USE_LDS ? blockIdx.x : 0)) * 2) * nv;
    float* g_cnt = g_votes + nv;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < 2 * nv; i += kStream) s_votes[i] = 0.0f;
        __syncthreads();
    }
    float* votes = USE_LDS ? s_votes : g_votes;
    float* cnt = USE_LDS ? s_cnt : g_cnt;

    const float* lg = logits + ((int64_t)b * V + v) * HW;
    const int32_t* vd = vid + (int64_t)v * HW * 3;
    const float* br = bary + (int64_t)v * HW * 3;
    const int64_t ngroups = HW >> 2;  // HW % 4 == 0 checked by the launcher
    // The loop is a chain of dependent memory round trips unless the next group's seven 16-byte loads are issued BEFORE this
    // group's votes (atomics order the iterations for the compiler): one group ahead in registers, clamped index so that the
    // prefetch is always legal.  48 - 86 us -> see profiles/r04_lift_microbench.json.
    const int64_t gstep = (int64_t)gridDim.x * kStream;
    int64_t g = (int64_t)blockIdx.x * kStream + threadIdx.x;
    float4 nx4, nw0, nw1, nw2;
    int4 ni0, ni1, ni2;
    auto fetch = [&](int64_t gg) __attribute__((always_inline)) {
        gg = gg < ngroups ? gg : ngroups - 1;
        nx4 = reinterpret_cast<const float4*>(lg)[gg];
        const int4* ip = reinterpret_cast<const int4*>(vd + 12 * gg);
        const float4* wp = reinterpret_cast<const float4*>(br + 12 * gg);
        ni0 = ip[0]; ni1 = ip[1]; ni2 = ip[2];
        nw0 = wp[0]; nw1 = wp[1]; nw2 = wp[2];
    };
    if (g < ngroups) fetch(g);
    for (; g < ngroups; g += gstep) {
        const float4 x4 = nx4, w0 = nw0, w1 = nw1, w2 = nw2;
        const int4 i0 = ni0, i1 = ni1, i2 = ni2;
        fetch(g + gstep);
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
        const int ids[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
        const float ws_[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        // consecutive pixels of a row mostly lie in the same triangle: votes of a run of pixels with the same vertex triple are
        // summed in registers and flushed once (LDS float atomics to one address serialise: they were a third of this kernel)
        int ca = -1, cb = -1, cc = -1;
        float rv[3] = {0.f, 0.f, 0.f}, rw[3] = {0.f, 0.f, 0.f};
        auto flush = [&]() __attribute__((always_inline)) {
            if (ca < 0) return;
            const int t3[3] = {ca, cb, cc};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                atomicAdd(&votes[t3[k]], rv[k]);
                atomicAdd(&cnt[t3[k]], rw[k]);
            }
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = ids[3 * j], bb = ids[3 * j + 1], c = ids[3 * j + 2];
            if (!triple_ok(a, bb, c, nv)) continue;
            float x = xs[j];
            if (MODE == 0) x = clamp_nan(x, -param, param);
            const float m = sigmoid_f32(x);
            if (MODE == 1 && !(m > param)) continue;
            if (a != ca || bb != cb || c != cc) {
                flush();
                ca = a; cb = bb; cc = c;
#pragma unroll
                for (int k = 0; k < 3; ++k) rv[k] = rw[k] = 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float w = ws_[3 * j + k];
                rv[k] += w * m;
                rw[k] += w;
            }
        }
        flush();
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * nv; i += kStream) g_votes[i] = s_votes[i];
    }
}

// The votes of an (image, view) arrive as `chunks` slabs [2][n] (one per block of the streaming kernel).  Block = 32 consecutive
// vertices x 8 chunk groups: thread (i, cg) sums the slabs cg, cg + 8, ... (independent loads, 128-byte rows per half wave), the
// eight partial sums meet in LDS and are added in group order - a fixed summation order, no atomics.
template <int MODE>
__global__ __launch_bounds__(kBlock) void lift_finalize_kernel(const float* __restrict__ ws, int V, int n, int chunks,
                                                               float* __restrict__ out, float* __restrict__ nviews) {
    constexpr int kPts = 32, kGrp = kBlock / kPts;
    __shared__ float s_v[kGrp][kPts], s_c[kGrp][kPts];
    const int li = threadIdx.x % kPts, cg = threadIdx.x / kPts;
    const int i = blockIdx.x * kPts + li;
    const int b = blockIdx.y;
    const int ic = i < n ? i : n - 1;
    float pred = 0.0f, seen = 0.0f;
    for (int v = 0; v < V; ++v) {
        const float* base = ws + (((int64_t)b * V + v) * chunks * 2) * n;
        float votes = 0.0f, cnt = 0.0f;
#pragma unroll 4
        for (int c = cg; c < chunks; c += kGrp) {
            votes += base[(int64_t)c * 2 * n + ic];
            cnt += base[(int64_t)c * 2 * n + n + ic];
        }
        s_v[cg][li] = votes;
        s_c[cg][li] = cnt;
        __syncthreads();
        if (cg == 0) {
            votes = 0.0f;
            cnt = 0.0f;
#pragma unroll
            for (int k = 0; k < kGrp; ++k) {
                votes += s_v[k][li];
                cnt += s_c[k][li];
            }
            if (cnt > 0.0f) {
                pred += votes / cnt;
                seen += 1.0f;
            }
        }
        __syncthreads();
    }
    if (cg != 0 || i >= n) return;
    if (seen > 0.0f) pred /= seen;
    if (MODE == 0) pred = clamp_nan(pred, 0.0f, 1.0f);
    out[(int64_t)b * n + i] = pred;
    if (nviews) nviews[(int64_t)b * n + i] = seen;
}

// point-cloud lift: votes += value, cnt += 1 per mapped pixel
template <bool USE_LDS>
__global__ __launch_bounds__(kStream) void lift_points_kernel(const float* __restrict__ probs,
                                                             const int32_t* __restrict__ pid, int pid_batched, int V,
                                                             int64_t HW, int np, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_votes = reinterpret_cast<float*>(smem);
    float* s_cnt = s_votes + np;
    const int v = blockIdx.y, b = blockIdx.z;
    float* g_votes = ws + ((((int64_t)b * V + v) * (USE_LDS ? gridDim.x : 1) + (USE_LDS ? blockIdx.x : 0)) * 2) * np;  // (see lift_dense_kernel)
    float* g_cnt = g_votes + np;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < 2 * np; i += kStream) s_votes[i] = 0.0f;
        __syncthreads();
    }
    float* votes = USE_LDS ? s_votes : g_votes;
    float* cnt = USE_LDS ? s_cnt : g_cnt;
    const float* pr = probs + ((int64_t)b * V + v) * HW;
    const int32_t* mp = pid + ((int64_t)(pid_batched ? b : 0) * V + v) * HW;
    const int64_t ngroups = HW >> 2;
    // two groups ahead in registers (see lift_dense_kernel): the loads of the following iterations are in flight while this one votes
    const int64_t gstep = (int64_t)gridDim.x * kStream;
    int64_t g = (int64_t)blockIdx.x * kStream + threadIdx.x;
    float4 px[2];
    int4 pi[2];
    auto fetch = [&](int slot, int64_t gg) __attribute__((always_inline)) {
        gg = gg < ngroups ? gg : ngroups - 1;
        px[slot] = reinterpret_cast<const float4*>(pr)[gg];
        pi[slot] = reinterpret_cast<const int4*>(mp)[gg];
    };
    if (g < ngroups) {
        fetch(0, g);
        fetch(1, g + gstep);
    }
    for (; g < ngroups; g += gstep) {
        const float4 x4 = px[0];
        const int4 i4 = pi[0];
        px[0] = px[1];
        pi[0] = pi[1];
        fetch(1, g + 2 * gstep);
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
        const int ids[4] = {i4.x, i4.y, i4.z, i4.w};
        // runs of equal ids among the thread's 4 consecutive pixels are summed in registers first (a splat covers several pixels
        // of a row): LDS float atomics to one address serialise, they were 40 % of this kernel
        int cur = -1;
        float rs = 0.0f, rc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int id = (unsigned)ids[j] < (unsigned)np ? ids[j] : -1;  // -1 (and anything out of range) = no point
            if (id != cur) {
                if (cur >= 0) {
                    atomicAdd(&votes[cur], rs);
                    atomicAdd(&cnt[cur], rc);
                }
                cur = id;
                rs = 0.0f;
                rc = 0.0f;
            }
            rs += xs[j];
            rc += 1.0f;
        }
        if (cur >= 0) {
            atomicAdd(&votes[cur], rs);
            atomicAdd(&cnt[cur], rc);
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * np; i += kStream) g_votes[i] = s_votes[i];
    }
}

constexpr size_t kLdsBudget = 160 * 1024;

static int g_lift_bpc = 1;  // blocks per CU of the streaming kernels (benchmark hook: ivlm_lift_stream_blocks_per_cu)
constexpr int kMaxChunksPerView = 256;  // bound of the slab count the workspace is sized for

inline int dense_chunks(int B, int V, int64_t HW, int bpc = 0) {
    // ~bpc blocks per CU over the whole launch, at least 1, at most one block per 1024 pixels
    int64_t want = ((bpc > 0 ? bpc : g_lift_bpc) * 256 + (int64_t)B * V - 1) / ((int64_t)B * V);
    int64_t maxc = (HW / 4 + kStream - 1) / kStream;
    if (want > maxc) want = maxc;
    if (want > kMaxChunksPerView) want = kMaxChunksPerView;
    if (want < 1) want = 1;
    return (int)want;
}

// slabs per (image, view) the workspace must hold: the largest chunk count dense_chunks can pick for this B * V (any HW, bpc <= 4)
inline size_t ws_chunks(int B, int V) {
    int64_t want = (4 * 256 + (int64_t)B * V - 1) / ((int64_t)B * V);
    if (want > kMaxChunksPerView) want = kMaxChunksPerView;
    return (size_t)(want < 1 ? 1 : want);
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
    plan_count_kernel<3><<<grid, kBlock, 0, st>>>(vid, V, HW, Nv, row_cnt);
    plan_scan_kernel<<<1, 1024, 0, st>>>(row_cnt, R, row_ptr, nnz_out);
    plan_fill_kernel<3><<<grid, kBlock, 0, st>>>(vid, bary, V, HW, Nv, row_ptr, row_cnt,
                                                 reinterpret_cast<uint32_t*>(ent_pix), ent_w, cap);
    const size_t lds = (sizeof(uint32_t) + sizeof(float)) * kSortLdsMax;
    plan_sort_kernel<<<R < 8192 ? R : 8192, kBlock, lds, st>>>(row_ptr, R, reinterpret_cast<uint32_t*>(ent_pix),
                                                              ent_w);
    return ivlm_launch_status();
}

// point-major plan of a pixel -> point map (pid int32 [V,HW], -1 / out of range = no point): rows sorted by pixel, weights 1;
// evaluated by ivlm_lift_mesh_plan with mode 2 (the plain mean of the map's values per view, then over the views that see the point)
int ivlm_lift_points_plan_build(const int32_t* pid, int V, int64_t HW, int Np, int32_t* row_ptr, int32_t* ent_pix, float* ent_w,
                                int64_t cap, int32_t* nnz_out, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(pid && row_ptr && ent_pix && ent_w && workspace);
    IVLM_CHECK_ARG(V > 0 && HW > 0 && Np > 0 && HW < (1ll << 30) && cap > 0);
    IVLM_CHECK_ARG((int64_t)V * Np < (1ll << 31) - 2 && (int64_t)V * HW < (1ll << 31));
    if (workspace_bytes < ivlm_lift_plan_workspace_bytes(V, HW, Np)) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    const int R = V * Np;
    int32_t* row_cnt = static_cast<int32_t*>(workspace);
    IVLM_HIP_TRY(hipMemsetAsync(row_cnt, 0, sizeof(int32_t) * (size_t)R, st));
    const int64_t n = (int64_t)V * HW;
    const int grid = (int)((n + kBlock - 1) / kBlock < 4096 ? (n + kBlock - 1) / kBlock : 4096);
    plan_count_kernel<1><<<grid, kBlock, 0, st>>>(pid, V, HW, Np, row_cnt);
    plan_scan_kernel<<<1, 1024, 0, st>>>(row_cnt, R, row_ptr, nnz_out);
    plan_fill_kernel<1><<<grid, kBlock, 0, st>>>(pid, nullptr, V, HW, Np, row_ptr, row_cnt, reinterpret_cast<uint32_t*>(ent_pix), ent_w, cap);
    const size_t lds = (sizeof(uint32_t) + sizeof(float)) * kSortLdsMax;
    plan_sort_kernel<<<R < 8192 ? R : 8192, kBlock, lds, st>>>(row_ptr, R, reinterpret_cast<uint32_t*>(ent_pix), ent_w);
    return ivlm_launch_status();
}

int ivlm_lift_mesh_plan(const float* logits, const int32_t* row_ptr, const int32_t* ent_pix, const float* ent_w,
                        int B, int V, int64_t HW, int Nv, int mode, float param, float* out, float* nviews,
                        ivlm_stream_t stream) {
    IVLM_CHECK_ARG(logits && row_ptr && ent_pix && ent_w && out);
    IVLM_CHECK_ARG(B > 0 && V > 0 && HW > 0 && Nv > 0 && B <= 65535 && (mode == 0 || mode == 1 || mode == 2));
    dim3 grid(8 * ((Nv + 7) / 8), B);
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    if (mode == 0)
        ivlm_launch(lift_plan_kernel<0>, dim3(grid), dim3(kBlock), 0, st, logits, row_ptr, ent_pix, ent_w, V, HW, Nv, param, out, nviews);
    else if (mode == 2)
        ivlm_launch(lift_plan_kernel<2>, dim3(grid), dim3(kBlock), 0, st, logits, row_ptr, ent_pix, ent_w, V, HW, Nv, param, out, nviews);
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
    if (B <= 0 || V <= 0 || Nv <= 0) return 0;
    return sizeof(float) * 2 * (size_t)B * V * Nv * ws_chunks(B, V);  // one [2][Nv] slab per block of an (image, view)
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
    const size_t lds = sizeof(float) * 2 * (size_t)Nv;
    const bool use_lds = lds <= kLdsBudget - 1024;
    int chunks = dense_chunks(B, V, HW);
    if ((size_t)chunks > ws_chunks(B, V)) chunks = (int)ws_chunks(B, V);
    if (!use_lds) IVLM_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(float) * 2 * (size_t)B * V * Nv, st));  // (atomics path: one slab per view)
    dim3 grid(chunks, V, B);
#define IVLM_LAUNCH_DENSE(MODE, LDS)                                                                        \
    do {                                                                                                    \
        auto kfn = lift_dense_kernel<MODE, LDS>;                                                            \
        if (LDS && lds > 64 * 1024)                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        kfn<<<grid, kStream, LDS ? lds : 0, st>>>(logits, vid, bary, V, HW, Nv, param, ws);                   \
    } while (0)
    if (mode == 0) {
        if (use_lds) IVLM_LAUNCH_DENSE(0, true); else IVLM_LAUNCH_DENSE(0, false);
    } else {
        if (use_lds) IVLM_LAUNCH_DENSE(1, true); else IVLM_LAUNCH_DENSE(1, false);
    }
#undef IVLM_LAUNCH_DENSE
    dim3 fgrid((Nv + 31) / 32, B);
    if (mode == 0)
        lift_finalize_kernel<0><<<fgrid, kBlock, 0, st>>>(ws, V, Nv, use_lds ? chunks : 1, out, nviews);
    else
        lift_finalize_kernel<1><<<fgrid, kBlock, 0, st>>>(ws, V, Nv, use_lds ? chunks : 1, out, nviews);
    return ivlm_launch_status();
}

int ivlm_lift_stream_blocks_per_cu(int bpc) {  // benchmark hook; returns the previous value
    const int prev = g_lift_bpc;
    if (bpc > 0) g_lift_bpc = bpc;
    return prev;
}

size_t ivlm_lift_points_workspace_bytes(int B, int V, int Np) {
    if (B <= 0 || V <= 0 || Np <= 0) return 0;
    return sizeof(float) * 2 * (size_t)B * V * Np * ws_chunks(B, V);
}

int ivlm_lift_points(const float* probs, const int32_t* pid, int pid_batched, int B, int V, int64_t HW, int Np,
                     float* out, float* nviews, void* workspace, size_t workspace_bytes, ivlm_stream_t stream) {
    IVLM_CHECK_ARG(probs && pid && out && workspace);
    IVLM_CHECK_ARG(B > 0 && V > 0 && HW > 0 && Np > 0 && HW % 4 == 0 && B <= 65535 && V <= 65535);
    const size_t need = ivlm_lift_points_workspace_bytes(B, V, Np);
    if (workspace_bytes < need) return IVLM_ERR_WORKSPACE;
    hipStream_t st = ivlm_stream(stream);
    ivlm_enter();
    float* ws = static_cast<float*>(workspace);
    const size_t lds = sizeof(float) * 2 * (size_t)Np;
    const bool use_lds = lds <= kLdsBudget - 1024;
    int chunks = dense_chunks(B, V, HW);
    if ((size_t)chunks > ws_chunks(B, V)) chunks = (int)ws_chunks(B, V);
    if (!use_lds) IVLM_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(float) * 2 * (size_t)B * V * Np, st));
    dim3 grid(chunks, V, B);
    if (use_lds) {
        auto kfn = lift_points_kernel<true>;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        kfn<<<grid, kStream, lds, st>>>(probs, pid, pid_batched, V, HW, Np, ws);
    } else {
        lift_points_kernel<false><<<grid, kStream, 0, st>>>(probs, pid, pid_batched, V, HW, Np, ws);
    }
    dim3 fgrid((Np + 31) / 32, B);
    lift_finalize_kernel<1><<<fgrid, kBlock, 0, st>>>(ws, V, Np, use_lds ? chunks : 1, out, nviews);  // MODE 1: no clamp
    return ivlm_launch_status();
}

}  // extern "C"
