// Steps right after the path in both callers, kept on the device:
//   contact_prf   binary contact precision / recall / F1 per sample  (utils/eval_utils.py:63-94, threshold 0.5,
//                 gt > 0), one block per sample, integer counts => exact
//   spmv_csr      SMPL -> SMPL-X contact transfer (utils/utils.py:428-443 `convert_contacts`: a dense
//                 [10475 x 6890] bmm, 289 MB) as a CSR SpMV over the ~3 non-zeros per row of that matrix
//   h_geo_metric  geodesic false-positive / false-negative distances (utils/eval_utils.py:129-151) over the 190 MB
//                 vertex-to-vertex distance matrix, streamed once per sample
#include <algorithm>

#include "kernels.h"

namespace ivlm {
namespace {

__global__ __launch_bounds__(256) void contact_prf_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                          int n, float thr, float* __restrict__ out /*[B,3]*/) {
    __shared__ int s[3][4];
    const int b = blockIdx.x;
    int tp = 0, pp = 0, ap = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const bool p = pred[(int64_t)b * n + i] >= thr, g = gt[(int64_t)b * n + i] > 0.0f;
        tp += p && g;
        pp += p;
        ap += g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tp += __shfl_xor(tp, off, 64);
        pp += __shfl_xor(pp, off, 64);
        ap += __shfl_xor(ap, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s[0][threadIdx.x >> 6] = tp;
        s[1][threadIdx.x >> 6] = pp;
        s[2][threadIdx.x >> 6] = ap;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (float)(s[0][0] + s[0][1] + s[0][2] + s[0][3]);
        const float p = (float)(s[1][0] + s[1][1] + s[1][2] + s[1][3]);
        const float a = (float)(s[2][0] + s[2][1] + s[2][2] + s[2][3]);
        const float precision = t / (p + 1e-10f), recall = t / (a + 1e-10f);
        out[3 * b] = 2.0f * precision * recall / (precision + recall + 1e-10f);
        out[3 * b + 1] = precision;
        out[3 * b + 2] = recall;
    }
}

// y[b, r] = sum_j val[j] * x[b, col[j]], j in [row_ptr[r], row_ptr[r+1])
__global__ __launch_bounds__(256) void spmv_csr_kernel(const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col, const float* __restrict__ val,
                                                       const float* __restrict__ x, int rows, int cols,
                                                       float* __restrict__ y) {
    const int r = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (r >= rows) return;
    float acc = 0.0f;
    for (int j = row_ptr[r]; j < row_ptr[r + 1]; ++j) acc += val[j] * x[(int64_t)b * cols + col[j]];
    y[(int64_t)b * rows + r] = acc;
}


// ---- get_h_geo_metric (utils/eval_utils.py:129-151): geodesic false-positive / false-negative distances ---------------
// Per sample: columns = vertices with gt == 1 (all columns if there are none), rows = vertices with pred >= 0.5 (all rows if
// none); E = D[rows][:, columns];  fp = mean over rows of min over columns, fn = mean over columns of min over rows.
// D is the 6890 x 6890 fp32 geodesic matrix (190 MB): the selected rows are streamed ONCE - every block owns a contiguous
// range of rows, a thread owns the columns t, t + 256, ...; the row minimum over the selected columns is a block
// reduction, the column minima over the selected rows stay in registers for the whole range and are merged with one
// atomicMin per (block, column) on an order-preserving integer key.  HBM-bound: rows_selected * Nv * 4 bytes.
constexpr int kGeoCols = 32;  // columns per thread: Nv <= 8192

__device__ __forceinline__ uint32_t f32_order_key(float f) {  // monotonic float -> uint (also for negative values)
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// sel[0..n) = row selection, sel[n..2n) = column selection, with the reference's "none selected -> all" rule
__global__ __launch_bounds__(1024) void geo_select_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n,
                                                          uint8_t* __restrict__ sel, uint32_t* __restrict__ colkey,
                                                          float* __restrict__ rowmin) {
    __shared__ int any_row, any_col;
    if (threadIdx.x == 0) any_row = any_col = 0;
    __syncthreads();
    int ar = 0, ac = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        ar |= pred[i] >= 0.5f;
        ac |= gt[i] == 1.0f;
    }
    if (ar) any_row = 1;
    if (ac) any_col = 1;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        sel[i] = any_row ? (pred[i] >= 0.5f) : 1;
        sel[n + i] = any_col ? (gt[i] == 1.0f) : 1;
        colkey[i] = 0xffffffffu;
        rowmin[i] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void geo_min_kernel(const float* __restrict__ dist, int n, const uint8_t* __restrict__ sel,
                                                      int rows_per_block, float* __restrict__ rowmin,
                                                      uint32_t* __restrict__ colkey) {
    __shared__ float red[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint8_t* rsel = sel;
    const uint8_t* csel = sel + n;
    uint32_t cmask = 0;
    float cm[kGeoCols];
#pragma unroll
    for (int j = 0; j < kGeoCols; ++j) {
        const int c = t + 256 * j;
        cm[j] = __builtin_inff();
        if (c < n && csel[c]) cmask |= 1u << j;
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, n);
    for (int r = r0; r < r1; ++r) {
        if (!rsel[r]) continue;  // block-uniform
        const float* row = dist + (int64_t)r * n;
        float rm = __builtin_inff();
#pragma unroll
        for (int j = 0; j < kGeoCols; ++j) {
            const int c = t + 256 * j;
            if (c < n) {
                const float d = row[c];
                cm[j] = fminf(cm[j], d);
                if ((cmask >> j) & 1u) rm = fminf(rm, d);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rm = fminf(rm, __shfl_xor(rm, o, 64));
        if (lane == 0) red[wave] = rm;
        __syncthreads();
        if (t == 0) rowmin[r] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < kGeoCols; ++j) {
        const int c = t + 256 * j;
        if (c < n && cm[j] < __builtin_inff()) atomicMin(colkey + c, f32_order_key(cm[j]));
    }
}

// out[0] = mean_{rows selected} rowmin, out[1] = mean_{columns selected} colmin (fp64 accumulation, one block)
__global__ __launch_bounds__(1024) void geo_finish_kernel(const float* __restrict__ rowmin, const uint32_t* __restrict__ colkey,
                                                          const uint8_t* __restrict__ sel, int n, float* __restrict__ out) {
    __shared__ double sh[4][16];
    double sr = 0.0, sc = 0.0, nr = 0.0, nc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        if (sel[i]) { sr += rowmin[i]; nr += 1.0; }
        if (sel[n + i]) { sc += f32_from_key(colkey[i]); nc += 1.0; }
    }
    double v[4] = {sr, nr, sc, nc};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if (lane == 0) sh[k][wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k)
            for (int w = 0; w < 16; ++w) a[k] += sh[k][w];
        out[0] = (float)(a[0] / a[1]);
        out[1] = (float)(a[2] / a[3]);
    }
}

}  // namespace

size_t h_geo_workspace_bytes(int n) { return (size_t)n * (4 + 4 + 2) + 64; }

int h_geo_metric(const float* dist, const float* pred, const float* gt, int B, int n, float* out, void* ws, size_t ws_bytes,
                 hipStream_t st) {
    if (!dist || !pred || !gt || !out || !ws || B <= 0 || n <= 0) return IVLM_ERR_INVALID_ARG;
    if (n > 256 * kGeoCols) return IVLM_ERR_UNSUPPORTED;
    if (ws_bytes < h_geo_workspace_bytes(n)) return IVLM_ERR_WORKSPACE;
    float* rowmin = static_cast<float*>(ws);
    uint32_t* colkey = reinterpret_cast<uint32_t*>(rowmin + n);
    uint8_t* sel = reinterpret_cast<uint8_t*>(colkey + n);
    const int blocks = std::min(n, 512);
    const int rpb = (n + blocks - 1) / blocks;
    for (int b = 0; b < B; ++b) {
        geo_select_kernel<<<1, 1024, 0, st>>>(pred + (int64_t)b * n, gt + (int64_t)b * n, n, sel, colkey, rowmin);
        geo_min_kernel<<<(n + rpb - 1) / rpb, 256, 0, st>>>(dist, n, sel, rpb, rowmin, colkey);
        geo_finish_kernel<<<1, 1024, 0, st>>>(rowmin, colkey, sel, n, out + 2 * b);
    }
    return ivlm_launch_status();
}

int contact_prf(const float* gt, const float* pred, int B, int n, float thr, float* out, hipStream_t st) {
    if (!gt || !pred || !out || B <= 0 || n <= 0) return IVLM_ERR_INVALID_ARG;
    contact_prf_kernel<<<B, 256, 0, st>>>(gt, pred, n, thr, out);
    return ivlm_launch_status();
}

int spmv_csr(const int32_t* row_ptr, const int32_t* col, const float* val, const float* x, int B, int rows, int cols,
             float* y, hipStream_t st) {
    if (!row_ptr || !col || !val || !x || !y || B <= 0 || rows <= 0 || cols <= 0) return IVLM_ERR_INVALID_ARG;
    spmv_csr_kernel<<<dim3((rows + 255) / 256, B), 256, 0, st>>>(row_ptr, col, val, x, rows, cols, y);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" {
size_t ivlm_h_geo_workspace_bytes(int n) { return ivlm::h_geo_workspace_bytes(n); }
int ivlm_h_geo_metric(const float* dist, const float* pred, const float* gt, int B, int n, float* out, void* workspace,
                      size_t workspace_bytes, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::h_geo_metric(dist, pred, gt, B, n, out, workspace, workspace_bytes, ivlm_stream(s));
}
int ivlm_contact_prf(const float* gt, const float* pred, int B, int n, float thr, float* out, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::contact_prf(gt, pred, B, n, thr, out, ivlm_stream(s));
}
int ivlm_spmv_csr(const int32_t* row_ptr, const int32_t* col, const float* val, const float* x, int B, int rows,
                  int cols, float* y, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::spmv_csr(row_ptr, col, val, x, B, rows, cols, y, ivlm_stream(s));
}
}
