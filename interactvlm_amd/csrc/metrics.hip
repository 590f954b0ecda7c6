// Steps right after the path in both callers, kept on the device:
//   contact_prf   binary contact precision / recall / F1 per sample  (utils/eval_utils.py:63-94, threshold 0.5,
//                 gt > 0), one block per sample, integer counts => exact
//   spmv_csr      SMPL -> SMPL-X contact transfer (utils/utils.py:428-443 `convert_contacts`: a dense
//                 [10475 x 6890] bmm, 289 MB) as a CSR SpMV over the ~3 non-zeros per row of that matrix
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

}  // namespace

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
