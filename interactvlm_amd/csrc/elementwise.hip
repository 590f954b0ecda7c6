// Data-movement / elementwise kernels of the contact-inference path (gfx950).  All HBM-bound: 16-byte
// coalesced accesses, no LDS, grid-stride.
//   im2col_nchw        patch embedding as GEMM   (SAM image_encoder.py:404-426; HF CLIP patch_embedding)
//   im2col3x3_nhwc     SAM neck 3x3 conv as GEMM (image_encoder.py:92-108)
//   gather_rows        window partition / unpartition(+residual), token-embedding gather, CLS drop
//                      (image_encoder.py:263-318; llava_arch.py:185-208)
//   add_rows           x + pe with row-modulo broadcast (transformer.py:160-176)
//   dense_pe           PositionEmbeddingRandom over the 64x64 grid (prompt_encoder.py:189-229)
//   rope_kv            rotate-half RoPE on q,k in the fused qkv buffer + KV-cache append (HF LlamaAttention)
//   mask_dot           masks = hyper_in @ upscaled  (mask_decoder.py:150-153) with the pixel un-shuffle of
//                      the two k2s2 transposed convolutions
#include "kernels.h"

namespace ivlm {
namespace {

constexpr int kT = 256;

inline int grid_for(int64_t work_items, int max_blocks = 8192) {
    int64_t b = (work_items + kT - 1) / kT;
    if (b < 1) b = 1;
    return (int)(b < max_blocks ? b : max_blocks);
}

// out[(b,gy,gx), (c,ky,kx)] = x[b,c,gy*s+ky,gx*s+kx], zero-padded to Kpad columns. 8 outputs per thread.
__global__ __launch_bounds__(kT) void im2col_nchw_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B,
                                                         int C, int H, int W, int ks, int stride, int Kpad) {
    const int gh = (H - ks) / stride + 1, gw = (W - ks) / stride + 1;
    const int K = C * ks * ks;
    const int64_t total = (int64_t)B * gh * gw * (Kpad >> 3);
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int c8 = (int)(i % (Kpad >> 3));
        const int64_t row = i / (Kpad >> 3);
        const int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((int64_t)gw * gh));
        bf16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = c8 * 8 + e;
            bf16_t val = 0;
            if (col < K) {
                const int c = col / (ks * ks), r = col - c * ks * ks, ky = r / ks, kx = r - ky * ks;
                val = x[(((int64_t)b * C + c) * H + gy * stride + ky) * W + gx * stride + kx];
            }
            v[e] = val;
        }
        *reinterpret_cast<uint4*>(out + row * Kpad + c8 * 8) = *reinterpret_cast<const uint4*>(v);
    }
}

// x [B,H,W,C] -> out [(b,y,x), (ky,kx,c)] with zero padding 1; C % 8 == 0
__global__ __launch_bounds__(kT) void im2col3x3_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                            int B, int H, int W, int C, int64_t ldx, int64_t ldo) {
    const int c8n = C >> 3;
    const int64_t total = (int64_t)B * H * W * 9 * c8n;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int c8 = (int)(i % c8n);
        const int tap = (int)((i / c8n) % 9);
        const int64_t row = i / ((int64_t)c8n * 9);
        const int xx = (int)(row % W), yy = (int)((row / W) % H), b = (int)(row / ((int64_t)W * H));
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (sy >= 0 && sy < H && sx >= 0 && sx < W)
            v = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + sy) * W + sx) * ldx + c8 * 8);
        *reinterpret_cast<uint4*>(out + row * ldo + tap * C + c8 * 8) = v;
    }
}

// ---- 8-element row fragments in either storage type ---------------------------------------------------------------
// kind: 0 = bf16, 1 = fp32, 2 (stores only) = "split": the row is written as [hi(cols) | lo(cols)] bf16 with
// x = hi + lo to 2^-17 - the A operand of a GEMM against [W | W] (K' = 2K), i.e. an fp32-activation GEMM on the bf16 MFMA
// path (used where the FLOPs are negligible and the precision is not: the SAM mask decoder).
__device__ __forceinline__ void load8(const void* base, int kind, int64_t elem, float* f) {
    if (kind == 1) {
        const float4* p = reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
        const float4 a = p[0], b = p[1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(base) + elem);
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
}
// 8 fp32 values -> 8 e4m3 bytes of x / scale, clamped to the format's +-448
__device__ __forceinline__ uint2 pack_fp8x8(const float* f, float inv_scale) {
    float c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(f[j] * inv_scale, -448.0f), 448.0f);
    uint32_t a = 0, b = 0;
    a = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], a, false);
    a = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], a, true);
    b = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], b, false);
    b = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], b, true);
    return make_uint2(a, b);
}

__device__ __forceinline__ void store8(void* base, int kind, int64_t elem, int cols, const float* f, float inv_scale = 1.0f) {
    if (kind == 3) {  // e4m3 bytes
        *reinterpret_cast<uint2*>(static_cast<uint8_t*>(base) + elem) = pack_fp8x8(f, inv_scale);
        return;
    }
    if (kind == 1) {
        float4* p = reinterpret_cast<float4*>(static_cast<float*>(base) + elem);
        p[0] = make_float4(f[0], f[1], f[2], f[3]);
        p[1] = make_float4(f[4], f[5], f[6], f[7]);
        return;
    }
    bf16_t* o = static_cast<bf16_t*>(base) + elem;
    *reinterpret_cast<uint4*>(o) = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                                              pack_bf16x2(f[6], f[7]));
    if (kind == 2) {
        float l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) l[j] = f[j] - bf16_to_f32(f32_to_bf16(f[j]));
        *reinterpret_cast<uint4*>(o + cols) = make_uint4(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]),
                                                         pack_bf16x2(l[4], l[5]), pack_bf16x2(l[6], l[7]));
    }
}

// dst[r] = (idx ? (idx[r] >= 0 ? src[idx[r]] : 0) : src[r]) (+ add[r]);  cols % 8 == 0; row strides in elements
__global__ __launch_bounds__(kT) void gather_rows_kernel(void* __restrict__ dst, int dst_kind, int64_t ldd,
                                                         const void* __restrict__ src, int src_kind, int64_t lds_,
                                                         const int32_t* __restrict__ idx, const void* __restrict__ add,
                                                         int add_kind, int64_t lda, int64_t rows, int cols,
                                                         const float* __restrict__ scale) {
    const int c8n = cols >> 3;
    const int64_t total = rows * c8n;
    const float inv_scale = scale ? 1.0f / *scale : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int c8 = (int)(i % c8n);
        const int64_t r = i / c8n;
        const int64_t s = idx ? (int64_t)idx[r] : r;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (s >= 0) load8(src, src_kind, s * lds_ + c8 * 8, v);
        if (add) {
            float a[8];
            load8(add, add_kind, r * lda + c8 * 8, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += a[j];
        }
        store8(dst, dst_kind, r * ldd + c8 * 8, cols, v, inv_scale);
    }
}

// max |x| of a tensor into *out (fp32 bits; *out >= 0 initialised by the caller): per-tensor fp8 scale calibration
__global__ __launch_bounds__(kT) void amax_kernel(const void* __restrict__ x, int kind, int64_t n8, float* __restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kT) {
        float v[8];
        load8(x, kind, i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));  // m >= 0: bit order = value order
}

// dst[idx[r]] = row (one bf16 row broadcast to a list of destination rows): the q|k|v rows of the zero-padded window positions
// are the bias alone (image_encoder.py:179-183: padded tokens are zeros AFTER norm1), so they are filled, not computed
__global__ __launch_bounds__(kT) void fill_rows_kernel(bf16_t* __restrict__ dst, int64_t ldd, const int32_t* __restrict__ idx,
                                                       int64_t n_idx, const bf16_t* __restrict__ row, int cols) {
    const int c8n = cols >> 3;
    const int64_t total = n_idx * c8n;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int c8 = (int)(i % c8n);
        const int64_t r = idx[i / c8n];
        *reinterpret_cast<uint4*>(dst + r * ldd + c8 * 8) = *reinterpret_cast<const uint4*>(row + c8 * 8);
    }
}

// bf16 -> IEEE fp16, 8 elements per thread (the tail one by one)
__global__ __launch_bounds__(kT) void bf16_to_f16_kernel(const bf16_t* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kT) {
        const uint4 u = *reinterpret_cast<const uint4*>(src + i * 8);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        uint4 o;
        uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int j = 0; j < 4; ++j) ow[j] = pack_f16x2(__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u));
        *reinterpret_cast<uint4*>(dst + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = n8 * 8 + threadIdx.x;
        dst[i] = (uint16_t)(pack_f16x2(bf16_to_f32(src[i]), 0.f) & 0xffffu);
    }
}

// out[r] = a[r] (+|*) b[r % b_rows]
__global__ __launch_bounds__(kT) void add_rows_kernel(void* __restrict__ out, int out_kind, const void* __restrict__ a,
                                                      int a_kind, const void* __restrict__ b, int b_kind, int64_t rows,
                                                      int cols, int64_t b_rows, int op) {
    const int c8n = cols >> 3;
    const int64_t total = rows * c8n;
    const int64_t ldo = out_kind == 2 ? 2 * (int64_t)cols : cols;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int c8 = (int)(i % c8n);
        const int64_t r = i / c8n;
        float va[8], vb[8];
        load8(a, a_kind, r * cols + c8 * 8, va);
        load8(b, b_kind, (r % b_rows) * cols + c8 * 8, vb);
#pragma unroll
        for (int j = 0; j < 8; ++j) va[j] = op ? va[j] * vb[j] : va[j] + vb[j];
        store8(out, out_kind, r * ldo + c8 * 8, cols, va);
    }
}

// pe[(y,x), c]: c < F -> sin, c >= F -> cos of 2*pi*((2*(x+.5)/w-1)*g[0,c'] + (2*(y+.5)/h-1)*g[1,c'])
__global__ __launch_bounds__(kT) void dense_pe_kernel(const float* __restrict__ gauss /*[2,F]*/,
                                                      void* __restrict__ pe /*[h*w, 2F]*/, int pe_f32, int h, int w, int F) {
    const int64_t total = (int64_t)h * w * 2 * F;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int c = (int)(i % (2 * F));
        const int64_t p = i / (2 * F);
        const int xx = (int)(p % w), yy = (int)(p / w);
        const int f = c < F ? c : c - F;
        // fp32 throughout (the reference's bf16 model would round the grid itself; we stay closer to fp32)
        const float cx = 2.0f * (((float)xx + 0.5f) / (float)w) - 1.0f;
        const float cy = 2.0f * (((float)yy + 0.5f) / (float)h) - 1.0f;
        float t = cx * gauss[f] + cy * gauss[F + f];
        t = 6.283185307179586f * t;
        const float val = c < F ? sinf(t) : cosf(t);
        if (pe_f32) static_cast<float*>(pe)[i] = val;
        else static_cast<bf16_t*>(pe)[i] = f32_to_bf16(val);
    }
}

// qkv [T, 3, H, D] (row stride ld): rotate-half RoPE in place on q and k at positions pos0 + t; append
// roped k and v to the caches [Tmax, H, D] at row pos0 + t.  One thread per (t, h, pair j < D/2).
// cos/sin table [T, D/2] in fp32 (HF computes them in fp32 as well)
__global__ __launch_bounds__(kT) void rope_table_kernel(float* __restrict__ ct, float* __restrict__ st, int T, int D,
                                                        float theta) {
    const int half = D >> 1;
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= T * half) return;
    const int j = i % half, pos = i / half;
    const float ang = (float)pos * powf(theta, -(float)(2 * j) / (float)D);
    ct[i] = cosf(ang);
    st[i] = sinf(ang);
}

// F16: qkv and the caches hold IEEE halves (the fp16-operand prefill)
template <bool F16>
__global__ __launch_bounds__(kT) void rope_kv_kernel(bf16_t* __restrict__ qkv, int64_t ld, int T, int H, int D,
                                                     int pos0, float theta, bf16_t* __restrict__ kcache,
                                                     bf16_t* __restrict__ vcache, const float* __restrict__ ct,
                                                     const float* __restrict__ stab) {
    const int half = D >> 1;
    const int64_t total = (int64_t)T * H * half;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int j = (int)(i % half);
        const int h = (int)((i / half) % H);
        const int t = (int)(i / ((int64_t)half * H));
        const int pos = pos0 + t;
        // HF: inv_freq = theta^(-2j/D) (fp32), freqs = pos*inv_freq, cos/sin in fp32; we keep fp32 products and
        // round q/k once (the reference's bf16 model rounds cos/sin and every product: more noise, same maths)
        float c, s;
        if (ct) {
            c = ct[pos * half + j];
            s = stab[pos * half + j];
        } else {
            const float ang = (float)pos * powf(theta, -(float)(2 * j) / (float)D);
            c = cosf(ang);
            s = sinf(ang);
        }
        bf16_t* row = qkv + (int64_t)t * ld;
        bf16_t* q = row + h * D;
        bf16_t* k = row + (int64_t)H * D + h * D;
        const bf16_t* v = row + 2 * (int64_t)H * D + h * D;
        const float q0 = h16_to_f32<F16>(q[j]), q1 = h16_to_f32<F16>(q[j + half]);
        const float k0 = h16_to_f32<F16>(k[j]), k1 = h16_to_f32<F16>(k[j + half]);
        // q*cos + rotate_half(q)*sin
        const bf16_t qa = f32_to_h16<F16>(q0 * c - q1 * s), qb = f32_to_h16<F16>(q1 * c + q0 * s);
        const bf16_t ka = f32_to_h16<F16>(k0 * c - k1 * s), kb = f32_to_h16<F16>(k1 * c + k0 * s);
        q[j] = qa;
        q[j + half] = qb;
        k[j] = ka;
        k[j + half] = kb;
        if (kcache) {
            bf16_t* kc = kcache + ((int64_t)pos * H + h) * D;
            bf16_t* vc = vcache + ((int64_t)pos * H + h) * D;
            kc[j] = ka;
            kc[j + half] = kb;
            vc[j] = v[j];
            vc[j + half] = v[j + half];
        }
    }
}

// "parity" precision of rope_kv: qkv rows are [hi(3HD) | lo(3HD)] bf16 (the split output of the q|k|v GEMM); q and k are rotated
// in fp32 on hi + lo and written back as hi + lo; K / V are appended to hi + lo cache planes (no rounding of the cached rows).
__global__ __launch_bounds__(kT) void rope_kv_split_kernel(bf16_t* __restrict__ qkv, int64_t ld, int T, int H, int D, int pos0,
                                                           bf16_t* __restrict__ kcache, bf16_t* __restrict__ kcache_lo,
                                                           bf16_t* __restrict__ vcache, bf16_t* __restrict__ vcache_lo,
                                                           const float* __restrict__ ct, const float* __restrict__ stab) {
    const int half = D >> 1;
    const int64_t total = (int64_t)T * H * half;
    const int64_t lo = 3 * (int64_t)H * D;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int j = (int)(i % half);
        const int h = (int)((i / half) % H);
        const int t = (int)(i / ((int64_t)half * H));
        const int pos = pos0 + t;
        const float c = ct[pos * half + j], s = stab[pos * half + j];
        bf16_t* row = qkv + (int64_t)t * ld;
        bf16_t* q = row + h * D;
        bf16_t* k = row + (int64_t)H * D + h * D;
        const bf16_t* v = row + 2 * (int64_t)H * D + h * D;
        auto val = [&](const bf16_t* p, int e) { return bf16_to_f32(p[e]) + bf16_to_f32(p[lo + e]); };
        const float q0 = val(q, j), q1 = val(q, j + half), k0 = val(k, j), k1 = val(k, j + half);
        const float qa = q0 * c - q1 * s, qb = q1 * c + q0 * s, ka = k0 * c - k1 * s, kb = k1 * c + k0 * s;
        auto put = [&](bf16_t* p, int e, float x, bf16_t* ch, bf16_t* cl) {
            const bf16_t hi = f32_to_bf16(x), l = f32_to_bf16(x - bf16_to_f32(hi));
            p[e] = hi;
            p[lo + e] = l;
            if (ch) {
                ch[e] = hi;
                cl[e] = l;
            }
        };
        const int64_t crow = ((int64_t)pos * H + h) * D;
        put(q, j, qa, nullptr, nullptr);
        put(q, j + half, qb, nullptr, nullptr);
        put(k, j, ka, kcache ? kcache + crow : nullptr, kcache ? kcache_lo + crow : nullptr);
        put(k, j + half, kb, kcache ? kcache + crow : nullptr, kcache ? kcache_lo + crow : nullptr);
        if (kcache) {
            vcache[crow + j] = v[j];
            vcache[crow + j + half] = v[j + half];
            vcache_lo[crow + j] = v[lo + j];
            vcache_lo[crow + j + half] = v[lo + j + half];
        }
    }
}

// up [B, gh, gw, 2,2, 2,2, C] (two k2s2 transposed convs, channels last)  x  hyper [B, C]
//   -> low [B, 4gh, 4gw] fp32 at (4y + 2dy + dy2, 4x + 2dx + dx2)
__global__ __launch_bounds__(kT) void mask_dot_kernel(const void* __restrict__ up, const void* __restrict__ hyper, int kind,
                                                      float* __restrict__ low, int B, int gh, int gw, int C) {
    const int64_t total = (int64_t)B * gh * gw * 16;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int sub = (int)(i & 15);
        const int64_t cell = i >> 4;
        const int xx = (int)(cell % gw), yy = (int)((cell / gw) % gh), b = (int)(cell / ((int64_t)gw * gh));
        const int dy = sub >> 3, dx = (sub >> 2) & 1, dy2 = (sub >> 1) & 1, dx2 = sub & 1;
        float acc = 0.0f;
        for (int c = 0; c < C; c += 8) {
            float a[8], h[8];
            load8(up, kind, i * C + c, a);
            load8(hyper, kind, (int64_t)b * C + c, h);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += a[j] * h[j];
        }
        const int Y = 4 * yy + 2 * dy + dy2, X = 4 * xx + 2 * dx + dx2;
        low[((int64_t)b * 4 * gh + Y) * (4 * gw) + X] = acc;
    }
}

// uint8 HWC crop -> (x - mean) / std, zero-padded to [3, OH, OW] (pad AFTER the normalisation, like the reference)
template <typename T>
__global__ __launch_bounds__(kT) void normalize_pad_kernel(const uint8_t* __restrict__ src, int W, int y0, int x0,
                                                           int ch, int cw, float m0, float m1, float m2, float s0,
                                                           float s1, float s2, T* __restrict__ out, int OH, int OW) {
    const int64_t total = (int64_t)OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
        const int x = (int)(i % OW), y = (int)(i / OW);
        float r = 0.f, g = 0.f, b = 0.f;
        if (y < ch && x < cw) {
            const uint8_t* p = src + ((int64_t)(y0 + y) * W + (x0 + x)) * 3;
            r = ((float)p[0] - m0) / s0;
            g = ((float)p[1] - m1) / s1;
            b = ((float)p[2] - m2) / s2;
        }
        if (sizeof(T) == 2) {
            reinterpret_cast<bf16_t*>(out)[i] = f32_to_bf16(r);
            reinterpret_cast<bf16_t*>(out)[total + i] = f32_to_bf16(g);
            reinterpret_cast<bf16_t*>(out)[2 * total + i] = f32_to_bf16(b);
        } else {
            reinterpret_cast<float*>(out)[i] = r;
            reinterpret_cast<float*>(out)[total + i] = g;
            reinterpret_cast<float*>(out)[2 * total + i] = b;
        }
    }
}

}  // namespace

int normalize_pad_u8(const uint8_t* src, int H, int W, int y0, int x0, int ch, int cw, const float* mean3,
                     const float* std3, void* out, int out_bf16, int OH, int OW, hipStream_t st) {
    if (!src || !out || !mean3 || !std3 || y0 < 0 || x0 < 0 || y0 + ch > H || x0 + cw > W || ch > OH || cw > OW)
        return IVLM_ERR_INVALID_ARG;
    const int grid = grid_for((int64_t)OH * OW);
    if (out_bf16)
        normalize_pad_kernel<bf16_t><<<grid, kT, 0, st>>>(src, W, y0, x0, ch, cw, mean3[0], mean3[1], mean3[2], std3[0],
                                                          std3[1], std3[2], static_cast<bf16_t*>(out), OH, OW);
    else
        normalize_pad_kernel<float><<<grid, kT, 0, st>>>(src, W, y0, x0, ch, cw, mean3[0], mean3[1], mean3[2], std3[0],
                                                         std3[1], std3[2], static_cast<float*>(out), OH, OW);
    return ivlm_launch_status();
}

int im2col_nchw(const bf16_t* x, bf16_t* out, int B, int C, int H, int W, int ks, int stride, int Kpad,
                hipStream_t st) {
    if (!x || !out || (Kpad & 7) || Kpad < C * ks * ks) return IVLM_ERR_INVALID_ARG;
    const int gh = (H - ks) / stride + 1, gw = (W - ks) / stride + 1;
    im2col_nchw_kernel<<<grid_for((int64_t)B * gh * gw * (Kpad >> 3)), kT, 0, st>>>(x, out, B, C, H, W, ks, stride, Kpad);
    return ivlm_launch_status();
}
int im2col3x3_nhwc(const bf16_t* x, bf16_t* out, int B, int H, int W, int C, hipStream_t st, int64_t ldx, int64_t ldo) {
    if (!x || !out || (C & 7)) return IVLM_ERR_INVALID_ARG;
    if (ldx == 0) ldx = C;
    if (ldo == 0) ldo = 9 * (int64_t)C;
    if (ldx < C || ldo < 9 * (int64_t)C || ((ldx | ldo) & 7)) return IVLM_ERR_INVALID_ARG;
    im2col3x3_nhwc_kernel<<<grid_for((int64_t)B * H * W * 9 * (C >> 3)), kT, 0, st>>>(x, out, B, H, W, C, ldx, ldo);
    return ivlm_launch_status();
}
int gather_rows(void* dst, int dst_kind, int64_t ldd, const void* src, int src_kind, int64_t lds_, const int32_t* idx,
                const void* add, int add_kind, int64_t lda, int64_t rows, int cols, hipStream_t st, const float* scale) {
    if (!dst || !src || rows <= 0 || (cols & 7) || (ldd & 7) || (lds_ & 7) || (lda & 7)) return IVLM_ERR_INVALID_ARG;
    if (dst_kind < 0 || dst_kind > 3 || (src_kind & ~1) || (add_kind & ~1)) return IVLM_ERR_INVALID_ARG;
    if (dst_kind == 2 && ldd < 2 * (int64_t)cols) return IVLM_ERR_INVALID_ARG;
    if (dst_kind == 3 && !scale) return IVLM_ERR_INVALID_ARG;
    gather_rows_kernel<<<grid_for(rows * (cols >> 3)), kT, 0, st>>>(dst, dst_kind, ldd, src, src_kind, lds_, idx, add,
                                                                    add_kind, lda, rows, cols, scale);
    return ivlm_launch_status();
}
int amax(const void* x, int kind, int64_t n, float* out, hipStream_t st) {
    if (!x || !out || n <= 0 || (n & 7) || (kind & ~1)) return IVLM_ERR_INVALID_ARG;
    amax_kernel<<<grid_for(n >> 3, 2048), kT, 0, st>>>(x, kind, n >> 3, out);
    return ivlm_launch_status();
}
int add_rows(void* out, int out_kind, const void* a, int a_kind, const void* b, int b_kind, int64_t rows, int cols,
             int64_t b_rows, hipStream_t st, int op) {
    if (!out || !a || !b || rows <= 0 || b_rows <= 0 || (cols & 7)) return IVLM_ERR_INVALID_ARG;
    if (out_kind < 0 || out_kind > 2 || (a_kind & ~1) || (b_kind & ~1)) return IVLM_ERR_INVALID_ARG;
    add_rows_kernel<<<grid_for(rows * (cols >> 3)), kT, 0, st>>>(out, out_kind, a, a_kind, b, b_kind, rows, cols, b_rows, op);
    return ivlm_launch_status();
}
int fill_rows(bf16_t* dst, int64_t ldd, const int32_t* idx, int64_t n_idx, const bf16_t* row, int cols, hipStream_t st) {
    if (n_idx == 0) return IVLM_OK;  // (an empty index tensor has a null data pointer: nothing to fill, not an error)
    if (!dst || !idx || !row || n_idx < 0 || (cols & 7) || (ldd & 7)) return IVLM_ERR_INVALID_ARG;
    fill_rows_kernel<<<grid_for(n_idx * (cols >> 3)), kT, 0, st>>>(dst, ldd, idx, n_idx, row, cols);
    return ivlm_launch_status();
}
int dense_pe(const float* gauss, void* pe, int pe_f32, int h, int w, int F, hipStream_t st) {
    if (!gauss || !pe) return IVLM_ERR_INVALID_ARG;
    dense_pe_kernel<<<grid_for((int64_t)h * w * 2 * F), kT, 0, st>>>(gauss, pe, pe_f32, h, w, F);
    return ivlm_launch_status();
}
int rope_kv(bf16_t* qkv, int64_t ld, int T, int H, int D, int pos0, float theta, bf16_t* kcache, bf16_t* vcache,
            hipStream_t st, const float* cos_tab, const float* sin_tab, int f16) {
    if (!qkv || T <= 0 || (D & 1) || (kcache && !vcache) || (cos_tab && !sin_tab)) return IVLM_ERR_INVALID_ARG;
    if (f16)
        rope_kv_kernel<true><<<grid_for((int64_t)T * H * (D >> 1)), kT, 0, st>>>(qkv, ld, T, H, D, pos0, theta, kcache, vcache,
                                                                                 cos_tab, sin_tab);
    else
        rope_kv_kernel<false><<<grid_for((int64_t)T * H * (D >> 1)), kT, 0, st>>>(qkv, ld, T, H, D, pos0, theta, kcache, vcache,
                                                                                  cos_tab, sin_tab);
    return ivlm_launch_status();
}
int rope_kv_split(bf16_t* qkv, int64_t ld, int T, int H, int D, int pos0, bf16_t* kcache, bf16_t* kcache_lo, bf16_t* vcache,
                  bf16_t* vcache_lo, const float* cos_tab, const float* sin_tab, hipStream_t st) {
    if (!qkv || T <= 0 || (D & 1) || !cos_tab || !sin_tab || ld < 6LL * H * D) return IVLM_ERR_INVALID_ARG;
    if (kcache && (!kcache_lo || !vcache || !vcache_lo)) return IVLM_ERR_INVALID_ARG;
    rope_kv_split_kernel<<<grid_for((int64_t)T * H * (D >> 1)), kT, 0, st>>>(qkv, ld, T, H, D, pos0, kcache, kcache_lo, vcache,
                                                                             vcache_lo, cos_tab, sin_tab);
    return ivlm_launch_status();
}
int rope_table(float* cos_tab, float* sin_tab, int T, int D, float theta, hipStream_t st) {
    if (!cos_tab || !sin_tab || T <= 0 || (D & 1)) return IVLM_ERR_INVALID_ARG;
    rope_table_kernel<<<(T * (D >> 1) + kT - 1) / kT, kT, 0, st>>>(cos_tab, sin_tab, T, D, theta);
    return ivlm_launch_status();
}
int mask_dot(const void* up, const void* hyper, int kind, float* low, int B, int gh, int gw, int C, hipStream_t st) {
    if (!up || !hyper || !low || (C & 7) || (kind & ~1)) return IVLM_ERR_INVALID_ARG;
    mask_dot_kernel<<<grid_for((int64_t)B * gh * gw * 16), kT, 0, st>>>(up, hyper, kind, low, B, gh, gw, C);
    return ivlm_launch_status();
}

}  // namespace ivlm

extern "C" {
#define BF(p) static_cast<bf16_t*>(p)
#define CBF(p) static_cast<const bf16_t*>(p)
int ivlm_im2col_nchw(const void* x, void* out, int B, int C, int H, int W, int ks, int stride, int Kpad,
                     ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::im2col_nchw(CBF(x), BF(out), B, C, H, W, ks, stride, Kpad, ivlm_stream(s));
}
int ivlm_im2col3x3_nhwc(const void* x, void* out, int B, int H, int W, int C, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::im2col3x3_nhwc(CBF(x), BF(out), B, H, W, C, ivlm_stream(s));
}
int ivlm_im2col3x3_nhwc_strided(const void* x, int64_t ldx, void* out, int64_t ldo, int B, int H, int W, int C, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::im2col3x3_nhwc(CBF(x), BF(out), B, H, W, C, ivlm_stream(s), ldx, ldo);
}
int ivlm_gather_rows(void* dst, int dst_kind, int64_t ldd, const void* src, int src_dtype, int64_t lds_, const int32_t* idx,
                     const void* add, int add_dtype, int64_t lda, int64_t rows, int cols, const float* fp8_scale,
                     ivlm_stream_t s) {
    ivlm_enter();
    const int k = dst_kind == IVLM_F32 ? 1 : (dst_kind == IVLM_BF16 ? 0 : (dst_kind == IVLM_BF16_SPLIT ? 2 : (dst_kind == IVLM_FP8 ? 3 : -1)));
    return ivlm::gather_rows(dst, k, ldd, src, src_dtype == IVLM_F32, lds_, idx, add, add_dtype == IVLM_F32, lda, rows, cols,
                             ivlm_stream(s), fp8_scale);
}
int ivlm_amax(const void* x, int dtype, int64_t n, float* out, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::amax(x, dtype == IVLM_F32, n, out, ivlm_stream(s));
}
int ivlm_add_rows(void* out, int out_kind, const void* a, int a_dtype, const void* b, int b_dtype, int64_t rows, int cols,
                  int64_t b_rows, int op, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::add_rows(out, out_kind == IVLM_F32 ? 1 : (out_kind == IVLM_BF16 ? 0 : (out_kind == IVLM_BF16_SPLIT ? 2 : -1)),
                          a, a_dtype == IVLM_F32, b, b_dtype == IVLM_F32, rows, cols, b_rows, ivlm_stream(s), op);
}
int ivlm_fill_rows(void* dst, int64_t ldd, const int32_t* idx, int64_t n_idx, const void* row, int cols, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::fill_rows(BF(dst), ldd, idx, n_idx, CBF(row), cols, ivlm_stream(s));
}
int ivlm_dense_pe(const void* gauss, void* pe, int pe_dtype, int h, int w, int F, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::dense_pe(static_cast<const float*>(gauss), pe, pe_dtype == IVLM_F32, h, w, F, ivlm_stream(s));
}
int ivlm_rope_kv(void* qkv, int64_t ld, int T, int H, int D, int pos0, float theta, void* kcache, void* vcache,
                 const float* cos_tab, const float* sin_tab, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::rope_kv(BF(qkv), ld, T, H, D, pos0, theta, BF(kcache), BF(vcache), ivlm_stream(s), cos_tab, sin_tab);
}
int ivlm_rope_kv_f16(void* qkv, int64_t ld, int T, int H, int D, int pos0, float theta, void* kcache, void* vcache,
                     const float* cos_tab, const float* sin_tab, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::rope_kv(BF(qkv), ld, T, H, D, pos0, theta, BF(kcache), BF(vcache), ivlm_stream(s), cos_tab, sin_tab, 1);
}
int ivlm_rope_kv_split(void* qkv, int64_t ld, int T, int H, int D, int pos0, void* kcache, void* kcache_lo, void* vcache,
                       void* vcache_lo, const float* cos_tab, const float* sin_tab, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::rope_kv_split(BF(qkv), ld, T, H, D, pos0, BF(kcache), BF(kcache_lo), BF(vcache), BF(vcache_lo), cos_tab, sin_tab,
                               ivlm_stream(s));
}
int ivlm_bf16_to_f16(const void* src, void* dst, int64_t n, ivlm_stream_t s) {
    ivlm_enter();
    if (n == 0) return IVLM_OK;
    if (!src || !dst || n < 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return IVLM_ERR_INVALID_ARG;
    ivlm::bf16_to_f16_kernel<<<ivlm::grid_for((n >> 3) + 1), ivlm::kT, 0, ivlm_stream(s)>>>(CBF(src), static_cast<uint16_t*>(dst), n);
    return ivlm_launch_status();
}
int ivlm_rope_table(float* cos_tab, float* sin_tab, int T, int D, float theta, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::rope_table(cos_tab, sin_tab, T, D, theta, ivlm_stream(s));
}
int ivlm_mask_dot(const void* up, const void* hyper, int dtype, float* low, int B, int gh, int gw, int C, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::mask_dot(up, hyper, dtype == IVLM_F32, low, B, gh, gw, C, ivlm_stream(s));
}
int ivlm_normalize_pad_u8(const uint8_t* src, int H, int W, int y0, int x0, int ch, int cw, const float* mean3_host,
                          const float* std3_host, void* out, int out_bf16, int OH, int OW, ivlm_stream_t s) {
    ivlm_enter();
    return ivlm::normalize_pad_u8(src, H, W, y0, x0, ch, cw, mean3_host, std3_host, out, out_bf16, OH, OW,
                                  ivlm_stream(s));
}
#undef BF
#undef CBF
}  // extern "C"
