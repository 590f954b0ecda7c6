#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ float wave_total(float v) {
    v = dpp_add<0xb1, 0xf>(v);
    v = dpp_add<0x4e, 0xf>(v);
    v = dpp_add<0x124, 0xf>(v);
    v = dpp_add<0x128, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__global__ void k(const float* in, float* out, const uint32_t* a, const uint32_t* b, float* dout) {
    float v = in[threadIdx.x];
    out[threadIdx.x] = wave_total(v);
    float acc = 0.f;
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[threadIdx.x]), __builtin_bit_cast(bf16x2_t, b[threadIdx.x]), acc, false);
    dout[threadIdx.x] = acc;
}
int main() {
    float h[64], *d, *o, ho[64]; double ref = 0;
    for (int i = 0; i < 64; ++i) { h[i] = (float)(i * i % 17) - 3.25f; ref += h[i]; }
    uint32_t ha[64], hb[64], *da, *db; float *dd, hd[64];
    for (int i = 0; i < 64; ++i) { ha[i] = 0x3f804000u + (i << 16); hb[i] = 0x40003f80u; }  // lo/hi bf16 pairs
    hipMalloc(&d, 256); hipMalloc(&o, 256); hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice); hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, da, db, dd);
    hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, 256, hipMemcpyDeviceToHost);
    printf("ref %f got %f %f\n", ref, ho[0], ho[63]);
    auto bf = [](uint32_t x) { uint32_t u = x << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int i = 0; i < 3; ++i) printf("dot2 lane %d: got %f ref %f\n", i, hd[i], bf(ha[i] & 0xffff) * bf(hb[i] & 0xffff) + bf(ha[i] >> 16) * bf(hb[i] >> 16));
    return 0;
}
