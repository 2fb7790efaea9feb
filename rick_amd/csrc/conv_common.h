// Device helpers shared by the MFMA convolution kernels (conv.hip, convt2.hip): bf16 hi/lo split, the swizzled
// [row][32 k] LDS image, vector loads, XCD-aware block remap.
#pragma once
#include "common.h"
#include <type_traits>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CV_BM 128
#define CV_BN 128
#define CV_CK 32
#define CV_WTILE_BYTES (CV_BM * CV_CK * 2)      // one of hi / lo: 8 KB
#define CV_WSTEP_BYTES (2 * CV_WTILE_BYTES)     // hi + lo: 16 KB

__host__ __device__ __forceinline__ int cv_swz(int kg, int row) { return kg ^ (((row >> 2) & 1) << 1); }

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {   // v_cvt_pk_bf16_f32
    bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return *reinterpret_cast<unsigned *>(&r);
}

// x = hi + lo with hi = x truncated to bf16 (exactly representable, so lo = x - hi is exact in
// fp32) and lo rounded to nearest bf16: |x - hi - lo| <= 2^-17 |x|, unbiased.  10 VALU ops per 4
// elements (v_and, v_perm, v_pk_add, v_cvt_pk).  SPLIT == 1 (plain bf16): hi is rounded to nearest.
template <int SPLIT>
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo) {
    if (SPLIT == 1) {
        hi.x = pack_bf16_rne(v.x, v.y);
        hi.y = pack_bf16_rne(v.z, v.w);
        lo.x = lo.y = 0;
        return;
    }
    const unsigned ux = __float_as_uint(v.x), uy = __float_as_uint(v.y), uz = __float_as_uint(v.z),
                   uw = __float_as_uint(v.w);
    hi.x = __builtin_amdgcn_perm(uy, ux, 0x07060302);
    hi.y = __builtin_amdgcn_perm(uw, uz, 0x07060302);
    lo.x = pack_bf16_rne(v.x - __uint_as_float(ux & 0xffff0000u), v.y - __uint_as_float(uy & 0xffff0000u));
    lo.y = pack_bf16_rne(v.z - __uint_as_float(uz & 0xffff0000u), v.w - __uint_as_float(uw & 0xffff0000u));
}

// XCD-aware bijective remap of the linear block id: blocks that share an XCD (id % 8) get a
// contiguous range of logical tiles, so neighbouring position tiles of one co-tile share L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// 16-byte load of 4 consecutive channels.  VEC (channel count % 4 == 0): one float4 load from an address
// that is always safe (callers substitute the tensor base for out-of-range items) — no branch, so the
// compiler has no reason to wait vmcnt(0) per element.  !VEC: guarded scalar loads (odd channel counts).
template <bool VEC>
__device__ __forceinline__ float4 load4(const float *p, bool ok, int c, int C) {
    if (VEC) return *reinterpret_cast<const float4 *>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        v.x = p[0];
        if (c + 1 < C) v.y = p[1];
        if (c + 2 < C) v.z = p[2];
        if (c + 3 < C) v.w = p[3];
    }
    return v;
}

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {   // f(integral_constant<int, I>) ... f(<N-1>): indices usable as constants
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

