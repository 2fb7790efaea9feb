// VALU issue cost of the fp32 -> 16-bit hi/lo split variants (one wave per SIMD, everything in registers):
//   mix   : 8 x v_fma_mixlo/mixhi_f16 per 4 elements (scale included)            [conv_common.h split4s]
//   cvt   : 2 v_pk_mul + 2 v_cvt_pk_f16_f32 + 4 v_cvt_f32_f16 + 2 v_pk_add + 2 v_cvt_pk_f16_f32
//   bf16  : round 2's bf16 split (perm / and / pk_add / cvt_pk_bf16) + 2 v_pk_mul for the scale
// build: hipcc -O3 --offload-arch=gfx950 -I../../rick_amd/csrc -I../../include split_rate.hip -o split_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "conv_common.h"

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {
    bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return *reinterpret_cast<unsigned *>(&r);
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *in, uint4 *out, float s, int iters) {
    float4 v = in[threadIdx.x];
    uint2 ah = make_uint2(0, 0), al = ah;
    const float su = cv_uniform(s);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            uint2 hi, lo;
            if (MODE == 0) split4s<2>(v, su, hi, lo);
            else if (MODE == 1) {
                const float4 w = scale4(v, su);
                const f16x2 h0 = __builtin_convertvector((f32x2_t){w.x, w.y}, f16x2), h1 = __builtin_convertvector((f32x2_t){w.z, w.w}, f16x2);
                hi.x = *reinterpret_cast<const unsigned *>(&h0); hi.y = *reinterpret_cast<const unsigned *>(&h1);
                lo.x = pack_f16_rne(w.x - (float)h0[0], w.y - (float)h0[1]);
                lo.y = pack_f16_rne(w.z - (float)h1[0], w.w - (float)h1[1]);
            } else {
                const float4 w = scale4(v, su);
                const unsigned ux = __float_as_uint(w.x), uy = __float_as_uint(w.y), uz = __float_as_uint(w.z), uw = __float_as_uint(w.w);
                hi.x = __builtin_amdgcn_perm(uy, ux, 0x07060302);
                hi.y = __builtin_amdgcn_perm(uw, uz, 0x07060302);
                lo.x = pack_bf16_rne(w.x - __uint_as_float(ux & 0xffff0000u), w.y - __uint_as_float(uy & 0xffff0000u));
                lo.y = pack_bf16_rne(w.z - __uint_as_float(uz & 0xffff0000u), w.w - __uint_as_float(uw & 0xffff0000u));
            }
            ah.x ^= hi.x; ah.y ^= hi.y; al.x ^= lo.x; al.y ^= lo.y;
            v.x += 1.0f; v.y += 0.5f; v.z -= 0.25f; v.w += 2.0f;      // 4 more VALU per round: same in every mode
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = make_uint4(ah.x, ah.y, al.x, al.y);
}
int main() {
    float4 *in; uint4 *out;
    hipMalloc(&in, 256 * 16); hipMalloc(&out, 1024 * 256 * 16);
    hipMemset(in, 0, 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, in, out, 4.0f, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, in, out, 4.0f, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, 0, in, out, 4.0f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        // 1024 blocks x 4 waves on 256 CUs x 4 SIMDs = 4 waves per SIMD (issue-bound); splits per wave = iters * 16
        const double ns_per_split = best * 1e6 / (iters * 16.0) / 4.0;   // per wave-split per SIMD
        printf("%s: %.3f ms, %.2f ns per 4-element split per SIMD (~%.1f cycles at 2.4 GHz)\n", mode == 0 ? "mix " : mode == 1 ? "cvt " : "bf16", best, ns_per_split, ns_per_split * 2.4);
    }
    return 0;
}
