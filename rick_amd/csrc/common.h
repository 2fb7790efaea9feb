// Shared helpers for the gfx950 kernels of librick_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rick_hip.h"

#define RICK_LAUNCH_STATUS()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        return e__ == hipSuccess ? 0 : 1000 + (int)e__;        \
    } while (0)

// Kernels that need more than the default 64 KB of dynamic LDS: raise the function's limit to the CU's 160 KB ONCE per kernel
// (template instantiation) and device — the call is a driver round trip (~1 us), harmless under graph replay but paid per launch
// in eager issue.  The static flags belong to the expansion site, i.e. to one kernel variant.
#define RICK_LDS160_ONCE(kernel_ptr)                                                                         \
    do {                                                                                                     \
        static bool done__[16];                                                                              \
        int dev__ = 0;                                                                                       \
        if (hipGetDevice(&dev__) != hipSuccess || dev__ < 0 || dev__ >= 16 || !done__[dev__]) {              \
            (void)hipFuncSetAttribute((const void *)(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (dev__ >= 0 && dev__ < 16) done__[dev__] = true;                                              \
        }                                                                                                    \
    } while (0)

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__host__ __device__ __forceinline__ int floor_div_i(int a, int b) {
    int c = a / b;
    if (c * b > a) c--;
    return c;
}

// 64-lane wavefront sum via DPP-free shuffles (wave64 on gfx950).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float *red /* >= 4 floats of LDS */) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// 16-byte load through a pointer that is KNOWN to address global memory.  Pointers read from descriptor tables are generic
// to the compiler (FLAT loads: 64-bit per-lane address pairs, LDS / scratch aperture checks); this form yields
// `global_load_dwordx4 v, v_off, s[base]`.
typedef float rick_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_global4(const float *p) {
    const rick_f32x4 v = *(const rick_f32x4 __attribute__((address_space(1))) *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
