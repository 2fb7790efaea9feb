// What does TRAPSTS.EXCP record for an fp32 -> fp16 conversion that overflows (with and without MODE.FP16_OVFL)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, unsigned *out, int clamp, int excp_en) {
    if (clamp) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    if (excp_en) __builtin_amdgcn_s_setreg((8 << 11) | (12 << 6) | 1, 0);   // MODE.EXCP_EN bits [20:12] left at 0 anyway
    const unsigned before = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 3);
    const float a = in[threadIdx.x], b = in[threadIdx.x + 64];
    f16x2 r = __builtin_convertvector((f32x2){a, b}, f16x2);
    const unsigned bits = *reinterpret_cast<unsigned *>(&r);
    __builtin_amdgcn_s_sleep(2);
    const unsigned after = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 3);
    const float big = a * 1e30f * 1e30f;        // fp32 overflow for comparison
    const unsigned after2 = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 3);
    if (threadIdx.x == 0) {
        out[0] = before; out[1] = after; out[2] = bits; out[3] = after2; out[4] = __float_as_uint(big);
        out[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 1);   // MODE
    }
}
int main() {
    float h[128];
    for (int i = 0; i < 128; i++) h[i] = 1.0f;
    h[5] = 1e5f;
    float *d; unsigned *o, ho[6];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 24);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int clamp = 0; clamp < 2; clamp++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, clamp, 0);
        hipMemcpy(ho, o, 24, hipMemcpyDeviceToHost);
        printf("clamp=%d trapsts before=%08x after cvt=%08x (lane0 bits %08x) after f32 ovf=%08x big=%08x mode=%08x\n", clamp, ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
    }
    h[5] = 1.0f;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1, 0);
    hipMemcpy(ho, o, 24, hipMemcpyDeviceToHost);
    printf("no spike: trapsts before=%08x after cvt=%08x after f32 mul=%08x\n", ho[0], ho[1], ho[3]);
    return 0;
}
