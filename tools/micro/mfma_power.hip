// Sustained rate of the fp16 MFMA shapes under the board's power envelope, operands resident in registers:
//   shape 0: v_mfma_f32_16x16x32_f16, wave tile 64 x 128 (4 x 8 accumulator tiles) — the conv kernels' form
//   shape 1: v_mfma_f32_32x32x16_f16, wave tile 64 x 128 (2 x 4 accumulator tiles), two k-halves per 32-deep chunk
// in the 3-product split form (hi*hi + hi*lo + lo*hi) and over operand data of different activity:
//   zero | +-1 | randn hi with 2^-11-sized random lo parts (what the split produces)
// 2 waves per SIMD (256 threads, 2 blocks per CU), 1024 blocks.  Prints issued TFLOP/s.
// build: hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// operands: [set][frag][lane] f16x8; sets rotate every iteration so the inputs of consecutive MFMAs differ
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const f16x8 *__restrict__ ops, float *__restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 ah[4], al[4], bh[8], bl[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ah[i] = ops[(0 * 8 + i) * 64 + lane];
        al[i] = ops[(1 * 8 + i) * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        bh[j] = ops[(2 * 8 + j) * 64 + lane];
        bl[j] = ops[(3 * 8 + j) * 64 + lane];
    }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        // 64 x 128 x 32 per iteration: row tiles 2, col tiles 4, k-halves 2; fragment (tile, k-half) = one f16x8
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int p = 0; p < 3; p++)          // product outer: 8 independent accumulators between dependent MFMAs
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const f16x8 a = p == 0 ? al[i * 2 + kh] : ah[i * 2 + kh];
                            const f16x8 b = p == 1 ? bl[j * 2 + kh] : bh[j * 2 + kh];
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
                        }
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) s += acc[i][j][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float randn() {
    float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main() {
    const int nfrag = 4 * 8 * 64 * 8;     // halves
    _Float16 *h = (_Float16 *)malloc(nfrag * 2);
    f16x8 *ops;
    float *out;
    hipMalloc(&ops, nfrag * 2);
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[3] = {"zero", "+-1", "randn hi / 2^-11 lo"};
    const int iters = 4000;
    for (int data = 0; data < 3; data++) {
        for (int i = 0; i < nfrag; i++) {
            const int set = i / (8 * 64 * 8);              // 0: ah, 1: al, 2: bh, 3: bl
            const bool lo = set & 1;
            float v = 0.f;
            if (data == 1) v = lo ? 0.f : ((rand() & 1) ? 1.f : -1.f);
            if (data == 2) v = lo ? randn() * 4.f * 0.00028f : randn() * 4.f * 0.35f;   // exponent target [4, 8): typical values ~1.4
            h[i] = (_Float16)v;
        }
        hipMemcpy(ops, h, nfrag * 2, hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; shape++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, ops, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, ops, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;       // (first launch: cold clocks)
            }
            const double flops = 1024.0 * 4 * iters * 96.0 * 2.0 * 16 * 16 * 32;
            printf("%-22s %s: %8.2f ms  %7.1f TFLOP/s issued (%6.1f algorithmic at 3 products)\n", names[data],
                   shape == 0 ? "16x16x32" : "32x32x16", best, flops / best * 1e-9, flops / best * 1e-9 / 3);
        }
    }
    return 0;
}
