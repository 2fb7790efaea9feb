// EqualLinear WITH gradients on a short batch (B <= 16): the discriminator's two final layers
// (model_probe_tune.py:139-173, 699-702: [B, 8192] -> 512 -> 1) in every D pass of the training loop, forward, data
// gradient and weight gradient, as three products closed under differentiation (R1 differentiates them twice):
//
//     P1(x, W)  [B, O] = alpha * x W^T (+ bias_mul * bias)          rick_linear_fwd_f32
//     P2(g, W)  [B, K] = alpha * g W                                rick_linear_dgrad_f32
//     P3(g, x)  [O, K] = alpha * g^T x   (+ column sums of g)       rick_linear_wgrad_f32
//
// Every one streams the [O, K] matrix once (16 MB for the 8192 -> 512 layer: HBM-bound, the batch rides along in
// registers / LDS) and sums in a FIXED order — no atomics, no library-chosen split: bit-reproducible run to run.  The
// rocBLAS / hipBLASLt calls these replace (torch.addmm) may pick split-K solutions that accumulate with atomics for
// such skinny shapes (M = batch, K = 8192): a run-to-run difference in the last bit of one logit is enough to flip an
// `array_equal` between two data-parallel runs (tests/test_gpu_dp.py, round 4's red test).
#include "common.h"

#define LN_MAXB 16
#define LN_COLS 8            // output columns per block of the forward product (2 per wave)
#define LN_KSLICE 1024       // K columns per block of the forward product
#define LN_OROWS 8           // weight rows per block of the weight gradient

// ---- P1: block (o tile, k slice): x slice [B][<=1024] in LDS, a wave owns 2 columns, a lane float4s along k
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *__restrict__ out,
                                                         float *__restrict__ part, int B, int K, int O, float alpha,
                                                         float bias_mul) {
    extern __shared__ float sx[];                                   // [B][LN_KSLICE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = (int)blockIdx.y * LN_KSLICE;
    const int len = K - k0 < LN_KSLICE ? K - k0 : LN_KSLICE;       // % 4 == 0
    for (int j = threadIdx.x * 4; j < B * len; j += 1024) {
        const int b = j / len, k = j - b * len;
        *reinterpret_cast<float4 *>(sx + b * LN_KSLICE + k) = *reinterpret_cast<const float4 *>(x + (int64_t)b * K + k0 + k);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int o = (int)blockIdx.x * LN_COLS + wave * 2 + r;
        if (o >= O) break;                                          // wave-uniform
        const float *wr = W + (int64_t)o * K + k0;
        float acc[LN_MAXB];
#pragma unroll
        for (int b = 0; b < LN_MAXB; b++) acc[b] = 0.f;
        for (int k = lane * 4; k < len; k += 256) {
            const float4 wv = *reinterpret_cast<const float4 *>(wr + k);
#pragma unroll
            for (int b = 0; b < LN_MAXB; b++)
                if (b < B) {
                    const float4 xv = *reinterpret_cast<const float4 *>(sx + b * LN_KSLICE + k);
                    acc[b] = __builtin_fmaf(wv.x, xv.x, acc[b]);
                    acc[b] = __builtin_fmaf(wv.y, xv.y, acc[b]);
                    acc[b] = __builtin_fmaf(wv.z, xv.z, acc[b]);
                    acc[b] = __builtin_fmaf(wv.w, xv.w, acc[b]);
                }
        }
#pragma unroll
        for (int b = 0; b < LN_MAXB; b++)
            if (b < B) {
                const float v = wave_sum(acc[b]);
                if (lane == 0) {
                    if (part) part[((int64_t)blockIdx.y * B + b) * O + o] = v;
                    else out[(int64_t)b * O + o] = v * alpha + (bias ? bias[o] * bias_mul : 0.f);
                }
            }
    }
}

// second stage of a split product: out[i] = alpha * (part[0][i] + part[1][i] + ...) (+ bias), slices in index order
__global__ __launch_bounds__(256) void linear_finish_kernel(const float *__restrict__ part, int S, int64_t n, int cols,
                                                            const float *__restrict__ bias, float bias_mul, float alpha,
                                                            float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = part[i];
    for (int s = 1; s < S; s++) v += part[(int64_t)s * n + i];
    out[i] = v * alpha + (bias ? bias[i % cols] * bias_mul : 0.f);
}

extern "C" int64_t rick_linear_fwd_workspace_floats(int B, int K, int O) {
    if (B < 1 || K < 4 || O < 1) return -1;
    const int S = cdiv(K, LN_KSLICE);
    return S > 1 ? (int64_t)S * B * O : 0;
}

extern "C" int rick_linear_fwd_f32(const float *x, const float *W, const float *bias, float *out, int B, int K, int O,
                                   float alpha, float bias_mul, float *workspace, void *stream) {
    if (!x || !W || !out || B < 1 || B > LN_MAXB || K < 4 || (K & 3) || O < 1) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)W) & 15) return RICK_EINVAL;
    const int S = cdiv(K, LN_KSLICE);
    if (S > 1 && !workspace) return RICK_EINVAL;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)cdiv(O, LN_COLS), (unsigned)S), dim3(256), (size_t)B * LN_KSLICE * 4, (hipStream_t)stream, x, W,
                       bias, out, S > 1 ? workspace : nullptr, B, K, O, alpha, bias_mul);
    if (S > 1) {
        const int64_t n = (int64_t)B * O;
        hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, workspace, S, n, O,
                           bias, bias_mul, alpha, out);
    }
    RICK_LAUNCH_STATUS();
}

// ---- P2: block (k tile of 1024, slice of R weight rows): g slice [B][R] in LDS, a thread owns 4 consecutive k
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float *__restrict__ g, const float *__restrict__ W,
                                                           float *__restrict__ out, float *__restrict__ part, int B, int K,
                                                           int O, int R, float alpha) {
    extern __shared__ float sg[];                                   // [R][B]
    const int o0 = (int)blockIdx.y * R;
    const int rows = O - o0 < R ? O - o0 : R;
    for (int j = threadIdx.x; j < rows * B; j += 256) {
        const int r = j / B, b = j - r * B;
        sg[j] = g[(int64_t)b * O + o0 + r];
    }
    __syncthreads();
    const int k = ((int)blockIdx.x * 256 + threadIdx.x) * 4;
    if (k >= K) return;
    float4 acc[LN_MAXB];
#pragma unroll
    for (int b = 0; b < LN_MAXB; b++) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *wp = W + (int64_t)o0 * K + k;
    int r = 0;
    for (; r + 4 <= rows; r += 4) {                                 // four independent 16-byte loads in flight
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) wv[u] = *reinterpret_cast<const float4 *>(wp + (int64_t)(r + u) * K);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int b = 0; b < LN_MAXB; b++)
                if (b < B) {
                    const float gv = sg[(r + u) * B + b];
                    acc[b].x = __builtin_fmaf(gv, wv[u].x, acc[b].x);
                    acc[b].y = __builtin_fmaf(gv, wv[u].y, acc[b].y);
                    acc[b].z = __builtin_fmaf(gv, wv[u].z, acc[b].z);
                    acc[b].w = __builtin_fmaf(gv, wv[u].w, acc[b].w);
                }
    }
    for (; r < rows; r++) {
        const float4 wv = *reinterpret_cast<const float4 *>(wp + (int64_t)r * K);
#pragma unroll
        for (int b = 0; b < LN_MAXB; b++)
            if (b < B) {
                const float gv = sg[r * B + b];
                acc[b].x = __builtin_fmaf(gv, wv.x, acc[b].x);
                acc[b].y = __builtin_fmaf(gv, wv.y, acc[b].y);
                acc[b].z = __builtin_fmaf(gv, wv.z, acc[b].z);
                acc[b].w = __builtin_fmaf(gv, wv.w, acc[b].w);
            }
    }
#pragma unroll
    for (int b = 0; b < LN_MAXB; b++)
        if (b < B) {
            if (part)
                *reinterpret_cast<float4 *>(part + ((int64_t)blockIdx.y * B + b) * K + k) = acc[b];
            else
                *reinterpret_cast<float4 *>(out + (int64_t)b * K + k) =
                    make_float4(acc[b].x * alpha, acc[b].y * alpha, acc[b].z * alpha, acc[b].w * alpha);
        }
}

// rows of W per block: enough blocks to cover the chip (>= ~256), at least 32 rows each — every row slice writes and the second
// stage re-reads B x K partial sums, so 8-row slices moved 2 x 16 MB of partials next to the 16 MB weight stream of the
// 8192 -> 512 layer at B = 8; 32 rows: 2 x 4 MB
static int ln_dgrad_rows(int K, int O) {
    const int kb = cdiv(K, 1024);
    int slices = 256 / kb;
    if (slices < 1) slices = 1;
    int R = cdiv(O, slices);
    if (R < 32) R = 32;
    if (R > 256) R = 256;
    return R;
}

extern "C" int64_t rick_linear_dgrad_workspace_floats(int B, int K, int O) {
    if (B < 1 || K < 4 || O < 1) return -1;
    const int S = cdiv(O, ln_dgrad_rows(K, O));
    return S > 1 ? (int64_t)S * B * K : 0;
}

extern "C" int rick_linear_dgrad_f32(const float *g, const float *W, float *out, int B, int K, int O, float alpha,
                                     float *workspace, void *stream) {
    if (!g || !W || !out || B < 1 || B > LN_MAXB || K < 4 || (K & 3) || O < 1) return RICK_EINVAL;
    if (((uintptr_t)W | (uintptr_t)out | (uintptr_t)workspace) & 15) return RICK_EINVAL;
    const int R = ln_dgrad_rows(K, O), S = cdiv(O, R);
    if (S > 1 && !workspace) return RICK_EINVAL;
    hipLaunchKernelGGL(linear_dgrad_kernel, dim3((unsigned)cdiv(K, 1024), (unsigned)S), dim3(256), (size_t)R * B * 4, (hipStream_t)stream,
                       g, W, out, S > 1 ? workspace : nullptr, B, K, O, R, alpha);
    if (S > 1) {
        const int64_t n = (int64_t)B * K;
        hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, workspace, S, n, K,
                           (const float *)nullptr, 0.f, alpha, out);
    }
    RICK_LAUNCH_STATUS();
}

// ---- P3: block (k tile of 1024, 8 weight rows): a thread keeps x[b, k .. k+3] of every b in registers
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                           float *__restrict__ gw, float *__restrict__ gb, int B, int K, int O,
                                                           float alpha, float bias_mul, int accumulate) {
    __shared__ float sg[LN_OROWS * LN_MAXB];                        // [8][B]
    const int o0 = (int)blockIdx.y * LN_OROWS;
    const int rows = O - o0 < LN_OROWS ? O - o0 : LN_OROWS;
    if ((int)threadIdx.x < rows * B) {
        const int r = threadIdx.x / B, b = threadIdx.x - r * B;
        sg[threadIdx.x] = g[(int64_t)b * O + o0 + r];
    }
    __syncthreads();
    if (gb && blockIdx.x == 0 && (int)threadIdx.x < rows) {         // bias gradient: bias_mul * sum_b g[b, o]
        float s = 0.f;
        for (int b = 0; b < B; b++) s += sg[threadIdx.x * B + b];
        s = __fmul_rn(s, bias_mul);
        gb[o0 + threadIdx.x] = accumulate ? gb[o0 + threadIdx.x] + s : s;
    }
    const int k = ((int)blockIdx.x * 256 + threadIdx.x) * 4;
    if (!gw || k >= K) return;
    float4 xv[LN_MAXB];
#pragma unroll
    for (int b = 0; b < LN_MAXB; b++)
        if (b < B) xv[b] = *reinterpret_cast<const float4 *>(x + (int64_t)b * K + k);
    for (int r = 0; r < rows; r++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < LN_MAXB; b++)
            if (b < B) {
                const float gv = sg[r * B + b];
                v.x = __builtin_fmaf(gv, xv[b].x, v.x);
                v.y = __builtin_fmaf(gv, xv[b].y, v.y);
                v.z = __builtin_fmaf(gv, xv[b].z, v.z);
                v.w = __builtin_fmaf(gv, xv[b].w, v.w);
            }
        float4 *dst = reinterpret_cast<float4 *>(gw + (int64_t)(o0 + r) * K + k);
        // (the product is rounded on its own in both modes: accumulating equals .grad + the plain result bit for bit)
        const float4 va = make_float4(__fmul_rn(v.x, alpha), __fmul_rn(v.y, alpha), __fmul_rn(v.z, alpha), __fmul_rn(v.w, alpha));
        if (accumulate) {
            const float4 old = *dst;
            *dst = make_float4(old.x + va.x, old.y + va.y, old.z + va.z, old.w + va.w);
        } else {
            *dst = va;
        }
    }
}

extern "C" int rick_linear_wgrad_f32(const float *g, const float *x, float *gw, float *gb, int B, int K, int O, float alpha,
                                     float bias_mul, int accumulate, void *stream) {
    if (!g || !x || (!gw && !gb) || B < 1 || B > LN_MAXB || K < 4 || (K & 3) || O < 1) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gw) & 15) return RICK_EINVAL;
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)(gw ? cdiv(K, 1024) : 1), (unsigned)cdiv(O, LN_OROWS)), dim3(256), 0,
                       (hipStream_t)stream, g, x, gw, gb, B, K, O, alpha, bias_mul, accumulate);
    RICK_LAUNCH_STATUS();
}
