// Style demodulation coefficients of the modulated convolution (model_probe_tune.py:246-252) and their first-order
// gradient, as four small kernels instead of ~25 tensor-algebra launches per layer and pass:
//     wsq[o,i] = scale^2 * sum_k w[o,i,k]^2                      (depends on the weights only)
//     d[b,o]   = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + eps)
//     t[b,o]   = -0.5 * d^3 * gd ;  gs[b,i] = 2 s[b,i] sum_o t[b,o] wsq[o,i] ;
//     gw[o,i,k] = 2 scale^2 w[o,i,k] * sum_b t[b,o] s[b,i]^2
// Everything here is latency-bound ([B,512] / [512,512] operands); the point is the launch count.
#include "common.h"

#define MOD_MAXB 32

__global__ __launch_bounds__(256) void wsq_kernel(const float *__restrict__ w, float *__restrict__ wsq, int64_t OI, int K,
                                                  float scale2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < OI; i += (int64_t)gridDim.x * 256) {
        const float *p = w + i * K;
        float s = 0.f;
        for (int k = 0; k < K; k++) s = __builtin_fmaf(p[k], p[k], s);
        wsq[i] = s * scale2;
    }
}

extern "C" int rick_wsq_f32(const float *w, float *wsq, int O, int I, int K, float scale, void *stream) {
    if (!w || !wsq || O <= 0 || I <= 0 || K <= 0) return RICK_EINVAL;
    const int64_t OI = (int64_t)O * I;
    int64_t nb = cdiv64(OI, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(wsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, wsq, OI, K, scale * scale);
    RICK_LAUNCH_STATUS();
}

// one wavefront per output channel o: the wsq row is read once (4 loads in flight per lane) and dotted with
// s[b,:]^2 for every b; s^2 is staged in LDS once per block.
__global__ __launch_bounds__(256) void demod_kernel(const float *__restrict__ s, const float *__restrict__ wsq,
                                                    float *__restrict__ d, int B, int I, int O, float eps) {
    extern __shared__ float s2[];   // [B][I]
    for (int j = threadIdx.x; j < B * I; j += 256) {
        const float v = s[j];
        s2[j] = v * v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    float acc[MOD_MAXB];
#pragma unroll
    for (int b = 0; b < MOD_MAXB; b++) acc[b] = 0.f;
    const float *wr = wsq + (int64_t)o * I;
    int i = lane;
    for (; i + 192 < I; i += 256) {
        const float w0 = wr[i], w1 = wr[i + 64], w2 = wr[i + 128], w3 = wr[i + 192];
#pragma unroll
        for (int b = 0; b < MOD_MAXB; b++)
            if (b < B) {
                const float *sb = s2 + b * I + i;
                acc[b] = __builtin_fmaf(sb[0], w0, acc[b]);
                acc[b] = __builtin_fmaf(sb[64], w1, acc[b]);
                acc[b] = __builtin_fmaf(sb[128], w2, acc[b]);
                acc[b] = __builtin_fmaf(sb[192], w3, acc[b]);
            }
    }
    for (; i < I; i += 64) {
        const float wv = wr[i];
#pragma unroll
        for (int b = 0; b < MOD_MAXB; b++)
            if (b < B) acc[b] = __builtin_fmaf(s2[b * I + i], wv, acc[b]);
    }
#pragma unroll
    for (int b = 0; b < MOD_MAXB; b++)
        if (b < B) {
            const float tot = wave_sum(acc[b]);
            if (lane == 0) d[(int64_t)b * O + o] = rsqrtf(tot + eps);
        }
}

extern "C" int rick_demod_f32(const float *s, const float *wsq, float *d, int B, int I, int O, float eps, void *stream) {
    if (!s || !wsq || !d || B <= 0 || B > MOD_MAXB || I <= 0 || O <= 0) return RICK_EINVAL;
    const size_t lds = (size_t)B * I * sizeof(float);
    if (lds > 64 * 1024) return RICK_EINVAL;
    hipLaunchKernelGGL(demod_kernel, dim3(cdiv(O, 4)), dim3(256), lds, (hipStream_t)stream, s, wsq, d, B, I, O, eps);
    RICK_LAUNCH_STATUS();
}

// gs[b,i] = 2 s[b,i] * sum_o t[b,o] wsq[o,i],  t = -0.5 d^3 gd.  Block = 64 consecutive input channels (one
// coalesced 256-byte piece of every wsq row) x 16 waves that split the o range (4 rows in flight per lane);
// fixed-order LDS combine.
#define DBS_WAVES 16
__global__ __launch_bounds__(64 * DBS_WAVES) void demod_bwd_s_kernel(const float *__restrict__ s, const float *__restrict__ wsq,
                                                                    const float *__restrict__ d, const float *__restrict__ gd,
                                                                    float *__restrict__ gs, int B, int I, int O) {
    extern __shared__ float tl[];   // [B][O] t values, then [DBS_WAVES][B][64] wave partials
    float *part = tl + B * O;
    for (int j = threadIdx.x; j < B * O; j += 64 * DBS_WAVES) {
        const float dv = d[j];
        tl[j] = -0.5f * dv * dv * dv * gd[j];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int o_per = (O + DBS_WAVES - 1) / DBS_WAVES, o0 = wave * o_per, o1 = o0 + o_per < O ? o0 + o_per : O;
    float acc[MOD_MAXB];
#pragma unroll
    for (int b = 0; b < MOD_MAXB; b++) acc[b] = 0.f;
    if (i < I) {
        const float *wc = wsq + i;
        int o = o0;
        for (; o + 3 < o1; o += 4) {
            const float w0 = wc[(int64_t)o * I], w1 = wc[(int64_t)(o + 1) * I], w2 = wc[(int64_t)(o + 2) * I],
                        w3 = wc[(int64_t)(o + 3) * I];
#pragma unroll
            for (int b = 0; b < MOD_MAXB; b++)
                if (b < B) {
                    const float *tb = tl + b * O + o;
                    acc[b] = __builtin_fmaf(tb[0], w0, acc[b]);
                    acc[b] = __builtin_fmaf(tb[1], w1, acc[b]);
                    acc[b] = __builtin_fmaf(tb[2], w2, acc[b]);
                    acc[b] = __builtin_fmaf(tb[3], w3, acc[b]);
                }
        }
        for (; o < o1; o++) {
            const float wv = wc[(int64_t)o * I];
#pragma unroll
            for (int b = 0; b < MOD_MAXB; b++)
                if (b < B) acc[b] = __builtin_fmaf(tl[b * O + o], wv, acc[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < MOD_MAXB; b++)
        if (b < B) part[(wave * B + b) * 64 + lane] = acc[b];
    __syncthreads();
    for (int j = threadIdx.x; j < B * 64; j += 64 * DBS_WAVES) {
        const int b = j >> 6, l = j & 63, ii = blockIdx.x * 64 + l;
        if (ii < I) {
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < DBS_WAVES; wv++) tot += part[(wv * B + b) * 64 + l];
            gs[(int64_t)b * I + ii] = 2.f * s[(int64_t)b * I + ii] * tot;
        }
    }
}

extern "C" int rick_demod_bwd_s_f32(const float *s, const float *wsq, const float *d, const float *gd, float *gs, int B,
                                    int I, int O, void *stream) {
    if (!s || !wsq || !d || !gd || !gs || B <= 0 || B > MOD_MAXB || I <= 0 || O <= 0) return RICK_EINVAL;
    const size_t lds = ((size_t)B * O + DBS_WAVES * (size_t)B * 64) * sizeof(float);
    if (lds > 160 * 1024) return RICK_EINVAL;
    (void)hipFuncSetAttribute((const void *)demod_bwd_s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(demod_bwd_s_kernel, dim3(cdiv(I, 64)), dim3(64 * DBS_WAVES), lds, (hipStream_t)stream, s, wsq, d, gd, gs,
                       B, I, O);
    RICK_LAUNCH_STATUS();
}

// gw[o,i,k] = 2 scale^2 w[o,i,k] * sum_b t[b,o] s[b,i]^2
__global__ __launch_bounds__(256) void demod_bwd_w_kernel(const float *__restrict__ w, const float *__restrict__ s,
                                                          const float *__restrict__ d, const float *__restrict__ gd,
                                                          float *__restrict__ gw, int B, int I, int O, int K, float scale2) {
    const int64_t OI = (int64_t)O * I;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < OI; j += (int64_t)gridDim.x * 256) {
        const int o = (int)(j / I), i = (int)(j - (int64_t)o * I);
        float acc = 0.f;
        for (int b = 0; b < B; b++) {
            const float dv = d[(int64_t)b * O + o], sv = s[(int64_t)b * I + i];
            acc = __builtin_fmaf(-0.5f * dv * dv * dv * gd[(int64_t)b * O + o], sv * sv, acc);
        }
        const float f = 2.f * scale2 * acc;
        for (int k = 0; k < K; k++) gw[j * K + k] = w[j * K + k] * f;
    }
}

extern "C" int rick_demod_bwd_w_f32(const float *w, const float *s, const float *d, const float *gd, float *gw, int B, int I,
                                    int O, int K, float scale, void *stream) {
    if (!w || !s || !d || !gd || !gw || B <= 0 || I <= 0 || O <= 0 || K <= 0) return RICK_EINVAL;
    int64_t nb = cdiv64((int64_t)O * I, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(demod_bwd_w_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, s, d, gd, gw, B, I, O, K,
                       scale * scale);
    RICK_LAUNCH_STATUS();
}
