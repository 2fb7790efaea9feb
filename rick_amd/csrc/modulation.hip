// Style demodulation coefficients of the modulated convolution (model_probe_tune.py:246-252) and their first-order
// gradient, as four small kernels instead of ~25 tensor-algebra launches per layer and pass:
//     wsq[o,i] = scale^2 * sum_k w[o,i,k]^2                      (depends on the weights only)
//     d[b,o]   = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + eps)
//     t[b,o]   = -0.5 * d^3 * gd ;  gs[b,i] = 2 s[b,i] sum_o t[b,o] wsq[o,i] ;
//     gw[o,i,k] = 2 scale^2 w[o,i,k] * sum_b t[b,o] s[b,i]^2
// Everything here is latency-bound ([B,512] / [512,512] operands); the point is the launch count.
#include "common.h"

#define MOD_MAXB 32
#define MB_MAXB 8         // modulation / demodulation banks: batch <= 8

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
    RICK_LDS160_ONCE(demod_bwd_s_kernel);
    hipLaunchKernelGGL(demod_bwd_s_kernel, dim3(cdiv(I, 64)), dim3(64 * DBS_WAVES), lds, (hipStream_t)stream, s, wsq, d, gd, gs,
                       B, I, O);
    RICK_LAUNCH_STATUS();
}

// gw[o,i,k] = 2 scale^2 w[o,i,k] * sum_b t[b,o] s[b,i]^2
__global__ __launch_bounds__(256) void demod_bwd_w_kernel(const float *__restrict__ w, const float *__restrict__ s,
                                                          const float *__restrict__ d, const float *__restrict__ gd,
                                                          float *__restrict__ gw, int B, int I, int O, int K, float scale2,
                                                          int accumulate) {
    const int64_t OI = (int64_t)O * I;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < OI; j += (int64_t)gridDim.x * 256) {
        const int o = (int)(j / I), i = (int)(j - (int64_t)o * I);
        float acc = 0.f;
        for (int b = 0; b < B; b++) {
            const float dv = d[(int64_t)b * O + o], sv = s[(int64_t)b * I + i];
            acc = __builtin_fmaf(-0.5f * dv * dv * dv * gd[(int64_t)b * O + o], sv * sv, acc);
        }
        const float f = 2.f * scale2 * acc;
        if (accumulate)
            for (int k = 0; k < K; k++) gw[j * K + k] += w[j * K + k] * f;
        else
            for (int k = 0; k < K; k++) gw[j * K + k] = w[j * K + k] * f;
    }
}

extern "C" int rick_demod_bwd_w_f32(const float *w, const float *s, const float *d, const float *gd, float *gw, int B, int I,
                                    int O, int K, float scale, int accumulate, void *stream) {
    if (!w || !s || !d || !gd || !gw || B <= 0 || I <= 0 || O <= 0 || K <= 0) return RICK_EINVAL;
    int64_t nb = cdiv64((int64_t)O * I, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(demod_bwd_w_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, s, d, gd, gw, B, I, O, K,
                       scale * scale, accumulate);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------
// Demodulation bank: the three kernels above for EVERY demodulated convolution of a generator in one launch each (13 layers
// at 256 px: 13 wsq + 13 demod launches of 5-6 us per forward, 13 weight-gradient launches of ~10 us per backward — 0.43 ms of
// a train iteration).  Same arithmetic per element, in the same order (bit-identical to the per-layer kernels).  wsq depends
// on the weights only: the host recomputes it once per update of the network (rick_wsq_multi_f32, with the weight packs) and
// every forward in between reads it.  Layer l: s_l [B, I] at float offset s_off of the modulation bank's flat output,
// d_l / gd_l [B, O] at d_off of the flat coefficient / gradient buffer; blocks [blk_*, next layer's blk_*).
#define WSQ_KMAX 9
__global__ __launch_bounds__(256) void wsq_multi_kernel(const rick_demod_desc *__restrict__ descs, int n) {
    int l = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_wsq) l = i;
    const rick_demod_desc ds = descs[l];
    const int64_t OI = (int64_t)ds.O * ds.I;
    const int64_t i0 = (int64_t)((int)blockIdx.x - ds.blk_wsq) * 256;
    const int K = ds.K;
    if (K <= WSQ_KMAX) {      // the block's 256 * K consecutive floats through LDS (coalesced), then the same k-ordered sum per (o, i)
        __shared__ float sw[256 * WSQ_KMAX];
        const int npair = OI - i0 < 256 ? (int)(OI - i0) : 256;
        const float *w = ds.w + i0 * K;
        for (int e = threadIdx.x; e < npair * K; e += 256) sw[e] = w[e];
        __syncthreads();
        if ((int)threadIdx.x < npair) {
            const float *p = sw + threadIdx.x * K;
            float s = 0.f;
            for (int k = 0; k < K; k++) s = __builtin_fmaf(p[k], p[k], s);
            ds.wsq[i0 + threadIdx.x] = s * ds.scale2;
        }
        return;
    }
    const int64_t i = i0 + threadIdx.x;
    if (i >= OI) return;
    const float *p = ds.w + i * K;
    float s = 0.f;
    for (int k = 0; k < K; k++) s = __builtin_fmaf(p[k], p[k], s);
    ds.wsq[i] = s * ds.scale2;
}

__global__ __launch_bounds__(256) void demod_multi_kernel(const float *__restrict__ s_flat, float *__restrict__ d_flat,
                                                          const rick_demod_desc *__restrict__ descs, int n, int B, float eps) {
    extern __shared__ float s2[];   // [B][I]
    int l = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_demod) l = i;
    const rick_demod_desc ds = descs[l];
    const int I = ds.I, O = ds.O;
    const float *s = s_flat + ds.s_off;
    for (int j = threadIdx.x; j < B * I; j += 256) {
        const float v = s[j];
        s2[j] = v * v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int o = ((int)blockIdx.x - ds.blk_demod) * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    float acc[MB_MAXB];
#pragma unroll
    for (int b = 0; b < MB_MAXB; b++) acc[b] = 0.f;
    const float *wr = ds.wsq + (int64_t)o * I;
    int i = lane;
    for (; i + 192 < I; i += 256) {
        const float w0 = wr[i], w1 = wr[i + 64], w2 = wr[i + 128], w3 = wr[i + 192];
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++)
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
        for (int b = 0; b < MB_MAXB; b++)
            if (b < B) acc[b] = __builtin_fmaf(s2[b * I + i], wv, acc[b]);
    }
#pragma unroll
    for (int b = 0; b < MB_MAXB; b++)
        if (b < B) {
            const float tot = wave_sum(acc[b]);
            if (lane == 0) d_flat[ds.d_off + (int64_t)b * O + o] = rsqrtf(tot + eps);
        }
}

__global__ __launch_bounds__(256) void demod_bwd_w_multi_kernel(const float *__restrict__ s_flat, const float *__restrict__ d_flat,
                                                                const float *__restrict__ gd_flat,
                                                                const rick_demod_desc *__restrict__ descs, int n, int B) {
    int l = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_wsq) l = i;
    const rick_demod_desc ds = descs[l];
    if (!ds.gw) return;                                   // frozen layer
    const int I = ds.I, O = ds.O, K = ds.K;
    __shared__ float fl[256];
    const int64_t j0 = (int64_t)((int)blockIdx.x - ds.blk_wsq) * 256, OI = (int64_t)O * I;
    const int64_t j = j0 + threadIdx.x;
    const float *s = s_flat + ds.s_off, *d = d_flat + ds.d_off, *gd = gd_flat + ds.d_off;
    float f = 0.f;
    if (j < OI) {
        const int o = (int)(j / I), i = (int)(j - (int64_t)o * I);
        float acc = 0.f;
        for (int b = 0; b < B; b++) {
            const float dv = d[(int64_t)b * O + o], sv = s[(int64_t)b * I + i];
            acc = __builtin_fmaf(-0.5f * dv * dv * dv * gd[(int64_t)b * O + o], sv * sv, acc);
        }
        f = 2.f * ds.scale2 * acc;
    }
    fl[threadIdx.x] = f;
    __syncthreads();
    // the block's 256 (o, i) pairs are 256 * K consecutive floats of w / gw: walk them lane by lane (coalesced), the factor of a
    // pair from LDS — the per-pair loop over k touched 4 bytes of every 36 per lane (117 us for a 256-px generator; same values:
    // gw += w * f element by element)
    const int npair = OI - j0 < 256 ? (int)(OI - j0) : 256;
    const float *w = ds.w + j0 * K;
    float *gw = ds.gw + j0 * K;
    for (int e = threadIdx.x; e < npair * K; e += 256) gw[e] += w[e] * fl[e / K];      // (always accumulates: gw is the parameter's .grad)
}

__global__ __launch_bounds__(64 * DBS_WAVES) void demod_bwd_s_multi_kernel(const float *__restrict__ s_flat,
                                                                          const float *__restrict__ d_flat,
                                                                          const float *__restrict__ gd_flat, float *__restrict__ gs_flat,
                                                                          const rick_demod_desc *__restrict__ descs, int n, int B) {
    extern __shared__ float tl[];   // [B][O] t values, then [DBS_WAVES][B][64] wave partials
    int l = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_bwd_s) l = i;
    const rick_demod_desc ds = descs[l];
    const int I = ds.I, O = ds.O;
    const float *s = s_flat + ds.s_off, *d = d_flat + ds.d_off, *gd = gd_flat + ds.d_off, *wsq = ds.wsq;
    float *gs = gs_flat + ds.s_off;
    float *part = tl + B * O;
    for (int j = threadIdx.x; j < B * O; j += 64 * DBS_WAVES) {
        const float dv = d[j];
        tl[j] = -0.5f * dv * dv * dv * gd[j];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x - ds.blk_bwd_s;
    const int i = blk * 64 + lane;
    const int o_per = (O + DBS_WAVES - 1) / DBS_WAVES, o0 = wave * o_per, o1 = o0 + o_per < O ? o0 + o_per : O;
    float acc[MB_MAXB];
#pragma unroll
    for (int b = 0; b < MB_MAXB; b++) acc[b] = 0.f;
    if (i < I) {
        const float *wc = wsq + i;
        int o = o0;
        for (; o + 3 < o1; o += 4) {
            const float w0 = wc[(int64_t)o * I], w1 = wc[(int64_t)(o + 1) * I], w2 = wc[(int64_t)(o + 2) * I],
                        w3 = wc[(int64_t)(o + 3) * I];
#pragma unroll
            for (int b = 0; b < MB_MAXB; b++)
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
            for (int b = 0; b < MB_MAXB; b++)
                if (b < B) acc[b] = __builtin_fmaf(tl[b * O + o], wv, acc[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < MB_MAXB; b++)
        if (b < B) part[(wave * B + b) * 64 + lane] = acc[b];
    __syncthreads();
    for (int j = threadIdx.x; j < B * 64; j += 64 * DBS_WAVES) {
        const int b = j >> 6, ll = j & 63, ii = blk * 64 + ll;
        if (ii < I) {
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < DBS_WAVES; wv++) tot += part[(wv * B + b) * 64 + ll];
            gs[(int64_t)b * I + ii] = 2.f * s[(int64_t)b * I + ii] * tot;
        }
    }
}

extern "C" int rick_demod_blocks_bwd_s(int I) { return cdiv(I, 64); }

extern "C" int rick_demod_bwd_s_multi_f32(const float *s_flat, const float *d_flat, const float *gd_flat, float *gs_flat,
                                          const rick_demod_desc *descs_device, int n, int total_blocks, int B, int max_O,
                                          void *stream) {
    if (!s_flat || !d_flat || !gd_flat || !gs_flat || !descs_device || n < 1 || total_blocks < 1 || B < 1 || B > MB_MAXB || max_O < 1)
        return RICK_EINVAL;
    const size_t lds = ((size_t)B * max_O + DBS_WAVES * (size_t)B * 64) * sizeof(float);
    if (lds > 160 * 1024) return RICK_EINVAL;
    RICK_LDS160_ONCE(demod_bwd_s_multi_kernel);
    hipLaunchKernelGGL(demod_bwd_s_multi_kernel, dim3((unsigned)total_blocks), dim3(64 * DBS_WAVES), lds, (hipStream_t)stream, s_flat,
                       d_flat, gd_flat, gs_flat, descs_device, n, B);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_demod_blocks_wsq(int O, int I) { return (int)cdiv64((int64_t)O * I, 256); }
extern "C" int rick_demod_blocks(int O) { return cdiv(O, 4); }

extern "C" int rick_wsq_multi_f32(const rick_demod_desc *descs_device, int n, int total_blocks, void *stream) {
    if (!descs_device || n < 1 || total_blocks < 1) return RICK_EINVAL;
    hipLaunchKernelGGL(wsq_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs_device, n);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_demod_multi_f32(const float *s_flat, float *d_flat, const rick_demod_desc *descs_device, int n,
                                    int total_blocks, int B, int max_I, float eps, void *stream) {
    if (!s_flat || !d_flat || !descs_device || n < 1 || total_blocks < 1 || B < 1 || B > MB_MAXB || max_I < 1) return RICK_EINVAL;
    const size_t lds = (size_t)B * max_I * sizeof(float);
    if (lds > 64 * 1024) return RICK_EINVAL;
    hipLaunchKernelGGL(demod_multi_kernel, dim3((unsigned)total_blocks), dim3(256), lds, (hipStream_t)stream, s_flat, d_flat,
                       descs_device, n, B, eps);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_demod_bwd_w_multi_f32(const float *s_flat, const float *d_flat, const float *gd_flat,
                                          const rick_demod_desc *descs_device, int n, int total_blocks, int B, void *stream) {
    if (!s_flat || !d_flat || !gd_flat || !descs_device || n < 1 || total_blocks < 1 || B < 1 || B > MB_MAXB) return RICK_EINVAL;
    hipLaunchKernelGGL(demod_bwd_w_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, s_flat, d_flat,
                       gd_flat, descs_device, n, B);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------
// Modulation bank: the style -> per-channel scale linears of EVERY modulated convolution of the generator
// (ModulatedConv2d.modulation = EqualLinear(style_dim, in_channel, bias_init=1), model_probe_tune.py:233,246; 13
// StyledConvs + 7 ToRGBs at 256 px) in ONE launch, and their weight / bias gradients in one more — instead of one
// rocBLAS GEMM per layer forward and two GEMMs + a column sum + scalings per layer backward (~180 launches of 4-8 us
// per generator forward + backward).
//   s_l[b, c]  = scale * sum_k lat[b, idx_l, k] * W_l[c, k] + bias_l[c]
//   gW_l[c, k] = scale * sum_b gs_l[b, c] * lat[b, idx_l, k] ;   gb_l[c] = sum_b gs_l[b, c]
// Layer l owns blocks [blk_begin, blk_begin + ceil(C_l / 32)); s_l / gs_l are [B, C_l] contiguous at float offset
// io_off of the flat output / gradient-input buffer, gW_l / gb_l at gw_off / gb_off of the flat gradient buffer.
// K % 256 == 0, B <= 8.
#define MB_ROWS 32       // channels per block
// acc = fma(a, b, acc) as ONE scalar-lane instruction the vectoriser cannot pair up (see modbank_fwd_kernel)
#define MB_FMAC(acc, a, b) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))

__global__ __launch_bounds__(256) void modbank_fwd_kernel(const float *__restrict__ lat, int B, int n_latent, int K,
                                                          const rick_modbank_desc *__restrict__ descs, int n, float scale,
                                                          float *__restrict__ out) {
    extern __shared__ float sl[];   // [B][K] latent rows of this layer
    int d = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_begin) d = i;
    const rick_modbank_desc ds = descs[d];
    for (int j = threadIdx.x; j < B * K; j += 256) {
        const int b = j / K, k = j - b * K;
        sl[j] = lat[((int64_t)b * n_latent + ds.lat_idx) * K + k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = ((int)blockIdx.x - ds.blk_begin) * MB_ROWS + wave * (MB_ROWS / 4);
    // the wave's 8 weight rows are fetched together, one k slice (256 floats) at a time: 8 independent 16-byte loads in flight per
    // lane instead of a dependent load -> reduce chain per row (the launch is latency-bound: 40 -> ~10 us for the 20 layers of a
    // 256-px generator).  Per (row, b) the products are accumulated in the same k order as before: same values.
    //
    // The multiply-adds are written as explicit v_fmac_f32 (MB_FMAC).  Left to the compiler this loop becomes 128 v_pk_fma_f32
    // (two rows per instruction, the latent element broadcast by op_sel) — and THAT form returned 1-3 wrong outputs (off by one
    // lane's partial sum) in 3-8 % of its launches whenever another process kept the same GPU busy, never when it ran alone
    // (round 5: tools/stress_ops.py, tools/stress_bank.py; it is what turned tests/test_gpu_dp.py red on the driver's box in
    // round 4: two ranks share cuda:0 there).  Same loop, same loads, scalar FMAs: 0 differences in 30 000 launches under the
    // same load; so was the round-3 one-row-at-a-time form.  No other kernel of the library showed the effect (18 op families
    // and full G / D passes, 2 000 launches each).  Cause not established (the ISA's waits and hazards read correctly).
    constexpr int RW = MB_ROWS / 4;
    float acc[RW][MB_MAXB];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++) acc[r][b] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        float4 wv[RW];
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const int c = c0 + r < ds.C ? c0 + r : ds.C - 1;      // (rows past the layer re-read its last row; never stored)
            wv[r] = *reinterpret_cast<const float4 *>(ds.w + (int64_t)c * K + k);
        }
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++)
            if (b < B) {
                const float4 lv = *reinterpret_cast<const float4 *>(sl + b * K + k);
#pragma unroll
                for (int r = 0; r < RW; r++) {
                    MB_FMAC(acc[r][b], wv[r].x, lv.x);
                    MB_FMAC(acc[r][b], wv[r].y, lv.y);
                    MB_FMAC(acc[r][b], wv[r].z, lv.z);
                    MB_FMAC(acc[r][b], wv[r].w, lv.w);
                }
            }
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int c = c0 + r;
        if (c >= ds.C) break;                       // wave-uniform
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++)
            if (b < B) {
                const float v = wave_sum(acc[r][b]);
                if (lane == 0) out[ds.io_off + (int64_t)b * ds.C + c] = v * scale + (ds.b ? ds.b[c] : 0.f);
            }
    }
}

__global__ __launch_bounds__(256) void modbank_bwd_kernel(const float *__restrict__ lat, const float *__restrict__ gs, int B,
                                                          int n_latent, int K, const rick_modbank_desc *__restrict__ descs,
                                                          int n, float scale, float *__restrict__ grad, int accumulate) {
    __shared__ float sg[MB_MAXB][MB_ROWS];
    int d = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_begin) d = i;
    const rick_modbank_desc ds = descs[d];
    if (ds.gw_off < 0 && ds.gb_off < 0) return;          // frozen layer: gs of this layer is not even read
    const int c0 = ((int)blockIdx.x - ds.blk_begin) * MB_ROWS;
    if (threadIdx.x < MB_ROWS * MB_MAXB) {
        const int b = threadIdx.x / MB_ROWS, r = threadIdx.x % MB_ROWS;
        sg[b][r] = (b < B && c0 + r < ds.C) ? gs[ds.io_off + (int64_t)b * ds.C + c0 + r] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < MB_ROWS && c0 + (int)threadIdx.x < ds.C && ds.gb_off >= 0) {
        float s = 0.f;
        for (int b = 0; b < B; b++) s += sg[b][threadIdx.x];
        float *dst = grad + ds.gb_off + c0 + threadIdx.x;
        *dst = accumulate ? *dst + s : s;
    }
    if (ds.gw_off < 0) return;
    // thread = (4 consecutive k, row r): gW[c0 + r, k..k+3]
    const int kq = K / 4;
    for (int it = threadIdx.x; it < MB_ROWS * kq; it += 256) {
        const int r = it / kq, k = (it - r * kq) * 4;
        if (c0 + r >= ds.C) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < B; b++) {
            const float g = sg[b][r];
            const float4 lv = *reinterpret_cast<const float4 *>(lat + ((int64_t)b * n_latent + ds.lat_idx) * K + k);
            a.x = __builtin_fmaf(g, lv.x, a.x);
            a.y = __builtin_fmaf(g, lv.y, a.y);
            a.z = __builtin_fmaf(g, lv.z, a.z);
            a.w = __builtin_fmaf(g, lv.w, a.w);
        }
        float4 *dst = reinterpret_cast<float4 *>(grad + ds.gw_off + (int64_t)(c0 + r) * K + k);
        float4 o = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
        if (accumulate) {
            const float4 p = *dst;
            o = make_float4(p.x + o.x, p.y + o.y, p.z + o.z, p.w + o.w);
        }
        *dst = o;
    }
}

extern "C" int rick_modbank_blocks(int C) { return cdiv(C, MB_ROWS); }

extern "C" int rick_modbank_fwd_f32(const float *lat, int B, int n_latent, int K, const rick_modbank_desc *descs_device, int n,
                                    int total_blocks, float scale, float *out, void *stream) {
    if (!lat || !descs_device || !out || B < 1 || B > MB_MAXB || K < 256 || (K & 255) || n < 1 || total_blocks < 1) return RICK_EINVAL;
    hipLaunchKernelGGL(modbank_fwd_kernel, dim3((unsigned)total_blocks), dim3(256), (size_t)B * K * 4, (hipStream_t)stream, lat, B,
                       n_latent, K, descs_device, n, scale, out);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_modbank_bwd_f32(const float *lat, const float *gs, int B, int n_latent, int K,
                                    const rick_modbank_desc *descs_device, int n, int total_blocks, float scale, float *grad,
                                    int accumulate, void *stream) {
    if (!lat || !gs || !descs_device || !grad || ((uintptr_t)grad & 15) || B < 1 || B > MB_MAXB || K < 256 || (K & 255) || n < 1 || total_blocks < 1)
        return RICK_EINVAL;
    hipLaunchKernelGGL(modbank_bwd_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, lat, gs, B, n_latent, K,
                       descs_device, n, scale, grad, accumulate);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------
// EqualLinear on a short batch: the 8-layer mapping network (model_probe_tune.py:139-173, 418-428) maps [B <= 16, 512]
// latents — per layer a GEMM with 8 rows, a bias scaling and the fused activation, i.e. three launches of 4-8 us for
// 4 MFLOP.  One launch per layer here, PixelNorm (model_probe_tune.py:92-98) folded into the first:
//   x'[b,:] = pixelnorm ? x[b,:] * rsqrt(mean_k x[b,k]^2 + 1e-8) : x[b,:]
//   out[b,o] = act(scale * sum_k x'[b,k] W[o,k] + bias[o] * bias_mul),  act = gain * leaky_relu(., slope) or identity
// A wave owns EL_CPW output columns: W rows are read once (float4 per lane), x' sits in LDS, sums finish with shuffles.
#define EL_MAXB 16
#define EL_CPW 2

__global__ __launch_bounds__(256) void equal_linear_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                           const float *__restrict__ bias, float *__restrict__ out, int B,
                                                           int K, int O, float scale, float bias_mul, int act, float slope,
                                                           float gain, int pixelnorm) {
    extern __shared__ float sx[];   // [B][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x * 4; j < B * K; j += 1024)
        *reinterpret_cast<float4 *>(sx + j) = *reinterpret_cast<const float4 *>(x + j);
    __syncthreads();
    if (pixelnorm) {
        for (int b = wave; b < B; b += 4) {
            float q = 0.f;
            for (int k = lane; k < K; k += 64) q = __builtin_fmaf(sx[b * K + k], sx[b * K + k], q);
            q = wave_sum(q);
            const float rn = rsqrtf(q / (float)K + 1e-8f);
            for (int k = lane; k < K; k += 64) sx[b * K + k] *= rn;
        }
        __syncthreads();
    }
    const int o0 = ((int)blockIdx.x * 4 + wave) * EL_CPW;
#pragma unroll
    for (int r = 0; r < EL_CPW; r++) {
        const int o = o0 + r;
        if (o >= O) break;                          // wave-uniform
        const float *wr = W + (int64_t)o * K;
        float acc[EL_MAXB];
#pragma unroll
        for (int b = 0; b < EL_MAXB; b++) acc[b] = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const float4 wv = *reinterpret_cast<const float4 *>(wr + k);
#pragma unroll
            for (int b = 0; b < EL_MAXB; b++)
                if (b < B) {
                    const float4 xv = *reinterpret_cast<const float4 *>(sx + b * K + k);
                    acc[b] = __builtin_fmaf(wv.x, xv.x, acc[b]);
                    acc[b] = __builtin_fmaf(wv.y, xv.y, acc[b]);
                    acc[b] = __builtin_fmaf(wv.z, xv.z, acc[b]);
                    acc[b] = __builtin_fmaf(wv.w, xv.w, acc[b]);
                }
        }
        const float bo = bias ? bias[o] * bias_mul : 0.f;
#pragma unroll
        for (int b = 0; b < EL_MAXB; b++)
            if (b < B) {
                float v = wave_sum(acc[b]) * scale + bo;
                if (act) v = (v > 0.f ? v : v * slope) * gain;
                if (lane == 0) out[(int64_t)b * O + o] = v;
            }
    }
}

extern "C" int rick_equal_linear_f32(const float *x, const float *W, const float *bias, float *out, int B, int K, int O,
                                     float scale, float bias_mul, int act, float slope, float gain, int pixelnorm,
                                     void *stream) {
    if (!x || !W || !out || B < 1 || B > EL_MAXB || K < 4 || (K & 3) || O < 1) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)W) & 15) return RICK_EINVAL;
    hipLaunchKernelGGL(equal_linear_kernel, dim3((unsigned)cdiv(O, 4 * EL_CPW)), dim3(256), (size_t)B * K * 4, (hipStream_t)stream,
                       x, W, bias, out, B, K, O, scale, bias_mul, act, slope, gain, pixelnorm);
    RICK_LAUNCH_STATUS();
}
