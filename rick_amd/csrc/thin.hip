// Thin (J <= 4) channel products for the RGB side of the networks — HBM-bound, so no MFMA:
// each NHWC activation element is read exactly once, dot products are finished with
// wavefront shuffles, reductions over pixels are two-stage and deterministic.
//   ToRGB (model_probe_tune.py:351-370):  rgb[n,j,p] = sum_c x[n,p,c] * (scale*w[j,c]*s[n,c]) + ...
//   D input conv (model_probe_tune.py:679): y[n,p,co] = sum_j img[n,j,p] * (scale*w[co,j])
#include "conv_common.h"

#define THIN_MAXJ 4

// ToRGB form (model_probe_tune.py:246-248, 366-370): the per-sample weight is formed on the fly from the shared [J, C]
// weight and the style, W[n,j,c] = (wscale * w[j,c]) * s[n,c] (the reference's association), and bias[j] is added to the
// output before `add` (the upsampled skip image) — the [N, J, C] weight tensor, its two multiplies and the two
// adds per layer never run as launches of their own.  smod == NULL: plain per-sample / shared W as given.
struct ThinMod {
    const float *smod;   // [N, C] or NULL
    float wscale;
    const float *bias;   // [J] or NULL
};

__device__ __forceinline__ float4 thin_w(const float *__restrict__ Wrow, const ThinMod &m, int n, int C, int c) {
    float4 w = *reinterpret_cast<const float4 *>(Wrow + c);
    if (m.smod) {
        const float4 sv = *reinterpret_cast<const float4 *>(m.smod + (int64_t)n * C + c);
        w = make_float4((m.wscale * w.x) * sv.x, (m.wscale * w.y) * sv.y, (m.wscale * w.z) * sv.z, (m.wscale * w.w) * sv.w);
    }
    return w;
}

// t[n,j,p] = sum_c x[n,p,c] * W[n,j,c] (+ add[n,j,p]);  LPP lanes cooperate on one pixel.
// A lane owns the same NQ channel quads for every pixel it visits, so its W values live in registers
// (NQ x J float4) and the loop body is one 16-byte load of x per quad; UNR pixels are in flight per lane.
template <int NQ>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                       int64_t w_bstride, const float *__restrict__ add,
                                                       float *__restrict__ t, int64_t P, int C, int J, int lpp, ThinMod m) {
    constexpr int UNR = 4;
    const int n = blockIdx.y;
    const int pix_per_block = 256 / lpp;
    const int sub = threadIdx.x % lpp, pl = threadIdx.x / lpp;
    const float *Wn = W + (int64_t)n * w_bstride;
    float4 wr[NQ][THIN_MAXJ];
#pragma unroll
    for (int qd = 0; qd < NQ; qd++)
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++)
            // rows j >= J re-read row J-1 (their sums are never stored): no branch, so the 4 x NQ loads batch
            wr[qd][j] = thin_w(Wn + (int64_t)(j < J ? j : J - 1) * C, m, n, C, (sub + qd * lpp) * 4);
    const float *xn = x + (int64_t)n * P * C + sub * 4;
    const int64_t stride = (int64_t)gridDim.x * pix_per_block;
    for (int64_t p0 = (int64_t)blockIdx.x * pix_per_block + pl; p0 < P; p0 += stride * UNR) {
        float4 xv[UNR][NQ];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t p = p0 + u * stride;
#pragma unroll
            for (int qd = 0; qd < NQ; qd++)
                xv[u][qd] = *reinterpret_cast<const float4 *>(xn + (p < P ? p : 0) * C + qd * lpp * 4);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t p = p0 + u * stride;
            float acc[THIN_MAXJ] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qd = 0; qd < NQ; qd++)
#pragma unroll
                for (int j = 0; j < THIN_MAXJ; j++)
                    acc[j] += xv[u][qd].x * wr[qd][j].x + xv[u][qd].y * wr[qd][j].y + xv[u][qd].z * wr[qd][j].z +
                              xv[u][qd].w * wr[qd][j].w;
#pragma unroll
            for (int j = 0; j < THIN_MAXJ; j++)
                for (int off = lpp >> 1; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off, 64);
            if (p < P && sub < J) {
                // lane `sub` writes output channel j = sub (select without dynamic register indexing)
                float v = acc[0];
                if (sub == 1) v = acc[1];
                if (sub == 2) v = acc[2];
                if (sub == 3) v = acc[3];
                const int64_t o = ((int64_t)n * J + sub) * P + p;
                if (m.bias) v += m.bias[sub];
                if (add) v += add[o];
                t[o] = v;
            }
        }
    }
}

// Matrix-core form (C = 16 * NT, P % 16 == 0): the product is a [16 pixels x C] x [C x 16] GEMM per wave step with 3 of the 16
// output columns in use — wasteful in FLOPs, free in time: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) needs
// 32 of them per 16 pixels x 128 channels = 25 % of the matrix pipe at HBM rate, and there is no cross-lane reduction and no
// per-pixel VALU left (the lanes-per-pixel form spends 3-5 shuffle steps x 4 accumulators per pixel: 2.9 TB/s at 128 channels,
// 1.4 at 512).  A lane (pixel i = lane % 16, k = lane / 16) loads 16 bytes = channels 16 tb + 4 k .. + 3 of its pixel — the 4 k-lanes
// of a pixel read 64 contiguous bytes — and component j feeds MFMA (tb, j), whose four K indices are the channels 16 tb + 4 k + j;
// the B operand holds the matching weights W[o][16 tb + 4 k + j] (o = lane % 16, zero for o >= J), modulated once per block.
// The result tile D[pixel][o] leaves as float4 stores of 4 consecutive pixels from the lanes with o < J.
typedef float thin_f32x4 __attribute__((ext_vector_type(4)));
template <int NT, int UNR>
__global__ __launch_bounds__(256) void thin_fwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                            int64_t w_bstride, const float *__restrict__ add,
                                                            float *__restrict__ t, int64_t P, int C, int J, ThinMod m) {
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, k = lane >> 4;
    const float *Wn = W + (int64_t)n * w_bstride;
    float4 breg[NT];
#pragma unroll
    for (int tb = 0; tb < NT; tb++) {
        const float4 wv = thin_w(Wn + (int64_t)(i < J ? i : 0) * C, m, n, C, 16 * tb + 4 * k);
        breg[tb] = i < J ? wv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *xn = x + (int64_t)n * P * C + 4 * k;
    const int64_t ngroups = P >> 4;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    for (int64_t g0 = (int64_t)blockIdx.x * 4 + wave; g0 < ngroups; g0 += gstride * UNR) {
        float4 xv[UNR][NT];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t g = g0 + u * gstride;
            const float *xp = xn + ((g < ngroups ? g : g0) * 16 + i) * C;
#pragma unroll
            for (int tb = 0; tb < NT; tb++) xv[u][tb] = *reinterpret_cast<const float4 *>(xp + 16 * tb);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t g = g0 + u * gstride;
            thin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tb = 0; tb < NT; tb++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][tb].x, breg[tb].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][tb].y, breg[tb].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][tb].z, breg[tb].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][tb].w, breg[tb].w, acc, 0, 0, 0);
            }
            if (g < ngroups && i < J) {      // D[pixel 4 k + r][o = i]
                const int64_t o = ((int64_t)n * J + i) * P + g * 16 + 4 * k;
                float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if (m.bias) {
                    const float bv = m.bias[i];
                    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                }
                if (add) {
                    const float4 av = *reinterpret_cast<const float4 *>(add + o);
                    v.x += av.x; v.y += av.y; v.z += av.z; v.w += av.w;
                }
                *reinterpret_cast<float4 *>(t + o) = v;
            }
        }
    }
}

// Generic form (any C % 4 == 0): lanes stride over the channel quads, W re-read through L1.
__global__ __launch_bounds__(256) void thin_fwd_generic_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                               int64_t w_bstride, const float *__restrict__ add,
                                                               float *__restrict__ t, int64_t P, int C, int J, int lpp,
                                                               ThinMod m) {
    const int n = blockIdx.y;
    const int pix_per_block = 256 / lpp;
    const int sub = threadIdx.x % lpp, pl = threadIdx.x / lpp;
    const float *Wn = W + (int64_t)n * w_bstride;
    const int C4 = C >> 2;
    for (int64_t p0 = (int64_t)blockIdx.x * pix_per_block; p0 < P; p0 += (int64_t)gridDim.x * pix_per_block) {
        const int64_t p = p0 + pl;
        float acc[THIN_MAXJ] = {0.f, 0.f, 0.f, 0.f};
        if (p < P) {
            const float4 *xp = reinterpret_cast<const float4 *>(x + ((int64_t)n * P + p) * C);
            for (int c4 = sub; c4 < C4; c4 += lpp) {
                const float4 xv = xp[c4];
#pragma unroll
                for (int j = 0; j < THIN_MAXJ; j++)
                    if (j < J) {
                        const float4 wv = thin_w(Wn + (int64_t)j * C, m, n, C, c4 * 4);
                        acc[j] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++)
            for (int off = lpp >> 1; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off, 64);
        if (p < P && sub < J) {
            float v = acc[0];
            if (sub == 1) v = acc[1];
            if (sub == 2) v = acc[2];
            if (sub == 3) v = acc[3];
            const int64_t o = ((int64_t)n * J + sub) * P + p;
            if (m.bias) v += m.bias[sub];
            if (add) v += add[o];
            t[o] = v;
        }
    }
}

static int thin_fwd_launch(const float *x, const float *W, int64_t w_bstride, const float *add, float *t, int N, int64_t P, int C,
                           int J, ThinMod m, void *stream) {
    if (!x || !W || !t || N <= 0 || P <= 0 || C <= 0 || C % 4 || J < 1 || J > THIN_MAXJ || N > 65535) return RICK_EINVAL;
    const int C4 = C / 4;
    hipStream_t st = (hipStream_t)stream;
    // matrix-core form: whole 16-pixel groups, 64 ... 512 channels in blocks of 16, 16-byte aligned operands
    if ((P & 15) == 0 && (C == 64 || C == 128 || C == 256 || C == 512) &&
        (((uintptr_t)x | (uintptr_t)t | (uintptr_t)(add ? add : t) | (uintptr_t)W | (uintptr_t)(m.smod ? m.smod : W)) & 15) == 0 && (w_bstride & 3) == 0) {
        int64_t nbm = cdiv64(P >> 4, 4 * (C <= 128 ? 2 : 1));
        if (nbm > 8192) nbm = 8192;
        const dim3 gridm((unsigned)nbm, N), blkm(256);
        if (C == 64) hipLaunchKernelGGL((thin_fwd_mfma_kernel<4, 2>), gridm, blkm, 0, st, x, W, w_bstride, add, t, P, C, J, m);
        else if (C == 128) hipLaunchKernelGGL((thin_fwd_mfma_kernel<8, 2>), gridm, blkm, 0, st, x, W, w_bstride, add, t, P, C, J, m);
        else if (C == 256) hipLaunchKernelGGL((thin_fwd_mfma_kernel<16, 1>), gridm, blkm, 0, st, x, W, w_bstride, add, t, P, C, J, m);
        else hipLaunchKernelGGL((thin_fwd_mfma_kernel<32, 1>), gridm, blkm, 0, st, x, W, w_bstride, add, t, P, C, J, m);
        RICK_LAUNCH_STATUS();
    }
    // fast path: lanes per pixel = a power of two (>= 4: J lanes store) dividing C/4, every lane owns nq whole quads
    int lpp = 64;
    while (lpp > 4 && (C4 % lpp || lpp > C4)) lpp >>= 1;
    // 4 channel quads per lane where the channel count allows: 3-5 shuffle steps per dot product instead of 5-6 and four
    // 16-byte loads per lane and pixel in flight (128 ch @256^2: 68.5 -> 46.4 us, 256 ch @128^2: 53.1 -> 40.0 us; 8 quads: slower)
    while (lpp > 4 && C4 / lpp < 4 && !(C4 % (lpp >> 1))) lpp >>= 1;
    const int nq = C4 / lpp;
    if (C4 % lpp || lpp > C4 || nq > 8) {
        lpp = 64;
        while (lpp > C4) lpp >>= 1;
        if (lpp < J) lpp = 4;
        int64_t nbg = cdiv64(P, 256 / lpp);
        if (nbg > 4096) nbg = 4096;
        hipLaunchKernelGGL(thin_fwd_generic_kernel, dim3((unsigned)nbg, N), dim3(256), 0, st, x, W, w_bstride, add, t, P, C,
                           J, lpp, m);
        RICK_LAUNCH_STATUS();
    }
    int64_t nb = cdiv64(P, (int64_t)(256 / lpp) * 4);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const dim3 grid((unsigned)nb, N), blk(256);
#define THIN_FWD_LAUNCH(Q) hipLaunchKernelGGL(thin_fwd_kernel<Q>, grid, blk, 0, st, x, W, w_bstride, add, t, P, C, J, lpp, m)
    switch (nq) {
    case 1: THIN_FWD_LAUNCH(1); break;
    case 2: THIN_FWD_LAUNCH(2); break;
    case 3: THIN_FWD_LAUNCH(3); break;
    case 4: THIN_FWD_LAUNCH(4); break;
    case 5: THIN_FWD_LAUNCH(5); break;
    case 6: THIN_FWD_LAUNCH(6); break;
    case 7: THIN_FWD_LAUNCH(7); break;
    default: THIN_FWD_LAUNCH(8); break;
    }
#undef THIN_FWD_LAUNCH
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_thin_fwd_f32(const float *x, const float *W, int64_t w_bstride, const float *add, float *t,
                                 int N, int64_t P, int C, int J, void *stream) {
    const ThinMod none = {nullptr, 1.f, nullptr};
    return thin_fwd_launch(x, W, w_bstride, add, t, N, P, C, J, none, stream);
}

extern "C" int rick_torgb_fwd_f32(const float *x, const float *w, const float *s, float wscale, const float *bias,
                                  const float *add, float *t, int N, int64_t P, int C, int J, void *stream) {
    if (!s || (((uintptr_t)s | (uintptr_t)w) & 15)) return RICK_EINVAL;
    const ThinMod m = {s, wscale, bias};
    return thin_fwd_launch(x, w, 0, add, t, N, P, C, J, m, stream);
}

// x[n,p,c] = sum_j t[n,j,p] * W[n,j,c]     (ACC: x += ..., the second gradient arriving at a branch point added where it is produced)
template <bool ACC>
__global__ __launch_bounds__(256) void thin_bwdx_kernel(const float *__restrict__ t, const float *__restrict__ W,
                                                        int64_t w_bstride, float *__restrict__ x, int64_t P, int C, int J,
                                                        ThinMod m) {
    const int n = blockIdx.y;
    const int C4 = C >> 2;
    const int64_t total4 = P * C4;
    const float *Wn = W + (int64_t)n * w_bstride;
    const float *tn = t + (int64_t)n * J * P;
    float4 *xn = reinterpret_cast<float4 *>(x + (int64_t)n * P * C);
    for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * 256) {
        const int64_t p = i4 / C4;
        const int c4 = (int)(i4 - p * C4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float tv[THIN_MAXJ];
        float4 wv[THIN_MAXJ];
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) {   // unconditional loads (row J-1 again for j >= J), zero weight after
            const int jj = j < J ? j : J - 1;
            tv[j] = tn[(int64_t)jj * P + p];
            wv[j] = thin_w(Wn + (int64_t)jj * C, m, n, C, c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) {
            const float tj = j < J ? tv[j] : 0.f;
            acc.x += tj * wv[j].x; acc.y += tj * wv[j].y; acc.z += tj * wv[j].z; acc.w += tj * wv[j].w;
        }
        if (ACC) {
            const float4 o = xn[i4];
            acc = make_float4(o.x + acc.x, o.y + acc.y, o.z + acc.z, o.w + acc.w);
        }
        xn[i4] = acc;
    }
}

// The discriminator's input layer in one pass (model_probe_tune.py:679 + its FusedLeakyReLU): x = gain * lrelu(sum_j t_j W[j,c]
// + b[c]), written as fp32 and — for the first ResBlock's convolutions — as a split image (conv_common.h).  Same operation order
// as thin_bwdx_kernel followed by bias_act_kernel: bit-identical values, without writing the 128-channel map twice and reading
// it once in between.
__global__ __launch_bounds__(256) void d_input_kernel(const float *__restrict__ t, const float *__restrict__ W,
                                                      const float *__restrict__ bias, float *__restrict__ x, int64_t P, int C,
                                                      int J, float slope, float gain, rick_split_out xo) {
    const int n = blockIdx.y;
    const int C4 = C >> 2;
    const int64_t total4 = P * C4;
    const float *tn = t + (int64_t)n * J * P;
    float4 *xn = reinterpret_cast<float4 *>(x + (int64_t)n * P * C);
    float sscale = 1.f, am = 0.f;
    if (xo.split_out) {
        const cv_split_hdr h = cv_split_header(xo.bound0, xo.bound1, xo.bound_coef);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *reinterpret_cast<cv_split_hdr *>(xo.split_hdr) = h;
        sscale = cv_uniform(h.scale);
    }
    unsigned char *sn = (unsigned char *)xo.split_out + (int64_t)n * P * C * 4;
    for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * 256) {
        const int64_t p = i4 / C4;
        const int c4 = (int)(i4 - p * C4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float tv[THIN_MAXJ];
        float4 wv[THIN_MAXJ];
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) {
            const int jj = j < J ? j : J - 1;
            tv[j] = tn[(int64_t)jj * P + p];
            wv[j] = *reinterpret_cast<const float4 *>(W + (int64_t)jj * C + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) {
            const float tj = j < J ? tv[j] : 0.f;
            acc.x += tj * wv[j].x; acc.y += tj * wv[j].y; acc.z += tj * wv[j].z; acc.w += tj * wv[j].w;
        }
        const float4 bv = *reinterpret_cast<const float4 *>(bias + c4 * 4);
        acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
        float4 y;
        y.x = (acc.x > 0.f ? acc.x : acc.x * slope) * gain;
        y.y = (acc.y > 0.f ? acc.y : acc.y * slope) * gain;
        y.z = (acc.z > 0.f ? acc.z : acc.z * slope) * gain;
        y.w = (acc.w > 0.f ? acc.w : acc.w * slope) * gain;
        xn[i4] = y;
        if (xo.split_out) {
            cv_split_store4(sn + i4 * 16, 0, y, sscale);
            am = amax4(am, y);
        }
    }
    if (xo.split_out) cv_sat_check(am, sscale);
}

extern "C" int rick_d_input_f32(const float *t, const float *W, const float *bias, float *x, int N, int64_t P, int C, int J,
                                float slope, float gain, const rick_split_out *ex, void *stream) {
    if (!x || !W || !t || !bias || N <= 0 || P <= 0 || C <= 0 || C % 4 || J < 1 || J > THIN_MAXJ || N > 65535) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)bias) % 16) return RICK_EINVAL;
    rick_split_out xo = {nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, 0, 0, nullptr, nullptr, 0.f, 1.f, nullptr};
    if (ex) {
        xo = *ex;
        if (xo.split_out && (!xo.split_hdr || !xo.bound0 || !(xo.bound_coef > 0.f) || ((uintptr_t)xo.split_out % 16))) return RICK_EINVAL;
    }
    int64_t nb = cdiv64(P * (C / 4), 256 * 2);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(d_input_kernel, dim3((unsigned)nb, N), dim3(256), 0, (hipStream_t)stream, t, W, bias, x, P, C, J, slope, gain, xo);
    RICK_LAUNCH_STATUS();
}

// Column-owner form of thin_bwdx (C / 4 divides 256): a thread owns one channel quad — its J modulated weight rows live in
// registers for the whole launch — and walks the block's pixels, 256 / (C / 4) of them per step, TB_UNR steps in flight.  The
// round-4 form (one output quad per thread and iteration: an int64 division, J weight loads with the style multiply and J
// scalar loads for 16 bytes stored, 8 192 single-iteration blocks at 512 ch @64^2) ran 2.7 TB/s there; same sums, same order.
#define TB_UNR 4
template <bool ACC>
__global__ __launch_bounds__(256) void thin_bwdx_cols_kernel(const float *__restrict__ t, const float *__restrict__ W,
                                                             int64_t w_bstride, float *__restrict__ x, int64_t P, int C, int J,
                                                             ThinMod m, int pix_per_block) {
    const int n = blockIdx.y;
    const int C4 = C >> 2, rows = 256 / C4;
    const int c4 = threadIdx.x % C4, r = threadIdx.x / C4;
    const float *Wn = W + (int64_t)n * w_bstride;
    const float *tn = t + (int64_t)n * J * P;
    float4 *xn = reinterpret_cast<float4 *>(x + (int64_t)n * P * C);
    float4 wv[THIN_MAXJ];
#pragma unroll
    for (int j = 0; j < THIN_MAXJ; j++) wv[j] = thin_w(Wn + (int64_t)(j < J ? j : J - 1) * C, m, n, C, c4 * 4);
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < P ? p0 + pix_per_block : P;
    auto one = [&](int64_t p, const float (&tv)[THIN_MAXJ], const float4 old) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) {
            const float tj = j < J ? tv[j] : 0.f;
            acc.x += tj * wv[j].x; acc.y += tj * wv[j].y; acc.z += tj * wv[j].z; acc.w += tj * wv[j].w;
        }
        if (ACC) acc = make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w);
        xn[p * C4 + c4] = acc;
    };
    int64_t p = p0 + r;
    for (; p + (TB_UNR - 1) * rows < p1; p += TB_UNR * rows) {
        float tv[TB_UNR][THIN_MAXJ];
        float4 old[TB_UNR];
#pragma unroll
        for (int u = 0; u < TB_UNR; u++) {
#pragma unroll
            for (int j = 0; j < THIN_MAXJ; j++) tv[u][j] = tn[(int64_t)(j < J ? j : J - 1) * P + p + u * rows];
            old[u] = ACC ? xn[(p + u * rows) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < TB_UNR; u++) one(p + u * rows, tv[u], old[u]);
    }
    for (; p < p1; p += rows) {
        float tv[THIN_MAXJ];
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) tv[j] = tn[(int64_t)(j < J ? j : J - 1) * P + p];
        one(p, tv, ACC ? xn[p * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

CV_DEFINE_SAT_ACCESSOR(rick_sat_thin)

static int thin_bwdx_launch(const float *t, const float *W, int64_t w_bstride, float *x, int N, int64_t P, int C, int J, ThinMod m,
                            void *stream, bool acc = false) {
    if (!x || !W || !t || N <= 0 || P <= 0 || C <= 0 || C % 4 || J < 1 || J > THIN_MAXJ || N > 65535) return RICK_EINVAL;
    const int C4 = C / 4;
    if (C4 <= 256 && 256 % C4 == 0) {
        // ~1 024 blocks (4 per CU) with at least TB_UNR steps each
        const int rows = 256 / C4;
        int64_t ppb = cdiv64(P * N, 1024);      // (512 / 1024 / 2048 / 4096 blocks measured: 8.4 / 8.2 / 9.3 / 9.4 us at 512 ch @64^2)
        if (ppb < (int64_t)TB_UNR * rows) ppb = (int64_t)TB_UNR * rows;
        ppb = cdiv64(ppb, rows) * rows;
        const unsigned nbx = (unsigned)cdiv64(P, ppb);
        if (acc) hipLaunchKernelGGL(thin_bwdx_cols_kernel<true>, dim3(nbx, N), dim3(256), 0, (hipStream_t)stream, t, W, w_bstride, x, P, C, J, m, (int)ppb);
        else hipLaunchKernelGGL(thin_bwdx_cols_kernel<false>, dim3(nbx, N), dim3(256), 0, (hipStream_t)stream, t, W, w_bstride, x, P, C, J, m, (int)ppb);
        RICK_LAUNCH_STATUS();
    }
    int64_t nb = cdiv64(P * (C / 4), 256);
    if (nb > 4096) nb = 4096;
    if (acc) hipLaunchKernelGGL(thin_bwdx_kernel<true>, dim3((unsigned)nb, N), dim3(256), 0, (hipStream_t)stream, t, W, w_bstride, x, P, C, J, m);
    else hipLaunchKernelGGL(thin_bwdx_kernel<false>, dim3((unsigned)nb, N), dim3(256), 0, (hipStream_t)stream, t, W, w_bstride, x, P, C, J, m);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_thin_bwdx_f32(const float *t, const float *W, int64_t w_bstride, float *x, int N, int64_t P, int C,
                                  int J, void *stream) {
    const ThinMod none = {nullptr, 1.f, nullptr};
    return thin_bwdx_launch(t, W, w_bstride, x, N, P, C, J, none, stream);
}

extern "C" int rick_torgb_bwdx_f32(const float *g, const float *w, const float *s, float wscale, float *gx, int N, int64_t P,
                                   int C, int J, void *stream) {
    if (!s || (((uintptr_t)s | (uintptr_t)w) & 15)) return RICK_EINVAL;
    const ThinMod m = {s, wscale, nullptr};
    return thin_bwdx_launch(g, w, 0, gx, N, P, C, J, m, stream);
}

extern "C" int rick_torgb_bwdx_acc_f32(const float *g, const float *w, const float *s, float wscale, float *gx, int N, int64_t P,
                                       int C, int J, void *stream) {
    if (!s || (((uintptr_t)s | (uintptr_t)w | (uintptr_t)gx) & 15)) return RICK_EINVAL;
    const ThinMod m = {s, wscale, nullptr};
    return thin_bwdx_launch(g, w, 0, gx, N, P, C, J, m, stream, true);
}

// G[n,j,c] = sum_p t[n,j,p] * x[n,p,c];  partials [blk][n][j][c]
// (64 rows per block: a 64 x 64 map of 4 images gives 256 blocks; with 256 rows the 16 x 4 blocks of that launch read their
// 33 MB at 0.75 TB/s)
#define THINW_ROWS 64
extern "C" int rick_thin_wgrad_blocks(int64_t P) {
    int64_t nb = cdiv64(P, THINW_ROWS);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    if (cdiv64(P, nb) > 2048) nb = cdiv64(P, 2048);      // the block's slice of t sits in LDS (J x pixels)
    return (int)nb;
}

// float4 over channels: thread owns 4 consecutive channels of a row-lane, J x 4 accumulators.  The block's slice of t
// (J x pixels) is staged in LDS once (coalesced) — the per-row scalar loads of round 4 sat in front of every FMA group — and
// TW_UNR rows are in flight per thread (same per-thread summation order: rows in increasing p).
#define TW_UNR 8
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float *__restrict__ t, const float *__restrict__ x,
                                                         float *__restrict__ partials, int64_t P, int C, int J) {
    extern __shared__ float lds[];   // [256 * 4] reduction scratch, then [J][ppb] slice of t
    const int n = blockIdx.y, N = gridDim.y, nb = gridDim.x;
    const int64_t ppb = cdiv64(P, nb);
    const int64_t p0 = (int64_t)blockIdx.x * ppb, p1 = p0 + ppb < P ? p0 + ppb : P;
    const int np = (int)(p1 - p0);
    const float *xn = x + (int64_t)n * P * C;
    const float *tn = t + (int64_t)n * J * P;
    float *pb = partials + ((int64_t)blockIdx.x * N + n) * J * C;
    float *st = lds + 1024;                                   // [THIN_MAXJ][ppb]; rows j >= J hold row J - 1 (never stored)
    for (int e = threadIdx.x; e < THIN_MAXJ * np; e += 256) {
        const int j = e / np, i = e - j * np;
        st[j * (int)ppb + i] = tn[(int64_t)(j < J ? j : J - 1) * P + p0 + i];
    }
    __syncthreads();
    const int ncol = C >> 2;
    for (int cbase = 0; cbase < ncol; cbase += 256) {
        const int cg = ncol - cbase < 256 ? ncol - cbase : 256;
        const int rpb = 256 / cg;
        const int lane_c = threadIdx.x % cg, lane_r = threadIdx.x / cg;
        float4 acc[THIN_MAXJ];
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_r < rpb) {
            auto row = [&](int i, const float4 xv) {
#pragma unroll
                for (int j = 0; j < THIN_MAXJ; j++) {   // accumulators of rows j >= J are never stored
                    const float tv = st[j * (int)ppb + i];
                    acc[j].x += tv * xv.x; acc[j].y += tv * xv.y; acc[j].z += tv * xv.z; acc[j].w += tv * xv.w;
                }
            };
            const float *xc = xn + p0 * C + (int64_t)(cbase + lane_c) * 4;
            int i = lane_r;
            for (; i + (TW_UNR - 1) * rpb < np; i += TW_UNR * rpb) {
                float4 xv[TW_UNR];
#pragma unroll
                for (int u = 0; u < TW_UNR; u++) xv[u] = *reinterpret_cast<const float4 *>(xc + (int64_t)(i + u * rpb) * C);
#pragma unroll
                for (int u = 0; u < TW_UNR; u++) row(i + u * rpb, xv[u]);
            }
            for (; i < np; i += rpb) row(i, *reinterpret_cast<const float4 *>(xc + (int64_t)i * C));
        }
#pragma unroll
        for (int j = 0; j < THIN_MAXJ; j++) {
            if (j >= J) break;
            __syncthreads();
            reinterpret_cast<float4 *>(lds)[threadIdx.x] = lane_r < rpb ? acc[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            if (threadIdx.x < cg) {
                float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int rr = 0; rr < rpb; rr++) {
                    const float4 v = reinterpret_cast<float4 *>(lds)[rr * cg + threadIdx.x];
                    sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
                }
                *reinterpret_cast<float4 *>(pb + (int64_t)j * C + (cbase + threadIdx.x) * 4) = sacc;
            }
        }
    }
}

// out[i] = sum_b partials[b][i]: 32 outputs x 8 groups per block, group g adds blocks g, g + 8, ... (four loads in flight), the
// eight group sums are added in index order — a fixed order, and 48-192 blocks instead of the 6-24 single-thread-per-output
// blocks of round 4 (21 us for 1 536 outputs x 256 partials: longer than the 134 MB pass that produced them).
__global__ __launch_bounds__(256) void thin_partial_sum_kernel(const float *__restrict__ partials, float *__restrict__ out,
                                                               int nb, int64_t n) {
    __shared__ float red[8][32];
    const int lo = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + lo;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int b = grp;
        for (; b + 24 < nb; b += 32) {
            s0 += partials[(int64_t)b * n + i];
            s1 += partials[(int64_t)(b + 8) * n + i];
            s2 += partials[(int64_t)(b + 16) * n + i];
            s3 += partials[(int64_t)(b + 24) * n + i];
        }
        for (; b < nb; b += 8) s0 += partials[(int64_t)b * n + i];
    }
    red[grp][lo] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < n) {
        float s = red[0][lo];
#pragma unroll
        for (int g = 1; g < 8; g++) s += red[g][lo];
        out[i] = s;
    }
}

extern "C" int rick_thin_wgrad_f32(const float *t, const float *x, float *G, int N, int64_t P, int C, int J,
                                   float *partials, void *stream) {
    if (!x || !G || !t || !partials || N <= 0 || P <= 0 || C <= 0 || (C & 3) || J < 1 || J > THIN_MAXJ || N > 65535) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)partials) % 16) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nb = rick_thin_wgrad_blocks(P);
    const size_t lds = (1024 + (size_t)THIN_MAXJ * cdiv64(P, nb)) * sizeof(float);
    if (lds > 64 * 1024) return RICK_EINVAL;
    hipLaunchKernelGGL(thin_wgrad_kernel, dim3(nb, N), dim3(256), lds, st, t, x, partials, P, C, J);
    const int64_t n = (int64_t)N * J * C;
    hipLaunchKernelGGL(thin_partial_sum_kernel, dim3((unsigned)cdiv64(n, 32)), dim3(256), 0, st, partials, G, nb, n);
    RICK_LAUNCH_STATUS();
}
