// HBM-bound kernels of the hot path: fused bias/noise/LeakyReLU (+ one-pass backward with
// wavefront-shuffle reductions), modulation helpers, ResBlock merge, minibatch-stddev,
// Fisher grad^2 accumulation, per-filter reduction, masked Adam, EMA.
#include "conv_common.h"

// ------------------------------------------------------------------------ bias + act (fwd)
// Semantics: op/fused_bias_act_kernel.cu:18-49 (+ fused NoiseInjection, model_probe_tune.py:293-298)
struct BiasActParams {
    int64_t n, step_b, size_b, n_div, hw_div, noise_nb, noise_hw;
    int act, grad;
    float alpha, scale;
};

__device__ __forceinline__ float act_apply(float v, float r, int mode, float alpha) {
    switch (mode) {
        case 30: return v > 0.f ? v : v * alpha;
        case 31: return r > 0.f ? v : v * alpha;
        case 12:
        case 32: return 0.f;
        default: return v;
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void bias_act_kernel(const float *__restrict__ x, const float *__restrict__ b,
                                                       const float *__restrict__ ref, float *__restrict__ out,
                                                       const float *__restrict__ noise, const float *__restrict__ nw,
                                                       BiasActParams p) {
    const int mode = p.act * 10 + p.grad;
    const float nwv = noise ? nw[0] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (VEC4) {
        // requires n % 4 == 0, step_b == 1, size_b % 4 == 0, hw_div % 4 == 0 (channels-last rows)
        const int64_t n4 = p.n >> 2;
        for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += stride) {
            const int64_t i = i4 << 2;
            float4 v = reinterpret_cast<const float4 *>(x)[i4];
            if (b) {
                const float4 bv = *reinterpret_cast<const float4 *>(b + (i % p.size_b));
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (noise) {
                const float nv = nwv * noise[((i / p.n_div) % p.noise_nb) * p.noise_hw + (i / p.hw_div) % p.noise_hw];
                v.x += nv; v.y += nv; v.z += nv; v.w += nv;
            }
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ref) r = reinterpret_cast<const float4 *>(ref)[i4];
            float4 y;
            y.x = act_apply(v.x, r.x, mode, p.alpha) * p.scale;
            y.y = act_apply(v.y, r.y, mode, p.alpha) * p.scale;
            y.z = act_apply(v.z, r.z, mode, p.alpha) * p.scale;
            y.w = act_apply(v.w, r.w, mode, p.alpha) * p.scale;
            reinterpret_cast<float4 *>(out)[i4] = y;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += stride) {
            float v = x[i];
            if (b) v += b[(i / p.step_b) % p.size_b];
            if (noise) v += nwv * noise[((i / p.n_div) % p.noise_nb) * p.noise_hw + (i / p.hw_div) % p.noise_hw];
            const float r = ref ? ref[i] : 0.f;
            out[i] = act_apply(v, r, mode, p.alpha) * p.scale;
        }
    }
}

static inline int ew_grid(int64_t work_items) {
    int64_t g = cdiv64(work_items, 256);
    if (g > 256 * 16) g = 256 * 16;   // 16 blocks per CU, grid-stride beyond
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int rick_bias_act_f32(const float *x, const float *bias, const float *ref, float *out,
                                 int64_t n, int64_t step_b, int64_t size_b, int act, int grad,
                                 float alpha, float scale, const float *noise, const float *nw,
                                 int64_t n_div, int64_t hw_div, int64_t noise_nb, int64_t noise_hw,
                                 void *stream) {
    if (!x || !out || n < 0 || (bias && (step_b <= 0 || size_b <= 0))) return RICK_EINVAL;
    if (noise && (!nw || n_div <= 0 || hw_div <= 0 || noise_nb <= 0 || noise_hw <= 0)) return RICK_EINVAL;
    if (n == 0) return 0;
    BiasActParams p{n, step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1, n_div, hw_div, noise_nb, noise_hw,
                    act, grad, alpha, scale};
    const bool vec = (n % 4 == 0) && (!bias || (step_b == 1 && size_b % 4 == 0)) &&
                     (!noise || (hw_div % 4 == 0 && n_div % 4 == 0)) &&
                     (((uintptr_t)x | (uintptr_t)out | (uintptr_t)(ref ? ref : x) | (uintptr_t)(bias ? bias : x)) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(bias_act_kernel<true>, dim3(ew_grid(n / 4)), dim3(256), 0, st, x, bias, ref, out, noise, nw, p);
    else
        hipLaunchKernelGGL(bias_act_kernel<false>, dim3(ew_grid(n)), dim3(256), 0, st, x, bias, ref, out, noise, nw, p);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------- bias + act (backward)
// One pass over g / ref: writes gx and per-block partial sums for gb[C] and gnw.
// Block = 256 threads handling a contiguous slab of rows; thread t owns column group
// (t % cg) and walks rows t / cg, t / cg + rpb, ...  (cg = min(C/4 or C, 256) lanes per row).
#define BAB_ROWS_PER_BLOCK 16

extern "C" int rick_bias_act_bwd_blocks(int64_t rows, int C) {
    (void)C;
    int64_t nb = cdiv64(rows, BAB_ROWS_PER_BLOCK);
    if (nb > 512) nb = 512;        // (512 / 1024 / 2048 / 4096 blocks: 4.76 / 4.52 / 4.66 / 4.56 TB/s at 128 ch @256^2, tools/bench_actbwd.py)
    if (nb < 1) nb = 1;
    return (int)nb;
}

// SPL: the adjoint leaves as split images instead of an fp32 tensor (its only consumers are the data- and weight-gradient
// MFMA kernels): image 1 = the adjoint itself, optional image 2 = g * mul2 (the same incoming gradient on its way into a
// parallel linear branch — the ResBlock skip path), both with bounds derived from one measured maximum of g.
struct BabSplit {
    unsigned char *s1, *s2;
    cv_split_hdr *h1, *h2;
    const float *bound;          // max |g| (exact or an upper bound)
    float coef1, coef2, mul2;
    const float *cs1;            // NULL, or [images, C]: image 1 holds adjoint * cs1[n, c] (a demodulation scale folded in)
    int64_t rows_per_img;
    float *f32;                  // NULL, or: also write the (unscaled) adjoint as fp32
    const float *bound1;         // NULL, or the amax word that bounds image 1 by itself (then coef1 = 1)
};

// DOT (the generator's demodulated layers): the same pass also reduces sum_rows adjoint * t per channel and block, t = the
// convolution's output reconstructed from the saved activation output (y / gain or y / (gain * slope), minus noise and bias) —
// the demodulation gradient d/d(d) of y = act(d * conv + ...) is sum_hw adjoint * conv_out / d.  Round 4 read the adjoint and y
// a second time for it (hw_dot_kernel<.., ACT>: 8 B per element, 0.23 ms per G step).  Blocks never straddle an image here (the
// host checks), so the per-image sums are the column sums of the image's block range (bab_dot_reduce_kernel).
struct BabDot {
    float *dpart;                // NULL: off; else [blocks][C]
    const float *bias;           // [C] or NULL
    const float *noise_w;        // [1] or NULL (no noise term)
    float inv_gain, inv_gain_slope;
};

template <bool VEC4, bool SPL = false>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float *__restrict__ g, const float *__restrict__ ref,
                                                           float *__restrict__ gx, const float *__restrict__ noise,
                                                           float *__restrict__ partials, int64_t rows, int C,
                                                           int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw,
                                                           float alpha, float scale, int want_gb, BabSplit sp, BabDot dot) {
    extern __shared__ float lds[];   // [256 * (VEC4 ? 4 : 1)] column partials + 4 for block_sum
    const bool DOT = VEC4 && !SPL && dot.dpart != nullptr;
    const float dot_nw = (DOT && dot.noise_w) ? dot.noise_w[0] : 0.f;
    float sc1 = 1.f, sc2 = 1.f, am1 = 0.f, am2 = 0.f;
    if (SPL) {
        const cv_split_hdr h1 = cv_split_header(sp.bound1 ? sp.bound1 : sp.bound, nullptr, sp.coef1);
        sc1 = cv_uniform(h1.scale);
        if (blockIdx.x == 0 && threadIdx.x == 0) *sp.h1 = h1;
        if (sp.s2) {
            const cv_split_hdr h2 = cv_split_header(sp.bound, nullptr, sp.coef2);
            sc2 = cv_uniform(h2.scale);
            if (blockIdx.x == 0 && threadIdx.x == 0) *sp.h2 = h2;
        }
    }
    auto put = [&](float *op, int64_t row, int c, const float4 o, const float4 gv) {
        if (!SPL) {
            *reinterpret_cast<float4 *>(op) = o;
            return;
        }
        if (sp.f32) *reinterpret_cast<float4 *>(sp.f32 + row * C + c) = o;
        float4 o1 = o;
        if (sp.cs1) {
            const float4 cs = *reinterpret_cast<const float4 *>(sp.cs1 + (row / sp.rows_per_img) * C + c);
            o1 = make_float4(o.x * cs.x, o.y * cs.y, o.z * cs.z, o.w * cs.w);
        }
        cv_split_store4(sp.s1 + row * C * 4, c, o1, sc1);
        am1 = amax4(am1, o1);
        if (sp.s2) {
            const float4 g2 = make_float4(gv.x * sp.mul2, gv.y * sp.mul2, gv.z * sp.mul2, gv.w * sp.mul2);
            cv_split_store4(sp.s2 + row * C * 4, c, g2, sc2);
            am2 = amax4(am2, g2);
        }
    };
    const int W = VEC4 ? 4 : 1;
    const int ncol = C / W;                       // column groups per row
    const int nb = gridDim.x;
    const int64_t rows_per_block = cdiv64(rows, nb);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float nsum = 0.f;
    float *pb = partials + (int64_t)blockIdx.x * (C + 1);
    const bool per_sample = noise_nb * rows_per_img == rows;
    // column groups are processed in chunks of up to 256 lanes
    for (int cbase = 0; cbase < ncol; cbase += 256) {
        const int cg = ncol - cbase < 256 ? ncol - cbase : 256;   // lanes per row in this chunk
        const int rpb = 256 / cg;                                  // rows handled concurrently
        const int lane_c = threadIdx.x % cg, lane_r = threadIdx.x / cg;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float dacc[4] = {0.f, 0.f, 0.f, 0.f};
        float4 dbias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DOT && dot.bias && lane_r < rpb) dbias = *reinterpret_cast<const float4 *>(dot.bias + (cbase + lane_c) * 4);
        auto dot_add = [&](const float4 o, const float4 y, float nv) {
            const float nb_ = dot_nw * nv;
            dacc[0] += o.x * ((y.x > 0.f ? y.x * dot.inv_gain : y.x * dot.inv_gain_slope) - nb_ - dbias.x);
            dacc[1] += o.y * ((y.y > 0.f ? y.y * dot.inv_gain : y.y * dot.inv_gain_slope) - nb_ - dbias.y);
            dacc[2] += o.z * ((y.z > 0.f ? y.z * dot.inv_gain : y.z * dot.inv_gain_slope) - nb_ - dbias.z);
            dacc[3] += o.w * ((y.w > 0.f ? y.w * dot.inv_gain : y.w * dot.inv_gain_slope) - nb_ - dbias.w);
        };
        if (lane_r < rpb) {
            // noise index of row r = (image % noise_nb) * noise_hw + pixel, advanced incrementally (no division per row)
            int64_t r = r0 + lane_r;
            int64_t img = noise ? r / rows_per_img : 0;
            int64_t pix = noise ? r - img * rows_per_img : 0;   // host guarantees noise_hw == rows_per_img
            img = noise ? img % noise_nb : 0;
            const float *gp = g + r * C + (int64_t)(cbase + lane_c) * W;
            const float *rp = ref + r * C + (int64_t)(cbase + lane_c) * W;
            float *op = gx + r * C + (int64_t)(cbase + lane_c) * W;
            const int64_t step = (int64_t)rpb * C;
            if (VEC4) {
                // 4 rows per iteration: 8 independent 16-byte loads in flight per thread
                for (; r + 3 * rpb < r1; r += 4 * rpb) {
                    float4 gv[4], rv[4];
                    float nv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        gv[u] = *reinterpret_cast<const float4 *>(gp + u * step);
                        rv[u] = *reinterpret_cast<const float4 *>(rp + u * step);
                    }
                    if (noise && per_sample) {   // one noise map per image: the noise index is the row index
#pragma unroll
                        for (int u = 0; u < 4; u++) nv[u] = noise[r + (int64_t)u * rpb];
                    } else if (noise) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            nv[u] = noise[img * noise_hw + pix];
                            pix += rpb;
                            while (pix >= rows_per_img) {
                                pix -= rows_per_img;
                                img = img + 1 == noise_nb ? 0 : img + 1;
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        float4 o;
                        o.x = gv[u].x * (rv[u].x > 0.f ? 1.f : alpha) * scale;
                        o.y = gv[u].y * (rv[u].y > 0.f ? 1.f : alpha) * scale;
                        o.z = gv[u].z * (rv[u].z > 0.f ? 1.f : alpha) * scale;
                        o.w = gv[u].w * (rv[u].w > 0.f ? 1.f : alpha) * scale;
                        put(op + u * step, r + (int64_t)u * rpb, (cbase + lane_c) * 4, o, gv[u]);
                        acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
                        nsum += (o.x + o.y + o.z + o.w) * nv[u];
                        if (DOT) dot_add(o, rv[u], nv[u]);
                    }
                    gp += 4 * step;
                    rp += 4 * step;
                    op += 4 * step;
                }
            }
            for (; r < r1; r += rpb) {
                float nv = 0.f;
                if (noise && per_sample) {
                    nv = noise[r];
                } else if (noise) {
                    nv = noise[img * noise_hw + pix];
                    pix += rpb;
                    while (pix >= rows_per_img) {
                        pix -= rows_per_img;
                        img = img + 1 == noise_nb ? 0 : img + 1;
                    }
                }
                if (VEC4) {
                    const float4 gv = *reinterpret_cast<const float4 *>(gp);
                    const float4 rv = *reinterpret_cast<const float4 *>(rp);
                    float4 o;
                    o.x = gv.x * (rv.x > 0.f ? 1.f : alpha) * scale;
                    o.y = gv.y * (rv.y > 0.f ? 1.f : alpha) * scale;
                    o.z = gv.z * (rv.z > 0.f ? 1.f : alpha) * scale;
                    o.w = gv.w * (rv.w > 0.f ? 1.f : alpha) * scale;
                    put(op, r, (cbase + lane_c) * 4, o, gv);
                    acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
                    nsum += (o.x + o.y + o.z + o.w) * nv;
                    if (DOT) dot_add(o, rv, nv);
                } else {
                    const float o = gp[0] * (rp[0] > 0.f ? 1.f : alpha) * scale;
                    op[0] = o;
                    acc[0] += o;
                    nsum += o * nv;
                }
                gp += step;
                rp += step;
                op += step;
            }
        }
        if (want_gb) {
            __syncthreads();
            for (int j = 0; j < W; j++) lds[threadIdx.x * W + j] = (lane_r < rpb) ? acc[j] : 0.f;
            __syncthreads();
            // reduce over the rpb row-lanes sharing a column group
            if (threadIdx.x < cg) {
                for (int j = 0; j < W; j++) {
                    float s = 0.f;
                    for (int rr = 0; rr < rpb; rr++) s += lds[(rr * cg + threadIdx.x) * W + j];
                    pb[(cbase + threadIdx.x) * W + j] = s;
                }
            }
        }
        if (DOT) {
            __syncthreads();
            for (int j = 0; j < W; j++) lds[threadIdx.x * W + j] = (lane_r < rpb) ? dacc[j] : 0.f;
            __syncthreads();
            if (threadIdx.x < cg) {
                for (int j = 0; j < W; j++) {
                    float s = 0.f;
                    for (int rr = 0; rr < rpb; rr++) s += lds[(rr * cg + threadIdx.x) * W + j];
                    dot.dpart[(int64_t)blockIdx.x * C + (cbase + threadIdx.x) * W + j] = s;
                }
            }
        }
    }
    if (noise && partials) {     // (DOT passes the noise for its reconstruction even when no parameter gradient is wanted)
        const float tot = block_sum_256(nsum, lds + 256 * W);
        if (threadIdx.x == 0) pb[C] = tot;
    }
    if (SPL) {
        cv_sat_check(am1, sc1);
        cv_sat_check(am2, sc2);
    }
}

// gd[n, c] = (sum of the image's block partials) / divisor[n, c]: block (8 columns, image), row groups summed by a fixed tree
__global__ __launch_bounds__(256) void bab_dot_reduce_kernel(const float *__restrict__ dpart, const float *__restrict__ divisor,
                                                             float *__restrict__ gd, int bpi, int C) {
    constexpr int CPB = 8, G = 256 / CPB;
    __shared__ float red[256];
    const int cl = threadIdx.x % CPB, grp = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl, n = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const float *pp = dpart + (int64_t)n * bpi * C + c;
        float s0 = 0.f, s1 = 0.f;
        int b = grp;
        for (; b + G < bpi; b += 2 * G) {
            s0 += pp[(int64_t)b * C];
            s1 += pp[(int64_t)(b + G) * C];
        }
        for (; b < bpi; b += G) s0 += pp[(int64_t)b * C];
        s = s0 + s1;
    }
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        if (grp < off) red[threadIdx.x] += red[threadIdx.x + off * CPB];
        __syncthreads();
    }
    if (grp == 0 && c < C) gd[(int64_t)n * C + c] = red[cl] / divisor[(int64_t)n * C + c];
}

// Destination options of the column-sum stage: columns >= split go to out2 (the noise-strength gradient behind the C bias
// columns), the sum may be divided elementwise (demodulation gradient: sum / d) and added to what the destination
// holds (gradient sink: the destination is the parameter's .grad).
struct ColsumOut {
    float *out2;
    int split;
    const float *divisor;
    int accumulate;
};

// out[c] = sum_b partials[b*stride + col0 + c].  Block = CPB columns x (256 / CPB) row groups; a thread walks
// its group's partial rows with 4 loads in flight, the group sums are combined by a fixed LDS tree
// (deterministic).  CPB = 8 keeps many rows in parallel (the stage is latency-bound: few KB..MB of partials),
// CPB = 1 spends the whole block on a single column (the noise-strength gradient).
template <int CPB>
__global__ __launch_bounds__(256) void partial_colsum_kernel(const float *__restrict__ partials, float *__restrict__ out,
                                                             int nb, int stride, int ncols, int col0, ColsumOut o) {
    constexpr int G = 256 / CPB;
    __shared__ float red[256];
    const int cl = threadIdx.x % CPB, grp = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    float s = 0.f;
    if (c < ncols) {
        const float *pp = partials + col0 + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = grp;
        for (; b + 3 * G < nb; b += 4 * G) {
            s0 += pp[(int64_t)b * stride];
            s1 += pp[(int64_t)(b + G) * stride];
            s2 += pp[(int64_t)(b + 2 * G) * stride];
            s3 += pp[(int64_t)(b + 3 * G) * stride];
        }
        for (; b < nb; b += G) s0 += pp[(int64_t)b * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        if (grp < off) red[threadIdx.x] += red[threadIdx.x + off * CPB];
        __syncthreads();
    }
    if (grp == 0 && c < ncols) {
        float v = red[cl];
        if (o.divisor) v /= o.divisor[c];
        float *dst = c < o.split ? out + c : o.out2 + (c - o.split);
        *dst = o.accumulate ? *dst + v : v;
    }
}

static void launch_colsum(const float *partials, float *out, int nb, int stride, int ncols, int col0, hipStream_t st,
                          const float *divisor = nullptr, int accumulate = 0, float *out2 = nullptr, int split = 0) {
    const ColsumOut o = {out2, out2 ? split : ncols, divisor, accumulate};
    if (ncols == 1)
        hipLaunchKernelGGL(partial_colsum_kernel<1>, dim3(1), dim3(256), 0, st, partials, out, nb, stride, ncols, col0, o);
    else
        hipLaunchKernelGGL(partial_colsum_kernel<8>, dim3(cdiv(ncols, 8)), dim3(256), 0, st, partials, out, nb, stride, ncols,
                           col0, o);
}

static int bias_act_bwd_run(const float *g, const float *ref, float *gx, float *gb, float *gnw,
                            const float *noise, int64_t rows, int C, int64_t rows_per_img,
                            int64_t noise_nb, int64_t noise_hw, float alpha, float scale,
                            float *partials, int accumulate, void *stream, const BabSplit *sp, const BabDot *dotp = nullptr);

// Several column sums in ONE launch (rick_colsum_multi_f32): the second stages of the bias / noise-strength gradients of a whole
// backward pass, whose results nobody reads before the optimiser.  Each item is summed exactly as partial_colsum_kernel<8> (or <1>
// for a single column) sums it — same row groups, same four-way accumulation, same LDS tree: bit-identical.  The items travel
// by value in the kernel arguments (no device table: safe under hipGraph capture).
struct ColsumBatch {
    rick_colsum_item it[RICK_COLSUM_MAX];
    int blk_begin[RICK_COLSUM_MAX];
    int n;
};

__global__ __launch_bounds__(256) void colsum_multi_kernel(const ColsumBatch b) {
    __shared__ float red[256];
    int l = 0;
    for (int i = 1; i < b.n; i++)
        if ((int)blockIdx.x >= b.blk_begin[i]) l = i;
    const rick_colsum_item it = b.it[l];
    const int CPB = it.ncols == 1 ? 1 : 8, G = 256 / CPB;
    const int cl = threadIdx.x % CPB, grp = threadIdx.x / CPB;
    const int c = ((int)blockIdx.x - b.blk_begin[l]) * CPB + cl;
    float s = 0.f;
    if (c < it.ncols) {
        const float *pp = it.partials + it.col0 + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = grp;
        for (; r + 3 * G < it.nb; r += 4 * G) {
            s0 += pp[(int64_t)r * it.stride];
            s1 += pp[(int64_t)(r + G) * it.stride];
            s2 += pp[(int64_t)(r + 2 * G) * it.stride];
            s3 += pp[(int64_t)(r + 3 * G) * it.stride];
        }
        for (; r < it.nb; r += G) s0 += pp[(int64_t)r * it.stride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = G / 2; off > 0; off >>= 1) {
        if (grp < off) red[threadIdx.x] += red[threadIdx.x + off * CPB];
        __syncthreads();
    }
    if (grp == 0 && c < it.ncols) {
        const float v = red[cl];
        const int split = it.out2 ? it.split : it.ncols;
        float *dst = c < split ? it.out + c : it.out2 + (c - split);
        *dst = it.accumulate ? *dst + v : v;
    }
}

extern "C" int rick_colsum_multi_f32(const rick_colsum_item *items, int n, void *stream) {
    if (!items || n < 1) return RICK_EINVAL;
    for (int base = 0; base < n; base += RICK_COLSUM_MAX) {
        ColsumBatch b;
        const int m = n - base < RICK_COLSUM_MAX ? n - base : RICK_COLSUM_MAX;
        int blk = 0;
        for (int i = 0; i < m; i++) {
            const rick_colsum_item &it = items[base + i];
            if (!it.partials || !it.out || it.nb < 1 || it.ncols < 1 || it.stride < it.ncols) return RICK_EINVAL;
            b.it[i] = it;
            b.blk_begin[i] = blk;
            blk += it.ncols == 1 ? 1 : cdiv(it.ncols, 8);
        }
        b.n = m;
        hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, b);
    }
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_colsum_f32(const float *partials, float *out, int64_t rows, int stride, int ncols, int accumulate, void *stream) {
    if (!partials || !out || rows < 1 || rows > 0x7fffffff || stride < ncols || ncols < 1) return RICK_EINVAL;
    launch_colsum(partials, out, (int)rows, stride, ncols, 0, (hipStream_t)stream, nullptr, accumulate);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_bias_act_bwd_f32(const float *g, const float *ref, float *gx, float *gb, float *gnw,
                                     const float *noise, int64_t rows, int C, int64_t rows_per_img,
                                     int64_t noise_nb, int64_t noise_hw, float alpha, float scale,
                                     float *partials, int accumulate, void *stream) {
    return bias_act_bwd_run(g, ref, gx, gb, gnw, noise, rows, C, rows_per_img, noise_nb, noise_hw, alpha, scale, partials,
                            accumulate, stream, nullptr);
}

// 1 when the blocks of rick_bias_act_bwd_f32 never straddle an image of `rows_per_img` rows (needed by the _dot form)
extern "C" int rick_bias_act_bwd_dot_ok(int64_t rows, int C, int64_t rows_per_img) {
    if (rows <= 0 || C <= 0 || (C & 3) || rows_per_img <= 0 || rows % rows_per_img) return 0;
    const int nb = rick_bias_act_bwd_blocks(rows, C);
    const int64_t rpb = cdiv64(rows, nb);
    return (rows % nb == 0 && rows_per_img % rpb == 0) ? 1 : 0;
}

// rick_bias_act_bwd_f32 that ALSO returns the demodulation gradient of the layer the activation belongs to, from the same pass:
//   gd[n, c] = (sum over the image's rows of gx * t) / divisor[n, c],  t = unact(ref) - noise_w * noise - bias
// (unact(y) = y / scale for y > 0, y / (scale * alpha) otherwise).  dpartials: rick_bias_act_bwd_blocks() * C floats.
extern "C" int rick_bias_act_bwd_dot_f32(const float *g, const float *ref, float *gx, float *gb, float *gnw,
                                         const float *noise, int64_t rows, int C, int64_t rows_per_img,
                                         int64_t noise_nb, int64_t noise_hw, float alpha, float scale,
                                         float *partials, int accumulate, const float *bias, const float *noise_w,
                                         const float *divisor, float *gd, float *dpartials, void *stream) {
    if (!divisor || !gd || !dpartials || scale == 0.f || alpha == 0.f || !rick_bias_act_bwd_dot_ok(rows, C, rows_per_img)) return RICK_EINVAL;
    if (noise_w && !noise) return RICK_EINVAL;
    if (((uintptr_t)g | (uintptr_t)ref | (uintptr_t)gx | (uintptr_t)(bias ? bias : g)) % 16) return RICK_EINVAL;
    const BabDot dot = {dpartials, bias, noise_w, 1.f / scale, 1.f / (scale * alpha)};
    const int rc = bias_act_bwd_run(g, ref, gx, gb, gnw, noise, rows, C, rows_per_img, noise_nb, noise_hw, alpha, scale, partials,
                                    accumulate, stream, nullptr, &dot);
    if (rc) return rc;
    const int nb = rick_bias_act_bwd_blocks(rows, C);
    const int N = (int)(rows / rows_per_img), bpi = nb / N;
    hipLaunchKernelGGL(bab_dot_reduce_kernel, dim3((unsigned)cdiv(C, 8), (unsigned)N), dim3(256), 0, (hipStream_t)stream, dpartials,
                       divisor, gd, bpi, C);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_bias_act_bwd_split2_f32(const float *g, const float *ref, void *out1, float *hdr1, void *out2, float *hdr2,
                                            float mul2, const float *amax_g, const float *chan_scale, const float *bound1,
                                            float *out_f32, float *gb, float *gnw, const float *noise,
                                            int64_t rows, int C, int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw,
                                            float alpha, float scale, float *partials, int accumulate, void *stream);

// The activation adjoint written as split images: out1 = g * (ref > 0 ? 1 : alpha) * scale (bound |scale| * *amax_g) and,
// when out2 != NULL, out2 = g * mul2 (bound |mul2| * *amax_g).  C % 32 == 0; gb / gnw as in rick_bias_act_bwd_f32.
extern "C" int rick_bias_act_bwd_split_f32(const float *g, const float *ref, void *out1, float *hdr1, void *out2, float *hdr2,
                                           float mul2, const float *amax_g, float *gb, float *gnw, const float *noise,
                                           int64_t rows, int C, int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw,
                                           float alpha, float scale, float *partials, int accumulate, void *stream) {
    return rick_bias_act_bwd_split2_f32(g, ref, out1, hdr1, out2, hdr2, mul2, amax_g, nullptr, nullptr, nullptr, gb, gnw, noise, rows,
                                        C, rows_per_img, noise_nb, noise_hw, alpha, scale, partials, accumulate, stream);
}

// ... with image 1 scaled per (image, channel) by chan_scale[n, c] — then bounded by the amax word `bound1` alone (the caller
// combines |scale| * max |chan_scale| * max |g| on the device: rick_bound_tail_f32) — and the unscaled adjoint optionally
// written as fp32 too (out_f32): the generator's layers, whose demodulation gradient still reads the fp32 adjoint.
extern "C" int rick_bias_act_bwd_split2_f32(const float *g, const float *ref, void *out1, float *hdr1, void *out2, float *hdr2,
                                            float mul2, const float *amax_g, const float *chan_scale, const float *bound1,
                                            float *out_f32, float *gb, float *gnw, const float *noise,
                                            int64_t rows, int C, int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw,
                                            float alpha, float scale, float *partials, int accumulate, void *stream) {
    if (!out1 || !hdr1 || (!amax_g && !bound1) || (C & 3) || (out2 && (!hdr2 || mul2 == 0.f || !amax_g)) || scale == 0.f) return RICK_EINVAL;
    if (((uintptr_t)out1 | (uintptr_t)(out2 ? out2 : out1) | (uintptr_t)g | (uintptr_t)ref | (uintptr_t)(out_f32 ? out_f32 : (float *)out1)) % 16)
        return RICK_EINVAL;
    if (chan_scale && (!bound1 || rows_per_img <= 0 || ((uintptr_t)chan_scale % 16))) return RICK_EINVAL;
    const float slope = fabsf(alpha) > 1.f ? fabsf(alpha) : 1.f;
    const BabSplit sp = {(unsigned char *)out1, (unsigned char *)out2, (cv_split_hdr *)hdr1, (cv_split_hdr *)hdr2, amax_g,
                         bound1 ? 1.f : fabsf(scale) * slope, fabsf(mul2), mul2, chan_scale,
                         rows_per_img > 0 ? rows_per_img : 1, out_f32, bound1};
    return bias_act_bwd_run(g, ref, (float *)out1, gb, gnw, noise, rows, C, rows_per_img, noise_nb, noise_hw, alpha, scale,
                            partials, accumulate, stream, &sp);
}

static int bias_act_bwd_run(const float *g, const float *ref, float *gx, float *gb, float *gnw,
                            const float *noise, int64_t rows, int C, int64_t rows_per_img,
                            int64_t noise_nb, int64_t noise_hw, float alpha, float scale,
                            float *partials, int accumulate, void *stream, const BabSplit *sp, const BabDot *dotp) {
    if (!g || !ref || !gx || rows <= 0 || C <= 0 || ((gb || gnw) && !partials)) return RICK_EINVAL;
    if (gnw && !noise) return RICK_EINVAL;
    if (noise && (rows_per_img <= 0 || noise_nb <= 0 || noise_hw != rows_per_img)) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nb = rick_bias_act_bwd_blocks(rows, C);
    const bool vec = (C % 4 == 0) && (((uintptr_t)g | (uintptr_t)ref | (uintptr_t)gx) % 16 == 0);
    const size_t lds = (256 * 4 + 8) * sizeof(float);
    const BabDot nodot = {nullptr, nullptr, nullptr, 1.f, 1.f};
    const BabDot dot = dotp ? *dotp : nodot;
    if (dot.dpart && (sp || !vec)) return RICK_EINVAL;
    const float *nz = (gnw || (dot.dpart && dot.noise_w)) ? noise : nullptr;     // (the noise values: for gnw and for the reconstruction)
    const BabSplit none = {nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, 1.f, 1.f, nullptr, 1, nullptr, nullptr};
    if (sp && vec)
        hipLaunchKernelGGL((bias_act_bwd_kernel<true, true>), dim3(nb), dim3(256), lds, st, g, ref, gx, nz, partials, rows, C,
                           rows_per_img, noise_nb, noise_hw, alpha, scale, gb ? 1 : 0, *sp, nodot);
    else if (sp)
        return RICK_EINVAL;
    else if (vec)
        hipLaunchKernelGGL((bias_act_bwd_kernel<true, false>), dim3(nb), dim3(256), lds, st, g, ref, gx, nz, partials, rows, C,
                           rows_per_img, noise_nb, noise_hw, alpha, scale, gb ? 1 : 0, none, dot);
    else
        hipLaunchKernelGGL((bias_act_bwd_kernel<false, false>), dim3(nb), dim3(256), lds, st, g, ref, gx, nz, partials, rows, C,
                           rows_per_img, noise_nb, noise_hw, alpha, scale, gb ? 1 : 0, none, nodot);
    // one second-stage launch for both parameter gradients; accumulate: gb / gnw are the parameters' .grad (gradient sink)
    if (accumulate & 2) RICK_LAUNCH_STATUS();      // the caller sums the partial rows later (rick_colsum_multi_f32)
    if (gb && gnw) launch_colsum(partials, gb, nb, C + 1, C + 1, 0, st, nullptr, accumulate & 1, gnw, C);
    else if (gb) launch_colsum(partials, gb, nb, C + 1, C, 0, st, nullptr, accumulate & 1);
    else if (gnw) launch_colsum(partials, gnw, nb, C + 1, 1, C, st, nullptr, accumulate & 1);
    RICK_LAUNCH_STATUS();
}

// --------------------------------------------------------------------- modulation helpers
// y[n,p,c] = x[n,p,c] * s[n,c]
__global__ __launch_bounds__(256) void chan_scale_kernel(const float *__restrict__ x, const float *__restrict__ s,
                                                         float *__restrict__ y, int64_t PC, int C, int64_t total4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += stride) {
        const int64_t i = i4 << 2;
        const int64_t n = i / PC;
        const int c = (int)(i % C);
        const float4 xv = reinterpret_cast<const float4 *>(x)[i4];
        const float4 sv = *reinterpret_cast<const float4 *>(s + n * C + c);
        reinterpret_cast<float4 *>(y)[i4] = make_float4(xv.x * sv.x, xv.y * sv.y, xv.z * sv.z, xv.w * sv.w);
    }
}
__global__ __launch_bounds__(256) void chan_scale_scalar_kernel(const float *__restrict__ x, const float *__restrict__ s,
                                                                float *__restrict__ y, int64_t PC, int C, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
        y[i] = x[i] * s[(i / PC) * C + i % C];
}

extern "C" int rick_chan_scale_f32(const float *x, const float *s, float *y, int N, int64_t P, int C, void *stream) {
    if (!x || !s || !y || N <= 0 || P <= 0 || C <= 0) return RICK_EINVAL;
    const int64_t total = (int64_t)N * P * C;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)s | (uintptr_t)y) % 16 == 0))
        hipLaunchKernelGGL(chan_scale_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, st, x, s, y, P * C, C, total / 4);
    else
        hipLaunchKernelGGL(chan_scale_scalar_kernel, dim3(ew_grid(total)), dim3(256), 0, st, x, s, y, P * C, C, total);
    RICK_LAUNCH_STATUS();
}

// d[n,c] = sum_p a[n,p,c]*b[n,p,c].  grid (blocks_p, N); partials [blk][n][c]
#define HWDOT_ROWS 16
extern "C" int rick_hw_dot_blocks(int64_t P) {
    int64_t nb = cdiv64(P, HWDOT_ROWS);
    if (nb > 512) nb = 512;
    return (int)(nb < 1 ? 1 : nb);
}

// ACT: b holds the OUTPUT y of a fused conv + noise + bias + LeakyReLU tail; the factor is the pre-tail value
// reconstructed on the fly,  lrelu^-1(y) - noise_w * noise[n,p] - bias[c]  (demodulation gradient of the fused
// StyledConv: sum_p gz * conv_out without keeping conv_out).
struct HwDotAct {
    const float *bias, *noise, *noise_w;
    int noise_nb;
    float inv_gain, inv_gain_slope;
    const float *scale;   // SCALE variant: [N][C] factor of the extra output
    float *scaled;        // SCALE variant: scaled[n,p,c] = a[n,p,c] * scale[n,c] written in the same pass
};

__device__ __forceinline__ float4 hwdot_unact(float4 y, float4 bias, float nv, const HwDotAct &t) {
    float4 r;
    r.x = (y.x > 0.f ? y.x * t.inv_gain : y.x * t.inv_gain_slope) - nv - bias.x;
    r.y = (y.y > 0.f ? y.y * t.inv_gain : y.y * t.inv_gain_slope) - nv - bias.y;
    r.z = (y.z > 0.f ? y.z * t.inv_gain : y.z * t.inv_gain_slope) - nv - bias.z;
    r.w = (y.w > 0.f ? y.w * t.inv_gain : y.w * t.inv_gain_slope) - nv - bias.w;
    return r;
}

template <bool VEC4, bool ACT, bool SCALE = false>
__global__ __launch_bounds__(256) void hw_dot_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                     float *__restrict__ partials, int64_t P, int C, HwDotAct t) {
    extern __shared__ float lds[];   // [256 * 4]
    constexpr int W = VEC4 ? 4 : 1;
    const int n = blockIdx.y, N = gridDim.y, nb = gridDim.x;
    const int64_t ppb = cdiv64(P, nb);
    const int64_t p0 = (int64_t)blockIdx.x * ppb, p1 = p0 + ppb < P ? p0 + ppb : P;
    const float *an = a + (int64_t)n * P * C, *bn = b + (int64_t)n * P * C;
    float *pb = partials + ((int64_t)blockIdx.x * N + n) * C;
    const int ncol = C / W;
    for (int cbase = 0; cbase < ncol; cbase += 256) {
        const int cg = ncol - cbase < 256 ? ncol - cbase : 256;
        const int rpb = 256 / cg;
        const int lane_c = threadIdx.x % cg, lane_r = threadIdx.x / cg;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane_r < rpb) {
            int64_t p = p0 + lane_r;
            const int64_t step = (int64_t)rpb * C;
            const float *ap = an + p * C + (int64_t)(cbase + lane_c) * W, *bp = bn + p * C + (int64_t)(cbase + lane_c) * W;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float nwv = 0.f;
            const float *nz = nullptr;
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
            float *cp = nullptr;
            if (SCALE) {
                sc4 = *reinterpret_cast<const float4 *>(t.scale + (int64_t)n * C + (cbase + lane_c) * 4);
                cp = t.scaled + (int64_t)n * P * C + p * C + (int64_t)(cbase + lane_c) * 4;
            }
            if (ACT) {
                if (t.bias) bias4 = *reinterpret_cast<const float4 *>(t.bias + (cbase + lane_c) * 4);
                if (t.noise) {
                    nwv = t.noise_w[0];
                    nz = t.noise + (int64_t)(t.noise_nb == 1 ? 0 : n) * P;
                }
            }
            if (VEC4)
                for (; p + 3 * rpb < p1; p += 4 * rpb) {   // 8 independent 16-byte loads in flight per thread
                    float4 av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        av[u] = *reinterpret_cast<const float4 *>(ap + u * step);
                        bv[u] = *reinterpret_cast<const float4 *>(bp + u * step);
                    }
                    if (ACT) {
#pragma unroll
                        for (int u = 0; u < 4; u++) bv[u] = hwdot_unact(bv[u], bias4, nz ? nwv * nz[p + u * rpb] : 0.f, t);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        acc[0] += av[u].x * bv[u].x; acc[1] += av[u].y * bv[u].y;
                        acc[2] += av[u].z * bv[u].z; acc[3] += av[u].w * bv[u].w;
                        if (SCALE)
                            *reinterpret_cast<float4 *>(cp + u * step) =
                                make_float4(av[u].x * sc4.x, av[u].y * sc4.y, av[u].z * sc4.z, av[u].w * sc4.w);
                    }
                    ap += 4 * step;
                    bp += 4 * step;
                    if (SCALE) cp += 4 * step;
                }
            for (; p < p1; p += rpb) {
                if (VEC4) {
                    const float4 av = *reinterpret_cast<const float4 *>(ap);
                    float4 bv = *reinterpret_cast<const float4 *>(bp);
                    if (ACT) bv = hwdot_unact(bv, bias4, nz ? nwv * nz[p] : 0.f, t);
                    acc[0] += av.x * bv.x; acc[1] += av.y * bv.y; acc[2] += av.z * bv.z; acc[3] += av.w * bv.w;
                    if (SCALE) {
                        *reinterpret_cast<float4 *>(cp) = make_float4(av.x * sc4.x, av.y * sc4.y, av.z * sc4.z, av.w * sc4.w);
                        cp += step;
                    }
                } else {
                    acc[0] += ap[0] * bp[0];
                }
                ap += step;
                bp += step;
            }
        }
        __syncthreads();
        for (int j = 0; j < W; j++) lds[threadIdx.x * W + j] = lane_r < rpb ? acc[j] : 0.f;
        __syncthreads();
        if (threadIdx.x < cg)
            for (int j = 0; j < W; j++) {
                float s = 0.f;
                for (int rr = 0; rr < rpb; rr++) s += lds[(rr * cg + threadIdx.x) * W + j];
                pb[(cbase + threadIdx.x) * W + j] = s;
            }
    }
}

extern "C" int rick_hw_dot_f32(const float *a, const float *b, float *d, int N, int64_t P, int C,
                               float *partials, const float *divisor, void *stream) {
    if (!a || !b || !d || !partials || N <= 0 || P <= 0 || C <= 0 || N > 65535) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nb = rick_hw_dot_blocks(P);
    const HwDotAct none = {nullptr, nullptr, nullptr, 1, 1.f, 1.f, nullptr, nullptr};
    if (C % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) % 16 == 0))
        hipLaunchKernelGGL((hw_dot_kernel<true, false>), dim3(nb, N), dim3(256), 1024 * sizeof(float), st, a, b, partials, P, C, none);
    else
        hipLaunchKernelGGL((hw_dot_kernel<false, false>), dim3(nb, N), dim3(256), 1024 * sizeof(float), st, a, b, partials, P, C, none);
    launch_colsum(partials, d, nb, N * C, N * C, 0, st, divisor);
    RICK_LAUNCH_STATUS();
}

// d[n,c] = sum_p a*b  and  scaled[n,p,c] = a[n,p,c] * scale[n,c] in ONE pass over a (modulated-conv backward: the style
// gradient and the style-scaled data gradient both come from the unscaled data gradient).
extern "C" int rick_hw_dot_scale_f32(const float *a, const float *b, float *d, const float *scale, float *scaled, int N,
                                     int64_t P, int C, float *partials, void *stream) {
    if (!a || !b || !d || !scale || !scaled || !partials || N <= 0 || P <= 0 || C <= 0 || N > 65535 || (C & 3)) return RICK_EINVAL;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)scale | (uintptr_t)scaled) % 16) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nb = rick_hw_dot_blocks(P);
    const HwDotAct t = {nullptr, nullptr, nullptr, 1, 1.f, 1.f, scale, scaled};
    hipLaunchKernelGGL((hw_dot_kernel<true, false, true>), dim3(nb, N), dim3(256), 1024 * sizeof(float), st, a, b, partials, P, C, t);
    launch_colsum(partials, d, nb, N * C, N * C, 0, st);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_hw_dot_act_f32(const float *g, const float *y, float *d, int N, int64_t P, int C, const float *bias,
                                   const float *noise, const float *noise_w, int noise_nb, float slope, float gain,
                                   float *partials, const float *divisor, void *stream) {
    if (!g || !y || !d || !partials || N <= 0 || P <= 0 || C <= 0 || N > 65535 || (C & 3) || gain == 0.f || slope == 0.f)
        return RICK_EINVAL;
    if ((((uintptr_t)g | (uintptr_t)y | (uintptr_t)(bias ? bias : g)) % 16) || (noise && (!noise_w || (noise_nb != 1 && noise_nb != N))))
        return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nb = rick_hw_dot_blocks(P);
    const HwDotAct t = {bias, noise, noise_w, noise_nb, 1.f / gain, 1.f / (gain * slope), nullptr, nullptr};
    hipLaunchKernelGGL((hw_dot_kernel<true, true>), dim3(nb, N), dim3(256), 1024 * sizeof(float), st, g, y, partials, P, C, t);
    launch_colsum(partials, d, nb, N * C, N * C, 0, st, divisor);
    RICK_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void add_scale_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        float *__restrict__ y, int64_t n, float alpha) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n4 = n >> 2;
    for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += stride) {
        float4 av = reinterpret_cast<const float4 *>(a)[i4];
        if (b) {
            const float4 bv = reinterpret_cast<const float4 *>(b)[i4];
            av.x += bv.x; av.y += bv.y; av.z += bv.z; av.w += bv.w;
        }
        reinterpret_cast<float4 *>(y)[i4] = make_float4(av.x * alpha, av.y * alpha, av.z * alpha, av.w * alpha);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        y[i] = (a[i] + (b ? b[i] : 0.f)) * alpha;
}

// y = (a + b) * alpha as fp32 AND as a split image (the ResBlock merge feeds the next block's convolutions), bound
// |alpha| * (*amax_a + *amax_b).  Rows of C channels, C % 32 == 0.
__global__ __launch_bounds__(256) void add_scale_split_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                              float *__restrict__ y, unsigned char *__restrict__ sy,
                                                              cv_split_hdr *__restrict__ hdr, const float *__restrict__ ba,
                                                              const float *__restrict__ bb, float coef, int64_t n4, float alpha) {
    const cv_split_hdr h = cv_split_header(ba, bb, coef);
    if (blockIdx.x == 0 && threadIdx.x == 0) *hdr = h;
    const float sc = cv_uniform(h.scale);
    float am = 0.f;
    for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += (int64_t)gridDim.x * 256) {
        float4 av = reinterpret_cast<const float4 *>(a)[i4];
        const float4 bv = reinterpret_cast<const float4 *>(b)[i4];
        av.x += bv.x; av.y += bv.y; av.z += bv.z; av.w += bv.w;
        const float4 o = make_float4(av.x * alpha, av.y * alpha, av.z * alpha, av.w * alpha);
        reinterpret_cast<float4 *>(y)[i4] = o;
        cv_split_store4(sy + i4 * 16, 0, o, sc);
        am = amax4(am, o);
    }
    cv_sat_check(am, sc);
}

extern "C" int rick_add_scale_split_f32(const float *a, const float *b, float *y, void *y_split, float *hdr,
                                        const float *amax_a, const float *amax_b, int64_t rows, int C, float alpha, void *stream) {
    if (!a || !b || !y || !y_split || !hdr || !amax_a || !amax_b || rows < 0 || C <= 0 || (C & 3) || alpha == 0.f) return RICK_EINVAL;
    if (rows == 0) return 0;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y | (uintptr_t)y_split | (uintptr_t)hdr) % 16 != 0) return RICK_EINVAL;
    const int64_t n4 = rows * (C / 4);
    hipLaunchKernelGGL(add_scale_split_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, a, b, y,
                       (unsigned char *)y_split, (cv_split_hdr *)hdr, amax_a, amax_b, fabsf(alpha), n4, alpha);
    RICK_LAUNCH_STATUS();
}

CV_DEFINE_SAT_ACCESSOR(rick_sat_elementwise)

extern "C" int rick_add_scale_f32(const float *a, const float *b, float *y, int64_t n, float alpha, void *stream) {
    if (!a || !y || n < 0) return RICK_EINVAL;
    if (n == 0) return 0;
    if (((uintptr_t)a | (uintptr_t)y | (uintptr_t)(b ? b : a)) % 16 != 0) return RICK_EINVAL;
    hipLaunchKernelGGL(add_scale_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, alpha);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------- minibatch stddev
// x[B, P, C] (NHWC, P = H*W), group = B (model_probe_tune.py:748-756 with group == batch):
//   sd[p,c] = sqrt(mean_b (x - mean_b x)^2 + 1e-8);  stat = mean_{p,c} sd
//   out[b,p,0:C] = x[b,p,:], out[b,p,C] = stat.
// One 1024-thread block per statistics group (the tensor is B x 16 x 512: latency-bound, so the work is spread over
// 16 wavefronts and the groups over blocks); wavefront-shuffle + fixed-order LDS reduction (deterministic).
#define MBSTD_THREADS 1024
__device__ __forceinline__ float block_sum_1024(float v, float *red /*[17]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < MBSTD_THREADS / 64; w++) t += red[w];
        red[16] = t;
    }
    __syncthreads();
    return red[16];
}

__global__ __launch_bounds__(MBSTD_THREADS) void mbstd_fwd_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                                  float *__restrict__ stat, int B, int P, int C, int groups) {
    __shared__ float red[17];
    const int PC = P * C, C1 = C + 1, gs = B / groups;
    const int gi = blockIdx.x;
    const float *xg = x + (int64_t)gi * gs * PC;
    float local = 0.f;
    for (int i = threadIdx.x; i < PC; i += MBSTD_THREADS) {
        float mean = 0.f;
        for (int b = 0; b < gs; b++) mean += xg[(int64_t)b * PC + i];
        mean /= gs;
        float var = 0.f;
        for (int b = 0; b < gs; b++) {
            const float d = xg[(int64_t)b * PC + i] - mean;
            var += d * d;
        }
        local += sqrtf(var / gs + 1e-8f);
    }
    const float tot = block_sum_1024(local, red) / PC;
    if (threadIdx.x == 0) stat[gi] = tot;
    float *og = out + (int64_t)gi * gs * P * C1;
    for (int i = threadIdx.x; i < gs * P * C1; i += MBSTD_THREADS) {
        const int c = i % C1;
        const int bp = i / C1;
        og[i] = c < C ? xg[(int64_t)bp * C + c] : tot;
    }
}

// gx[b,p,c] = gout[b,p,c] + G/(P*C) * (x[b,p,c]-mean[p,c]) / (gs * sd[p,c]),  G = sum_{b,p in group} gout[b,p,C]
__global__ __launch_bounds__(MBSTD_THREADS) void mbstd_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gout,
                                                                  float *__restrict__ gx, int B, int P, int C, int groups) {
    __shared__ float red[17];
    const int PC = P * C, C1 = C + 1, gs = B / groups;
    const int gi = blockIdx.x;
    const float *xg = x + (int64_t)gi * gs * PC;
    const float *gg = gout + (int64_t)gi * gs * P * C1;
    float *gxg = gx + (int64_t)gi * gs * PC;
    float gl = 0.f;
    for (int i = threadIdx.x; i < gs * P; i += MBSTD_THREADS) gl += gg[(int64_t)i * C1 + C];
    const float G = block_sum_1024(gl, red) / PC;
    for (int i = threadIdx.x; i < PC; i += MBSTD_THREADS) {
        const int p = i / C, c = i - p * C;
        float mean = 0.f;
        for (int b = 0; b < gs; b++) mean += xg[(int64_t)b * PC + i];
        mean /= gs;
        float var = 0.f;
        for (int b = 0; b < gs; b++) {
            const float d = xg[(int64_t)b * PC + i] - mean;
            var += d * d;
        }
        const float sd = sqrtf(var / gs + 1e-8f);
        for (int b = 0; b < gs; b++) {
            const float d = xg[(int64_t)b * PC + i] - mean;
            gxg[(int64_t)b * PC + i] = gg[((int64_t)b * P + p) * C1 + c] + G * d / (gs * sd);
        }
    }
}

extern "C" int rick_mbstd_fwd_f32(const float *x, float *out, float *stat, int B, int P, int C, int groups, void *stream) {
    if (!x || !out || !stat || B <= 0 || P <= 0 || C <= 0 || groups <= 0 || B % groups) return RICK_EINVAL;
    hipLaunchKernelGGL(mbstd_fwd_kernel, dim3(groups), dim3(MBSTD_THREADS), 0, (hipStream_t)stream, x, out, stat, B, P, C, groups);
    RICK_LAUNCH_STATUS();
}
extern "C" int rick_mbstd_bwd_f32(const float *x, const float *gout, float *gx, int B, int P, int C, int groups, void *stream) {
    if (!x || !gout || !gx || B <= 0 || P <= 0 || C <= 0 || groups <= 0 || B % groups) return RICK_EINVAL;
    hipLaunchKernelGGL(mbstd_bwd_kernel, dim3(groups), dim3(MBSTD_THREADS), 0, (hipStream_t)stream, x, gout, gx, B, P, C, groups);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------ Fisher / optimiser
__global__ __launch_bounds__(256) void sq_acc_kernel(float *__restrict__ acc, const float *__restrict__ g, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = g[i];
        acc[i] = __builtin_fmaf(v, v, acc[i]);
    }
}
extern "C" int rick_sq_accumulate_f32(float *acc, const float *g, int64_t n, void *stream) {
    if (!acc || !g || n < 0) return RICK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sq_acc_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, acc, g, n);
    RICK_LAUNCH_STATUS();
}

// One wavefront per filter: lanes stride over (outer, inner), shuffle-reduce.
__global__ __launch_bounds__(256) void filter_reduce_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                            int64_t outer, int64_t outer_stride, int64_t nfilters,
                                                            int64_t filter_stride, int64_t inner, float scale) {
    const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nfilters) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int64_t o = 0; o < outer; o++) {
        const float *base = x + o * outer_stride + f * filter_stride;
        for (int64_t i = lane; i < inner; i += 64) s += base[i];
    }
    s = wave_sum(s);
    if (lane == 0) out[f] = s * scale;
}
extern "C" int rick_filter_reduce_f32(const float *x, float *out, int64_t outer, int64_t outer_stride,
                                      int64_t nfilters, int64_t filter_stride, int64_t inner, float scale,
                                      void *stream) {
    if (!x || !out || outer <= 0 || nfilters <= 0 || inner <= 0) return RICK_EINVAL;
    hipLaunchKernelGGL(filter_reduce_kernel, dim3((unsigned)cdiv64(nfilters, 4)), dim3(256), 0, (hipStream_t)stream,
                       x, out, outer, outer_stride, nfilters, filter_stride, inner, scale);
    RICK_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void masked_adam_kernel(float *__restrict__ p, float *__restrict__ g,
                                                          float *__restrict__ m, float *__restrict__ v,
                                                          const uint8_t *__restrict__ mask, int64_t n, float lr,
                                                          float beta1, float beta2, float eps, float bc1, float bc2,
                                                          const float *__restrict__ bc_dev) {
    if (bc_dev) {   // bias corrections kept on the device (rick_adam_prepare_f32): the launch is replayable in a hipGraph
        bc1 = bc_dev[0];
        bc2 = bc_dev[1];
    }
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gv = g[i], pv = p[i];
        if (mask) {
            const uint8_t mk = mask[i];
            if (mk & 2) pv = 0.f;
            if (mk & 3) { gv = 0.f; g[i] = 0.f; }
        }
        // torch.optim.Adam single-tensor: m.lerp_(g, 1-b1); v = b2*v + (1-b2) g^2
        const float mv = m[i] + (gv - m[i]) * (1.f - beta1);
        const float vv = v[i] * beta2 + (1.f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        p[i] = pv - step_size * (mv / denom);
    }
}
extern "C" int rick_masked_adam_f32(float *p, float *g, float *m, float *v, const uint8_t *mask, int64_t n,
                                    float lr, float beta1, float beta2, float eps, float bc1, float bc2,
                                    void *stream) {
    if (!p || !g || !m || !v || n < 0) return RICK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(masked_adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, mask, n,
                       lr, beta1, beta2, eps, bc1, bc2, (const float *)nullptr);
    RICK_LAUNCH_STATUS();
}

// steps[first .. first+count) += 1 and bc = {1 - beta1^t, 1 - beta2^t} with t = the new step count of steps[first]
// (every parameter of the range has the same count).  Keeping the step counter on the device makes an optimiser
// step a pure function of device state, so a captured hipGraph of the whole train step can be replayed.
__global__ void adam_prepare_kernel(int *__restrict__ steps, int first, int count, float beta1, float beta2,
                                    float *__restrict__ bc) {
    const int t = steps[first] + 1;
    __syncthreads();
    for (int j = threadIdx.x; j < count; j += blockDim.x) steps[first + j] += 1;
    if (threadIdx.x == 0) {
        bc[0] = (float)(1.0 - pow((double)beta1, (double)t));
        bc[1] = (float)(1.0 - pow((double)beta2, (double)t));
    }
}
extern "C" int rick_adam_prepare_f32(int *steps, int first, int count, float beta1, float beta2, float *bc, void *stream) {
    if (!steps || !bc || first < 0 || count < 1) return RICK_EINVAL;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, steps, first, count, beta1, beta2, bc);
    RICK_LAUNCH_STATUS();
}
extern "C" int rick_masked_adam_dev_f32(float *p, float *g, float *m, float *v, const uint8_t *mask, int64_t n,
                                        float lr, float beta1, float beta2, float eps, const float *bc, void *stream) {
    if (!p || !g || !m || !v || !bc || n < 0) return RICK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(masked_adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, mask, n,
                       lr, beta1, beta2, eps, 1.f, 1.f, bc);
    RICK_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void ema_kernel(float *__restrict__ e, const float *__restrict__ p, int64_t n, float decay) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float om = 1.f - decay;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) e[i] = e[i] * decay + p[i] * om;
}
extern "C" int rick_ema_f32(float *ema, const float *p, int64_t n, float decay, void *stream) {
    if (!ema || !p || n < 0) return RICK_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(ema_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, ema, p, n, decay);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_abi_version(void) { return 1; }
