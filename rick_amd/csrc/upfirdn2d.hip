// upfirdn2d for gfx950: upsample (zero insertion) -> pad/crop -> 2-D FIR -> downsample.
//
// Semantics follow the reference extension (op/upfirdn2d_kernel.cu:49-105, :209-240):
// input viewed as [major, in_h, in_w, minor]; for output (oy, ox)
//     mid_y = oy*down_y + up_y - 1 - pad_y0,  in_y0 = floor_div(mid_y, up_y),
//     taps  y = 0..h-1 read input row in_y0 + y with kernel row (mid_y + kh - (in_y0+1)*up_y) - y*up_y
// (same along x).  Rows/cols outside the input contribute nothing; here they are staged as
// zeros in LDS, which leaves the fmaf chain bit-identical to skipping them.  Accumulation
// order is y-outer / x-inner with one fmaf per tap, matching oracle/csrc/oracle_ops.c.
//
// Two kernels, both LDS-staged with coalesced HBM reads:
//   planar  (minor == 1, e.g. the RGB skip path): 16x64 output tile per 256-thread block
//   nhwc    (minor % 4 == 0): 64-channel slab x TOHxTOW pixel tile, float4 per lane
#include "conv_common.h"
#include <stdlib.h>

struct UfdParams {
    int in_h, in_w, minor, kh, kw;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    int out_h, out_w;
    int tih, tiw;       // LDS input-tile extents
    int toh, tow;       // output tile
    int tiles_x;
};

// ----------------------------------------------------------------------------- planar
__global__ __launch_bounds__(256) void upfirdn2d_planar_kernel(const float *__restrict__ in,
                                                               const float *__restrict__ kern,
                                                               float *__restrict__ out, UfdParams p) {
    extern __shared__ float smem[];
    float *sk = smem;                       // [kh*kw]
    float *sx = smem + p.kh * p.kw;         // [tih][tiw]
    const int tile_x = blockIdx.x % p.tiles_x, tile_y = blockIdx.x / p.tiles_x;
    const int64_t major = blockIdx.y;
    const int oy0 = tile_y * p.toh, ox0 = tile_x * p.tow;
    const int iy_lo = floor_div_i(oy0 * p.down_y + p.up_y - 1 - p.pad_y0, p.up_y);
    const int ix_lo = floor_div_i(ox0 * p.down_x + p.up_x - 1 - p.pad_x0, p.up_x);

    for (int i = threadIdx.x; i < p.kh * p.kw; i += 256) sk[i] = kern[i];
    const float *src = in + major * (int64_t)p.in_h * p.in_w;
    for (int i = threadIdx.x; i < p.tih * p.tiw; i += 256) {
        const int r = i / p.tiw, c = i - r * p.tiw;
        const int iy = iy_lo + r, ix = ix_lo + c;
        float v = 0.f;
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = src[(int64_t)iy * p.in_w + ix];
        sx[i] = v;
    }
    __syncthreads();

    float *dst = out + major * (int64_t)p.out_h * p.out_w;
    for (int i = threadIdx.x; i < p.toh * p.tow; i += 256) {
        const int ty = i / p.tow, tx = i - ty * p.tow;
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy >= p.out_h || ox >= p.out_w) continue;
        const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int in_y = floor_div_i(mid_y, p.up_y);
        const int h = floor_div_i(mid_y + p.kh, p.up_y) - in_y;
        const int ky0 = mid_y + p.kh - (in_y + 1) * p.up_y;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
        const int in_x = floor_div_i(mid_x, p.up_x);
        const int w = floor_div_i(mid_x + p.kw, p.up_x) - in_x;
        const int kx0 = mid_x + p.kw - (in_x + 1) * p.up_x;
        const float *xr = sx + (in_y - iy_lo) * p.tiw + (in_x - ix_lo);
        float v = 0.f;
        for (int y = 0; y < h; y++) {
            const float *kr = sk + (ky0 - y * p.up_y) * p.kw + kx0;
            for (int x = 0; x < w; x++) v = __builtin_fmaf(xr[x], kr[-x * p.up_x], v);
            xr += p.tiw;
        }
        dst[(int64_t)oy * p.out_w + ox] = v;
    }
}

// ------------------------------------------------------------------------------- NHWC
// blockIdx.x: pixel tile, blockIdx.y: 64-channel slab, blockIdx.z: image.
// XO: extended result handling (rick_split_out): add into `out`, fold max |result| into a word, write a split image.
static const rick_split_out kNoSplitOut = {nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, 0, 0, nullptr, nullptr, 0.f, 1.f, nullptr};

template <bool XO>
__device__ __forceinline__ void ufd_store(float *dst, unsigned char *spix, int c, float4 v, const rick_split_out &xo, float sscale,
                                          float &am, float &ams, int64_t n = 0, int C = 0, int64_t eoff = 0, float4 *bsum = nullptr) {
    if (!XO) {
        *reinterpret_cast<float4 *>(dst) = v;
        return;
    }
    if (xo.adj_ref) {          // fused activation adjoint (same expression as bias_act_bwd_kernel: g * (ref > 0 ? 1 : slope) * gain)
        const float4 r = *reinterpret_cast<const float4 *>(xo.adj_ref + eoff);
        v.x = v.x * (r.x > 0.f ? 1.f : xo.adj_slope) * xo.adj_gain;
        v.y = v.y * (r.y > 0.f ? 1.f : xo.adj_slope) * xo.adj_gain;
        v.z = v.z * (r.z > 0.f ? 1.f : xo.adj_slope) * xo.adj_gain;
        v.w = v.w * (r.w > 0.f ? 1.f : xo.adj_slope) * xo.adj_gain;
        if (bsum) { bsum->x += v.x; bsum->y += v.y; bsum->z += v.z; bsum->w += v.w; }
    }
    if (xo.accumulate) {
        const float4 o = *reinterpret_cast<const float4 *>(dst);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    am = amax4(am, v);
    if (!xo.no_f32) *reinterpret_cast<float4 *>(dst) = v;
    if (xo.split_out) {
        if (xo.chan_scale) {
            const float4 cs = *reinterpret_cast<const float4 *>(xo.chan_scale + n * C + c);
            v = make_float4(v.x * cs.x, v.y * cs.y, v.z * cs.z, v.w * cs.w);
        }
        ams = amax4(ams, v);
        cv_split_store4(spix, c, v, sscale);
    }
}

template <bool XO>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ kern,
                                                             float *__restrict__ out, UfdParams p, int cb4, rick_split_out xo) {
    extern __shared__ float smem[];
    float sscale = 1.f, am = 0.f, ams = 0.f;
    if (XO && xo.split_out) {
        const cv_split_hdr h = cv_split_header(xo.bound0, xo.bound1, xo.bound_coef);
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *reinterpret_cast<cv_split_hdr *>(xo.split_hdr) = h;
        sscale = cv_uniform(h.scale);
    }
    float *sk = smem;                                            // [kh*kw] (padded to x4)
    float4 *sx = reinterpret_cast<float4 *>(smem + ((p.kh * p.kw + 3) & ~3));   // [tih][tiw][cb4]
    const int tile_x = blockIdx.x % p.tiles_x, tile_y = blockIdx.x / p.tiles_x;
    const int c0 = blockIdx.y * cb4 * 4;
    const int64_t n = blockIdx.z;
    const int oy0 = tile_y * p.toh, ox0 = tile_x * p.tow;
    const int iy_lo = floor_div_i(oy0 * p.down_y + p.up_y - 1 - p.pad_y0, p.up_y);
    const int ix_lo = floor_div_i(ox0 * p.down_x + p.up_x - 1 - p.pad_x0, p.up_x);

    for (int i = threadIdx.x; i < p.kh * p.kw; i += 256) sk[i] = kern[i];
    const float *src = in + n * (int64_t)p.in_h * p.in_w * p.minor + c0;
    const int nload = p.tih * p.tiw * cb4;
    for (int i = threadIdx.x; i < nload; i += 256) {
        const int c4 = i % cb4, pix = i / cb4;
        const int r = pix / p.tiw, c = pix - r * p.tiw;
        const int iy = iy_lo + r, ix = ix_lo + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w)
            v = *reinterpret_cast<const float4 *>(src + ((int64_t)iy * p.in_w + ix) * p.minor + c4 * 4);
        sx[i] = v;
    }
    __syncthreads();

    const int nout = p.toh * p.tow * cb4;
    for (int i = threadIdx.x; i < nout; i += 256) {
        const int c4 = i % cb4, pix = i / cb4;
        const int ty = pix / p.tow, tx = pix - ty * p.tow;
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy >= p.out_h || ox >= p.out_w) continue;
        const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int in_y = floor_div_i(mid_y, p.up_y);
        const int h = floor_div_i(mid_y + p.kh, p.up_y) - in_y;
        const int ky0 = mid_y + p.kh - (in_y + 1) * p.up_y;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
        const int in_x = floor_div_i(mid_x, p.up_x);
        const int w = floor_div_i(mid_x + p.kw, p.up_x) - in_x;
        const int kx0 = mid_x + p.kw - (in_x + 1) * p.up_x;
        const float4 *xr = sx + ((in_y - iy_lo) * p.tiw + (in_x - ix_lo)) * cb4 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int y = 0; y < h; y++) {
            const float *kr = sk + (ky0 - y * p.up_y) * p.kw + kx0;
            for (int x = 0; x < w; x++) {
                const float kv = kr[-x * p.up_x];
                const float4 xv = xr[x * cb4];
                v.x = __builtin_fmaf(xv.x, kv, v.x);
                v.y = __builtin_fmaf(xv.y, kv, v.y);
                v.z = __builtin_fmaf(xv.z, kv, v.z);
                v.w = __builtin_fmaf(xv.w, kv, v.w);
            }
            xr += p.tiw * cb4;
        }
        const int64_t po = n * (int64_t)p.out_h * p.out_w + (int64_t)oy * p.out_w + ox;
        ufd_store<XO>(out + po * p.minor + c0 + c4 * 4, (unsigned char *)xo.split_out + po * p.minor * 4, c0 + c4 * 4, v, xo, sscale, am, ams, n, p.minor);
    }
    if (XO) {
        if (xo.amax) cv_amax_publish(am, xo.amax, smem);
        if (xo.split_out) cv_sat_check(ams, sscale);
    }
}

// ------------------------------------------------------------- NHWC, 4x4 FIR, up == 1 (hot path)
// The blur after every transposed conv, the blurs of the discriminator ResBlocks and the skip-path
// decimation are all 4x4 kernels with up == 1 and down in {1, 2} on >= 64 channels.  Compile-time tile,
// tap count and strides: 16 unrolled fmaf per output float4, y-outer / x-inner exactly like the generic
// kernel and the C oracle (bit-identical results).
// CB4: float4 per pixel of the block's channel slab (16: 64 channels, 8 x 8 tile — the default; 8: 32 channels — still whole
// 128-byte lines per pixel — on a 16 x 16 tile, round 6: the 3-pixel halo then costs 19^2 / 16^2 = 1.41 input pixels per output
// instead of 11^2 / 8^2 = 1.89).  MEASURED (tools/r6_ufd_eval.sh, profiles/r06_ufd_tile16.txt): the hypothesis that the L2 -> CU side
// ((halo + 1) x the algorithmic bytes) limits this kernel is wrong — pad-(1,1) blur 55.0 vs 51.6 us at 128 ch @256^2, 24.8 vs 23.4
// at 512 ch @64^2, pad-(2,2) 52.4 vs 55.8 / 17.8 vs 16.3; bench.py 169.4 vs 170.2 images/s (two alternating same-box pairs).  The
// 16 x 16 form stays behind rick_conv_tuning(RICK_TUNE_UFD_TILE16) with its bit-equality test; 8 x 8 x 64 ships.
template <int DOWN, int TOH, int TOW, bool TAIL, bool XO = false, int CB4 = 16>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_k4_kernel(const float *__restrict__ in,
                                                                const float *__restrict__ kern,
                                                                float *__restrict__ out, UfdParams p,
                                                                rick_conv_epilogue tail, rick_split_out xo) {
    constexpr int TIH = (TOH - 1) * DOWN + 4, TIW = (TOW - 1) * DOWN + 4;
    constexpr int PG = 256 / CB4;                 // pixel groups: threads that share a channel quad walk pixels PG apart
    __shared__ float4 sx[TIH * TIW * CB4];
    float sscale = 1.f, am = 0.f, ams = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (XO && xo.split_out) {
        const cv_split_hdr h = cv_split_header(xo.bound0, xo.bound1, xo.bound_coef);
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *reinterpret_cast<cv_split_hdr *>(xo.split_hdr) = h;
        sscale = cv_uniform(h.scale);
    }
    // blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of tiles (row-major), so the
    // halo rows / columns that neighbouring tiles share are re-read from the same L2 (placement only)
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tile_x = tile % p.tiles_x, tile_y = tile / p.tiles_x;
    const int c0 = blockIdx.y * (CB4 * 4);
    const int64_t n = blockIdx.z;
    const int oy0 = tile_y * TOH, ox0 = tile_x * TOW;
    const int iy_lo = oy0 * DOWN - p.pad_y0, ix_lo = ox0 * DOWN - p.pad_x0;
    float kr[16];
#pragma unroll
    for (int i = 0; i < 16; i++) kr[i] = kern[i];
    const float *src = in + n * (int64_t)p.in_h * p.in_w * p.minor + c0;
    const int c4 = threadIdx.x & (CB4 - 1);
    const int pgrp = threadIdx.x / CB4;
    // stage the input tile: all of a thread's 16-byte loads are issued first (clamped addresses, no branch),
    // then written to LDS — one HBM round trip per block instead of one per staged pixel
    constexpr int NLD = (TIH * TIW + PG - 1) / PG;
    float4 stage[NLD];
    unsigned okm = 0;
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int pix = pgrp + PG * k;
        const int r = pix / TIW, c = pix - r * TIW;
        const int iy = iy_lo + r, ix = ix_lo + c;
        const bool ok = pix < TIH * TIW && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        okm |= (ok ? 1u : 0u) << k;
        stage[k] = *reinterpret_cast<const float4 *>(ok ? src + ((int64_t)iy * p.in_w + ix) * p.minor + c4 * 4 : in);
    }
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int pix = pgrp + PG * k;
        if (pix < TIH * TIW) sx[pix * CB4 + c4] = ((okm >> k) & 1u) ? stage[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // Outputs of a thread.  down == 1: four vertically adjacent pixels of one column, so the 7 x 4 input window is read
    // from LDS once (28 ds_read_b128 instead of 64) — every output still accumulates its 16 taps y-outer / x-inner.
    constexpr int NOUT = TOH * TOW / PG;
    constexpr bool COLUMN = DOWN == 1 && (TOH % 4) == 0 && ((TOH / 4) * TOW) % PG == 0;
    constexpr int NCG = COLUMN ? (TOH / 4) * TOW / PG : 1;        // column groups (4 vertically adjacent outputs) per thread
    static_assert(!COLUMN || NOUT == 4 * NCG, "column groups cover the tile");
    float4 acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; j++) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (COLUMN) {
#pragma unroll
        for (int q = 0; q < NCG; q++) {
            const int cg = pgrp + PG * q;
            const float4 *xr = sx + (((cg / TOW) * 4) * TIW + (cg % TOW)) * CB4 + c4;
#pragma unroll
            for (int r = 0; r < 7; r++) {
                float4 xv[4];
#pragma unroll
                for (int x = 0; x < 4; x++) xv[x] = xr[(r * TIW + x) * CB4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int y = r - o;
                    if (y < 0 || y > 3) continue;
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const float kv = kr[(3 - y) * 4 + (3 - x)];
                        acc[q * 4 + o].x = __builtin_fmaf(xv[x].x, kv, acc[q * 4 + o].x);
                        acc[q * 4 + o].y = __builtin_fmaf(xv[x].y, kv, acc[q * 4 + o].y);
                        acc[q * 4 + o].z = __builtin_fmaf(xv[x].z, kv, acc[q * 4 + o].z);
                        acc[q * 4 + o].w = __builtin_fmaf(xv[x].w, kv, acc[q * 4 + o].w);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NOUT; j++) {
            const int pix = pgrp + PG * j;
            const int ty = pix / TOW, tx = pix % TOW;
            const float4 *xr = sx + ((ty * DOWN) * TIW + tx * DOWN) * CB4 + c4;
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const float kv = kr[(3 - y) * 4 + (3 - x)];
                    const float4 xv = xr[(y * TIW + x) * CB4];
                    acc[j].x = __builtin_fmaf(xv.x, kv, acc[j].x);
                    acc[j].y = __builtin_fmaf(xv.y, kv, acc[j].y);
                    acc[j].z = __builtin_fmaf(xv.z, kv, acc[j].z);
                    acc[j].w = __builtin_fmaf(xv.w, kv, acc[j].w);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < NOUT; j++) {
        const int cg = pgrp + PG * (j >> 2);          // (COLUMN: output j = column group j / 4, row j % 4 of it)
        const int ty = COLUMN ? (cg / TOW) * 4 + (j & 3) : (pgrp + PG * j) / TOW;
        const int tx = COLUMN ? (cg % TOW) : (pgrp + PG * j) % TOW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        float4 v = acc[j];
        if (oy < p.out_h && ox < p.out_w) {
            if (TAIL) {   // fused NoiseInjection + bias + LeakyReLU (same operation order as rick_bias_act_f32)
                if (tail.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(tail.bias + c0 + c4 * 4);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (tail.noise) {
                    const float nv = tail.noise_w[0] *
                                     tail.noise[(tail.noise_nb == 1 ? 0 : n) * p.out_h * p.out_w + (int64_t)oy * p.out_w + ox];
                    v.x += nv; v.y += nv; v.z += nv; v.w += nv;
                }
                if (tail.act) {
                    v.x = (v.x > 0.f ? v.x : v.x * tail.slope) * tail.gain;
                    v.y = (v.y > 0.f ? v.y : v.y * tail.slope) * tail.gain;
                    v.z = (v.z > 0.f ? v.z : v.z * tail.slope) * tail.gain;
                    v.w = (v.w > 0.f ? v.w : v.w * tail.slope) * tail.gain;
                }
            }
            const int64_t po = n * (int64_t)p.out_h * p.out_w + (int64_t)oy * p.out_w + ox;
            ufd_store<XO>(out + po * p.minor + c0 + c4 * 4, (unsigned char *)xo.split_out + po * p.minor * 4, c0 + c4 * 4, v, xo, sscale, am, ams, n, p.minor,
                          po * p.minor + c0 + c4 * 4, &bsum);
        }
    }
    if (XO) {
        if (xo.adj_partials) {     // per-channel sums of the block's tile (the bias gradient's first stage): PG pixel rows -> 1
            __syncthreads();
            sx[threadIdx.x] = bsum;                      // [pixel group 0..PG-1][c4 0..CB4-1]
            __syncthreads();
            if (threadIdx.x < CB4) {
                float4 t = sx[threadIdx.x];
#pragma unroll
                for (int r = 1; r < PG; r++) {
                    const float4 u = sx[r * CB4 + threadIdx.x];
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                const int64_t row = n * gridDim.x + tile;
                *reinterpret_cast<float4 *>(xo.adj_partials + row * p.minor + c0 + threadIdx.x * 4) = t;
            }
        }
        if (xo.amax) cv_amax_publish(am, xo.amax, reinterpret_cast<float *>(sx));
        if (xo.split_out) cv_sat_check(ams, sscale);
    }
}

static int tile_in_extent(int tile_out, int down, int k, int up) {
    // rows touched by tile_out consecutive outputs: <= ((tile_out-1)*down + k - 1)/up + 2
    return ((tile_out - 1) * down + k - 1) / up + 2;
}

static int upfirdn2d_impl(const float *input, const float *kernel, float *out, int64_t major, int in_h, int in_w, int minor,
                          int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                          int pad_y1, const rick_conv_epilogue *tail, void *stream, const rick_split_out *xo = nullptr);

// 16 x 16 output tiles for the 4x4 up = 1 down = 1 form (rick_conv_tuning RICK_TUNE_UFD_TILE16; outputs of >= 32 x 32)
static bool ufd_tile16(int out_h, int out_w) { return rick_internal_tune(RICK_TUNE_UFD_TILE16) && out_h >= 32 && out_w >= 32; }

extern "C" int rick_upfirdn2d_f32(const float *input, const float *kernel, float *out,
                                  int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                                  int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream) {
    return upfirdn2d_impl(input, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                          pad_y0, pad_y1, nullptr, stream);
}

extern "C" int rick_upfirdn2d_act_f32(const float *input, const float *kernel, float *out,
                                      int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                                      int up_x, int up_y, int down_x, int down_y,
                                      int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                      const rick_conv_epilogue *tail, void *stream) {
    if (!tail) return RICK_EINVAL;
    return upfirdn2d_impl(input, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                          pad_y0, pad_y1, tail, stream);
}

// The same FIR with the extended result handling (split image / accumulate / running maximum): channels-last, minor % 64 == 0
// for the 4x4 up = 1 form, minor % 4 == 0 for the generic one (split images need minor % 32 == 0).
extern "C" int rick_upfirdn2d_ex_f32(const float *input, const float *kernel, float *out,
                                     int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                                     int up_x, int up_y, int down_x, int down_y,
                                     int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                     const rick_conv_epilogue *tail, const rick_split_out *ex, void *stream) {
    if (!ex || (minor & 3)) return RICK_EINVAL;
    if (ex->split_out && (!ex->split_hdr || !ex->bound0 || (minor & 31) || !(ex->bound_coef > 0.f) || ((uintptr_t)ex->split_out % 16)))
        return RICK_EINVAL;
    if (!out && !(ex->no_f32 && !ex->accumulate)) return RICK_EINVAL;
    if (ex->adj_ref && !(kh == 4 && kw == 4 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && minor % 64 == 0 && !tail &&
                         ((uintptr_t)ex->adj_ref % 16) == 0 && (!ex->adj_partials || ((uintptr_t)ex->adj_partials % 16) == 0)))
        return RICK_EINVAL;
    return upfirdn2d_impl(input, kernel, out ? out : (float *)ex->split_out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x,
                          down_y, pad_x0, pad_x1, pad_y0, pad_y1, tail, stream, ex);
}

static int upfirdn2d_impl(const float *input, const float *kernel, float *out, int64_t major, int in_h, int in_w, int minor,
                          int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                          int pad_y1, const rick_conv_epilogue *tail, void *stream, const rick_split_out *xo) {
    if (!input || !kernel || !out || major <= 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 ||
        kw <= 0 || up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
        return RICK_EINVAL;
    UfdParams p;
    p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y;
    p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (p.out_h <= 0 || p.out_w <= 0) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (minor % 64 == 0 && kh == 4 && kw == 4 && up_x == 1 && up_y == 1 && down_x == down_y && (down_x == 1 || down_x == 2) &&
        major <= 65535 && minor / 64 <= 65535 && (((uintptr_t)input | (uintptr_t)out) % 16 == 0)) {
        const rick_conv_epilogue none = {nullptr, nullptr, nullptr, 1, 0, 0.f, 1.f, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr};
        if (tail) {
            if (tail->noise && (!tail->noise_w || (tail->noise_nb != 1 && tail->noise_nb != major))) return RICK_EINVAL;
            if (tail->bias && ((uintptr_t)tail->bias % 16)) return RICK_EINVAL;
        }
        if (down_x == 1 && ufd_tile16(p.out_h, p.out_w)) {        // 16 x 16 tile on 32-channel slabs (see the kernel's comment)
            if (2 * (int64_t)(minor / 64) > 65535) return RICK_EINVAL;
            p.tiles_x = cdiv(p.out_w, 16);
            dim3 grid(p.tiles_x * cdiv(p.out_h, 16), minor / 32, (unsigned)major);
            if (xo && tail) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 16, 16, true, true, 8>), grid, dim3(256), 0, st, input, kernel, out, p, *tail, *xo);
            else if (xo) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 16, 16, false, true, 8>), grid, dim3(256), 0, st, input, kernel, out, p, none, *xo);
            else if (tail) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 16, 16, true, false, 8>), grid, dim3(256), 0, st, input, kernel, out, p, *tail, kNoSplitOut);
            else hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 16, 16, false, false, 8>), grid, dim3(256), 0, st, input, kernel, out, p, none, kNoSplitOut);
        } else if (down_x == 1) {
            p.tiles_x = cdiv(p.out_w, 8);
            dim3 grid(p.tiles_x * cdiv(p.out_h, 8), minor / 64, (unsigned)major);
            if (xo && tail) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 8, 8, true, true>), grid, dim3(256), 0, st, input, kernel, out, p, *tail, *xo);
            else if (xo) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 8, 8, false, true>), grid, dim3(256), 0, st, input, kernel, out, p, none, *xo);
            else if (tail) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 8, 8, true>), grid, dim3(256), 0, st, input, kernel, out, p, *tail, kNoSplitOut);
            else hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<1, 8, 8, false>), grid, dim3(256), 0, st, input, kernel, out, p, none, kNoSplitOut);
        } else {
            p.tiles_x = cdiv(p.out_w, 8);
            dim3 grid(p.tiles_x * cdiv(p.out_h, 4), minor / 64, (unsigned)major);
            if (xo && tail) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<2, 4, 8, true, true>), grid, dim3(256), 0, st, input, kernel, out, p, *tail, *xo);
            else if (xo) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<2, 4, 8, false, true>), grid, dim3(256), 0, st, input, kernel, out, p, none, *xo);
            else if (tail) hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<2, 4, 8, true>), grid, dim3(256), 0, st, input, kernel, out, p, *tail, kNoSplitOut);
            else hipLaunchKernelGGL((upfirdn2d_nhwc_k4_kernel<2, 4, 8, false>), grid, dim3(256), 0, st, input, kernel, out, p, none, kNoSplitOut);
        }
        RICK_LAUNCH_STATUS();
    }
    if (tail) return RICK_EINVAL;   // the fused tail exists on the channels-last 4x4 path only
    if (minor % 4 == 0) {
        const int cb4 = (minor >= 64 ? 64 : minor) / 4;
        if (minor % (cb4 * 4) != 0) goto planar_like;    // e.g. minor = 72: fall through to generic
        int toh = 8, tow = 8;
        for (;;) {
            p.toh = toh; p.tow = tow;
            p.tih = tile_in_extent(toh, down_y, kh, up_y);
            p.tiw = tile_in_extent(tow, down_x, kw, up_x);
            size_t lds = (size_t)((kh * kw + 3) & ~3) * 4 + (size_t)p.tih * p.tiw * cb4 * 16;
            if (lds <= 60 * 1024 || (toh == 1 && tow == 1)) {
                if (lds > 64 * 1024) return RICK_EINVAL;
                p.tiles_x = cdiv(p.out_w, tow);
                if (major > 65535 || minor / (cb4 * 4) > 65535) return RICK_EINVAL;
                dim3 grid(p.tiles_x * cdiv(p.out_h, toh), minor / (cb4 * 4), (unsigned)major);
                if (xo) hipLaunchKernelGGL(upfirdn2d_nhwc_kernel<true>, grid, dim3(256), lds, st, input, kernel, out, p, cb4, *xo);
                else hipLaunchKernelGGL(upfirdn2d_nhwc_kernel<false>, grid, dim3(256), lds, st, input, kernel, out, p, cb4, kNoSplitOut);
                RICK_LAUNCH_STATUS();
            }
            if (toh >= tow && toh > 1) toh >>= 1; else tow >>= 1;
        }
    }
planar_like:
    if (xo) return RICK_EINVAL;           // the extended result handling exists on the channels-last kernels only
    if (minor != 1) return RICK_EINVAL;   // callers re-layout to planar or channels-last x4
    {
        int toh = 16, tow = 64;
        for (;;) {
            p.toh = toh; p.tow = tow;
            p.tih = tile_in_extent(toh, down_y, kh, up_y);
            p.tiw = tile_in_extent(tow, down_x, kw, up_x);
            size_t lds = (size_t)(kh * kw + p.tih * p.tiw) * 4;
            if (lds <= 60 * 1024 || (toh == 1 && tow == 1)) {
                if (lds > 64 * 1024) return RICK_EINVAL;
                p.tiles_x = cdiv(p.out_w, tow);
                // grid.y carries `major` (N*C); split very large majors over grid.z-free loop
                if (major > 65535 * 1LL) {
                    // launch in slabs of 65535 planes
                    for (int64_t m0 = 0; m0 < major; m0 += 65535) {
                        int64_t mm = major - m0 < 65535 ? major - m0 : 65535;
                        dim3 grid(p.tiles_x * cdiv(p.out_h, toh), (unsigned)mm);
                        hipLaunchKernelGGL(upfirdn2d_planar_kernel, grid, dim3(256), lds, st,
                                           input + m0 * (int64_t)in_h * in_w, kernel,
                                           out + m0 * (int64_t)p.out_h * p.out_w, p);
                    }
                    RICK_LAUNCH_STATUS();
                }
                dim3 grid(p.tiles_x * cdiv(p.out_h, toh), (unsigned)major);
                hipLaunchKernelGGL(upfirdn2d_planar_kernel, grid, dim3(256), lds, st, input, kernel, out, p);
                RICK_LAUNCH_STATUS();
            }
            if (toh >= tow && toh > 1) toh >>= 1; else tow >>= 1;
        }
    }
}

CV_DEFINE_SAT_ACCESSOR(rick_sat_upfirdn2d)

extern "C" int64_t rick_upfirdn2d_adjoint_rows(int64_t major, int out_h, int out_w) {
    const int tile = ufd_tile16(out_h, out_w) ? 16 : 8;
    return major * cdiv(out_w, tile) * cdiv(out_h, tile);
}
