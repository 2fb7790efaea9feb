// Implicit-GEMM convolution family on gfx950 MFMA (v_mfma_f32_16x16x32_bf16).
//
// One kernel covers every convolution of the StyleGAN2 G/D hot path and its data gradient
// (model_probe_tune.py:122,265,274,280): the host describes a launch as a grid of output
// "positions" plus a tap table (rick_conv_geom, include/rick_hip.h).
//
// GEMM view per block:  D[128 co][128 positions] += W[128 co][K] * X[K][128 positions],
// K = (32-channel chunk, tap).  fp32 activations are read from HBM ONCE per block and chunk
// as a spatial patch (tile + halo), converted to bf16 hi/lo on the way into LDS and re-used
// by all taps (9x re-use for 3x3), so neither im2col traffic nor the fp32->bf16x2 split is
// paid per tap.  Weights are pre-packed (rick_conv_pack_weight) into the exact swizzled LDS
// image of the A operand, so staging them is a linear 16-byte copy.
//
// Precision: split=2 multiplies hi*hi + hi*lo + lo*hi with fp32 accumulation (~2^-16
// relative per product, fp32-grade for the 1e-3 parity bar); split=1 is plain bf16.
//
// Wave tiling: 256 threads = 4 waves in 2(co) x 2(pos); each wave owns 64 co x 64 positions
// = 4x4 MFMA tiles (16 accumulators of 4 VGPRs).  Output rows (co) land 4-consecutive per
// lane, so NHWC stores are float4.
//
// LDS bank conflicts: both operands are [row][32 k] bf16 (64 B rows) read as ds_read_b128 per
// lane (row = lane&15, k-group = lane>>4).  16-byte slot index is XOR-swizzled with bit 2 of
// the row: slot = kg ^ (((row>>2)&1)<<1)  -> conflict-free for 16 consecutive rows at any
// offset (checked by simulation against the gfx950 ds_read_b128 lane groups).
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CV_BM 128
#define CV_BN 128
#define CV_CK 32
#define CV_WTILE_BYTES (CV_BM * CV_CK * 2)      // one of hi / lo: 8 KB
#define CV_WSTEP_BYTES (2 * CV_WTILE_BYTES)     // hi + lo: 16 KB

__host__ __device__ __forceinline__ int cv_swz(int kg, int row) { return kg ^ (((row >> 2) & 1) << 1); }

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {   // v_cvt_pk_bf16_f32
    bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return *reinterpret_cast<unsigned *>(&r);
}

// x = hi + lo with hi = x truncated to bf16 (exactly representable, so lo = x - hi is exact in
// fp32) and lo rounded to nearest bf16: |x - hi - lo| <= 2^-17 |x|, unbiased.  10 VALU ops per 4
// elements (v_and, v_perm, v_pk_add, v_cvt_pk).  SPLIT == 1 (plain bf16): hi is rounded to nearest.
template <int SPLIT>
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo) {
    if (SPLIT == 1) {
        hi.x = pack_bf16_rne(v.x, v.y);
        hi.y = pack_bf16_rne(v.z, v.w);
        lo.x = lo.y = 0;
        return;
    }
    const unsigned ux = __float_as_uint(v.x), uy = __float_as_uint(v.y), uz = __float_as_uint(v.z),
                   uw = __float_as_uint(v.w);
    hi.x = __builtin_amdgcn_perm(uy, ux, 0x07060302);
    hi.y = __builtin_amdgcn_perm(uw, uz, 0x07060302);
    lo.x = pack_bf16_rne(v.x - __uint_as_float(ux & 0xffff0000u), v.y - __uint_as_float(uy & 0xffff0000u));
    lo.y = pack_bf16_rne(v.z - __uint_as_float(uz & 0xffff0000u), v.w - __uint_as_float(uw & 0xffff0000u));
}

// ------------------------------------------------------------------------------------------
// Weight packing.  Packed layout: block (cotile, chunk, slice) at
//   ((cotile * nchunks + chunk) * nslices + slice) * 16 KB : [hi 128x32 bf16][lo 128x32 bf16],
//   element (row r, k) at byte r*64 + cv_swz(k>>3, r)*16 + (k&7)*2.
extern "C" int64_t rick_conv_packed_bytes(int Co, int Ci, int nslices) {
    return (int64_t)cdiv(Co, CV_BM) * cdiv(Ci, CV_CK) * nslices * CV_WSTEP_BYTES;
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, int64_t s_co, int64_t s_ci,
                                                          int64_t s_t, int Co, int Ci, int nslices, float scale,
                                                          unsigned short *__restrict__ packed, int64_t total, int split) {
    const int nchunks = (Ci + CV_CK - 1) / CV_CK;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i & 31);
        const int r = (int)((i >> 5) & 127);
        int64_t blk = i >> 12;
        const int slice = (int)(blk % nslices);
        blk /= nslices;
        const int chunk = (int)(blk % nchunks);
        const int cot = (int)(blk / nchunks);
        const int co = cot * CV_BM + r, ci = chunk * CV_CK + k;
        float v = 0.f;
        if (co < Co && ci < Ci) v = w[co * s_co + ci * s_ci + slice * s_t] * scale;
        unsigned short h, l;
        if (split == 1) {
            h = f32_to_bf16_rne(v);
            l = 0;
        } else {
            h = (unsigned short)(__float_as_uint(v) >> 16);
            l = f32_to_bf16_rne(v - bf16_to_f32(h));
        }
        const int64_t base = (i >> 12) * (CV_WSTEP_BYTES / 2);
        const int off = r * 32 + cv_swz(k >> 3, r) * 8 + (k & 7);
        packed[base + off] = h;
        packed[base + CV_WTILE_BYTES / 2 + off] = l;
    }
}

extern "C" int rick_conv_pack_weight(const float *w, int64_t s_co, int64_t s_ci, int64_t s_t, int Co, int Ci,
                                     int nslices, float scale, int split, void *packed, void *stream) {
    if (!w || !packed || Co <= 0 || Ci <= 0 || nslices <= 0 || (split != 1 && split != 2)) return RICK_EINVAL;
    const int64_t total = (int64_t)cdiv(Co, CV_BM) * cdiv(Ci, CV_CK) * nslices * CV_BM * CV_CK;
    int64_t nb = cdiv64(total, 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, s_co, s_ci, s_t, Co,
                       Ci, nslices, scale, (unsigned short *)packed, total, split);
    RICK_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
struct ConvTiling {
    int tw_log2, th_log2;       // position tile = (1<<tw) x (1<<th) x nb images = 128 (igemm) / 64 (wgrad)
    int nb;                     // images per tile
    int ntx, nty, ntn;          // tiles along x, y, image groups
    int dymin, dxmin;
    int PH, PW, NPP;            // patch extents (input pixels), NPP = nb*PH*PW
    int nchunks, ncot;
    int nsplit, cps;            // igemm split-K over channel chunks: splits, chunks per split
};

static int ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

static int make_tiling(const rick_conv_geom *g, int tile_positions, ConvTiling *t) {
    if (g->ntaps < 1 || g->ntaps > RICK_MAX_TAPS) return RICK_EINVAL;
    const int tl = ilog2_ceil(tile_positions);
    int tw = ilog2_ceil(g->GW);
    if (tw > 4) tw = 4;
    if (tw < 2) tw = 2;
    if (tw > tl) tw = tl;
    int th = ilog2_ceil(g->GH);
    if (th > tl - tw) th = tl - tw;
    t->tw_log2 = tw;
    t->th_log2 = th;
    t->nb = tile_positions >> (tw + th);
    t->ntx = cdiv(g->GW, 1 << tw);
    t->nty = cdiv(g->GH, 1 << th);
    t->ntn = cdiv(g->N, t->nb);
    int dymin = g->dy[0], dymax = g->dy[0], dxmin = g->dx[0], dxmax = g->dx[0];
    for (int i = 1; i < g->ntaps; i++) {
        dymin = g->dy[i] < dymin ? g->dy[i] : dymin;
        dymax = g->dy[i] > dymax ? g->dy[i] : dymax;
        dxmin = g->dx[i] < dxmin ? g->dx[i] : dxmin;
        dxmax = g->dx[i] > dxmax ? g->dx[i] : dxmax;
    }
    t->dymin = dymin;
    t->dxmin = dxmin;
    t->PH = ((1 << th) - 1) * g->is + (dymax - dymin) + 1;
    t->PW = ((1 << tw) - 1) * g->is + (dxmax - dxmin) + 1;
    t->NPP = t->nb * t->PH * t->PW;
    t->nchunks = cdiv(g->Ci, CV_CK);
    t->ncot = cdiv(g->Co, CV_BM);
    t->nsplit = 1;
    t->cps = t->nchunks;
    return 0;
}

// Split-K plan for launches with too few blocks to fill 256 CUs (the 4x4..32x32, 512-channel layers:
// K = 4608 is long while there are only 4..128 output tiles).
static void igemm_plan_split(ConvTiling *t) {
    const int base = t->ntx * t->nty * t->ntn * t->ncot;
    if (base >= 384 || t->nchunks < 2) return;
    int want = cdiv(768, base);
    if (want > t->nchunks) want = t->nchunks;
    t->cps = cdiv(t->nchunks, want);
    t->nsplit = cdiv(t->nchunks, t->cps);
}

// XCD-aware bijective remap of the linear block id: blocks that share an XCD (id % 8) get a
// contiguous range of logical tiles, so neighbouring position tiles of one co-tile share L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Stage the fp32 input patch of one 32-channel chunk into LDS as bf16 hi (+ lo).
template <int SPLIT>
__device__ __forceinline__ void stage_patch(const float *__restrict__ x, const float *__restrict__ iscale,
                                            unsigned char *ph, unsigned char *pl, const rick_conv_geom &g,
                                            const ConvTiling &t, int n0, int iy0, int ix0, int chunk) {
    const int items = t.NPP * 8;
    const int phw = t.PH * t.PW;
    const bool vec = (g.Ci & 3) == 0;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int pix = it >> 3, c4 = it & 7;
        const int nbi = pix / phw;
        const int rem = pix - nbi * phw;
        const int py = rem / t.PW, px = rem - py * t.PW;
        const int n = n0 + nbi, iy = iy0 + py, ix = ix0 + px;
        const int ci = chunk * CV_CK + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW && ci < g.Ci) {
            const float *src = x + (((int64_t)n * g.IH + iy) * g.IW + ix) * g.Ci + ci;
            if (vec) {
                v = *reinterpret_cast<const float4 *>(src);
            } else {
                v.x = src[0];
                if (ci + 1 < g.Ci) v.y = src[1];
                if (ci + 2 < g.Ci) v.z = src[2];
                if (ci + 3 < g.Ci) v.w = src[3];
            }
            if (iscale) {
                const float *sp = iscale + (int64_t)n * g.Ci + ci;
                v.x *= sp[0];
                if (ci + 1 < g.Ci) v.y *= sp[1];
                if (ci + 2 < g.Ci) v.z *= sp[2];
                if (ci + 3 < g.Ci) v.w *= sp[3];
            }
        }
        uint2 hi, lo;
        split4<SPLIT>(v, hi, lo);
        const int off = pix * 64 + cv_swz(c4 >> 1, pix) * 16 + (c4 & 1) * 8;
        *reinterpret_cast<uint2 *>(ph + off) = hi;
        if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + off) = lo;
    }
}

template <int SPLIT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const float *__restrict__ x,
                                                            const unsigned char *__restrict__ wpk,
                                                            float *__restrict__ out, const float *__restrict__ iscale,
                                                            const float *__restrict__ oscale, float *__restrict__ ws,
                                                            const rick_conv_geom g, const ConvTiling t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wbuf = smem;                               // [2][16 KB]
    unsigned char *ph = smem + 2 * CV_WSTEP_BYTES;            // [NPP][64 B]
    unsigned char *pl = ph + ((t.NPP * 64 + 15) & ~15);

    const int nwg = gridDim.x;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int npos_tiles = t.ntx * t.nty * t.ntn;
    int pt = lid % npos_tiles;
    const int split = (lid / npos_tiles) % t.nsplit;
    const int cot = lid / (npos_tiles * t.nsplit);
    const int c_begin = split * t.cps;
    const int c_end = c_begin + t.cps < t.nchunks ? c_begin + t.cps : t.nchunks;
    const int tx_i = pt % t.ntx;
    pt /= t.ntx;
    const int ty_i = pt % t.nty;
    const int tn_i = pt / t.nty;
    const int gx0 = tx_i << t.tw_log2, gy0 = ty_i << t.th_log2, n0 = tn_i * t.nb;
    const int iy0 = gy0 * g.is + t.dymin, ix0 = gx0 * g.is + t.dxmin;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kg = lane >> 4;

    // per-lane LDS byte offsets of the A rows (weights)
    int a_off[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = wm * 64 + i * 16 + l15;
        a_off[i] = row * 64 + cv_swz(kg, row) * 16;
    }
    // per-lane patch pixel index of the 4 position columns
    int pb[4];
    const int tw_mask = (1 << t.tw_log2) - 1, th_mask = (1 << t.th_log2) - 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pos = wn * 64 + j * 16 + l15;
        const int px = pos & tw_mask, py = (pos >> t.tw_log2) & th_mask, nbi = pos >> (t.tw_log2 + t.th_log2);
        pb[j] = (nbi * t.PH + py * g.is) * t.PW + px * g.is;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};


    const int nks = (c_end - c_begin) * g.ntaps;
    // packed-weight base of this co-tile; k-step (chunk c, tap tt) lives at block (c*nslices + wt[tt])
    const unsigned char *wbase = wpk + (int64_t)cot * t.nchunks * g.nslices * CV_WSTEP_BYTES;
    constexpr int WCOPY = (SPLIT == 2 ? CV_WSTEP_BYTES : CV_WTILE_BYTES) / (256 * 16);   // uint4 per thread: 4 or 2

    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;   // named (not an array): stays in VGPRs
#define CV_WLOAD(src)                                                   \
    do {                                                                \
        w0 = (src)[threadIdx.x];                                        \
        w1 = (src)[256 + threadIdx.x];                                  \
        if (WCOPY == 4) {                                               \
            w2 = (src)[512 + threadIdx.x];                              \
            w3 = (src)[768 + threadIdx.x];                              \
        }                                                               \
    } while (0)
#define CV_WSTORE(dst)                                                  \
    do {                                                                \
        reinterpret_cast<uint4 *>(dst)[threadIdx.x] = w0;               \
        reinterpret_cast<uint4 *>(dst)[256 + threadIdx.x] = w1;         \
        if (WCOPY == 4) {                                               \
            reinterpret_cast<uint4 *>(dst)[512 + threadIdx.x] = w2;     \
            reinterpret_cast<uint4 *>(dst)[768 + threadIdx.x] = w3;     \
        }                                                               \
    } while (0)
    // ---- prologue: patch(0), W(0)
    stage_patch<SPLIT>(x, iscale, ph, pl, g, t, n0, iy0, ix0, c_begin);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(
            wbase + ((int64_t)c_begin * g.nslices + g.wt[0]) * CV_WSTEP_BYTES);
        CV_WLOAD(src);
        CV_WSTORE(wbuf);
    }
    __syncthreads();

    int chunk = c_begin, tap = 0;
    for (int ks = 0; ks < nks; ks++) {
        // prefetch W(ks+1) into registers
        int nchunk = chunk, ntap = tap + 1;
        if (ntap == g.ntaps) { ntap = 0; nchunk++; }
        const bool more = ks + 1 < nks;
        if (more) {
            const uint4 *src = reinterpret_cast<const uint4 *>(
                wbase + ((int64_t)nchunk * g.nslices + g.wt[ntap]) * CV_WSTEP_BYTES);
            CV_WLOAD(src);
        }
        // ---- compute k-step ks
        {
            const unsigned char *wb = wbuf + (ks & 1) * CV_WSTEP_BYTES;
            const int toff = (g.dy[tap] - t.dymin) * t.PW + (g.dx[tap] - t.dxmin);
            bf16x8 ahi[4], alo[4], bhi[4], blo[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ahi[i] = *reinterpret_cast<const bf16x8 *>(wb + a_off[i]);
                if (SPLIT == 2) alo[i] = *reinterpret_cast<const bf16x8 *>(wb + CV_WTILE_BYTES + a_off[i]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int pp = pb[j] + toff;
                const int off = pp * 64 + cv_swz(kg, pp) * 16;
                bhi[j] = *reinterpret_cast<const bf16x8 *>(ph + off);
                if (SPLIT == 2) blo[j] = *reinterpret_cast<const bf16x8 *>(pl + off);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (SPLIT == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[i], bhi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[i], blo[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[i], bhi[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            if (ntap == 0) {   // next k-step starts a new channel chunk: restage the patch
                __syncthreads();
                stage_patch<SPLIT>(x, iscale, ph, pl, g, t, n0, iy0, ix0, nchunk);
            }
            unsigned char *wd = wbuf + ((ks + 1) & 1) * CV_WSTEP_BYTES;
            CV_WSTORE(wd);
        }
        __syncthreads();
        chunk = nchunk;
        tap = ntap;
    }

    // ---- epilogue: out[n, pix, co] = alpha * oscale[n,co] * acc
    const bool covec = (g.Co & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pos = wn * 64 + j * 16 + l15;
        const int px = pos & tw_mask, py = (pos >> t.tw_log2) & th_mask, nbi = pos >> (t.tw_log2 + t.th_log2);
        const int n = n0 + nbi, gy = gy0 + py, gx = gx0 + px;
        if (n >= g.N || gy >= g.GH || gx >= g.GW) continue;
        if (t.nsplit > 1) {   // raw partial sums -> workspace [split][n, gy, gx][Co]; scaled in the reduce kernel
            float *wrow = ws + (((int64_t)split * g.N + n) * g.GH * g.GW + (int64_t)gy * g.GW + gx) * g.Co;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int co = cot * CV_BM + wm * 64 + i * 16 + kg * 4;
                if (co >= g.Co) continue;
                if (covec) {
                    *reinterpret_cast<float4 *>(wrow + co) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else {
                    wrow[co] = acc[i][j][0];
                    if (co + 1 < g.Co) wrow[co + 1] = acc[i][j][1];
                    if (co + 2 < g.Co) wrow[co + 2] = acc[i][j][2];
                    if (co + 3 < g.Co) wrow[co + 3] = acc[i][j][3];
                }
            }
            continue;
        }
        const int64_t opix = ((int64_t)n * g.OH + gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0;
        float *orow = out + opix * g.Co;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int co = cot * CV_BM + wm * 64 + i * 16 + kg * 4;
            if (co >= g.Co) continue;
            f32x4 v = acc[i][j] * g.alpha;
            if (oscale) {
                const float *sp = oscale + (int64_t)n * g.Co + co;
                v[0] *= sp[0];
                if (co + 1 < g.Co) v[1] *= sp[1];
                if (co + 2 < g.Co) v[2] *= sp[2];
                if (co + 3 < g.Co) v[3] *= sp[3];
            }
            if (covec) {
                *reinterpret_cast<float4 *>(orow + co) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                orow[co] = v[0];
                if (co + 1 < g.Co) orow[co + 1] = v[1];
                if (co + 2 < g.Co) orow[co + 2] = v[2];
                if (co + 3 < g.Co) orow[co + 3] = v[3];
            }
        }
    }
}

// out[n, pix(gy,gx), co] = alpha * oscale[n,co] * sum_s ws[s][n,gy,gx][co]
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ out,
                                                                  const float *__restrict__ oscale, rick_conv_geom g,
                                                                  int nsplit) {
    const int64_t per = (int64_t)g.N * g.GH * g.GW * g.Co;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % g.Co);
        int64_t pos = i / g.Co;
        const int gx = (int)(pos % g.GW);
        pos /= g.GW;
        const int gy = (int)(pos % g.GH);
        const int n = (int)(pos / g.GH);
        float s = 0.f;
        for (int sp = 0; sp < nsplit; sp++) s += ws[sp * per + i];
        s *= g.alpha;
        if (oscale) s *= oscale[(int64_t)n * g.Co + co];
        out[(((int64_t)n * g.OH + gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0) * g.Co + co] = s;
    }
}

static int check_geom(const rick_conv_geom *g) {
    if (!g || g->N <= 0 || g->IH <= 0 || g->IW <= 0 || g->Ci <= 0 || g->OH <= 0 || g->OW <= 0 || g->Co <= 0 ||
        g->GH <= 0 || g->GW <= 0 || g->is <= 0 || g->os <= 0 || g->ntaps < 1 || g->ntaps > RICK_MAX_TAPS ||
        g->nslices < 1 || (g->split != 1 && g->split != 2))
        return RICK_EINVAL;
    for (int i = 0; i < g->ntaps; i++)
        if (g->wt[i] < 0 || g->wt[i] >= g->nslices) return RICK_EINVAL;
    if ((g->GH - 1) * g->os + g->oy0 >= g->OH || (g->GW - 1) * g->os + g->ox0 >= g->OW) return RICK_EINVAL;
    return 0;
}

extern "C" int64_t rick_conv_igemm_workspace_bytes(const rick_conv_geom *g) {
    if (check_geom(g)) return -1;
    ConvTiling t;
    if (make_tiling(g, CV_BN, &t)) return -1;
    igemm_plan_split(&t);
    return t.nsplit > 1 ? (int64_t)t.nsplit * g->N * g->GH * g->GW * g->Co * 4 : 0;
}

extern "C" int rick_conv_igemm_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                   const float *oscale, const rick_conv_geom *g, void *workspace, void *stream) {
    if (!x || !packed_w || !out || check_geom(g)) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed_w) % 16) return RICK_EINVAL;
    ConvTiling t;
    if (make_tiling(g, CV_BN, &t)) return RICK_EINVAL;
    igemm_plan_split(&t);
    if (t.nsplit > 1 && (!workspace || ((uintptr_t)workspace % 16))) return RICK_EINVAL;
    const size_t lds = 2 * CV_WSTEP_BYTES + 2 * (size_t)((t.NPP * 64 + 15) & ~15);
    if (lds > 160 * 1024) return RICK_EINVAL;
    const int64_t nwg = (int64_t)t.ntx * t.nty * t.ntn * t.ncot * t.nsplit;
    if (nwg > 0x7fffffff) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    if (g->split == 2) {
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void *)conv_igemm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr2 = true;
        }
        hipLaunchKernelGGL(conv_igemm_kernel<2>, dim3((unsigned)nwg), dim3(256), lds, st, x,
                           (const unsigned char *)packed_w, out, iscale, oscale, ws, *g, t);
    } else {
        static bool attr1 = false;
        if (!attr1) {
            (void)hipFuncSetAttribute((const void *)conv_igemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr1 = true;
        }
        hipLaunchKernelGGL(conv_igemm_kernel<1>, dim3((unsigned)nwg), dim3(256), lds, st, x,
                           (const unsigned char *)packed_w, out, iscale, oscale, ws, *g, t);
    }
    if (t.nsplit > 1) {
        const int64_t per = (int64_t)g->N * g->GH * g->GW * g->Co;
        int64_t nb = cdiv64(per, 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, t.nsplit);
    }
    RICK_LAUNCH_STATUS();
}

// ==========================================================================================
// Weight gradient.  GEMM view per block: D[128 co][NT taps x 32 ci] += GY^T[128 co][K] * X[K][..],
// K = positions (64 per tile, two 32-deep MFMA k-steps), split-K over position tiles.
// Both operands are k-strided in NHWC memory (k = position), so they are staged row-major
// [position][channel] and consumed through ds_read_b64_tr_b16 transposing reads.
//   gy image : [64 pos][128 co] bf16, 256 B rows, 32-byte granules XOR-swizzled with
//              key(r) = ((r>>3)&1)*4 + (r&3)  (conflict-free for the 8 rows a half-wave reads)
//   x  patch : same image as the igemm kernel ([pixel][32 ci], cv_swz slots)
#define WG_TILE 64
#define WG_GY_BYTES (WG_TILE * CV_BM * 2)   // 16 KB (one of hi / lo)
#define WG_PMAX 12                          // patch float4 per thread held in registers (NPP <= 384 pixels)

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

__device__ __forceinline__ bf16x8 tr_read2(const unsigned char *base, int off0, int off1) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_ptr)(base + off0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_ptr)(base + off1));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ int wg_key(int r) { return (((r >> 3) & 1) << 2) + (r & 3); }

template <int NT, int SPLIT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                         float *__restrict__ ws, const float *__restrict__ ascale,
                                                         const float *__restrict__ bscale, const rick_conv_geom g,
                                                         const ConvTiling t, int nsplit, int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *gh = smem;                          // gy hi
    unsigned char *gl = smem + WG_GY_BYTES;            // gy lo
    unsigned char *ph = smem + 2 * WG_GY_BYTES;
    unsigned char *pl = ph + ((t.NPP * 64 + 15) & ~15);

    const int lid = blockIdx.x;
    const int chunk = lid % t.nchunks;
    const int cot = (lid / t.nchunks) % t.ncot;
    const int split = lid / (t.nchunks * t.ncot);
    const int ntiles = t.ntx * t.nty * t.ntn;
    const int tile_begin = split * tiles_per_split;
    const int tile_end = tile_begin + tiles_per_split < ntiles ? tile_begin + tiles_per_split : ntiles;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int tw_mask = (1 << t.tw_log2) - 1, th_mask = (1 << t.th_log2) - 1;

    // A (gy^T) row byte offsets for the 4 (kk, h) row groups this lane addresses, column part per co tile
    int a_row[2][2];     // [kk][h] -> r*256, with key
    int a_key[2][2];
    int pbase[2][2];     // patch pixel of position r
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int r = kk * 32 + G * 8 + h * 4 + q;
            a_row[kk][h] = r * 256;
            a_key[kk][h] = wg_key(r) * 32;
            const int px = r & tw_mask, py = (r >> t.tw_log2) & th_mask, nbi = r >> (t.tw_log2 + t.th_log2);
            pbase[kk][h] = (nbi * t.PH + py * g.is) * t.PW + px * g.is;
        }
    const int b_kg = wn * 2 + (p >> 1), b_sub = (p & 1) * 8;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int tt = 0; tt < NT; tt++) acc[i][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline: the fp32 gy tile (8 float4 / thread) and input patch (<= WG_PMAX float4 / thread)
    // of tile t+1 are fetched into registers while the MFMAs of tile t run; conversion to bf16 hi/lo and
    // the LDS writes happen after the barrier that retires tile t.
    float4 gq[8];
    float4 pq[WG_PMAX];
    const int p_items = t.NPP * 8;
    const int phw = t.PH * t.PW;
    const bool xvec = (g.Ci & 3) == 0, gvec = (g.Co & 3) == 0;

    auto load_tile = [&](int tile) {
        int pt = tile;
        const int tx_i = pt % t.ntx;
        pt /= t.ntx;
        const int ty_i = pt % t.nty;
        const int tn_i = pt / t.nty;
        const int gx0 = tx_i << t.tw_log2, gy0 = ty_i << t.th_log2, n0 = tn_i * t.nb;
        const int iy0 = gy0 * g.is + t.dymin, ix0 = gx0 * g.is + t.dxmin;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int it = threadIdx.x + k * 256;
            const int r = it >> 5, c4 = it & 31;
            const int px = r & tw_mask, py = (r >> t.tw_log2) & th_mask, nbi = r >> (t.tw_log2 + t.th_log2);
            const int n = n0 + nbi, yy = gy0 + py, xx = gx0 + px;
            const int co = cot * CV_BM + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < g.N && yy < g.GH && xx < g.GW && co < g.Co) {
                const int64_t opix = ((int64_t)n * g.OH + yy * g.os + g.oy0) * g.OW + xx * g.os + g.ox0;
                const float *src = gy + opix * g.Co + co;
                if (gvec) {
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    v.x = src[0];
                    if (co + 1 < g.Co) v.y = src[1];
                    if (co + 2 < g.Co) v.z = src[2];
                    if (co + 3 < g.Co) v.w = src[3];
                }
                if (ascale) {
                    const float *sp = ascale + (int64_t)n * g.Co + co;
                    v.x *= sp[0];
                    if (co + 1 < g.Co) v.y *= sp[1];
                    if (co + 2 < g.Co) v.z *= sp[2];
                    if (co + 3 < g.Co) v.w *= sp[3];
                }
            }
            gq[k] = v;
        }
#pragma unroll
        for (int k = 0; k < WG_PMAX; k++) {
            const int it = threadIdx.x + k * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < p_items) {
                const int pix = it >> 3, c4 = it & 7;
                const int nbi = pix / phw;
                const int rem = pix - nbi * phw;
                const int py = rem / t.PW, px = rem - py * t.PW;
                const int n = n0 + nbi, iy = iy0 + py, ix = ix0 + px;
                const int ci = chunk * CV_CK + c4 * 4;
                if (n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW && ci < g.Ci) {
                    const float *src = x + (((int64_t)n * g.IH + iy) * g.IW + ix) * g.Ci + ci;
                    if (xvec) {
                        v = *reinterpret_cast<const float4 *>(src);
                    } else {
                        v.x = src[0];
                        if (ci + 1 < g.Ci) v.y = src[1];
                        if (ci + 2 < g.Ci) v.z = src[2];
                        if (ci + 3 < g.Ci) v.w = src[3];
                    }
                    if (bscale) {
                        const float *sp = bscale + (int64_t)n * g.Ci + ci;
                        v.x *= sp[0];
                        if (ci + 1 < g.Ci) v.y *= sp[1];
                        if (ci + 2 < g.Ci) v.z *= sp[2];
                        if (ci + 3 < g.Ci) v.w *= sp[3];
                    }
                }
            }
            pq[k] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int it = threadIdx.x + k * 256;
            const int r = it >> 5, c4 = it & 31;
            uint2 hi, lo;
            split4<SPLIT>(gq[k], hi, lo);
            const int off = r * 256 + ((c4 * 8) ^ (wg_key(r) * 32));
            *reinterpret_cast<uint2 *>(gh + off) = hi;
            if (SPLIT == 2) *reinterpret_cast<uint2 *>(gl + off) = lo;
        }
#pragma unroll
        for (int k = 0; k < WG_PMAX; k++) {
            const int it = threadIdx.x + k * 256;
            if (it < p_items) {
                const int pix = it >> 3, c4 = it & 7;
                uint2 hi, lo;
                split4<SPLIT>(pq[k], hi, lo);
                const int off = pix * 64 + cv_swz(c4 >> 1, pix) * 16 + (c4 & 1) * 8;
                *reinterpret_cast<uint2 *>(ph + off) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + off) = lo;
            }
        }
    };

    if (tile_begin < tile_end) load_tile(tile_begin);
    for (int tile = tile_begin; tile < tile_end; tile++) {
        __syncthreads();   // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (tile + 1 < tile_end) load_tile(tile + 1);
        // ---- two 32-deep k-steps
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 ahi[4], alo[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int cb = (wm * 64 + i * 16 + p * 4) * 2;
                const int o0 = a_row[kk][0] + (cb ^ a_key[kk][0]);
                const int o1 = a_row[kk][1] + (cb ^ a_key[kk][1]);
                ahi[i] = tr_read2(gh, o0, o1);
                if (SPLIT == 2) alo[i] = tr_read2(gl, o0, o1);
            }
#pragma unroll
            for (int tt = 0; tt < NT; tt++) {
                if (tt < g.ntaps) {
                    const int toff = (g.dy[tt] - t.dymin) * t.PW + (g.dx[tt] - t.dxmin);
                    const int pp0 = pbase[kk][0] + toff, pp1 = pbase[kk][1] + toff;
                    const int o0 = pp0 * 64 + cv_swz(b_kg, pp0) * 16 + b_sub;
                    const int o1 = pp1 * 64 + cv_swz(b_kg, pp1) * 16 + b_sub;
                    const bf16x8 bhi = tr_read2(ph, o0, o1);
                    bf16x8 blo;
                    if (SPLIT == 2) blo = tr_read2(pl, o0, o1);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (SPLIT == 2) {
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[i], bhi, acc[i][tt], 0, 0, 0);
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[i], blo, acc[i][tt], 0, 0, 0);
                        }
                        acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[i], bhi, acc[i][tt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- partial tile -> workspace [split][cot][chunk][tap][128 co][32 ci]
    float *wsb = ws + (((int64_t)split * t.ncot + cot) * t.nchunks + chunk) * g.ntaps * (CV_BM * CV_CK);
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
        if (tt < g.ntaps) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int co = wm * 64 + i * 16 + G * 4 + r;
                    const int ci = wn * 16 + (lane & 15);
                    wsb[(tt * CV_BM + co) * CV_CK + ci] = acc[i][tt][r];
                }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ gw,
                                                           int64_t s_co, int64_t s_ci, int64_t s_t, int Co, int Ci,
                                                           int ncot, int nchunks, int ntaps, int nsplit, float alpha,
                                                           int accumulate, rick_conv_geom g) {
    const int64_t per_split = (int64_t)ncot * nchunks * ntaps * CV_BM * CV_CK;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_split; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i & 31);
        const int r = (int)((i >> 5) & 127);
        int64_t blk = i >> 12;
        const int tt = (int)(blk % ntaps);
        blk /= ntaps;
        const int chunk = (int)(blk % nchunks);
        const int cot = (int)(blk / nchunks);
        const int co = cot * CV_BM + r, ci = chunk * CV_CK + k;
        if (co >= Co || ci >= Ci) continue;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; sp++) s += ws[sp * per_split + i];
        float *dst = gw + co * s_co + ci * s_ci + g.wt[tt] * s_t;
        const float v = s * alpha;
        *dst = accumulate ? *dst + v : v;
    }
}

static void wgrad_plan(const rick_conv_geom *g, ConvTiling *t, int *nsplit, int *tps) {
    make_tiling(g, WG_TILE, t);
    const int ntiles = t->ntx * t->nty * t->ntn;
    int want = 768 / (t->ncot * t->nchunks);
    if (want < 1) want = 1;
    if (want > ntiles) want = ntiles;
    *tps = cdiv(ntiles, want);
    *nsplit = cdiv(ntiles, *tps);
}

extern "C" int64_t rick_conv_wgrad_workspace_bytes(const rick_conv_geom *g) {
    if (check_geom(g)) return -1;
    ConvTiling t;
    int nsplit, tps;
    wgrad_plan(g, &t, &nsplit, &tps);
    return (int64_t)nsplit * t.ncot * t.nchunks * g->ntaps * CV_BM * CV_CK * 4;
}

template <int NT>
static void launch_wgrad(const float *x, const float *gy, float *ws, const float *ascale, const float *bscale,
                         const rick_conv_geom *g, const ConvTiling &t, int nsplit, int tps, size_t lds, hipStream_t st) {
    const unsigned nwg = (unsigned)(nsplit * t.ncot * t.nchunks);
    if (g->split == 2) {
        (void)hipFuncSetAttribute((const void *)conv_wgrad_kernel<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((conv_wgrad_kernel<NT, 2>), dim3(nwg), dim3(256), lds, st, x, gy, ws, ascale, bscale, *g, t,
                           nsplit, tps);
    } else {
        (void)hipFuncSetAttribute((const void *)conv_wgrad_kernel<NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((conv_wgrad_kernel<NT, 1>), dim3(nwg), dim3(256), lds, st, x, gy, ws, ascale, bscale, *g, t,
                           nsplit, tps);
    }
}

extern "C" int rick_conv_wgrad_f32(const float *x, const float *gy, float *gw, int64_t s_co, int64_t s_ci, int64_t s_t,
                                   const float *ascale, const float *bscale, const rick_conv_geom *g, int accumulate,
                                   void *workspace, void *stream) {
    if (!x || !gy || !gw || !workspace || check_geom(g)) return RICK_EINVAL;
    if (g->ntaps > 9) return RICK_EINVAL;
    ConvTiling t;
    int nsplit, tps;
    wgrad_plan(g, &t, &nsplit, &tps);
    const size_t lds = 2 * WG_GY_BYTES + 2 * (size_t)((t.NPP * 64 + 15) & ~15);
    if (lds > 160 * 1024 || t.NPP * 8 > WG_PMAX * 256) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    if (g->ntaps == 1) launch_wgrad<1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (g->ntaps == 2) launch_wgrad<2>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (g->ntaps <= 4) launch_wgrad<4>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else launch_wgrad<9>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    const int64_t per_split = (int64_t)t.ncot * t.nchunks * g->ntaps * CV_BM * CV_CK;
    int64_t nb = cdiv64(per_split, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ws, gw, s_co, s_ci, s_t, g->Co, g->Ci,
                       t.ncot, t.nchunks, g->ntaps, nsplit, g->alpha, accumulate, *g);
    RICK_LAUNCH_STATUS();
}
