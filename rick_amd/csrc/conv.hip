// Implicit-GEMM convolution family on gfx950 MFMA (v_mfma_f32_16x16x32_f16).
//
// One kernel covers every convolution of the StyleGAN2 G/D hot path and its data gradient
// (model_probe_tune.py:122,265,274,280): the host describes a launch as a grid of output
// "positions" plus a tap table (rick_conv_geom, include/rick_hip.h).
//
// GEMM view per block:  D[128 co][128 positions] += W[128 co][K] * X[K][128 positions],
// K = (32-channel chunk, tap).  fp32 activations are read from HBM ONCE per block and chunk
// as a spatial patch (tile + halo), converted to fp16 hi/lo on the way into LDS and re-used
// by all taps (9x re-use for 3x3), so neither im2col traffic nor the fp32->fp16x2 split is
// paid per tap.  Weights are pre-packed (rick_conv_pack_weight) into the exact swizzled LDS
// image of the A operand, so staging them is a linear 16-byte copy.
//
// Precision: split=2 multiplies hi*hi + hi*lo + lo*hi with fp32 accumulation, operands scaled by a per-block
// power of two so that the fp16 range is centred on the block's data (conv_common.h: ~2^-22 relative per
// product, 64x finer than a bf16 hi/lo split at the same three MFMAs); split=1 is plain fp16.
//
// Wave tiling: 256 threads = 4 waves in 2(co) x 2(pos); each wave owns 64 co x 64 positions
// = 4x4 MFMA tiles (16 accumulators of 4 VGPRs).  Output rows (co) land 4-consecutive per
// lane, so NHWC stores are float4.
//
// LDS bank conflicts: both operands are [row][32 k] fp16 (64 B rows) read as ds_read_b128 per
// lane (row = lane&15, k-group = lane>>4).  16-byte slot index is XOR-swizzled with bit 2 of
// the row: slot = kg ^ (((row>>2)&1)<<1)  -> conflict-free for 16 consecutive rows at any
// offset (checked by simulation against the gfx950 ds_read_b128 lane groups).
#include "conv_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------
// Weight packing.  Packed layout: block (cotile, chunk, slice) at
//   ((cotile * nchunks + chunk) * nslices + slice) * 16 KB : [hi 128x32 fp16][lo 128x32 fp16],
//   element (row r, k) at byte r*64 + cv_swz(k>>3, r)*16 + (k&7)*2,
// followed by a 64-byte trailer {float unscale = 2^-e, float scale = 2^e}: the tensor is packed as (w * scale) * 2^e
// with e from a sample of the tensor (cv_pow2_scale), and every kernel that multiplies with the packed image folds
// `unscale` into its output factor.  "Weights" are not always O(1) parameters: the second-order terms (R1, path length)
// run gradients through the A operand, so the exponent is taken from the data, on the device, per pack.
__host__ __device__ static inline int64_t packed_tile_bytes(int Co, int Ci, int nslices) {
    return (int64_t)cdiv(Co, CV_BM) * cdiv(Ci, CV_CK) * nslices * CV_WSTEP_BYTES;
}
extern "C" int64_t rick_conv_packed_bytes(int Co, int Ci, int nslices) {
    return packed_tile_bytes(Co, Ci, nslices) + CV_WTRAILER_BYTES;
}

#define PK_SAMPLES 32    // per thread: up to 8192 strided samples of the tensor

// One block: sampled amax of (w * scale) -> {2^-e, 2^e} into the trailer.
__device__ __forceinline__ void pack_exponent(const float *__restrict__ w, int64_t s_co, int64_t s_ci, int64_t s_t, int Co,
                                              int Ci, int nslices, float scale, float *__restrict__ trailer) {
    __shared__ float red[4];
    const int64_t total = (int64_t)Co * Ci * nslices;
    int64_t stride = total / (PK_SAMPLES * 256);
    if (stride < 1) stride = 1;
    stride |= 1;                                    // odd: walks every tap slice and channel residue
    float m = 0.f;
    for (int j = 0; j < PK_SAMPLES; j++) {
        const int64_t i = ((int64_t)j * 256 + threadIdx.x) * stride;
        if (i < total) {
            const int sl = (int)(i % nslices);
            const int64_t r = i / nslices;
            const int ci = (int)(r % Ci), co = (int)(r / Ci);
            m = fmaxf(m, fabsf(w[co * s_co + ci * s_ci + sl * s_t] * scale));
        }
    }
    m = block_amax(m, red);
    if (threadIdx.x == 0) {
        float sc, un;
        cv_pow2_scale(m, sc, un);
        trailer[0] = un;
        trailer[1] = sc;
    }
    if (threadIdx.x >= 2 && threadIdx.x < CV_WTRAILER_BYTES / 4) trailer[threadIdx.x] = 0.f;   // (packed images compare equal byte for byte)
}

__global__ __launch_bounds__(256) void pack_exponent_kernel(const float *__restrict__ w, int64_t s_co, int64_t s_ci,
                                                            int64_t s_t, int Co, int Ci, int nslices, float scale,
                                                            float *__restrict__ trailer) {
    pack_exponent(w, s_co, s_ci, s_t, Co, Ci, nslices, scale, trailer);
}

__global__ __launch_bounds__(256) void pack_exponent_multi_kernel(const rick_pack_desc *__restrict__ descs) {
    const rick_pack_desc ds = descs[blockIdx.x];
    pack_exponent(ds.w, ds.s_co, ds.s_ci, ds.s_t, ds.Co, ds.Ci, ds.nslices, ds.scale,
                  reinterpret_cast<float *>((unsigned char *)ds.packed + packed_tile_bytes(ds.Co, ds.Ci, ds.nslices)));
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, int64_t s_co, int64_t s_ci,
                                                          int64_t s_t, int Co, int Ci, int nslices, float scale,
                                                          unsigned short *__restrict__ packed, int64_t total, int split,
                                                          const float *__restrict__ trailer) {
    const int nchunks = (Ci + CV_CK - 1) / CV_CK;
    const float pscale = trailer[1];
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
        if (co < Co && ci < Ci) v = w[co * s_co + ci * s_ci + slice * s_t] * scale * pscale;
        unsigned short h, l;
        split1(v, h, l, split);
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
    float *trailer = reinterpret_cast<float *>((unsigned char *)packed + packed_tile_bytes(Co, Ci, nslices));
    int64_t nb = cdiv64(total, 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_exponent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, s_co, s_ci, s_t, Co, Ci, nslices,
                       scale, trailer);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, s_co, s_ci, s_t, Co,
                       Ci, nslices, scale, (unsigned short *)packed, total, split, (const float *)trailer);
    RICK_LAUNCH_STATUS();
}

// Many weights in ONE launch (all convolutions of a network after an optimiser step): block b serves the
// descriptor d with blk_begin[d] <= b < blk_begin[d+1]; a thread owns one (co, ci) of a 128 x 32 tile and
// walks the tap slices (contiguous in memory for [O, I, kh, kw] parameters).
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const rick_pack_desc *__restrict__ descs, int n, int split) {
    int d = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_begin) d = i;
    const rick_pack_desc ds = descs[d];
    const int nchunks = (ds.Ci + CV_CK - 1) / CV_CK;
    const float pscale =
        reinterpret_cast<const float *>((const unsigned char *)ds.packed + packed_tile_bytes(ds.Co, ds.Ci, ds.nslices))[1];
    const int i = ((int)blockIdx.x - ds.blk_begin) * 256 + threadIdx.x;   // over (cotile, chunk, r, k)
    const int k = i & 31, r = (i >> 5) & 127, tile = i >> 12;
    const int chunk = tile % nchunks, cot = tile / nchunks;
    const int co = cot * CV_BM + r, ci = chunk * CV_CK + k;
    const bool ok = co < ds.Co && ci < ds.Ci;
    const float *src = ds.w + (ok ? co * ds.s_co + ci * ds.s_ci : 0);
    unsigned short *dst = (unsigned short *)ds.packed + (int64_t)tile * ds.nslices * (CV_WSTEP_BYTES / 2) +
                          r * 32 + cv_swz(k >> 3, r) * 8 + (k & 7);
    for (int sl = 0; sl < ds.nslices; sl++) {
        float v = src[sl * ds.s_t] * ds.scale * pscale;
        if (!ok) v = 0.f;
        unsigned short h, l;
        split1(v, h, l, split);
        dst[(int64_t)sl * (CV_WSTEP_BYTES / 2)] = h;
        dst[(int64_t)sl * (CV_WSTEP_BYTES / 2) + CV_WTILE_BYTES / 2] = l;
    }
}

extern "C" int rick_conv_pack_blocks(int Co, int Ci) { return cdiv(Co, CV_BM) * cdiv(Ci, CV_CK) * (CV_BM * CV_CK / 256); }

extern "C" int rick_conv_pack_weights_multi(const rick_pack_desc *descs_device, int n, int total_blocks, int split,
                                            void *stream) {
    if (!descs_device || n < 1 || total_blocks < 1 || (split != 1 && split != 2)) return RICK_EINVAL;
    hipLaunchKernelGGL(pack_exponent_multi_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, descs_device);
    hipLaunchKernelGGL(pack_weight_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       descs_device, n, split);
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
    int nbe;                    // images per tile actually used (min(nb, N)); patch holds nbe images
    int nsplit, cps;            // igemm split-K over channel chunks: splits, chunks per split
    int pkb;                    // wgrad: bit of the patch pixel index that keys the slot swizzle
    int debug;                  // ablation switches for tools/bench_conv.py (RICK_CONV_DEBUG); 0 in production
};

// Ablation switches (tools/bench_conv.py) exist only in builds made with -DRICK_ABLATION; the production library never
// reads the environment, so a stray variable cannot change results.
static int ablation_env(const char *name, int dflt) {
#ifdef RICK_ABLATION
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// A 64-byte page of zeros in device memory: out-of-range staging items load from it, so their registers need no
// zero-select afterwards (4 VALU per item; the conv kernels are bound by the issue of their staging instructions).
__device__ __attribute__((aligned(64))) float g_zero_page[16];

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
    t->nbe = t->nb < g->N ? t->nb : g->N;
    t->ntx = cdiv(g->GW, 1 << tw);
    t->nty = cdiv(g->GH, 1 << th);
    t->ntn = cdiv(g->N, t->nbe);
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
    t->NPP = t->nbe * t->PH * t->PW;
    t->nchunks = cdiv(g->Ci, CV_CK);
    t->ncot = cdiv(g->Co, CV_BM);
    t->nsplit = 1;
    t->cps = t->nchunks;
    t->pkb = (g->is == 1 && tw == 4) ? 3 : 2;
    t->debug = ablation_env("RICK_CONV_DEBUG", 0);
    return 0;
}

// Stride-2 inputs need a patch of ~4x the position tile; a 64-position block (each wave 64 co x 32 positions)
// keeps it at ~37 KB so two blocks still fit per CU.
static size_t igemm_lds_bytes(const ConvTiling &t, bool has_iscale) {
    (void)has_iscale;   // the scale table is always present (filled with 1 when the conv has no input scale)
    return 2 * CV_WSTEP_BYTES + 2 * (size_t)(t.NPP + 1) * 64 + (size_t)((t.NPP + 3) & ~3) * 4 +
           (size_t)t.nbe * t.cps * CV_CK * 4;
}

static int igemm_tile_positions(const rick_conv_geom *g) { return (g->is >= 2 && g->ntaps > 1) ? 64 : CV_BN; }

// Split-K plan for launches with too few blocks to fill 256 CUs (the 4x4..64x64 layers with few output tiles but a
// long K = Ci x taps).  Two blocks are resident per CU, so the chip runs the grid in "waves" of 512 blocks; the
// plan minimises  waves x (k-steps per block + fixed block cost)  over the split factor — e.g. 128 output tiles
// with 16 chunks run best as 4 splits (exactly one wave of 512 blocks: measured 69 us vs 82 us for 6 splits).
static void igemm_plan_split(ConvTiling *t, int ntaps) {
    static const int fixed = ablation_env("RICK_SPLITK_FIXED", 32);
    const int base = t->ntx * t->nty * t->ntn * t->ncot;
    if (t->nchunks < 2) return;
    int best_cps = t->nchunks, best_cost = 1 << 30;
    const int smax = t->nchunks < 16 ? t->nchunks : 16;
    for (int s = 1; s <= smax; s++) {
        const int cps = cdiv(t->nchunks, s), se = cdiv(t->nchunks, cps);
        const int waves = cdiv(base * se, 512);
        const int cost = waves * (cps * ntaps + fixed) + (se > 1 ? 6 : 0);
        if (cost < best_cost) {
            best_cost = cost;
            best_cps = cps;
        }
    }
    t->cps = best_cps;
    t->nsplit = cdiv(t->nchunks, best_cps);
}

// Patch pixel -> (image-in-tile, row, col) table, built once per block so the staging loops need no
// integer divisions: entry = nbi << 20 | py << 10 | px.
__device__ __forceinline__ void build_patch_table(unsigned *ptab, const ConvTiling &t) {
    const int phw = t.PH * t.PW;
    for (int pix = threadIdx.x; pix < t.NPP; pix += 256) {
        const int nbi = pix / phw;
        const int rem = pix - nbi * phw;
        const int py = rem / t.PW, px = rem - py * t.PW;
        ptab[pix] = ((unsigned)nbi << 20) | ((unsigned)py << 10) | (unsigned)px;
    }
}

// ==========================================================================================
// Forward / data-gradient kernel.
#define IG_PMAX 10        // non-DEEP: patch float4 per thread prefetched in registers (NPP <= 320 pixels)
#define IG_PSET_DEEP 6     // DEEP: two register sets of 6 (NPP <= 192 pixels), chunks prefetched two ahead
#define IG_DEEP_NPP (32 * IG_PSET_DEEP)

typedef __attribute__((address_space(3))) unsigned char cv_lds_u8;
typedef __attribute__((address_space(1))) const unsigned char cv_gbl_u8;

template <int N>
__device__ __forceinline__ void cv_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void cv_lds_barrier() {   // raw barrier: no vmcnt(0) (an LDS-DMA in flight survives it)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int SPLIT, bool VEC, bool DEEP, int NJ, int NT, int WDMA = 0>
__device__ __forceinline__ void igemm_body(const float *__restrict__ x, const unsigned char *__restrict__ wpk,
                                           float *__restrict__ out, const float *__restrict__ iscale,
                                           const float *__restrict__ oscale, float *__restrict__ ws,
                                           const rick_conv_geom &g, const ConvTiling &t, const int bid, const int nwg,
                                           const rick_conv_epilogue &epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wbuf = smem;                               // [2][16 KB]
    unsigned char *ph = smem + (WDMA == 3 ? 3 : 2) * CV_WSTEP_BYTES;   // [NPP + 1][64 B]  (WDMA = 3: a ring of 3 weight tiles)
    unsigned char *pl = ph + (t.NPP + 1) * 64;                // (+1: spare row for out-of-patch items)
    unsigned *ptab = reinterpret_cast<unsigned *>(pl + (t.NPP + 1) * 64);             // [NPP]
    float *sct = reinterpret_cast<float *>(ptab + ((t.NPP + 3) & ~3));                // [nbe][cps * 32] input scales
    build_patch_table(ptab, t);

    const int lid = xcd_remap(bid, nwg);
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
    const int gx0 = tx_i << t.tw_log2, gy0 = ty_i << t.th_log2, n0 = tn_i * t.nbe;
    const int iy0 = gy0 * g.is + t.dymin, ix0 = gx0 * g.is + t.dxmin;
    const int cspan = t.cps * CV_CK;

    // per-(image, input channel) scales of this block's images and channel range (1 without an input scale): read from HBM
    // once, multiplied by the block exponent below, then applied from LDS while the patch is converted — no global-load
    // round trip and no branch per patch item
    if (iscale) {
        for (int i = threadIdx.x; i < t.nbe * cspan; i += 256) {
            const int nbi = i / cspan, c = c_begin * CV_CK + (i - nbi * cspan);
            sct[i] = (n0 + nbi < g.N && c < g.Ci) ? iscale[(int64_t)(n0 + nbi) * g.Ci + c] : 0.f;
        }
    }
    __syncthreads();   // patch table and scale table complete
    // operand exponent of this block (conv_common.h), set by block_exponent() below once the first chunk is in registers
    float xscale = 1.f, unscale = 1.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kg = lane >> 4;

    int a_off[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = wm * 64 + i * 16 + l15;
        a_off[i] = row * 64 + cv_swz(kg, row) * 16;
    }
    int pb[NJ];   // NJ = position tiles of 16 per wave: 4 (128-position block) or 2 (64-position block, stride-2 input)
    const int tw_mask = (1 << t.tw_log2) - 1, th_mask = (1 << t.th_log2) - 1;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int pos = wn * (NJ * 16) + j * 16 + l15;
        const int px = pos & tw_mask, py = (pos >> t.tw_log2) & th_mask;
        int nbi = pos >> (t.tw_log2 + t.th_log2);
        nbi = nbi < t.nbe ? nbi : t.nbe - 1;     // image slots beyond nbe are masked in the epilogue; keep LDS reads in range
        pb[j] = (nbi * t.PH + py * g.is) * t.PW + px * g.is;
    }

    // ---- patch staging state.  The tile is fixed for the block, so validity, source offset (relative to
    // the tile-origin pointer) and LDS offset of each of this thread's patch items are computed once.
    // DEEP (patches of <= 160 pixels: 1x1 convs and the parity classes of transposed convs, whose chunks
    // last only 1-4 k-steps): the IG_PMAX register slots form TWO sets of 5 and chunks are prefetched two
    // ahead, so a load has two chunks' worth of MFMAs to land instead of one.
    // register prefetch slots per set: the unrolled 128-position form only serves patches of <= 192 pixels
    constexpr int PSET = (DEEP || (NT > 0 && NJ == 4)) ? IG_PSET_DEEP : IG_PMAX;
    constexpr int PREGS = DEEP ? 2 * IG_PSET_DEEP : PSET;
    const int p_items = t.NPP * 8;
    const int c4 = threadIdx.x & 7;                      // 256 % 8 == 0: same channel quad for all items
    const float *xt = x + (((int64_t)n0 * g.IH + iy0) * g.IW + ix0) * g.Ci + c4 * 4;
    int p_rel[PSET], p_lds[PSET], p_sc[PSET];
    unsigned p_ok = 0;
    float4 pq[PREGS];
#pragma unroll
    for (int k = 0; k < PSET; k++) {
        const int pix = (threadIdx.x >> 3) + 32 * k;
        p_rel[k] = 0;
        p_sc[k] = 0;
        // items past the patch land in a spare row behind it, so the LDS writes need no guard
        p_lds[k] = (pix < t.NPP ? pix : t.NPP) * 64 + cv_swz(c4 >> 1, pix) * 16 + (c4 & 1) * 8;
        if (pix < t.NPP) {
            const unsigned e = ptab[pix];
            const int nbi = (int)(e >> 20), py = (int)((e >> 10) & 1023), px = (int)(e & 1023);
            const int n = n0 + nbi, iy = iy0 + py, ix = ix0 + px;
            if (n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
                p_ok |= 1u << k;
                p_rel[k] = ((nbi * g.IH + py) * g.IW + px) * g.Ci;
                p_sc[k] = nbi * cspan + c4 * 4;
            }
        }
    }
    unsigned cur_ok[2] = {0, 0};   // validity of the items held in each register set (p_ok restricted by ci < Ci)
    auto issue_patch = [&](int chunk, auto SET) {   // raw loads only; consumers live in commit_patch
        constexpr int S = decltype(SET)::value;
        const int ci = chunk * CV_CK + c4 * 4;
        const unsigned okm = ci < g.Ci ? p_ok : 0u;
        cur_ok[S] = okm;
#pragma unroll
        for (int k = 0; k < PSET; k++) {
            const bool ok = (okm >> k) & 1u;
            pq[S * PSET + k] = load4<VEC>(ok ? xt + p_rel[k] + chunk * CV_CK : (VEC ? g_zero_page : x), ok, ci, g.Ci);
        }
    };
    auto commit_patch = [&](int chunk, auto SET) {
        constexpr int S = decltype(SET)::value;
        // input scale x block exponent.  ONE block-uniform branch around the whole item loop (two straight-line copies):
        // layers without an input scale multiply by the exponent from an SGPR and read no table.  With vector loads an
        // out-of-range item has read the zero page and needs no select.
        auto items = [&](auto ISC) {
#pragma unroll
            for (int k = 0; k < PSET; k++) {
                const bool ok = (cur_ok[S] >> k) & 1u;
                float4 v = pq[S * PSET + k];
                if (!VEC && !ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                if constexpr (decltype(ISC)::value)
                    split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sct + p_sc[k] + (chunk - c_begin) * CV_CK), hi, lo);
                else split4s<SPLIT>(v, xscale, hi, lo);
                *reinterpret_cast<uint2 *>(ph + p_lds[k]) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + p_lds[k]) = lo;
            }
        };
        if (iscale) items(std::true_type{});
        else items(std::false_type{});
        if (!DEEP) {
            // patches larger than IG_PMAX*32 pixels (stride-2 geometry): remaining items, synchronously
            for (int it = threadIdx.x + 256 * PSET; it < p_items; it += 256) {
                const int pix = it >> 3;
                const unsigned e = ptab[pix];
                const int n = n0 + (int)(e >> 20), iy = iy0 + (int)((e >> 10) & 1023), ix = ix0 + (int)(e & 1023);
                const int ci = chunk * CV_CK + c4 * 4;
                const bool ok = n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW && ci < g.Ci;
                float4 v = load4<VEC>(ok ? x + (((int64_t)n * g.IH + iy) * g.IW + ix) * g.Ci + ci : x, ok, ci, g.Ci);
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                if (iscale) split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sct + (int)(e >> 20) * cspan + (chunk - c_begin) * CV_CK + c4 * 4), hi, lo);
                else split4s<SPLIT>(v, xscale, hi, lo);      // (the scale table only exists with an input scale)
                const int off = pix * 64 + cv_swz(c4 >> 1, pix) * 16 + (c4 & 1) * 8;
                *reinterpret_cast<uint2 *>(ph + off) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + off) = lo;
            }
        }
    };
    // ---- operand exponent of this block: amax of |input scale * x| over the first channel chunk's patch, which
    // issue_patch has just put into the registers of set 0 (no extra loads: 32 channels x the whole patch), reduced over
    // the block -> x * 2^e (conv_common.h).  An all-zero first chunk (padding, pruned channels) falls back to explicit
    // samples over all of the block's chunks.  Called between issue_patch(c_begin) and commit_patch(c_begin).
    auto block_exponent = [&]() {
        float *red = reinterpret_cast<float *>(pl + t.NPP * 64);      // the spare row: not written before the first commit
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < PSET; k++) {
            float4 v = pq[k];
            if (!VEC && !((cur_ok[0] >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iscale) v = mul4(v, *reinterpret_cast<const float4 *>(sct + p_sc[k]));
            m = amax4(m, v);
        }
        m = block_amax(m, red);
        if (m == 0.f) {                 // block-uniform
            const int ncl = c_end - c_begin;
#pragma unroll 1
            for (int sidx = 0; sidx < 8; sidx++) {
                const int pix = (threadIdx.x >> 3) + 32 * ((sidx * 5 + 1) % PSET);
                const int chunk = c_begin + (sidx * ncl) / 8;
                const int ci = chunk * CV_CK + c4 * 4;
                bool ok = pix < t.NPP && ci < g.Ci;
                int nbi = 0, n = 0, iy = 0, ix = 0;
                if (ok) {
                    const unsigned e = ptab[pix];
                    nbi = (int)(e >> 20);
                    n = n0 + nbi;
                    iy = iy0 + (int)((e >> 10) & 1023);
                    ix = ix0 + (int)(e & 1023);
                    ok = n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                }
                float4 v = load4<VEC>(ok ? x + (((int64_t)n * g.IH + iy) * g.IW + ix) * g.Ci + ci : (VEC ? g_zero_page : x), ok, ci, g.Ci);
                if (iscale) v = mul4(v, *reinterpret_cast<const float4 *>(sct + nbi * cspan + (chunk - c_begin) * CV_CK + c4 * 4));
                if (ok) m = amax4(m, v);
            }
            __syncthreads();            // every thread has read `red`
            m = block_amax(m, red);
        }
        float xs, xu;
        cv_pow2_scale(m, xs, xu);
        xscale = cv_uniform(xs);
        // packed-weight exponent (trailer of the packed image)
        unscale = cv_uniform(xu * *reinterpret_cast<const float *>(wpk + (int64_t)t.ncot * t.nchunks * g.nslices * CV_WSTEP_BYTES));
        if (iscale) {                   // fold 2^e into the scale table
            for (int i = threadIdx.x; i < t.nbe * cspan; i += 256) sct[i] *= xscale;
            __syncthreads();
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nks = (c_end - c_begin) * g.ntaps;
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

    if constexpr (NT > 0) {
        // ---- straight-line form for a compile-time tap count (the 3x3 layers that carry the FLOPs).  One channel
        // chunk = NT fully unrolled k-steps; every k-step loads the next weight tile and a slice of the NEXT chunk's
        // patch items, unconditionally (past the end of the range the last chunk is staged again and never used).
        // With no branch between VMEM instructions hipcc keeps exact vmcnt counts: the weight store at the end of a
        // k-step waits only for its own 4 loads, and the patch loads stay in flight until the chunk boundary
        // (up to NT-1 k-steps) instead of being drained by the first weight wait.
        static_assert(!DEEP, "the unrolled form prefetches one chunk ahead");
        constexpr int IPT = (PSET + NT - 1) / NT;
        auto issue_item = [&](auto KC, int chunk) {
            constexpr int K = decltype(KC)::value;
            const bool ok = (cur_ok[0] >> K) & 1u;
            pq[K] = load4<VEC>(ok ? xt + p_rel[K] + chunk * CV_CK : (VEC ? g_zero_page : x), ok, chunk * CV_CK + c4 * 4, g.Ci);
        };
        // WDMA: the packed weight tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write
        // pass).  WDMA = 3: a ring of three slots, the tile of k-step ks + 2 is issued right behind the barrier that
        // opens k-step ks and stays in flight across the next barrier (counted s_waitcnt vmcnt, raw s_barrier).
        // WDMA = 2 (stride-2 form, whose 37 KB patch leaves no room for a third slot at two blocks per CU): the two
        // existing slots, tile ks + 1 issued behind the barrier of k-step ks.
        static_assert(WDMA == 0 || WDMA == 2 || (WDMA == 3 && NT % 3 == 0), "ring slot = tap % 3");
        static_assert(WDMA == 0 || SPLIT == 2, "the DMA copies whole hi + lo tiles");
        const int wave = threadIdx.x >> 6;
        auto issue_wdma = [&](int wchunk, int wtap, int slot) {
            const unsigned char *src = wbase + ((int64_t)wchunk * g.nslices + g.wt[wtap]) * CV_WSTEP_BYTES;
            unsigned char *dst = wbuf + slot * CV_WSTEP_BYTES;
#pragma unroll
            for (int q = 0; q < 4; q++)
                __builtin_amdgcn_global_load_lds((cv_gbl_u8 *)(src + (q * 256 + threadIdx.x) * 16),
                                                 (cv_lds_u8 *)(dst + (q * 256 + wave * 64) * 16), 16, 0, 0);
        };
        if constexpr (WDMA != 0) {
            issue_wdma(c_begin, 0, 0);
            if constexpr (WDMA == 3) issue_wdma(c_begin, 1, 1);
            issue_patch(c_begin, S0{});
            block_exponent();                     // (first use of the patch registers: hipcc drains the VM counter here)
            commit_patch(c_begin, S0{});
        } else {
        issue_patch(c_begin, S0{});
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(
                wbase + ((int64_t)c_begin * g.nslices + g.wt[0]) * CV_WSTEP_BYTES);
            CV_WLOAD(src);
            CV_WSTORE(wbuf);
        }
        block_exponent();
        commit_patch(c_begin, S0{});
        __syncthreads();
        }
        for (int chunk = c_begin; chunk < c_end; chunk++) {
            const int cnext = chunk + 1 < c_end ? chunk + 1 : chunk;
            const int par = ((chunk - c_begin) * NT) & 1;
            auto kstep = [&](auto TC) {
                constexpr int tap = decltype(TC)::value;
                if constexpr (WDMA == 3) {
                    // tile `tap` was issued two k-steps ago; behind it in the VM queue: that k-step's patch items, the 4
                    // DMA instructions of tile tap + 1 and the previous k-step's patch items
                    constexpr int t1 = (tap + NT - 1) % NT, t2 = (tap + NT - 2) % NT;
                    constexpr int p1 = PSET - t1 * IPT < 0 ? 0 : (PSET - t1 * IPT < IPT ? PSET - t1 * IPT : IPT);
                    constexpr int p2 = PSET - t2 * IPT < 0 ? 0 : (PSET - t2 * IPT < IPT ? PSET - t2 * IPT : IPT);
                    cv_wait_vm<4 + p1 + p2>();
                    cv_lds_barrier();                 // all pieces of tile `tap` landed; slot (tap + 2) % 3 has no reader left
                    const int wchunk = tap + 2 < NT ? chunk : cnext, wtap = tap + 2 < NT ? tap + 2 : tap + 2 - NT;
                    issue_wdma(wchunk, wtap, (tap + 2) % 3);
                } else if constexpr (WDMA == 2) {
                    // tile `tap` was issued one k-step ago; behind it: that k-step's patch items
                    constexpr int t1 = (tap + NT - 1) % NT;
                    constexpr int p1 = PSET - t1 * IPT < 0 ? 0 : (PSET - t1 * IPT < IPT ? PSET - t1 * IPT : IPT);
                    cv_wait_vm<p1>();
                    cv_lds_barrier();                 // tile `tap` landed; the other slot (read in the previous k-step) is free
                    const int wchunk = tap + 1 < NT ? chunk : cnext, wtap = tap + 1 < NT ? tap + 1 : 0;
                    issue_wdma(wchunk, wtap, (par + tap + 1) & 1);
                } else {   // weights of the next k-step -> registers
                    const int wchunk = tap + 1 < NT ? chunk : cnext, wtap = tap + 1 < NT ? tap + 1 : 0;
                    const uint4 *src = reinterpret_cast<const uint4 *>(
                        wbase + ((int64_t)wchunk * g.nslices + g.wt[wtap]) * CV_WSTEP_BYTES);
                    CV_WLOAD(src);
                }
                if constexpr (tap == 0) cur_ok[0] = cnext * CV_CK + c4 * 4 < g.Ci ? p_ok : 0u;
                if constexpr (tap * IPT < PSET) issue_item(std::integral_constant<int, tap * IPT>{}, cnext);
                if constexpr (IPT > 1 && tap * IPT + 1 < PSET) issue_item(std::integral_constant<int, tap * IPT + 1>{}, cnext);
                if constexpr (IPT > 2 && tap * IPT + 2 < PSET) issue_item(std::integral_constant<int, tap * IPT + 2>{}, cnext);
                static_assert(IPT <= 3 || NT >= 4, "patch items per k-step");
                if constexpr (IPT > 3) {   // few taps: the remaining items of this slice
                    auto rest = [&](auto KC) {
                        constexpr int K = decltype(KC)::value + 3;
                        if constexpr (K < IPT && tap * IPT + K < PSET) issue_item(std::integral_constant<int, tap * IPT + K>{}, cnext);
                    };
                    static_for<0, 8>(rest);
                }
                {
                    const unsigned char *wb = wbuf + (WDMA == 3 ? tap % 3 : (par + tap) & 1) * CV_WSTEP_BYTES;
                    const int toff = (g.dy[tap] - t.dymin) * t.PW + (g.dx[tap] - t.dxmin);
                    f16x8 ahi[4], alo[4], bhi[NJ], blo[NJ];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        ahi[i] = *reinterpret_cast<const f16x8 *>(wb + a_off[i]);
                        if (SPLIT == 2) alo[i] = *reinterpret_cast<const f16x8 *>(wb + CV_WTILE_BYTES + a_off[i]);
                    }
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        const int pp = pb[j] + toff;
                        const int off = pp * 64 + cv_swz(kg, pp) * 16;
                        bhi[j] = *reinterpret_cast<const f16x8 *>(ph + off);
                        if (SPLIT == 2) blo[j] = *reinterpret_cast<const f16x8 *>(pl + off);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++) {
                            if (SPLIT == 2) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo[j], acc[i][j], 0, 0, 0);
                            }
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi[j], acc[i][j], 0, 0, 0);
                        }
                }
                if constexpr (WDMA != 0) {
                    if constexpr (tap == NT - 1) {   // chunk boundary: all waves are done with the patch
                        cv_lds_barrier();
                        commit_patch(cnext, S0{});
                    }
                } else {
                if constexpr (tap == NT - 1) {   // chunk boundary: all waves are done with the patch
                    __syncthreads();
                    commit_patch(cnext, S0{});
                }
                CV_WSTORE(wbuf + ((par + tap + 1) & 1) * CV_WSTEP_BYTES);
                __syncthreads();
                }
            };
            static_for<0, NT>(kstep);
        }
    } else {
        // ---- prologue: patch(c_begin) [+ patch(c_begin+1) when DEEP], W(0)
        issue_patch(c_begin, S0{});
        if (DEEP && c_begin + 1 < c_end) issue_patch(c_begin + 1, S1{});
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(
                wbase + ((int64_t)c_begin * g.nslices + g.wt[0]) * CV_WSTEP_BYTES);
            CV_WLOAD(src);
            CV_WSTORE(wbuf);
        }
        block_exponent();
        commit_patch(c_begin, S0{});
        __syncthreads();

        int chunk = c_begin, tap = 0;
        for (int ks = 0; ks < nks; ks++) {
            int nchunk = chunk, ntap = tap + 1;
            if (ntap == g.ntaps) { ntap = 0; nchunk++; }
            const bool more = ks + 1 < nks;
            if (more && !(t.debug & 4)) {   // weights of k-step ks+1 -> registers
                const uint4 *src = reinterpret_cast<const uint4 *>(
                    wbase + ((int64_t)nchunk * g.nslices + g.wt[ntap]) * CV_WSTEP_BYTES);
                CV_WLOAD(src);
            }
            if (tap == 0 && !(t.debug & 2)) {   // patch prefetch: next chunk (two ahead when DEEP) -> registers
                const int cpre = chunk + (DEEP ? 2 : 1);
                if (cpre < c_end) {
                    if (DEEP && ((cpre - c_begin) & 1)) issue_patch(cpre, S1{});
                    else issue_patch(cpre, S0{});
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // prefetches are issued before the MFMAs, consumed after
            if (!(t.debug & 1)) {
                const unsigned char *wb = wbuf + (ks & 1) * CV_WSTEP_BYTES;
                const int toff = (g.dy[tap] - t.dymin) * t.PW + (g.dx[tap] - t.dxmin);
                f16x8 ahi[4], alo[4], bhi[NJ], blo[NJ];
    #pragma unroll
                for (int i = 0; i < 4; i++) {
                    ahi[i] = *reinterpret_cast<const f16x8 *>(wb + a_off[i]);
                    if (SPLIT == 2) alo[i] = *reinterpret_cast<const f16x8 *>(wb + CV_WTILE_BYTES + a_off[i]);
                }
    #pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const int pp = pb[j] + toff;
                    const int off = pp * 64 + cv_swz(kg, pp) * 16;
                    bhi[j] = *reinterpret_cast<const f16x8 *>(ph + off);
                    if (SPLIT == 2) blo[j] = *reinterpret_cast<const f16x8 *>(pl + off);
                }
    #pragma unroll
                for (int i = 0; i < 4; i++)
    #pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        if (SPLIT == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi[j], acc[i][j], 0, 0, 0);
                    }
            }
            if (more) {
                if (ntap == 0 && !(t.debug & 2)) {   // next k-step starts a new channel chunk: all waves are done with the patch
                    __syncthreads();
                    if (DEEP && ((nchunk - c_begin) & 1)) commit_patch(nchunk, S1{});
                    else commit_patch(nchunk, S0{});
                }
                unsigned char *wd = wbuf + ((ks + 1) & 1) * CV_WSTEP_BYTES;
                if (!(t.debug & 4)) CV_WSTORE(wd);
            }
            __syncthreads();
            chunk = nchunk;
            tap = ntap;
        }
    }

    // ---- epilogue.  Instantiated twice (with / without output scales) so the scale loads of a j-column
    // are unconditional in their variant: hipcc otherwise sinks each into its own branch + vmcnt(0).
    const bool covec = (g.Co & 3) == 0;
    const float oalpha = g.alpha * unscale;   // exact: the exponents are powers of two
    const float nwv = epi.noise ? epi.noise_w[0] : 0.f;
    auto epilogue = [&](auto HAS_OS, auto HAS_EP) {
        constexpr bool OS = decltype(HAS_OS)::value;
        constexpr bool EP = decltype(HAS_EP)::value;   // fused bias (+ noise) + LeakyReLU tail (rick_conv_epilogue)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int pos = wn * (NJ * 16) + j * 16 + l15;
            const int px = pos & tw_mask, py = (pos >> t.tw_log2) & th_mask, nbi = pos >> (t.tw_log2 + t.th_log2);
            const int n = n0 + nbi, gy = gy0 + py, gx = gx0 + px;
            if (nbi >= t.nbe || n >= g.N || gy >= g.GH || gx >= g.GW) continue;
            if (t.nsplit > 1) {   // raw partial sums -> workspace [split][n, gy, gx][Co]; scaled in the reduce kernel
                float *wrow = ws + (((int64_t)split * g.N + n) * g.GH * g.GW + (int64_t)gy * g.GW + gx) * g.Co;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int co = cot * CV_BM + wm * 64 + i * 16 + kg * 4;
                    if (co >= g.Co) continue;
                    // (partial sums leave the block without its operand exponents: blocks of one output tile may differ)
                    if (covec) {
                        *reinterpret_cast<float4 *>(wrow + co) =
                            make_float4(acc[i][j][0] * unscale, acc[i][j][1] * unscale, acc[i][j][2] * unscale, acc[i][j][3] * unscale);
                    } else {
                        wrow[co] = acc[i][j][0] * unscale;
                        if (co + 1 < g.Co) wrow[co + 1] = acc[i][j][1] * unscale;
                        if (co + 2 < g.Co) wrow[co + 2] = acc[i][j][2] * unscale;
                        if (co + 3 < g.Co) wrow[co + 3] = acc[i][j][3] * unscale;
                    }
                }
                continue;
            }
            const int64_t opix = ((int64_t)n * g.OH + gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0;
            float *orow = out + opix * g.Co;
            if (covec) {   // 4 batched float4 scale loads (clamped address), then 4 float4 stores
                float4 sc[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int co = cot * CV_BM + wm * 64 + i * 16 + kg * 4;
                    sc[i] = make_float4(oalpha, oalpha, oalpha, oalpha);
                    if (OS) {
                        const float4 o = *reinterpret_cast<const float4 *>(oscale + (int64_t)n * g.Co + (co < g.Co ? co : 0));
                        sc[i] = make_float4(o.x * oalpha, o.y * oalpha, o.z * oalpha, o.w * oalpha);
                    }
                }
                float nv = 0.f;
                if (EP && epi.noise)
                    nv = nwv * epi.noise[(int64_t)(epi.noise_nb == 1 ? 0 : n) * g.OH * g.OW +
                                         (int64_t)(gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int co = cot * CV_BM + wm * 64 + i * 16 + kg * 4;
                    if (co < g.Co) {
                        float4 v = make_float4(acc[i][j][0] * sc[i].x, acc[i][j][1] * sc[i].y, acc[i][j][2] * sc[i].z,
                                               acc[i][j][3] * sc[i].w);
                        if (EP) {   // same operation order as rick_bias_act_f32: + bias, + noise, LeakyReLU, gain
                            if (epi.bias) {
                                const float4 bv = *reinterpret_cast<const float4 *>(epi.bias + co);
                                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            }
                            v.x += nv; v.y += nv; v.z += nv; v.w += nv;
                            if (epi.act) {
                                v.x = (v.x > 0.f ? v.x : v.x * epi.slope) * epi.gain;
                                v.y = (v.y > 0.f ? v.y : v.y * epi.slope) * epi.gain;
                                v.z = (v.z > 0.f ? v.z : v.z * epi.slope) * epi.gain;
                                v.w = (v.w > 0.f ? v.w : v.w * epi.slope) * epi.gain;
                            }
                        }
                        *reinterpret_cast<float4 *>(orow + co) = v;
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int co = cot * CV_BM + wm * 64 + i * 16 + kg * 4;
                if (co >= g.Co) continue;
                f32x4 v = acc[i][j] * oalpha;
                if (OS) {
                    const float *sp = oscale + (int64_t)n * g.Co + co;
                    v[0] *= sp[0];
                    if (co + 1 < g.Co) v[1] *= sp[1];
                    if (co + 2 < g.Co) v[2] *= sp[2];
                    if (co + 3 < g.Co) v[3] *= sp[3];
                }
                orow[co] = v[0];
                if (co + 1 < g.Co) orow[co + 1] = v[1];
                if (co + 2 < g.Co) orow[co + 2] = v[2];
                if (co + 3 < g.Co) orow[co + 3] = v[3];
            }
        }
    };
    const bool has_ep = epi.bias || epi.noise || epi.act;   // (host: only with Co % 4 == 0)
    if (oscale && has_ep) epilogue(std::true_type{}, std::true_type{});
    else if (oscale) epilogue(std::true_type{}, std::false_type{});
    else if (has_ep) epilogue(std::false_type{}, std::true_type{});
    else epilogue(std::false_type{}, std::false_type{});
}

template <int SPLIT, bool VEC, bool DEEP, int NJ, int NT, int WDMA = 0>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const float *__restrict__ x,
                                                            const unsigned char *__restrict__ wpk,
                                                            float *__restrict__ out, const float *__restrict__ iscale,
                                                            const float *__restrict__ oscale, float *__restrict__ ws,
                                                            const rick_conv_geom g, const ConvTiling t,
                                                            const rick_conv_epilogue epi) {
    igemm_body<SPLIT, VEC, DEEP, NJ, NT, WDMA>(x, wpk, out, iscale, oscale, ws, g, t, blockIdx.x, gridDim.x, epi);
}

// Several geometries (the output-parity classes of a transposed convolution) in ONE launch: block ranges
// [blk_end[c-1], blk_end[c]) run class c.  Short-K classes (1 or 2 taps) overlap with the long ones instead
// of each paying its own launch, fill and tail.
#define IG_MAXCLS 8
struct IgemmMulti {
    int ncls;
    int deep;                        // all classes have NPP <= 160: two-ahead patch prefetch
    int blk_end[IG_MAXCLS];
    int64_t ws_off[IG_MAXCLS];       // float offset of each class's split-K workspace
    rick_conv_geom g[IG_MAXCLS];
    ConvTiling t[IG_MAXCLS];
};

template <int SPLIT, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_igemm_multi_kernel(const float *__restrict__ x,
                                                                  const unsigned char *__restrict__ wpk,
                                                                  float *__restrict__ out,
                                                                  const float *__restrict__ iscale,
                                                                  const float *__restrict__ oscale,
                                                                  float *__restrict__ ws, const IgemmMulti m) {
    int c = 0, start = 0;
    for (int i = 0; i + 1 < m.ncls; i++)
        if ((int)blockIdx.x >= m.blk_end[i]) {
            c = i + 1;
            start = m.blk_end[i];
        }
    // every parity class of a transposed conv has a patch of <= 160 pixels (tile + at most one halo row/col)
    // (the host only uses this kernel when every class qualifies for the two-ahead prefetch)
    const rick_conv_epilogue none = {nullptr, nullptr, nullptr, 1, 0, 0.f, 1.f};
    igemm_body<SPLIT, VEC, true, 4, 0>(x, wpk, out, iscale, oscale, ws + m.ws_off[c], m.g[c], m.t[c], (int)blockIdx.x - start,
                                       m.blk_end[c] - start, none);
}

// out[n, pix(gy,gx), co] = alpha * oscale[n,co] * sum_s ws[s][n,gy,gx][co]
// VEC4 (Co % 4 == 0): one float4 of 4 consecutive channels per thread and split, 32-bit index math (the guard in
// check_geom keeps N*OH*OW*Co below 2^31).
template <bool VEC4>
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ out,
                                                                  const float *__restrict__ oscale, rick_conv_geom g,
                                                                  int nsplit, rick_conv_epilogue epi) {
    constexpr int W = VEC4 ? 4 : 1;
    const float nwv = epi.noise ? epi.noise_w[0] : 0.f;
    const unsigned per = (unsigned)g.N * g.GH * g.GW * g.Co;
    const unsigned cow = (unsigned)g.Co / W;
    for (unsigned iw = blockIdx.x * 256 + threadIdx.x; iw < per / W; iw += gridDim.x * 256) {
        const unsigned cq = iw % cow;
        unsigned pos = iw / cow;
        const unsigned gx = pos % g.GW;
        pos /= g.GW;
        const unsigned gy = pos % g.GH;
        const unsigned n = pos / g.GH;
        const unsigned co = cq * W;
        const int64_t o = (((int64_t)n * g.OH + gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0) * g.Co + co;
        if (VEC4) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 *src = reinterpret_cast<const float4 *>(ws) + iw;
            for (int sp = 0; sp < nsplit; sp++) {
                const float4 v = src[(size_t)sp * (per / 4)];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            float4 sc = make_float4(g.alpha, g.alpha, g.alpha, g.alpha);
            if (oscale) {
                const float4 os4 = *reinterpret_cast<const float4 *>(oscale + (size_t)n * g.Co + co);
                sc = make_float4(os4.x, os4.y, os4.z, os4.w);
                s.x *= g.alpha; s.y *= g.alpha; s.z *= g.alpha; s.w *= g.alpha;
            }
            float4 v = make_float4(s.x * sc.x, s.y * sc.y, s.z * sc.z, s.w * sc.w);
            if (epi.bias) {
                const float4 bv = *reinterpret_cast<const float4 *>(epi.bias + co);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (epi.noise) {
                const float nv = nwv * epi.noise[(size_t)(epi.noise_nb == 1 ? 0 : n) * g.OH * g.OW +
                                                 (size_t)(gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0];
                v.x += nv; v.y += nv; v.z += nv; v.w += nv;
            }
            if (epi.act) {
                v.x = (v.x > 0.f ? v.x : v.x * epi.slope) * epi.gain;
                v.y = (v.y > 0.f ? v.y : v.y * epi.slope) * epi.gain;
                v.z = (v.z > 0.f ? v.z : v.z * epi.slope) * epi.gain;
                v.w = (v.w > 0.f ? v.w : v.w * epi.slope) * epi.gain;
            }
            *reinterpret_cast<float4 *>(out + o) = v;
        } else {
            float s = 0.f;
            for (int sp = 0; sp < nsplit; sp++) s += ws[(size_t)sp * per + iw];
            s *= g.alpha;
            if (oscale) s *= oscale[(size_t)n * g.Co + co];
            out[o] = s;
        }
    }
}

static const rick_conv_epilogue kNoEpilogue = {nullptr, nullptr, nullptr, 1, 0, 0.f, 1.f};

static void launch_splitk_reduce(const float *ws, float *out, const float *oscale, const rick_conv_geom *g, int nsplit,
                                 hipStream_t st, const rick_conv_epilogue &epi = kNoEpilogue) {
    const int64_t per = (int64_t)g->N * g->GH * g->GW * g->Co;
    const bool vec = (g->Co & 3) == 0 && (((uintptr_t)ws | (uintptr_t)out | (uintptr_t)(oscale ? oscale : out)) % 16) == 0;
    int64_t nb = cdiv64(vec ? per / 4 : per, 256);
    if (nb > 8192) nb = 8192;
    if (vec) hipLaunchKernelGGL(igemm_splitk_reduce_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, nsplit, epi);
    else hipLaunchKernelGGL(igemm_splitk_reduce_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, nsplit, kNoEpilogue);
}

static int check_geom(const rick_conv_geom *g) {
    if (!g || g->N <= 0 || g->IH <= 0 || g->IW <= 0 || g->Ci <= 0 || g->OH <= 0 || g->OW <= 0 || g->Co <= 0 ||
        g->GH <= 0 || g->GW <= 0 || g->is <= 0 || g->os <= 0 || g->ntaps < 1 || g->ntaps > RICK_MAX_TAPS ||
        g->nslices < 1 || (g->split != 1 && g->split != 2))
        return RICK_EINVAL;
    for (int i = 0; i < g->ntaps; i++)
        if (g->wt[i] < 0 || g->wt[i] >= g->nslices) return RICK_EINVAL;
    if ((g->GH - 1) * g->os + g->oy0 >= g->OH || (g->GW - 1) * g->os + g->ox0 >= g->OW) return RICK_EINVAL;
    // 32-bit element offsets are used inside a tile
    if ((int64_t)g->N * g->IH * g->IW * g->Ci >= (1LL << 31) || (int64_t)g->N * g->OH * g->OW * g->Co >= (1LL << 31))
        return RICK_EINVAL;
    return 0;
}

extern "C" int64_t rick_conv_igemm_workspace_bytes(const rick_conv_geom *g) {
    if (check_geom(g)) return -1;
    ConvTiling t;
    if (make_tiling(g, igemm_tile_positions(g), &t)) return -1;
    igemm_plan_split(&t, g->ntaps);
    return t.nsplit > 1 ? (int64_t)t.nsplit * g->N * g->GH * g->GW * g->Co * 4 : 0;
}

template <int SPLIT, bool VEC, bool DEEP, int NJ, int NT = 0, int WDMA = 0>
static void launch_igemm_k(unsigned nwg, size_t lds, hipStream_t st, const float *x, const unsigned char *wp, float *out,
                           const float *iscale, const float *oscale, float *ws, const rick_conv_geom *g,
                           const ConvTiling &t, const rick_conv_epilogue &epi) {
    (void)hipFuncSetAttribute((const void *)conv_igemm_kernel<SPLIT, VEC, DEEP, NJ, NT, WDMA>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_igemm_kernel<SPLIT, VEC, DEEP, NJ, NT, WDMA>), dim3(nwg), dim3(256), lds + (WDMA == 3 ? CV_WSTEP_BYTES : 0), st,
                       x, wp, out, iscale, oscale, ws, *g, t, epi);
}

template <int SPLIT, bool VEC>
static void launch_igemm(unsigned nwg, size_t lds, hipStream_t st, const float *x, const unsigned char *wp, float *out,
                         const float *iscale, const float *oscale, float *ws, const rick_conv_geom *g,
                         const ConvTiling &t, const rick_conv_epilogue &epi) {
    // production path (fp16x3, vector loads), 3x3 layers with a full grid: the straight-line 9-tap k-loop.  Its
    // loop body carries 0.65 non-MFMA VALU + 0.2 SALU instructions per MFMA against 1.4 + 1.2 (stride 1) / 2.2 + 2.3
    // (stride 2) of the generic tap loop (SQ_INSTS_* counters, profiles/r02_pmc_conv.json): +3..8 % on the Ci >= 256
    // stride-1 layers, +7 % at Ci = 128, +7..13 % on the stride-2 layers (tools/abl_u9.sh).
    static const int no_unroll = ablation_env("RICK_IGEMM_NOUNROLL", 0);
    static const int u9_minchunks = ablation_env("RICK_U9_MINCHUNKS", 4), u9_s2 = ablation_env("RICK_U9_S2", 1);
    static const int u9_split = ablation_env("RICK_U9_SPLIT", 4);    // split-K launches too when a split keeps >= 4 chunks (+5..13 %)
    const bool u9 = SPLIT == 2 && VEC && g->ntaps == 9 && !no_unroll && !t.debug && igemm_tile_positions(g) == CV_BN &&
                    t.NPP <= IG_DEEP_NPP && (t.nsplit == 1 || (u9_split && t.cps >= u9_split)) && t.cps >= u9_minchunks;
    const bool u9s2 = SPLIT == 2 && VEC && g->ntaps == 9 && u9_s2 && igemm_tile_positions(g) == 64 && t.NPP <= 32 * IG_PMAX &&
                      (t.nsplit == 1 || (u9_split && t.cps >= u9_split)) && t.cps >= u9_minchunks;
    // (stride 2: a k-step has half the MFMAs per weight tile, so the register-staged weight store weighed twice as much:
    // +11..17 % with the tiles by LDS-DMA into the two existing slots)
    if (u9s2 && ablation_env("RICK_WDMA2", 1)) launch_igemm_k<2, true, false, 2, 9, 2>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (u9s2) launch_igemm_k<2, true, false, 2, 9>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (igemm_tile_positions(g) == 64) launch_igemm_k<SPLIT, VEC, false, 2>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    // weights by LDS-DMA into a 3-slot ring (+3..5 % on the Ci >= 256 layers; keeps two blocks per CU)
    else if (u9 && ablation_env("RICK_WDMA", 1) && lds + CV_WSTEP_BYTES <= 80 * 1024)
        launch_igemm_k<2, true, false, 4, 9, 3>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (u9) launch_igemm_k<2, true, false, 4, 9>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (t.NPP <= IG_DEEP_NPP) launch_igemm_k<SPLIT, VEC, true, 4>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else launch_igemm_k<SPLIT, VEC, false, 4>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
}

extern "C" int rick_conv_igemm_act_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                       const float *oscale, const rick_conv_geom *g, const rick_conv_epilogue *epilogue,
                                       void *workspace, void *stream);

extern "C" int rick_conv_igemm_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                   const float *oscale, const rick_conv_geom *g, void *workspace, void *stream) {
    return rick_conv_igemm_act_f32(x, packed_w, out, iscale, oscale, g, nullptr, workspace, stream);
}

extern "C" int rick_conv_igemm_act_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                       const float *oscale, const rick_conv_geom *g, const rick_conv_epilogue *epilogue,
                                       void *workspace, void *stream) {
    if (!x || !packed_w || !out || check_geom(g)) return RICK_EINVAL;
    rick_conv_epilogue epi = kNoEpilogue;
    if (epilogue) {
        epi = *epilogue;
        if (epi.noise && (!epi.noise_w || (epi.noise_nb != 1 && epi.noise_nb != g->N))) return RICK_EINVAL;
        // the tail is applied on float4 channel groups: Co % 4 == 0, 16-byte aligned bias
        if ((epi.bias || epi.noise || epi.act) && ((g->Co & 3) || ((uintptr_t)(epi.bias ? epi.bias : x) % 16))) return RICK_EINVAL;
    }
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed_w | (uintptr_t)(iscale ? iscale : x) | (uintptr_t)(oscale ? oscale : x)) % 16)
        return RICK_EINVAL;
    ConvTiling t;
    if (make_tiling(g, igemm_tile_positions(g), &t)) return RICK_EINVAL;
    igemm_plan_split(&t, g->ntaps);
    if (t.nsplit > 1 && (!workspace || ((uintptr_t)workspace % 16))) return RICK_EINVAL;
    const size_t lds = igemm_lds_bytes(t, iscale != nullptr);
    if (lds > 160 * 1024 || t.PH > 1023 || t.PW > 1023) return RICK_EINVAL;
    const int64_t nwg = (int64_t)t.ntx * t.nty * t.ntn * t.ncot * t.nsplit;
    if (nwg > 0x7fffffff) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const unsigned char *wp = (const unsigned char *)packed_w;
    const bool vec = (g->Ci & 3) == 0;
    if (g->split == 2) {
        if (vec) launch_igemm<2, true>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
        else launch_igemm<2, false>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    } else {
        if (vec) launch_igemm<1, true>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
        else launch_igemm<1, false>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    }
    if (t.nsplit > 1) launch_splitk_reduce(ws, out, oscale, g, t.nsplit, st, epi);
    RICK_LAUNCH_STATUS();
}

static int plan_multi(const rick_conv_geom *geoms, int ngeom, IgemmMulti *m, size_t *lds_max, int64_t *ws_floats) {
    if (!geoms || ngeom < 1 || ngeom > IG_MAXCLS) return RICK_EINVAL;
    m->ncls = ngeom;
    m->deep = 1;
    int64_t blocks = 0, wsf = 0;
    size_t lmax = 0;
    for (int c = 0; c < ngeom; c++) {
        const rick_conv_geom *g = &geoms[c];
        if (check_geom(g) || g->Ci != geoms[0].Ci || g->Co != geoms[0].Co || g->split != geoms[0].split) return RICK_EINVAL;
        if (make_tiling(g, CV_BN, &m->t[c])) return RICK_EINVAL;
        igemm_plan_split(&m->t[c], g->ntaps);
        const ConvTiling &t = m->t[c];
        const size_t lds = igemm_lds_bytes(t, true);
        if (lds > 160 * 1024 || t.PH > 1023 || t.PW > 1023) return RICK_EINVAL;
        lmax = lds > lmax ? lds : lmax;
        if (t.NPP > IG_DEEP_NPP) m->deep = 0;
        blocks += (int64_t)t.ntx * t.nty * t.ntn * t.ncot * t.nsplit;
        if (blocks > 0x7fffffff) return RICK_EINVAL;
        m->blk_end[c] = (int)blocks;
        m->g[c] = *g;
        m->ws_off[c] = wsf;
        if (t.nsplit > 1) wsf += ((int64_t)t.nsplit * g->N * g->GH * g->GW * g->Co + 63) & ~63LL;
    }
    for (int c = ngeom; c < IG_MAXCLS; c++) {
        m->blk_end[c] = m->blk_end[ngeom - 1];
        m->ws_off[c] = 0;
    }
    *lds_max = lmax;
    *ws_floats = wsf;
    return 0;
}

extern "C" int64_t rick_conv_igemm_multi_workspace_bytes(const rick_conv_geom *geoms, int ngeom) {
    IgemmMulti m;
    size_t lds;
    int64_t wsf;
    if (plan_multi(geoms, ngeom, &m, &lds, &wsf)) return -1;
    return wsf * 4;
}

template <int SPLIT, bool VEC>
static void launch_igemm_multi(size_t lds, hipStream_t st, const float *x, const unsigned char *wp, float *out,
                               const float *iscale, const float *oscale, float *ws, const IgemmMulti &m) {
    (void)hipFuncSetAttribute((const void *)conv_igemm_multi_kernel<SPLIT, VEC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_igemm_multi_kernel<SPLIT, VEC>), dim3((unsigned)m.blk_end[m.ncls - 1]), dim3(256), lds, st, x,
                       wp, out, iscale, oscale, ws, m);
}

extern "C" int rick_conv_igemm_multi_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                         const float *oscale, const rick_conv_geom *geoms, int ngeom, void *workspace,
                                         void *stream) {
    if (!x || !packed_w || !out) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed_w | (uintptr_t)(iscale ? iscale : x) | (uintptr_t)(oscale ? oscale : x)) % 16)
        return RICK_EINVAL;
    IgemmMulti m;
    size_t lds;
    int64_t wsf;
    if (plan_multi(geoms, ngeom, &m, &lds, &wsf)) return RICK_EINVAL;
    if (wsf > 0 && (!workspace || ((uintptr_t)workspace % 16))) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const unsigned char *wp = (const unsigned char *)packed_w;
    const bool vec = (geoms[0].Ci & 3) == 0;
    if (!m.deep || ngeom == 1) {   // large patches: one ordinary launch per class; a single class needs no multi launch
        for (int c = 0; c < ngeom; c++) {
            const int rc = rick_conv_igemm_f32(x, packed_w, out, iscale, oscale, &geoms[c],
                                               m.t[c].nsplit > 1 ? (void *)(ws + m.ws_off[c]) : nullptr, stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (geoms[0].split == 2) {
        if (vec) launch_igemm_multi<2, true>(lds, st, x, wp, out, iscale, oscale, ws, m);
        else launch_igemm_multi<2, false>(lds, st, x, wp, out, iscale, oscale, ws, m);
    } else {
        if (vec) launch_igemm_multi<1, true>(lds, st, x, wp, out, iscale, oscale, ws, m);
        else launch_igemm_multi<1, false>(lds, st, x, wp, out, iscale, oscale, ws, m);
    }
    for (int c = 0; c < ngeom; c++)
        if (m.t[c].nsplit > 1) launch_splitk_reduce(ws + m.ws_off[c], out, oscale, &geoms[c], m.t[c].nsplit, st);
    RICK_LAUNCH_STATUS();
}

// ==========================================================================================
// Weight gradient.  GEMM view per block: D[128 co][NT taps x 32 ci] += GY^T[128 co][K] * X[K][..],
// K = positions (64 per tile, two 32-deep MFMA k-steps), split-K over position tiles.
// Both operands are k-strided in NHWC memory (k = position), so they are staged row-major
// [position][channel] and consumed through ds_read_b64_tr_b16 transposing reads.
//   gy image : [64 pos][128 co] fp16, 256 B rows, 32-byte granules XOR-swizzled with
//              key(r) = ((r>>3)&1)*4 + (r&3)  (conflict-free for the 8 rows a half-wave reads)
//   x  patch : [pixel][32 ci] like the igemm kernel's, 16-byte slots XOR-swizzled with bit `pkb` of the pixel index:
//              a transposing read takes 32 B of 8 patch rows per 32-lane group — rows p..p+3 and p+8..p+11 of a
//              16-wide position tile — so the two runs must differ in their slot key: bit 3 (pkb = 3, conflict-free;
//              the igemm's bit 2 makes every such read 2-way: 31 % of the LDS cycles were conflicts in round 2);
//              narrower tiles and stride-2 geometries keep bit 2 (tools/lds_sim.py)
#define WG_TILE 64
#define WG_GY_BYTES (WG_TILE * CV_BM * 2)   // 16 KB (one of hi / lo)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ f16x8 tr_read2(const unsigned char *base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 r = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, r);
}

__device__ __forceinline__ int wg_key(int r) { return (((r >> 3) & 1) << 2) + (r & 3); }
__device__ __forceinline__ int wg_pswz(int kg, int pix, int kb) { return kg ^ (((pix >> kb) & 1) << 1); }

// FAST (pipelined form; the host selects it for layers whose position grid is an exact multiple of the tile and whose
// channel counts fill the 128 x 32 block — every 3x3 / 1x1 convolution of both networks at >= 8x8): staging is stripped to
// what such a layer needs.  gy items are always valid (one unconditional load, no mask bookkeeping); x items test two
// unsigned compares (halo of the padding) and read the zero page when outside; conversion is the fp16 split plus, for
// the modulated layers only, the scale multiply — no zero-select.  ~15 / ~22 instructions per item instead of ~30.  The
// kernel is bound by the ISSUE of exactly these instructions (one wave per SIMD, 2-3 of them per MFMA, and an MFMA
// leaves room for ~2), not by the matrix pipe: measured +20...27 % (280 -> 330-345 TFLOP/s on the 64^2...256^2 layers).
template <int NT, int SPLIT, bool VEC, int PMAX, bool PIPE, int FAST = 0>   // FAST: 1 = no per-channel scales, 2 = with
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                         float *__restrict__ ws, const float *__restrict__ ascale,
                                                         const float *__restrict__ bscale, const rick_conv_geom g,
                                                         const ConvTiling t, int nsplit, int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // one operand buffer = [gy hi 16 KB][gy lo 16 KB][patch hi (NPP+1) x 64 B][patch lo]; PIPE keeps two of them
    // (+1 patch row: spare row for out-of-patch items)
    const int bufsz = 2 * WG_GY_BYTES + 2 * (t.NPP + 1) * 64;
    unsigned char *gh = smem;                          // gy hi
    unsigned char *gl = smem + WG_GY_BYTES;            // gy lo
    unsigned char *ph = smem + 2 * WG_GY_BYTES;
    unsigned char *pl = ph + (t.NPP + 1) * 64;
    unsigned *ptab = reinterpret_cast<unsigned *>(smem + (PIPE ? 2 : 1) * bufsz);
    float *sA = reinterpret_cast<float *>(ptab + ((t.NPP + 3) & ~3));   // [N][128 co] scales of gy (1 if none)
    float *sB = sA + g.N * CV_BM;                                        // [N][32 ci]  scales of x
    build_patch_table(ptab, t);

    // XCD-aware mapping: blocks with equal blockIdx % 8 share an XCD (and its L2).  Every XCD owns its own
    // slices of the position range; all (co-tile, chunk) blocks of a slice run there, so a gy tile is fetched
    // into ONE L2 and re-used by the nchunks blocks that need it.  (Placement only affects speed.)
    int split, cc;
    if ((nsplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, spx = nsplit >> 3;
        split = xcd + 8 * (j % spx);
        cc = j / spx;
    } else {
        split = blockIdx.x / (t.nchunks * t.ncot);
        cc = blockIdx.x % (t.nchunks * t.ncot);
    }
    const int chunk = cc % t.nchunks;
    const int cot = cc / t.nchunks;
    const int ntiles = t.ntx * t.nty * t.ntn;
    const int tile_begin = split * tiles_per_split;
    const int tile_end = tile_begin + tiles_per_split < ntiles ? tile_begin + tiles_per_split : ntiles;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int tw_mask = (1 << t.tw_log2) - 1, th_mask = (1 << t.th_log2) - 1;

    int a_row[2][2], a_key[2][2], pbase[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int r = kk * 32 + G * 8 + h * 4 + q;
            a_row[kk][h] = r * 256;
            a_key[kk][h] = wg_key(r) * 32;
            const int px = r & tw_mask, py = (r >> t.tw_log2) & th_mask;
            int nbi = r >> (t.tw_log2 + t.th_log2);
            nbi = nbi < t.nbe ? nbi : t.nbe - 1;   // masked rows (zero gy) still read finite patch data
            pbase[kk][h] = (nbi * t.PH + py * g.is) * t.PW + px * g.is;
        }
    const int b_kg = wn * 2 + (p >> 1), b_sub = (p & 1) * 8;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int tt = 0; tt < NT; tt++) acc[i][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging state (tile-invariant parts computed once per thread)
    float4 gq[8];
    float4 pq[PMAX];
    int g_rel[8], g_lds[8], p_rel[PMAX], p_lds[PMAX];
    unsigned g_pyx[8], p_pyx[PMAX];
    const int gc4 = threadIdx.x & 31, pc4 = threadIdx.x & 7;
    const int co_base = cot * CV_BM, ci_base = chunk * CV_CK;
    const int gco = co_base + gc4 * 4, pci = ci_base + pc4 * 4;
    // per-(image, channel) scales of this block's channel ranges -> LDS once (applied in store_tile without
    // a global-load round trip per item)
    for (int i = threadIdx.x; i < g.N * CV_BM; i += 256) {
        const int n = i >> 7, co = co_base + (i & 127);
        sA[i] = !ascale ? 1.f : co < g.Co ? ascale[(int64_t)n * g.Co + co] : 0.f;
    }
    for (int i = threadIdx.x; i < g.N * CV_CK; i += 256) {
        const int n = i >> 5, ci = ci_base + (i & 31);
        sB[i] = !bscale ? 1.f : ci < g.Ci ? bscale[(int64_t)n * g.Ci + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int r = (threadIdx.x >> 5) + 8 * k;
        const int px = r & tw_mask, py = (r >> t.tw_log2) & th_mask, nbi = r >> (t.tw_log2 + t.th_log2);
        g_rel[k] = ((nbi * g.OH + py * g.os) * g.OW + px * g.os) * g.Co;
        g_pyx[k] = (gco < g.Co && nbi < t.nbe) ? (((unsigned)nbi << 20) | ((unsigned)py << 10) | (unsigned)px) : 0xffffffffu;
        g_lds[k] = r * 256 + ((gc4 * 8) ^ (wg_key(r) * 32));
    }
#pragma unroll
    for (int k = 0; k < PMAX; k++) {
        const int pix = (threadIdx.x >> 3) + 32 * k;
        p_rel[k] = 0;
        p_pyx[k] = 0xffffffffu;
        p_lds[k] = (pix < t.NPP ? pix : t.NPP) * 64 + wg_pswz(pc4 >> 1, pix, t.pkb) * 16 + (pc4 & 1) * 8;
        if (pix < t.NPP) {
            const unsigned e = ptab[pix];
            p_rel[k] = (((int)(e >> 20) * g.IH + (int)((e >> 10) & 1023)) * g.IW + (int)(e & 1023)) * g.Ci;
            if (pci < g.Ci) p_pyx[k] = e;
        }
    }

    // ---- operand exponents of this block (conv_common.h): amax of gy * ascale and of x * bscale over 4 of the block's
    // position tiles x 2 staging items each (4096 values per operand), then 2^e is folded into the LDS scale tables.
    float punscale;   // 2^-(e_gy + e_x): applied to the partial tile on its way to the workspace
    float xsa, xsb;   // 2^e_gy, 2^e_x (SGPRs: the multiplier of layers without per-channel scales)
    {
        float ma = 0.f, mb = 0.f;
        const int nt = tile_end - tile_begin;
#pragma unroll
        for (int sidx = 0; sidx < 4; sidx++) {
            int pt = tile_begin + (sidx * nt) / 4;
            const int tx_i = pt % t.ntx;
            pt /= t.ntx;
            const int ty_i = pt % t.nty;
            const int tn_i = pt / t.nty;
            const int gx0 = tx_i << t.tw_log2, gy0 = ty_i << t.th_log2, n0 = tn_i * t.nbe;
            const int iy0 = gy0 * g.is + t.dymin, ix0 = gx0 * g.is + t.dxmin;
            const float *gbase = gy + (((int64_t)n0 * g.OH + gy0 * g.os + g.oy0) * g.OW + gx0 * g.os + g.ox0) * g.Co + gco;
            const float *xbase = x + (((int64_t)n0 * g.IH + iy0) * g.IW + ix0) * g.Ci + pci;
            const int nrem = g.N - n0, yrem = g.GH - gy0, xrem = g.GW - gx0;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                {
                    const int k = (2 * sidx + 5 * kk + 1) & 7;
                    const unsigned e = g_pyx[k];
                    const int nbi = (int)(e >> 20), py = (int)((e >> 10) & 1023), px = (int)(e & 1023);
                    const bool ok = nt > 0 && e != 0xffffffffu && nbi < nrem && py < yrem && px < xrem;
                    float4 v = load4<VEC>(ok ? gbase + g_rel[k] : (VEC ? g_zero_page : gy), ok, gco, g.Co);
                    v = mul4(v, *reinterpret_cast<const float4 *>(sA + (ok ? (n0 + nbi) * CV_BM : 0) + gc4 * 4));
                    if (ok) ma = amax4(ma, v);
                }
                {
                    const int k = (3 * sidx + 7 * kk + 2) % PMAX;
                    const unsigned e = p_pyx[k];
                    const int nbi = (int)(e >> 20), iy = iy0 + (int)((e >> 10) & 1023), ix = ix0 + (int)(e & 1023);
                    const bool ok = nt > 0 && e != 0xffffffffu && nbi < nrem && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                    float4 v = load4<VEC>(ok ? xbase + p_rel[k] : (VEC ? g_zero_page : x), ok, pci, g.Ci);
                    v = mul4(v, *reinterpret_cast<const float4 *>(sB + (ok ? (n0 + nbi) * CV_CK : 0) + pc4 * 4));
                    if (ok) mb = amax4(mb, v);
                }
            }
        }
        float *red = reinterpret_cast<float *>(smem);      // (operand buffers are not in use yet)
        ma = block_amax(ma, red);
        mb = block_amax(mb, red + 8);
        float sa, ua, sb, ub;
        cv_pow2_scale(ma, sa, ua);
        cv_pow2_scale(mb, sb, ub);
        punscale = cv_uniform(ua * ub);
        xsa = cv_uniform(sa);
        xsb = cv_uniform(sb);
        __syncthreads();                                    // every thread has read `red`
        for (int i = threadIdx.x; i < g.N * CV_BM; i += 256) sA[i] *= sa;
        for (int i = threadIdx.x; i < g.N * CV_CK; i += 256) sB[i] *= sb;
        __syncthreads();
    }

    // load_tile only ISSUES raw 16-byte loads (safe address for out-of-range items) and records a validity
    // bitmask; every consumer of the loaded registers (scale, zero-select, fp16 split) lives in store_tile,
    // which runs after the MFMA phase of the previous tile — so the loads stay in flight behind the MFMAs.
    unsigned okmask = 0;
    int st_n0 = 0;
    auto load_tile = [&](int tile) {
        int pt = tile;
        const int tx_i = pt % t.ntx;
        pt /= t.ntx;
        const int ty_i = pt % t.nty;
        const int tn_i = pt / t.nty;
        const int gx0 = tx_i << t.tw_log2, gy0 = ty_i << t.th_log2, n0 = tn_i * t.nbe;
        const int iy0 = gy0 * g.is + t.dymin, ix0 = gx0 * g.is + t.dxmin;
        const float *gbase = gy + (((int64_t)n0 * g.OH + gy0 * g.os + g.oy0) * g.OW + gx0 * g.os + g.ox0) * g.Co + gco;
        const float *xbase = x + (((int64_t)n0 * g.IH + iy0) * g.IW + ix0) * g.Ci + pci;
        const int nrem = g.N - n0, yrem = g.GH - gy0, xrem = g.GW - gx0;
        unsigned m = 0;
        st_n0 = n0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned e = g_pyx[k];
            const int nbi = (int)(e >> 20), py = (int)((e >> 10) & 1023), px = (int)(e & 1023);
            const bool ok = e != 0xffffffffu && nbi < nrem && py < yrem && px < xrem;
            m |= (ok ? 1u : 0u) << k;
            gq[k] = load4<VEC>(ok ? gbase + g_rel[k] : gy, ok, gco, g.Co);
        }
#pragma unroll
        for (int k = 0; k < PMAX; k++) {
            const unsigned e = p_pyx[k];
            const int nbi = (int)(e >> 20), iy = iy0 + (int)((e >> 10) & 1023), ix = ix0 + (int)(e & 1023);
            const bool ok = e != 0xffffffffu && nbi < nrem && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
            m |= (ok ? 1u : 0u) << (8 + k);
            pq[k] = load4<VEC>(ok ? xbase + p_rel[k] : x, ok, pci, g.Ci);
        }
        okmask = m;
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool ok = (okmask >> k) & 1u;
            float4 v = gq[k];
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sA + (ok ? (st_n0 + (int)(g_pyx[k] >> 20)) * CV_BM : 0) + gc4 * 4), hi, lo);
            *reinterpret_cast<uint2 *>(gh + g_lds[k]) = hi;
            if (SPLIT == 2) *reinterpret_cast<uint2 *>(gl + g_lds[k]) = lo;
        }
#pragma unroll
        for (int k = 0; k < PMAX; k++) {
            const bool ok = (okmask >> (8 + k)) & 1u;
            float4 v = pq[k];
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sB + (ok ? (st_n0 + (int)(p_pyx[k] >> 20)) * CV_CK : 0) + pc4 * 4), hi, lo);
            *reinterpret_cast<uint2 *>(ph + p_lds[k]) = hi;
            if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + p_lds[k]) = lo;
        }
    };

    if (!PIPE) {
        if (tile_begin < tile_end) load_tile(tile_begin);
        for (int tile = tile_begin; tile < tile_end; tile++) {
            __syncthreads();   // previous tile fully consumed
            if (!(t.debug & 6) || tile == tile_begin) store_tile();
            __syncthreads();
            if (tile + 1 < tile_end && !(t.debug & 10)) load_tile(tile + 1);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads ahead of the MFMA phase
            if (t.debug & 1) continue;
    #pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                f16x8 ahi[4], alo[4];
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
                        const int o0 = pp0 * 64 + wg_pswz(b_kg, pp0, t.pkb) * 16 + b_sub;
                        const int o1 = pp1 * 64 + wg_pswz(b_kg, pp1, t.pkb) * 16 + b_sub;
                        const f16x8 bhi = tr_read2(ph, o0, o1);
                        f16x8 blo;
                        if (SPLIT == 2) blo = tr_read2(pl, o0, o1);
    #pragma unroll
                        for (int i = 0; i < 4; i++) {
                            if (SPLIT == 2) {
                                acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi, acc[i][tt], 0, 0, 0);
                                acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo, acc[i][tt], 0, 0, 0);
                            }
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi, acc[i][tt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    } else {
        // ---- software-pipelined form (two operand buffers in LDS).  While the MFMAs of tile t run from buffer
        // t&1, the same wave converts the raw registers of tile t+1 into buffer (t+1)&1 — one staging item per
        // (k-half, tap) slot, placed in program order between the MFMA groups so its VALU / LDS-write work issues
        // in the shadow of the matrix pipe — and re-issues each register's global load for tile t+2 as soon as
        // the register is free.  One barrier per tile; a load has a whole tile period to land.
        constexpr int NITEM = 8 + PMAX, NSLOT = 2 * NT, IPS = (NITEM + NSLOT - 1) / NSLOT;
        const float *gbase = gy, *xbase = x;     // bases of the tile being LOADED
        int l_n0 = 0, l_nrem = 0, l_yrem = 0, l_xrem = 0, l_iy0 = 0, l_ix0 = 0;
        unsigned mask_ld = 0, mask_cv = 0;       // validity of the registers being loaded / converted
        int cv_n0 = 0;
        // a tile inside ONE image (every layer >= 8x8) has one gy / x scale vector per thread: kept in registers,
        // fetched from the LDS table once per tile instead of once per staged item
        const bool one_img = t.nbe == 1;
        float4 sa_cv = make_float4(1.f, 1.f, 1.f, 1.f), sb_cv = sa_cv;
        auto set_tile = [&](int tile) {
            int pt = tile;
            const int tx_i = pt % t.ntx;
            pt /= t.ntx;
            const int ty_i = pt % t.nty;
            const int tn_i = pt / t.nty;
            const int gx0 = tx_i << t.tw_log2, gy0 = ty_i << t.th_log2;
            l_n0 = tn_i * t.nbe;
            l_iy0 = gy0 * g.is + t.dymin;
            l_ix0 = gx0 * g.is + t.dxmin;
            gbase = gy + (((int64_t)l_n0 * g.OH + gy0 * g.os + g.oy0) * g.OW + gx0 * g.os + g.ox0) * g.Co + gco;
            xbase = x + (((int64_t)l_n0 * g.IH + l_iy0) * g.IW + l_ix0) * g.Ci + pci;
            l_nrem = g.N - l_n0;
            l_yrem = g.GH - gy0;
            l_xrem = g.GW - gx0;
            mask_ld = 0;
        };
        auto issue_item = [&](auto KC) {         // raw load of staging item K of the tile selected by set_tile
            constexpr int K = decltype(KC)::value;
            if constexpr (FAST && K < 8) {
                gq[K] = *reinterpret_cast<const float4 *>(gbase + g_rel[K]);
            } else if constexpr (FAST) {
                constexpr int P = K - 8;
                const unsigned e = p_pyx[P];
                const unsigned iy = (unsigned)(l_iy0 + (int)((e >> 10) & 1023)), ix = (unsigned)(l_ix0 + (int)(e & 1023));
                const bool ok = e != 0xffffffffu && iy < (unsigned)g.IH && ix < (unsigned)g.IW;
                pq[P] = *reinterpret_cast<const float4 *>(ok ? xbase + p_rel[P] : g_zero_page);
            } else if constexpr (K < 8) {
                const unsigned e = g_pyx[K];
                const int nbi = (int)(e >> 20), py = (int)((e >> 10) & 1023), px = (int)(e & 1023);
                const bool ok = e != 0xffffffffu && nbi < l_nrem && py < l_yrem && px < l_xrem;
                mask_ld |= (ok ? 1u : 0u) << K;
                gq[K] = load4<VEC>(ok ? gbase + g_rel[K] : gy, ok, gco, g.Co);
            } else {
                constexpr int P = K - 8;
                const unsigned e = p_pyx[P];
                const int nbi = (int)(e >> 20), iy = l_iy0 + (int)((e >> 10) & 1023), ix = l_ix0 + (int)(e & 1023);
                const bool ok = e != 0xffffffffu && nbi < l_nrem && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                mask_ld |= (ok ? 1u : 0u) << K;
                pq[P] = load4<VEC>(ok ? xbase + p_rel[P] : x, ok, pci, g.Ci);
            }
        };
        auto convert_item = [&](auto KC, unsigned char *buf, auto ONE) {   // raw register K -> scaled fp16 hi/lo in `buf`
            constexpr int K = decltype(KC)::value;
            constexpr bool ONE_IMG = decltype(ONE)::value;
            const bool ok = (mask_cv >> K) & 1u;
            if constexpr (FAST) {
                uint2 hi, lo;
                if constexpr (K < 8) {
                    const float4 v = gq[K];
                    if constexpr (FAST == 2) {     // modulated layers (G) carry per-(image, channel) scales (x block exponent)
                        if constexpr (ONE_IMG) split4v<SPLIT>(v, sa_cv, hi, lo);
                        else split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sA + (cv_n0 + (int)(g_pyx[K] >> 20)) * CV_BM + gc4 * 4), hi, lo);
                    } else split4s<SPLIT>(v, xsa, hi, lo);   // block exponent from an SGPR
                    *reinterpret_cast<uint2 *>(buf + g_lds[K]) = hi;
                    *reinterpret_cast<uint2 *>(buf + WG_GY_BYTES + g_lds[K]) = lo;
                } else {
                    const float4 v = pq[K - 8];
                    if constexpr (FAST == 2) {     // (an out-of-range item read the zero page: 0 * scale stays 0)
                        if constexpr (ONE_IMG) split4v<SPLIT>(v, sb_cv, hi, lo);
                        else split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sB + (cv_n0 + (int)((p_pyx[K - 8] >> 20) & 15)) * CV_CK + pc4 * 4), hi, lo);
                    } else split4s<SPLIT>(v, xsb, hi, lo);
                    *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + p_lds[K - 8]) = hi;
                    *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + (t.NPP + 1) * 64 + p_lds[K - 8]) = lo;
                }
            } else if constexpr (K < 8) {
                float4 v = gq[K];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                if constexpr (ONE_IMG) split4v<SPLIT>(v, sa_cv, hi, lo);
                else split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sA + (ok ? (cv_n0 + (int)(g_pyx[K] >> 20)) * CV_BM : 0) + gc4 * 4), hi, lo);
                *reinterpret_cast<uint2 *>(buf + g_lds[K]) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(buf + WG_GY_BYTES + g_lds[K]) = lo;
            } else {
                constexpr int P = K - 8;
                float4 v = pq[P];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                if constexpr (ONE_IMG) split4v<SPLIT>(v, sb_cv, hi, lo);
                else split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sB + (ok ? (cv_n0 + (int)(p_pyx[P] >> 20)) * CV_CK : 0) + pc4 * 4), hi, lo);
                *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + p_lds[P]) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + (t.NPP + 1) * 64 + p_lds[P]) = lo;
            }
        };
        auto for_items = [&](auto LO, auto fn) {  // fn(K) for the IPS items of one slot, K static
            constexpr int L = decltype(LO)::value;
            if constexpr (L < NITEM) fn(std::integral_constant<int, L>{});
            if constexpr (IPS > 1 && L + 1 < NITEM) fn(std::integral_constant<int, L + 1>{});
            if constexpr (IPS > 2 && L + 2 < NITEM) fn(std::integral_constant<int, L + 2>{});
            if constexpr (IPS > 3 && L + 3 < NITEM) fn(std::integral_constant<int, L + 3>{});
            if constexpr (IPS > 4 && L + 4 < NITEM) fn(std::integral_constant<int, L + 4>{});
            if constexpr (IPS > 5 && L + 5 < NITEM) fn(std::integral_constant<int, L + 5>{});
            if constexpr (IPS > 6 && L + 6 < NITEM) fn(std::integral_constant<int, L + 6>{});
            if constexpr (IPS > 7 && L + 7 < NITEM) fn(std::integral_constant<int, L + 7>{});
            if constexpr (IPS > 8 && L + 8 < NITEM) fn(std::integral_constant<int, L + 8>{});
            if constexpr (IPS > 9 && L + 9 < NITEM) fn(std::integral_constant<int, L + 9>{});
            static_assert(IPS <= 10, "items per slot");
        };
        auto for_slots = [&](auto fn) { static_for<0, NSLOT>(fn); };   // fn(slot), slot static
        // prologue: tile_begin -> buffer 0, raw registers <- tile_begin + 1
        if (tile_begin < tile_end) {
            set_tile(tile_begin);
            for_slots([&](auto S) { for_items(std::integral_constant<int, decltype(S)::value * IPS>{}, [&](auto K) { issue_item(K); }); });
            mask_cv = mask_ld;
            cv_n0 = l_n0;
            sa_cv = *reinterpret_cast<const float4 *>(sA + (cv_n0 < g.N ? cv_n0 : 0) * CV_BM + gc4 * 4);
            sb_cv = *reinterpret_cast<const float4 *>(sB + (cv_n0 < g.N ? cv_n0 : 0) * CV_CK + pc4 * 4);
            for_slots([&](auto S) {
                for_items(std::integral_constant<int, decltype(S)::value * IPS>{}, [&](auto K) { convert_item(K, smem, std::false_type{}); });
            });
            set_tile(tile_begin + 1 < tile_end ? tile_begin + 1 : tile_end - 1);
            for_slots([&](auto S) { for_items(std::integral_constant<int, decltype(S)::value * IPS>{}, [&](auto K) { issue_item(K); }); });
        }
        __syncthreads();
        // two copies of the tile loop (tile within one image / spanning images): a run-time select per item would
        // put a branch between the loads and cost the exact vmcnt counts
        auto run = [&](auto ONE) {
            for (int tile = tile_begin; tile < tile_end; tile++) {
                const int cur = (tile - tile_begin) & 1;
                const unsigned char *bgh = smem + cur * bufsz, *bgl = bgh + WG_GY_BYTES;
                const unsigned char *bph = bgh + 2 * WG_GY_BYTES, *bpl = bph + (t.NPP + 1) * 64;
                unsigned char *nbuf = smem + (cur ^ 1) * bufsz;
                mask_cv = mask_ld;                   // the registers hold tile + 1
                cv_n0 = l_n0;
                sa_cv = *reinterpret_cast<const float4 *>(sA + (cv_n0 < g.N ? cv_n0 : 0) * CV_BM + gc4 * 4);
                sb_cv = *reinterpret_cast<const float4 *>(sB + (cv_n0 < g.N ? cv_n0 : 0) * CV_CK + pc4 * 4);
                // No branch around the staging work: past the end of the range the last tile is simply staged again
                // (never consumed).  With straight-line VMEM traffic hipcc's waitcnt pass keeps exact counts
                // (vmcnt(NITEM-1) per converted register); any branch here makes it fall back to vmcnt(0) per slot.
                set_tile(tile + 2 < tile_end ? tile + 2 : tile_end - 1);
                f16x8 ahi[4], alo[4];
                for_slots([&](auto SC) {
                    constexpr int S = decltype(SC)::value, kk = S / NT, tt = S % NT;
                    if constexpr (tt == 0) {
    #pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int cb = (wm * 64 + i * 16 + p * 4) * 2;
                            const int o0 = a_row[kk][0] + (cb ^ a_key[kk][0]);
                            const int o1 = a_row[kk][1] + (cb ^ a_key[kk][1]);
                            ahi[i] = tr_read2(bgh, o0, o1);
                            if (SPLIT == 2) alo[i] = tr_read2(bgl, o0, o1);
                        }
                    }
                    for_items(std::integral_constant<int, S * IPS>{}, [&](auto K) { convert_item(K, nbuf, ONE); });
                    for_items(std::integral_constant<int, S * IPS>{}, [&](auto K) { issue_item(K); });
                    if (tt < g.ntaps) {
                        const int ts = tt < g.ntaps ? tt : 0;
                        const int toff = (g.dy[ts] - t.dymin) * t.PW + (g.dx[ts] - t.dxmin);
                        const int pp0 = pbase[kk][0] + toff, pp1 = pbase[kk][1] + toff;
                        const int o0 = pp0 * 64 + wg_pswz(b_kg, pp0, t.pkb) * 16 + b_sub;
                        const int o1 = pp1 * 64 + wg_pswz(b_kg, pp1, t.pkb) * 16 + b_sub;
                        const f16x8 bhi = tr_read2(bph, o0, o1);
                        f16x8 blo;
                        if (SPLIT == 2) blo = tr_read2(bpl, o0, o1);
    #pragma unroll
                        for (int i = 0; i < 4; i++) {
                            if (SPLIT == 2) {
                                acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi, acc[i][tt], 0, 0, 0);
                                acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo, acc[i][tt], 0, 0, 0);
                            }
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi, acc[i][tt], 0, 0, 0);
                        }
                    }
                });
                __syncthreads();   // buffer `cur` fully consumed, buffer cur^1 fully written
            }
        };
        if (one_img) run(std::true_type{});
        else run(std::false_type{});
    }

    // ---- partial tile -> workspace [split][cot][chunk][tap][32 ci][128 co]: a lane's 4 accumulator
    // registers are 4 consecutive co of one ci, so the [ci][co] order makes every store a float4
    float *wsb = ws + (((int64_t)split * t.ncot + cot) * t.nchunks + chunk) * g.ntaps * (CV_BM * CV_CK);
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
        if (tt < g.ntaps) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int co = wm * 64 + i * 16 + G * 4;
                const int ci = wn * 16 + (lane & 15);
                *reinterpret_cast<float4 *>(wsb + (tt * CV_CK + ci) * CV_BM + co) =
                    make_float4(acc[i][tt][0] * punscale, acc[i][tt][1] * punscale, acc[i][tt][2] * punscale, acc[i][tt][3] * punscale);
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
        const int r = (int)(i & 127);              // co within tile (fastest in the partial layout)
        const int k = (int)((i >> 7) & 31);        // ci within chunk
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

// Split-K plan of the weight gradient: blocks = co-tiles x chunks x splits, one 400-register block per CU, so the grid
// runs in waves of 256 blocks.  Minimise  waves x (tiles per block + fixed block cost)  plus a small charge per split
// for the partial tiles the second stage has to read (measured: 512x512 @64^2, batch 4 runs best as 4 splits of 64
// tiles = one wave, not 8 x 32).  Multiples of 8 splits get the XCD-aware block mapping and win ties.
static void wgrad_plan(const rick_conv_geom *g, ConvTiling *t, int *nsplit, int *tps) {
    make_tiling(g, WG_TILE, t);
    const int ntiles = t->ntx * t->nty * t->ntn;
    const int cc = t->ncot * t->nchunks;
    int best_tps = ntiles, best_cost = 1 << 30;
    const int smax = ntiles < 64 ? ntiles : 64;
    for (int s = 1; s <= smax; s++) {
        const int tp = cdiv(ntiles, s), se = cdiv(ntiles, tp);
        const int waves = cdiv(cc * se, 256);
        const int cost = 2 * waves * (tp + 8) + se - ((se & 7) == 0 ? 1 : 0);
        if (cost < best_cost) {
            best_cost = cost;
            best_tps = tp;
        }
    }
    *tps = best_tps;
    *nsplit = cdiv(ntiles, best_tps);
}

extern "C" int64_t rick_conv_wgrad_workspace_bytes(const rick_conv_geom *g) {
    if (check_geom(g)) return -1;
    ConvTiling t;
    int nsplit, tps;
    wgrad_plan(g, &t, &nsplit, &tps);
    return (int64_t)nsplit * t.ncot * t.nchunks * g->ntaps * CV_BM * CV_CK * 4;
}

template <int NT, int SPLIT, bool VEC, int PMAX, bool PIPE, int FAST = 0>
static void launch_wgrad_k(const float *x, const float *gy, float *ws, const float *ascale, const float *bscale,
                           const rick_conv_geom *g, const ConvTiling &t, int nsplit, int tps, size_t lds, hipStream_t st) {
    const unsigned nwg = (unsigned)(nsplit * t.ncot * t.nchunks);
    (void)hipFuncSetAttribute((const void *)conv_wgrad_kernel<NT, SPLIT, VEC, PMAX, PIPE, FAST>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_wgrad_kernel<NT, SPLIT, VEC, PMAX, PIPE, FAST>), dim3(nwg), dim3(256), lds, st, x, gy, ws, ascale,
                       bscale, *g, t, nsplit, tps);
}

static size_t wgrad_lds_bytes(const rick_conv_geom *g, const ConvTiling &t, bool pipe) {
    const size_t buf = 2 * WG_GY_BYTES + 2 * (size_t)(t.NPP + 1) * 64;
    return (pipe ? 2 : 1) * buf + (size_t)((t.NPP + 3) & ~3) * 4 + (size_t)g->N * (CV_BM + CV_CK) * 4;
}
// The software-pipelined form needs two operand buffers in LDS; production path only (fp16x3, vector loads).
static bool wgrad_use_pipe(const rick_conv_geom *g, const ConvTiling &t) {
    static const int off = ablation_env("RICK_WGRAD_NOPIPE", 0);
    return !off && g->split == 2 && ((g->Ci | g->Co) & 3) == 0 && wgrad_lds_bytes(g, t, true) <= 160 * 1024;
}

template <int NT>
static void launch_wgrad(const float *x, const float *gy, float *ws, const float *ascale, const float *bscale,
                         const rick_conv_geom *g, const ConvTiling &t, int nsplit, int tps, hipStream_t st) {
    const bool vec = ((g->Ci | g->Co) & 3) == 0;
    const bool small = t.NPP <= 4 * 32;
    const bool pipe = wgrad_use_pipe(g, t);
    const size_t lds = wgrad_lds_bytes(g, t, pipe);
    // FAST: position grid an exact multiple of the tile, full 128 x 32 channel blocks
    const bool fast = pipe && !(g->Co % CV_BM) && !(g->Ci % CV_CK) && !(g->GH & ((1 << t.th_log2) - 1)) &&
                      !(g->GW & ((1 << t.tw_log2) - 1)) && !(g->N % t.nbe) && (t.nb == t.nbe);
    const bool scaled = ascale != nullptr || bscale != nullptr;
    // ONE chain: exactly one kernel per call
    if (g->split == 1 && vec) launch_wgrad_k<NT, 1, true, 12, false>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (!vec) launch_wgrad_k<NT, 2, false, 12, false>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (fast && small && scaled) launch_wgrad_k<NT, 2, true, 4, true, 2>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (fast && small) launch_wgrad_k<NT, 2, true, 4, true, 1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (fast && scaled) launch_wgrad_k<NT, 2, true, 12, true, 2>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (fast) launch_wgrad_k<NT, 2, true, 12, true, 1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (small && pipe) launch_wgrad_k<NT, 2, true, 4, true>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (small) launch_wgrad_k<NT, 2, true, 4, false>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else if (pipe) launch_wgrad_k<NT, 2, true, 12, true>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
    else launch_wgrad_k<NT, 2, true, 12, false>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
}

extern "C" int rick_conv_wgrad_f32(const float *x, const float *gy, float *gw, int64_t s_co, int64_t s_ci, int64_t s_t,
                                   const float *ascale, const float *bscale, const rick_conv_geom *g, int accumulate,
                                   void *workspace, void *stream) {
    if (!x || !gy || !gw || !workspace || check_geom(g)) return RICK_EINVAL;
    if (g->ntaps > 9) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)(ascale ? ascale : x) | (uintptr_t)(bscale ? bscale : x)) % 16) return RICK_EINVAL;
    ConvTiling t;
    int nsplit, tps;
    wgrad_plan(g, &t, &nsplit, &tps);
    if (wgrad_lds_bytes(g, t, false) > 160 * 1024 || t.NPP > 12 * 32 || t.PH > 1023 || t.PW > 1023) return RICK_EINVAL;
    if (g->split == 1 && (((g->Ci | g->Co) & 3) != 0)) return RICK_EINVAL;   // plain-fp16 option: vector path only
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    if (g->ntaps == 1) launch_wgrad<1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
    else if (g->ntaps <= 4) launch_wgrad<4>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
    else launch_wgrad<9>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
    const int64_t per_split = (int64_t)t.ncot * t.nchunks * g->ntaps * CV_BM * CV_CK;
    int64_t nb = cdiv64(per_split, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ws, gw, s_co, s_ci, s_t, g->Co, g->Ci,
                       t.ncot, t.nchunks, g->ntaps, nsplit, g->alpha, accumulate, *g);
    RICK_LAUNCH_STATUS();
}
