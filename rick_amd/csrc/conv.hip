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
#include "conv_tiling.h"

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

#define PK_SAMPLES 8     // per thread: up to 2048 strided samples of the tensor (one block: the kernel is latency-bound)

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
    cv_fp16_saturate();
    float satm = 0.f;
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
        satm = fmaxf(satm, fabsf(v));
        unsigned short h, l;
        split1(v, h, l, split);
        const int64_t base = (i >> 12) * (CV_WSTEP_BYTES / 2);
        const int off = r * 32 + cv_swz(k >> 3, r) * 8 + (k & 7);
        packed[base + off] = h;
        packed[base + CV_WTILE_BYTES / 2 + off] = l;
    }
    cv_sat_report(satm);
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
#define PK_NS_MAX 16      // tap slices staged through LDS (RICK_MAX_TAPS); more: the per-thread tap walk
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const rick_pack_desc *__restrict__ descs, int n, int split) {
    __shared__ float sw[256 * PK_NS_MAX];
    int d = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_begin) d = i;
    const rick_pack_desc ds = descs[d];
    cv_fp16_saturate();
    const int nchunks = (ds.Ci + CV_CK - 1) / CV_CK;
    const float pscale =
        reinterpret_cast<const float *>((const unsigned char *)ds.packed + packed_tile_bytes(ds.Co, ds.Ci, ds.nslices))[1];
    const int blk = (int)blockIdx.x - ds.blk_begin;      // 16 blocks per 128 x 32 tile: 8 rows x 32 k each
    float satm = 0.f;
    if (ds.nslices <= PK_NS_MAX) {
        // A block's 8 rows x 32 k x nslices values are read in MEMORY order (coalesced runs of 32 * nslices floats for a
        // [O, I, kh, kw] parameter, 8 * nslices for its transposed view) into LDS as [r][k][slice]; then one thread per
        // (slice, row, 16-byte granule) converts 8 consecutive k and writes the granule's hi and lo halves as two 16-byte
        // stores.  (One thread per (row, k) walking the slices read 4 bytes of every 36 per lane and wrote 2-byte pieces:
        // 203 us per network.)  Same value per element — w * scale * 2^e, same conversion — so the image is byte-identical.
        // (Round 5: four of these blocks per launched block — 32 rows staged at once, 4 x longer runs, a quarter of the blocks —
        // measured SLOWER, same box: 171 / 158 us against 146 / 137 us for the generator / discriminator; the launch lives on
        // having many small blocks in flight.  Not kept.)
        const int ns = ds.nslices, tile = blk >> 4, rg = blk & 15;
        const int chunk = tile % nchunks, cot = tile / nchunks;
        const int co0 = cot * CV_BM + rg * 8, ci0 = chunk * CV_CK;
        const bool k_fast = llabs(ds.s_ci) <= llabs(ds.s_co);
        for (int e = threadIdx.x; e < 256 * ns; e += 256) {
            const int sl = e % ns, q = e / ns;
            const int r = k_fast ? q >> 5 : q & 7, k = k_fast ? q & 31 : q >> 3;
            const int co = co0 + r, ci = ci0 + k;
            float v = 0.f;
            if (co < ds.Co && ci < ds.Ci) v = ds.w[co * ds.s_co + ci * ds.s_ci + sl * ds.s_t] * ds.scale * pscale;
            sw[(r * 32 + k) * ns + sl] = v;
        }
        __syncthreads();
        unsigned char *base = (unsigned char *)ds.packed + (int64_t)tile * ns * CV_WSTEP_BYTES;
        for (int it = threadIdx.x; it < 32 * ns; it += 256) {
            const int sl = it >> 5, r = (it >> 2) & 7, gq = it & 3;
            const float *src = sw + (r * 32 + gq * 8) * ns + sl;
            unsigned short h[8], l[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const float v = src[jj * ns];
                satm = fmaxf(satm, fabsf(v));
                split1(v, h[jj], l[jj], split);
            }
            const int row = rg * 8 + r;
            unsigned char *dst = base + (int64_t)sl * CV_WSTEP_BYTES + row * 64 + cv_swz(gq, row) * 16;
            *reinterpret_cast<uint4 *>(dst) = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16),
                                                         h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
            *reinterpret_cast<uint4 *>(dst + CV_WTILE_BYTES) = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16),
                                                                          l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
        }
        cv_sat_report(satm);
        return;
    }
    const int i = blk * 256 + threadIdx.x;   // over (cotile, chunk, r, k)
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
        satm = fmaxf(satm, fabsf(v));
        unsigned short h, l;
        split1(v, h, l, split);
        dst[(int64_t)sl * (CV_WSTEP_BYTES / 2)] = h;
        dst[(int64_t)sl * (CV_WSTEP_BYTES / 2) + CV_WTILE_BYTES / 2] = l;
    }
    cv_sat_report(satm);
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

#define IG_PMAX 10        // non-DEEP: patch float4 per thread prefetched in registers (NPP <= 320 pixels)
#define IG_PSET_DEEP 6     // DEEP: two register sets of 6 (NPP <= 192 pixels), chunks prefetched two ahead
#define IG_DEEP_NPP (32 * IG_PSET_DEEP)

// Stride-2 inputs need a patch of ~4x the position tile; a 64-position block (each wave 64 co x 32 positions)
// keeps it at ~37 KB so two blocks still fit per CU.
static size_t igemm_lds_bytes(const ConvTiling &t, bool has_iscale) {
    (void)has_iscale;   // the scale table is always present (filled with 1 when the conv has no input scale)
    return 2 * CV_WSTEP_BYTES + 2 * (size_t)(t.NPP + 1) * 64 + (size_t)((t.NPP + 3) & ~3) * 4 +
           (size_t)t.nbe * t.cps * CV_CK * 4 + 64;      // (+64: the block-exponent reduction scratch)
}

static int igemm_tile_positions(const rick_conv_geom *g) { return (g->is >= 2 && g->ntaps > 1) ? 64 : CV_BN; }

// Eight-wave form (igemm_body, NW = 8): 3x3 stride-1 geometries whose 16 x 16 position tiles fill the chip by themselves.
// LDS: 4 weight slots + two patch buffers + tables (one block per CU).
#define IG_W8_POS 256
static size_t igemm_w8_lds_bytes(const ConvTiling &t) {
    return 4 * CV_WSTEP_BYTES + 4 * (size_t)(t.NPP + 1) * 64 + (size_t)((t.NPP + 3) & ~3) * 4 + (size_t)t.nbe * t.cps * CV_CK * 4 + 64;
}
// Measured and NOT the default (round 6, profiles/r06_w8_microbench.txt, batch 8, both forms interleaved in one process): the
// eight-wave block gains only where the K loop is long AND the operands are converted in the kernel — 512 -> 512 @64^2 fp32
// operands +4.7 % forward / +4.2 % data gradient, 512 -> 256 +6.1 % — TIES on split images (470.3 vs 469.9 TFLOP/s) and LOSES on
// short K loops (128 -> 128 @256^2: -10 ... -15 %; 256 -> 256 @128^2: -2.4 ... +0.9 %): one block per CU has no neighbour block
// to cover its prologue and its 128 KB epilogue.  The counters say why the tie (profiles/r06_w8_pmc.txt, 512 -> 512 @64^2 split
// image): four-wave 72.1 % MFMA-busy at 1.74 GHz, eight-wave 64.3 % at 1.84 GHz — the chip trades clock against matrix-pipe
// occupancy at constant power, so a schedule that keeps the pipe fuller buys nothing.  In bench.py mode 1 (fp32 operands with
// >= 16 channel chunks) measured 168.80 vs 168.79 images/s.  Mode 0 ships; the form stays behind rick_conv_tuning with its
// parity tests (tests/test_gpu_ops.py: fp64 at 2e-6, bit-equal to the four-wave form on split images).
// Eight-wave STRIDE-2 form (igemm_body, WDMA = 5): 3x3 stride-2 geometries with Co % 256 == 0 whose 16 x 8 position tiles fill
// the chip; t->ncot counts 256-channel block columns.  rick_conv_tuning RICK_TUNE_IGEMM_S2W8: 0 never, 1 channel chunks >= 8, 2 all.
static size_t igemm_s2w8_lds_bytes(const ConvTiling &t) {
    return 4 * CV_WSTEP_BYTES + 2 * (size_t)(t.NPP + 1) * 64 + (size_t)((t.NPP + 3) & ~3) * 4 + (size_t)t.nbe * t.cps * CV_CK * 4 + 64;
}
static bool igemm_s2w8_plan(const rick_conv_geom *g, ConvTiling *t) {
    const int mode = rick_internal_tune(RICK_TUNE_IGEMM_S2W8);
    if (!mode || g->split != 2 || g->ntaps != 9 || g->is != 2 || (g->Ci & 3) || (g->Co & 255) || g->GH < 8 || g->GW < 16) return false;
    if (make_tiling(g, CV_BN, t)) return false;
    if (t->nb != 1 || t->NPP > 64 * 9 || t->nchunks < (mode == 1 ? 8 : 4)) return false;
    t->ncot = g->Co / 256;
    if (t->ntx * t->nty * t->ntn * t->ncot < rick_internal_tune(RICK_TUNE_IGEMM_W8_MINBLK)) return false;
    if ((int64_t)g->N * g->IH * g->IW * g->Ci * 4 >= (1LL << 32) - 256) return false;
    return igemm_s2w8_lds_bytes(*t) <= 160 * 1024 && t->PH <= 1023 && t->PW <= 1023;
}

static bool igemm_w8_plan(const rick_conv_geom *g, ConvTiling *t, bool pkx) {
    const int w8 = rick_internal_tune(RICK_TUNE_IGEMM_W8), w8_minblk = rick_internal_tune(RICK_TUNE_IGEMM_W8_MINBLK);
    if (w8 == 1 && (pkx || g->Ci < 16 * CV_CK)) return false;
    if (!w8 || g->split != 2 || g->ntaps != 9 || g->is != 1 || (g->Ci & 3) || g->GH < 16 || g->GW < 16) return false;
    if (make_tiling(g, IG_W8_POS, t)) return false;
    if (t->nb != 1 || t->NPP > 64 * IG_PSET_DEEP || t->nchunks < 4) return false;
    if (t->ntx * t->nty * t->ntn * t->ncot < w8_minblk) return false;     // (short grids keep the 128-position blocks and split-K)
    if ((int64_t)g->N * g->IH * g->IW * g->Ci * 4 >= (1LL << 32) - 256) return false;   // (32-bit byte offsets into the input)
    return igemm_w8_lds_bytes(*t) <= 160 * 1024 && t->PH <= 1023 && t->PW <= 1023;
}

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


// ==========================================================================================
// Second stage of a split-K launch, one float4 (4 consecutive output channels of one pixel): alpha, output scale and the fused
// bias / noise / LeakyReLU tail on the summed partials.  ONE definition for the stand-alone reduce kernel and the in-launch
// fix-up (the block that arrives last for a tile, cv_splitk_arrive): both forms produce the same bits.
__device__ __forceinline__ float4 splitk_finish4(float4 s, unsigned n, unsigned co, unsigned gy, unsigned gx, const rick_conv_geom &g,
                                                 const float *__restrict__ oscale, const rick_conv_epilogue &epi, float nwv) {
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
    return v;
}

// ==========================================================================================
// Forward / data-gradient kernel.

typedef unsigned int igemm_u32x4 __attribute__((ext_vector_type(4)));
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

// PKX: the input is a split image (conv_common.h): `iscale` points at its header {2^e, 2^-e, ..}, per-channel input scales
// are already folded in by the producer, a staging item is the 16 bytes {hi x 4 | lo x 4} of 4 channels, copied to LDS as it is.
// NW = 8 (WDMA = 4): the eight-wave form of the straight-line 3x3 stride-1 loop — 512 threads on a 128 co x 256 position block,
// waves 2 (co) x 4 (positions), ONE block per CU (two waves per SIMD like two of the four-wave blocks, but one weight tile serves
// all eight waves, the 16 x 16 tile's 18 x 18 patch replaces two 10 x 18 ones, and the LDS this frees holds a fourth weight slot
// and a SECOND patch buffer: the B fragments and the first A row of k-step ks + 1 are read from LDS during k-step ks, so the
// MFMAs behind a barrier start at once instead of behind an LDS round trip).  Same MFMA order per accumulator as the four-wave
// form (bit-identical on split images).
template <int SPLIT, bool VEC, bool DEEP, int NJ, int NT, int WDMA = 0, bool PKX = false, int NW = 4>
__device__ __forceinline__ void igemm_body(const float *__restrict__ x, const unsigned char *__restrict__ wpk,
                                           float *__restrict__ out, const float *__restrict__ iscale,
                                           const float *__restrict__ oscale, float *__restrict__ ws,
                                           const rick_conv_geom &g, const ConvTiling &t, const int bid, const int nwg,
                                           const rick_conv_epilogue &epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cv_fp16_saturate();
    constexpr int NTHR = NW * 64;
    static_assert(NW == 4 || (NW == 8 && (WDMA == 4 || WDMA == 5) && NJ == 4 && NT == 9 && SPLIT == 2 && VEC && !DEEP), "eight-wave forms");
    static_assert((WDMA == 4 || WDMA == 5) == (NW == 8), "WDMA = 4 / 5 are the eight-wave forms");
    constexpr int COT = WDMA == 5 ? 2 : 1;                  // 128-channel co tiles per block (WDMA = 5: 256 co x 128 positions, stride 2)
    unsigned char *wbuf = smem;                               // [2][16 KB]
    unsigned char *ph = smem + (WDMA == 5 ? 4 : WDMA >= 3 ? WDMA : 2) * CV_WSTEP_BYTES;   // [NPP + 1][64 B]  (WDMA = 3 / 4: a ring of 3 / 4 weight tiles; 5: 2 slots of 2 tiles)
    unsigned char *pl = ph + (t.NPP + 1) * 64;                // (+1: spare row for out-of-patch items)
    const int pbuf_bytes = 2 * (t.NPP + 1) * 64;              // WDMA = 4: two patch buffers [hi | lo], buffer b at ph + b * pbuf_bytes
    unsigned *ptab = reinterpret_cast<unsigned *>(pl + (t.NPP + 1) * 64 + (WDMA == 4 ? pbuf_bytes : 0));   // [NPP]
    float *sct = reinterpret_cast<float *>(ptab + ((t.NPP + 3) & ~3));                // [nbe][cps * 32] input scales
    build_patch_table(ptab, t, NTHR);

    const int lid = xcd_remap(bid, nwg);
    const int ncot128 = (g.Co + CV_BM - 1) / CV_BM;          // co tiles of the packed weight image (t.ncot counts BLOCK columns)
    const int npos_tiles = t.ntx * t.nty * t.ntn;
    // Block order.  Position tile fastest gives each co tile its own XCDs: a co tile's weights stay in few L2s, but every input
    // patch crosses the fabric ncot times.  Stride-2 launches (37 KB patches, 4x the input per output position) run the co tile
    // fastest instead, so that the ncot blocks of a position tile fetch its patch into ONE L2: 609 -> 377 MB on 256 -> 512
    // @129^2, +2-3 %.  Stride 1 keeps the old order (co tile fastest: 163 -> 153 MB and -0.2 % in bench.py: the 9.4 MB of
    // packed weights, then needed by every XCD, take what the input saves), and so do split-K launches (-7 % otherwise).
    int pt, split, cot;
    if (t.nsplit > 1 || g.is < 2 || (t.debug & 32)) {
        pt = lid % npos_tiles;
        split = (lid / npos_tiles) % t.nsplit;
        cot = lid / (npos_tiles * t.nsplit);
    } else {
        cot = lid % t.ncot;
        pt = (lid / t.ncot) % npos_tiles;
        split = lid / (t.ncot * npos_tiles);
    }
    const int c_begin = split * t.cps;
    const int c_end = c_begin + t.cps < t.nchunks ? c_begin + t.cps : t.nchunks;
    const int pt_lin = pt;
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
    static_assert(!PKX || (SPLIT == 2 && VEC), "split-image input: fp16x3, vector path");
    if (!PKX && iscale) {
        for (int i = threadIdx.x; i < t.nbe * cspan; i += NTHR) {
            const int nbi = i / cspan, c = c_begin * CV_CK + (i - nbi * cspan);
            sct[i] = (n0 + nbi < g.N && c < g.Ci) ? iscale[(int64_t)(n0 + nbi) * g.Ci + c] : 0.f;
        }
    }
    __syncthreads();   // patch table and scale table complete
    // operand exponent of this block (conv_common.h), set by block_exponent() below once the first chunk is in registers
    float xscale = 1.f, unscale = 1.f;
    float satm = 0.f;          // largest |scaled operand| this thread converts (cv_sat_report)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = WDMA == 5 ? wave >> 1 : NW == 8 ? wave >> 2 : wave >> 1, wn = WDMA == 5 ? wave & 1 : NW == 8 ? wave & 3 : wave & 1;
    const int l15 = lane & 15, kg = lane >> 4;

    int a_off[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (WDMA == 5 ? wm & 1 : wm) * 64 + i * 16 + l15;       // (WDMA = 5: wave rows 0..255 = tile wm >> 1, row (wm & 1) * 64 + ..)
        a_off[i] = (WDMA == 5 ? (wm >> 1) * CV_WSTEP_BYTES : 0) + row * 64 + cv_swz(kg, row) * 16;
    }
    int pb[NJ];   // NJ = position tiles of 16 per wave: 4 (128-position block) or 2 (64-position block, stride-2 input)
    const int tw_mask = (1 << t.tw_log2) - 1, th_mask = (1 << t.th_log2) - 1;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int pos = wn * (NJ * 16) + j * 16 + l15;
        const int px = pos & tw_mask, py = (pos >> t.tw_log2) & th_mask;
        int nbi = pos >> (t.tw_log2 + t.th_log2);
        nbi = nbi < t.nbe ? nbi : t.nbe - 1;     // image slots beyond nbe are masked in the epilogue; keep LDS reads in range
        pb[j] = (nbi * t.PH + py * g.is) * t.PW + (g.is == 2 ? px : px * g.is);   // (stride 2: de-interleaved columns, conv_tiling.h)
    }

    // ---- patch staging state.  The tile is fixed for the block, so validity, source offset (relative to
    // the tile-origin pointer) and LDS offset of each of this thread's patch items are computed once.
    // DEEP (patches of <= 160 pixels: 1x1 convs and the parity classes of transposed convs, whose chunks
    // last only 1-4 k-steps): the IG_PMAX register slots form TWO sets of 5 and chunks are prefetched two
    // ahead, so a load has two chunks' worth of MFMAs to land instead of one.
    // register prefetch slots per set: the unrolled 128-position form only serves patches of <= 192 pixels
    // (eight waves: 64 pixels per item round, 6 rounds cover the 18 x 18 patch of a 16 x 16 tile)
    constexpr int PSET = WDMA == 5 ? 9 : (DEEP || (NT > 0 && NJ == 4)) ? IG_PSET_DEEP : IG_PMAX;   // (WDMA = 5: 33 x 17 patch, 64 pixels per round)
    constexpr int PREGS = DEEP ? 2 * IG_PSET_DEEP : PSET;
    const int p_items = t.NPP * 8;
    const int c4 = threadIdx.x & 7;                      // NTHR % 8 == 0: same channel quad for all items
    const float *xt = x + (((int64_t)n0 * g.IH + iy0) * g.IW + ix0) * g.Ci + c4 * 4;
    int p_rel[PSET], p_lds[PSET], p_sc[PSET];
    unsigned p_ok = 0;
    float4 pq[PREGS];
#pragma unroll
    for (int k = 0; k < PSET; k++) {
        const int pix = (threadIdx.x >> 3) + (NTHR / 8) * k;
        p_rel[k] = 0;
        p_sc[k] = 0;
        // items past the patch land in a spare row behind it, so the LDS writes need no guard
        const int slot = pix < t.NPP ? cv_patch_slot(pix, ptab[pix], t, g.is) : t.NPP;
        p_lds[k] = slot * 64 + cv_swz(c4 >> 1, slot) * 16 + (c4 & 1) * 8;
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
                if constexpr (PKX) {      // the item already IS {hi x 4 | lo x 4}
                    const float4 pv = pq[S * PSET + k];
                    *reinterpret_cast<float2 *>(ph + p_lds[k]) = make_float2(pv.x, pv.y);
                    *reinterpret_cast<float2 *>(pl + p_lds[k]) = make_float2(pv.z, pv.w);
                    continue;
                }
                const bool ok = (cur_ok[S] >> k) & 1u;
                float4 v = pq[S * PSET + k];
                if (!VEC && !ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                if constexpr (decltype(ISC)::value)
                    split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sct + p_sc[k] + (chunk - c_begin) * CV_CK), hi, lo, satm);
                else split4s<SPLIT>(v, xscale, hi, lo, satm);
                *reinterpret_cast<uint2 *>(ph + p_lds[k]) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + p_lds[k]) = lo;
            }
        };
        if (!PKX && iscale) items(std::true_type{});
        else items(std::false_type{});
        if (!DEEP && !PKX) {
            // patches larger than IG_PMAX*32 pixels (stride-2 geometry): remaining items, synchronously
            for (int it = threadIdx.x + NTHR * PSET; it < p_items; it += NTHR) {
                const int pix = it >> 3;
                const unsigned e = ptab[pix];
                const int n = n0 + (int)(e >> 20), iy = iy0 + (int)((e >> 10) & 1023), ix = ix0 + (int)(e & 1023);
                const int ci = chunk * CV_CK + c4 * 4;
                const bool ok = n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW && ci < g.Ci;
                float4 v = load4<VEC>(ok ? x + (((int64_t)n * g.IH + iy) * g.IW + ix) * g.Ci + ci : x, ok, ci, g.Ci);
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 hi, lo;
                if (iscale) split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sct + (int)(e >> 20) * cspan + (chunk - c_begin) * CV_CK + c4 * 4), hi, lo, satm);
                else split4s<SPLIT>(v, xscale, hi, lo, satm);      // (the scale table only exists with an input scale)
                const int slot = cv_patch_slot(pix, e, t, g.is);
                const int off = slot * 64 + cv_swz(c4 >> 1, slot) * 16 + (c4 & 1) * 8;
                *reinterpret_cast<uint2 *>(ph + off) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + off) = lo;
            }
        }
    };
    // ---- operand exponent of this block: amax of |input scale * x| over the first channel chunk's patch, which
    // issue_patch has just put into the registers of set 0 (no extra loads: 32 channels x the whole patch), and over
    // samples of the block's other chunks, reduced over the block -> x * 2^e (conv_common.h).  Called between
    // issue_patch(c_begin) and commit_patch(c_begin).
    auto block_exponent = [&]() {
        if constexpr (PKX) {   // the image's own exponent (header) x the packed weights' (trailer); nothing to measure
            unscale = cv_uniform(iscale[1] * *reinterpret_cast<const float *>(wpk + (int64_t)ncot128 * t.nchunks * g.nslices * CV_WSTEP_BYTES));
            return;
        }
        float *red = sct + t.nbe * cspan;                             // 16 floats behind the scale table, used for nothing else
#ifdef RICK_ABLATION
        if (t.debug & 8) {   // timing-only ablation (RICK_CONV_DEBUG=8): no exponent (values are wrong for data far from 1)
            unscale = *reinterpret_cast<const float *>(wpk + (int64_t)ncot128 * t.nchunks * g.nslices * CV_WSTEP_BYTES);
            return;
        }
#endif
        // the block's OTHER chunks: 8 float4 per thread (2 048 x 4 values), every thread group of 8 starting at its own
        // chunk so that all chunks are visited — channel groups of very different magnitude (pruned filters, dead
        // channels of a gradient) are the case a first-chunk exponent gets wrong: 1e-4 x smaller first 32 channels were
        // enough for inf.  Issued before the first chunk is reduced: one memory latency for both.
        const int ncl = c_end - c_begin;
        float4 sv[8];
        if (ncl > 1) {
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) {
                const int k = (sidx * 5 + 1) % PSET;
                const int chunk = c_begin + 1 + (sidx + (int)(threadIdx.x >> 3)) % (ncl - 1);
                const int ci = chunk * CV_CK + c4 * 4;
                const bool ok = ((p_ok >> k) & 1u) && ci < g.Ci;
                sv[sidx] = load4<VEC>(ok ? xt + p_rel[k] + chunk * CV_CK : (VEC ? g_zero_page : x), ok, ci, g.Ci);
                if (!VEC && !ok) sv[sidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < PSET; k++) {
            float4 v = pq[k];
            if (!VEC && !((cur_ok[0] >> k) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iscale) v = mul4(v, *reinterpret_cast<const float4 *>(sct + p_sc[k]));
            m = amax4(m, v);
        }
        if (ncl > 1) {
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) {
                const int k = (sidx * 5 + 1) % PSET;
                const int chunk = c_begin + 1 + (sidx + (int)(threadIdx.x >> 3)) % (ncl - 1);
                float4 v = sv[sidx];
                if (iscale) v = mul4(v, *reinterpret_cast<const float4 *>(sct + p_sc[k] + (chunk - c_begin) * CV_CK));
                m = amax4(m, v);
            }
        }
        m = block_amax(m, red);
        float xs, xu;
        cv_pow2_scale(m, xs, xu);
        xscale = cv_uniform(xs);
        // packed-weight exponent (trailer of the packed image)
        unscale = cv_uniform(xu * *reinterpret_cast<const float *>(wpk + (int64_t)ncot128 * t.nchunks * g.nslices * CV_WSTEP_BYTES));
        if (iscale) {                   // fold 2^e into the scale table
            for (int i = threadIdx.x; i < t.nbe * cspan; i += NTHR) sct[i] *= xscale;
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
    const unsigned char *wbase = wpk + (int64_t)cot * COT * t.nchunks * g.nslices * CV_WSTEP_BYTES;
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

    if constexpr (WDMA == 4) {
        // ---- eight-wave form (NW = 8): 4-slot weight ring by LDS-DMA, two patch buffers, fragments prefetched across the barrier.
        // State at the top of k-step ks (behind its barrier): weight tiles ks and ks + 1 have landed, tile ks + 2 is in flight;
        // the patch of the current chunk is complete in buffer (chunk - c_begin) & 1; the fragments of the FIRST quadrant of
        // k-step ks (A rows 0-1, B columns 0-1: 8 of its 16 ds_read_b128) are in registers, read during k-step ks - 1.
        // K-step ks then
        //   issues the DMA of tile ks + 3 into the slot of tile ks - 1 (its last reader finished before the barrier),
        //   issues one patch item of the NEXT chunk (taps 0..5) and converts / stores the one issued two k-steps ago (taps 2..7)
        //     into the OTHER patch buffer — the next chunk is complete before the barrier that opens tap 8, whose prefetch reads it,
        //   walks the wave's 4 x 4 MFMA tiles as four quadrants (rows 0-1 | 2-3) x (columns 0-1 | 2-3) in the order
        //     Q00 Q01 Q11 Q10, reading one operand half ahead of each (B columns 2-3 under Q00, A rows 2-3 under Q01, the first
        //     quadrant of k-step ks + 1 under Q10): at most 64 fragment registers live, like the four-wave form, and the 12 MFMAs
        //     behind a barrier never wait for LDS,
        //   waits for tile ks + 2 (counted vmcnt) and joins the barrier.
        static_assert(NT >= 3 && PSET + 2 < NT, "patch items: issued at taps 0 .. PSET-1, stored at taps 2 .. PSET+1 < NT-1");
        // patch items without per-item registers: item k of this thread is patch pixel (tid >> 3) + 64 k, whose LDS offset is the
        // item-0 offset + 4096 k (64 pixels on: same swizzle key) and whose source offset comes from the pixel table in LDS
        const int pix0 = threadIdx.x >> 3;
        const int lds0 = pix0 * 64 + cv_swz(c4 >> 1, pix0) * 16 + (c4 & 1) * 8;
        const int lds_spare = t.NPP * 64 + cv_swz(c4 >> 1, t.NPP) * 16 + (c4 & 1) * 8;
        auto item_lds = [&](int k) { return pix0 + 64 * k < t.NPP ? lds0 + 4096 * k : lds_spare; };
        // loads through a buffer descriptor over the whole input with range-checked 32-bit byte offsets: an item in the zero
        // padding takes an offset beyond the descriptor and the hardware returns zeros (no branch, no 64-bit address arithmetic)
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x), 0, (unsigned)((int64_t)g.N * g.IH * g.IW * g.Ci * 4), 0x00020000);
        const unsigned xt_bytes = (unsigned)((((n0 * g.IH + iy0) * g.IW + ix0) * g.Ci + c4 * 4) * 4);   // (modular: may lie "before" x)
        auto issue_item = [&](auto KC, int chunk) {
            constexpr int K = decltype(KC)::value;
            const int pix = pix0 + 64 * K;
            const unsigned e = ptab[pix < t.NPP ? pix : 0];                             // (one image per tile: nbi = 0)
            const unsigned rel = (unsigned)(((int)((e >> 10) & 1023) * g.IW + (int)(e & 1023)) * g.Ci * 4);
            const bool ok = (cur_ok[0] >> K) & 1u;
            const igemm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (xt_bytes + rel + (unsigned)chunk * (CV_CK * 4)) | (ok ? 0u : 0xfffffff0u), 0, 0);   // (OR, not a select of the sum: hipcc would sink the table read into a branch)
            pq[K] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        };
        auto issue_wdma8 = [&](int wchunk, int wtap, int slot) {
            const unsigned char *src = wbase + ((int64_t)wchunk * g.nslices + g.wt[wtap]) * CV_WSTEP_BYTES;
            unsigned char *dst = wbuf + slot * CV_WSTEP_BYTES;
#pragma unroll
            for (int q = 0; q < 2; q++)
                __builtin_amdgcn_global_load_lds((cv_gbl_u8 *)(src + (q * NTHR + threadIdx.x) * 16),
                                                 (cv_lds_u8 *)(dst + (q * NTHR + wave * 64) * 16), 16, 0, 0);
        };
        const int plo = (t.NPP + 1) * 64;                    // lo half of a patch buffer
        struct Frag2 { f16x8 h[2], l[2]; };                  // two 16-row fragments, hi and lo
        auto read_a = [&](const unsigned char *wb, int half) {
            Frag2 f;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f.h[i] = *reinterpret_cast<const f16x8 *>(wb + a_off[half * 2 + i]);
                f.l[i] = *reinterpret_cast<const f16x8 *>(wb + CV_WTILE_BYTES + a_off[half * 2 + i]);
            }
            return f;
        };
        auto read_b = [&](const unsigned char *pbase, int tap, int half) {
            Frag2 f;
            const int toff = (g.dy[tap] - t.dymin) * t.PW + (g.dx[tap] - t.dxmin);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int pp = pb[half * 2 + j] + toff;
                const int off = pp * 64 + cv_swz(kg, pp) * 16;
                f.h[j] = *reinterpret_cast<const f16x8 *>(pbase + off);
                f.l[j] = *reinterpret_cast<const f16x8 *>(pbase + plo + off);
            }
            return f;
        };
        auto quad = [&](const Frag2 &a, const Frag2 &b, auto IH, auto JH) {
            constexpr int i0 = decltype(IH)::value * 2, j0 = decltype(JH)::value * 2;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l[i], b.h[j], acc[i0 + i][j0 + j], 0, 0, 0);
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[i], b.l[j], acc[i0 + i][j0 + j], 0, 0, 0);
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[i], b.h[j], acc[i0 + i][j0 + j], 0, 0, 0);
                }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        auto mainloop = [&](auto ISC) {
            auto commit_item = [&](auto KC, int chunk, unsigned char *pbase) {
                constexpr int K = decltype(KC)::value;
                unsigned char *dh = pbase + item_lds(K);
                if constexpr (PKX) {      // the item already IS {hi x 4 | lo x 4}
                    const float4 pv = pq[K];
                    *reinterpret_cast<float2 *>(dh) = make_float2(pv.x, pv.y);
                    *reinterpret_cast<float2 *>(dh + plo) = make_float2(pv.z, pv.w);
                } else {
                    uint2 hi, lo;
                    if constexpr (decltype(ISC)::value)
                        split4v<SPLIT>(pq[K], *reinterpret_cast<const float4 *>(sct + c4 * 4 + (chunk - c_begin) * CV_CK), hi, lo, satm);
                    else split4s<SPLIT>(pq[K], xscale, hi, lo, satm);
                    *reinterpret_cast<uint2 *>(dh) = hi;
                    *reinterpret_cast<uint2 *>(dh + plo) = lo;
                }
            };
            // prologue: tiles 0..2, the first chunk's patch (all items at once: the block exponent needs them), first quadrant of k-step 0
            issue_wdma8(c_begin, 0, 0);
            issue_wdma8(c_begin, 1, 1);
            issue_wdma8(c_begin, 2, 2);
            issue_patch(c_begin, S0{});
            block_exponent();
            static_for<0, PSET>([&](auto KC) { commit_item(KC, c_begin, ph); });
            cv_wait_vm<0>();
            cv_lds_barrier();
            Frag2 a0 = read_a(wbuf, 0), b0 = read_b(ph, 0, 0);
            int ks0 = 0;
            for (int chunk = c_begin; chunk < c_end; chunk++, ks0 += NT) {
                const int cnext = chunk + 1 < c_end ? chunk + 1 : chunk;
                const int lb = (chunk - c_begin) & 1;
                unsigned char *pcur = ph + lb * pbuf_bytes, *pnxt = ph + (lb ^ 1) * pbuf_bytes;
                auto kstep = [&](auto TC) {
                    constexpr int tap = decltype(TC)::value;
                    const int ks = ks0 + tap;
                    issue_wdma8(tap + 3 < NT ? chunk : cnext, (tap + 3) % NT, (ks + 3) & 3);
                    asm volatile("" ::: "memory");       // (keeps the VM queue in source order: the counted wait below relies on it)
                    if constexpr (tap == 0) cur_ok[0] = cnext * CV_CK + c4 * 4 < g.Ci ? p_ok : 0u;
                    if constexpr (tap < PSET) issue_item(std::integral_constant<int, tap < PSET ? tap : 0>{}, cnext);
                    if constexpr (tap >= 2 && tap - 2 < PSET) commit_item(std::integral_constant<int, tap >= 2 ? tap - 2 : 0>{}, cnext, pnxt);
                    const unsigned char *wb = wbuf + (ks & 3) * CV_WSTEP_BYTES;
                    const unsigned char *wbn = wbuf + ((ks + 1) & 3) * CV_WSTEP_BYTES;
                    const Frag2 b1 = read_b(pcur, tap, 1);
                    quad(a0, b0, I0{}, I0{});
                    const Frag2 a1 = read_a(wb, 1);
                    quad(a0, b1, I0{}, I1{});
                    quad(a1, b1, I1{}, I1{});
                    const Frag2 na = read_a(wbn, 0);
                    const Frag2 nb = read_b(tap + 1 < NT ? pcur : pnxt, (tap + 1) % NT, 0);
                    quad(a1, b0, I1{}, I0{});
                    a0 = na;
                    b0 = nb;
                    // tile ks + 2 (issued at the top of k-step ks - 1) has landed when at most the VM operations issued behind
                    // it are outstanding: that k-step's patch item, this k-step's two DMA instructions and patch item
                    constexpr int n1 = (tap >= 1 && tap - 1 < PSET) ? 1 : 0, n0 = tap < PSET ? 1 : 0;
                    cv_wait_vm<2 + n0 + n1>();
                    cv_lds_barrier();
                };
                static_for<0, NT>(kstep);
            }
            cv_wait_vm<0>();       // (the ring runs three tiles ahead: nothing may land in LDS after the block has left)
        };
        if (!PKX && iscale) mainloop(std::true_type{});
        else mainloop(std::false_type{});
    } else if constexpr (WDMA == 5) {
        // ---- eight-wave STRIDE-2 form (round 6, verdict item 2b): 256 co x 128 positions, waves 4 (co) x 2 (positions) of 64 x 64.
        // The four-wave stride-2 block is 128 co x 64 positions (each wave 64 x 32: 2 MFMAs per ds_read_b128, its 37 KB patch staged
        // per 128 co); here a wave has the stride-1 shape (3 MFMAs per read) and the 72 KB patch of a 16 x 8 tile is staged ONCE for
        // 256 co.  LDS: 2 slots x 2 weight tiles (64 KB, LDS-DMA, tile pair of k-step ks + 1 issued behind the barrier of ks) + the
        // patch (one buffer; the next chunk's 9 items per thread wait in registers, one issued per tap, stored at the chunk boundary).
        static_assert(NT == 9 && PSET == NT, "one patch item per tap");
        const int pix0 = threadIdx.x >> 3;
        auto item_lds = [&](int k) {
            const int pix = pix0 + 64 * k;
            const int slot = pix < t.NPP ? cv_patch_slot(pix, ptab[pix < t.NPP ? pix : 0], t, 2) : t.NPP;
            return slot * 64 + cv_swz(c4 >> 1, slot) * 16 + (c4 & 1) * 8;
        };
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x), 0, (unsigned)((int64_t)g.N * g.IH * g.IW * g.Ci * 4), 0x00020000);
        const unsigned xt_bytes = (unsigned)((((n0 * g.IH + iy0) * g.IW + ix0) * g.Ci + c4 * 4) * 4);
        auto issue_item = [&](auto KC, int chunk) {
            constexpr int K = decltype(KC)::value;
            const int pix = pix0 + 64 * K;
            const unsigned e = ptab[pix < t.NPP ? pix : 0];
            const unsigned rel = (unsigned)(((int)((e >> 10) & 1023) * g.IW + (int)(e & 1023)) * g.Ci * 4);
            const bool ok = (cur_ok[0] >> K) & 1u;
            const igemm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (xt_bytes + rel + (unsigned)chunk * (CV_CK * 4)) | (ok ? 0u : 0xfffffff0u), 0, 0);
            pq[K] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        };
        const int64_t cot_stride = (int64_t)t.nchunks * g.nslices * CV_WSTEP_BYTES;      // packed bytes of one 128-channel co tile
        auto issue_wdma5 = [&](int wchunk, int wtap, int slot) {
            const unsigned char *src = wbase + ((int64_t)wchunk * g.nslices + g.wt[wtap]) * CV_WSTEP_BYTES;
            unsigned char *dst = wbuf + slot * (2 * CV_WSTEP_BYTES);
#pragma unroll
            for (int q = 0; q < 4; q++)          // pieces 0..1023: co tile 2 cot, 1024..2047: co tile 2 cot + 1
                __builtin_amdgcn_global_load_lds((cv_gbl_u8 *)(src + (q >> 1) * cot_stride + ((q & 1) * NTHR + threadIdx.x) * 16),
                                                 (cv_lds_u8 *)(dst + (q * NTHR + wave * 64) * 16), 16, 0, 0);
        };
        struct Frag2 { f16x8 h[2], l[2]; };
        auto read_a = [&](const unsigned char *wb, int half) {
            Frag2 f;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f.h[i] = *reinterpret_cast<const f16x8 *>(wb + a_off[half * 2 + i]);
                f.l[i] = *reinterpret_cast<const f16x8 *>(wb + CV_WTILE_BYTES + a_off[half * 2 + i]);
            }
            return f;
        };
        auto read_b = [&](int tap, int half) {
            Frag2 f;
            const int toff = (g.dy[tap] - t.dymin) * t.PW + cv_patch_col(g.dx[tap] - t.dxmin, t.PW, 2);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int pp = pb[half * 2 + j] + toff;
                const int off = pp * 64 + cv_swz(kg, pp) * 16;
                f.h[j] = *reinterpret_cast<const f16x8 *>(ph + off);
                f.l[j] = *reinterpret_cast<const f16x8 *>(pl + off);
            }
            return f;
        };
        auto quad = [&](const Frag2 &a, const Frag2 &b, auto IH, auto JH) {
            constexpr int i0 = decltype(IH)::value * 2, j0 = decltype(JH)::value * 2;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l[i], b.h[j], acc[i0 + i][j0 + j], 0, 0, 0);
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[i], b.l[j], acc[i0 + i][j0 + j], 0, 0, 0);
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h[i], b.h[j], acc[i0 + i][j0 + j], 0, 0, 0);
                }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        auto mainloop = [&](auto ISC) {
            auto commit_item = [&](auto KC, int chunk) {
                constexpr int K = decltype(KC)::value;
                const int off = item_lds(K);
                if constexpr (PKX) {
                    const float4 pv = pq[K];
                    *reinterpret_cast<float2 *>(ph + off) = make_float2(pv.x, pv.y);
                    *reinterpret_cast<float2 *>(pl + off) = make_float2(pv.z, pv.w);
                } else {
                    uint2 hi, lo;
                    if constexpr (decltype(ISC)::value)
                        split4v<SPLIT>(pq[K], *reinterpret_cast<const float4 *>(sct + c4 * 4 + (chunk - c_begin) * CV_CK), hi, lo, satm);
                    else split4s<SPLIT>(pq[K], xscale, hi, lo, satm);
                    *reinterpret_cast<uint2 *>(ph + off) = hi;
                    *reinterpret_cast<uint2 *>(pl + off) = lo;
                }
            };
            issue_wdma5(c_begin, 0, 0);
            issue_patch(c_begin, S0{});
            block_exponent();
            static_for<0, PSET>([&](auto KC) { commit_item(KC, c_begin); });
            cv_wait_vm<0>();
            cv_lds_barrier();
            int ks0 = 0;
            for (int chunk = c_begin; chunk < c_end; chunk++, ks0 += NT) {
                const int cnext = chunk + 1 < c_end ? chunk + 1 : chunk;
                auto kstep = [&](auto TC) {
                    constexpr int tap = decltype(TC)::value;
                    const int ks = ks0 + tap;
                    if (tap > 0 || ks0 > 0) {       // (compile-time for tap > 0; the block's very first k-step follows the prologue's barrier)
                        cv_wait_vm<1>();            // tile pair ks (issued one k-step ago); behind it: that k-step's patch item
                        cv_lds_barrier();           // all pieces landed; slot (ks + 1) & 1 (read in k-step ks - 1) is free
                    }
                    issue_wdma5(tap + 1 < NT ? chunk : cnext, (tap + 1) % NT, (ks + 1) & 1);
                    asm volatile("" ::: "memory");
                    if constexpr (tap == 0) cur_ok[0] = cnext * CV_CK + c4 * 4 < g.Ci ? p_ok : 0u;
                    issue_item(TC, cnext);
                    const unsigned char *wb = wbuf + (ks & 1) * (2 * CV_WSTEP_BYTES);
                    const Frag2 a0 = read_a(wb, 0), b0 = read_b(tap, 0);
                    const Frag2 b1 = read_b(tap, 1);
                    quad(a0, b0, I0{}, I0{});
                    const Frag2 a1 = read_a(wb, 1);
                    quad(a0, b1, I0{}, I1{});
                    quad(a1, b1, I1{}, I1{});
                    quad(a1, b0, I1{}, I0{});
                    if constexpr (tap == NT - 1) {     // chunk boundary: every wave is done with the patch
                        cv_lds_barrier();
                        static_for<0, PSET>([&](auto KC) { commit_item(KC, cnext); });
                    }
                };
                static_for<0, NT>(kstep);
            }
            cv_wait_vm<0>();
        };
        if (!PKX && iscale) mainloop(std::true_type{});
        else mainloop(std::false_type{});
    } else if constexpr (NT > 0) {
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
                    const int toff = (g.dy[tap] - t.dymin) * t.PW + cv_patch_col(g.dx[tap] - t.dxmin, t.PW, g.is);
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
                const int toff = (g.dy[tap] - t.dymin) * t.PW + cv_patch_col(g.dx[tap] - t.dxmin, t.PW, g.is);
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
    float out_amax = 0.f;      // running max |out| of this thread's stores (epi.amax)
    // split image of the output (the next modulated convolution's operand, its style folded in): epi.split_out
    float so_scale = 1.f, so_amax = 0.f;
    if (epi.split_out) {
        const cv_split_hdr h = cv_split_header(epi.split_bound, nullptr, epi.split_coef);
        if (bid == 0 && threadIdx.x == 0) *reinterpret_cast<cv_split_hdr *>(epi.split_hdr) = h;
        so_scale = cv_uniform(h.scale);
    }
    const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        ws, 0, t.nsplit > 1 ? (unsigned)((int64_t)t.nsplit * g.N * g.GH * g.GW * g.Co * 4) : 0u, 0x00020000);
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
                    const int co = cot * (COT * CV_BM) + wm * 64 + i * 16 + kg * 4;
                    if (co >= g.Co) continue;
                    // (partial sums leave the block without its operand exponents: blocks of one output tile may differ)
                    if (covec && t.tickets) {      // in-launch fix-up: WRITE-THROUGH (sc1) stores, so that no block pays a release fence
                        const f32x4 pv = {acc[i][j][0] * unscale, acc[i][j][1] * unscale, acc[i][j][2] * unscale, acc[i][j][3] * unscale};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(igemm_u32x4, pv), ws_rsrc,
                                                               (unsigned)((wrow + co - ws) * 4), 0, 16 /* sc1 */);
                    } else if (covec) {
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
                    const int co = cot * (COT * CV_BM) + wm * 64 + i * 16 + kg * 4;
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
                    const int co = cot * (COT * CV_BM) + wm * 64 + i * 16 + kg * 4;
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
                        out_amax = amax4(out_amax, v);
                        if (epi.split_out) {
                            if (epi.split_scale) {
                                const float4 cs = *reinterpret_cast<const float4 *>(epi.split_scale + (int64_t)n * g.Co + co);
                                v = make_float4(v.x * cs.x, v.y * cs.y, v.z * cs.z, v.w * cs.w);
                            }
                            so_amax = amax4(so_amax, v);
                            cv_split_store4(reinterpret_cast<unsigned char *>(epi.split_out) + opix * g.Co * 4, co, v, so_scale);
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int co = cot * (COT * CV_BM) + wm * 64 + i * 16 + kg * 4;
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
                out_amax = fmaxf(out_amax, fabsf(v[0]));
                if (co + 1 < g.Co) { orow[co + 1] = v[1]; out_amax = fmaxf(out_amax, fabsf(v[1])); }
                if (co + 2 < g.Co) { orow[co + 2] = v[2]; out_amax = fmaxf(out_amax, fabsf(v[2])); }
                if (co + 3 < g.Co) { orow[co + 3] = v[3]; out_amax = fmaxf(out_amax, fabsf(v[3])); }
            }
        }
    };
    const bool has_ep = epi.bias || epi.noise || epi.act;   // (host: only with Co % 4 == 0)
    if (oscale && has_ep) epilogue(std::true_type{}, std::true_type{});
    else if (oscale) epilogue(std::true_type{}, std::false_type{});
    else if (has_ep) epilogue(std::false_type{}, std::true_type{});
    else epilogue(std::false_type{}, std::false_type{});
    cv_sat_report(satm);
    if (epi.split_out) cv_sat_check(so_amax, so_scale);
    if (epi.amax && t.nsplit == 1) cv_amax_publish(out_amax, epi.amax, reinterpret_cast<float *>(smem));     // (split-K: the second stage measures)
    if (VEC && t.nsplit > 1 && t.tickets) {
        // ---- split-K fix-up inside the launch (rick_conv_tuning RICK_TUNE_SPLITK_FUSED; host: Co % 4 == 0, 16-byte aligned
        // buffers).  The block that arrives last for this output tile sums the nsplit partial tiles in split order — the order
        // and arithmetic of igemm_splitk_reduce_kernel — and applies the second stage; 8 positions x 32 float4 of the 128-channel
        // tile per pass (512-byte runs per position).  MEASURED SLOWER than the second-stage launch and switched off (round 6,
        // profiles/r06_splitk_fused.txt): with a release fence per block +7 ... +36 us per launch (every block writes back its
        // XCD's L2), with write-through stores and no release still +8 (4^2) ... +17 us (8^2, 16^2): ONE block reads the tile's
        // nsplit partial tiles (0.5 - 1 MB) at the 60 - 100 GB/s a single block gets across XCDs, where the second-stage kernel
        // spreads the same bytes over the chip in 6 us.
        if (cv_splitk_arrive<true>(t.tickets + cot * npos_tiles + pt_lin, t.nsplit, reinterpret_cast<unsigned *>(smem))) {
            const size_t per4 = (size_t)g.N * g.GH * g.GW * g.Co / 4;
            const float4 *ws4 = reinterpret_cast<const float4 *>(ws);
            float ram = 0.f;
            for (int it = threadIdx.x; it < NJ * 32 * 32; it += NTHR) {
                const int pos = it >> 5, co = cot * CV_BM + (it & 31) * 4;
                const int px = pos & tw_mask, py = (pos >> t.tw_log2) & th_mask, nbi = pos >> (t.tw_log2 + t.th_log2);
                const int n = n0 + nbi, gy = gy0 + py, gx = gx0 + px;
                if (co >= g.Co || nbi >= t.nbe || n >= g.N || gy >= g.GH || gx >= g.GW) continue;
                const float4 *src = ws4 + ((((size_t)n * g.GH + gy) * g.GW + gx) * g.Co + co) / 4;
                const float4 s = cv_sum_splits4(src, (int64_t)per4, t.nsplit);
                const float4 v = splitk_finish4(s, (unsigned)n, (unsigned)co, (unsigned)gy, (unsigned)gx, g, oscale, epi, nwv);
                *reinterpret_cast<float4 *>(out + (((int64_t)n * g.OH + gy * g.os + g.oy0) * g.OW + gx * g.os + g.ox0) * g.Co + co) = v;
                ram = amax4(ram, v);
            }
            if (epi.amax) cv_amax_publish(ram, epi.amax, reinterpret_cast<float *>(smem) + 16);
        }
    }
}

template <int SPLIT, bool VEC, bool DEEP, int NJ, int NT, int WDMA = 0, bool PKX = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void conv_igemm_kernel(const float *__restrict__ x,
                                                            const unsigned char *__restrict__ wpk,
                                                            float *__restrict__ out, const float *__restrict__ iscale,
                                                            const float *__restrict__ oscale, float *__restrict__ ws,
                                                            const rick_conv_geom g, const ConvTiling t,
                                                            const rick_conv_epilogue epi) {
    igemm_body<SPLIT, VEC, DEEP, NJ, NT, WDMA, PKX, NW>(x, wpk, out, iscale, oscale, ws, g, t, blockIdx.x, gridDim.x, epi);
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
    const rick_conv_epilogue none = {nullptr, nullptr, nullptr, 1, 0, 0.f, 1.f, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr};
    igemm_body<SPLIT, VEC, true, 4, 0>(x, wpk, out, iscale, oscale, ws + m.ws_off[c], m.g[c], m.t[c], (int)blockIdx.x - start,
                                       m.blk_end[c] - start, none);
}

// out[n, pix(gy,gx), co] = alpha * oscale[n,co] * sum_s ws[s][n,gy,gx][co]
// VEC4 (Co % 4 == 0): one float4 of 4 consecutive channels per thread and split, 32-bit index math (the guard in
// check_geom keeps N*OH*OW*Co below 2^31).
template <bool VEC4, bool AMAX = false>
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ out,
                                                                  const float *__restrict__ oscale, rick_conv_geom g,
                                                                  int nsplit, rick_conv_epilogue epi) {
    constexpr int W = VEC4 ? 4 : 1;
    const float nwv = epi.noise ? epi.noise_w[0] : 0.f;
    const unsigned per = (unsigned)g.N * g.GH * g.GW * g.Co;
    const unsigned cow = (unsigned)g.Co / W;
    float ram = 0.f;
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
            const float4 s = cv_sum_splits4(reinterpret_cast<const float4 *>(ws) + iw, (int64_t)(per / 4), nsplit);
            const float4 v = splitk_finish4(s, n, co, gy, gx, g, oscale, epi, nwv);
            *reinterpret_cast<float4 *>(out + o) = v;
            if (AMAX) ram = amax4(ram, v);
        } else {
            float s = 0.f;
            for (int sp = 0; sp < nsplit; sp++) s += ws[(size_t)sp * per + iw];
            s *= g.alpha;
            if (oscale) s *= oscale[(size_t)n * g.Co + co];
            out[o] = s;
            if (AMAX) ram = fmaxf(ram, fabsf(s));
        }
    }
    if (AMAX) {
        __shared__ float red[4];
        cv_amax_publish(ram, epi.amax, red);
    }
}

static const rick_conv_epilogue kNoEpilogue = {nullptr, nullptr, nullptr, 1, 0, 0.f, 1.f, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr};

static void launch_splitk_reduce(const float *ws, float *out, const float *oscale, const rick_conv_geom *g, int nsplit,
                                 hipStream_t st, const rick_conv_epilogue &epi = kNoEpilogue) {
    const int64_t per = (int64_t)g->N * g->GH * g->GW * g->Co;
    const bool vec = (g->Co & 3) == 0 && (((uintptr_t)ws | (uintptr_t)out | (uintptr_t)(oscale ? oscale : out)) % 16) == 0;
    int64_t nb = cdiv64(vec ? per / 4 : per, 256);
    if (nb > 8192) nb = 8192;
    rick_conv_epilogue e2 = kNoEpilogue;
    e2.amax = epi.amax;
    if (vec && epi.amax) hipLaunchKernelGGL((igemm_splitk_reduce_kernel<true, true>), dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, nsplit, epi);
    else if (vec) hipLaunchKernelGGL((igemm_splitk_reduce_kernel<true, false>), dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, nsplit, epi);
    else if (epi.amax) hipLaunchKernelGGL((igemm_splitk_reduce_kernel<false, true>), dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, nsplit, e2);
    else hipLaunchKernelGGL((igemm_splitk_reduce_kernel<false, false>), dim3((unsigned)nb), dim3(256), 0, st, ws, out, oscale, *g, nsplit, e2);
}


extern "C" int64_t rick_conv_igemm_workspace_bytes(const rick_conv_geom *g) {
    if (check_geom(g)) return -1;
    ConvTiling t;
    // (the four-wave plan's need, whatever rick_conv_tuning says now: callers cache this per geometry; the eight-wave form uses none)
    if (make_tiling(g, igemm_tile_positions(g), &t)) return -1;
    igemm_plan_split(&t, g->ntaps);
    return t.nsplit > 1 ? (int64_t)t.nsplit * g->N * g->GH * g->GW * g->Co * 4 : 0;
}

template <int SPLIT, bool VEC, bool DEEP, int NJ, int NT = 0, int WDMA = 0, bool PKX = false, int NW = 4>
static void launch_igemm_k(unsigned nwg, size_t lds, hipStream_t st, const float *x, const unsigned char *wp, float *out,
                           const float *iscale, const float *oscale, float *ws, const rick_conv_geom *g,
                           const ConvTiling &t, const rick_conv_epilogue &epi) {
    RICK_LDS160_ONCE((conv_igemm_kernel<SPLIT, VEC, DEEP, NJ, NT, WDMA, PKX, NW>));
    hipLaunchKernelGGL((conv_igemm_kernel<SPLIT, VEC, DEEP, NJ, NT, WDMA, PKX, NW>), dim3(nwg), dim3(NW * 64), lds + (WDMA == 3 ? CV_WSTEP_BYTES : 0), st,
                       x, wp, out, iscale, oscale, ws, *g, t, epi);
}

template <int SPLIT, bool VEC>
static int launch_igemm(unsigned nwg, size_t lds, hipStream_t st, const float *x, const unsigned char *wp, float *out,
                        const float *iscale, const float *oscale, float *ws, const rick_conv_geom *g,
                        const ConvTiling &t, const rick_conv_epilogue &epi, bool pkx = false) {
    // production path (fp16x3, vector loads), 3x3 layers with a full grid: the straight-line 9-tap k-loop.  Its
    // loop body carries 0.65 non-MFMA VALU + 0.2 SALU instructions per MFMA against 1.4 + 1.2 (stride 1) / 2.2 + 2.3
    // (stride 2) of the generic tap loop (SQ_INSTS_* counters, profiles/r02_pmc_conv.json): +3..8 % on the Ci >= 256
    // stride-1 layers, +7 % at Ci = 128, +7..13 % on the stride-2 layers (tools/abl_u9.sh).
    static const int no_unroll = ablation_env("RICK_IGEMM_NOUNROLL", 0);
    static const int u9_minchunks = ablation_env("RICK_U9_MINCHUNKS", 4), u9_s2 = ablation_env("RICK_U9_S2", 1);
    static const int u9_split = ablation_env("RICK_U9_SPLIT", 4);    // split-K launches too when a split keeps >= 4 chunks (+5..13 %)
    const bool u9 = SPLIT == 2 && VEC && g->ntaps == 9 && !no_unroll && !(t.debug & 7) && igemm_tile_positions(g) == CV_BN &&
                    t.NPP <= IG_DEEP_NPP && (t.nsplit == 1 || (u9_split && t.cps >= u9_split)) && t.cps >= u9_minchunks;
    const bool u9s2 = SPLIT == 2 && VEC && g->ntaps == 9 && u9_s2 && igemm_tile_positions(g) == 64 && t.NPP <= 32 * IG_PMAX &&
                      (t.nsplit == 1 || (u9_split && t.cps >= u9_split)) && t.cps >= u9_minchunks;
    // (stride 2: a k-step has half the MFMAs per weight tile, so the register-staged weight store weighed twice as much:
    // +11..17 % with the tiles by LDS-DMA into the two existing slots)
    if constexpr (SPLIT == 2 && VEC) {
        if (pkx) {   // split-image input (`iscale` = its header): the three forms that carry the FLOPs
            if (u9s2) launch_igemm_k<2, true, false, 2, 9, 2, true>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
            else if (igemm_tile_positions(g) == 64) {          // stride-2 launches with short K per block (split-K): generic tap loop
                if (t.NPP > 32 * IG_PMAX) return RICK_EINVAL;
                launch_igemm_k<2, true, false, 2, 0, 0, true>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
            }
            else if (u9 && lds + CV_WSTEP_BYTES <= 80 * 1024) launch_igemm_k<2, true, false, 4, 9, 3, true>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
            else if (t.NPP <= IG_DEEP_NPP) launch_igemm_k<2, true, true, 4, 0, 0, true>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
            else return RICK_EINVAL;
            return 0;
        }
    }
    if (pkx) return RICK_EINVAL;
    if (u9s2 && ablation_env("RICK_WDMA2", 1)) launch_igemm_k<2, true, false, 2, 9, 2>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (u9s2) launch_igemm_k<2, true, false, 2, 9>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (igemm_tile_positions(g) == 64) launch_igemm_k<SPLIT, VEC, false, 2>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    // weights by LDS-DMA into a 3-slot ring (+3..5 % on the Ci >= 256 layers; keeps two blocks per CU)
    else if (u9 && ablation_env("RICK_WDMA", 1) && lds + CV_WSTEP_BYTES <= 80 * 1024)
        launch_igemm_k<2, true, false, 4, 9, 3>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (u9) launch_igemm_k<2, true, false, 4, 9>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else if (t.NPP <= IG_DEEP_NPP) launch_igemm_k<SPLIT, VEC, true, 4>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    else launch_igemm_k<SPLIT, VEC, false, 4>(nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi);
    return 0;
}

extern "C" int rick_conv_igemm_act_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                       const float *oscale, const rick_conv_geom *g, const rick_conv_epilogue *epilogue,
                                       void *workspace, void *stream);

extern "C" int rick_conv_igemm_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                   const float *oscale, const rick_conv_geom *g, void *workspace, void *stream) {
    return rick_conv_igemm_act_f32(x, packed_w, out, iscale, oscale, g, nullptr, workspace, stream);
}

static int igemm_run(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                     const rick_conv_geom *g, const rick_conv_epilogue *epilogue, void *workspace, void *stream, bool pkx);

extern "C" int rick_conv_igemm_act_f32(const float *x, const void *packed_w, float *out, const float *iscale,
                                       const float *oscale, const rick_conv_geom *g, const rick_conv_epilogue *epilogue,
                                       void *workspace, void *stream) {
    return igemm_run(x, packed_w, out, iscale, oscale, g, epilogue, workspace, stream, false);
}

extern "C" int rick_conv_igemm_split_f32(const void *x_split, const float *x_hdr, const void *packed_w, float *out,
                                         const float *oscale, const rick_conv_geom *g, const rick_conv_epilogue *epilogue,
                                         void *workspace, void *stream) {
    if (!x_hdr || !g || (g->Ci & 31) || g->split != 2) return RICK_EINVAL;
    return igemm_run((const float *)x_split, packed_w, out, x_hdr, oscale, g, epilogue, workspace, stream, true);
}

// 1 when rick_conv_igemm_split_f32 has a kernel form for this geometry (mirrors launch_igemm's selection)
extern "C" int rick_conv_igemm_split_supported(const rick_conv_geom *g) {
    if (check_geom(g) || (g->Ci & 31) || g->split != 2) return 0;
    ConvTiling t;
    if (igemm_w8_plan(g, &t, true) || igemm_s2w8_plan(g, &t)) return 1;
    if (make_tiling(g, igemm_tile_positions(g), &t)) return 0;
    igemm_plan_split(&t, g->ntaps);
    const size_t lds = igemm_lds_bytes(t, false);
    if (lds > 160 * 1024 || t.PH > 1023 || t.PW > 1023) return 0;
    const bool fullk = t.nsplit == 1 || t.cps >= 4;
    if (igemm_tile_positions(g) == 64) return t.NPP <= 32 * IG_PMAX;
    if (g->ntaps == 9 && t.NPP <= IG_DEEP_NPP && fullk && t.cps >= 4 && lds + CV_WSTEP_BYTES <= 80 * 1024) return 1;
    return t.NPP <= IG_DEEP_NPP;
}

static int igemm_run(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                     const rick_conv_geom *g, const rick_conv_epilogue *epilogue, void *workspace, void *stream, bool pkx) {
    if (!x || !packed_w || !out || check_geom(g)) return RICK_EINVAL;
    rick_conv_epilogue epi = kNoEpilogue;
    if (epilogue) {
        epi = *epilogue;
        if (epi.noise && (!epi.noise_w || (epi.noise_nb != 1 && epi.noise_nb != g->N))) return RICK_EINVAL;
        // the tail is applied on float4 channel groups: Co % 4 == 0, 16-byte aligned bias
        if ((epi.bias || epi.noise || epi.act) && ((g->Co & 3) || ((uintptr_t)(epi.bias ? epi.bias : x) % 16))) return RICK_EINVAL;
        if (epi.split_out && ((g->Co & 3) || !epi.split_hdr || !epi.split_bound || !(epi.split_coef > 0.f) ||
                              (((uintptr_t)epi.split_out | (uintptr_t)(epi.split_scale ? epi.split_scale : x)) % 16)))
            return RICK_EINVAL;
    }
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed_w | (uintptr_t)(iscale ? iscale : x) | (uintptr_t)(oscale ? oscale : x)) % 16)
        return RICK_EINVAL;
    ConvTiling t;
    hipStream_t st = (hipStream_t)stream;
    if (igemm_w8_plan(g, &t, pkx)) {          // eight-wave 128 co x 256 position blocks (igemm_body, NW = 8)
        const int64_t nwg8 = (int64_t)t.ntx * t.nty * t.ntn * t.ncot;
        const size_t lds8 = igemm_w8_lds_bytes(t);
        const unsigned char *wp8 = (const unsigned char *)packed_w;
        if (pkx) launch_igemm_k<2, true, false, 4, 9, 4, true, 8>((unsigned)nwg8, lds8, st, x, wp8, out, iscale, oscale, nullptr, g, t, epi);
        else launch_igemm_k<2, true, false, 4, 9, 4, false, 8>((unsigned)nwg8, lds8, st, x, wp8, out, iscale, oscale, nullptr, g, t, epi);
        RICK_LAUNCH_STATUS();
    }
    if (igemm_s2w8_plan(g, &t)) {             // eight-wave 256 co x 128 position blocks, stride 2 (igemm_body, WDMA = 5)
        const int64_t nwg8 = (int64_t)t.ntx * t.nty * t.ntn * t.ncot;
        const size_t lds8 = igemm_s2w8_lds_bytes(t);
        const unsigned char *wp8 = (const unsigned char *)packed_w;
        if (pkx) launch_igemm_k<2, true, false, 4, 9, 5, true, 8>((unsigned)nwg8, lds8, st, x, wp8, out, iscale, oscale, nullptr, g, t, epi);
        else launch_igemm_k<2, true, false, 4, 9, 5, false, 8>((unsigned)nwg8, lds8, st, x, wp8, out, iscale, oscale, nullptr, g, t, epi);
        RICK_LAUNCH_STATUS();
    }
    if (make_tiling(g, igemm_tile_positions(g), &t)) return RICK_EINVAL;
    igemm_plan_split(&t, g->ntaps);
    if (t.nsplit > 1 && (!workspace || ((uintptr_t)workspace % 16))) return RICK_EINVAL;
    if (t.nsplit > 1 && epi.split_out) return RICK_EINVAL;          // (the split-K second stage writes no image)
    const size_t lds = igemm_lds_bytes(t, iscale != nullptr);
    if (lds > 160 * 1024 || t.PH > 1023 || t.PW > 1023) return RICK_EINVAL;
    const int64_t nwg = (int64_t)t.ntx * t.nty * t.ntn * t.ncot * t.nsplit;
    if (nwg > 0x7fffffff) return RICK_EINVAL;
    float *ws = (float *)workspace;
    const unsigned char *wp = (const unsigned char *)packed_w;
    const bool vec = (g->Ci & 3) == 0;
    if (t.nsplit > 1 && (g->Co & 3) == 0 && vec && (int64_t)t.nsplit * g->N * g->GH * g->GW * g->Co * 4 < (1LL << 32))   // second stage inside the launch?
        t.tickets = rick_internal_tickets(t.ntx * t.nty * t.ntn * t.ncot, stream);     // (NULL unless rick_conv_tuning switched it on)
    int rc;
    if (g->split == 2) {
        if (vec) rc = launch_igemm<2, true>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi, pkx);
        else rc = launch_igemm<2, false>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi, pkx);
    } else {
        if (vec) rc = launch_igemm<1, true>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi, pkx);
        else rc = launch_igemm<1, false>((unsigned)nwg, lds, st, x, wp, out, iscale, oscale, ws, g, t, epi, pkx);
    }
    if (rc) return rc;
    if (t.nsplit > 1 && !t.tickets) launch_splitk_reduce(ws, out, oscale, g, t.nsplit, st, epi);
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
    RICK_LDS160_ONCE((conv_igemm_multi_kernel<SPLIT, VEC>));
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

CV_DEFINE_SAT_ACCESSOR(rick_sat_conv)
