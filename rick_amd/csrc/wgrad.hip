// Weight gradients of the convolution family on gfx950 MFMA (v_mfma_f32_16x16x32_f16); see conv.hip for the forward /
// data-gradient kernels and conv_common.h for the fp16 hi/lo split.
#include "conv_tiling.h"

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
//              narrower stride-1 tiles keep bit 2.  Stride-2 patches store every row as [even columns | odd columns]
//              (conv_tiling.h cv_patch_col), which turns the gather of every other pixel into the stride-1 pattern: bit 3
//              is conflict-free there as well (row-major rows: 2-way on every read, 28 % of the LDS cycles; after: 8 % —
//              and the same kernel time, tools/bench_split.py: the stride-2 form is bound by the issue of its 12-item
//              staging, not by LDS) (tools/lds_sim.py)
#define WG_TILE 64
#ifndef WG_BUF12
#define WG_BUF12 1      // stride-2 patches (PMAX = 12): range-checked buffer loads as well, but one basic block per SLOT
#endif
#define WG_GY_BYTES (WG_TILE * CV_BM * 2)   // 16 KB (one of hi / lo)
#ifndef WG_PIPE_B
#define WG_PIPE_B 0      // experiment (round 4), measured and left off: see PIPE_B in the tile loop
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ f16x8 tr_read2(const unsigned char *base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + off1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 r = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, r);
}

typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {   // buffer_load_dwordx4 ... offen
    const wg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ int wg_key(int r) { return (((r >> 3) & 1) << 2) + (r & 3); }
__device__ __forceinline__ int wg_pswz(int kg, int pix, int kb) { return kg ^ (((pix >> kb) & 1) << 1); }

// FAST (pipelined form; the host selects it for layers whose position grid is an exact multiple of the tile and whose
// channel counts fill the 128 x 32 block — every 3x3 / 1x1 convolution of both networks at >= 8x8): staging is stripped to
// what such a layer needs.  gy items are always valid (one unconditional load, no mask bookkeeping); x items test two
// unsigned compares (halo of the padding) and take a buffer-load offset beyond the descriptor's range when outside (the
// hardware returns zeros); conversion is the fp16 split with the scale folded in — no zero-select.  ~15 / ~22 instructions per item instead of ~30.  The
// kernel is bound by the ISSUE of exactly these instructions (one wave per SIMD, 2-3 of them per MFMA, and an MFMA
// leaves room for ~2), not by the matrix pipe: measured +20...27 % (280 -> 330-345 TFLOP/s on the 64^2...256^2 layers).
// PK (FAST == 1 only): bit 0 = gy is a split image, bit 1 = x is a split image (conv_common.h; `ascale` / `bscale` then point
// at the image's header instead of a scale table).  A split image stores the fp16 hi / lo halves of 4 channels in the 16 bytes
// the fp32 tensor uses for them, so the staging items keep their addresses and their LDS slots; an item is copied to LDS as it
// is (two ds_write_b64, no conversion) and the operand's exponent comes from the header.
template <int NT, int SPLIT, bool VEC, int PMAX, bool PIPE, int FAST = 0, int PK = 0>   // FAST: 1 = no per-channel scales, 2 = with
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                         float *__restrict__ ws, const float *__restrict__ ascale,
                                                         const float *__restrict__ bscale, const rick_conv_geom g,
                                                         const ConvTiling t, int nsplit, int tiles_per_split) {
    static_assert(PK == 0 || (FAST == 1 && SPLIT == 2 && PIPE), "split-image operands: the scale-free pipelined form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cv_fp16_saturate();
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
    // into ONE L2 and re-used by the nchunks blocks that need it.  With 1 / 2 / 4 splits a slice is shared by 8 / 4 / 2
    // XCDs, each with a contiguous (co-tile major) part of its blocks: an XCD then reads 1 / 8 ... 1 / 2 of the slice's gy
    // tiles instead of all of them (512 x 512 @64^2, batch 8: 628 -> ~200 MB from HBM).  (Placement only affects speed.)
    int split, cc;
    const int ncc = t.nchunks * t.ncot;
    if ((nsplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, spx = nsplit >> 3;
        split = xcd + 8 * (j % spx);
        cc = j / spx;
    } else if ((nsplit == 1 || nsplit == 2 || nsplit == 4) && ncc % (8 / nsplit) == 0 && !(t.debug & 16)) {
        const int xps = 8 / nsplit, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        split = xcd / xps;
        cc = (xcd % xps) * (ncc / xps) + j;
    } else {
        split = blockIdx.x / ncc;
        cc = blockIdx.x % ncc;
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
            pbase[kk][h] = (nbi * t.PH + py * g.is) * t.PW + (g.is == 2 ? px : px * g.is);   // (stride 2: de-interleaved columns, conv_tiling.h)
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
        sA[i] = (!ascale || (PK & 1)) ? 1.f : co < g.Co ? ascale[(int64_t)n * g.Co + co] : 0.f;
    }
    for (int i = threadIdx.x; i < g.N * CV_CK; i += 256) {
        const int n = i >> 5, ci = ci_base + (i & 31);
        sB[i] = (!bscale || (PK & 2)) ? 1.f : ci < g.Ci ? bscale[(int64_t)n * g.Ci + ci] : 0.f;
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
        const int slot = pix < t.NPP ? cv_patch_slot(pix, ptab[pix], t, g.is) : t.NPP;
        p_lds[k] = slot * 64 + wg_pswz(pc4 >> 1, slot, t.pkb) * 16 + (pc4 & 1) * 8;
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
    // Saturation tracking (conv_common.h) without a live VGPR — this kernel has none to spare (two extra ones cost 24-200
    // spilled registers): a converted item whose raw maximum exceeds thr = 65504 / (largest scale x exponent of the block's
    // table) sets the wave's bit mask, which lives in SGPRs (a ballot result is wave-uniform).
    [[maybe_unused]] float thr_a = 3.0e38f, thr_b = 3.0e38f;
    unsigned long long sat_bits = 0;
    // Shipping build: OFF in this kernel — its whole-tile forms sit at exactly 512 registers, the three compares per item cost
    // 17-21 spilled registers and 1.5 % of a train step (same-box A/B).  The forward / data-gradient kernel (conv.hip) tracks
    // always: every tensor this kernel converts is also an operand there in the same step (x in fprop, gy in dgrad).  The
    // experiment build (make abl, -DRICK_ABLATION) tracks here too; tools/stability.py runs on it.
#ifdef RICK_ABLATION
#define WG_TRACK(v, thr) sat_bits |= __builtin_amdgcn_ballot_w64(fmaxf(fmaxf(fabsf((v).x), fabsf((v).y)), fmaxf(fabsf((v).z), fabsf((v).w))) > (thr))
#else
#define WG_TRACK(v, thr) (void)0
#endif
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
                if constexpr ((PK & 1) == 0) {
                    const int k = (2 * sidx + 5 * kk + 1) & 7;
                    const unsigned e = g_pyx[k];
                    const int nbi = (int)(e >> 20), py = (int)((e >> 10) & 1023), px = (int)(e & 1023);
                    const bool ok = nt > 0 && e != 0xffffffffu && nbi < nrem && py < yrem && px < xrem;
                    float4 v = load4<VEC>(ok ? gbase + g_rel[k] : (VEC ? g_zero_page : gy), ok, gco, g.Co);
                    v = mul4(v, *reinterpret_cast<const float4 *>(sA + (ok ? (n0 + nbi) * CV_BM : 0) + gc4 * 4));
                    if (ok) ma = amax4(ma, v);
                }
                if constexpr ((PK & 2) == 0) {
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
        if constexpr ((PK & 1) != 0) { sa = 1.f; ua = ascale[1]; }      // header {2^e, 2^-e, ..} of the split image
        if constexpr ((PK & 2) != 0) { sb = 1.f; ub = bscale[1]; }
        punscale = cv_uniform(ua * ub);
        xsa = cv_uniform(sa);
        xsb = cv_uniform(sb);
        __syncthreads();                                    // every thread has read `red`
        float ta = 0.f, tb = 0.f;
        for (int i = threadIdx.x; i < g.N * CV_BM; i += 256) ta = fmaxf(ta, fabsf(sA[i] *= sa));
        for (int i = threadIdx.x; i < g.N * CV_CK; i += 256) tb = fmaxf(tb, fabsf(sB[i] *= sb));
        thr_a = cv_uniform(65504.f / fmaxf(block_amax(ta, red), 1e-30f));
        thr_b = cv_uniform(65504.f / fmaxf(block_amax(tb, red + 8), 1e-30f));
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
            WG_TRACK(v, thr_a);
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
            WG_TRACK(v, thr_b);
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
                        const int toff = (g.dy[tt] - t.dymin) * t.PW + cv_patch_col(g.dx[tt] - t.dxmin, t.PW, g.is);
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
        // FAST: buffer descriptors over the whole tensors (the host only selects FAST below 4 GB per tensor)
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(gy), 0, FAST ? (unsigned)((int64_t)g.N * g.OH * g.OW * g.Co * 4) : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x), 0, FAST ? (unsigned)((int64_t)g.N * g.IH * g.IW * g.Ci * 4) : 0u, 0x00020000);
        int g_toff = 0, x_toff = 0;              // byte offset of the loading tile's origin (+ this thread's channel quad)
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
            if constexpr (FAST != 0) {           // (32-bit: FAST tensors are < 4 GB; the x origin may lie before the tensor — halo)
                g_toff = ((((l_n0 * g.OH + gy0 * g.os + g.oy0) * g.OW + gx0 * g.os + g.ox0) * g.Co) + gco) * 4;
                x_toff = ((((l_n0 * g.IH + l_iy0) * g.IW + l_ix0) * g.Ci) + pci) * 4;
            }
            l_nrem = g.N - l_n0;
            l_yrem = g.GH - gy0;
            l_xrem = g.GW - gx0;
            mask_ld = 0;
        };
        auto issue_item = [&](auto KC) {         // raw load of staging item K of the tile selected by set_tile
            constexpr int K = decltype(KC)::value;
            if constexpr (FAST && K < 8) {
                // buffer loads: one descriptor per tensor, the tile's origin and the item in a 32-bit byte offset
                if constexpr (PMAX <= 4 || WG_BUF12) gq[K] = buf_load4(g_rsrc, (unsigned)(g_toff + g_rel[K] * 4));
                else gq[K] = *reinterpret_cast<const float4 *>(gbase + g_rel[K]);
            } else if constexpr (FAST) {
                constexpr int P = K - 8;
                const unsigned e = p_pyx[P];
                const unsigned iy = (unsigned)(l_iy0 + (int)((e >> 10) & 1023)), ix = (unsigned)(l_ix0 + (int)(e & 1023));
                const bool ok = (e != 0xffffffffu) & (iy < (unsigned)g.IH) & (ix < (unsigned)g.IW);
                // an item outside the image (zero padding) gets an offset beyond the descriptor's range: the hardware's range
                // check returns zeros.  One v_cndmask on a 32-bit offset — the pointer select this replaces was compiled into a
                // branch per item (s_and_saveexec / s_cbranch_execz around the 64-bit address arithmetic), which cut the tile's code
                // into basic blocks and kept the scheduler from spreading the staging work under the MFMAs.
                // (PMAX = 12, the stride-2 patches: one basic block per tile needs more registers than the 512 there are — 54-89
                // spilled in the loop, measured 227 -> 204 TF — so those keep the pointer select and its per-item branch)
                if constexpr (PMAX <= 4 || WG_BUF12) pq[P] = buf_load4(x_rsrc, ok ? (unsigned)(x_toff + p_rel[P] * 4) : 0xFFFFFFF0u);
                else pq[P] = *reinterpret_cast<const float4 *>(ok ? xbase + p_rel[P] : g_zero_page);
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
                if constexpr (K < 8 && (PK & 1) != 0) {            // split image: the item already IS {hi x 4 | lo x 4}
                    const float4 v = gq[K];
                    *reinterpret_cast<float2 *>(buf + g_lds[K]) = make_float2(v.x, v.y);
                    *reinterpret_cast<float2 *>(buf + WG_GY_BYTES + g_lds[K]) = make_float2(v.z, v.w);
                } else if constexpr (K >= 8 && (PK & 2) != 0) {
                    const float4 v = pq[K - 8];
                    *reinterpret_cast<float2 *>(buf + 2 * WG_GY_BYTES + p_lds[K - 8]) = make_float2(v.x, v.y);
                    *reinterpret_cast<float2 *>(buf + 2 * WG_GY_BYTES + (t.NPP + 1) * 64 + p_lds[K - 8]) = make_float2(v.z, v.w);
                } else if constexpr (K < 8) {
                    const float4 v = gq[K];
                    WG_TRACK(v, thr_a);
                    if constexpr (FAST == 2) {     // modulated layers (G) carry per-(image, channel) scales (x block exponent)
                        if constexpr (ONE_IMG) split4v_mix<SPLIT>(v, sa_cv, hi, lo);
                        else split4v_mix<SPLIT>(v, *reinterpret_cast<const float4 *>(sA + (cv_n0 + (int)(g_pyx[K] >> 20)) * CV_BM + gc4 * 4), hi, lo);
                    } else split4s_mix<SPLIT>(v, xsa, hi, lo);   // block exponent from an SGPR
                    *reinterpret_cast<uint2 *>(buf + g_lds[K]) = hi;
                    *reinterpret_cast<uint2 *>(buf + WG_GY_BYTES + g_lds[K]) = lo;
                } else {
                    const float4 v = pq[K - 8];
                    WG_TRACK(v, thr_b);
                    if constexpr (FAST == 2) {     // (an out-of-range item read the zero page: 0 * scale stays 0)
                        if constexpr (ONE_IMG) split4v_mix<SPLIT>(v, sb_cv, hi, lo);
                        else split4v_mix<SPLIT>(v, *reinterpret_cast<const float4 *>(sB + (cv_n0 + (int)((p_pyx[K - 8] >> 20) & 15)) * CV_CK + pc4 * 4), hi, lo);
                    } else split4s_mix<SPLIT>(v, xsb, hi, lo);
                    *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + p_lds[K - 8]) = hi;
                    *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + (t.NPP + 1) * 64 + p_lds[K - 8]) = lo;
                }
            } else if constexpr (K < 8) {
                float4 v = gq[K];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                WG_TRACK(v, thr_a);
                uint2 hi, lo;
                if constexpr (ONE_IMG) split4v<SPLIT>(v, sa_cv, hi, lo);
                else split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sA + (ok ? (cv_n0 + (int)(g_pyx[K] >> 20)) * CV_BM : 0) + gc4 * 4), hi, lo);
                *reinterpret_cast<uint2 *>(buf + g_lds[K]) = hi;
                if (SPLIT == 2) *reinterpret_cast<uint2 *>(buf + WG_GY_BYTES + g_lds[K]) = lo;
            } else {
                constexpr int P = K - 8;
                float4 v = pq[P];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                WG_TRACK(v, thr_b);
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
                // ONEBB (FAST, small patches): the tile is straight-line code — no branch per slot or item (compile-time tap count,
                // buffer loads with range-checked offsets) — so hipcc's scheduler spreads the staging work of a slot (fp16 split,
                // LDS stores, the next global load) and the LDS reads under the MFMAs of its neighbours: 276 -> 300-308 TF at
                // N = 4, 315-329 -> 338-351 TF at N = 8.  Measured and rejected on top of it: B fragments software-pipelined by
                // hand one slot ahead with sched_barrier(0) between slots (-3 %), and a sched_group_barrier template for the
                // whole tile (1 MFMA : 2 VALU with the LDS / VMEM instructions pinned: the scheduler then front-loads ~170
                // SALU / VALU instructions and emits 11 % more code).
                constexpr bool ONEBB = FAST != 0 && PMAX <= 4;
                // PIPE_B (both operands split images, whole-tile form): with no conversion work left to spread, hipcc's
                // scheduler hoists ALL 104 B-fragment reads of the tile to its top, runs out of VGPRs and shuttles the fragments
                // through AGPRs — 208 v_accvgpr_{write,read,mov} per 216 MFMAs in the ISA (0.96 of the 1.21 non-MFMA VALU per
                // MFMA the counters showed).  Here the fragments of slot s + 1 are read explicitly before the MFMAs of slot s and
                // a sched_barrier closes every slot: bounded live ranges, 174 instead of 208 moves — and SLOWER: 380 vs 410 TF on
                // 512 x 512 @64^2 (batch 8): the fences cost more issue slack than the moves do, as in round 3.  Off (WG_PIPE_B).
                constexpr bool PIPE_B = WG_PIPE_B && PK == 3 && ONEBB;
                f16x8 bq_hi, bq_lo;
                auto read_b = [&](auto SN) {
                    constexpr int S2 = decltype(SN)::value, kk2 = S2 / NT, tt2 = S2 % NT;
                    const int toff = (g.dy[tt2] - t.dymin) * t.PW + cv_patch_col(g.dx[tt2] - t.dxmin, t.PW, g.is);
                    const int pp0 = pbase[kk2][0] + toff, pp1 = pbase[kk2][1] + toff;
                    const int o0 = pp0 * 64 + wg_pswz(b_kg, pp0, t.pkb) * 16 + b_sub;
                    const int o1 = pp1 * 64 + wg_pswz(b_kg, pp1, t.pkb) * 16 + b_sub;
                    bq_hi = tr_read2(bph, o0, o1);
                    bq_lo = tr_read2(bpl, o0, o1);
                };
                if constexpr (PIPE_B) read_b(std::integral_constant<int, 0>{});
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
                    if constexpr (PIPE_B) {
                        const f16x8 bhi = bq_hi, blo = bq_lo;
                        if constexpr (S + 1 < NSLOT) read_b(std::integral_constant<int, S + 1>{});
    #pragma unroll
                        for (int i = 0; i < 4; i++) {
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi, acc[i][tt], 0, 0, 0);
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo, acc[i][tt], 0, 0, 0);
                            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi, acc[i][tt], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (ONEBB || tt < g.ntaps) {     // (ONEBB: the host guarantees ntaps == NT)
                        const int ts = (ONEBB || tt < g.ntaps) ? tt : 0;
                        const int toff = (g.dy[ts] - t.dymin) * t.PW + cv_patch_col(g.dx[ts] - t.dxmin, t.PW, g.is);
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

    if (sat_bits != 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_cv_sat, 1u);
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

// ==========================================================================================
// The whole-tile split-image form on EIGHT waves (round 5; verdict round 4 item 4).  Same block (128 co x 9 taps x 32 ci), same
// LDS images, same fragment reads and the same MFMA order per accumulator as conv_wgrad_kernel<.., PK = 3> — bit-identical
// partial tiles — but a wave owns 32 co x 16 ci x 9 taps = 72 accumulator registers and half the staging items, so the kernel
// fits 256 registers (230 / 246, no scratch) and TWO waves share a SIMD; the four-wave form sits at 472-512 registers, one wave
// per SIMD, and hides its LDS / VMEM latencies by hand-placing staging items between MFMA groups.  Here the hardware
// interleaves two waves: two LDS operand buffers, one barrier per tile, the registers of tile t + 1 copied into the other
// buffer and the loads of tile t + 2 issued before the MFMAs of tile t.  Measured against the four-wave form (batch 8, both
// operands split images, tools/bench_split.py, same box): stride 2 254 -> 224 us (128 x 256 @256^2), 247 -> 218 (256 x 512
// @128^2), 138 -> 123 (512 x 512 @64^2) = +12..14 %; stride 1 394 -> 380 (256 x 256 @128^2), 411 -> 393 (128 x 128 @256^2),
// 114 -> 110 (512 x 512 @32^2) = +3..4 %, and 385 -> 393 us (-2 %) on 512 x 512 @64^2.  fp32 operands (the conversion then runs
// under the partner wave's MFMAs): 530 -> 480, 486 -> 420, 501 -> 426, 146 -> 123 us stride 1, 312 -> 271, 308 -> 254, 172 -> 139
// us stride 2 (+10..24 %); 1 x 1: 136 -> 96 / 93 -> 81 us (fp32 / images, 128 x 256 @128^2).  In situ (bench.py, same box,
// experiment build with RICK_WGRAD8 = 0 / 1): 165.5 -> 169.4 images/s.  Measured and not kept: the two waves of a SIMD taking
// the staging and the MFMA half of a tile period in opposite order (w < 4 stage first, w >= 4 multiply first): stride-1 fp32
// 474 -> 564, 407 -> 435 us, the rest unchanged — the hardware's own interleaving of two in-phase waves does better.
// PK = 3: both operands split images (`ascale` / `bscale` = their headers).  PK = 0: fp32 operands split on the fly — FAST = 2 with
// per-(image, channel) scale tables (the generator's modulated layers: ascale = demodulation, bscale = style), FAST = 1 without;
// one image per tile (every layer >= 8 x 8), so a tile's scale vectors live in registers.  The conversion work of one wave
// (12 VALU per staged float4) now runs under the OTHER wave's MFMAs instead of between this wave's own.
template <int NT, int PM8, int FAST, int PK>
__global__ __launch_bounds__(512) void conv_wgrad8_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                          float *__restrict__ ws, const float *__restrict__ ascale,
                                                          const float *__restrict__ bscale, const rick_conv_geom g,
                                                          const ConvTiling t, int nsplit, int tiles_per_split) {
    static_assert(PK == 0 || PK == 3, "both operands fp32, or both split images");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cv_fp16_saturate();
    const int bufsz = 2 * WG_GY_BYTES + 2 * (t.NPP + 1) * 64;
    unsigned *ptab = reinterpret_cast<unsigned *>(smem + 2 * bufsz);
    float *sA = reinterpret_cast<float *>(ptab + ((t.NPP + 3) & ~3));   // [N][128 co] scales of gy (PK = 0)
    float *sB = sA + g.N * CV_BM;                                        // [N][32 ci]  scales of x
    build_patch_table(ptab, t);
    int split, cc;
    const int ncc = t.nchunks * t.ncot;
    if ((nsplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, spx = nsplit >> 3;
        split = xcd + 8 * (j % spx);
        cc = j / spx;
    } else if ((nsplit == 1 || nsplit == 2 || nsplit == 4) && ncc % (8 / nsplit) == 0) {
        const int xps = 8 / nsplit, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        split = xcd / xps;
        cc = (xcd % xps) * (ncc / xps) + j;
    } else {
        split = blockIdx.x / ncc;
        cc = blockIdx.x % ncc;
    }
    const int chunk = cc % t.nchunks, cot = cc / t.nchunks;
    const int ntiles = t.ntx * t.nty * t.ntn;
    const int tile_begin = split * tiles_per_split;
    const int tile_end = tile_begin + tiles_per_split < ntiles ? tile_begin + tiles_per_split : ntiles;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;                     // wm: 32-co quarter, wn: 16-ci half
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
            nbi = nbi < t.nbe ? nbi : t.nbe - 1;
            pbase[kk][h] = (nbi * t.PH + py * g.is) * t.PW + (g.is == 2 ? px : px * g.is);
        }
    const int b_kg = wn * 2 + (p >> 1), b_sub = (p & 1) * 8;
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int tt = 0; tt < NT; tt++) acc[i][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging maps for 512 threads: 4 gy items (rows (tid >> 5) + 16 k), PM8 patch items (pixels (tid >> 3) + 64 k)
    float4 gq[4], pq[PM8];
    int g_rel[4], g_lds[4], p_rel[PM8], p_lds[PM8];
    unsigned p_pyx[PM8];
    const int gc4 = threadIdx.x & 31, pc4 = threadIdx.x & 7;
    const int gco = cot * CV_BM + gc4 * 4, pci = chunk * CV_CK + pc4 * 4;
    if constexpr (PK == 0) {
        for (int i = threadIdx.x; i < g.N * CV_BM; i += 512) {
            const int n = i >> 7, co = cot * CV_BM + (i & 127);
            sA[i] = !ascale ? 1.f : co < g.Co ? ascale[(int64_t)n * g.Co + co] : 0.f;
        }
        for (int i = threadIdx.x; i < g.N * CV_CK; i += 512) {
            const int n = i >> 5, ci = chunk * CV_CK + (i & 31);
            sB[i] = !bscale ? 1.f : ci < g.Ci ? bscale[(int64_t)n * g.Ci + ci] : 0.f;
        }
    }
    __syncthreads();                                             // patch table (and scale tables) complete
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = (threadIdx.x >> 5) + 16 * k;
        const int px = r & tw_mask, py = (r >> t.tw_log2) & th_mask, nbi = r >> (t.tw_log2 + t.th_log2);
        g_rel[k] = ((nbi * g.OH + py * g.os) * g.OW + px * g.os) * g.Co;
        g_lds[k] = r * 256 + ((gc4 * 8) ^ (wg_key(r) * 32));
    }
#pragma unroll
    for (int k = 0; k < PM8; k++) {
        const int pix = (threadIdx.x >> 3) + 64 * k;
        p_rel[k] = 0;
        p_pyx[k] = 0xffffffffu;
        const int slot = pix < t.NPP ? cv_patch_slot(pix, ptab[pix], t, g.is) : t.NPP;
        p_lds[k] = slot * 64 + wg_pswz(pc4 >> 1, slot, t.pkb) * 16 + (pc4 & 1) * 8;
        if (pix < t.NPP) {
            const unsigned e = ptab[pix];
            p_rel[k] = (((int)(e >> 20) * g.IH + (int)((e >> 10) & 1023)) * g.IW + (int)(e & 1023)) * g.Ci;
            p_pyx[k] = e;
        }
    }
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(gy), 0, (unsigned)((int64_t)g.N * g.OH * g.OW * g.Co * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (unsigned)((int64_t)g.N * g.IH * g.IW * g.Ci * 4), 0x00020000);
    int g_toff = 0, x_toff = 0, l_iy0 = 0, l_ix0 = 0, l_n0 = 0;
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
        g_toff = ((((l_n0 * g.OH + gy0 * g.os + g.oy0) * g.OW + gx0 * g.os + g.ox0) * g.Co) + gco) * 4;
        x_toff = ((((l_n0 * g.IH + l_iy0) * g.IW + l_ix0) * g.Ci) + pci) * 4;
    };
    auto issue = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) gq[k] = buf_load4(g_rsrc, (unsigned)(g_toff + g_rel[k] * 4));
#pragma unroll
        for (int k = 0; k < PM8; k++) {
            const unsigned e = p_pyx[k];
            const unsigned iy = (unsigned)(l_iy0 + (int)((e >> 10) & 1023)), ix = (unsigned)(l_ix0 + (int)(e & 1023));
            const bool ok = (e != 0xffffffffu) & (iy < (unsigned)g.IH) & (ix < (unsigned)g.IW);
            pq[k] = buf_load4(x_rsrc, ok ? (unsigned)(x_toff + p_rel[k] * 4) : 0xFFFFFFF0u);     // (beyond the range: zeros)
        }
    };
    // ---- operand exponents (PK = 0; conv_common.h): amax of gy * ascale and x * bscale over 4 of the block's tiles x 2 staging
    // items each, then 2^e is folded into the scale tables; split images bring theirs in the header
    float punscale, xsa = 1.f, xsb = 1.f;
    [[maybe_unused]] float thr_a8 = 3.0e38f, thr_b8 = 3.0e38f;      // (RICK_ABLATION: saturation thresholds, see the exponent block)
    [[maybe_unused]] unsigned long long sat_bits8 = 0;
#ifdef RICK_ABLATION
#define WG_TRACK8(v, thr) sat_bits8 |= __builtin_amdgcn_ballot_w64(fmaxf(fmaxf(fabsf((v).x), fabsf((v).y)), fmaxf(fabsf((v).z), fabsf((v).w))) > (thr))
#else
#define WG_TRACK8(v, thr) (void)0
#endif
    if constexpr (PK == 3) {
        punscale = cv_uniform(ascale[1] * bscale[1]);
    } else {
        float ma = 0.f, mb = 0.f;
        const int nt = tile_end - tile_begin;
        if (nt > 0) {
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) {
                set_tile(tile_begin + (sidx * nt) / 4);
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int k = (sidx + 3 * kk + 1) & 3;
                    float4 v = buf_load4(g_rsrc, (unsigned)(g_toff + g_rel[k] * 4));
                    v = mul4(v, *reinterpret_cast<const float4 *>(sA + l_n0 * CV_BM + gc4 * 4));
                    ma = amax4(ma, v);
                    const int kp = (3 * sidx + 7 * kk + 2) % PM8;
                    const unsigned e = p_pyx[kp];
                    const unsigned iy = (unsigned)(l_iy0 + (int)((e >> 10) & 1023)), ix = (unsigned)(l_ix0 + (int)(e & 1023));
                    const bool ok = (e != 0xffffffffu) & (iy < (unsigned)g.IH) & (ix < (unsigned)g.IW);
                    float4 w = buf_load4(x_rsrc, ok ? (unsigned)(x_toff + p_rel[kp] * 4) : 0xFFFFFFF0u);
                    w = mul4(w, *reinterpret_cast<const float4 *>(sB + l_n0 * CV_CK + pc4 * 4));
                    mb = amax4(mb, w);
                }
            }
        }
        float *red = reinterpret_cast<float *>(smem);            // (operand buffers are not in use yet)
        ma = block_amax(ma, red);
        mb = block_amax(mb, red + 8);
        float sa, ua, sb, ub;
        cv_pow2_scale(ma, sa, ua);
        cv_pow2_scale(mb, sb, ub);
        punscale = cv_uniform(ua * ub);
        xsa = cv_uniform(sa);
        xsb = cv_uniform(sb);
        __syncthreads();                                         // every thread has read `red`
#ifdef RICK_ABLATION
        // saturation tracking like the four-wave kernel's (WG_TRACK above; experiment build only — tools/stability.py runs on it):
        // thr = 65504 / (largest scale x exponent of the block's table); a converted item above it sets the wave's SGPR mask
        float ta = 0.f, tb = 0.f;
        for (int i = threadIdx.x; i < g.N * CV_BM; i += 512) ta = fmaxf(ta, fabsf(sA[i] *= sa));
        for (int i = threadIdx.x; i < g.N * CV_CK; i += 512) tb = fmaxf(tb, fabsf(sB[i] *= sb));
        thr_a8 = cv_uniform(65504.f / fmaxf(block_amax(ta, red), 1e-30f));
        thr_b8 = cv_uniform(65504.f / fmaxf(block_amax(tb, red + 8), 1e-30f));
#else
        for (int i = threadIdx.x; i < g.N * CV_BM; i += 512) sA[i] *= sa;
        for (int i = threadIdx.x; i < g.N * CV_CK; i += 512) sB[i] *= sb;
#endif
        __syncthreads();
    }
    float4 sa_cv = make_float4(1.f, 1.f, 1.f, 1.f), sb_cv = sa_cv;   // scale vectors of the tile whose registers are held
    auto tile_scales = [&]() {
        if constexpr (PK == 0 && FAST == 2) {
            const int n = l_n0 < g.N ? l_n0 : 0;
            sa_cv = *reinterpret_cast<const float4 *>(sA + n * CV_BM + gc4 * 4);
            sb_cv = *reinterpret_cast<const float4 *>(sB + n * CV_CK + pc4 * 4);
        }
    };
    auto store = [&](unsigned char *buf) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (PK == 3) {                             // a split-image item IS {hi x 4 | lo x 4}
                *reinterpret_cast<float2 *>(buf + g_lds[k]) = make_float2(gq[k].x, gq[k].y);
                *reinterpret_cast<float2 *>(buf + WG_GY_BYTES + g_lds[k]) = make_float2(gq[k].z, gq[k].w);
            } else {
                uint2 hi, lo;
                WG_TRACK8(gq[k], thr_a8);
                if constexpr (FAST == 2) split4v_mix<2>(gq[k], sa_cv, hi, lo);
                else split4s_mix<2>(gq[k], xsa, hi, lo);
                *reinterpret_cast<uint2 *>(buf + g_lds[k]) = hi;
                *reinterpret_cast<uint2 *>(buf + WG_GY_BYTES + g_lds[k]) = lo;
            }
        }
#pragma unroll
        for (int k = 0; k < PM8; k++) {
            if constexpr (PK == 3) {
                *reinterpret_cast<float2 *>(buf + 2 * WG_GY_BYTES + p_lds[k]) = make_float2(pq[k].x, pq[k].y);
                *reinterpret_cast<float2 *>(buf + 2 * WG_GY_BYTES + (t.NPP + 1) * 64 + p_lds[k]) = make_float2(pq[k].z, pq[k].w);
            } else {
                uint2 hi, lo;
                WG_TRACK8(pq[k], thr_b8);
                if constexpr (FAST == 2) split4v_mix<2>(pq[k], sb_cv, hi, lo);
                else split4s_mix<2>(pq[k], xsb, hi, lo);
                *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + p_lds[k]) = hi;
                *reinterpret_cast<uint2 *>(buf + 2 * WG_GY_BYTES + (t.NPP + 1) * 64 + p_lds[k]) = lo;
            }
        }
    };
    if (tile_begin < tile_end) {
        set_tile(tile_begin);
        issue();
        tile_scales();
        store(smem);
        set_tile(tile_begin + 1 < tile_end ? tile_begin + 1 : tile_end - 1);
        issue();
    }
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; tile++) {
        const int cur = (tile - tile_begin) & 1;
        const unsigned char *bgh = smem + cur * bufsz, *bgl = bgh + WG_GY_BYTES;
        const unsigned char *bph = bgh + 2 * WG_GY_BYTES, *bpl = bph + (t.NPP + 1) * 64;
        tile_scales();                                           // (l_n0 still belongs to tile + 1, whose registers are held)
        store(smem + (cur ^ 1) * bufsz);
        set_tile(tile + 2 < tile_end ? tile + 2 : tile_end - 1);
        issue();                                                 // tile + 2
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            f16x8 ahi[2], alo[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int cb = (wm * 32 + i * 16 + p * 4) * 2;
                const int o0 = a_row[kk][0] + (cb ^ a_key[kk][0]);
                const int o1 = a_row[kk][1] + (cb ^ a_key[kk][1]);
                ahi[i] = tr_read2(bgh, o0, o1);
                alo[i] = tr_read2(bgl, o0, o1);
            }
#pragma unroll
            for (int tt = 0; tt < NT; tt++) {
                const int toff = (g.dy[tt] - t.dymin) * t.PW + cv_patch_col(g.dx[tt] - t.dxmin, t.PW, g.is);
                const int pp0 = pbase[kk][0] + toff, pp1 = pbase[kk][1] + toff;
                const int o0 = pp0 * 64 + wg_pswz(b_kg, pp0, t.pkb) * 16 + b_sub;
                const int o1 = pp1 * 64 + wg_pswz(b_kg, pp1, t.pkb) * 16 + b_sub;
                const f16x8 bhi = tr_read2(bph, o0, o1);
                const f16x8 blo = tr_read2(bpl, o0, o1);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi, acc[i][tt], 0, 0, 0);
                    acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo, acc[i][tt], 0, 0, 0);
                    acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi, acc[i][tt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float *wsb = ws + (((int64_t)split * t.ncot + cot) * t.nchunks + chunk) * g.ntaps * (CV_BM * CV_CK);
#pragma unroll
    for (int tt = 0; tt < NT; tt++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int co = wm * 32 + i * 16 + G * 4;
            const int ci = wn * 16 + (lane & 15);
            *reinterpret_cast<float4 *>(wsb + (tt * CV_CK + ci) * CV_BM + co) =
                make_float4(acc[i][tt][0] * punscale, acc[i][tt][1] * punscale, acc[i][tt][2] * punscale, acc[i][tt][3] * punscale);
        }
#ifdef RICK_ABLATION
    if (sat_bits8 != 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_cv_sat, 1u);
#endif
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

// Second stage for the two dense destination layouts, written in runs of consecutive floats.  The partial tiles are
// [tap][ci 32][co 128] (co fastest: what the MFMA accumulators hold); the gradient tensor is [co][ci][slice] (TR = false:
// s_ci = nslices, s_t = 1) or [ci][co][slice] (TR = true: s_co = nslices, s_t = 1), so the element-per-thread kernel above
// writes 4 bytes every Ci*9*4 B (or 36 B).  Here a block owns 32 co x 8 ci x taps of one (co tile, chunk) tile — 16 blocks
// per tile, enough of them in flight per CU to keep the partial-sum reads streaming —, sums the splits with float4 loads
// along co, transposes through LDS and writes runs of 8 * nslices (TR: 32 * nslices) consecutive floats.  Same
// summation order as the kernel above: bit-identical results.
template <bool TR>
__global__ __launch_bounds__(256) void wgrad_reduce_tile_kernel(const float *__restrict__ ws, float *__restrict__ gw,
                                                                int64_t s_outer, int Co, int Ci, int nchunks, int ntaps,
                                                                int nsl, int nsplit, float alpha, int accumulate,
                                                                rick_conv_geom g, int64_t per_split) {
    extern __shared__ float wr_tile[];
    const int pc = blockIdx.x & 3, pk = (blockIdx.x >> 2) & 3;   // 32-co quarter, 8-ci quarter of the tile
    const int chunk = (blockIdx.x >> 4) % nchunks, cot = (blockIdx.x >> 4) / nchunks;
    const float4 *src = reinterpret_cast<const float4 *>(ws + ((int64_t)(cot * nchunks + chunk) * ntaps) * CV_CK * CV_BM);
    const int64_t split4 = per_split >> 2;
    const int run = (TR ? 32 : 8) * nsl;                         // consecutive destination floats per LDS row
    const int pitch = run + 1;
    const int c4 = threadIdx.x & 7;                              // 8 float4 = the block's 32 co
    for (int row = threadIdx.x >> 3; row < ntaps * 8; row += 32) {
        const int tt = row >> 3, kl = row & 7;
        const float4 *p = src + ((int64_t)(tt * CV_CK + pk * 8 + kl) * CV_BM + pc * 32) / 4 + c4;
        const float4 a = cv_sum_splits4(p, split4, nsplit);
        const int sl = g.wt[tt];
        if (!TR) {
            float *d = wr_tile + (c4 * 4) * pitch + kl * nsl + sl;
            d[0] = a.x; d[pitch] = a.y; d[2 * pitch] = a.z; d[3 * pitch] = a.w;
        } else {
            float *d = wr_tile + kl * pitch + (c4 * 4) * nsl + sl;
            d[0] = a.x; d[nsl] = a.y; d[2 * nsl] = a.z; d[3 * nsl] = a.w;
        }
    }
    __syncthreads();
    unsigned smask = 0;                                          // slices this launch produces (a tap subset leaves the others alone)
    for (int tt = 0; tt < ntaps; tt++) smask |= 1u << g.wt[tt];
    const int nrows = TR ? 8 : 32;
    for (int e = threadIdx.x; e < nrows * run; e += 256) {
        const int rl = e / run, c = e - rl * run;
        const int inner = c / nsl, sl = c - inner * nsl;
        // TR: row = ci, inner = co;  else: row = co, inner = ci
        const int co = cot * CV_BM + pc * 32 + (TR ? inner : rl);
        const int ci = chunk * CV_CK + pk * 8 + (TR ? rl : inner);
        if (co >= Co || ci >= Ci || !((smask >> sl) & 1u)) continue;
        float *dst = gw + (int64_t)(TR ? ci : co) * s_outer + (int64_t)(TR ? co : ci) * nsl + sl;
        const float v = wr_tile[rl * pitch + c] * alpha;
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

template <int NT, int SPLIT, bool VEC, int PMAX, bool PIPE, int FAST = 0, int PK = 0>
static void launch_wgrad_k(const float *x, const float *gy, float *ws, const float *ascale, const float *bscale,
                           const rick_conv_geom *g, const ConvTiling &t, int nsplit, int tps, size_t lds, hipStream_t st) {
    const unsigned nwg = (unsigned)(nsplit * t.ncot * t.nchunks);
    RICK_LDS160_ONCE((conv_wgrad_kernel<NT, SPLIT, VEC, PMAX, PIPE, FAST, PK>));
    hipLaunchKernelGGL((conv_wgrad_kernel<NT, SPLIT, VEC, PMAX, PIPE, FAST, PK>), dim3(nwg), dim3(256), lds, st, x, gy, ws, ascale,
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

// FAST: position grid an exact multiple of the tile, full 128 x 32 channel blocks
static bool wgrad_fast(const rick_conv_geom *g, const ConvTiling &t, int NT) {
    return wgrad_use_pipe(g, t) && g->ntaps == NT &&
           (int64_t)g->N * g->IH * g->IW * g->Ci < (1LL << 29) && (int64_t)g->N * g->OH * g->OW * g->Co < (1LL << 29) &&   // (32-bit byte offsets of the buffer loads)
           !(g->Co % CV_BM) && !(g->Ci % CV_CK) && !(g->GH & ((1 << t.th_log2) - 1)) &&
           !(g->GW & ((1 << t.tw_log2) - 1)) && !(g->N % t.nbe) && (t.nb == t.nbe);
}

static size_t wgrad8_lds_bytes(const rick_conv_geom *g, const ConvTiling &t) {
    return 2 * (2 * WG_GY_BYTES + 2 * (size_t)(t.NPP + 1) * 64) + (size_t)((t.NPP + 3) & ~3) * 4 + (size_t)g->N * (CV_BM + CV_CK) * 4;
}
template <int NT, int PM8, int FAST, int PK>
static void launch_wgrad8(const float *x, const float *gy, float *ws, const float *ascale, const float *bscale,
                          const rick_conv_geom *g, const ConvTiling &t, int nsplit, int tps, hipStream_t st) {
    if constexpr (NT == 9 || NT == 1) {
        const unsigned nwg = (unsigned)(nsplit * t.ncot * t.nchunks);
        RICK_LDS160_ONCE((conv_wgrad8_kernel<NT, PM8, FAST, PK>));
        hipLaunchKernelGGL((conv_wgrad8_kernel<NT, PM8, FAST, PK>), dim3(nwg), dim3(512), wgrad8_lds_bytes(g, t), st, x, gy, ws, ascale,
                           bscale, *g, t, nsplit, tps);
    }
}

template <int NT>
static int launch_wgrad(const float *x, const float *gy, float *ws, const float *ascale, const float *bscale,
                        const rick_conv_geom *g, const ConvTiling &t, int nsplit, int tps, hipStream_t st, int pk = 0) {
    const bool vec = ((g->Ci | g->Co) & 3) == 0;
    const bool small = t.NPP <= 4 * 32;
    const bool pipe = wgrad_use_pipe(g, t);
    const size_t lds = wgrad_lds_bytes(g, t, pipe);
    const bool fast = wgrad_fast(g, t, NT);
    static const int wg8 = ablation_env("RICK_WGRAD8", 1);              // (A/B switch of the experiment build)
    if (pk) {   // split-image operands (`ascale` / `bscale` carry the headers): only the whole-tile form
        if (!fast || g->split != 2) return RICK_EINVAL;
        if (pk == 3 && (NT == 9 || NT == 1) && wg8 && t.nbe == 1 && wgrad8_lds_bytes(g, t) <= 160 * 1024) {
            // both operands split images: the eight-wave form
            if (small) launch_wgrad8<NT, 2, 1, 3>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
            else launch_wgrad8<NT, 6, 1, 3>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
            return 0;
        }
        if (small) {
            if (pk == 3) launch_wgrad_k<NT, 2, true, 4, true, 1, 3>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
            else if (pk == 2) launch_wgrad_k<NT, 2, true, 4, true, 1, 2>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
            else launch_wgrad_k<NT, 2, true, 4, true, 1, 1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
        } else {
            if (pk == 3) launch_wgrad_k<NT, 2, true, 12, true, 1, 3>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
            else if (pk == 2) launch_wgrad_k<NT, 2, true, 12, true, 1, 2>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
            else launch_wgrad_k<NT, 2, true, 12, true, 1, 1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, lds, st);
        }
        return 0;
    }
    const bool scaled = ascale != nullptr || bscale != nullptr;
    if (fast && vec && g->split == 2 && (NT == 9 || NT == 1) && wg8 && t.nbe == 1 && wgrad8_lds_bytes(g, t) <= 160 * 1024) {
        // fp32 operands, whole tiles, one image per tile: the eight-wave form (conversion under the other wave's MFMAs)
        if (small && scaled) launch_wgrad8<NT, 2, 2, 0>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
        else if (small) launch_wgrad8<NT, 2, 1, 0>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
        else if (scaled) launch_wgrad8<NT, 6, 2, 0>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
        else launch_wgrad8<NT, 6, 1, 0>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st);
        return 0;
    }
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
    return 0;
}

static int wgrad_run(const float *x, const float *gy, float *gw, int64_t s_co, int64_t s_ci, int64_t s_t,
                     const float *ascale, const float *bscale, const rick_conv_geom *g, int accumulate,
                     void *workspace, void *stream, int pk);

extern "C" int rick_conv_wgrad_f32(const float *x, const float *gy, float *gw, int64_t s_co, int64_t s_ci, int64_t s_t,
                                   const float *ascale, const float *bscale, const rick_conv_geom *g, int accumulate,
                                   void *workspace, void *stream) {
    return wgrad_run(x, gy, gw, s_co, s_ci, s_t, ascale, bscale, g, accumulate, workspace, stream, 0);
}

// 1 when the geometry runs the whole-tile form that accepts split-image operands (rick_conv_wgrad_split_f32)
extern "C" int rick_conv_wgrad_split_supported(const rick_conv_geom *g) {
    if (check_geom(g) || g->ntaps > 9 || g->split != 2) return 0;
    ConvTiling t;
    int nsplit, tps;
    wgrad_plan(g, &t, &nsplit, &tps);
    if (t.NPP > 12 * 32) return 0;
    return wgrad_fast(g, t, g->ntaps == 1 ? 1 : g->ntaps <= 4 ? 4 : 9) ? 1 : 0;
}

extern "C" int rick_conv_wgrad_split_f32(const void *x, const float *x_hdr, const void *gy, const float *gy_hdr, float *gw,
                                         int64_t s_co, int64_t s_ci, int64_t s_t, const rick_conv_geom *g, int accumulate,
                                         void *workspace, void *stream) {
    const int pk = (gy_hdr ? 1 : 0) | (x_hdr ? 2 : 0);
    if (!pk) return RICK_EINVAL;
    return wgrad_run((const float *)x, (const float *)gy, gw, s_co, s_ci, s_t, gy_hdr, x_hdr, g, accumulate, workspace, stream, pk);
}

static int wgrad_run(const float *x, const float *gy, float *gw, int64_t s_co, int64_t s_ci, int64_t s_t,
                     const float *ascale, const float *bscale, const rick_conv_geom *g, int accumulate,
                     void *workspace, void *stream, int pk) {
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
    int rc;
    if (g->ntaps == 1) rc = launch_wgrad<1>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st, pk);
    else if (g->ntaps <= 4) rc = launch_wgrad<4>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st, pk);
    else rc = launch_wgrad<9>(x, gy, ws, ascale, bscale, g, t, nsplit, tps, st, pk);
    if (rc) return rc;
    const int64_t per_split = (int64_t)t.ncot * t.nchunks * g->ntaps * CV_BM * CV_CK;
    const int nsl = g->nslices;
    const unsigned ntile = (unsigned)(t.ncot * t.nchunks * 16);
    // (the tile kernel needs >= 512 blocks to keep the partial-sum reads streaming: 4-5 us faster on the 512 x 512 x 9
    // gradients, slower than the element-per-thread kernel on small tensors with many splits)
    static const int tile_min = ablation_env("RICK_WR_TILE", 32);
    const bool tile_reduce = tile_min > 0 && t.ncot * t.nchunks >= tile_min;
    if (tile_reduce && s_t == 1 && s_ci == nsl && s_co >= (int64_t)g->Ci * nsl && nsl <= 16) {          // [co][ci][slice]
        hipLaunchKernelGGL(wgrad_reduce_tile_kernel<false>, dim3(ntile), dim3(256), 32 * (8 * nsl + 1) * 4, st, ws, gw,
                           s_co, g->Co, g->Ci, t.nchunks, g->ntaps, nsl, nsplit, g->alpha, accumulate, *g, per_split);
        RICK_LAUNCH_STATUS();
    }
    if (tile_reduce && s_t == 1 && s_co == nsl && s_ci >= (int64_t)g->Co * nsl && nsl <= 16) {          // [ci][co][slice]
        hipLaunchKernelGGL(wgrad_reduce_tile_kernel<true>, dim3(ntile), dim3(256), 8 * (32 * nsl + 1) * 4, st, ws, gw,
                           s_ci, g->Co, g->Ci, t.nchunks, g->ntaps, nsl, nsplit, g->alpha, accumulate, *g, per_split);
        RICK_LAUNCH_STATUS();
    }
    int64_t nb = cdiv64(per_split, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ws, gw, s_co, s_ci, s_t, g->Co, g->Ci,
                       t.ncot, t.nchunks, g->ntaps, nsplit, g->alpha, accumulate, *g);
    RICK_LAUNCH_STATUS();
}

CV_DEFINE_SAT_ACCESSOR(rick_sat_wgrad)
