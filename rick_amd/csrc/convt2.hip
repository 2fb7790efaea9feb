// Transposed 3x3 stride-2 convolution (padding 0) on gfx950 MFMA — every output-parity class from ONE staged patch.
//
//   out[n, 2*iy + ky, 2*ix + kx, co] += W[co][ci][ky][kx] * (iscale[n,ci] * x[n, iy, ix, ci])      (* alpha * oscale[n,co])
//
// This is F.conv_transpose2d(stride=2) of the upsampling StyledConvs (model_probe_tune.py:257-268) and the data
// gradient of the discriminator's stride-2 3x3 convolutions (:608-630).  An output pixel of parity (py, px) only
// receives the taps with ky = py (mod 2), kx = px (mod 2): 4 / 2 / 2 / 1 of the 9 taps.  The generic kernel
// (conv.hip) runs the four classes as four block ranges and each of them stages and converts (fp32 -> fp16 hi/lo)
// the same input patch again; here a 512-thread block owns one tile of the input-aligned position grid
// (gy, gx) in [0, IH] x [0, IW], stages the (TH+1) x (TW+1) patch ONCE per 32-channel chunk, and its 8 waves are
// (parity class) x (64-channel half of the 128-channel co tile), each with a 64 co x 128 positions accumulator:
//
//   wave 0,1: class (0,0), 4 taps   |  wave 4,5: class (1,1), 1 tap      <- waves w and w+4 share a SIMD:
//   wave 2,3: class (0,1), 2 taps   |  wave 6,7: class (1,0), 2 taps         5 / 5 / 4 / 4 taps per SIMD
//
// so all 9 taps of a chunk are multiplied against one staged patch (the same staging : MFMA ratio as a stride-1
// 3x3 convolution).  Weights are private to a wave (its class's taps, its 64 rows), so they never touch LDS: a wave
// loads its A fragments straight from the packed image in global memory — the packed layout makes every such
// load one contiguous, fully used 1 KB — one tap ahead, into the registers the previous tap has just released.
// The only LDS traffic is the B operand (patch), and the only barrier is one per channel chunk (double-buffered
// patch).  Precision: fp16x3 (hi*hi + hi*lo + lo*hi with a per-block power-of-two operand exponent, fp32 accumulate)
// or plain fp16, as in conv.hip.
#include "conv_common.h"
#include <stdio.h>
#include <stdlib.h>

#define CT_THREADS 512
#define CT_PITEMS 3                       // patch float4 per thread: up to 192 patch pixels
#define CT_MAX_NPP (CT_PITEMS * (CT_THREADS / 8))
// time of a block that computes 4 / 2 of its tile's 8 fragment columns, % of a whole-tile block (measured,
// tools/abl_ct2_subcal.sh: 67-70 % / 64 % — staging and the weight stream do not shrink with the columns)
#define CT_SUB2_PCT 68
#define CT_SUB4_PCT 64

// zeros that out-of-range patch items load (no zero-select afterwards; see conv.hip)
__device__ __attribute__((aligned(64))) float g_ct2_zero_page[16];

struct Ct2Plan {
    int N, IH, IW, Ci, OH, OW, Co;
    int TW, TH, NB;                 // position tile: TW x TH positions of NB images (TW*TH*NB <= 128)
    int ntx, nty, ntn;
    int PH, PW, NPP;                // patch = (TH+1) x (TW+1) input pixels per image, NB images
    int nchunks, ncot, nsplit, cps;
    float alpha;
    // work items [0, nfull) run as whole-tile blocks; every later one is shared by subq blocks, each with 8 / subq of the
    // tile's 8 fragment columns, in a second launch (ct2_plan: the last, partly filled round of a grid then takes a
    // fraction of a round's time).  item0: first work item of the launch.
    int nfull, subq, item0;
    // lane slot (j * 16 + l15) -> tile position (bit 7: slot unused), 4 slots per word; a permutation that makes every
    // ds_read_b128 of the patch conflict-free (ct2_position_map)
    unsigned posw[32];
    float *amax;                    // NULL, or an amax word: max |out| is folded into it (conv_common.h)
};

// PKX: x is a split image (conv_common.h), `iscale` its header; staging copies 16-byte granules (see conv.hip).
template <int SPLIT, int NJ, bool PKX = false>
__global__ __launch_bounds__(CT_THREADS) void convt2_kernel(const float *__restrict__ x,
                                                           const unsigned char *__restrict__ wpk,
                                                           float *__restrict__ out, const float *__restrict__ iscale,
                                                           const float *__restrict__ oscale, float *__restrict__ ws,
                                                           const Ct2Plan P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cv_fp16_saturate();
    const int pbuf = P.NPP * 64;                              // one of hi / lo of one patch buffer
    float *sct = reinterpret_cast<float *>(smem + 4 * pbuf);  // [NB][cps * 32] input scales
    unsigned char *spos = reinterpret_cast<unsigned char *>(sct + P.NB * P.cps * CV_CK);   // [128] lane slot -> tile position
    if (threadIdx.x < 32) {     // (a select chain over the kernel argument: a dynamic index would send the struct to scratch)
        unsigned w = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) w = (int)threadIdx.x == i ? P.posw[i] : w;
        reinterpret_cast<unsigned *>(spos)[threadIdx.x] = w;
    }

    // work item, and the fragment columns [jbase, jbase + NJ) of its tile this block computes (NJ < 8: the launch of the
    // grid's last, partly filled round — 8 / NJ consecutive blocks, on one XCD, share a work item)
    const int r = xcd_remap(blockIdx.x, gridDim.x);
    const int lid = P.item0 + r / (8 / NJ);
    const int jbase = (r % (8 / NJ)) * NJ;
    const int npos_tiles = P.ntx * P.nty * P.ntn;
    int pt = lid % npos_tiles;
    const int split = (lid / npos_tiles) % P.nsplit;
    const int cot = lid / (npos_tiles * P.nsplit);
    const int c_begin = split * P.cps;
    const int c_end = c_begin + P.cps < P.nchunks ? c_begin + P.cps : P.nchunks;
    const int tx_i = pt % P.ntx;
    pt /= P.ntx;
    const int ty_i = pt % P.nty;
    const int tn_i = pt / P.nty;
    const int gx0 = tx_i * P.TW, gy0 = ty_i * P.TH, n0 = tn_i * P.NB;

    const int cspan = P.cps * CV_CK;

    // ---- roles
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, kg = lane >> 4;
    const int h = wave & 1;                                   // 64-row half of the co tile
    const int wp = wave >> 1;
    const int cls = wp ^ (wp >> 1);                           // py*2 + px of wave pairs {0,1},{2,3},{4,5},{6,7}: (0,0) (0,1) (1,1) (1,0)
    const int py = cls >> 1, px = cls & 1;
    const int nky = py ? 1 : 2, nkx = px ? 1 : 2;
    const int ntap = nky * nkx;

    // ---- patch staging state (tile is fixed for the block)
    const int c4 = threadIdx.x & 7;
    const int iy0 = gy0 - 1, ix0 = gx0 - 1;                   // input pixel of patch (0, 0)
    const float *xt = x + (((int64_t)n0 * P.IH + iy0) * P.IW + ix0) * P.Ci + c4 * 4;
    // item k of a thread is patch pixel (tid >> 3) + 64 k: same swizzle key (bit 2 of the pixel), LDS offset + k * 4096
    int p_rel[CT_PITEMS];
    unsigned p_ok = 0, p_nbi = 0;                             // validity bits; image-in-tile of each item, 8 bits each
    const int pix0 = threadIdx.x >> 3;
    const int p_lds0 = pix0 * 64 + cv_swz(c4 >> 1, pix0) * 16 + (c4 & 1) * 8;
    const int phw = P.PH * P.PW;
#pragma unroll
    for (int k = 0; k < CT_PITEMS; k++) {
        const int pix = pix0 + (CT_THREADS / 8) * k;
        p_rel[k] = 0;
        if (pix < P.NPP) {
            const int nbi = pix / phw, rem = pix - nbi * phw;
            const int ry = rem / P.PW, rx = rem - ry * P.PW;
            const int n = n0 + nbi, iy = iy0 + ry, ix = ix0 + rx;
            if (n < P.N && iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW) {
                p_ok |= 1u << k;
                p_nbi |= (unsigned)nbi << (8 * k);
                p_rel[k] = ((nbi * P.IH + ry) * P.IW + rx) * P.Ci;
            }
        }
    }
    // per-(image, input channel) scales of the block's images and channel range; multiplied by the block exponent below
    if (!PKX && iscale) {
        for (int i = threadIdx.x; i < P.NB * cspan; i += CT_THREADS) {
            const int nbi = i / cspan, c = c_begin * CV_CK + (i - nbi * cspan);
            sct[i] = (n0 + nbi < P.N && c < P.Ci) ? iscale[(int64_t)(n0 + nbi) * P.Ci + c] : 0.f;
        }
    }
    float unscale = 1.f, xscale = 1.f;                        // set by block_exponent() once the first chunk is in registers
    // saturation tracking without a live VGPR (see wgrad.hip): raw item maximum against 65504 / (largest scale x exponent)
    [[maybe_unused]] float sat_thr = 3.0e38f;
    unsigned long long sat_bits = 0;
    float4 pq[CT_PITEMS];
    unsigned cur_ok = 0;
    auto issue_patch = [&](int chunk) {
        const int ci = chunk * CV_CK + c4 * 4;
        cur_ok = ci < P.Ci ? p_ok : 0u;
#pragma unroll
        for (int k = 0; k < CT_PITEMS; k++) {
            const bool ok = (cur_ok >> k) & 1u;
            pq[k] = *reinterpret_cast<const float4 *>(ok ? xt + p_rel[k] + chunk * CV_CK : g_ct2_zero_page);
        }
    };
    auto commit_patch = [&](int chunk, unsigned char *ph) {
        unsigned char *pl = ph + pbuf;
        const float *sc = sct + (chunk - c_begin) * CV_CK + c4 * 4;
        auto items = [&](auto ISC) {   // one block-uniform branch around the item loop: input scale x 2^e from the table, or 2^e from an SGPR
#pragma unroll
            for (int k = 0; k < CT_PITEMS; k++) {
                const float4 v = pq[k];       // (an out-of-range item has read the zero page)
#ifdef RICK_ABLATION      // (shipping build: off here, as in wgrad.hip — this kernel spills already; conv.hip tracks always)
                if constexpr (!PKX) sat_bits |= __builtin_amdgcn_ballot_w64(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) > sat_thr);
#endif
                if constexpr (PKX) {      // the item already IS {hi x 4 | lo x 4}
                    if (pix0 + (CT_THREADS / 8) * k < P.NPP) {
                        *reinterpret_cast<float2 *>(ph + p_lds0 + k * 4096) = make_float2(v.x, v.y);
                        *reinterpret_cast<float2 *>(pl + p_lds0 + k * 4096) = make_float2(v.z, v.w);
                    }
                    continue;
                }
                uint2 hi, lo;
                if constexpr (decltype(ISC)::value) split4v<SPLIT>(v, *reinterpret_cast<const float4 *>(sc + ((p_nbi >> (8 * k)) & 255u) * cspan), hi, lo);
                else split4s<SPLIT>(v, xscale, hi, lo);
                if (pix0 + (CT_THREADS / 8) * k < P.NPP) {        // (a branch around an LDS store is harmless; loads stay unconditional)
                    *reinterpret_cast<uint2 *>(ph + p_lds0 + k * 4096) = hi;
                    if (SPLIT == 2) *reinterpret_cast<uint2 *>(pl + p_lds0 + k * 4096) = lo;
                }
            }
        };
        if (!PKX && iscale) items(std::true_type{});
        else items(std::false_type{});
    };

    // ---- operand exponent of this block (conv_common.h): amax of |input scale * x| over the first channel chunk's patch
    // (already in registers: 32 channels x the whole window) and over samples of the other chunks, reduced over the
    // block -> x * 2^e.
    auto block_exponent = [&]() {
        if constexpr (PKX) {
            unscale = cv_uniform(iscale[1] * *reinterpret_cast<const float *>(wpk + (int64_t)P.ncot * P.nchunks * 9 * CV_WSTEP_BYTES));
            return;
        }
        float *red = reinterpret_cast<float *>(smem + 2 * pbuf);   // second patch buffer: not written before the chunk loop
        // samples of the block's other chunks (see conv.hip): 8 float4 per thread, issued before the first chunk is reduced
        const int ncl = c_end - c_begin;
        float4 sv[8];
        if (ncl > 1) {
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) {
                const int k = sidx % CT_PITEMS;
                const int chunk = c_begin + 1 + (sidx + (int)(threadIdx.x >> 3)) % (ncl - 1);
                const bool ok = ((p_ok >> k) & 1u) && chunk * CV_CK + c4 * 4 < P.Ci;
                sv[sidx] = *reinterpret_cast<const float4 *>(ok ? xt + p_rel[k] + chunk * CV_CK : g_ct2_zero_page);
            }
        }
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < CT_PITEMS; k++) {
            float4 v = pq[k];
            if (iscale) v = mul4(v, *reinterpret_cast<const float4 *>(sct + ((p_nbi >> (8 * k)) & 255u) * cspan + c4 * 4));
            m = amax4(m, v);
        }
        if (ncl > 1) {
#pragma unroll
            for (int sidx = 0; sidx < 8; sidx++) {
                const int k = sidx % CT_PITEMS;
                const int chunk = c_begin + 1 + (sidx + (int)(threadIdx.x >> 3)) % (ncl - 1);
                float4 v = sv[sidx];
                if (iscale) v = mul4(v, *reinterpret_cast<const float4 *>(sct + ((p_nbi >> (8 * k)) & 255u) * cspan + (chunk - c_begin) * CV_CK + c4 * 4));
                m = amax4(m, v);
            }
        }
        m = block_amax(m, red);
        float xs, xu;
        cv_pow2_scale(m, xs, xu);
        xscale = cv_uniform(xs);
        // packed-weight exponent (trailer of the packed image)
        unscale = cv_uniform(xu * *reinterpret_cast<const float *>(wpk + (int64_t)P.ncot * P.nchunks * 9 * CV_WSTEP_BYTES));
        sat_thr = 65504.f / xscale;
        if (iscale) {                                         // fold 2^e into the scale table
            float tm = 0.f;
            for (int i = threadIdx.x; i < P.NB * cspan; i += CT_THREADS) tm = fmaxf(tm, fabsf(sct[i] *= xscale));
            __syncthreads();
            sat_thr = cv_uniform(65504.f / fmaxf(block_amax(tm, red), 1e-30f));
            __syncthreads();
        }
    };

    // ---- B operand (patch) read offsets: lane slot j*16 + l15 -> tile position (position map) -> patch pixel (ty + 1, tx + 1)
    __syncthreads();                                          // position map in LDS
    const int tpos = P.TW * P.TH;
    int pb[8];
    unsigned slotpos[2] = {0, 0};                             // this lane's 8 table entries (positions of slots j * 16 + l15)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned ent = spos[(jbase + (j < NJ ? j : 0)) * 16 + l15];
        slotpos[j >> 2] |= ent << (8 * (j & 3));
        const int pos = (int)(ent & 127u);
        int nbi = pos / tpos;
        const int rem = pos - nbi * tpos;
        const int ty = rem / P.TW, tx = rem - ty * P.TW;
        nbi = nbi < P.NB ? nbi : P.NB - 1;                    // tile slots beyond NB images are masked in the epilogue
        pb[j] = (nbi * P.PH + ty + 1) * P.PW + tx + 1;
    }
    // ---- A operand (weights) fragment offset inside one packed 16 KB tap tile: row = h*64 + i*16 + l15
    const int arow = h * 64 + l15;
    const int a_off = arow * 64 + cv_swz(kg, arow) * 16;      // + i * 1024 (the swizzle key (row >> 2) & 1 does not depend on i)
    const unsigned char *wbase = wpk + (int64_t)cot * P.nchunks * 9 * CV_WSTEP_BYTES + a_off;
    auto tap_slice = [&](int t) {                             // weight slice ky*3 + kx of this class's tap t
        const int ty = t / nkx, tx = t - ty * nkx;
        return (py ? 1 : 2 * ty) * 3 + (px ? 1 : 2 * tx);
    };
    auto tap_off = [&](int t) {                               // patch offset (dy*PW + dx): ky == 2 reads the row above
        const int ty = t / nkx, tx = t - ty * nkx;
        return -((!py && ty) ? P.PW : 0) - ((!px && tx) ? 1 : 0);
    };

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A fragments of the NEXT tap roll into the registers the current tap releases: a tap multiplies positions 0-63
    // (all four 16-row groups), then positions 64-127; in that second half row group i is finished after its 12 MFMAs
    // and its registers are reloaded at once — 48 MFMAs (>= 768 cycles) before the next tap touches them again.
    // The order of the groups is pinned with sched_barrier (hipcc otherwise interleaves the groups and ends up waiting
    // vmcnt(0) for all eight loads at the top of every tap).
    f16x8 ahi[4], alo[4];
    auto load_a = [&](auto IC, const unsigned char *wt) {
        constexpr int i = decltype(IC)::value;
        ahi[i] = *reinterpret_cast<const f16x8 *>(wt + i * 1024);
        if (SPLIT == 2) alo[i] = *reinterpret_cast<const f16x8 *>(wt + CV_WTILE_BYTES + i * 1024);
    };
    auto mma_tap = [&](const unsigned char *ph, const unsigned char *pl, int toff, const unsigned char *wnext) {
        constexpr int NJH = NJ > 4 ? 2 : 1, NJJ = NJ < 4 ? NJ : 4;
#pragma unroll
        for (int jh = 0; jh < NJH; jh++) {
            f16x8 bhi[NJJ], blo[NJJ];
#pragma unroll
            for (int jj = 0; jj < NJJ; jj++) {
                const int pp = pb[jh * 4 + jj] + toff;
                const int off = pp * 64 + cv_swz(kg, pp) * 16;
                bhi[jj] = *reinterpret_cast<const f16x8 *>(ph + off);
                if (SPLIT == 2) blo[jj] = *reinterpret_cast<const f16x8 *>(pl + off);
            }
            static_for<0, 4>([&](auto IC) {
                constexpr int i = decltype(IC)::value;
#pragma unroll
                for (int jj = 0; jj < NJJ; jj++) {
                    f32x4 &a = acc[i][jh * 4 + jj];
                    if (SPLIT == 2) {
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[i], bhi[jj], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], blo[jj], a, 0, 0, 0);
                    }
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[i], bhi[jj], a, 0, 0, 0);
                }
                if (jh == NJH - 1) load_a(IC, wnext);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };

    // ---- prologue: patch(c_begin) -> buffer 0, A fragments of (c_begin, tap 0)
    issue_patch(c_begin);
    {
        const unsigned char *wt = wbase + ((int64_t)c_begin * 9 + tap_slice(0)) * CV_WSTEP_BYTES;
        static_for<0, 4>([&](auto IC) { load_a(IC, wt); });
    }
    __syncthreads();                                          // scale table complete
    block_exponent();
    commit_patch(c_begin, smem);
    __syncthreads();

    for (int chunk = c_begin; chunk < c_end; chunk++) {
        const int cur = (chunk - c_begin) & 1;
        const unsigned char *ph = smem + cur * 2 * pbuf, *pl = ph + pbuf;
        const int cnext = chunk + 1 < c_end ? chunk + 1 : chunk;     // past the end the last chunk is staged again (never used)
        issue_patch(cnext);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < ntap; t++) {
            // the tap after this one: next tap of the chunk, else tap 0 of the next chunk
            const int tn = t + 1 < ntap ? t + 1 : 0;
            const int cn = t + 1 < ntap ? chunk : cnext;
            mma_tap(ph, pl, tap_off(t), wbase + ((int64_t)cn * 9 + tap_slice(tn)) * CV_WSTEP_BYTES);
        }
        commit_patch(cnext, smem + (cur ^ 1) * 2 * pbuf);
        __syncthreads();   // buffer cur^1 written by everyone, buffer cur read by everyone
    }

    // ---- epilogue: class (py, px), position (gy, gx) -> output pixel (2*gy + py, 2*gx + px)
    const int GHc = P.IH + 1 - py, GWc = P.IW + 1 - px;
    const int64_t osz = (int64_t)P.N * P.OH * P.OW * P.Co;
    if (sat_bits != 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_cv_sat, 1u);
    const float oalpha = P.alpha * unscale;                   // exact: the exponents are powers of two
    float out_amax = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const unsigned ent = (slotpos[j >> 2] >> (8 * (j & 3))) & 255u;
        const int pos = (int)(ent & 127u);
        const int nbi = pos / tpos, rem = pos - nbi * tpos;
        const int ty = rem / P.TW, tx = rem - ty * P.TW;
        const int n = n0 + nbi, gy = gy0 + ty, gx = gx0 + tx;
        const int oy = 2 * gy + py, ox = 2 * gx + px;
        if ((ent & 128u) || nbi >= P.NB || n >= P.N || gy >= GHc || gx >= GWc || oy >= P.OH || ox >= P.OW) continue;
        const int64_t opix = ((int64_t)n * P.OH + oy) * P.OW + ox;
        if (P.nsplit > 1) {     // raw partial sums in output layout; scaled by the reduce kernel
            float *wrow = ws + (int64_t)split * osz + opix * P.Co;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int co = cot * CV_BM + h * 64 + i * 16 + kg * 4;
                if (co < P.Co)
                    *reinterpret_cast<float4 *>(wrow + co) = make_float4(acc[i][j][0] * unscale, acc[i][j][1] * unscale,
                                                                         acc[i][j][2] * unscale, acc[i][j][3] * unscale);
            }
            continue;
        }
        float *orow = out + opix * P.Co;
        float4 sc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int co = cot * CV_BM + h * 64 + i * 16 + kg * 4;
            sc[i] = make_float4(oalpha, oalpha, oalpha, oalpha);
            if (oscale) {
                const float4 o = *reinterpret_cast<const float4 *>(oscale + (int64_t)n * P.Co + (co < P.Co ? co : 0));
                sc[i] = make_float4(o.x * oalpha, o.y * oalpha, o.z * oalpha, o.w * oalpha);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int co = cot * CV_BM + h * 64 + i * 16 + kg * 4;
            if (co < P.Co) {
                const float4 o = make_float4(acc[i][j][0] * sc[i].x, acc[i][j][1] * sc[i].y, acc[i][j][2] * sc[i].z, acc[i][j][3] * sc[i].w);
                *reinterpret_cast<float4 *>(orow + co) = o;
                if constexpr (PKX) out_amax = amax4(out_amax, o);        // (only the split-image entry takes an amax word)
            }
        }
    }
    if constexpr (PKX)
        if (P.amax && P.nsplit == 1) cv_amax_publish(out_amax, P.amax, reinterpret_cast<float *>(smem));
}

// out[i] = alpha * oscale[n, co] * sum_s ws[s][i]   (ws in output layout; Co % 4 == 0)
__global__ __launch_bounds__(256) void convt2_reduce_kernel(const float *__restrict__ ws, float *__restrict__ out,
                                                            const float *__restrict__ oscale, int64_t n4, int64_t per_img4,
                                                            int co4, int nsplit, float alpha, float *__restrict__ amax) {
    float am = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 *src = reinterpret_cast<const float4 *>(ws) + i;
        float4 s = cv_sum_splits4(src, n4, nsplit);
        float4 sc = make_float4(alpha, alpha, alpha, alpha);
        if (oscale) {
            const int64_t n = i / per_img4;
            const int cq = (int)(i % co4);
            const float4 o = *(reinterpret_cast<const float4 *>(oscale) + n * co4 + cq);
            s.x *= alpha; s.y *= alpha; s.z *= alpha; s.w *= alpha;
            sc = o;
        }
        const float4 o = make_float4(s.x * sc.x, s.y * sc.y, s.z * sc.z, s.w * sc.w);
        reinterpret_cast<float4 *>(out)[i] = o;
        am = amax4(am, o);
    }
    if (amax) {
        __shared__ float red[4];
        cv_amax_publish(am, amax, red);
    }
}

static size_t ct2_lds_bytes(const Ct2Plan &p) { return 4 * (size_t)p.NPP * 64 + (size_t)p.NB * p.cps * CV_CK * 4 + 128; }

// Which tile position each lane slot (j, l15) of a B-operand read serves.  ds_read_b128 is serviced in four fixed groups
// of 16 lanes (MI355X_MICROARCH.md, LDS): {l15 in A = 0-3, 12-15 with k-group 2h; l15 in B = 4-11 with k-group 2h + 1} and
// the mirror image.  With 64-byte pixel rows and the slot key on bit 2 of the pixel index, a group is conflict-free iff
// its 8 A-lanes read pixels that are distinct mod 8, and its 8 B-lanes likewise.  16 consecutive pixels satisfy that;
// the (TW + 1)-pitched windows of this kernel's odd-sized tiles (6 x 5 x 4, 10 x 3 x 4, 11 x 11 ...) do not — every read
// was 2-way (46 % of the LDS cycles in round 2).  The tap offsets are constant shifts of the pixel index, which keep
// "distinct mod 8", so ONE assignment serves all taps: sort the tile's positions by (pixel index mod 8) and give the
// s-th position of every residue class to set s (16 sets of 8 = 8 j's x {A, B}).  Needs <= 16 positions per residue:
// the window pitch is padded by 0-3 unused columns until that holds (else the identity map: correct, with conflicts).
static void ct2_position_map(Ct2Plan *p) {
    static const int laneA[8] = {0, 1, 2, 3, 12, 13, 14, 15}, laneB[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    const int npos = p->TW * p->TH * p->NB, tpos = p->TW * p->TH;
    unsigned char map[128];
    for (int i = 0; i < 128; i++) map[i] = (unsigned char)(i < npos ? i : 128 | (npos - 1));
    const int base_pw = p->TW + 1;
    for (int pad = 0; pad <= 3; pad++) {
        const int PW = base_pw + pad, npp = p->NB * p->PH * PW;
        Ct2Plan q = *p;
        q.PW = PW;
        q.NPP = npp;
        if (npp > CT_MAX_NPP || ct2_lds_bytes(q) > 160 * 1024) break;
        int cls[8][128], cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int pos = 0; pos < npos; pos++) {
            const int nbi = pos / tpos, rem = pos % tpos, ty = rem / p->TW, tx = rem % p->TW;
            const int r = ((nbi * p->PH + ty + 1) * PW + tx + 1) & 7;
            cls[r][cnt[r]++] = pos;
        }
        int mx = 0;
        for (int r = 0; r < 8; r++) mx = cnt[r] > mx ? cnt[r] : mx;
        if (mx > 16) continue;
        for (int s = 0; s < 16; s++) {
            const int j = s >> 1;
            const int *lanes = (s & 1) ? laneB : laneA;
            int any = -1;
            for (int r = 0; r < 8 && any < 0; r++)
                if (s < cnt[r]) any = cls[r][s];
            for (int r = 0; r < 8; r++)     // a missing residue re-reads a pixel of its own set (same address: broadcast)
                map[j * 16 + lanes[r]] = (unsigned char)(s < cnt[r] ? cls[r][s] : 128 | (any >= 0 ? any : 0));
        }
        p->PW = PW;
        p->NPP = npp;
        break;
    }
    for (int i = 0; i < 32; i++)
        p->posw[i] = map[4 * i] | (map[4 * i + 1] << 8) | (map[4 * i + 2] << 16) | ((unsigned)map[4 * i + 3] << 24);
}

// Tile and split-K plan.  One 512-thread block (256 registers per thread) is resident per CU, so the grid runs in
// waves of 256 blocks: minimise  waves x (chunks per block x 9 taps + fixed cost)  over the position-tile shape
// (any TW x TH x NB with <= 128 positions: the class grids are (2^k + 1)-sized, powers of two would waste up to
// 2/3 of a tile on the small layers) and the channel split.
static int ct2_plan(int N, int IH, int IW, int Ci, int Co, int OH, int OW, float alpha, Ct2Plan *p) {
    if (N <= 0 || IH <= 0 || IW <= 0 || Ci <= 0 || Co <= 0 || (Ci & 3) || (Co & 3)) return RICK_EINVAL;
    if (OH < 2 * IH || OH > 2 * IH + 1 || OW < 2 * IW || OW > 2 * IW + 1) return RICK_EINVAL;
    if ((int64_t)N * IH * IW * Ci >= (1LL << 31) || (int64_t)N * OH * OW * Co >= (1LL << 31)) return RICK_EINVAL;
    p->N = N; p->IH = IH; p->IW = IW; p->Ci = Ci; p->OH = OH; p->OW = OW; p->Co = Co; p->alpha = alpha;
    p->nchunks = cdiv(Ci, CV_CK);
    p->ncot = cdiv(Co, CV_BM);
    const int GH = IH + 1, GW = IW + 1;
    long best = -1;
    for (int tw = 1; tw <= GW && tw <= 128; tw++)
        for (int th = 1; th <= GH && th * tw <= 128; th++) {
            int nb = 128 / (tw * th);
            if (nb > N) nb = N;
            const int npp = nb * (th + 1) * (tw + 1);
            if (npp > CT_MAX_NPP) continue;

            const long tiles = (long)cdiv(GW, tw) * cdiv(GH, th) * cdiv(N, nb) * p->ncot;
            for (int s = 1; s <= p->nchunks && s <= 16; s++) {
                const int cps = cdiv(p->nchunks, s), se = cdiv(p->nchunks, cps);
                // (two double-buffered hi/lo patch images + the [nb][cps * 32] scale table must fit the LDS)
                if (4 * (size_t)npp * 64 + (size_t)nb * cps * CV_CK * 4 > 160 * 1024) continue;
                // rough time model in ns: a block needs ~6 us per channel chunk (9 taps x 96 MFMAs per wave) + ~4 us fixed and
                // the grid runs in rounds of 256.  The last, partly filled round would cost a whole round: its R work items
                // are shared by 2 or 4 blocks each (4 / 2 of the tile's 8 fragment columns; CT_SUB2_PCT / CT_SUB4_PCT of a
                // whole block's time) when they still fit one round — 286 blocks: 2 rounds -> 1.64.
                const long blocks = tiles * se, full = blocks / 256, rest = blocks % 256;
                const long tb = cps * 6000L + 4000L;
                long tail = rest ? tb : 0;
                int subq = 1;
                if (rest && 2 * rest <= 256 && tb * CT_SUB2_PCT / 100 < tail) { tail = tb * CT_SUB2_PCT / 100; subq = 2; }
                if (rest && 4 * rest <= 256 && tb * CT_SUB4_PCT / 100 < tail) { tail = tb * CT_SUB4_PCT / 100; subq = 4; }
                // a split writes and re-reads se partial copies of the output (~4 TB/s) plus a second launch.  Patch rows
                // shorter than 8 pixels stage poorly (tie-break towards wide tiles).
                long cost = full * tb + tail + tiles * se * 40L + (tw < 8 ? 8 - tw : 0);
                if (se > 1) cost += (long)((double)N * OH * OW * Co * 4.0 * (2 * se + 1) / 4000.0) + 3000L;
                if (best < 0 || cost < best) {
                    best = cost;
                    p->TW = tw; p->TH = th; p->NB = nb; p->cps = cps; p->nsplit = se;
                    p->subq = subq;
                }
            }
        }
    if (best < 0) return RICK_EINVAL;
#ifdef RICK_ABLATION   // tile / split override for tools/bench_conv.py: RICK_CT2_TILE="tw,th,nb,nsplit"
    if (const char *ov = getenv("RICK_CT2_TILE")) {
        int tw, th, nb, ns;
        if (sscanf(ov, "%d,%d,%d,%d", &tw, &th, &nb, &ns) == 4 && tw * th * nb <= 128 && nb * (th + 1) * (tw + 1) <= CT_MAX_NPP) {
            p->TW = tw; p->TH = th; p->NB = nb < N ? nb : N;
            p->cps = cdiv(p->nchunks, ns); p->nsplit = cdiv(p->nchunks, p->cps);
            p->subq = 1;
        }
    }
    if (const char *ov = getenv("RICK_CT2_SUBQ")) p->subq = atoi(ov) == 4 ? 4 : atoi(ov) == 2 ? 2 : 1;   // (forced, even past one round)
#endif
    p->ntx = cdiv(GW, p->TW);
    p->nty = cdiv(GH, p->TH);
    p->ntn = cdiv(N, p->NB);
    {
        const long blocks = (long)p->ntx * p->nty * p->ntn * p->ncot * p->nsplit;
        p->nfull = p->subq > 1 ? (int)(blocks / 256 * 256) : (int)blocks;
    }
    p->PH = p->TH + 1;
    p->PW = p->TW + 1;
    p->NPP = p->NB * p->PH * p->PW;
    ct2_position_map(p);
    return 0;
}

extern "C" int64_t rick_convt2_workspace_bytes(int N, int IH, int IW, int Ci, int Co, int OH, int OW) {
    Ct2Plan p;
    if (ct2_plan(N, IH, IW, Ci, Co, OH, OW, 1.f, &p)) return -1;
    return p.nsplit > 1 ? (int64_t)p.nsplit * N * OH * OW * Co * 4 : 0;
}

static int convt2_run(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                      int N, int IH, int IW, int Ci, int Co, int OH, int OW, int split, float alpha,
                      void *workspace, void *stream, bool pkx, float *amax = nullptr);

extern "C" int rick_convt2_f32(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                               int N, int IH, int IW, int Ci, int Co, int OH, int OW, int split, float alpha,
                               void *workspace, void *stream) {
    return convt2_run(x, packed_w, out, iscale, oscale, N, IH, IW, Ci, Co, OH, OW, split, alpha, workspace, stream, false);
}

extern "C" int rick_convt2_split_f32(const void *x_split, const float *x_hdr, const void *packed_w, float *out,
                                     const float *oscale, int N, int IH, int IW, int Ci, int Co, int OH, int OW, float alpha,
                                     float *amax, void *workspace, void *stream) {
    if (!x_hdr || (Ci & 31)) return RICK_EINVAL;
    return convt2_run((const float *)x_split, packed_w, out, x_hdr, oscale, N, IH, IW, Ci, Co, OH, OW, 2, alpha, workspace, stream, true, amax);
}

static int convt2_run(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                      int N, int IH, int IW, int Ci, int Co, int OH, int OW, int split, float alpha,
                      void *workspace, void *stream, bool pkx, float *amax) {
    if (!x || !packed_w || !out || (split != 1 && split != 2)) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed_w | (uintptr_t)(iscale ? iscale : x) | (uintptr_t)(oscale ? oscale : x)) % 16)
        return RICK_EINVAL;
    Ct2Plan p;
    if (ct2_plan(N, IH, IW, Ci, Co, OH, OW, alpha, &p)) return RICK_EINVAL;
    p.amax = amax;
    if (p.nsplit > 1 && (!workspace || ((uintptr_t)workspace % 16))) return RICK_EINVAL;
    const size_t lds = ct2_lds_bytes(p);
    if (lds > 160 * 1024) return RICK_EINVAL;
    const int64_t nitems = (int64_t)p.ntx * p.nty * p.ntn * p.ncot * p.nsplit;
    if (nitems * p.subq > 0x7fffffff) return RICK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&](auto kern, int item0, int64_t nwg) {
        if (nwg <= 0) return;
        p.item0 = item0;
        RICK_LDS160_ONCE(kern);         // (generic lambda: one set of flags per kernel variant)
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(CT_THREADS), lds, st, x, (const unsigned char *)packed_w, out, iscale,
                           oscale, (float *)workspace, p);
    };
    const int64_t nsub = (nitems - p.nfull) * p.subq;
    if (pkx) {
        launch(convt2_kernel<2, 8, true>, 0, p.nfull);
        if (p.subq == 2) launch(convt2_kernel<2, 4, true>, p.nfull, nsub);
        else if (p.subq == 4) launch(convt2_kernel<2, 2, true>, p.nfull, nsub);
    } else if (split == 2) {
        launch(convt2_kernel<2, 8>, 0, p.nfull);
        if (p.subq == 2) launch(convt2_kernel<2, 4>, p.nfull, nsub);
        else if (p.subq == 4) launch(convt2_kernel<2, 2>, p.nfull, nsub);
    } else {
        launch(convt2_kernel<1, 8>, 0, p.nfull);
        if (p.subq == 2) launch(convt2_kernel<1, 4>, p.nfull, nsub);
        else if (p.subq == 4) launch(convt2_kernel<1, 2>, p.nfull, nsub);
    }
    if (p.nsplit > 1) {
        const int64_t n4 = (int64_t)N * OH * OW * Co / 4;
        int64_t nb = cdiv64(n4, 256);
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(convt2_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float *)workspace, out, oscale, n4,
                           (int64_t)OH * OW * Co / 4, Co / 4, p.nsplit, alpha, amax);
    }
    RICK_LAUNCH_STATUS();
}

// Plan introspection for tools/bench_conv.py and the tests: {TW, TH, NB, tiles, nsplit, cps, whole-tile blocks, blocks per
// remaining work item}.
extern "C" int rick_convt2_plan(int N, int IH, int IW, int Ci, int Co, int OH, int OW, int *out8) {
    Ct2Plan p;
    const int rc = ct2_plan(N, IH, IW, Ci, Co, OH, OW, 1.f, &p);
    if (rc || !out8) return RICK_EINVAL;
    out8[0] = p.TW; out8[1] = p.TH; out8[2] = p.NB; out8[3] = p.ntx * p.nty * p.ntn * p.ncot; out8[4] = p.nsplit; out8[5] = p.cps;
    out8[6] = p.nfull; out8[7] = p.subq;
    return 0;
}

// The lane-slot -> position table of the plan and its (padded) window pitch, for tools/lds_sim.py and the tests.
extern "C" int rick_convt2_posmap(int N, int IH, int IW, int Ci, int Co, int OH, int OW, unsigned char *out128, int *pitch) {
    Ct2Plan p;
    const int rc = ct2_plan(N, IH, IW, Ci, Co, OH, OW, 1.f, &p);
    if (rc || !out128 || !pitch) return RICK_EINVAL;
    for (int i = 0; i < 128; i++) out128[i] = (unsigned char)((p.posw[i >> 2] >> (8 * (i & 3))) & 255u);
    *pitch = p.PW;
    return 0;
}

CV_DEFINE_SAT_ACCESSOR(rick_sat_convt2)
