/* rick_hip.h — C ABI of librick_hip.so: the MI355X (gfx950) kernels behind the RICK
 * StyleGAN2 train-step hot path.
 *
 * Conventions (all entry points):
 *   - plain device pointers + explicit sizes; no framework types; fp32 tensors unless said;
 *   - launched asynchronously on `stream` (a hipStream_t passed as void*), never
 *     synchronises, never allocates; workspaces are caller-provided;
 *   - returns 0 on success, RICK_EINVAL for bad arguments, or 1000 + hipError_t from
 *     hipGetLastError() after the launch (the reference's extensions raise through
 *     TORCH_CHECK -> RuntimeError, op/upfirdn2d.cpp:8,15-16; the Python binding in
 *     rick_amd/_lib.py turns a non-zero status into RuntimeError);
 *   - re-entrant, no global mutable state (the reference ops are called from
 *     DataParallel worker threads, train_dynamic_update_prune.py:941-944).
 *
 * Activation layout: 4-D activations are channels-last ("NHWC": [N, H, W, C] in memory);
 * this is the reference extension's own [major, H, W, minor] view
 * (op/upfirdn2d_kernel.cu:209-240) with major = N, minor = C.  RGB-side tensors (3 channels)
 * stay planar ([N, 3, H, W], i.e. major = N*3, minor = 1).
 */
#ifndef RICK_HIP_H
#define RICK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RICK_EINVAL 22
#define RICK_MAX_TAPS 16

int rick_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * upfirdn2d — replaces upfirdn2d_op.upfirdn2d(input[M,H,W,minor], kernel, up_x, up_y,
 * down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1) (op/upfirdn2d.cpp:12-19,
 * op/upfirdn2d_kernel.cu:209-369).  out is [major, out_h, out_w, minor] with
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (kernel.cu:237-240); the caller
 * allocates it.  Forward, backward (flipped kernel, up<->down, op/upfirdn2d.py:19-60) and
 * double-backward are the same entry with different parameters.  Tap order and fmaf
 * accumulation match oracle/csrc/oracle_ops.c bit for bit. */
int rick_upfirdn2d_f32(const float *input, const float *kernel, float *out,
                       int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                       int up_x, int up_y, int down_x, int down_y,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream);

/* The same operator in the reference's other dtypes (op/upfirdn2d_kernel.cu:311-367 dispatches half / float / double):
 * dtype 1 = float64, 2 = float16 (fp32 accumulation); planar [major, H, W] layout (minor == 1), kernel taps in the same
 * dtype.  Correctness-first (one thread per output): for `.double()` gradcheck and `.half()` inference of drop-in users. */
int rick_upfirdn2d_any(const void *input, const void *kernel, void *out, int dtype, int64_t major, int in_h, int in_w,
                       int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                       int pad_y1, void *stream);

/* ---------------------------------------------------------------------------------------
 * fused bias + activation — replaces fused.fused_bias_act(input, bias, refer, act, grad,
 * alpha, scale) (op/fused_bias_act.cpp:11-21, op/fused_bias_act_kernel.cu:18-99):
 *   v = x[i] + (bias ? bias[(i / step_b) % size_b] : 0)
 *         + (noise ? nw[0] * noise[((i / n_div) % noise_nb) * noise_hw + (i / hw_div) % noise_hw] : 0)
 *   act*10+grad: 30: v>0 ? v : v*alpha   31: ref>0 ? v : v*alpha   10/11: v   12/32: 0
 *   out[i] = y * scale
 * NULL bias / ref / noise mean "absent" (the reference passes empty tensors).  The noise
 * term is this build's fusion of NoiseInjection (model_probe_tune.py:293-298) into the same
 * pass; nw is a device pointer to the scalar noise strength. */
int rick_bias_act_f32(const float *x, const float *bias, const float *ref, float *out,
                      int64_t n, int64_t step_b, int64_t size_b, int act, int grad,
                      float alpha, float scale,
                      const float *noise, const float *nw, int64_t n_div, int64_t hw_div,
                      int64_t noise_nb, int64_t noise_hw, void *stream);

/* Backward of the fused activation for a [rows, C] (channels-last) tensor in ONE pass:
 *   gx[r,c] = scale * g[r,c] * (ref[r,c] > 0 ? 1 : alpha)          (fused_act.py:28-30)
 *   gb[c]   = sum_r gx[r,c]                                          (fused_act.py:32-37)
 *   gnw     = sum_{r,c} gx[r,c] * noise[(r / rows_per_img % noise_nb)*noise_hw + r % noise_hw]
 * The per-channel / scalar sums use wavefront shuffles + a deterministic two-stage
 * reduction through `partials` (float[(C + 1) * rick_bias_act_bwd_blocks(rows, C)]).
 * gb / gnw / noise may be NULL.  accumulate != 0: the second stage ADDS the sums into gb / gnw (the caller passes
 * the parameters' gradient buffers: no temporary, no separate accumulation pass); one second-stage launch serves both. */
/* rick_bias_act_f32 without the noise term in dtype 1 = float64 / 2 = float16 (op/fused_bias_act_kernel.cu:79). */
int rick_bias_act_any(const void *x, const void *bias, const void *ref, void *out, int dtype, int64_t n, int64_t step_b,
                      int64_t size_b, int act, int grad, float alpha, float scale, void *stream);
int rick_bias_act_bwd_blocks(int64_t rows, int C);
int rick_bias_act_bwd_f32(const float *g, const float *ref, float *gx, float *gb, float *gnw,
                          const float *noise, int64_t rows, int C, int64_t rows_per_img,
                          int64_t noise_nb, int64_t noise_hw, float alpha, float scale,
                          float *partials, int accumulate, void *stream);
/* ... and, from the same pass, the demodulation gradient of the modulated convolution the activation follows
 * (model_probe_tune.py:246-252: y = act(d * conv(..) + noise + bias)):  gd[n, c] = (sum over the image's rows of gx * t) / divisor[n, c],
 * t = the convolution's output reconstructed from `ref` (ref / scale for ref > 0, ref / (scale * alpha) otherwise, minus
 * noise_w * noise and bias).  Needs rick_bias_act_bwd_dot_ok(): C % 4 == 0 and blocks that never straddle an image.
 * dpartials: rick_bias_act_bwd_blocks() * C floats.  (round 5: replaces a second read of gx and ref, rick_hw_dot_act_f32) */
int rick_bias_act_bwd_dot_ok(int64_t rows, int C, int64_t rows_per_img);
int rick_bias_act_bwd_dot_f32(const float *g, const float *ref, float *gx, float *gb, float *gnw, const float *noise,
                              int64_t rows, int C, int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw, float alpha,
                              float scale, float *partials, int accumulate, const float *bias, const float *noise_w,
                              const float *divisor, float *gd, float *dpartials, void *stream);

/* ---------------------------------------------------------------------------------------
 * Convolution family (replaces the F.conv2d / F.conv_transpose2d calls of
 * model_probe_tune.py:122,265,274,280 and their autograd).  One implicit-GEMM MFMA kernel,
 * parameterised by a tap table, covers 3x3/1x1, stride 1/2, transposed stride 2, and the
 * data-gradient of each.  fp32 in HBM; on the way into LDS an operand is multiplied by a
 * power of two (chosen per block from a sample of the block's own data, per packed tensor for
 * the weights) and split into fp16 hi + fp16 lo; the products hi*hi + hi*lo + lo*hi run on
 * v_mfma_f32_16x16x32_f16 with fp32 accumulation (split = 2: 2^-22 relative per product) or
 * hi*hi only (split = 1); the exponents are removed again in the epilogue (exact).
 *
 * Geometry: activations NHWC.  The launch covers a grid of GH x GW "positions" per image;
 * position (gy, gx) writes output pixel (gy*os + oy0, gx*os + ox0) of [N, OH, OW, Co] and
 * tap t reads input pixel (gy*is + dy[t], gx*is + dx[t]) of [N, IH, IW, Ci] (zero outside)
 * with packed weight slice wt[t]:
 *   out[n, pix, co] = alpha * oscale[n, co] * sum_t sum_ci W[wt[t]][co][ci] *
 *                     (iscale[n, ci] * x[n, inpix(t), ci])
 * oscale / iscale (NULL = 1) carry StyleGAN2's demodulation / modulation
 * (model_probe_tune.py:246-251) without materialising per-sample weights. */
typedef struct {
    int N, IH, IW, Ci;      /* input  [N, IH, IW, Ci] */
    int OH, OW, Co;         /* output [N, OH, OW, Co] */
    int GH, GW;             /* positions per image */
    int is, os, oy0, ox0;   /* input stride, output stride / offset */
    int ntaps;              /* taps used by this launch */
    int nslices;            /* tap slices in the packed weight / in gw (wt[t] < nslices) */
    int dy[RICK_MAX_TAPS], dx[RICK_MAX_TAPS], wt[RICK_MAX_TAPS];
    int split;              /* 1 = fp16, 2 = fp16x3 (fp32-grade) */
    float alpha;
} rick_conv_geom;

/* Packed-weight buffer size in bytes for `nslices` tap slices of a [Co, Ci] matrix (tiles + a 64-byte
 * trailer that holds the tensor's pack exponent). */
int64_t rick_conv_packed_bytes(int Co, int Ci, int nslices);
/* Pack W (element (co, ci, slice s) at w[co*s_co + ci*s_ci + s*s_t]) * scale into the MFMA
 * A-operand LDS image (fp16 hi/lo of W * scale * 2^e, swizzled, zero padded to 128 x 32 tiles; e from a
 * device-side sample of the tensor, stored in the trailer).  Two launches, no host synchronisation. */
int rick_conv_pack_weight(const float *w, int64_t s_co, int64_t s_ci, int64_t s_t,
                          int Co, int Ci, int nslices, float scale, int split, void *packed, void *stream);
/* The same packing for MANY weights in one launch (every convolution of a network after an optimiser step;
 * replaces one launch per layer and orientation).  `descs_device` is an array of n descriptors in DEVICE
 * memory; descriptor d owns blocks [blk_begin[d], blk_begin[d] + rick_conv_pack_blocks(Co, Ci)), in order;
 * total_blocks = the sum.  `w` need not be a whole tensor: any (s_co, s_ci, s_t) view, e.g. the transposed
 * view the data gradient uses. */
typedef struct {
    const float *w;
    int64_t s_co, s_ci, s_t;
    void *packed;           /* rick_conv_packed_bytes(Co, Ci, nslices) bytes */
    int Co, Ci, nslices;
    float scale;
    int blk_begin;
    int reserved;
} rick_pack_desc;
int rick_conv_pack_blocks(int Co, int Ci);
int rick_conv_pack_weights_multi(const rick_pack_desc *descs_device, int n, int total_blocks, int split, void *stream);
/* Launches with too few output tiles to fill the chip (the 4x4..32x32 512-channel layers) are
 * split over the channel-chunk dimension; partial sums go through `workspace`
 * (rick_conv_igemm_workspace_bytes bytes, 0 = not needed, may then be NULL) and a deterministic
 * second-stage kernel applies alpha / oscale. */
int64_t rick_conv_igemm_workspace_bytes(const rick_conv_geom *g);
/* Optional tail fused into the convolution's epilogue — replaces a separate rick_bias_act_f32 pass (one read and
 * one write of the whole activation) after EqualConv2d + FusedLeakyReLU (model_probe_tune.py:595-641) and after the
 * non-upsampling StyledConv (conv -> NoiseInjection -> FusedLeakyReLU, :314-348):
 *   out = gain * lrelu_slope(conv + bias[co] + noise_w[0] * noise[n % noise_nb][oy][ox])
 * in exactly that operation order (bit-identical to the two-pass form).  Needs Co % 4 == 0. */
typedef struct {
    const float *bias;      /* [Co] or NULL */
    const float *noise;     /* [noise_nb][OH*OW] or NULL */
    const float *noise_w;   /* device scalar, required with noise */
    int noise_nb;           /* 1 (shared map) or N */
    int act;                /* 0: none, 1: LeakyReLU(slope) * gain */
    float slope, gain;
    float *amax;            /* NULL, or an amax word: max |out| is folded into it (the exact bound the producer of the next
                             * split image needs; see "Split images" below) */
    /* also write out * split_scale[n, co] as a split image (the next modulated convolution's operand: its style scale
     * folded in; split_scale may be NULL), bounded by split_coef * (*split_bound) * ... — the caller's bound covers the scale.
     * Launches without split-K only (rick_conv_igemm_workspace_bytes(g) == 0). */
    void *split_out;
    float *split_hdr;
    const float *split_bound;   /* amax word */
    float split_coef;
    const float *split_scale;   /* [N, Co] or NULL */
} rick_conv_epilogue;
int rick_conv_igemm_f32(const float *x, const void *packed_w, float *out,
                        const float *iscale, const float *oscale,
                        const rick_conv_geom *g, void *workspace, void *stream);
/* rick_upfirdn2d_f32 with the same tail fused after the FIR (the blur that follows every upsampling StyledConv,
 * model_probe_tune.py:263-268,344-348): channels-last 4x4 taps, up = 1, minor % 64 == 0 only (else non-zero). */
int rick_upfirdn2d_act_f32(const float *input, const float *kernel, float *out,
                           int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                           int up_x, int up_y, int down_x, int down_y,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                           const rick_conv_epilogue *tail, void *stream);
/* rick_conv_igemm_f32 with the fused tail (epilogue may be NULL). */
int rick_conv_igemm_act_f32(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                            const rick_conv_geom *g, const rick_conv_epilogue *epilogue, void *workspace, void *stream);

/* Several geometries over the same tensors and weights (the 4 output-parity classes of a stride-2
 * transposed convolution) in one launch.  ngeom <= 4; all share Ci, Co, split. */
int64_t rick_conv_igemm_multi_workspace_bytes(const rick_conv_geom *geoms, int ngeom);
int rick_conv_igemm_multi_f32(const float *x, const void *packed_w, float *out,
                              const float *iscale, const float *oscale,
                              const rick_conv_geom *geoms, int ngeom, void *workspace, void *stream);

/* Transposed 3x3 stride-2 convolution, padding 0 — F.conv_transpose2d(stride=2) of the upsampling StyledConvs
 * (model_probe_tune.py:257-268) and the data gradient of the discriminator's stride-2 3x3 convolutions (:608-630):
 *   out[n, 2*iy+ky, 2*ix+kx, co] = alpha * oscale[n,co] * sum W[ky*3+kx][co][ci] * (iscale[n,ci] * x[n, iy, ix, ci])
 * All four output-parity classes come from ONE staged input patch per block (the generic multi-class launch above
 * stages it once per class).  packed_w = rick_conv_pack_weight(..., nslices = 9); out is [N, OH, OW, Co] with
 * OH in {2*IH, 2*IH+1} (likewise OW; the smaller size crops the last row / column); Ci % 4 == Co % 4 == 0.
 * Short grids are split over channel chunks through `workspace` (rick_convt2_workspace_bytes, 0 = not needed). */
int64_t rick_convt2_workspace_bytes(int N, int IH, int IW, int Ci, int Co, int OH, int OW);
int rick_convt2_f32(const float *x, const void *packed_w, float *out, const float *iscale, const float *oscale,
                    int N, int IH, int IW, int Ci, int Co, int OH, int OW, int split, float alpha,
                    void *workspace, void *stream);
/* The tile / split plan the launch above will use: out8 = {TW, TH, NB, tiles, nsplit, chunks per split, nfull, subq}:
 * tiles x nsplit work items; the first nfull run as one block each, every later one as subq blocks with 8 / subq of the
 * tile's fragment columns (the last, partly filled round of 256 blocks). */
int rick_convt2_plan(int N, int IH, int IW, int Ci, int Co, int OH, int OW, int *out8);
/* The plan's lane-slot -> tile-position table (128 entries, bit 7 = unused slot) and its patch-window pitch in pixels:
 * the permutation that makes the kernel's LDS reads bank-conflict-free (tools/lds_sim.py checks it). */
int rick_convt2_posmap(int N, int IH, int IW, int Ci, int Co, int OH, int OW, unsigned char *out128, int *pitch);

/* Weight gradient for the same geometry:
 *   gw[(co, ci, t)] = alpha * sum_{n, pos} (ascale[n,co] * gy[n, outpix(pos), co]) *
 *                                           (bscale[n,ci] * x[n, inpix(pos, t), ci])
 * written to gw[co*s_co + ci*s_ci + wt[t]*s_t].  Split-K over position tiles with a
 * deterministic second-stage reduction through `workspace`
 * (rick_conv_wgrad_workspace_bytes bytes).  `accumulate` != 0 adds into gw. */
int64_t rick_conv_wgrad_workspace_bytes(const rick_conv_geom *g);
int rick_conv_wgrad_f32(const float *x, const float *gy, float *gw,
                        int64_t s_co, int64_t s_ci, int64_t s_t,
                        const float *ascale, const float *bscale,
                        const rick_conv_geom *g, int accumulate, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Split images — activations / gradients stored pre-split for the MFMA kernels (no counterpart in the reference: its
 * convolutions are cuDNN's, model_probe_tune.py:122,265,274,280; this is how the fp16x3 arithmetic of this build avoids
 * converting the same fp32 operand once per co-tile block, again in dgrad and again in wgrad).
 * A split image of an NHWC fp32 tensor [npix, C] (C % 4 == 0) has the same byte size and addressing; the 16 bytes of
 * channels c .. c+3 hold {hi x 4 | lo x 4} as fp16, hi = fp16(v * 2^e), lo = fp16(v * 2^e - hi), with ONE
 * exponent per tensor in a 16-byte device header {2^e, 2^-e, bound, 0}.  The exponent comes from a guaranteed bound
 * coef * (*amax0 + *amax1) on |v| (amax1 may be NULL), placed in [2^13, 2^14): nothing can saturate, and values down to
 * 2^-10 of the bound keep 2^-22 relative precision.  A RUNNING MAXIMUM ("amax word": every `amax`, `amax0`, `bound0` ...
 * pointer below) is RICK_AMAX_FLOATS floats: RICK_AMAX_SLOTS slots, one per 128-byte line, the value being the maximum over
 * the slots (thousands of blocks folding into one address serialise in the L2).  Callers zero the word; rick_amax_f32 folds
 * max |x| into it with atomic max, the kernels that produce activations do the same in their epilogues. */
#define RICK_AMAX_SLOTS 16
#define RICK_AMAX_FLOATS (16 * 32)
int rick_amax_f32(const float *x, int64_t n, float *amax_word, void *stream);
/* Identity of the hipGraph capture `stream` is recording into (hipStreamGetCaptureInfo's id), 0 when it is not capturing.
 * Running maxima and headers a CAPTURED launch accumulates into must be zeroed by a fill inside the SAME graph: the host side
 * (rick_amd/op/split.py) starts a fresh block of words whenever this value changes. */
int rick_stream_capture_id(void *stream, unsigned long long *id);
/* What a producer kernel does with its result besides (or instead of) the fp32 store. */
typedef struct {
    void *split_out;        /* also write the result as a split image (NULL: no) */
    float *split_hdr;       /* its 16-byte header, published by the launch */
    const float *bound0;    /* |result| <= bound_coef * (*bound0 + *bound1); bound1 may be NULL */
    const float *bound1;
    float bound_coef;
    float *amax;            /* NULL, or: fold max |result| (after accumulation) into this word */
    int accumulate;         /* add the result to what the fp32 output holds (a second gradient arriving at a branch point) */
    int no_f32;             /* do not write the fp32 result (the output pointer may be NULL) */
    const float *chan_scale;/* NULL, or [N, C]: the split image holds result * chan_scale[n, c] (a modulated convolution's
                             * style / demodulation scale folded in; the bound must cover it) */
    /* activation adjoint fused behind the FIR (4x4 up = 1 form only): result <- result * (adj_ref > 0 ? 1 : adj_slope) * adj_gain
     * with adj_ref the activation's saved OUTPUT (same shape as the result), and the per-channel sums of that (the bias
     * gradient) reduced per block into adj_partials [blocks per channel slab = images * tiles][C]
     * (rick_upfirdn2d_adjoint_rows() rows; the caller column-sums them: rick_colsum_f32). */
    const float *adj_ref;
    float adj_slope, adj_gain;
    float *adj_partials;
} rick_split_out;
/* Saturation events since the last reset, host-synchronous: values that exceeded a split-image producer's bound (a caller's
 * mistake) and waves of the igemm / convt2 / wgrad kernels that clamped an fp32 -> fp16 conversion of an on-the-fly split
 * (an operand ~8 000 x above the sampled block maximum: the hardware's sticky OVERFLOW status, read once per wave).  Every
 * event means a finite but wrong product somewhere: 0 in every test and in tools/stability.py. */
int rick_saturation_count(unsigned *count, int reset);
/* Kernel-form switches (process-wide, host side; every form computes the same convolution — the eight-wave igemm is
 * bit-identical to the four-wave one on split images and differs by per-block operand exponents, 2^-22, on fp32 operands).
 * Used by the parity tests and the same-process A/B micro-benchmarks; returns the previous value, -1 for an unknown key.
 *   RICK_TUNE_IGEMM_W8: 3x3 stride-1 launches with >= RICK_TUNE_IGEMM_W8_MINBLK (default 192) blocks of 128 co x 256
 *   positions run the eight-wave form (conv.hip, igemm_body NW = 8) — 0 (default): never (four-wave 128 x 128 blocks: equal or
 *   faster end to end, conv.hip igemm_w8_plan); 1: fp32 operands with Ci >= 512; 2: wherever the form exists.
 *   RICK_TUNE_SPLITK_FUSED (default 0: measured 8 - 17 us per launch SLOWER than the second-stage launch, conv.hip): 1 = igemm
 *   split-K launches sum their partial tiles inside the launch — the block that arrives last for an output tile (one atomic
 *   ticket per tile) adds the partials in split order, exactly the sums of igemm_splitk_reduce_kernel. */
#define RICK_TUNE_IGEMM_W8 0
#define RICK_TUNE_IGEMM_W8_MINBLK 1
#define RICK_TUNE_SPLITK_FUSED 2
#define RICK_TUNE_IGEMM_S2W8 4      /* (default 2) 3x3 stride-2 launches with Co % 256 == 0 and >= RICK_TUNE_IGEMM_W8_MINBLK blocks: eight-wave 256 co x 128
                                     * position blocks (conv.hip, igemm_body WDMA = 5: +8 ... +16 % per launch, +1.05 % end to end); 1: Ci >= 256 only; 0: four-wave */
#define RICK_TUNE_UFD_TILE16 3      /* (default 0: measured 0.5 % slower end to end) 1 = 4x4 FIR, up = down = 1, outputs >= 32 x 32: 16 x 16 tiles on 32-channel slabs instead of 8 x 8 x 64 */
int rick_conv_tuning(int key, int value);
/* Producers.  rick_upfirdn2d_f32 / rick_upfirdn2d_act_f32 (tail may be NULL) with the extended result handling, channels-last
 * only; `out` may be NULL with ex->no_f32.  The activation adjoint (rick_bias_act_bwd_f32) leaving as split images:
 * out1 = g * (ref > 0 ? 1 : alpha) * scale and, when out2 != NULL, out2 = g * mul2 (the same gradient entering a parallel linear
 * branch), both bounded through *amax_g >= max |g|.  The ResBlock merge y = (a + b) * alpha (rick_add_scale_f32) written as
 * fp32 and as a split image. */
int rick_upfirdn2d_ex_f32(const float *input, const float *kernel, float *out,
                          int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                          int up_x, int up_y, int down_x, int down_y,
                          int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                          const rick_conv_epilogue *tail, const rick_split_out *ex, void *stream);
/* rows of adj_partials for an output of out_h x out_w pixels and `major` images (4x4 up = 1 down = 1 form: 8 x 8 tiles) */
int64_t rick_upfirdn2d_adjoint_rows(int64_t major, int out_h, int out_w);
/* out[c] (+)= sum_r partials[r * stride + c], c < ncols: the deterministic second stage of the per-block channel sums */
int rick_colsum_f32(const float *partials, float *out, int64_t rows, int stride, int ncols, int accumulate, void *stream);
/* Up to any number of such column sums in one launch per 32 items (bit-identical to rick_colsum_f32 item by item): out[c] (c < split,
 * or all columns when out2 is NULL) / out2[c - split] = (accumulate ? old : 0) + sum_r partials[r * stride + col0 + c], r < nb.
 * `items` is a HOST array: it travels in the kernel arguments, so the call may be captured into a hipGraph.  Used for the second
 * stages of the bias / noise-strength gradients of a whole backward pass (rick_bias_act_bwd_f32 with accumulate bit 1 set leaves
 * its partial rows [blocks][C + 1] unsummed: column C is the noise-strength term). */
#define RICK_COLSUM_MAX 32
typedef struct rick_colsum_item {
    const float *partials;
    float *out, *out2;
    int nb, stride, ncols, col0, split, accumulate;
} rick_colsum_item;
int rick_colsum_multi_f32(const rick_colsum_item *items, int n, void *stream);
int rick_bias_act_bwd_split_f32(const float *g, const float *ref, void *out1, float *hdr1, void *out2, float *hdr2,
                                float mul2, const float *amax_g, float *gb, float *gnw, const float *noise,
                                int64_t rows, int C, int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw,
                                float alpha, float scale, float *partials, int accumulate, void *stream);
/* The discriminator's input layer (model_probe_tune.py:679: 1x1 conv of the planar image t [N, J, P] with W [J, C], then
 * FusedLeakyReLU) in one pass: x[n,p,c] = gain * lrelu(sum_j t[n,j,p] * W[j,c] + bias[c]); ex (may be NULL) adds a split image. */
int rick_d_input_f32(const float *t, const float *W, const float *bias, float *x, int N, int64_t P, int C, int J,
                     float slope, float gain, const rick_split_out *ex, void *stream);
/* Bound of an activation tail for the producers above, evaluated on the device (one tiny launch, no host sync):
 *   out[slot 0] = max|mul| * gain * (c0 * amax(w0) + max|nw| * amax(w_noise) + max|bias|)
 * — |gain * lrelu(v + nw * noise + bias) * mul[n,c]| for |v| <= c0 * amax(w0).  Any of nw / w_noise, bias, mul may be NULL
 * (term dropped / factor 1).  out is an amax word (the other slots are zeroed). */
int rick_bound_tail_f32(float *out_word, const float *w0, float c0, const float *nw, const float *w_noise,
                        const float *bias, int nbias, float gain, const float *mul, int nmul, void *stream);
/* rick_bias_act_bwd_split_f32 for modulated layers: image 1 = adjoint * chan_scale[n, c] (chan_scale [rows / rows_per_img, C]),
 * bounded by the amax word bound1 alone (|scale| * max |chan_scale| * max |g|, combined by rick_bound_tail_f32); out_f32
 * (may be NULL) also receives the unscaled adjoint. */
int rick_bias_act_bwd_split2_f32(const float *g, const float *ref, void *out1, float *hdr1, void *out2, float *hdr2,
                                 float mul2, const float *amax_g, const float *chan_scale, const float *bound1,
                                 float *out_f32, float *gb, float *gnw, const float *noise,
                                 int64_t rows, int C, int64_t rows_per_img, int64_t noise_nb, int64_t noise_hw,
                                 float alpha, float scale, float *partials, int accumulate, void *stream);
int rick_add_scale_split_f32(const float *a, const float *b, float *y, void *y_split, float *hdr,
                             const float *amax_a, const float *amax_b, int64_t rows, int C, float alpha, void *stream);
int rick_split_pack_f32(const float *x, void *out, float *hdr, const float *amax0, const float *amax1, float coef,
                        int64_t npix, int C, void *stream);
int rick_split_unpack_f32(const void *split, const float *hdr, float *out, int64_t npix, int C, void *stream);
/* Weight gradient (rick_conv_wgrad_f32) with one or both operands given as split images: x_hdr / gy_hdr non-NULL marks
 * the operand as a split image (its per-channel scales, if the layer has any, are already folded in by the producer).
 * Only geometries for which rick_conv_wgrad_split_supported() returns 1 (whole 64-position tiles, Co % 128 == 0,
 * Ci % 32 == 0: every 3x3 / 1x1 layer of both networks at >= 8x8). */
int rick_conv_wgrad_split_supported(const rick_conv_geom *g);
/* rick_conv_igemm_act_f32 / rick_convt2_f32 with the input given as a split image (Ci % 32 == 0; per-channel input scales
 * are folded into the image by its producer).  The igemm entry serves the 3x3 stride-1 / stride-2 forms with a full grid and
 * the small-patch (1x1, parity-class) form; other geometries return RICK_EINVAL and take the fp32 entry. */
int rick_conv_igemm_split_supported(const rick_conv_geom *g);
int rick_conv_igemm_split_f32(const void *x_split, const float *x_hdr, const void *packed_w, float *out,
                              const float *oscale, const rick_conv_geom *g, const rick_conv_epilogue *epilogue,
                              void *workspace, void *stream);
int rick_convt2_split_f32(const void *x_split, const float *x_hdr, const void *packed_w, float *out, const float *oscale,
                          int N, int IH, int IW, int Ci, int Co, int OH, int OW, float alpha, float *amax /* word or NULL */,
                          void *workspace, void *stream);
int rick_conv_wgrad_split_f32(const void *x, const float *x_hdr, const void *gy, const float *gy_hdr, float *gw,
                              int64_t s_co, int64_t s_ci, int64_t s_t, const rick_conv_geom *g, int accumulate,
                              void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Thin (J <= 4 channel) products for the RGB side (ToRGB 1x1 modulated conv,
 * model_probe_tune.py:351-370; discriminator input conv, :679).  x is NHWC [N, P, C],
 * the thin tensor is planar [N, J, P], W is [N or 1, J, C] (w_bstride = J*C or 0).
 *   fwd :  t[n,j,p]  = sum_c x[n,p,c] * W[n,j,c]  (+ add[n,j,p] if add)
 *   bwdx:  x[n,p,c]  = sum_j t[n,j,p] * W[n,j,c]
 *   wgrad: G[n,j,c]  = sum_p t[n,j,p] * x[n,p,c]   (partials: float[N*J*C*rick_thin_wgrad_blocks]) */
int rick_thin_fwd_f32(const float *x, const float *W, int64_t w_bstride, const float *add,
                      float *t, int N, int64_t P, int C, int J, void *stream);
int rick_thin_bwdx_f32(const float *t, const float *W, int64_t w_bstride, float *x,
                       int N, int64_t P, int C, int J, void *stream);
/* ToRGB (model_probe_tune.py:246-248, 366-370) with the per-sample weight formed on the fly,
 * W[n,j,c] = (wscale * w[j,c]) * s[n,c]  (w [J, C] shared, s [N, C] the style; both 16-byte aligned):
 *   fwd :  t[n,j,p] = sum_c x[n,p,c] * W[n,j,c] + bias[j] (+ add[n,j,p])     (bias / add may be NULL)
 *   bwdx:  gx[n,p,c] = sum_j g[n,j,p] * W[n,j,c] */
int rick_torgb_fwd_f32(const float *x, const float *w, const float *s, float wscale, const float *bias,
                       const float *add, float *t, int N, int64_t P, int C, int J, void *stream);
int rick_torgb_bwdx_f32(const float *g, const float *w, const float *s, float wscale, float *gx, int N, int64_t P,
                        int C, int J, void *stream);
/* ... added INTO gx (gx[n,p,c] += ...): the activation that feeds ToRGB also feeds the next layer, whose data gradient is
 * already in gx — the second gradient arriving at the branch point is added by the kernel that produces it (same fp32 addition
 * as autograd's accumulation: torch.equal). */
int rick_torgb_bwdx_acc_f32(const float *g, const float *w, const float *s, float wscale, float *gx, int N, int64_t P,
                            int C, int J, void *stream);
int rick_thin_wgrad_blocks(int64_t P);
int rick_thin_wgrad_f32(const float *t, const float *x, float *G, int N, int64_t P, int C, int J,
                        float *partials, void *stream);

/* ---------------------------------------------------------------------------------------
 * Elementwise / reduction helpers on NHWC [N, P, C] tensors. */
/* y[n,p,c] = x[n,p,c] * s[n,c] */
int rick_chan_scale_f32(const float *x, const float *s, float *y, int N, int64_t P, int C, void *stream);
/* d[n,c] = sum_p a[n,p,c] * b[n,p,c] (/ divisor[n,c] when divisor != NULL); partials: float[N*C*rick_hw_dot_blocks(P)] */
int rick_hw_dot_blocks(int64_t P);
int rick_hw_dot_f32(const float *a, const float *b, float *d, int N, int64_t P, int C,
                    float *partials, const float *divisor, void *stream);
/* rick_hw_dot_f32 that also writes scaled[n,p,c] = a[n,p,c] * scale[n,c] in the same pass (C % 4 == 0). */
int rick_hw_dot_scale_f32(const float *a, const float *b, float *d, const float *scale, float *scaled, int N,
                          int64_t P, int C, float *partials, void *stream);
/* d[n,c] = sum_p g[n,p,c] * (lrelu^-1(y[n,p,c]) - noise_w[0]*noise[n % noise_nb][p] - bias[c]) where y is the output
 * of a fused tail gain*lrelu_slope(v + bias + noise_w*noise): the sum over pixels of g times the PRE-tail value v,
 * without keeping v (demodulation gradient of the fused StyledConv).  C % 4 == 0; partials and divisor as
 * rick_hw_dot_f32. */
int rick_hw_dot_act_f32(const float *g, const float *y, float *d, int N, int64_t P, int C, const float *bias,
                        const float *noise, const float *noise_w, int noise_nb, float slope, float gain,
                        float *partials, const float *divisor, void *stream);
/* y = (a + b) * alpha  (ResBlock merge, model_probe_tune.py:658); b may be NULL */
int rick_add_scale_f32(const float *a, const float *b, float *y, int64_t n, float alpha, void *stream);

/* Minibatch standard deviation (model_probe_tune.py:748-756) on NHWC x[B, P, C]: `groups` runs of
 * B/groups consecutive samples, each treated like one discriminator call whose stddev group is its whole
 * batch: out[B, P, C+1], channel C = mean_{p,c} sqrt(var_b(x) + 1e-8) of the sample's run; backward.
 * (groups = 2 lets D(fake) and D(real) of the D step share one pass with unchanged results.) */
int rick_mbstd_fwd_f32(const float *x, float *out, float *stat, int B, int P, int C, int groups, void *stream);
int rick_mbstd_bwd_f32(const float *x, const float *gout, float *gx, int B, int P, int C, int groups, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fisher information + optimiser side (train_dynamic_update_prune.py:252-269, 279-299,
 * 427-438, 68-73, 908-931). */
/* acc[i] += g[i]^2 */
int rick_sq_accumulate_f32(float *acc, const float *g, int64_t n, void *stream);
/* out[f] = scale * sum_{o, i} x[o*outer_stride + f*filter_stride + i], i < inner
 * (per-filter mean of a Fisher tensor: one wavefront-shuffle reduction per filter) */
int rick_filter_reduce_f32(const float *x, float *out, int64_t outer, int64_t outer_stride,
                           int64_t nfilters, int64_t filter_stride, int64_t inner,
                           float scale, void *stream);
/* Style demodulation of the modulated convolution (model_probe_tune.py:246-252; w is [O, I, K] contiguous):
 *   rick_wsq_f32          wsq[o,i] = scale^2 * sum_k w[o,i,k]^2
 *   rick_demod_f32        d[b,o]   = rsqrt(sum_i s[b,i]^2 wsq[o,i] + eps)                      (B <= 32)
 *   rick_demod_bwd_s_f32  gs[b,i]  = 2 s[b,i] sum_o t[b,o] wsq[o,i],   t = -0.5 d^3 gd
 *   rick_demod_bwd_w_f32  gw[o,i,k] = 2 scale^2 w[o,i,k] sum_b t[b,o] s[b,i]^2 */
int rick_wsq_f32(const float *w, float *wsq, int O, int I, int K, float scale, void *stream);
int rick_demod_f32(const float *s, const float *wsq, float *d, int B, int I, int O, float eps, void *stream);
int rick_demod_bwd_s_f32(const float *s, const float *wsq, const float *d, const float *gd, float *gs, int B, int I, int O,
                         void *stream);
int rick_demod_bwd_w_f32(const float *w, const float *s, const float *d, const float *gd, float *gw, int B, int I, int O,
                         int K, float scale, int accumulate, void *stream);
/* Modulation bank: s_l = EqualLinear_l(latent[:, idx_l]) for EVERY modulated convolution of the generator
 * (ModulatedConv2d.modulation, model_probe_tune.py:233,246) in one launch, and the weight / bias gradients of all of
 * them in one more.  lat is [B, n_latent, K]; layer l owns blocks [blk_begin, blk_begin + rick_modbank_blocks(C_l));
 *   fwd: out[io_off + b*C + c] = scale * sum_k lat[b, lat_idx, k] * w[c*K + k] + b[c]
 *   bwd: grad[gw_off + c*K + k] = scale * sum_b gs[io_off + b*C + c] * lat[b, lat_idx, k];  grad[gb_off + c] = sum_b gs[..]
 * (gw_off / gb_off < 0: no weight / bias gradient; a layer with neither is skipped and its gs is not read;
 * accumulate != 0: the gradients are ADDED to what grad holds — grad + offsets then address the parameters' own
 * gradient buffers).  `descs_device` lives in device memory.  K % 256 == 0, B <= 8, grad 16-byte aligned. */
typedef struct {
    const float *w;         /* [C, K] */
    const float *b;         /* [C] or NULL */
    int64_t io_off;         /* float offset of s_l / gs_l ([B, C] contiguous) */
    int64_t gw_off, gb_off; /* float offsets of the weight / bias gradient */
    int C, lat_idx, blk_begin, reserved;
} rick_modbank_desc;
int rick_modbank_blocks(int C);
int rick_modbank_fwd_f32(const float *lat, int B, int n_latent, int K, const rick_modbank_desc *descs_device, int n,
                         int total_blocks, float scale, float *out, void *stream);
int rick_modbank_bwd_f32(const float *lat, const float *gs, int B, int n_latent, int K, const rick_modbank_desc *descs_device,
                         int n, int total_blocks, float scale, float *grad, int accumulate, void *stream);
/* Demodulation bank: the demodulation coefficients of EVERY demodulated convolution of a generator (model_probe_tune.py:246-252)
 * in one launch, their weight gradient in one more, wsq[o,i] = scale^2 sum_k w[o,i,k]^2 in a third that only runs when the
 * weights have changed.  Same arithmetic and order per element as rick_wsq_f32 / rick_demod_f32 / rick_demod_bwd_w_f32
 * (bit-identical).  Layer l reads s_l [B, I] at float offset s_off of the modulation bank's flat output and owns d_l / gd_l
 * [B, O] at d_off of the flat coefficient / gradient buffers; gw (the parameter's gradient buffer, accumulated into) may be NULL.
 * Block ranges: [blk_wsq, next) of rick_demod_blocks_wsq(O, I) blocks for the wsq and weight-gradient launches,
 * [blk_demod, next) of rick_demod_blocks(O) for the coefficient launch, [blk_bwd_s, next) of rick_demod_blocks_bwd_s(I) for the
 * style gradient gs_l [B, I] (written at s_off of gs_flat).  B <= 8. */
typedef struct rick_demod_desc {
    const float *w;         /* [O, I, K] contiguous */
    float *wsq;             /* [O, I] */
    float *gw;              /* [O, I, K] or NULL */
    int64_t s_off, d_off;
    int O, I, K;
    float scale2;           /* (conv scale)^2 */
    int blk_wsq, blk_demod, blk_bwd_s, reserved;
} rick_demod_desc;
int rick_demod_blocks_wsq(int O, int I);
int rick_demod_blocks(int O);
int rick_wsq_multi_f32(const rick_demod_desc *descs_device, int n, int total_blocks, void *stream);
int rick_demod_multi_f32(const float *s_flat, float *d_flat, const rick_demod_desc *descs_device, int n, int total_blocks,
                         int B, int max_I, float eps, void *stream);
int rick_demod_bwd_w_multi_f32(const float *s_flat, const float *d_flat, const float *gd_flat,
                               const rick_demod_desc *descs_device, int n, int total_blocks, int B, void *stream);
int rick_demod_blocks_bwd_s(int I);
int rick_demod_bwd_s_multi_f32(const float *s_flat, const float *d_flat, const float *gd_flat, float *gs_flat,
                               const rick_demod_desc *descs_device, int n, int total_blocks, int B, int max_O, void *stream);
/* EqualLinear forward on a short batch (the mapping network, model_probe_tune.py:139-173, 418-428; no autograd):
 *   x' = pixelnorm ? x * rsqrt(mean_k x^2 + 1e-8) : x                                  (PixelNorm, :92-98)
 *   out[b,o] = act(scale * sum_k x'[b,k] W[o,k] + bias[o] * bias_mul);  act != 0: gain * leaky_relu(., slope)
 * x [B, K], W [O, K], bias [O] or NULL, out [B, O].  B <= 16, K % 4 == 0, x / W 16-byte aligned. */
int rick_equal_linear_f32(const float *x, const float *W, const float *bias, float *out, int B, int K, int O,
                          float scale, float bias_mul, int act, float slope, float gain, int pixelnorm, void *stream);
/* EqualLinear WITH gradients on a short batch (the discriminator's final layers, model_probe_tune.py:699-702, in every D pass;
 * F.linear(input, weight * scale, bias * lr_mul) of :157-168): three products closed under differentiation, each one pass
 * over the [O, K] matrix, summed in a fixed order (bit-reproducible run to run — the library GEMMs they replace may use
 * atomically accumulated split-K solutions for M = batch, K = 8192).  B <= 16, K % 4 == 0, matrices 16-byte aligned.
 *   fwd   out[b,o] = alpha * sum_k x[b,k] W[o,k] + bias_mul * bias[o]          (bias may be NULL)
 *   dgrad out[b,k] = alpha * sum_o g[b,o] W[o,k]
 *   wgrad gw[o,k] (+)= alpha * sum_b g[b,o] x[b,k];  gb[o] (+)= bias_mul * sum_b g[b,o]   (gw or gb may be NULL)
 * `workspace`: rick_linear_{fwd,dgrad}_workspace_floats() floats (0: may be NULL), 16-byte aligned. */
int64_t rick_linear_fwd_workspace_floats(int B, int K, int O);
int rick_linear_fwd_f32(const float *x, const float *W, const float *bias, float *out, int B, int K, int O, float alpha,
                        float bias_mul, float *workspace, void *stream);
int64_t rick_linear_dgrad_workspace_floats(int B, int K, int O);
int rick_linear_dgrad_f32(const float *g, const float *W, float *out, int B, int K, int O, float alpha, float *workspace,
                          void *stream);
int rick_linear_wgrad_f32(const float *g, const float *x, float *gw, float *gb, int B, int K, int O, float alpha,
                          float bias_mul, int accumulate, void *stream);

/* Masked Adam over a flat parameter buffer (mask bits: 1 = freeze (grad := 0),
 * 2 = prune (param := 0, grad := 0); mask may be NULL), torch.optim.Adam semantics
 * (no weight decay, no amsgrad), bias corrections passed in. */
int rick_masked_adam_f32(float *p, float *g, float *m, float *v, const uint8_t *mask, int64_t n,
                         float lr, float beta1, float beta2, float eps, float bc1, float bc2,
                         void *stream);
/* The same update with the step counters and bias corrections kept in DEVICE memory, so that a whole train step
 * (incl. the optimiser) can be captured once into a hipGraph and replayed: rick_adam_prepare_f32 increments
 * steps[first .. first+count) (int32, all equal) and writes bc = {1 - beta1^t, 1 - beta2^t} for the new count t;
 * rick_masked_adam_dev_f32 is rick_masked_adam_f32 reading the two corrections from `bc`. */
int rick_adam_prepare_f32(int *steps, int first, int count, float beta1, float beta2, float *bc, void *stream);
int rick_masked_adam_dev_f32(float *p, float *g, float *m, float *v, const uint8_t *mask, int64_t n,
                             float lr, float beta1, float beta2, float eps, const float *bc, void *stream);
/* ema[i] = ema[i]*decay + p[i]*(1-decay) */
int rick_ema_f32(float *ema, const float *p, int64_t n, float decay, void *stream);

/* ---------------------------------------------------------------------------------------
 * Data path (dataset.py:8-40, train_dynamic_update_prune.py:789-843).  The dataset stays resident in device memory as
 * uint8 [N, H, W, 3]; one launch builds a training batch: out[b, c, y, x] = (images[index[b], y, flip[b] ? W-1-x : x, c]
 * / 255 - 0.5) / 0.5 — ToTensor + RandomHorizontalFlip + Normalize(0.5, 0.5) of the reference transform, [B, 3, H, W]. */
int rick_image_batch_f32(const uint8_t *images, const int64_t *index, const uint8_t *flip, float *out,
                         int N, int H, int W, int B, void *stream);
/* PNG scanline reconstruction (filter types 0-4), in place, HOST memory: H rows of 1 filter byte + stride bytes. */
int rick_png_unfilter(uint8_t *data, int H, int stride, int bpp);

#ifdef __cplusplus
}
#endif
#endif /* RICK_HIP_H */
