// Device helpers shared by the MFMA convolution kernels (conv.hip, convt2.hip): fp16 hi/lo split, the swizzled
// [row][32 k] LDS image, vector loads, XCD-aware block remap.
#pragma once
#include "common.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define CV_BM 128
#define CV_BN 128
#define CV_CK 32
#define CV_WTILE_BYTES (CV_BM * CV_CK * 2)      // one of hi / lo: 8 KB
#define CV_WSTEP_BYTES (2 * CV_WTILE_BYTES)     // hi + lo: 16 KB
#define CV_WTRAILER_BYTES 64                    // behind the packed tiles: float[0] = 2^-e of the tensor's pack exponent

__host__ __device__ __forceinline__ int cv_swz(int kg, int row) { return kg ^ (((row >> 2) & 1) << 1); }

// ---- fp16 hi/lo split with a power-of-two operand exponent ------------------------------------------------------
// An fp32 operand x is multiplied by 2^e (e chosen per block from a sample of the block's own data, so that the sampled
// maximum lands in [2^T, 2^(T+1)), T = CV_EXP_TARGET), then split  x*2^e = hi + lo  with hi = fp16(x*2^e) and
// lo = fp16(x*2^e - hi), both rounded to nearest even (the difference is exact in fp32).  The three products
// hi*hi + hi*lo + lo*hi accumulate in ONE fp32 accumulator (lo carries no extra factor), the result is multiplied by
// 2^-e in the epilogue (exact).
//   * values within 2^(T+3) = 32x of the sampled maximum: |x - hi - lo| <= 2^-22 |x| (fp16 has 11 significant bits,
//     twice) - fp32-grade, 64x finer than a bf16 hi/lo split (2^-16) at the same three MFMAs and the same 12 conversion
//     instructions per 4 elements; measured on MI355X:
//     2-4e-7 of the output maximum on every layer shape, the level of an fp32 CPU convolution (tools/diag_precision.py);
//   * both roundings are to NEAREST on purpose: with hi rounded towards zero (v_cvt_pkrtz_f16_f32, which would saturate
//     instead of overflowing) lo always has the sign of x, the dropped lo*lo term always has the sign of the product,
//     and every product comes out ~1.2e-7 short - a bias that compounds through 25 layers (measured: logits 3e-6 off,
//     second-order bias gradients 3e-3 off); nearest rounding makes the dropped term sign-random;
//   * smaller values: lo becomes an fp16 subnormal (the MFMA does not flush them - measured), absolute error
//     <= 2^-25 / 2^T = 2^-27 of the sampled maximum;
//   * overflow needs a value 2^(13 - T) ~ 8 000 x above the sampled maximum of >= 4096 samples per block; such a value
//     saturates at +-65504 (MODE.FP16_OVFL, below: finite, wrong by its excess) and is COUNTED (cv_overflow_check):
//     rick_saturation_count() > 0 — tools/stability.py and the GPU tests fail on it.
#define CV_EXP_TARGET 2

__device__ __forceinline__ unsigned pack_f16_rne(float a, float b) {   // v_cvt_pk_f16_f32
    f16x2 r = __builtin_convertvector((f32x2_t){a, b}, f16x2);
    return *reinterpret_cast<unsigned *>(&r);
}

// amax (>= 0, finite) -> scale = 2^e and unscale = 2^-e with  amax * scale in [2^T, 2^(T+1));  amax == 0 -> 1, 1.
// (amax == 0 means every SAMPLED value of the block was zero — the first chunk whole plus 8 192 samples of the others in
// the igemm: a block of a masked / padded gradient map.  Unsampled non-zero values are then split as raw fp16: exact down
// to 6e-8, flushed below — an absolute error of at most 2^-25 on operands that the block's own sample calls zero; the split
// images (below) have no such case: their exponent comes from the exact maximum.)
template <int T>
__device__ __forceinline__ void cv_pow2_scale_t(float amax, float &scale, float &unscale) {
    int eb = (int)((__float_as_uint(amax) >> 23) & 0xffu);           // biased exponent (0 for zero / fp32 subnormals)
    if (amax == 0.f) eb = 127 + T;
    eb = eb < T + 1 ? T + 1 : eb;                                     // keeps both exponent fields in [1, 254]
    eb = eb > 253 ? 253 : eb;                                         // (inf / nan samples: finite scale, the nan propagates)
    scale = __uint_as_float((unsigned)(254 + T - eb) << 23);
    unscale = __uint_as_float((unsigned)(eb - T) << 23);
}
__device__ __forceinline__ void cv_pow2_scale(float amax, float &scale, float &unscale) {
    cv_pow2_scale_t<CV_EXP_TARGET>(amax, scale, unscale);
}

// ---- split images (split.hip): an activation / gradient tensor stored pre-split in HBM -------------------------------------
// A split image of an NHWC fp32 tensor [N, H, W, C] (C % 4 == 0) has the SAME size and addressing: the 16 bytes that hold
// channels c .. c+3 of a pixel in the fp32 tensor hold {hi x 4 | lo x 4} (8 x fp16) with hi = fp16(v * 2^e),
// lo = fp16(v * 2^e - hi) — exactly the two 8-byte pieces the MFMA kernels write into LDS when they split on the fly, so a
// consumer stages its usual 16-byte items with no VALU work, and any kernel that stores float4 can store an image instead.
// ONE exponent per tensor, kept in a 16-byte device header {2^e, 2^-e, bound, 0}.  The exponent is not sampled: the producer
// is handed a guaranteed BOUND on |v| (exact running maxima its own producers measured with atomic max, combined by the
// triangle inequality) and places it in [2^13, 2^14): no value can reach the fp16 maximum, and everything within 2^10 of the
// bound keeps both halves normal (2^-22 relative) — a wider window than the sampled per-block exponents above (2^5).
#define CV_SPLIT_TARGET 13
struct cv_split_hdr { float scale, unscale, bound, pad; };

// MODE.FP16_OVFL = 1: an fp32 -> fp16 conversion that overflows gives +-65504 instead of +-inf.  The operand exponent
// comes from a SAMPLE of the block's data (below); a value more than 2^(13 - T) = 8 188 times the largest sample would
// otherwise turn into inf and, through inf - inf in the `lo` part, into NaN in every output it touches.  With the bit set
// such a value saturates (finite, wrong by its excess) — called once at the start of every kernel that splits operands.
__device__ __forceinline__ void cv_fp16_saturate() {
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);       // hwreg(HW_REG_MODE, offset 23, size 1)
}

// a block-uniform float as an SGPR operand (the exponents come out of an LDS reduction, which the compiler treats as
// divergent: without this every scale multiply would hold VGPRs)
__device__ __forceinline__ float cv_uniform(float v) {
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

__device__ __forceinline__ float amax4(float m, const float4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// wave64 max, then across the block's waves through `red` (>= blockDim.x / 64 floats of LDS that nobody is reading or
// writing at the moment of the call); result in every thread.  ONE barrier.
__device__ __forceinline__ float block_amax(float v, float *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    const int nw = (int)blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
    for (int i = 1; i < nw; i++) m = fmaxf(m, red[i]);
    return m;
}

// (v * s) -> fp16 hi / lo pairs: 12 VALU per 4 elements, the scale included — 2 v_pk_mul_f32, 2 v_cvt_pk_f16_f32 (hi),
// 4 v_cvt_f32_f16, 2 v_pk_fma_f32 (x*s - hi, exact before its rounding), 2 v_cvt_pk_f16_f32 (lo) — all full-rate
// instructions: 41 issue cycles per split on one SIMD.  The shorter-looking form on v_fma_mixlo/mixhi_f16 (8 instructions:
// f16(v*s) and f16(v*s - hi) with the fp16 hi read back in place) was built first and is SLOWER: the mix instructions
// issue at half rate — 57 cycles per split (tools/micro/split_rate.hip on MI355X: 89 / 73 / 74 cycles per round for mix /
// this form / round 2's bf16 split, each round carrying 8 further VALU).  SPLIT == 1 (plain fp16): hi only.
// `s` per element (input-scale table x block exponent) ...
template <int SPLIT>
__device__ __forceinline__ void split4v(const float4 v, const float4 s, uint2 &hi, uint2 &lo) {
    const float4 w = make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w);
    const f16x2 h0 = __builtin_convertvector((f32x2_t){w.x, w.y}, f16x2);
    const f16x2 h1 = __builtin_convertvector((f32x2_t){w.z, w.w}, f16x2);
    hi.x = *reinterpret_cast<const unsigned *>(&h0);
    hi.y = *reinterpret_cast<const unsigned *>(&h1);
    if (SPLIT == 1) {
        lo.x = lo.y = 0;
        return;
    }
    lo.x = pack_f16_rne(w.x - (float)h0[0], w.y - (float)h0[1]);
    lo.y = pack_f16_rne(w.z - (float)h1[0], w.w - (float)h1[1]);
}
// ... or one block-uniform scale (the exponent alone) from an SGPR.
template <int SPLIT>
__device__ __forceinline__ void split4s(const float4 v, const float s, uint2 &hi, uint2 &lo) {
    split4v<SPLIT>(v, make_float4(s, s, s, s), hi, lo);
}
// Saturation tracking (2 v_max3 per 4 elements): `m` follows the largest |scaled operand| this thread has converted; a thread
// whose m reaches the fp16 maximum has produced a clamped (finite, wrong) value and reports it at the end of the kernel
// (cv_sat_report).  The exponent comes from a SAMPLE, so this is the only place the event can be seen.
__device__ __forceinline__ float cv_track4(float m, const float4 w) {
    return fmaxf(fmaxf(fmaxf(m, fabsf(w.x)), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w)));
}
template <int SPLIT>
__device__ __forceinline__ void split4v(const float4 v, const float4 s, uint2 &hi, uint2 &lo, float &m) {
    m = cv_track4(m, make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w));     // (the products are shared with the split below)
    split4v<SPLIT>(v, s, hi, lo);
}
template <int SPLIT>
__device__ __forceinline__ void split4s(const float4 v, const float s, uint2 &hi, uint2 &lo, float &m) {
    split4v<SPLIT>(v, make_float4(s, s, s, s), hi, lo, m);
}

// The same split in 8 instructions on v_fma_mixlo/mixhi_f16: f16(v*s) and f16(v*s - hi), the fp16 hi read back in place.
// Half-rate instructions (57 issue cycles against 41), but a third fewer of them: the weight-gradient kernel, whose single
// wave per SIMD places its staging instructions in the issue shadows of the MFMAs (each shadow takes ~2 instructions
// whatever they cost), gains from the count — 47.8 % MFMA-busy against 46.4 %, 2.09 -> 1.58 non-MFMA VALU per MFMA —
// while the two-wave kernels (igemm, convt2) do slightly better on the full-rate form above.
template <int SPLIT>
__device__ __forceinline__ void split4v_mix(const float4 v, const float4 s, uint2 &hi, uint2 &lo) {
    unsigned h0, h1, l0 = 0, l1 = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v.x), "v"(s.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v.y), "v"(s.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v.z), "v"(s.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v.w), "v"(s.w));
    if (SPLIT == 2) {
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "v"(s.x), "v"(h0));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "v"(s.y), "v"(h0));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "v"(s.z), "v"(h1));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "v"(s.w), "v"(h1));
    }
    hi = make_uint2(h0, h1);
    lo = make_uint2(l0, l1);
}
// (mix forms never materialise v * s in fp32: they track the RAW operand; the caller multiplies by a bound on the scale)
template <int SPLIT>
__device__ __forceinline__ void split4v_mix(const float4 v, const float4 s, uint2 &hi, uint2 &lo, float &mraw) {
    mraw = cv_track4(mraw, v);
    split4v_mix<SPLIT>(v, s, hi, lo);
}
template <int SPLIT>
__device__ __forceinline__ void split4s_mix(const float4 v, const float s, uint2 &hi, uint2 &lo);
template <int SPLIT>
__device__ __forceinline__ void split4s_mix(const float4 v, const float s, uint2 &hi, uint2 &lo, float &mraw) {
    mraw = cv_track4(mraw, v);
    split4s_mix<SPLIT>(v, s, hi, lo);
}
template <int SPLIT>
__device__ __forceinline__ void split4s_mix(const float4 v, const float s, uint2 &hi, uint2 &lo) {
    unsigned h0, h1, l0 = 0, l1 = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v.x), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v.y), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v.z), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v.w), "s"(s));
    if (SPLIT == 2) {
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "s"(s), "v"(h0));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "s"(s), "v"(h0));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "s"(s), "v"(h1));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "s"(s), "v"(h1));
    }
    hi = make_uint2(h0, h1);
    lo = make_uint2(l0, l1);
}

__device__ __forceinline__ void split1(float v, unsigned short &hi, unsigned short &lo, int split) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    hi = *reinterpret_cast<const unsigned short *>(&h);
    lo = split == 1 ? (unsigned short)0 : *reinterpret_cast<const unsigned short *>(&l);
}

// XCD-aware bijective remap of the linear block id: blocks that share an XCD (id % 8) get a
// contiguous range of logical tiles, so neighbouring position tiles of one co-tile share L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// 16-byte load of 4 consecutive channels.  VEC (channel count % 4 == 0): one float4 load from an address
// that is always safe (callers substitute the tensor base for out-of-range items) — no branch, so the
// compiler has no reason to wait vmcnt(0) per element.  !VEC: guarded scalar loads (odd channel counts).
template <bool VEC>
__device__ __forceinline__ float4 load4(const float *p, bool ok, int c, int C) {
    if (VEC) return *reinterpret_cast<const float4 *>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        v.x = p[0];
        if (c + 1 < C) v.y = p[1];
        if (c + 2 < C) v.z = p[2];
        if (c + 3 < C) v.w = p[3];
    }
    return v;
}

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {   // f(integral_constant<int, I>) ... f(<N-1>): indices usable as constants
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}


// ---- split-image producers (epilogues of the kernels that write activations / gradients) ---------------------------------
// header from the producer's bound  coef * (*b0 + *b1)  (b1 may be NULL); every thread evaluates it (block-uniform scalar
// work), ONE thread of the launch publishes it for the consumers
// A running maximum is kept in CV_AMAX_SLOTS words, one per 128-byte line (the launch's blocks are dealt over the slots):
// same-address atomics serialise in the L2 at ~70 ns each, which turned a launch of 16 K blocks folding into ONE word into a
// millisecond.  Readers take the maximum over the slots.
#define CV_AMAX_SLOTS 16
#define CV_AMAX_STRIDE 32
__device__ __forceinline__ float cv_amax_read(const float *w) {
    float m = w[0];
#pragma unroll
    for (int s = 1; s < CV_AMAX_SLOTS; s++) m = fmaxf(m, w[s * CV_AMAX_STRIDE]);
    return m;
}
__device__ __forceinline__ cv_split_hdr cv_split_header(const float *b0, const float *b1, float coef) {
    cv_split_hdr h;
    h.bound = coef * (cv_amax_read(b0) + (b1 ? cv_amax_read(b1) : 0.f));
    cv_pow2_scale_t<CV_SPLIT_TARGET>(h.bound, h.scale, h.unscale);
    h.pad = 0.f;
    return h;
}
// channels c .. c+3 (c % 4 == 0) of the pixel that starts at `pixel`: one 16-byte store at the float4's own address
__device__ __forceinline__ void cv_split_store4(unsigned char *pixel, int c, const float4 v, float s) {
    uint2 hi, lo;
    split4s<2>(v, s, hi, lo);
    *reinterpret_cast<uint4 *>(pixel + c * 4) = make_uint4(hi.x, hi.y, lo.x, lo.y);
}
// Saturation is impossible while the bound holds; a producer whose values exceed its bound (a caller's mistake, never silent)
// bumps this per-translation-unit counter, which rick_saturation_count() sums.
static __device__ unsigned g_cv_sat;
__device__ __forceinline__ void cv_sat_check(float thread_amax, float scale) {
    if (thread_amax * scale >= 65504.f) atomicAdd(&g_cv_sat, 1u);
}
// End of a kernel that splits on the fly: a thread whose tracked maximum reached the fp16 range's end bumps the counter.
// (TRAPSTS.EXCP would have been free, but this hardware does not accumulate IEEE exceptions while traps are disabled:
// tools/micro/trapsts.hip reads 0 after an fp32 -> fp16 and an fp32 overflow alike.)
__device__ __forceinline__ void cv_sat_report(float m) {
    if (m >= 65504.f) atomicAdd(&g_cv_sat, 1u);
}
// running maximum of a thread -> wave -> one atomic max on the float's bits (values are >= 0)
// The thread first reads the slot (device-scope load: from the L2, where the atomics act) and only issues the atomic when it
// would raise it.
// Called by every thread of the block: block maximum through `red` (>= blockDim.x / 64 floats of LDS; a barrier on either
// side, so it may alias buffers the block is done with), then ONE thread folds it into the block's slot.
__device__ __forceinline__ void cv_amax_publish(float m, float *word, float *cv_amax_red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) cv_amax_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = ((int)blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; i++) m = fmaxf(m, cv_amax_red[i]);
        const unsigned slot = (blockIdx.x + blockIdx.y * 5u + blockIdx.z * 11u) & (CV_AMAX_SLOTS - 1);
        unsigned *w = reinterpret_cast<unsigned *>(word) + slot * CV_AMAX_STRIDE;
        const unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, bits);
    }
}
// Sum of the nsplit partial float4 at p[sp * stride], sp = 0 .. nsplit-1, added IN SPLIT ORDER (the second-stage kernels'
// fixed summation order: results do not depend on the unrolling) with up to eight independent 16-byte loads in flight — the
// plain `for (sp) s += p[sp * stride]` loop is a chain of dependent round trips (runtime trip count: hipcc waits for each load
// before the next), which is what the 6 - 13 us of these launches were made of (round 6).
__device__ __forceinline__ float4 cv_sum_splits4(const float4 *__restrict__ p, int64_t stride, int nsplit) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(int64_t)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    if (sp + 4 <= nsplit) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = p[(int64_t)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        sp += 4;
    }
    if (sp + 2 <= nsplit) {
        const float4 v0 = p[(int64_t)sp * stride], v1 = p[(int64_t)(sp + 1) * stride];
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
        sp += 2;
    }
    if (sp < nsplit) {
        const float4 v0 = p[(int64_t)sp * stride];
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
    }
    return s;
}

// ---- in-launch split-K fix-up: arrival ticket (split.hip: rick_internal_tickets) -----------------------------------------------
// Called by every thread of a block AFTER it has stored its partial tile with plain stores.  Returns true in every thread of
// the block that arrives LAST for this tile; that block may then read all partial tiles with plain loads.  The hand-off is the
// placement-independent counter form of the guide (cdna_hip_programming.md, Projection GEMM item 2): every storing wave drains
// its stores, the block's barrier, ONE agent-scope release by lane 0 (+ an explicit vmcnt(0): ROCm 7.2 can drop the fence's own),
// a relaxed agent-scope add; the last arriver: one agent-scope acquire + vmcnt(0), then the barrier that lets the other waves
// load.  The counter goes back to 0 for the next launch that is handed this range (stream order makes that store visible).
// `flag`: 4 bytes of LDS nobody else uses until the block ends.
extern "C" unsigned *rick_internal_tickets(int n, void *stream);
extern "C" int rick_internal_tune(int key);          // current value of a rick_conv_tuning key (split.hip)
// WT: the partial tile was stored WRITE-THROUGH (sc1 stores: every byte of it) — the bytes are in memory once the storing
// wave's vmcnt has drained, and no release fence (an L2 write-back per block: measured +13 ... +43 us on a 64 ... 512-block
// launch, profiles/r06_splitk_fused.txt) is needed; the acquire of the last arriver stays.
template <bool WT = false>
__device__ __forceinline__ bool cv_splitk_arrive(unsigned *counter, int nsplit, unsigned *flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == (unsigned)(nsplit - 1) ? 1u : 0u;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
    }
    __syncthreads();
    return *flag != 0u;
}

#define CV_DEFINE_SAT_ACCESSOR(fn)                                                                    \
    extern "C" int fn(unsigned *count, int reset) {                                                   \
        unsigned *p = nullptr, v = 0, z = 0;                                                          \
        if (hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_cv_sat)) != hipSuccess) return 1;           \
        if (hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;                       \
        if (reset && hipMemcpy(p, &z, 4, hipMemcpyHostToDevice) != hipSuccess) return 1;              \
        *count = v;                                                                                   \
        return 0;                                                                                     \
    }
