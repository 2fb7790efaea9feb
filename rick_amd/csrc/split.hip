// Split images: fp32 NHWC tensors stored as fp16 hi/lo chunk pairs with one power-of-two exponent per tensor
// (conv_common.h).  Producers of activations / gradients write them next to (or instead of) the fp32 tensor; the MFMA
// kernels stage them with plain 16-byte copies.  This file: the exact running maximum (atomic max on the float's bits),
// the stand-alone fp32 -> split-image pass (layers whose producer is not fused yet, tests) and its inverse (tests).
#include "conv_common.h"

// |x| maximum of n floats, atomically folded into *word (non-negative floats order like their bit patterns).
__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ word) {
    float m = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        m = amax4(m, reinterpret_cast<const float4 *>(x)[i]);
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
    __shared__ float red[4];
    cv_amax_publish(m, word, red);
}

extern "C" int rick_amax_f32(const float *x, int64_t n, float *amax_word, void *stream) {
    if (!x || !amax_word || n < 0 || ((uintptr_t)x % 16)) return RICK_EINVAL;
    if (n == 0) return 0;
    int64_t nb = cdiv64(n >> 2, 256 * 8);
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, n, amax_word);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_stream_capture_id(void *stream, unsigned long long *id) {
    if (!id) return RICK_EINVAL;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long v = 0;
    const hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &st, &v);
    if (e != hipSuccess) return 1000 + (int)e;
    *id = (st == hipStreamCaptureStatusActive) ? (v ? v : ~0ull) : 0ull;
    return 0;
}

__global__ __launch_bounds__(256) void split_pack_kernel(const float *__restrict__ x, unsigned char *__restrict__ out,
                                                         cv_split_hdr *__restrict__ hdr, const float *__restrict__ a0,
                                                         const float *__restrict__ a1, float coef, int64_t n4) {
    const cv_split_hdr h = cv_split_header(a0, a1, coef);
    if (blockIdx.x == 0 && threadIdx.x == 0) *hdr = h;
    const float s = cv_uniform(h.scale);
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        cv_split_store4(out + i * 16, 0, v, s);
        m = amax4(m, v);
    }
    // a value above the bound means the caller's bound was no bound: counted, never silent (rick_saturation_count)
    cv_sat_check(m, s);
}

__global__ __launch_bounds__(256) void split_unpack_kernel(const unsigned char *__restrict__ pk, const cv_split_hdr *__restrict__ hdr,
                                                           float *__restrict__ out, int64_t n4) {
    const float u = hdr->unscale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f16x8 q = *reinterpret_cast<const f16x8 *>(pk + i * 16);
        reinterpret_cast<float4 *>(out)[i] = make_float4(((float)q[0] + (float)q[4]) * u, ((float)q[1] + (float)q[5]) * u,
                                                          ((float)q[2] + (float)q[6]) * u, ((float)q[3] + (float)q[7]) * u);
    }
}

// ---- kernel-form switches (include/rick_hip.h: rick_conv_tuning) -------------------------------------------------------------------
static int g_tune[5] = {0, 192, 0, 0, 2};
extern "C" int rick_conv_tuning(int key, int value) {
    if (key < 0 || key >= (int)(sizeof(g_tune) / sizeof(g_tune[0]))) return -1;
    const int prev = g_tune[key];
    g_tune[key] = value;
    return prev;
}
extern "C" int rick_internal_tune(int key) { return g_tune[key]; }

// ---- arrival tickets of the in-launch split-K fix-up (conv_common.h: cv_splitk_arrive) ------------------------------------------
// One zero-initialised, SELF-RESETTING 32-bit counter per output tile of a split-K launch: the block whose add returns
// nsplit - 1 is the last to arrive, sums the partial tiles in split order and puts the counter back to 0.  The counters live in
// one device buffer per GPU, zeroed once when it is created; a launch takes a range that no other launch in flight can hold:
//   * launches issued into a CAPTURING stream keep their range for the life of the process (graph replays re-use it; replays of
//     one node are ordered) — a bump allocator over the lower half that never wraps;
//   * eager launches take theirs from a ring over the upper half (2^19 counters: more than 8 000 launches in flight).
// NULL (no buffer yet while a capture is running, range exhausted, any runtime error): the caller runs its second-stage kernel.
#define TK_TOTAL (1u << 20)
#define TK_MAXDEV 16
static unsigned *g_tk_buf[TK_MAXDEV];
static unsigned g_tk_perm[TK_MAXDEV], g_tk_ring[TK_MAXDEV];
extern "C" unsigned *rick_internal_tickets(int n, void *stream) {
    int dev = 0;
    if (!g_tune[RICK_TUNE_SPLITK_FUSED] || n < 1 || n > (int)(TK_TOTAL / 4) || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TK_MAXDEV) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    const bool capturing = cs != hipStreamCaptureStatusNone;
    if (!g_tk_buf[dev]) {
        if (capturing) return nullptr;          // (allocation is not a capturable operation: the first eager launch creates the buffer)
        unsigned *p = nullptr;
        if (hipMalloc((void **)&p, TK_TOTAL * sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, TK_TOTAL * sizeof(unsigned)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_tk_buf[dev] = p;
    }
    const unsigned n32 = ((unsigned)n + 31u) & ~31u;      // whole 128-byte lines per launch
    if (capturing) {
        if (g_tk_perm[dev] + n32 > TK_TOTAL / 2) return nullptr;
        unsigned *r = g_tk_buf[dev] + g_tk_perm[dev];
        g_tk_perm[dev] += n32;
        return r;
    }
    if (g_tk_ring[dev] + n32 > TK_TOTAL / 2) g_tk_ring[dev] = 0;
    unsigned *r = g_tk_buf[dev] + TK_TOTAL / 2 + g_tk_ring[dev];
    g_tk_ring[dev] += n32;
    return r;
}

CV_DEFINE_SAT_ACCESSOR(rick_sat_split)
extern "C" int rick_sat_upfirdn2d(unsigned *, int);
extern "C" int rick_sat_elementwise(unsigned *, int);
extern "C" int rick_sat_conv(unsigned *, int);
extern "C" int rick_sat_wgrad(unsigned *, int);
extern "C" int rick_sat_convt2(unsigned *, int);
extern "C" int rick_sat_thin(unsigned *, int);

extern "C" int rick_saturation_count(unsigned *count, int reset) {
    if (!count) return RICK_EINVAL;
    unsigned a = 0, b = 0, c = 0, d = 0, e = 0, f = 0, g = 0;
    if (rick_sat_split(&a, reset) || rick_sat_upfirdn2d(&b, reset) || rick_sat_elementwise(&c, reset) || rick_sat_conv(&d, reset) ||
        rick_sat_wgrad(&e, reset) || rick_sat_convt2(&f, reset) || rick_sat_thin(&g, reset))
        return 1;
    *count = a + b + c + d + e + f + g;
    return 0;
}

extern "C" int rick_split_pack_f32(const float *x, void *out, float *hdr, const float *amax0, const float *amax1, float coef,
                                   int64_t npix, int C, void *stream) {
    if (!x || !out || !hdr || !amax0 || npix < 0 || C <= 0 || (C & 3) || !(coef > 0.f)) return RICK_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)hdr) % 16) return RICK_EINVAL;
    if (npix == 0) return 0;
    const int64_t n4 = npix * (C / 4);
    int64_t nb = cdiv64(n4, 256 * 8);
    nb = nb < 1 ? 1 : (nb > 8192 ? 8192 : nb);
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (unsigned char *)out,
                       (cv_split_hdr *)hdr, amax0, amax1, coef, n4);
    RICK_LAUNCH_STATUS();
}

extern "C" int rick_split_unpack_f32(const void *pk, const float *hdr, float *out, int64_t npix, int C, void *stream) {
    if (!pk || !hdr || !out || npix < 0 || C <= 0 || (C & 3)) return RICK_EINVAL;
    if (((uintptr_t)pk | (uintptr_t)out | (uintptr_t)hdr) % 16) return RICK_EINVAL;
    if (npix == 0) return 0;
    const int64_t n4 = npix * (C / 4);
    int64_t nb = cdiv64(n4, 256 * 8);
    nb = nb < 1 ? 1 : (nb > 8192 ? 8192 : nb);
    hipLaunchKernelGGL(split_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)pk,
                       (const cv_split_hdr *)hdr, out, n4);
    RICK_LAUNCH_STATUS();
}

// out = max|mul| * gain * (c0 * amax(w0) + max|nw| * amax(w_noise) + max|bias|)   (rick_hip.h: rick_bound_tail_f32)
__global__ __launch_bounds__(256) void bound_tail_kernel(float *__restrict__ out, const float *__restrict__ w0, float c0,
                                                         const float *__restrict__ nw, const float *__restrict__ w_noise,
                                                         const float *__restrict__ bias, int nbias, float gain,
                                                         const float *__restrict__ mul, int nmul) {
    __shared__ float red[8];
    float mb = 0.f, mm = 0.f;
    for (int i = threadIdx.x; i < nbias; i += 256) mb = fmaxf(mb, fabsf(bias[i]));
    for (int i = threadIdx.x; i < nmul; i += 256) mm = fmaxf(mm, fabsf(mul[i]));
    mb = block_amax(mb, red);
    mm = block_amax(mm, red + 4);
    for (int i = threadIdx.x; i < CV_AMAX_SLOTS * CV_AMAX_STRIDE; i += 256) out[i] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = c0 * cv_amax_read(w0) + mb;
        if (nw && w_noise) v += fabsf(nw[0]) * cv_amax_read(w_noise);
        out[0] = (mul ? mm : 1.f) * gain * v;
    }
}

extern "C" int rick_bound_tail_f32(float *out_word, const float *w0, float c0, const float *nw, const float *w_noise,
                                   const float *bias, int nbias, float gain, const float *mul, int nmul, void *stream) {
    if (!out_word || !w0 || !(c0 > 0.f) || !(gain > 0.f) || nbias < 0 || nmul < 0) return RICK_EINVAL;
    hipLaunchKernelGGL(bound_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out_word, w0, c0, nw, w_noise,
                       bias, bias ? nbias : 0, gain, mul, mul ? nmul : 0);
    RICK_LAUNCH_STATUS();
}
