// The two drop-in extensions in the reference's other dtypes (op/upfirdn2d_kernel.cu:311-367 and op/fused_bias_act_kernel.cu:79
// dispatch half / float / double): generic, correctness-first kernels — the training script never leaves fp32 (the fp32 entries
// rick_upfirdn2d_f32 / rick_bias_act_f32 are the product path), but a drop-in user's `.double()` gradcheck or `.half()` inference
// must not hit a dtype error.  Same index math as the fp32 kernels (and oracle/csrc/oracle_ops.c): tap order y-outer / x-inner,
// one fused multiply-add per tap; fp16 accumulates in fp32 and rounds once.
#include "common.h"
#include <hip/hip_fp16.h>

template <class T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

struct UfdG {
    int in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

// planar layout [major, H, W]; one thread per output element
template <class T>
__global__ __launch_bounds__(256) void upfirdn2d_any_kernel(const T *__restrict__ in, const T *__restrict__ kern, T *__restrict__ out,
                                                            int64_t total, UfdG p) {
    typedef typename Acc<T>::type A;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % p.out_w);
        const int64_t r = i / p.out_w;
        const int oy = (int)(r % p.out_h);
        const int64_t m = r / p.out_h;
        const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0, in_y = floor_div_i(mid_y, p.up_y);
        const int h = floor_div_i(mid_y + p.kh, p.up_y) - in_y, ky0 = mid_y + p.kh - (in_y + 1) * p.up_y;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0, in_x = floor_div_i(mid_x, p.up_x);
        const int w = floor_div_i(mid_x + p.kw, p.up_x) - in_x, kx0 = mid_x + p.kw - (in_x + 1) * p.up_x;
        const T *src = in + m * (int64_t)p.in_h * p.in_w;
        A v = 0;
        for (int y = 0; y < h; y++) {
            const int iy = in_y + y;
            for (int x = 0; x < w; x++) {
                const int ix = in_x + x;
                if (iy < 0 || iy >= p.in_h || ix < 0 || ix >= p.in_w) continue;
                v = fma((A)src[(int64_t)iy * p.in_w + ix], (A)kern[(ky0 - y * p.up_y) * p.kw + kx0 - x * p.up_x], v);
            }
        }
        out[i] = (T)v;
    }
}

// dtype: 1 = float64, 2 = float16 (0 = float32 is served by rick_upfirdn2d_f32).  Planar [major, H, W] only.
extern "C" int rick_upfirdn2d_any(const void *input, const void *kernel, void *out, int dtype, int64_t major, int in_h, int in_w,
                                  int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                  int pad_y1, void *stream) {
    if (!input || !kernel || !out || major <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || up_x <= 0 || up_y <= 0 ||
        down_x <= 0 || down_y <= 0 || (dtype != 1 && dtype != 2))
        return RICK_EINVAL;
    UfdG p = {in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, 0, 0};
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (p.out_h <= 0 || p.out_w <= 0) return RICK_EINVAL;
    const int64_t total = major * p.out_h * p.out_w;
    int64_t nb = cdiv64(total, 256);
    if (nb > 65535) nb = 65535;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 1)
        hipLaunchKernelGGL(upfirdn2d_any_kernel<double>, dim3((unsigned)nb), dim3(256), 0, st, (const double *)input,
                           (const double *)kernel, (double *)out, total, p);
    else
        hipLaunchKernelGGL(upfirdn2d_any_kernel<__half>, dim3((unsigned)nb), dim3(256), 0, st, (const __half *)input,
                           (const __half *)kernel, (__half *)out, total, p);
    RICK_LAUNCH_STATUS();
}

// fused_bias_act (op/fused_bias_act_kernel.cu:18-49): v = x[i] + bias[(i / step_b) % size_b]; act*10+grad as in rick_bias_act_f32
template <class T>
__global__ __launch_bounds__(256) void bias_act_any_kernel(const T *__restrict__ x, const T *__restrict__ b, const T *__restrict__ ref,
                                                           T *__restrict__ out, int64_t n, int64_t step_b, int64_t size_b, int mode,
                                                           float alpha, float scale) {
    typedef typename Acc<T>::type A;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        A v = (A)x[i];
        if (b) v += (A)b[(i / step_b) % size_b];
        const A r = ref ? (A)ref[i] : (A)0;
        A y;
        switch (mode) {
            case 30: y = v > 0 ? v : v * (A)alpha; break;
            case 31: y = r > 0 ? v : v * (A)alpha; break;
            case 12:
            case 32: y = 0; break;
            default: y = v;
        }
        out[i] = (T)(y * (A)scale);
    }
}

extern "C" int rick_bias_act_any(const void *x, const void *bias, const void *ref, void *out, int dtype, int64_t n, int64_t step_b,
                                 int64_t size_b, int act, int grad, float alpha, float scale, void *stream) {
    if (!x || !out || n < 0 || (bias && (step_b <= 0 || size_b <= 0)) || (dtype != 1 && dtype != 2)) return RICK_EINVAL;
    if (n == 0) return 0;
    int64_t nb = cdiv64(n, 256);
    if (nb > 65535) nb = 65535;
    hipStream_t st = (hipStream_t)stream;
    const int mode = act * 10 + grad;
    if (dtype == 1)
        hipLaunchKernelGGL(bias_act_any_kernel<double>, dim3((unsigned)nb), dim3(256), 0, st, (const double *)x, (const double *)bias,
                           (const double *)ref, (double *)out, n, step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1, mode, alpha, scale);
    else
        hipLaunchKernelGGL(bias_act_any_kernel<__half>, dim3((unsigned)nb), dim3(256), 0, st, (const __half *)x, (const __half *)bias,
                           (const __half *)ref, (__half *)out, n, step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1, mode, alpha, scale);
    RICK_LAUNCH_STATUS();
}
