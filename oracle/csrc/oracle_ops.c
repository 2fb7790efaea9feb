/* Oracle (test infrastructure, NOT product code): scalar C restatement of the
 * reference's two custom kernels, used to check the HIP kernels bit-for-bit.
 *
 *   oracle_upfirdn2d_f32  follows op/upfirdn2d_kernel.cu:49-105 (generic kernel) and the
 *                         output-size formula at :237-240; the floor division is the
 *                         reference's floor_div (:17-25).
 *   oracle_bias_act_f32   follows op/fused_bias_act_kernel.cu:18-49.
 *
 * Accumulation order is ky-outer / kx-inner with one fused multiply-add per tap
 * (fmaf), which is what nvcc emits for `v += x * k` and what the HIP kernel uses,
 * so results are comparable bit-for-bit.
 */
#include <math.h>
#include <stdint.h>

static int floor_div(int a, int b) {
    int c = a / b;
    if (c * b > a) c--;
    return c;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* input  [major, in_h, in_w]  (minor dim = 1, as every call site uses it)
 * kernel [kh, kw]; out [major, out_h, out_w] */
int oracle_upfirdn2d_f32(const float *input, const float *kernel, float *out,
                         int64_t major, int in_h, int in_w, int kh, int kw,
                         int up_x, int up_y, int down_x, int down_y,
                         int pad_x0, int pad_x1, int pad_y0, int pad_y1) {
    int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;   /* kernel.cu:237-240 */
    int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (out_h <= 0 || out_w <= 0) return 1;
    for (int64_t m = 0; m < major; m++) {
        for (int oy = 0; oy < out_h; oy++) {
            int mid_y = oy * down_y + up_y - 1 - pad_y0;
            int in_y = imin(imax(floor_div(mid_y, up_y), 0), in_h);
            int h = imin(imax(floor_div(mid_y + kh, up_y), 0), in_h) - in_y;
            int kernel_y = mid_y + kh - (in_y + 1) * up_y;
            for (int ox = 0; ox < out_w; ox++) {
                int mid_x = ox * down_x + up_x - 1 - pad_x0;
                int in_x = imin(imax(floor_div(mid_x, up_x), 0), in_w);
                int w = imin(imax(floor_div(mid_x + kw, up_x), 0), in_w) - in_x;
                int kernel_x = mid_x + kw - (in_x + 1) * up_x;
                float v = 0.0f;
                for (int y = 0; y < h; y++) {
                    for (int x = 0; x < w; x++) {
                        float xv = input[(m * in_h + in_y + y) * in_w + in_x + x];
                        float kv = kernel[(kernel_y - y * up_y) * kw + (kernel_x - x * up_x)];
                        v = fmaf(xv, kv, v);
                    }
                }
                out[(m * out_h + oy) * out_w + ox] = v;
            }
        }
    }
    return 0;
}

/* x[n] flattened, bias index = (i / step_b) % size_b; empty bias/ref = NULL. */
int oracle_bias_act_f32(const float *x, const float *b, const float *ref, float *out,
                        int64_t n, int64_t step_b, int64_t size_b,
                        int act, int grad, float alpha, float scale) {
    for (int64_t i = 0; i < n; i++) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        float r = ref ? ref[i] : 0.0f;
        float y;
        switch (act * 10 + grad) {
            default:
            case 10: y = v; break;
            case 11: y = v; break;
            case 12: y = 0.0f; break;
            case 30: y = (v > 0.0f) ? v : v * alpha; break;
            case 31: y = (r > 0.0f) ? v : v * alpha; break;
            case 32: y = 0.0f; break;
        }
        out[i] = y * scale;
    }
    return 0;
}
