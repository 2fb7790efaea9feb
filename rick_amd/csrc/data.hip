// Data path of the RICK loop on the device: the few-shot training set (10 images) and the test set stay
// resident in HBM as uint8 [N, H, W, 3]; a batch is ONE launch that gathers the sampled images, applies the horizontal
// flips and the ToTensor + Normalize(0.5, 0.5) arithmetic of the reference transform (train_dynamic_update_prune.py:
// 789-798) and writes the [B, 3, H, W] fp32 batch the discriminator consumes — no per-iteration host->device pixel copy.
// Host side: PNG scanline reconstruction (the PNG 'filter' step, RFC 2083 §6) for the decoder in rick_amd/data.py —
// the inflate itself is zlib's.
#include "common.h"

__global__ __launch_bounds__(256) void image_batch_kernel(const uint8_t *__restrict__ src, const int64_t *__restrict__ idx,
                                                          const uint8_t *__restrict__ flip, float *__restrict__ out, int H, int W,
                                                          int B) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)B * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / hw);
        const int64_t p = i - (int64_t)b * hw;
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        const int xs = flip[b] ? W - 1 - x : x;
        const uint8_t *s = src + ((idx[b] * H + y) * (int64_t)W + xs) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // ToTensor: uint8 -> float / 255; Normalize: (v - 0.5) / 0.5 — the same IEEE operations in the same order
            const float v = (float)s[c] / 255.0f;
            out[((int64_t)b * 3 + c) * hw + p] = (v - 0.5f) / 0.5f;
        }
    }
}

extern "C" int rick_image_batch_f32(const uint8_t *images, const int64_t *index, const uint8_t *flip, float *out, int N, int H, int W,
                                    int B, void *stream) {
    if (!images || !index || !flip || !out || N < 1 || H < 1 || W < 1 || B < 1) return RICK_EINVAL;
    int64_t nb = cdiv64((int64_t)B * H * W, 256);
    if (nb > 65535) nb = 65535;
    hipLaunchKernelGGL(image_batch_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, images, index, flip, out, H, W, B);
    RICK_LAUNCH_STATUS();
}

// PNG reconstruction in place: `data` holds H scanlines of (1 filter byte + stride bytes); on return bytes 1..stride of
// each scanline are the reconstructed samples.  bpp = bytes per complete pixel (>= 1).  Returns RICK_EINVAL for an unknown
// filter type.  (Host code; no GPU involved.)
extern "C" int rick_png_unfilter(uint8_t *data, int H, int stride, int bpp) {
    if (!data || H < 1 || stride < 1 || bpp < 1) return RICK_EINVAL;
    const int64_t line = (int64_t)stride + 1;
    for (int y = 0; y < H; y++) {
        uint8_t *cur = data + y * line + 1;
        const uint8_t *up = y ? data + (y - 1) * line + 1 : nullptr;
        switch (data[y * line]) {
        case 0: break;
        case 1:
            for (int i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            if (up) for (int i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + up[i]);
            break;
        case 3:
            for (int i = 0; i < stride; i++) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (int i = 0; i < stride; i++) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                cur[i] = (uint8_t)(cur[i] + (pa <= pb && pa <= pc ? a : pb <= pc ? b : c));
            }
            break;
        default: return RICK_EINVAL;
        }
    }
    return 0;
}
