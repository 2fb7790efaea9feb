// Run-to-run reproducibility of the modulation-bank forward kernel when TWO processes share the GPU (tests/test_gpu_dp.py runs two
// ranks on cuda:0): variants of the kernel are launched ITERS times on the same inputs and every output is compared with the
// variant's first output on the device.  Start two copies at once:  ./modbank_race & ./modbank_race & wait
//   V0: product form of round 4 (8 weight rows per wave fetched together; weight pointer read from the descriptor -> FLAT loads)
//   V1: round-3 form (one row at a time)
//   V2: V0 with the weight / bias pointers taken from ONE kernel-argument base + offsets (-> GLOBAL loads)
//   V3: V0 with the cross-lane sums done by DPP row operations + readlane instead of ds_bpermute
//   V4: V0 with every lane storing nothing but lane 0 computing the bias add BEFORE the reduction tail (no EXEC change while a
//       ds_bpermute is in flight)
// build: hipcc -O3 --offload-arch=gfx950 modbank_race.hip -o modbank_race
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <math.h>
#include <vector>

#define MB_MAXB 8
#define MB_ROWS 32
struct desc {
    const float *w, *b;
    int64_t io_off, w_off;
    int C, lat_idx, blk_begin, reserved;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// DPP / readlane reduction: row_shr within rows of 16, then the four row totals through readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
    // lane 15 of every row holds the row total
    const int iv = __builtin_bit_cast(int, v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 15));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 31));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 47));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 63));
    return (a + b) + (c + d);
}

template <int V>
__global__ __launch_bounds__(256) void modbank_fwd(const float *__restrict__ lat, int B, int n_latent, int K,
                                                   const desc *__restrict__ descs, int n, float scale, float *__restrict__ out,
                                                   const float *__restrict__ wbase, const float *__restrict__ bbase) {
    extern __shared__ float sl[];
    int d = 0;
    for (int i = 1; i < n; i++)
        if ((int)blockIdx.x >= descs[i].blk_begin) d = i;
    const desc ds = descs[d];
    for (int j = threadIdx.x; j < B * K; j += 256) {
        const int b = j / K, k = j - b * K;
        sl[j] = lat[((int64_t)b * n_latent + ds.lat_idx) * K + k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = ((int)blockIdx.x - ds.blk_begin) * MB_ROWS + wave * (MB_ROWS / 4);
    const float *W = V == 2 ? wbase + ds.w_off : ds.w;
    const float *Bv = V == 2 ? bbase + (ds.w_off / K) : ds.b;
    if (V == 1) {
        for (int r = 0; r < MB_ROWS / 4; r++) {
            const int c = c0 + r;
            if (c >= ds.C) break;
            const float *wr = W + (int64_t)c * K;
            float acc[MB_MAXB];
#pragma unroll
            for (int b = 0; b < MB_MAXB; b++) acc[b] = 0.f;
            for (int k = lane * 4; k < K; k += 256) {
                const float4 wv = *reinterpret_cast<const float4 *>(wr + k);
#pragma unroll
                for (int b = 0; b < MB_MAXB; b++)
                    if (b < B) {
                        const float4 lv = *reinterpret_cast<const float4 *>(sl + b * K + k);
                        acc[b] = __builtin_fmaf(wv.x, lv.x, acc[b]);
                        acc[b] = __builtin_fmaf(wv.y, lv.y, acc[b]);
                        acc[b] = __builtin_fmaf(wv.z, lv.z, acc[b]);
                        acc[b] = __builtin_fmaf(wv.w, lv.w, acc[b]);
                    }
            }
#pragma unroll
            for (int b = 0; b < MB_MAXB; b++)
                if (b < B) {
                    const float v = wave_sum(acc[b]);
                    if (lane == 0) out[ds.io_off + (int64_t)b * ds.C + c] = v * scale + (Bv ? Bv[c] : 0.f);
                }
        }
        return;
    }
    constexpr int RW = MB_ROWS / 4;
    float acc[RW][MB_MAXB];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++) acc[r][b] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        float4 wv[RW];
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const int c = c0 + r < ds.C ? c0 + r : ds.C - 1;
            wv[r] = *reinterpret_cast<const float4 *>(W + (int64_t)c * K + k);
        }
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++)
            if (b < B) {
                const float4 lv = *reinterpret_cast<const float4 *>(sl + b * K + k);
#pragma unroll
                for (int r = 0; r < RW; r++) {
                    acc[r][b] = __builtin_fmaf(wv[r].x, lv.x, acc[r][b]);
                    acc[r][b] = __builtin_fmaf(wv[r].y, lv.y, acc[r][b]);
                    acc[r][b] = __builtin_fmaf(wv[r].z, lv.z, acc[r][b]);
                    acc[r][b] = __builtin_fmaf(wv[r].w, lv.w, acc[r][b]);
                }
            }
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int c = c0 + r;
        if (c >= ds.C) break;
        float bias = 0.f;
        if (V == 4) bias = Bv ? Bv[c] : 0.f;          // every lane loads it, before the reduction
#pragma unroll
        for (int b = 0; b < MB_MAXB; b++)
            if (b < B) {
                const float v = V == 3 ? wave_sum_dpp(acc[r][b]) : wave_sum(acc[r][b]);
                if (V == 4) {
                    const float o = v * scale + bias;
                    if (lane == 0) out[ds.io_off + (int64_t)b * ds.C + c] = o;
                } else if (lane == 0) {
                    out[ds.io_off + (int64_t)b * ds.C + c] = v * scale + (Bv ? Bv[c] : 0.f);
                }
            }
    }
}

__global__ void compare(const float *a, const float *ref, int n, unsigned *bad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && __float_as_uint(a[i]) != __float_as_uint(ref[i])) atomicAdd(bad, 1u);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int L = 11, C = 512, K = 512, B = 2, NL = 8;
    std::vector<float> hw((size_t)L * C * K), hb((size_t)L * C), hl((size_t)B * NL * K);
    srand(7);
    for (auto &v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto &v : hb) v = 1.f;
    for (auto &v : hl) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    float *w, *b, *lat, *out, *ref;
    desc *dd;
    unsigned *bad;
    CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&lat, hl.size() * 4));
    CK(hipMalloc(&out, (size_t)L * B * C * 4)); CK(hipMalloc(&ref, (size_t)L * B * C * 4)); CK(hipMalloc(&dd, L * sizeof(desc)));
    CK(hipMalloc(&bad, 4 * 8));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lat, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    std::vector<desc> hd(L);
    for (int l = 0; l < L; l++) hd[l] = {w + (size_t)l * C * K, b + (size_t)l * C, (int64_t)l * B * C, (int64_t)l * C * K, C, l % NL, l * (C / MB_ROWS), 0};
    CK(hipMemcpy(dd, hd.data(), L * sizeof(desc), hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, 32));
    const int blocks = L * (C / MB_ROWS), n = L * B * C;
    const float scale = 1.f / sqrtf((float)K);
    for (int v = 0; v < 5; v++) {
        unsigned events = 0;
        for (int it = 0; it <= iters; it++) {
            float *dst = it == 0 ? ref : out;
            switch (v) {
            case 0: hipLaunchKernelGGL(modbank_fwd<0>, dim3(blocks), dim3(256), B * K * 4, 0, lat, B, NL, K, dd, L, scale, dst, w, b); break;
            case 1: hipLaunchKernelGGL(modbank_fwd<1>, dim3(blocks), dim3(256), B * K * 4, 0, lat, B, NL, K, dd, L, scale, dst, w, b); break;
            case 2: hipLaunchKernelGGL(modbank_fwd<2>, dim3(blocks), dim3(256), B * K * 4, 0, lat, B, NL, K, dd, L, scale, dst, w, b); break;
            case 3: hipLaunchKernelGGL(modbank_fwd<3>, dim3(blocks), dim3(256), B * K * 4, 0, lat, B, NL, K, dd, L, scale, dst, w, b); break;
            case 4: hipLaunchKernelGGL(modbank_fwd<4>, dim3(blocks), dim3(256), B * K * 4, 0, lat, B, NL, K, dd, L, scale, dst, w, b); break;
            }
            if (it) hipLaunchKernelGGL(compare, dim3((n + 255) / 256), dim3(256), 0, 0, out, ref, n, bad + v);
            (void)events;
        }
        CK(hipDeviceSynchronize());
        unsigned hbad = 0;
        CK(hipMemcpy(&hbad, bad + v, 4, hipMemcpyDeviceToHost));
        printf("[pid %d] V%d: %u differing elements over %d launches of %d outputs\n", (int)getpid(), v, hbad, iters, n);
        fflush(stdout);
    }
    return 0;
}
