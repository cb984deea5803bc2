// Pooling kernels (HBM-bound, 16-byte vector access, 8 channels per thread):
//   MaxPool2d(3,2,1)                         model/resnet.py:114
//   AdaptiveAvgPool2d(1)+AdaptiveMaxPool2d(1) model/resnet.py:122-123,130-131,266,274
#include "cs_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ amax, int N, int H, int W, int C, int P,
                                                          int Q) {
    const int CG = C / 8;
    const long long total = (long long)N * P * Q * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        long long t = idx / CG;
        const int q = (int)(t % Q); t /= Q;
        const int p = (int)(t % P);
        const long long n = t / P;
        float best[8];
        int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = p * 2 - 1 + kh;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = q * 2 - 1 + kw;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                load8<T>(x + ((n * H + iy) * (long long)W + ix) * C + cg * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // ATen rule: take if strictly greater (or NaN); the first valid tap seeds the max
                    if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
                }
                first = false;
            }
        }
        const long long o = ((n * P + p) * (long long)Q + q) * C + cg * 8;
        store8<T>(y + o, best);
        if (amax) {
            uint2 pk;
            pk.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
            pk.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
            *reinterpret_cast<uint2*>(amax + o) = pk;
        }
    }
}

// Gather form of the backward: each input pixel takes dy of the <= 4 windows that contain it where the window's recorded argmax is
// this pixel.  No atomics, no zero-fill pass.  (Round 1-2: one input pixel per thread, 1 + 2 + 2 + 4 = 9 window loads per 2 x 2 pixels.)
// Per 2 x 2 block of input pixels (round 3): the block (2a + i, 2b + j) is covered by exactly the four windows
// (a, b), (a, b+1), (a+1, b), (a+1, b+1); they are loaded ONCE (argmax bytes, dy, ReLU mask) and matched against the nine
// (window, tap) pairs of the block -- the pixel-per-thread kernel loaded 1 + 2 + 2 + 4 = 9 windows for the same four pixels and
// was bound by its L1 traffic (95 us for the 184 MB stem gradient of the ResNet-50 tile step, 3.1 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_block_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                                const T* __restrict__ y, T* __restrict__ dx, int N, int H, int W,
                                                                int C, int P, int Q) {
    const int CG = C / 8;
    const int Ha = (H + 1) / 2, Wb = (W + 1) / 2;
    const long long total = (long long)N * Ha * Wb * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        long long t = idx / CG;
        const int b = (int)(t % Wb); t /= Wb;
        const int a = (int)(t % Ha);
        const long long n = t / Ha;
        float acc[2][2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][j][e] = 0.f;
#pragma unroll
        for (int wp = 0; wp < 2; ++wp) {
            const int p = a + wp;
            if (p >= P) continue;
#pragma unroll
            for (int wq = 0; wq < 2; ++wq) {
                const int q = b + wq;
                if (q >= Q) continue;
                const long long o = ((n * P + p) * (long long)Q + q) * C + cg * 8;
                const uint2 pk = *reinterpret_cast<const uint2*>(amax + o);
                float g[8];
                load8<T>(dy + o, g);
                if (y) {
                    float yy[8];
                    load8<T>(y + o, yy);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
                }
                // pixel (2a + i, 2b + j) is tap (kh, kw) = (2a + i - (2p - 1), 2b + j - (2q - 1)) = (i + 1 - 2 wp, j + 1 - 2 wq) of this window
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int kh = i + 1 - 2 * wp;
                    if (kh < 0 || kh > 2) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int kw = j + 1 - 2 * wq;
                        if (kw < 0 || kw > 2) continue;
                        const uint32_t want = (uint32_t)(kh * 3 + kw);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const uint32_t am = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                            if (am == want) acc[i][j][e] += g[e];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = 2 * a + i;
            if (iy >= H) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ix = 2 * b + j;
                if (ix < W) store8<T>(dx + ((n * H + iy) * (long long)W + ix) * C + cg * 8, acc[i][j]);
            }
        }
    }
}

// Global avg+max: workgroup = 64 channel-groups x 4 row partitions.
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ feat,
                                                      int32_t* __restrict__ amax, int HW, int C, int with_max) {
    const int CG = C / 8;
    const int n = blockIdx.y;
    const int cg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    __shared__ float s_sum[4][64][8];
    __shared__ float s_max[4][64][8];
    __shared__ int s_idx[4][64][8];
    float sum[8], mx[8];
    int mi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[e] = 0.f; mx[e] = -INFINITY; mi[e] = 0x7fffffff; }
    if (cg < CG) {
        // four pixels in flight per thread (one dependent load per step made a 10 x 10 map a chain of 25 memory latencies: 17 us)
        const T* base = x + (long long)n * HW * C + cg * 8;
        int p = part;
        for (; p + 12 < HW; p += 16) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8<T>(base + (long long)(p + 4 * u) * C, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sum[e] += v[u][e];
                    if (mi[e] == 0x7fffffff || v[u][e] > mx[e] || v[u][e] != v[u][e]) { mx[e] = v[u][e]; mi[e] = p + 4 * u; }
                }
        }
        for (; p < HW; p += 4) {
            float v[8];
            load8<T>(base + (long long)p * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sum[e] += v[e];
                if (mi[e] == 0x7fffffff || v[e] > mx[e] || v[e] != v[e]) { mx[e] = v[e]; mi[e] = p; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s_sum[part][threadIdx.x & 63][e] = sum[e];
        s_max[part][threadIdx.x & 63][e] = mx[e];
        s_idx[part][threadIdx.x & 63][e] = mi[e];
    }
    __syncthreads();
    if (part == 0 && cg < CG) {
        const int l = threadIdx.x & 63;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = 0.f, m = -INFINITY;
            int bi = 0x7fffffff;
            for (int q = 0; q < 4; ++q) {
                s += s_sum[q][l][e];
                const float mq = s_max[q][l][e];
                const int iq = s_idx[q][l][e];
                if (iq == 0x7fffffff) continue;
                // first (smallest index) maximum wins ties, matching a sequential strict-> scan
                if (bi == 0x7fffffff || mq > m || (mq == m && iq < bi)) { m = mq; bi = iq; }
            }
            feat[(long long)n * C + cg * 8 + e] = s / (float)HW + (with_max ? m : 0.f);
            if (amax) amax[(long long)n * C + cg * 8 + e] = bi;
        }
    }
}

// Average only (the SE squeeze, HW up to 150*150): many workgroups per image, each reduces a slab of pixels and adds its
// partial mean with one atomic per channel (feat zeroed by the launcher).  grid = (channel-group chunks, N, pixel slabs).
template <typename T>
__global__ __launch_bounds__(256) void gap_avg_split_kernel(const T* __restrict__ x, float* __restrict__ feat, int HW, int C, int slab) {
    const int CG = C / 8;
    const int n = blockIdx.y;
    const int cg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int p0 = blockIdx.z * slab;
    int p1 = p0 + slab;
    if (p1 > HW) p1 = HW;
    __shared__ float red[4][64][8];
    float sum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = 0.f;
    if (cg < CG) {
        for (int p = p0 + part; p < p1; p += 4) {
            float v[8];
            load8<T>(x + ((long long)n * HW + p) * C + cg * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[part][threadIdx.x & 63][e] = sum[e];
    __syncthreads();
    if (part == 0 && cg < CG) {
        const int l = threadIdx.x & 63;
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            atomicAdd(feat + (long long)n * C + cg * 8 + e, (red[0][l][e] + red[1][l][e] + red[2][l][e] + red[3][l][e]) * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dfeat, const int32_t* __restrict__ amax,
                                                      const T* __restrict__ x, T* __restrict__ dx, int N, int HW, int C,
                                                      int relu_mask, int with_max) {
    const int CG = C / 8;
    const long long total = (long long)N * HW * CG;
    const float inv = 1.f / (float)HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        const long long t = idx / CG;
        const int p = (int)(t % HW);
        const long long n = t / HW;
        float v[8], g8[8];
        load8p(dfeat + n * C + cg * 8, 0.f, g8);
        int am[8];
        if (with_max) {
            const int4 a0 = *reinterpret_cast<const int4*>(amax + n * C + cg * 8);
            const int4 a1 = *reinterpret_cast<const int4*>(amax + n * C + cg * 8 + 4);
            am[0] = a0.x; am[1] = a0.y; am[2] = a0.z; am[3] = a0.w; am[4] = a1.x; am[5] = a1.y; am[6] = a1.z; am[7] = a1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g8[e] * inv + ((with_max && am[e] == p) ? g8[e] : 0.f);
        const long long o = (n * HW + p) * C + cg * 8;
        if (relu_mask) {
            float xx[8];
            load8<T>(x + o, xx);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xx[e] > 0.f ? v[e] : 0.f;
        }
        store8<T>(dx + o, v);
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int cs_maxpool3x3s2_fwd(const void* x, int dtype, void* y, uint8_t* argmax, int N, int H, int W, int C, int P,
                                   int Q, void* stream) {
    CS_CHECK_ARG(x && y, "maxpool_fwd: NULL tensor");
    CS_CHECK_ARG(C > 0 && C % 8 == 0, "maxpool_fwd: C must be a multiple of 8");
    CS_CHECK_ARG(P == (H + 2 - 3) / 2 + 1 && Q == (W + 2 - 3) / 2 + 1, "maxpool_fwd: P/Q do not match H/W");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * P * Q * (C / 8);
    if (dtype == CS_F32)
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, N, H, W, C, P, Q);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, N, H, W, C, P, Q);
    else
        CS_CHECK_ARG(false, "maxpool_fwd: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, const void* y_mask, int dtype, void* dx, int N,
                                   int H, int W, int C, int P, int Q, void* stream) {
    CS_CHECK_ARG(dy && argmax && dx, "maxpool_bwd: NULL tensor");
    CS_CHECK_ARG(C > 0 && C % 8 == 0, "maxpool_bwd: C must be a multiple of 8");
    CS_CHECK_ARG(P == (H + 2 - 3) / 2 + 1 && Q == (W + 2 - 3) / 2 + 1, "maxpool_bwd: P/Q do not match H/W");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    if (dtype == CS_F32)
        hipLaunchKernelGGL(maxpool_bwd_block_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)dy, argmax, (const float*)y_mask, (float*)dx, N, H, W, C, P, Q);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(maxpool_bwd_block_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)dy, argmax, (const bf16_t*)y_mask, (bf16_t*)dx, N, H, W, C, P, Q);
    else
        CS_CHECK_ARG(false, "maxpool_bwd: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_gap_avgmax_fwd(const void* x, int dtype, float* feat, int32_t* argmax, int N, int HW, int C, int with_max,
                                 void* stream) {
    CS_CHECK_ARG(x && feat, "gap_fwd: NULL tensor");
    CS_CHECK_ARG(N > 0 && HW > 0 && C > 0 && C % 8 == 0, "gap_fwd: bad extents");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((C / 8 + 63) / 64, N);
    if (!with_max && HW >= 1024) {
        int slabs = (HW + 511) / 512;
        if (slabs > 64) slabs = 64;
        const int slab = (HW + slabs - 1) / slabs;
        if (hipMemsetAsync(feat, 0, sizeof(float) * (size_t)N * C, st) != hipSuccess) { cs_set_error_("gap_fwd: memset failed"); return CS_ERR_LAUNCH; }
        dim3 g3((C / 8 + 63) / 64, N, (HW + slab - 1) / slab);
        if (dtype == CS_F32)
            hipLaunchKernelGGL(gap_avg_split_kernel<float>, g3, dim3(256), 0, st, (const float*)x, feat, HW, C, slab);
        else if (dtype == CS_BF16)
            hipLaunchKernelGGL(gap_avg_split_kernel<bf16_t>, g3, dim3(256), 0, st, (const bf16_t*)x, feat, HW, C, slab);
        else
            CS_CHECK_ARG(false, "gap_fwd: bad dtype");
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (dtype == CS_F32)
        hipLaunchKernelGGL(gap_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, feat, argmax, HW, C, with_max);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(gap_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, feat, argmax, HW, C, with_max);
    else
        CS_CHECK_ARG(false, "gap_fwd: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_gap_avgmax_bwd(const float* dfeat, const int32_t* argmax, const void* x, int dtype, void* dx, int N, int HW,
                                 int C, int relu_mask, int with_max, void* stream) {
    CS_CHECK_ARG(dfeat && dx && (argmax || !with_max), "gap_bwd: NULL tensor");
    CS_CHECK_ARG(!relu_mask || x, "gap_bwd: relu_mask needs x");
    CS_CHECK_ARG(N > 0 && HW > 0 && C > 0 && C % 8 == 0, "gap_bwd: bad extents");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * HW * (C / 8);
    if (dtype == CS_F32)
        hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, dfeat, argmax, (const float*)x, (float*)dx, N, HW, C, relu_mask, with_max);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(gap_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, dfeat, argmax, (const bf16_t*)x, (bf16_t*)dx, N, HW, C, relu_mask, with_max);
    else
        CS_CHECK_ARG(false, "gap_bwd: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}
