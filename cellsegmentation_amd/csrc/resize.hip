// Segmentation-decoder data movement (HBM-bound, 8 channels per thread):
//   F.interpolate(mode="bilinear", align_corners=True)   model/resnet.py:282,287,292,297,300
//   torch.cat([a, b], dim=1)                            model/resnet.py:284,289,294
#include "cs_common.h"

namespace {

struct Tap { int i0, i1; float w0, w1; };

// ATen's align_corners source index: scale = (in-1)/(out-1) in fp32, src = scale*dst,
// i0 = floor(src), i1 = i0 + (i0 < in-1), lambda = src - i0.
__device__ __forceinline__ Tap tap_of(int o, int in, int out) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)o;
    int i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
    float l = src - (float)i0;
    l = fminf(fmaxf(l, 0.f), 1.f);
    return Tap{i0, i1, 1.f - l, l};
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                           int P, int Q) {
    const int CG = C / 8;
    const long long total = (long long)N * P * Q * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        long long t = idx / CG;
        const int q = (int)(t % Q); t /= Q;
        const int p = (int)(t % P);
        const long long n = t / P;
        const Tap ty = tap_of(p, H, P), tx = tap_of(q, W, Q);
        float a[8], b[8], c[8], d[8], o[8];
        const T* base = x + n * (long long)H * W * C + cg * 8;
        load8<T>(base + ((long long)ty.i0 * W + tx.i0) * C, a);
        load8<T>(base + ((long long)ty.i0 * W + tx.i1) * C, b);
        load8<T>(base + ((long long)ty.i1 * W + tx.i0) * C, c);
        load8<T>(base + ((long long)ty.i1 * W + tx.i1) * C, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ty.w0 * (tx.w0 * a[e] + tx.w1 * b[e]) + ty.w1 * (tx.w0 * c[e] + tx.w1 * d[e]);
        store8<T>(y + ((n * P + p) * (long long)Q + q) * C + cg * 8, o);
    }
}

// Gather-form backward: an input pixel collects from the contiguous range of output rows/cols
// whose taps touch it (recomputed with the same fp32 formula, so fwd/bwd weights agree exactly).
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ mask, T* __restrict__ dx,
                                                           int N, int H, int W, int C, int P, int Q) {
    const int CG = C / 8;
    const long long total = (long long)N * H * W * CG;
    const float sy = H > 1 ? (float)(P - 1) / (float)(H - 1) : 0.f;   // inverse scales (out per in)
    const float sx = W > 1 ? (float)(Q - 1) / (float)(W - 1) : 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        long long t = idx / CG;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long long n = t / H;
        int p_lo = H > 1 ? (int)floorf((float)(iy - 1) * sy) - 1 : 0;
        int p_hi = H > 1 ? (int)ceilf((float)(iy + 1) * sy) + 1 : P - 1;
        int q_lo = W > 1 ? (int)floorf((float)(ix - 1) * sx) - 1 : 0;
        int q_hi = W > 1 ? (int)ceilf((float)(ix + 1) * sx) + 1 : Q - 1;
        if (p_lo < 0) p_lo = 0;
        if (q_lo < 0) q_lo = 0;
        if (p_hi > P - 1) p_hi = P - 1;
        if (q_hi > Q - 1) q_hi = Q - 1;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int p = p_lo; p <= p_hi; ++p) {
            const Tap ty = tap_of(p, H, P);
            const float wy = (ty.i0 == iy ? ty.w0 : 0.f) + (ty.i1 == iy ? ty.w1 : 0.f);
            if (wy == 0.f) continue;
            for (int q = q_lo; q <= q_hi; ++q) {
                const Tap tx = tap_of(q, W, Q);
                const float wx = (tx.i0 == ix ? tx.w0 : 0.f) + (tx.i1 == ix ? tx.w1 : 0.f);
                if (wx == 0.f) continue;
                float g[8];
                load8<T>(dy + ((n * P + p) * (long long)Q + q) * C + cg * 8, g);
                const float w = wy * wx;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += w * g[e];
            }
        }
        const long long o = ((n * H + iy) * (long long)W + ix) * C + cg * 8;
        if (mask) {
            float m8[8];
            load8<T>(mask + o, m8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = m8[e] > 0.f ? acc[e] : 0.f;
        }
        store8<T>(dx + o, acc);
    }
}

// out[m][0:Ca] = a[m], out[m][Ca:] = b[m]  (dir=0)   or the reverse split (dir=1; a/b nullable)
template <typename T>
__global__ __launch_bounds__(256) void concat_kernel(T* __restrict__ a, T* __restrict__ b, T* __restrict__ out, long long M, int Ca,
                                                     int Cb, int dir) {
    const int CG = (Ca + Cb) / 8;
    const long long total = M * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CG) * 8;
        const long long m = idx / CG;
        T* part = c < Ca ? (a ? a + m * Ca + c : nullptr) : (b ? b + m * Cb + (c - Ca) : nullptr);
        if (!part) continue;
        T* whole = out + m * (Ca + Cb) + c;
        if (sizeof(T) == 4) {
            const float4* s = reinterpret_cast<const float4*>(dir == 0 ? part : whole);
            float4* d = reinterpret_cast<float4*>(dir == 0 ? whole : part);
            d[0] = s[0]; d[1] = s[1];
        } else {
            *reinterpret_cast<uint4*>(dir == 0 ? whole : part) = *reinterpret_cast<const uint4*>(dir == 0 ? part : whole);
        }
    }
}

inline int grid_ew(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int cs_bilinear_ac_fwd(const void* x, int dtype, void* y, int N, int H, int W, int C, int P, int Q, void* stream) {
    CS_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && C > 0 && C % 8 == 0, "bilinear_fwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew((long long)N * P * Q * (C / 8));
    if (dtype == CS_F32)
        hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, P, Q);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, P, Q);
    else
        CS_CHECK_ARG(false, "bilinear_fwd: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_bilinear_ac_bwd(const void* dy, const void* mask, int dtype, void* dx, int N, int H, int W, int C, int P, int Q,
                                  void* stream) {
    CS_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && C > 0 && C % 8 == 0, "bilinear_bwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew((long long)N * H * W * (C / 8));
    if (dtype == CS_F32)
        hipLaunchKernelGGL(bilinear_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)mask, (float*)dx, N, H, W, C, P, Q);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(bilinear_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)mask, (bf16_t*)dx, N, H, W, C, P, Q);
    else
        CS_CHECK_ARG(false, "bilinear_bwd: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_concat_channels(const void* a, const void* b, int dtype, void* out, long long M, int Ca, int Cb, void* stream) {
    CS_CHECK_ARG(a && b && out && M > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "concat: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew(M * ((Ca + Cb) / 8));
    if (dtype == CS_F32)
        hipLaunchKernelGGL(concat_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)a, (float*)b, (float*)out, M, Ca, Cb, 0);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(concat_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (bf16_t*)a, (bf16_t*)b, (bf16_t*)out, M, Ca, Cb, 0);
    else
        CS_CHECK_ARG(false, "concat: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_split_channels(const void* whole, int dtype, void* a, void* b, long long M, int Ca, int Cb, void* stream) {
    CS_CHECK_ARG(whole && (a || b) && M > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "split: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew(M * ((Ca + Cb) / 8));
    if (dtype == CS_F32)
        hipLaunchKernelGGL(concat_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)a, (float*)b, (float*)whole, M, Ca, Cb, 1);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(concat_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (bf16_t*)a, (bf16_t*)b, (bf16_t*)whole, M, Ca, Cb, 1);
    else
        CS_CHECK_ARG(false, "split: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// =============================================================================================
// Tile gather: the step BEFORE the hot path.  Replaces, for a batch of tiles,
//   image[x:x+size, y:y+size]  (dataset/dataset.py:207-209)  ->  ToTensor (/255)  ->  Normalize(mean, std)
// (dataset/dataset.py:78-83) and the NCHW->NHWC staging: uint8 HWC images resident in HBM -> NHWC tiles in the
// compute dtype with the 3 colour channels padded to 8.  One thread per output pixel (3 bytes in, 16/32 bytes out):
// overlapping tiles (32-px tiles at stride 20) re-read the image from L2 instead of crossing PCIe 2.6x.
// =============================================================================================
namespace {

template <typename T>
__global__ __launch_bounds__(256) void tile_gather_kernel(const uint8_t* __restrict__ images, const int32_t* __restrict__ tile_img,
                                                          const int32_t* __restrict__ tile_rc, long long T_tiles, int H, int W, int size,
                                                          float m0, float m1, float m2, float s0, float s1, float s2,
                                                          T* __restrict__ out) {
    const long long total = T_tiles * size * size;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % size);
        const int y = (int)((idx / size) % size);
        const long long t = idx / ((long long)size * size);
        const int r = tile_rc[2 * t] + y, c = tile_rc[2 * t + 1] + x;
        const uint8_t* px = images + (((long long)tile_img[t] * H + r) * W + c) * 3;
        float v[8];
        // same fp32 operation order as ToTensor + Normalize: (u8 / 255 - mean) / std
        v[0] = ((float)px[0] / 255.0f - m0) / s0;
        v[1] = ((float)px[1] / 255.0f - m1) / s1;
        v[2] = ((float)px[2] / 255.0f - m2) / s2;
        v[3] = v[4] = v[5] = v[6] = v[7] = 0.f;
        store8<T>(out + idx * 8, v);
    }
}

}  // namespace

extern "C" int cs_tile_gather(const uint8_t* images, int n_images, int H, int W, const int32_t* tile_img, const int32_t* tile_rc,
                              long long n_tiles, int size, const float* host_mean3, const float* host_std3, int dtype, void* out,
                              void* stream) {
    CS_CHECK_ARG(images && tile_img && tile_rc && out && host_mean3 && host_std3, "tile_gather: NULL argument");
    CS_CHECK_ARG(n_images > 0 && H > 0 && W > 0 && n_tiles > 0 && size > 0 && size <= H && size <= W, "tile_gather: bad extents");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew(n_tiles * size * size);
    if (dtype == CS_F32)
        hipLaunchKernelGGL(tile_gather_kernel<float>, dim3(grid), dim3(256), 0, st, images, tile_img, tile_rc, n_tiles, H, W, size,
                           host_mean3[0], host_mean3[1], host_mean3[2], host_std3[0], host_std3[1], host_std3[2], (float*)out);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(tile_gather_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, images, tile_img, tile_rc, n_tiles, H, W, size,
                           host_mean3[0], host_mean3[1], host_mean3[2], host_std3[0], host_std3[1], host_std3[2], (bf16_t*)out);
    else
        CS_CHECK_ARG(false, "tile_gather: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}
