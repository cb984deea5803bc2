// MBConv pieces of the EfficientNet path (model/efficientnet.py:81-122 on top of torchvision 0.11.2's
// ConvNormActivation / SqueezeExcitation / StochasticDepth):
//   depthwise k x k convolution (k in {3,5}, stride in {1,2}), forward / data-grad / weight-grad
//   squeeze-excitation channel scaling and its backward
//   per-sample (row mode) stochastic-depth scale + residual add
// All HBM-bound NHWC kernels, 8 channels (one 16-byte access for bf16) per thread; k*k*2 FLOP per
// 2-4 bytes, so nothing here belongs on MFMA.
#include "cs_common.h"

namespace {

// y[n,oy,ox,c] = act( (sum_taps x[n,oy*s-p+kh,ox*s-p+kw,c] * w[kh][kw][c]) * scale[c] + shift[c] )
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int act, T* __restrict__ y, int N, int H, int W,
                                                     int C, int R, int stride, int pad, int P, int Q) {
    const int CG = C / 8;
    const long long total = (long long)N * P * Q * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        long long t = idx / CG;
        const int ox = (int)(t % Q); t /= Q;
        const int oy = (int)(t % P);
        const long long n = t / P;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < R; ++kh) {
            const int iy = oy * stride - pad + kh;
            if (iy < 0 || iy >= H) continue;
            for (int kw = 0; kw < R; ++kw) {
                const int ix = ox * stride - pad + kw;
                if (ix < 0 || ix >= W) continue;
                float v[8], wp[8];
                load8<T>(x + ((n * H + iy) * (long long)W + ix) * C + cg * 8, v);
                load8p(w + (kh * R + kw) * C + cg * 8, 0.f, wp);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e] * wp[e];
            }
        }
        float sc8[8], sh8[8];
        load8p(scale ? scale + cg * 8 : nullptr, 1.f, sc8);
        load8p(shift ? shift + cg * 8 : nullptr, 0.f, sh8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[e] * sc8[e] + sh8[e];
            if (act == CS_ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (act == CS_ACT_SILU) v = v / (1.f + __expf(-v));
            acc[e] = v;
        }
        store8<T>(y + ((n * P + oy) * (long long)Q + ox) * C + cg * 8, acc);
    }
}

// dx[n,iy,ix,c] = sum_taps dy[n,(iy+p-kh)/s,(ix+p-kw)/s,c] * w[kh][kw][c]   (gather form, no atomics)
template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int N,
                                                       int H, int W, int C, int R, int stride, int pad, int P, int Q) {
    const int CG = C / 8;
    const long long total = (long long)N * H * W * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        long long t = idx / CG;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long long n = t / H;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < R; ++kh) {
            const int ty = iy + pad - kh;
            if (ty < 0 || ty % stride != 0) continue;
            const int oy = ty / stride;
            if (oy >= P) continue;
            for (int kw = 0; kw < R; ++kw) {
                const int tx = ix + pad - kw;
                if (tx < 0 || tx % stride != 0) continue;
                const int ox = tx / stride;
                if (ox >= Q) continue;
                float g[8], wp[8];
                load8<T>(dy + ((n * P + oy) * (long long)Q + ox) * C + cg * 8, g);
                load8p(w + (kh * R + kw) * C + cg * 8, 0.f, wp);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += g[e] * wp[e];
            }
        }
        store8<T>(dx + ((n * H + iy) * (long long)W + ix) * C + cg * 8, acc);
    }
}

// dw[kh][kw][c] += sum over output pixels of dy * x.  Workgroup = (slab of output pixels) x (chunk of <= 64 channel groups)
// x (group of <= 9 filter taps); threads = W channel groups x (256/W) pixel lanes.  Pixels are the OUTER loop: dy is read once
// per pixel and each of the group's taps keeps its own 8-channel accumulator in registers (72 VGPRs); afterwards the pixel
// lanes are folded through LDS and the workgroup leaves ONE partial row slab[blockIdx.x][tap][c]; dw_wgrad_fold_kernel sums the
// rows.  (Round 1 left one atomicAdd per (tap, channel) instead: ~500 workgroups x 4608 atomics onto a few thousand addresses
// serialised in L2 -- 262 us average per EfficientNet-B3 layer, 189 us with the partial rows.  A sliding-window variant that
// walks output rows and loads only the new window column per pixel (4 instead of 10 loads) was measured SLOWER, 204 us: its
// loads depend on the previous step and one or two waves per SIMD cannot hide them -- the kernel is latency-, not request-bound.)
constexpr int kDwTaps = 9;
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ slab, int N,
                                                       int H, int W, int C, int R, int stride, int pad, int P, int Q, int pix_per_block) {
    const int CG = C / 8;
    const int cg0 = blockIdx.y * 64;
    const int width = (CG - cg0) < 64 ? (CG - cg0) : 64;
    const int lanes = 256 / width;
    const int cgl = threadIdx.x % width;
    const int pl = threadIdx.x / width;
    const int cg = cg0 + cgl;
    const bool live = pl < lanes;
    const int t0 = blockIdx.z * kDwTaps;
    const int nt = (R * R - t0) < kDwTaps ? (R * R - t0) : kDwTaps;
    const long long npix = (long long)N * P * Q;
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    float acc[kDwTaps][8];
#pragma unroll
    for (int t = 0; t < kDwTaps; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    if (live) {
        long long pp = p0 + pl;
        long long n = pp / ((long long)P * Q);
        int rem = (int)(pp - n * (long long)P * Q);
        int oy = rem / Q, ox = rem - oy * Q;
        for (; pp < p1; pp += lanes) {
            float g[8];
            load8<T>(dy + pp * C + cg * 8, g);
            const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
#pragma unroll
            for (int t = 0; t < kDwTaps; ++t) {
                if (t < nt) {
                    const int tap = t0 + t;
                    const int kh = tap / R, kw = tap - kh * R;
                    const int iy = iy0 + kh, ix = ix0 + kw;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                        float v[8];
                        load8<T>(x + ((n * H + iy) * (long long)W + ix) * C + cg * 8, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[t][e] += g[e] * v[e];
                    }
                }
            }
            ox += lanes;
            while (ox >= Q) { ox -= Q; if (++oy == P) { oy = 0; ++n; } }
        }
    }
    __shared__ float red[256][8];
#pragma unroll
    for (int t = 0; t < kDwTaps; ++t) {
        // nt is workgroup-uniform; no early exit so the loop fully unrolls and acc[t] stays in registers
        if (t < nt) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = live ? acc[t][e] : 0.f;
            __syncthreads();
            if ((int)threadIdx.x < width) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float sum = 0.f;
                    for (int r = 0; r < lanes; ++r) sum += red[r * width + threadIdx.x][e];
                    slab[((long long)blockIdx.x * (R * R) + t0 + t) * C + cg * 8 + e] = sum;
                }
            }
        }
    }
}

// dw[i] = sum_r slab[r][i]: workgroup = 16 columns x 16 row lanes, 8 loads in flight per thread
__global__ __launch_bounds__(256) void dw_wgrad_fold_kernel(const float* __restrict__ slab, int rows, int ncols, float* __restrict__ dw) {
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < ncols) {
        int r = rl;
        for (; r + 7 * 16 < rows; r += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(long long)(r + u * 16) * ncols + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < rows; r += 16) acc += slab[(long long)r * ncols + c];
    }
    __shared__ float red[16][17];
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][cl];
        dw[c] = t;
    }
}

// y[n,p,c] = x[n,p,c] * s[n,c]
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ y, int N,
                                                       int HW, int C) {
    const int CG = C / 8;
    const long long total = (long long)N * HW * CG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        const long long pix = idx / CG;
        const long long n = pix / HW;
        float v[8], sp[8];
        load8<T>(x + pix * C + cg * 8, v);
        load8p(s + n * C + cg * 8, 1.f, sp);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= sp[e];
        store8<T>(y + pix * C + cg * 8, v);
    }
}

// ds[n,c] = sum_p dy[n,p,c] * x[n,p,c]   (workgroup = (64 channel groups x 4 pixel lanes), one image)
template <typename T>
__global__ __launch_bounds__(256) void se_ds_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ ds, int HW, int C,
                                                    int slab) {
    // grid = (channel-group chunks, N, pixel slabs): partial sums combined with one atomic per channel (ds zeroed by launcher)
    const int CG = C / 8;
    const int n = blockIdx.y;
    const int cg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int p0 = blockIdx.z * slab;
    int p1 = p0 + slab;
    if (p1 > HW) p1 = HW;
    __shared__ float red[4][64][8];
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (cg < CG) {
        for (int p = p0 + part; p < p1; p += 4) {
            float g[8], v[8];
            const long long o = ((long long)n * HW + p) * C + cg * 8;
            load8<T>(dy + o, g);
            load8<T>(x + o, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += g[e] * v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[part][threadIdx.x & 63][e] = acc[e];
    __syncthreads();
    if (part == 0 && cg < CG) {
        const int l = threadIdx.x & 63;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(ds + (long long)n * C + cg * 8 + e, red[0][l][e] + red[1][l][e] + red[2][l][e] + red[3][l][e]);
    }
}

// dx[n,p,c] = dy[n,p,c] * s[n,c] + davg[n,c] / HW        (davg nullable)
template <typename T>
__global__ __launch_bounds__(256) void se_dx_kernel(const T* __restrict__ dy, const float* __restrict__ s, const float* __restrict__ davg,
                                                    T* __restrict__ dx, int N, int HW, int C) {
    const int CG = C / 8;
    const long long total = (long long)N * HW * CG;
    const float inv = 1.f / (float)HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % CG);
        const long long pix = idx / CG;
        const long long n = pix / HW;
        float g[8], s8[8], d8[8];
        load8<T>(dy + pix * C + cg * 8, g);
        load8p(s + n * C + cg * 8, 1.f, s8);
        load8p(davg ? davg + n * C + cg * 8 : nullptr, 0.f, d8);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = g[e] * s8[e] + d8[e] * inv;
        store8<T>(dx + pix * C + cg * 8, g);
    }
}

// y[n,p,c] = a[n,p,c] * rs[n] + b[n,p,c]      (StochasticDepth "row" + residual add; rs nullable = 1)
template <typename T>
__global__ __launch_bounds__(256) void rowscale_add_kernel(const T* __restrict__ a, const float* __restrict__ rs, const T* __restrict__ b,
                                                           T* __restrict__ y, long long per_row, long long total8) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total8; idx += (long long)gridDim.x * blockDim.x) {
        const long long o = idx * 8;
        const float s = rs ? rs[o / per_row] : 1.f;
        float va[8], vb[8];
        load8<T>(a + o, va);
        if (b) {
            load8<T>(b + o, vb);
#pragma unroll
            for (int e = 0; e < 8; ++e) va[e] = va[e] * s + vb[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) va[e] *= s;
        }
        store8<T>(y + o, va);
    }
}

// Depthwise forward that also leaves the per-channel sum / sum of squares of the STORED output (train-mode BN statistics) as one
// partial row per workgroup: partial[(2*b + {0,1}) * C + c] in fp64 -- folded by cs_bn_partial_fold.  Saves the separate
// cs_bn_stats pass over z (26 launches of ~100 us per EfficientNet-B3 step).  The grid is a multiple of CG / gcd(CG, 256)
// workgroups, so a thread keeps its 8 channels for its whole grid-stride walk and accumulates in registers.
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_stats_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                           double* __restrict__ partial, int N, int H, int W, int C, int R, int stride,
                                                           int pad, int P, int Q) {
    extern __shared__ float lsum[];          // [2][C]
    for (int i = threadIdx.x; i < 2 * C; i += 256) lsum[i] = 0.f;
    __syncthreads();
    const int CG = C / 8;
    const long long total = (long long)N * P * Q * CG;
    const long long first = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(first % CG);        // constant along the walk: gridDim.x * 256 is a multiple of CG
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    for (long long idx = first; idx < total; idx += (long long)gridDim.x * 256) {
        long long t = idx / CG;
        const int ox = (int)(t % Q); t /= Q;
        const int oy = (int)(t % P);
        const long long n = t / P;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < R; ++kh) {
            const int iy = oy * stride - pad + kh;
            if (iy < 0 || iy >= H) continue;
            for (int kw = 0; kw < R; ++kw) {
                const int ix = ox * stride - pad + kw;
                if (ix < 0 || ix >= W) continue;
                float v[8], wp[8];
                load8<T>(x + ((n * H + iy) * (long long)W + ix) * C + cg * 8, v);
                load8p(w + (kh * R + kw) * C + cg * 8, 0.f, wp);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e] * wp[e];
            }
        }
        store8<T>(y + ((n * P + oy) * (long long)Q + ox) * C + cg * 8, acc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float r = to_f32<T>(from_f32<T>(acc[e]));       // statistics of the stored (rounded) value
            s1[e] += r; s2[e] += r * r;
        }
    }
    if (first < total) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            atomicAdd(&lsum[cg * 8 + e], s1[e]);
            atomicAdd(&lsum[C + cg * 8 + e], s2[e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) partial[(long long)blockIdx.x * 2 * C + i] = (double)lsum[i];
}

inline int grid_ew(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define CS_T_SWITCH(dtype, NAME, F32, BF16)                                            \
    if (dtype == CS_F32) { F32; } else if (dtype == CS_BF16) { BF16; }                 \
    else { cs_set_error_(NAME ": bad dtype"); return CS_ERR_INVALID_ARG; }

static int check_dw(const CsConvGeom* g, const char* what) {
    if (!g || g->R != g->S || g->R < 1 || g->C % 8 != 0 || g->K != g->C || g->stride < 1 ||
        g->P != (g->H + 2 * g->pad - g->R) / g->stride + 1 || g->Q != (g->W + 2 * g->pad - g->S) / g->stride + 1) {
        cs_set_error_(what);
        return CS_ERR_INVALID_ARG;
    }
    return CS_OK;
}

extern "C" int cs_dwconv_fwd(const CsConvGeom* g, int dtype, const void* x, const float* w_hwc, const float* scale, const float* shift,
                             int act, void* y, void* stream) {
    int rc = check_dw(g, "dwconv_fwd: bad geometry (square filter, K == C, C % 8 == 0 required)");
    if (rc) return rc;
    CS_CHECK_ARG(x && w_hwc && y, "dwconv_fwd: NULL tensor");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew((long long)g->N * g->P * g->Q * (g->C / 8));
    CS_T_SWITCH(dtype, "dwconv_fwd",
                hipLaunchKernelGGL(dw_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, w_hwc, scale, shift, act, (float*)y,
                                   g->N, g->H, g->W, g->C, g->R, g->stride, g->pad, g->P, g->Q),
                hipLaunchKernelGGL(dw_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, w_hwc, scale, shift, act,
                                   (bf16_t*)y, g->N, g->H, g->W, g->C, g->R, g->stride, g->pad, g->P, g->Q));
    CS_LAUNCH_CHECK();
    return CS_OK;
}

static int dw_stats_grid(const CsConvGeom* g) {
    const int CG = g->C / 8;
    int a = CG, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }           // gcd(CG, 256)
    const int m = CG / a;                                      // workgroup-count granule
    long long want = ((long long)g->N * g->P * g->Q * CG + 255) / 256;
    static const int cap = cs_env_int_("CELLSEG_DW_STATS_BLOCKS", 1024);     // A/B experiments only
    if (want > cap) want = cap;
    long long grid = (want + m - 1) / m * m;
    if (grid < m) grid = m;
    return (int)grid;
}

extern "C" size_t cs_dwconv_fwd_stats_workspace(const CsConvGeom* g) {
    if (!g || g->C <= 0 || g->C % 8) return 0;
    return (size_t)dw_stats_grid(g) * 2 * (size_t)g->C * sizeof(double);
}

extern "C" int cs_dwconv_fwd_stats(const CsConvGeom* g, int dtype, const void* x, const float* w_hwc, void* y, double* partial,
                                   int* partial_rows, void* stream) {
    int rc = check_dw(g, "dwconv_fwd_stats: bad geometry (square filter, K == C, C % 8 == 0 required)");
    if (rc) return rc;
    CS_CHECK_ARG(x && w_hwc && y && partial && partial_rows, "dwconv_fwd_stats: NULL argument");
    CS_CHECK_ARG((size_t)2 * g->C * sizeof(float) <= 65536, "dwconv_fwd_stats: too many channels for the LDS fold");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = dw_stats_grid(g);
    const size_t lds = (size_t)2 * g->C * sizeof(float);
    *partial_rows = grid;
    CS_T_SWITCH(dtype, "dwconv_fwd_stats",
                hipLaunchKernelGGL(dw_fwd_stats_kernel<float>, dim3(grid), dim3(256), lds, st, (const float*)x, w_hwc, (float*)y, partial, g->N,
                                   g->H, g->W, g->C, g->R, g->stride, g->pad, g->P, g->Q),
                hipLaunchKernelGGL(dw_fwd_stats_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, (const bf16_t*)x, w_hwc, (bf16_t*)y, partial,
                                   g->N, g->H, g->W, g->C, g->R, g->stride, g->pad, g->P, g->Q));
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_dwconv_dgrad(const CsConvGeom* g, int dtype, const void* dy, const float* w_hwc, void* dx, void* stream) {
    int rc = check_dw(g, "dwconv_dgrad: bad geometry");
    if (rc) return rc;
    CS_CHECK_ARG(dy && w_hwc && dx, "dwconv_dgrad: NULL tensor");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew((long long)g->N * g->H * g->W * (g->C / 8));
    CS_T_SWITCH(dtype, "dwconv_dgrad",
                hipLaunchKernelGGL(dw_dgrad_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, w_hwc, (float*)dx, g->N, g->H,
                                   g->W, g->C, g->R, g->stride, g->pad, g->P, g->Q),
                hipLaunchKernelGGL(dw_dgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, w_hwc, (bf16_t*)dx, g->N, g->H,
                                   g->W, g->C, g->R, g->stride, g->pad, g->P, g->Q));
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// output-row slabs of the weight gradient: ~1024 workgroups in total (no atomics, so parallelism is free), <= 256 partial rows
static void dw_wgrad_split(const CsConvGeom* g, int& rows_per_block, unsigned& nslab) {
    const long long rows = (long long)g->N * g->P;
    const int chunks = (g->C / 8 + 63) / 64 * ((g->R * g->R + kDwTaps - 1) / kDwTaps);
    static const int dw_target = cs_env_int_("CELLSEG_DW_BLOCKS", 1024);   // A/B experiments only
    long long slabs = dw_target / chunks;
    if (slabs < 1) slabs = 1;
    // every slab is one fp32 partial row of R*R*C floats for the fold: at most 32 MiB of them, at least 256
    long long cap = (32ll << 20) / ((long long)g->R * g->R * g->C * 4);
    if (cap < 256) cap = 256;
    if (slabs > cap) slabs = cap;
    long long rpb = (rows + slabs - 1) / slabs;
    const long long min_rows = (64 + g->Q - 1) / g->Q;           // >= 64 pixels per workgroup
    if (rpb < min_rows) rpb = min_rows;
    rows_per_block = (int)rpb;
    nslab = (unsigned)((rows + rpb - 1) / rpb);
}

extern "C" size_t cs_dwconv_wgrad_workspace(const CsConvGeom* g) {
    if (!g || check_dw(g, "dwconv_wgrad_workspace: bad geometry")) return 0;
    int rpb; unsigned nslab;
    dw_wgrad_split(g, rpb, nslab);
    return (size_t)nslab * g->R * g->R * g->C * sizeof(float);
}

extern "C" int cs_dwconv_wgrad(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_hwc, float* workspace, void* stream) {
    int rc = check_dw(g, "dwconv_wgrad: bad geometry");
    if (rc) return rc;
    CS_CHECK_ARG(x && dy && dw_hwc && workspace, "dwconv_wgrad: NULL tensor (workspace: cs_dwconv_wgrad_workspace bytes)");
    CS_CHECK_ARG(dtype == CS_F32 || dtype == CS_BF16, "dwconv_wgrad: bad dtype");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rpb; unsigned nslab;
    dw_wgrad_split(g, rpb, nslab);
    const int chunks = (g->C / 8 + 63) / 64;
    const int tgroups = (g->R * g->R + kDwTaps - 1) / kDwTaps;
    dim3 grid(nslab, (unsigned)chunks, (unsigned)tgroups);
    const int ppb = rpb * g->Q;
    if (dtype == CS_F32)
        hipLaunchKernelGGL(dw_wgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)dy, workspace, g->N, g->H, g->W, g->C,
                           g->R, g->stride, g->pad, g->P, g->Q, ppb);
    else
        hipLaunchKernelGGL(dw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, workspace, g->N, g->H, g->W,
                           g->C, g->R, g->stride, g->pad, g->P, g->Q, ppb);
    CS_LAUNCH_CHECK();
    const int ncols = g->R * g->R * g->C;
    hipLaunchKernelGGL(dw_wgrad_fold_kernel, dim3((unsigned)((ncols + 15) / 16)), dim3(256), 0, st, workspace, (int)nslab, ncols, dw_hwc);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_se_scale(const void* x, int dtype, const float* s, void* y, int N, int HW, int C, void* stream) {
    CS_CHECK_ARG(x && s && y && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_ew((long long)N * HW * (C / 8));
    CS_T_SWITCH(dtype, "se_scale",
                hipLaunchKernelGGL(se_scale_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, s, (float*)y, N, HW, C),
                hipLaunchKernelGGL(se_scale_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, s, (bf16_t*)y, N, HW, C));
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_se_scale_bwd(const void* dy, const void* x, int dtype, const float* s, const float* davg, float* ds, void* dx, int N,
                               int HW, int C, int phase, void* stream) {
    /* phase 0: ds[n,c] = sum_p dy*x ; phase 1: dx = dy*s + davg/HW */
    CS_CHECK_ARG(dy && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale_bwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (phase == 0) {
        CS_CHECK_ARG(x && ds, "se_scale_bwd: phase 0 needs x and ds");
        int slabs = (HW + 511) / 512;
        if (slabs > 64) slabs = 64;
        const int slab = (HW + slabs - 1) / slabs;
        if (hipMemsetAsync(ds, 0, sizeof(float) * (size_t)N * C, st) != hipSuccess) { cs_set_error_("se_scale_bwd: memset failed"); return CS_ERR_LAUNCH; }
        dim3 grid((C / 8 + 63) / 64, N, (HW + slab - 1) / slab);
        CS_T_SWITCH(dtype, "se_scale_bwd",
                    hipLaunchKernelGGL(se_ds_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, ds, HW, C, slab),
                    hipLaunchKernelGGL(se_ds_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, ds, HW, C, slab));
    } else {
        CS_CHECK_ARG(s && dx, "se_scale_bwd: phase 1 needs s and dx");
        const int grid = grid_ew((long long)N * HW * (C / 8));
        CS_T_SWITCH(dtype, "se_scale_bwd",
                    hipLaunchKernelGGL(se_dx_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, s, davg, (float*)dx, N, HW, C),
                    hipLaunchKernelGGL(se_dx_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, s, davg, (bf16_t*)dx, N, HW, C));
    }
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_rowscale_add(const void* a, int dtype, const float* row_scale, const void* b, void* y, int N, long long per_row,
                               void* stream) {
    CS_CHECK_ARG(a && y && N > 0 && per_row > 0 && per_row % 8 == 0, "rowscale_add: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total8 = (long long)N * per_row / 8;
    const int grid = grid_ew(total8);
    CS_T_SWITCH(dtype, "rowscale_add",
                hipLaunchKernelGGL(rowscale_add_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)a, row_scale, (const float*)b,
                                   (float*)y, per_row, total8),
                hipLaunchKernelGGL(rowscale_add_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)a, row_scale, (const bf16_t*)b,
                                   (bf16_t*)y, per_row, total8));
    CS_LAUNCH_CHECK();
    return CS_OK;
}
