// MBConv pieces of the EfficientNet path (model/efficientnet.py:81-122 on top of torchvision 0.11.2's
// ConvNormActivation / SqueezeExcitation / StochasticDepth):
//   depthwise k x k convolution (k in {3,5}, stride in {1,2}), forward / data-grad / weight-grad
//   squeeze-excitation channel scaling and its backward
//   per-sample (row mode) stochastic-depth scale + residual add
// All HBM-bound NHWC kernels; k*k*2 FLOP per 2-4 bytes, so nothing here belongs on MFMA.  Three generations of depthwise kernels live
// here: element-per-thread (8 channels = one 16-byte access per thread; fp32 and odd geometries), channel-tiled (round 3, first half:
// the stride-2 data gradient per 2 x 2 block and one forward launch rule) and STRIPS (round 3, second half: bf16, k in {3, 5}, stride
// in {1, 2}: forward + statistics, stride-1 data gradient, weight gradient) -- see the comments at each family.
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
            else if (act == CS_ACT_SILU) v = silu_fast(v);
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
__global__ __launch_bounds__(256) void dw_wgrad_fold_kernel(const float* __restrict__ slab, int rows, int ncols, float* __restrict__ dw, int chan) {
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
        // column c = tap * C + channel; chan > 0: the parameter's own [C][1][R][S] layout (no permute + copy per depthwise layer afterwards)
        dw[chan > 0 ? (long long)(c % chan) * (ncols / chan) + c / chan : c] = t;
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

// out[n][c] += scale * sum over this workgroup's rows of a (* b): grid = (row blocks of ONE sample, N); a thread keeps one
// 8-channel group and every rpar-th row (the BN reductions' layout: all 256 lanes busy for any channel count -- the 64-group x 4-lane
// workgroups of se_ds_kernel / gap_fwd_kernel idle 72 % of their lanes on a 144-channel tensor), two rows in flight; LDS fold, one
// float atomic per (workgroup, channel) into the zeroed output.
template <typename T, bool PROD>
__global__ __launch_bounds__(256) void sample_rowsum_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ partial,
                                                            int HW, int C, int rows_per_block, int cw, float* __restrict__ out, float scale) {
    __shared__ float red[256][8];
    const int CG = C / 8;
    const int n = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > HW) r1 = HW;
    const T* an = a + (long long)n * HW * C;
    const T* bn = PROD ? b + (long long)n * HW * C : nullptr;
    {
        // grid = (row blocks of one sample, N, channel chunks of <= cw groups)
        const int cg0 = blockIdx.z * cw;
        const int width = (CG - cg0) < cw ? (CG - cg0) : cw;
        const int rpar = 256 / width;
        const int cg = cg0 + (int)(threadIdx.x % width);
        const int rr = threadIdx.x / width;
        const bool live = rr < rpar;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (live) {
            int r = r0 + rr;
            for (; r + rpar < r1; r += 2 * rpar) {
                float v0[8], v1[8];
                load8<T>(an + (long long)r * C + cg * 8, v0);
                load8<T>(an + (long long)(r + rpar) * C + cg * 8, v1);
                if constexpr (PROD) {
                    float w0[8], w1[8];
                    load8<T>(bn + (long long)r * C + cg * 8, w0);
                    load8<T>(bn + (long long)(r + rpar) * C + cg * 8, w1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += v0[e] * w0[e] + v1[e] * w1[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += v0[e] + v1[e];
                }
            }
            if (r < r1) {
                float v0[8];
                load8<T>(an + (long long)r * C + cg * 8, v0);
                if constexpr (PROD) {
                    float w0[8];
                    load8<T>(bn + (long long)r * C + cg * 8, w0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += v0[e] * w0[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += v0[e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = live ? acc[e] : 0.f;
        __syncthreads();
        // one partial row per workgroup, plain stores (no atomics: the fold below adds them in a fixed order, so the eval forward of an
        // EfficientNet -- and with it the adaptive top-k -- repeats bit for bit; a first version with float atomics did not)
        float* prow = partial + ((long long)n * gridDim.x + blockIdx.x) * C;
        for (int idx = threadIdx.x; idx < width * 8; idx += 256) {
            const int cl = idx >> 3, e = idx & 7;
            float t = 0.f;
            for (int q = 0; q < rpar; ++q) t += red[q * width + cl][e];
            // `out`: the launch has ONE row block per sample (small maps) and leaves the scaled sums themselves -- no fold launch
            if (out) out[(long long)n * C + (cg0 + cl) * 8 + e] = t * scale;
            else prow[(cg0 + cl) * 8 + e] = t;
        }
    }
}

// out[n][c] = scale * sum_b partial[n][b][c], b in ascending order
__global__ __launch_bounds__(256) void sample_rowsum_fold_kernel(const float* __restrict__ partial, float scale, float* __restrict__ out, int nblk,
                                                                 int C, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long n = idx / C;
    const int c = (int)(idx - n * C);
    const float* p = partial + n * nblk * C + c;
    float t = 0.f;
    int b = 0;
    for (; b + 7 < nblk; b += 8) {                       // eight loads in flight, added in ascending order (the same bits as one by one)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(b + u) * C];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < nblk; ++b) t += p[(long long)b * C];
    out[idx] = t * scale;
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
    // fixed-order fold of the threads that share a channel group (float LDS atomics added them in arrival order: not repeatable)
    __shared__ float fold[256][16];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        fold[threadIdx.x][e] = first < total ? s1[e] : 0.f;
        fold[threadIdx.x][8 + e] = first < total ? s2[e] : 0.f;
    }
    __syncthreads();
    const int t0 = (int)(((long long)blockIdx.x * 256) % CG);           // channel group of thread 0
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int a = i >= C ? 1 : 0, c = i - a * C;
        const int cgc = c >> 3, e = c & 7;
        float t = 0.f;
        for (int th = (cgc - t0 + CG) % CG; th < 256; th += CG) t += fold[th][a * 8 + e];
        lsum[i] = t;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) partial[(long long)blockIdx.x * 2 * C + i] = (double)lsum[i];
}

// ---- Round 3: channel-tiled depthwise kernels for the filter sizes EfficientNet uses (k in {3,5}, stride in {1,2}, "same" padding).
// The element-per-thread kernels above issue, per 16 bytes of output, k*k input loads AND 2*k*k weight loads through the vector L1
// (a thread's channel group changes with its flat index, so the filter cannot stay in registers): 1.2 KB of L1 traffic per 16 bytes
// stored -- L1-bound at ~0.9 TB/s on the EfficientNet-B3 step.  Here a workgroup is 32 channel groups (256 channels) x 8 pixel lanes:
//   * the filter slice [k*k][256] sits in LDS (25.6 KB for k = 5; both pixel lanes of a wave read the same address: broadcast);
//   * a thread produces TWO neighbouring output columns (forward / stride-1 gradient) or a 2 x 2 block of input pixels (stride-2
//     gradient) from one register window, so a 5x5 stride-1 output costs 15 input loads instead of 25 (+50 weight loads);
//   * the stride-2 data gradient is the gather form per 2 x 2 block: every (tap, output parity) pair is a compile-time constant and
//     each of the k*k taps is multiplied exactly once per block (the generic kernel walks k*k taps per pixel and rejects 3/4 of them).
constexpr int kDwCGT = 32;       // at most 32 channel groups per workgroup; the actual tile width cgt = ceil(CG / chunks) is a launch
                                 // argument and the workgroup runs 256 / cgt pixel lanes (EfficientNet-B3's depthwise layers have 5, 18, 24,
                                 // 36, 72, 102, 174 and 288 channel groups: a fixed 32-wide tile would idle 84 % of the lanes on the first)

template <int R, bool FLIP>
__device__ __forceinline__ void dw_stage_filter(const float* __restrict__ w, int C, int cg_base, int cgt, float (*wl)[kDwCGT][8]) {
    const int CG = C / 8;
    for (int i = threadIdx.x; i < R * R * cgt; i += 256) {
        const int tap = i / cgt, c_ = i - tap * cgt;
        const int cgx = cg_base + c_;
        float v[8];
        if (cgx < CG) load8p(w + (long long)(FLIP ? R * R - 1 - tap : tap) * C + cgx * 8, 0.f, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) wl[tap][c_][e] = v[e];
    }
}

// forward (FLIP = false) and stride-1 data gradient (FLIP = true, pad' = R - 1 - pad, source = dy): two output columns per thread.
// STATS: per-channel sum / sum of squares of the STORED values, one partial row per pixel block: partial[(2 * blockIdx.x + {0,1}) * C + c].
template <typename T, int R, int ST, bool STATS, bool FLIP>
__global__ __launch_bounds__(256) void dw_tile_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int act, T* __restrict__ y,
                                                      double* __restrict__ partial, int N, int H, int W, int C, int pad, int P, int Q,
                                                      int items_per_block, int cgt) {
    __shared__ float wl[R * R][kDwCGT][8];
    __shared__ float fold[STATS ? 2 : 1][STATS ? 256 : 1][8];
    const int npl = 256 / cgt;                       // pixel lanes
    const int cgl = threadIdx.x % cgt, pl = threadIdx.x / cgt;
    const int CG = C / 8;
    const int cg = blockIdx.y * cgt + cgl;
    const bool live = cg < CG && pl < npl;
    dw_stage_filter<R, FLIP>(w, C, blockIdx.y * cgt, cgt, wl);
    __syncthreads();
    constexpr int NC = ST + R;                       // input columns of the two outputs' window
    const int Qp = (Q + 1) / 2;
    const long long total = (long long)N * P * Qp;
    const long long i0 = (long long)blockIdx.x * items_per_block;
    long long i1 = i0 + items_per_block;
    if (i1 > total) i1 = total;
    float sc8[8], sh8[8], s1[8], s2[8];
    if (live) {
        load8p(scale ? scale + cg * 8 : nullptr, 1.f, sc8);
        load8p(shift ? shift + cg * 8 : nullptr, 0.f, sh8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (live) {
        for (long long it = i0 + pl; it < i1; it += npl) {
            const int oxp = (int)(it % Qp);
            const long long t = it / Qp;
            const int oy = (int)(t % P);
            const long long n = t / P;
            const int ox0 = 2 * oxp;
            const bool two = ox0 + 1 < Q;
            float a0[8], a1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
            const int ixb = ox0 * ST - pad;
#pragma unroll
            for (int kh = 0; kh < R; ++kh) {
                const int iy = oy * ST - pad + kh;
                if ((unsigned)iy < (unsigned)H) {
                    const T* rowp = x + ((n * H + iy) * (long long)W) * C + cg * 8;
                    float xv[NC][8];
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        const int ix = ixb + j;
                        if ((unsigned)ix < (unsigned)W) load8<T>(rowp + (long long)ix * C, xv[j]);
                        else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) xv[j][e] = 0.f;
                        }
                    }
#pragma unroll
                    for (int kw = 0; kw < R; ++kw) {
                        const float4 wa = *reinterpret_cast<const float4*>(&wl[kh * R + kw][cgl][0]);
                        const float4 wb = *reinterpret_cast<const float4*>(&wl[kh * R + kw][cgl][4]);
                        const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) { a0[e] += xv[kw][e] * wv[e]; a1[e] += xv[kw + ST][e] * wv[e]; }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v0 = a0[e] * sc8[e] + sh8[e], v1 = a1[e] * sc8[e] + sh8[e];
                if (act == CS_ACT_RELU) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
                else if (act == CS_ACT_SILU) { v0 = silu_fast(v0); v1 = silu_fast(v1); }
                a0[e] = v0; a1[e] = v1;
            }
            T* orow = y + ((n * P + oy) * (long long)Q + ox0) * C + cg * 8;
            store8<T>(orow, a0);
            if (two) store8<T>(orow + C, a1);
            if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r0 = to_f32<T>(from_f32<T>(a0[e]));       // statistics of the stored (rounded) values
                    const float r1 = two ? to_f32<T>(from_f32<T>(a1[e])) : 0.f;
                    s1[e] += r0 + r1; s2[e] += r0 * r0 + r1 * r1;
                }
            }
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { fold[0][threadIdx.x][e] = s1[e]; fold[1][threadIdx.x][e] = s2[e]; }
        __syncthreads();
        if (pl == 0 && live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                double t0 = 0.0, t1 = 0.0;
                for (int r = 0; r < npl; ++r) { t0 += (double)fold[0][r * cgt + cgl][e]; t1 += (double)fold[1][r * cgt + cgl][e]; }
                partial[(2LL * blockIdx.x) * C + cg * 8 + e] = t0;
                partial[(2LL * blockIdx.x + 1) * C + cg * 8 + e] = t1;
            }
        }
    }
}

// stride-2 data gradient, pad = (R - 1) / 2: a thread owns the 2 x 2 block of input pixels (2a + i, 2b + j); it reads the
// NW x NW window of dy rows a - (NW - 2) .. a + 1 (NW = (R + 1) / 2) and multiplies tap kh = i + pad + 2 * (NW - 2 - wy) (same in x)
// where that index is a tap of the filter -- all compile-time.
template <typename T, int R>
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int N,
                                                          int H, int W, int C, int P, int Q, int items_per_block, int cgt) {
    __shared__ float wl[R * R][kDwCGT][8];
    constexpr int PAD = (R - 1) / 2, NW = (R + 1) / 2;
    const int npl = 256 / cgt;
    const int cgl = threadIdx.x % cgt, pl = threadIdx.x / cgt;
    const int CG = C / 8;
    const int cg = blockIdx.y * cgt + cgl;
    dw_stage_filter<R, false>(w, C, blockIdx.y * cgt, cgt, wl);
    __syncthreads();
    if (cg >= CG || pl >= npl) return;
    const int Ha = (H + 1) / 2, Wb = (W + 1) / 2;
    const long long total = (long long)N * Ha * Wb;
    const long long i0 = (long long)blockIdx.x * items_per_block;
    long long i1 = i0 + items_per_block;
    if (i1 > total) i1 = total;
    for (long long it = i0 + pl; it < i1; it += npl) {
        const int b = (int)(it % Wb);
        const long long t = it / Wb;
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
        for (int wy = 0; wy < NW; ++wy) {
            const int oy = a - (NW - 2) + wy;
            if ((unsigned)oy >= (unsigned)P) continue;
#pragma unroll
            for (int wx = 0; wx < NW; ++wx) {
                const int ox = b - (NW - 2) + wx;
                if ((unsigned)ox >= (unsigned)Q) continue;
                float g[8];
                load8<T>(dy + ((n * P + oy) * (long long)Q + ox) * C + cg * 8, g);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int kh = i + PAD + 2 * (NW - 2 - wy);
                    if (kh < 0 || kh >= R) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int kw = j + PAD + 2 * (NW - 2 - wx);
                        if (kw < 0 || kw >= R) continue;
                        const float4 wa = *reinterpret_cast<const float4*>(&wl[kh * R + kw][cgl][0]);
                        const float4 wb = *reinterpret_cast<const float4*>(&wl[kh * R + kw][cgl][4]);
                        const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][j][e] += g[e] * wv[e];
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

// ---- Round 3 (second half): strip kernels.  A thread keeps ONE channel pair (a dword of bf16) and walks strips of TS output columns
// of one output row; a wave's 64 lanes are 64 neighbouring channel pairs, so every access is a coalesced 256-byte row piece.  Per
// filter row the strip's (TS - 1) * ST + R input dwords are loaded once, unpacked once and multiplied with v_pk_fma_f32 (two channels
// per instruction); all global accesses are raw-buffer loads whose offset is pushed out of range for padding columns / rows (no
// branches, no exec masking), and the loads of filter row kh + 1 are issued before the arithmetic of row kh.
// The element-per-thread weight gradient above multiplies every x element it loads (16 bytes) by ONE dy element: 28 L1 requests of
// 16 bytes per 32 bytes of HBM traffic and a dependent load per tap -- 0.8 TB/s, 3.5 ms of the 26 ms EfficientNet-B3 step.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f bf2_unpack(unsigned u) {
    v2f r;
    r.x = __uint_as_float(u << 16);
    r.y = __uint_as_float(u & 0xffff0000u);
    return r;
}
// XCD-aware block order: workgroup b runs on XCD b % 8 (each with its own L2); logical blocks are handed out so that one XCD walks a
// CONTIGUOUS eighth of the items -- neighbouring output rows share R - 1 of their R input rows, and with the plain order every input row
// was fetched into two or three L2s (counter traffic 2.2x the algorithmic bytes on the EfficientNet-B3 layers).
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb) {
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    return xcd * q + (xcd < r ? xcd : r) + (b >> 3);
}
constexpr unsigned kDwRowOut = 0x80000000u, kDwColOut = 0x40000000u;      // tensors below 1 GiB: either flag alone or both leave the range

// dw[kh][kw][c] partial rows: slab[blockIdx.x][R*R][C]; grid = (pixel blocks, channel-pair chunks); workgroup = cpt channel pairs x
// (256 / cpt) item lanes; item = (n, oy, strip of TS output columns).
template <int R, int ST, int TS>
__global__ __launch_bounds__(256) void dw_wgrad_strip_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ slab,
                                                             int N, int H, int W, int C, int pad, int P, int Q, int items_per_block, int cpt,
                                                             unsigned x_bytes, unsigned dy_bytes) {
    constexpr int NC = (TS - 1) * ST + R;
    constexpr int TB = 9;                                // taps folded per LDS pass
    __shared__ v2f red[TB][256];
    const int npl = 256 / cpt;
    const int cpl = threadIdx.x % cpt, pl = threadIdx.x / cpt;
    const int CP = C / 2;
    const int cp = blockIdx.y * cpt + cpl;
    const bool live = cp < CP && pl < npl;
    const int QS = (Q + TS - 1) / TS;
    const long long total = (long long)N * P * QS;
    const unsigned bx = xcd_block(blockIdx.x, gridDim.x);
    const long long i0 = (long long)bx * items_per_block;
    long long i1 = i0 + items_per_block;
    if (i1 > total) i1 = total;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, dy_bytes, 0x00020000);
    const unsigned Cb = (unsigned)C * 2u;                // bytes per pixel
    v2f acc[R * R];
#pragma unroll
    for (int t = 0; t < R * R; ++t) acc[t] = v2f{0.f, 0.f};
    if (live) {
        // (item -> (n, oy, strip) once, then carried: 64-bit divisions per item cost as much as a 3x3 item's arithmetic)
        int sx, oy, n;
        {
            const unsigned first = (unsigned)(i0 + pl);
            const unsigned t_ = first / (unsigned)QS;
            sx = (int)(first - t_ * (unsigned)QS);
            n = (int)(t_ / (unsigned)P);
            oy = (int)(t_ - (unsigned)n * (unsigned)P);
        }
        for (long long it = i0 + pl; it < i1; it += npl) {
            const int ox0 = sx * TS;
            // column offsets of the x window (padding columns pushed out of range) -- one set per item, shared by its R rows
            unsigned co[NC];
            const int ixb = ox0 * ST - pad;
#pragma unroll
            for (int j = 0; j < NC; ++j) co[j] = (unsigned)(ixb + j) < (unsigned)W ? (unsigned)(ixb + j) * Cb + (unsigned)cp * 4u : kDwColOut;
            unsigned gq[TS];
            const unsigned dyb = (unsigned)((n * P + oy) * Q + ox0) * Cb + (unsigned)cp * 4u;
#pragma unroll
            for (int j = 0; j < TS; ++j)
                gq[j] = __builtin_amdgcn_raw_buffer_load_b32(rdy, ox0 + j < Q ? dyb + (unsigned)j * Cb : kDwRowOut, 0, 0);
            auto row_base = [&](int kh) -> unsigned {
                const int iy = oy * ST - pad + kh;
                return (unsigned)iy < (unsigned)H ? (unsigned)((n * H + iy) * W) * Cb : kDwRowOut;
            };
            unsigned raw[2][NC];
            {
                const unsigned rb = row_base(0);
#pragma unroll
                for (int j = 0; j < NC; ++j) raw[0][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, rb + co[j], 0, 0);
            }
            v2f g[TS];
#pragma unroll
            for (int j = 0; j < TS; ++j) g[j] = bf2_unpack(gq[j]);
#pragma unroll
            for (int kh = 0; kh < R; ++kh) {
                if (kh + 1 < R) {
                    const unsigned rb = row_base(kh + 1);
#pragma unroll
                    for (int j = 0; j < NC; ++j) raw[(kh + 1) & 1][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, rb + co[j], 0, 0);
                }
                v2f xv[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) xv[j] = bf2_unpack(raw[kh & 1][j]);
#pragma unroll
                for (int kw = 0; kw < R; ++kw)
#pragma unroll
                    for (int j = 0; j < TS; ++j) acc[kh * R + kw] = __builtin_elementwise_fma(g[j], xv[j * ST + kw], acc[kh * R + kw]);
            }
            sx += npl;
            while (sx >= QS) { sx -= QS; if (++oy == P) { oy = 0; ++n; } }
        }
    }
    // fold the item lanes through LDS, TB taps per pass; thread (tap, channel pair) writes slab[blockIdx.x][tap][2cp .. 2cp+1]
#pragma unroll
    for (int t0 = 0; t0 < R * R; t0 += TB) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TB; ++t)
            if (t0 + t < R * R) red[t][threadIdx.x] = live ? acc[t0 + t] : v2f{0.f, 0.f};
        __syncthreads();
        const int nt = (R * R - t0) < TB ? (R * R - t0) : TB;
        for (int i = threadIdx.x; i < nt * cpt; i += 256) {
            const int t = i / cpt, c_ = i - t * cpt;
            const int cpx = blockIdx.y * cpt + c_;
            if (cpx < CP) {
                v2f sum = v2f{0.f, 0.f};
                for (int r = 0; r < npl; ++r) sum += red[t][r * cpt + c_];
                *reinterpret_cast<float2*>(slab + ((long long)bx * (R * R) + t0 + t) * C + 2 * cpx) = make_float2(sum.x, sum.y);
            }
        }
    }
}

// Forward (FLIP = false) and stride-1 data gradient (FLIP = true: source dy, pad' = R - 1 - pad, mirrored filter) on strips: the thread's
// R*R filter pairs stay in registers (its channel pair is fixed), TS accumulators, one unpacked input row at a time.
// STATS: per-channel sum / sum of squares of the STORED values, partial[(2 * blockIdx.x + {0,1}) * C + c] in fp64 (cs_bn_partial_fold).
template <int R, int ST, int TS, bool STATS, bool FLIP>
__global__ __launch_bounds__(256) void dw_conv_strip_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act, bf16_t* __restrict__ y,
                                                            double* __restrict__ partial, int N, int H, int W, int C, int pad, int P, int Q,
                                                            int items_per_block, int cpt, unsigned x_bytes, unsigned y_bytes) {
    constexpr int NC = (TS - 1) * ST + R;
    __shared__ v2f red[STATS ? 2 : 1][STATS ? 256 : 1];
    const int npl = 256 / cpt;
    const int cpl = threadIdx.x % cpt, pl = threadIdx.x / cpt;
    const int CP = C / 2;
    const int cp = blockIdx.y * cpt + cpl;
    const bool live = cp < CP && pl < npl;
    const int QS = (Q + TS - 1) / TS;
    const long long total = (long long)N * P * QS;
    const unsigned bx = xcd_block(blockIdx.x, gridDim.x);
    const long long i0 = (long long)bx * items_per_block;
    long long i1 = i0 + items_per_block;
    if (i1 > total) i1 = total;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, y_bytes, 0x00020000);
    const unsigned Cb = (unsigned)C * 2u;
    v2f s1 = v2f{0.f, 0.f}, s2 = v2f{0.f, 0.f};
    if (live) {
        v2f wv[R * R];
#pragma unroll
        for (int t = 0; t < R * R; ++t) {
            const float2 f = *reinterpret_cast<const float2*>(w + (long long)(FLIP ? R * R - 1 - t : t) * C + 2 * cp);
            wv[t] = v2f{f.x, f.y};
        }
        v2f sc = v2f{1.f, 1.f}, sh = v2f{0.f, 0.f};
        if (scale) { const float2 f = *reinterpret_cast<const float2*>(scale + 2 * cp); sc = v2f{f.x, f.y}; }
        if (shift) { const float2 f = *reinterpret_cast<const float2*>(shift + 2 * cp); sh = v2f{f.x, f.y}; }
        // (item -> (n, oy, strip) once, then carried: 64-bit divisions per item cost as much as a 3x3 item's arithmetic)
        int sx, oy, n;
        {
            const unsigned first = (unsigned)(i0 + pl);
            const unsigned t_ = first / (unsigned)QS;
            sx = (int)(first - t_ * (unsigned)QS);
            n = (int)(t_ / (unsigned)P);
            oy = (int)(t_ - (unsigned)n * (unsigned)P);
        }
        for (long long it = i0 + pl; it < i1; it += npl) {
            const int ox0 = sx * TS;
            unsigned co[NC];
            const int ixb = ox0 * ST - pad;
#pragma unroll
            for (int j = 0; j < NC; ++j) co[j] = (unsigned)(ixb + j) < (unsigned)W ? (unsigned)(ixb + j) * Cb + (unsigned)cp * 4u : kDwColOut;
            auto row_base = [&](int kh) -> unsigned {
                const int iy = oy * ST - pad + kh;
                return (unsigned)iy < (unsigned)H ? (unsigned)((n * H + iy) * W) * Cb : kDwRowOut;
            };
            unsigned raw[2][NC];
            {
                const unsigned rb = row_base(0);
#pragma unroll
                for (int j = 0; j < NC; ++j) raw[0][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, rb + co[j], 0, 0);
            }
            v2f acc[TS];
#pragma unroll
            for (int j = 0; j < TS; ++j) acc[j] = v2f{0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < R; ++kh) {
                if (kh + 1 < R) {
                    const unsigned rb = row_base(kh + 1);
#pragma unroll
                    for (int j = 0; j < NC; ++j) raw[(kh + 1) & 1][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, rb + co[j], 0, 0);
                }
                v2f xv[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) xv[j] = bf2_unpack(raw[kh & 1][j]);
#pragma unroll
                for (int kw = 0; kw < R; ++kw)
#pragma unroll
                    for (int j = 0; j < TS; ++j) acc[j] = __builtin_elementwise_fma(xv[j * ST + kw], wv[kh * R + kw], acc[j]);
            }
            const unsigned yb = (unsigned)((n * P + oy) * Q + ox0) * Cb + (unsigned)cp * 4u;
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                v2f v = __builtin_elementwise_fma(acc[j], sc, sh);
                if (act == CS_ACT_RELU) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; }
                else if (act == CS_ACT_SILU) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); }
                const unsigned u = pack_bf16x2(v.x, v.y);
                const bool in = ox0 + j < Q;
                __builtin_amdgcn_raw_buffer_store_b32(u, ry, in ? yb + (unsigned)j * Cb : kDwRowOut, 0, 0);
                if constexpr (STATS) {
                    const v2f r = in ? bf2_unpack(u) : v2f{0.f, 0.f};          // statistics of the stored (rounded) values
                    s1 += r;
                    s2 = __builtin_elementwise_fma(r, r, s2);
                }
            }
            sx += npl;
            while (sx >= QS) { sx -= QS; if (++oy == P) { oy = 0; ++n; } }
        }
    }
    if constexpr (STATS) {
        red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
        __syncthreads();
        if (pl == 0 && live) {
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            for (int r = 0; r < npl; ++r) {
                const v2f u1 = red[0][r * cpt + cpl], u2 = red[1][r * cpt + cpl];
                a0 += (double)u1.x; a1 += (double)u1.y; b0 += (double)u2.x; b1 += (double)u2.y;
            }
            double* d0 = partial + (2LL * bx) * C + 2 * cp;
            double* d1 = partial + (2LL * bx + 1) * C + 2 * cp;
            d0[0] = a0; d0[1] = a1; d1[0] = b0; d1[1] = b1;
        }
    }
}

// channel-pair tile of the strip kernels: cpt lanes x (256 / cpt) item lanes; the tile that keeps most lanes busy, at least 16 pairs
// (64 contiguous bytes per pixel)
inline void dw_strip_shape(int C, int& chunks, int& cpt) {
    const int CP = C / 2;
    int best_chunks = 1, best_cpt = CP < 256 ? CP : 256;
    double best = -1.0;
    for (int ch = (CP + 255) / 256; ch <= (CP + 15) / 16; ++ch) {
        const int t = (CP + ch - 1) / ch;
        if (t > 256) continue;
        const double lane_use = (double)((256 / t) * t) / 256.0, fill = (double)CP / ((double)ch * t);
        const double score = lane_use * fill - 0.0005 * ch;      // (ties: fewer chunks = longer contiguous runs)
        if (score > best) { best = score; best_chunks = ch; best_cpt = t; }
    }
    chunks = best_chunks; cpt = best_cpt;
}

// channel-group tile of the tiled kernels: chunks = ceil(CG / 32) workgroup columns of cgt = ceil(CG / chunks) channel groups
inline void dw_tile_shape(int C, int& chunks, int& cgt) {
    const int CG = C / 8;
    chunks = (CG + kDwCGT - 1) / kDwCGT;
    cgt = (CG + chunks - 1) / chunks;
}
// work items per workgroup of the tiled kernels: ~2048 workgroups in all, at least 4 items per pixel lane
inline int dw_tile_items(long long items, int chunks, int cgt) {
    long long blocks = 2048 / (chunks > 0 ? chunks : 1);
    if (blocks < 1) blocks = 1;
    long long per = (items + blocks - 1) / blocks;
    const int npl = 256 / cgt;
    if (per < 4 * npl) per = 4 * npl;
    return (int)per;
}
// Which launches take the tiled kernels: measured per EfficientNet-B3 layer (tools/dw_microbench.py, profiles/round3_notes.md).
// Both families are VALU-bound, not L1-bound as assumed (k = 5 layers: 25 multiply-adds + the bf16 unpacking per output element run
// at 0.45-0.8 TB/s either way), so the tile only pays where it removes work: every stride-2 data gradient (2-3x: 286 -> 176, 186 ->
// 62, 39 -> 18, 52 -> 25 us), the 3x3 stride-1 forwards (180 -> 138, 58 -> 47, 56 -> 34, 77 -> 49 us) and the narrow early layers.
inline bool dw_tiled_geometry(const CsConvGeom* g, int kind /* 0 forward, 1 data gradient */) {
    static const int off = cs_env_int_("CELLSEG_DW_UNTILED", 0);      // A/B experiments only: 1 = the element-per-thread kernels
    if (off || !((g->R == 3 || g->R == 5) && (g->stride == 1 || g->stride == 2) && g->pad == (g->R - 1) / 2)) return false;
    if (off == 2) return true;                                        // (2 = tiled wherever the geometry allows)
    if (kind == 1) return g->stride == 2 || g->C <= 48;
    return g->R == 3 && (g->stride == 1 || g->C >= 288);
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

// launch of dw_tile_kernel<T, R, ST, STATS, FLIP> on (source x [N][H][W][C]) -> (y [N][P][Q][C])
template <typename T, bool STATS, bool FLIP>
static void launch_dw_tile(int R, int ST, hipStream_t st, const void* x, const float* w, const float* scale, const float* shift, int act,
                           void* y, double* partial, int N, int H, int W, int C, int pad, int P, int Q, int* rows_out) {
    int chunks, cgt;
    dw_tile_shape(C, chunks, cgt);
    const long long items = (long long)N * P * ((Q + 1) / 2);
    const int per = dw_tile_items(items, chunks, cgt);
    dim3 grid((unsigned)((items + per - 1) / per), (unsigned)chunks);
    if (rows_out) *rows_out = (int)grid.x;
#define CS_DW_TILE(R_, S_)                                                                                                      \
    hipLaunchKernelGGL((dw_tile_kernel<T, R_, S_, STATS, FLIP>), grid, dim3(256), 0, st, (const T*)x, w, scale, shift, act, (T*)y, partial, \
                       N, H, W, C, pad, P, Q, per, cgt)
    if (R == 3 && ST == 1) CS_DW_TILE(3, 1);
    else if (R == 3) CS_DW_TILE(3, 2);
    else if (ST == 1) CS_DW_TILE(5, 1);
    else CS_DW_TILE(5, 2);
#undef CS_DW_TILE
}
// ---- strip kernels (bf16): geometry of a launch over (source N x H x W) -> (destination N x P x Q)
struct DwStrip { int ts, per, chunks, cpt; unsigned nblk; };
static bool dw_strip_ok(int R, int stride, long long src_elems, long long dst_elems, int dtype) {
    static const int off = cs_env_int_("CELLSEG_DW_NOSTRIP", 0);      // A/B experiments only
    return !off && dtype == CS_BF16 && (R == 3 || R == 5) && (stride == 1 || stride == 2) && src_elems * 2 < (1ll << 30) && dst_elems * 2 < (1ll << 30);
}
// strip length: the one of the two instantiated per stride that wastes fewer columns; `blocks_target` workgroups in all
static DwStrip dw_strip_plan(int N, int P, int Q, int C, int stride, long long blocks_target, long long rows_cap) {
    DwStrip d;
    const int a = stride == 1 ? 8 : 4, b = 5;
    const int wa = (Q + a - 1) / a * a, wb = (Q + b - 1) / b * b;
    d.ts = wb < wa ? b : a;
    dw_strip_shape(C, d.chunks, d.cpt);
    const long long items = (long long)N * P * ((Q + d.ts - 1) / d.ts);
    long long blocks = blocks_target / d.chunks;
    if (blocks < 1) blocks = 1;
    if (rows_cap > 0 && blocks > rows_cap) blocks = rows_cap;
    long long p_ = (items + blocks - 1) / blocks;
    const int npl = 256 / d.cpt;
    if (p_ < 2 * npl) p_ = 2 * npl;
    d.per = (int)p_;
    d.nblk = (unsigned)((items + p_ - 1) / p_);
    return d;
}
// forward launches that stay on the channel-tiled kernel: the 144-channel 3x3 stride-2 layer at 150 x 150 (194 vs 231 us, tools/dw_microbench.py)
static bool dw_strip_fwd_ok(const CsConvGeom* g, int dtype) {
    if (g->stride == 2 && g->R == 3 && g->C <= 144 && dw_tiled_geometry(g, 0)) return false;
    return dw_strip_ok(g->R, g->stride, (long long)g->N * g->H * g->W * g->C, (long long)g->N * g->P * g->Q * g->C, dtype);
}
static DwStrip dw_strip_plan_conv(int N, int P, int Q, int C, int stride) {
    return dw_strip_plan(N, P, Q, C, stride, 4096, 2048);            // (partial statistics rows: at most 2048 per launch)
}
template <bool STATS, bool FLIP>
static void launch_dw_strip(int R, int ST, hipStream_t st, const void* x, const float* w, const float* scale, const float* shift, int act,
                            void* y, double* partial, int N, int H, int W, int C, int pad, int P, int Q, int* rows_out) {
    const DwStrip d = dw_strip_plan_conv(N, P, Q, C, ST);
    if (rows_out) *rows_out = (int)d.nblk;
    const unsigned xb = (unsigned)((long long)N * H * W * C * 2), yb = (unsigned)((long long)N * P * Q * C * 2);
#define CS_DW_CS(R_, S_, T_)                                                                                                              \
    hipLaunchKernelGGL((dw_conv_strip_kernel<R_, S_, T_, STATS, FLIP>), dim3(d.nblk, (unsigned)d.chunks), dim3(256), 0, st, (const bf16_t*)x, w, \
                       scale, shift, act, (bf16_t*)y, partial, N, H, W, C, pad, P, Q, d.per, d.cpt, xb, yb)
    if (R == 3) {
        if (ST == 1) { if (d.ts == 8) CS_DW_CS(3, 1, 8); else CS_DW_CS(3, 1, 5); }
        else { if (d.ts == 4) CS_DW_CS(3, 2, 4); else CS_DW_CS(3, 2, 5); }
    } else {
        if (ST == 1) { if (d.ts == 8) CS_DW_CS(5, 1, 8); else CS_DW_CS(5, 1, 5); }
        else { if (d.ts == 4) CS_DW_CS(5, 2, 4); else CS_DW_CS(5, 2, 5); }
    }
#undef CS_DW_CS
}

static int dw_tile_rows(const CsConvGeom* g) {
    int chunks, cgt;
    dw_tile_shape(g->C, chunks, cgt);
    const long long items = (long long)g->N * g->P * ((g->Q + 1) / 2);
    const int per = dw_tile_items(items, chunks, cgt);
    return (int)((items + per - 1) / per);
}

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
    if (dw_strip_fwd_ok(g, dtype)) {
        launch_dw_strip<false, false>(g->R, g->stride, st, x, w_hwc, scale, shift, act, y, nullptr, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, nullptr);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (dw_tiled_geometry(g, 0) && (dtype == CS_F32 || dtype == CS_BF16)) {
        if (dtype == CS_F32) launch_dw_tile<float, false, false>(g->R, g->stride, st, x, w_hwc, scale, shift, act, y, nullptr, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, nullptr);
        else launch_dw_tile<bf16_t, false, false>(g->R, g->stride, st, x, w_hwc, scale, shift, act, y, nullptr, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, nullptr);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
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
    size_t strip_rows = 0;                               // (the dtype is not known here: room for whichever kernel serves the call)
    if (dw_strip_fwd_ok(g, CS_BF16))
        strip_rows = dw_strip_plan_conv(g->N, g->P, g->Q, g->C, g->stride).nblk;
    const size_t other_rows = dw_tiled_geometry(g, 0) ? (size_t)dw_tile_rows(g) : (size_t)dw_stats_grid(g);
    return (strip_rows > other_rows ? strip_rows : other_rows) * 2 * (size_t)g->C * sizeof(double);
}
extern "C" int cs_dwconv_fwd_stats(const CsConvGeom* g, int dtype, const void* x, const float* w_hwc, void* y, double* partial,
                                   int* partial_rows, void* stream) {
    int rc = check_dw(g, "dwconv_fwd_stats: bad geometry (square filter, K == C, C % 8 == 0 required)");
    if (rc) return rc;
    CS_CHECK_ARG(x && w_hwc && y && partial && partial_rows, "dwconv_fwd_stats: NULL argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dw_strip_fwd_ok(g, dtype)) {
        launch_dw_strip<true, false>(g->R, g->stride, st, x, w_hwc, nullptr, nullptr, CS_ACT_NONE, y, partial, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, partial_rows);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (dw_tiled_geometry(g, 0) && (dtype == CS_F32 || dtype == CS_BF16)) {
        if (dtype == CS_F32) launch_dw_tile<float, true, false>(g->R, g->stride, st, x, w_hwc, nullptr, nullptr, CS_ACT_NONE, y, partial, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, partial_rows);
        else launch_dw_tile<bf16_t, true, false>(g->R, g->stride, st, x, w_hwc, nullptr, nullptr, CS_ACT_NONE, y, partial, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, partial_rows);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    CS_CHECK_ARG((size_t)2 * g->C * sizeof(float) <= 65536, "dwconv_fwd_stats: too many channels for the LDS fold");
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
    if (g->stride == 1 && g->pad <= g->R - 1 && dw_strip_ok(g->R, 1, (long long)g->N * g->P * g->Q * g->C, (long long)g->N * g->H * g->W * g->C, dtype)) {
        // the gradient of a stride-1 convolution is the same convolution with the filter mirrored (source dy, destination dx)
        launch_dw_strip<false, true>(g->R, 1, st, dy, w_hwc, nullptr, nullptr, CS_ACT_NONE, dx, nullptr, g->N, g->P, g->Q, g->C, g->R - 1 - g->pad, g->H, g->W, nullptr);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (dw_tiled_geometry(g, 1) && (dtype == CS_F32 || dtype == CS_BF16)) {
        if (g->stride == 1) {
            // the gradient of a stride-1 "same" convolution is the same convolution with the filter mirrored (source dy, destination dx)
            if (dtype == CS_F32) launch_dw_tile<float, false, true>(g->R, 1, st, dy, w_hwc, nullptr, nullptr, CS_ACT_NONE, dx, nullptr, g->N, g->P, g->Q, g->C, g->R - 1 - g->pad, g->H, g->W, nullptr);
            else launch_dw_tile<bf16_t, false, true>(g->R, 1, st, dy, w_hwc, nullptr, nullptr, CS_ACT_NONE, dx, nullptr, g->N, g->P, g->Q, g->C, g->R - 1 - g->pad, g->H, g->W, nullptr);
        } else {
            int chunks, cgt;
            dw_tile_shape(g->C, chunks, cgt);
            const long long items = (long long)g->N * ((g->H + 1) / 2) * ((g->W + 1) / 2);
            const int per = dw_tile_items(items, chunks, cgt);
            dim3 grid((unsigned)((items + per - 1) / per), (unsigned)chunks);
#define CS_DW_S2(T_, R_) hipLaunchKernelGGL((dw_dgrad_s2_kernel<T_, R_>), grid, dim3(256), 0, st, (const T_*)dy, w_hwc, (T_*)dx, g->N, g->H, g->W, g->C, g->P, g->Q, per, cgt)
            if (dtype == CS_F32) { if (g->R == 3) CS_DW_S2(float, 3); else CS_DW_S2(float, 5); }
            else { if (g->R == 3) CS_DW_S2(bf16_t, 3); else CS_DW_S2(bf16_t, 5); }
#undef CS_DW_S2
        }
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
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

// strip weight gradient: the same planning as the strip convolutions (dw_strip_plan); partial rows: at most 32 MiB, at least 128 rows
static DwStrip dw_strip_plan_wgrad(const CsConvGeom* g) {
    long long cap = (32ll << 20) / ((long long)g->R * g->R * g->C * 4);
    if (cap < 128) cap = 128;
    return dw_strip_plan(g->N, g->P, g->Q, g->C, g->stride, 4096, cap);
}
static bool dw_strip_wgrad_ok(const CsConvGeom* g, int dtype) {
    return dw_strip_ok(g->R, g->stride, (long long)g->N * g->H * g->W * g->C, (long long)g->N * g->P * g->Q * g->C, dtype);
}

extern "C" size_t cs_dwconv_wgrad_workspace(const CsConvGeom* g) {
    if (!g || check_dw(g, "dwconv_wgrad_workspace: bad geometry")) return 0;
    int rpb; unsigned nslab;
    dw_wgrad_split(g, rpb, nslab);
    if ((g->R == 3 || g->R == 5) && (g->stride == 1 || g->stride == 2)) {      // (either kernel may serve the call: room for both)
        const unsigned nblk = dw_strip_plan_wgrad(g).nblk;
        if (nblk > nslab) nslab = nblk;
    }
    return (size_t)nslab * g->R * g->R * g->C * sizeof(float);
}

static int dwconv_wgrad_impl(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_hwc, float* workspace, int chan, void* stream) {
    int rc = check_dw(g, "dwconv_wgrad: bad geometry");
    if (rc) return rc;
    CS_CHECK_ARG(x && dy && dw_hwc && workspace, "dwconv_wgrad: NULL tensor (workspace: cs_dwconv_wgrad_workspace bytes)");
    CS_CHECK_ARG(dtype == CS_F32 || dtype == CS_BF16, "dwconv_wgrad: bad dtype");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ncols = g->R * g->R * g->C;
    if (dw_strip_wgrad_ok(g, dtype)) {
        const DwStrip d_ = dw_strip_plan_wgrad(g);
        const int ts = d_.ts, per = d_.per, sch = d_.chunks, cpt = d_.cpt;
        const unsigned nblk = d_.nblk;
        const unsigned xb = (unsigned)((long long)g->N * g->H * g->W * g->C * 2), yb = (unsigned)((long long)g->N * g->P * g->Q * g->C * 2);
#define CS_DW_STRIP(R_, S_, T_)                                                                                                          \
    hipLaunchKernelGGL((dw_wgrad_strip_kernel<R_, S_, T_>), dim3(nblk, (unsigned)sch), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, \
                       workspace, g->N, g->H, g->W, g->C, g->pad, g->P, g->Q, per, cpt, xb, yb)
        if (g->R == 3) {
            if (g->stride == 1) { if (ts == 8) CS_DW_STRIP(3, 1, 8); else CS_DW_STRIP(3, 1, 5); }
            else { if (ts == 4) CS_DW_STRIP(3, 2, 4); else CS_DW_STRIP(3, 2, 5); }
        } else {
            if (g->stride == 1) { if (ts == 8) CS_DW_STRIP(5, 1, 8); else CS_DW_STRIP(5, 1, 5); }
            else { if (ts == 4) CS_DW_STRIP(5, 2, 4); else CS_DW_STRIP(5, 2, 5); }
        }
#undef CS_DW_STRIP
        CS_LAUNCH_CHECK();
        hipLaunchKernelGGL(dw_wgrad_fold_kernel, dim3((unsigned)((ncols + 15) / 16)), dim3(256), 0, st, workspace, (int)nblk, ncols, dw_hwc, chan);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
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
    hipLaunchKernelGGL(dw_wgrad_fold_kernel, dim3((unsigned)((ncols + 15) / 16)), dim3(256), 0, st, workspace, (int)nslab, ncols, dw_hwc, chan);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_dwconv_wgrad(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_hwc, float* workspace, void* stream) {
    return dwconv_wgrad_impl(g, dtype, x, dy, dw_hwc, workspace, 0, stream);
}
extern "C" int cs_dwconv_wgrad_oihw(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_oihw, float* workspace, void* stream) {
    CS_CHECK_ARG(g && g->C > 0, "dwconv_wgrad_oihw: bad geometry");
    return dwconv_wgrad_impl(g, dtype, x, dy, dw_oihw, workspace, g->C, stream);
}

// Depthwise filters [C][1][R][S] (the parameters) -> [R][S][C] (what the depthwise kernels read), ALL layers of a network in one launch:
// one strided torch copy per layer and step was 26 launches of 4.4 us on EfficientNet-B3.  desc (device): n rows of CsDwStageDesc, `first`
// ascending; one thread per element.
__global__ __launch_bounds__(256) void dw_weights_hwc_multi_kernel(const CsDwStageDesc* __restrict__ desc, int n, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    const CsDwStageDesc d = desc[lo];
    const long long e = i - d.first;                 // destination element: tap * C + c
    const long long tap = e / d.C, c = e - tap * d.C;
    d.dst[e] = d.src[c * d.RS + tap];
}
extern "C" int cs_dw_weights_hwc_multi(const CsDwStageDesc* desc_dev, int n, long long total, void* stream) {
    CS_CHECK_ARG(desc_dev && n > 0 && total > 0, "dw_weights_hwc_multi: bad arguments");
    hipLaunchKernelGGL(dw_weights_hwc_multi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc_dev, n, total);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

static void sample_rowsum_shape(int N, int HW, int C, int& rpb, int& nblk, int& cw) {
    const int CG = C / 8;
    if (HW <= 512) {
        // small maps (EfficientNet stages 4-7 at 299 x 299: 19 x 19 and 10 x 10): ONE workgroup per (sample, chunk of <= 32 channel groups)
        // walks all rows -- no partial rows, no fold launch (7.6-9.8 us for squeeze + fold of 13-18 MB tensors, half of it the second launch)
        const int chunks = (CG + 31) / 32;
        cw = (CG + chunks - 1) / chunks;
        rpb = HW;
        nblk = 1;
        return;
    }
    // ~2048 workgroups over the N samples, at least 8 row steps per thread, at most 64 partial rows per sample
    cw = CG < 256 ? CG : 256;
    const int rpar = 256 / cw;
    int per_sample = 2048 / (N > 0 ? N : 1);
    if (per_sample < 1) per_sample = 1;
    if (per_sample > 64) per_sample = 64;
    rpb = (HW + per_sample - 1) / per_sample;
    if (rpb < 8 * rpar) rpb = 8 * rpar;
    nblk = (HW + rpb - 1) / rpb;
}

extern "C" size_t cs_sample_sum_workspace(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0 || C % 8) return 0;
    int rpb, nblk, cw;
    sample_rowsum_shape(N, HW, C, rpb, nblk, cw);
    return (size_t)N * nblk * C * sizeof(float);
}

extern "C" int cs_sample_sum(const void* a, const void* b, int dtype, float scale, float* out, float* workspace, int N, int HW, int C, void* stream) {
    CS_CHECK_ARG(a && out && workspace && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "sample_sum: bad arguments (workspace: cs_sample_sum_workspace bytes)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rpb, nblk, cw;
    sample_rowsum_shape(N, HW, C, rpb, nblk, cw);
    const int CG = C / 8;
    dim3 grid((unsigned)nblk, (unsigned)N, (unsigned)((CG + cw - 1) / cw));
    float* direct = nblk == 1 ? out : nullptr;
    if (dtype == CS_F32) {
        if (b) hipLaunchKernelGGL((sample_rowsum_kernel<float, true>), grid, dim3(256), 0, st, (const float*)a, (const float*)b, workspace, HW, C, rpb, cw, direct, scale);
        else hipLaunchKernelGGL((sample_rowsum_kernel<float, false>), grid, dim3(256), 0, st, (const float*)a, (const float*)nullptr, workspace, HW, C, rpb, cw, direct, scale);
    } else if (dtype == CS_BF16) {
        if (b) hipLaunchKernelGGL((sample_rowsum_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, workspace, HW, C, rpb, cw, direct, scale);
        else hipLaunchKernelGGL((sample_rowsum_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)nullptr, workspace, HW, C, rpb, cw, direct, scale);
    } else {
        cs_set_error_("sample_sum: bad dtype");
        return CS_ERR_INVALID_ARG;
    }
    CS_LAUNCH_CHECK();
    if (direct) return CS_OK;
    const long long total = (long long)N * C;
    hipLaunchKernelGGL(sample_rowsum_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, scale, out, nblk, C, total);
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
