// Layout / precision staging kernels around the conv family: boundary transposes, BatchNorm
// folding, weight staging, weight-gradient finalisation, per-channel column sums.
// All HBM-bound elementwise or small reductions.
#include "cs_common.h"

namespace {

// ---- NCHW fp32 -> NHWC T (channel padded) : one thread per (pixel, 8 stored channels) ----------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int Cp) {
    const int groups = Cp / 8 > 0 ? Cp / 8 : 1;
    const long long total = (long long)N * HW * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int gq = (int)(idx % groups);
        const long long pix = idx / groups;      // n*HW + p
        const long long n = pix / HW;
        const int p = (int)(pix - n * HW);
        if (Cp % 8 == 0) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = gq * 8 + e;
                v[e] = c < C ? x[(n * C + c) * HW + p] : 0.f;
            }
            store8<T>(y + pix * Cp + gq * 8, v);
        } else {
            // Cp == 4 (fp32 stem input)
            for (int c = 0; c < Cp; ++c) y[pix * Cp + c] = from_f32<T>(c < C ? x[(n * C + c) * HW + p] : 0.f);
        }
    }
}

// ---- NHWC T -> NCHW fp32: LDS-free, reads 8 channels per thread, writes are strided by HW but
// consecutive threads take consecutive pixels so each channel plane gets coalesced stores. -------
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ y, float* __restrict__ x, int N, int C, int HW, int Cp) {
    const int groups = (C + 7) / 8;
    const long long total = (long long)N * HW * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx % ((long long)N * HW);
        const int gq = (int)(idx / ((long long)N * HW));
        const long long n = pix / HW;
        const int p = (int)(pix - n * HW);
        for (int e = 0; e < 8; ++e) {
            const int c = gq * 8 + e;
            if (c < C) x[(n * C + c) * HW + p] = to_f32<T>(y[pix * Cp + c]);
        }
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               const float* conv_bias, float* scale, float* shift, float* rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // same operation order as ATen's eval batch_norm: invstd = 1/sqrt(var+eps)
    const float r = 1.0f / sqrtf(var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    const float s = g * r;
    if (scale) scale[c] = s;
    if (shift) shift[c] = b + ((conv_bias ? conv_bias[c] : 0.f) - mean[c]) * s;
    if (rstd) rstd[c] = r;
}

// ---- weights: fp32 [K][Cin][R][S] (x scale[k]) -> T [K][R][S][Cp] and/or T [Cin][R][S][Kp] ------
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ scale, int K, int Cin, int R,
                                   int S, int Cp, int Kp, T* __restrict__ w_khwc, T* __restrict__ w_chwk) {
    const int RS = R * S;
    if (w_khwc) {
        const long long total = (long long)Kp * RS * Cp;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(idx % Cp);
            const int rs = (int)((idx / Cp) % RS);
            const int k = (int)(idx / ((long long)Cp * RS));
            float v = 0.f;
            if (c < Cin && k < K) {
                v = w[((long long)k * Cin + c) * RS + rs];
                if (scale) v *= scale[k];
            }
            w_khwc[idx] = from_f32<T>(v);
        }
    }
    if (w_chwk) {
        const long long total = (long long)Cp * RS * Kp;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const int k = (int)(idx % Kp);
            const int rs = (int)((idx / Kp) % RS);
            const int c = (int)(idx / ((long long)Kp * RS));
            float v = 0.f;
            if (k < K && c < Cin) {
                v = w[((long long)k * Cin + c) * RS + rs];
                if (scale) v *= scale[k];
            }
            w_chwk[idx] = from_f32<T>(v);
        }
    }
}

// The same for bf16 through an LDS tile (round 3): the kernel above reads w with a stride of R*S floats between neighbouring lanes (a wave
// touches 9x the bytes it uses for a 3x3 filter) -- 21 us per layer and 0.3 ms per step on the segmentation decoder, whose 50 M trainable
// weights are re-staged every step.  Workgroup = 32 filters x 64 input channels x all taps: contiguous reads (64 * R*S floats per filter),
// contiguous 128-byte / 64-byte runs on the way out.
constexpr int kWpTK = 32, kWpTC = 64;
template <int RS>
__global__ __launch_bounds__(256) void weight_prep_tiled_kernel(const float* __restrict__ w, const float* __restrict__ scale, int K, int Cin,
                                                                int Cp, int Kp, bf16_t* __restrict__ w_khwc, bf16_t* __restrict__ w_chwk) {
    extern __shared__ bf16_t wt[];                       // [kWpTK][RS][kWpTC + 2]
    const int k0 = blockIdx.y * kWpTK, c0 = blockIdx.x * kWpTC;
    constexpr int pitch = kWpTC + 2;
    const int cw = (Cin - c0) < kWpTC ? (Cin - c0) : kWpTC;          // may be <= 0 in the padding columns of Cp
    const int run = cw > 0 ? cw * RS : 0;
    constexpr int ROW = kWpTC * RS, TOTAL = kWpTK * ROW;             // flat walk over the tile: 8 independent loads in flight per thread
    static_assert(TOTAL % (256 * 8) == 0, "tile walk");
    for (int e0 = threadIdx.x; e0 < TOTAL; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            const int kk = e / ROW, i = e - kk * ROW;
            const int k = k0 + kk;
            v[u] = (k < K && i < run) ? w[((long long)k * Cin + c0) * RS + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            const int kk = e / ROW, i = e - kk * ROW;
            const int cl = i / RS, rs = i - cl * RS;
            const int k = k0 + kk;
            const float sc = (scale && k < K) ? scale[k] : 1.f;
            wt[(kk * RS + rs) * pitch + cl] = from_f32<bf16_t>(v[u] * sc);
        }
    }
    __syncthreads();
    if (w_khwc) {
        // [k][rs][c0 .. c0+63]: one 128-byte run per (k, rs)
        for (int i = threadIdx.x; i < kWpTK * RS * (kWpTC / 2); i += 256) {
            const int cp = i % (kWpTC / 2), krs = i / (kWpTC / 2);
            const int kk = krs / RS, rs = krs - kk * RS;
            const int k = k0 + kk, c = c0 + 2 * cp;
            if (k < Kp && c < Cp)
                *reinterpret_cast<unsigned*>(w_khwc + ((long long)k * RS + rs) * Cp + c) = *reinterpret_cast<const unsigned*>(&wt[(kk * RS + rs) * pitch + 2 * cp]);
        }
    }
    if (w_chwk) {
        // [c][rs][k0 .. k0+31]: one 64-byte run per (c, rs)
        for (int i = threadIdx.x; i < kWpTC * RS * (kWpTK / 2); i += 256) {
            const int kp = i % (kWpTK / 2), crs = i / (kWpTK / 2);
            const int cl = crs / RS, rs = crs - cl * RS;
            const int c = c0 + cl, k = k0 + 2 * kp;
            if (c < Cp && k < Kp) {
                union { bf16_t h[2]; unsigned u; } cv;
                cv.h[0] = wt[((2 * kp) * RS + rs) * pitch + cl];
                cv.h[1] = wt[((2 * kp + 1) * RS + rs) * pitch + cl];
                *reinterpret_cast<unsigned*>(w_chwk + ((long long)c * RS + rs) * Kp + k) = cv.u;
            }
        }
    }
}

// ---- fused staging for an eval-mode Conv+BN: BN fold + both weight layouts in ONE launch.  Every thread recomputes
// scale[k] = gamma[k]/sqrt(var[k]+eps) for its element (cheap), workgroup 0 also writes scale/shift/rstd.
template <typename T>
__global__ void stage_conv_bn_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                     const float* __restrict__ conv_bias, int K, int Cin, int R, int S, int Cp, int Kp,
                                     T* __restrict__ w_khwc, T* __restrict__ w_chwk, float* __restrict__ scale,
                                     float* __restrict__ shift, float* __restrict__ rstd) {
    const int RS = R * S;
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
            float r = 0.f, sc = 0.f, sh = 0.f;
            if (k < K) {
                r = 1.0f / sqrtf(var[k] + eps);
                sc = (gamma ? gamma[k] : 1.f) * r;
                sh = (beta ? beta[k] : 0.f) + ((conv_bias ? conv_bias[k] : 0.f) - mean[k]) * sc;
            }
            scale[k] = sc; shift[k] = sh; rstd[k] = r;
        }
    }
    const long long t1 = w_khwc ? (long long)Kp * RS * Cp : 0;
    const long long t2 = w_chwk ? (long long)Cp * RS * Kp : 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < t1 + t2; idx += (long long)gridDim.x * blockDim.x) {
        int k, c, rs;
        const bool first = idx < t1;
        if (first) {
            c = (int)(idx % Cp); rs = (int)((idx / Cp) % RS); k = (int)(idx / ((long long)Cp * RS));
        } else {
            const long long j = idx - t1;
            k = (int)(j % Kp); rs = (int)((j / Kp) % RS); c = (int)(j / ((long long)Kp * RS));
        }
        float v = 0.f;
        if (k < K && c < Cin) v = w[((long long)k * Cin + c) * RS + rs] * ((gamma ? gamma[k] : 1.f) * (1.0f / sqrtf(var[k] + eps)));
        if (first) w_khwc[idx] = from_f32<T>(v);
        else w_chwk[idx - t1] = from_f32<T>(v);
    }
}

// ---- every layer of a network in one launch: workgroup -> layer by binary search over the block0 prefix table
constexpr int kStageElemsPerBlock = 2048;      // 256 threads x 8 staged elements

template <typename T>
__global__ __launch_bounds__(256) void stage_conv_bn_multi_kernel(const CsStageDesc* __restrict__ desc, int n) {
    __shared__ int which_s;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (desc[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        which_s = lo;
    }
    __syncthreads();
    const int which = which_s;
    const CsStageDesc d = desc[which];
    const int nb = (which + 1 < n ? desc[which + 1].block0 : (int)gridDim.x) - d.block0;
    const int b = (int)blockIdx.x - d.block0;
    const int RS = d.R * d.S;
    if (b == 0) {
        for (int k = threadIdx.x; k < d.Kp; k += blockDim.x) {
            float r = 0.f, sc = 0.f, sh = 0.f;
            if (k < d.K) {
                r = d.var ? 1.0f / sqrtf(d.var[k] + d.eps) : 1.f;          // (var == NULL: a layer without a folded BatchNorm)
                sc = (d.gamma ? d.gamma[k] : 1.f) * r;
                sh = (d.beta ? d.beta[k] : 0.f) + ((d.conv_bias ? d.conv_bias[k] : 0.f) - (d.mean ? d.mean[k] : 0.f)) * sc;
            }
            d.scale[k] = sc; d.shift[k] = sh; d.rstd[k] = r;
        }
    }
    T* w_khwc = reinterpret_cast<T*>(d.w_khwc);
    T* w_chwk = reinterpret_cast<T*>(d.w_chwk);
    // ---- packed operands of unpadded layers: through an LDS tile.  The element-wise path below reads the OIHW weights with a stride
    // of R*S floats (4 of every 32 fetched bytes used on the 3x3 layers -- it ran at 1.3 TB/s for the whole network); here a tile of
    // 16 output channels x 64 input channels x all taps is read as 16 contiguous runs, and both packed operands leave as 16-byte
    // pieces that are contiguous over 16 / 32 lanes.  Same arithmetic, bit-identical results.
    constexpr int kTileRS = 9;
    __shared__ float tile_s[16][64 * kTileRS + 1];
    __shared__ float scl_s[16];
    const bool tiled = sizeof(T) == 2 && RS <= kTileRS && d.Cp == d.Cin && d.Kp == d.K && d.Cin % 64 == 0 && d.K % 64 == 0;
    if (tiled) {
        const int n_ct = d.Cin / 64, n_tiles = (d.K / 16) * n_ct;
        const int run = 64 * RS;
        for (int tile = b; tile < n_tiles; tile += nb) {
            const int k0 = (tile / n_ct) * 16, c0 = (tile % n_ct) * 64;
            __syncthreads();
            // eight loads in flight per thread, then their LDS stores (written load, store, load, store ... hipcc waits for every load before
            // its store: 36 memory round trips per 3x3 tile and thread; the launch ran at 2.2 TB/s)
            for (int idx0 = threadIdx.x; idx0 < 16 * run; idx0 += 8 * 256) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = idx0 + 256 * u;
                    const int r = idx / run, o = idx - r * run;
                    v[u] = idx < 16 * run ? d.w[((long long)(k0 + r) * d.Cin + c0) * RS + o] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = idx0 + 256 * u;
                    const int r = idx / run, o = idx - r * run;
                    if (idx < 16 * run) tile_s[r][o] = v[u];
                }
            }
            if (threadIdx.x < 16) {
                const int k = k0 + threadIdx.x;
                scl_s[threadIdx.x] = (d.gamma ? d.gamma[k] : 1.f) * (d.var ? 1.0f / sqrtf(d.var[k] + d.eps) : 1.f);
            }
            __syncthreads();
            if (w_khwc && !d.fwd_packed) {
                // plain forward operand [k][tap][c]: 8 consecutive input channels per store, 128 contiguous bytes per (k, tap)
                for (int idx = threadIdx.x; idx < RS * 128; idx += 256) {
                    const int c8 = idx & 7, r16 = (idx >> 3) & 15, t = idx >> 7;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tile_s[r16][(c8 * 8 + e) * RS + t] * scl_s[r16];
                    store8<T>(w_khwc + ((long long)(k0 + r16) * RS + t) * d.Cp + c0 + c8 * 8, v);
                }
            } else if (w_khwc) {
                // rows = output channels: [32-row tile][64-column chunk][tap][16-deep step][lane][8]
                const int ncc = d.Cp / 64, cc = c0 / 64;
                for (int idx = threadIdx.x; idx < RS * 128; idx += 256) {
                    const int r16 = idx & 15, lh = (idx >> 4) & 1, k16 = (idx >> 5) & 3, t = idx >> 7;
                    const int row = k0 + r16;
                    const long long j = ((((long long)(row >> 5) * ncc + cc) * RS + t) * 4 + k16) * 512 + (lh * 32 + (row & 31)) * 8;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tile_s[r16][(k16 * 16 + 8 * lh + e) * RS + t] * scl_s[r16];
                    store8<T>(w_khwc + j, v);
                }
            }
            if (w_chwk && !d.bwd_packed) {
                // plain data-gradient operand [c][tap][k]: 8 consecutive output channels per store
                for (int idx = threadIdx.x; idx < RS * 128; idx += 256) {
                    const int kg = idx & 1, c = (idx >> 1) & 63, t = idx >> 7;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tile_s[kg * 8 + e][c * RS + t] * scl_s[kg * 8 + e];
                    store8<T>(w_chwk + ((long long)(c0 + c) * RS + t) * d.Kp + k0 + kg * 8, v);
                }
            } else if (w_chwk) {
                // rows = input channels, columns = output channels, taps mirrored
                const int ncc = d.Kp / 64, cc = k0 / 64, k16 = (k0 & 63) >> 4;
                for (int idx = threadIdx.x; idx < RS * 128; idx += 256) {
                    const int c32 = idx & 31, lh = (idx >> 5) & 1, rt2 = (idx >> 6) & 1, tp = idx >> 7;
                    const int c = c0 + rt2 * 32 + c32, ts = RS - 1 - tp;
                    const long long j = ((((long long)(c >> 5) * ncc + cc) * RS + tp) * 4 + k16) * 512 + (lh * 32 + c32) * 8;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tile_s[lh * 8 + e][(rt2 * 32 + c32) * RS + ts] * scl_s[lh * 8 + e];
                    store8<T>(w_chwk + j, v);
                }
            }
        }
        return;
    }
    const long long t1 = w_khwc ? (long long)d.Kp * RS * d.Cp : 0;
    const long long t2 = w_chwk ? (long long)d.Cp * RS * d.Kp : 0;
    for (long long idx = (long long)b * blockDim.x + threadIdx.x; idx < t1 + t2; idx += (long long)nb * blockDim.x) {
        int k, c, rs;
        const bool first = idx < t1;
        const long long j = first ? idx : idx - t1;
        if (first ? d.fwd_packed : d.bwd_packed) {
            // MFMA-fragment order of csrc/conv_v2.hip: [32-row tile][64-column chunk][tap][16-deep step][lane][8]; rows = destination
            // channels (k forward, c data gradient), columns = contraction channels; the data-gradient form mirrors the taps
            const int e = (int)(j & 7), lane = (int)((j >> 3) & 63), k16 = (int)((j >> 9) & 3);
            long long u = j >> 11;
            const int t = (int)(u % RS); u /= RS;
            const int ncc = (first ? d.Cp : d.Kp) / 64;
            const int cc = (int)(u % ncc);
            const int rt = (int)(u / ncc);
            const int row = rt * 32 + (lane & 31), col = cc * 64 + k16 * 16 + 8 * (lane >> 5) + e;
            if (first) { k = row; c = col; rs = t; }
            else { c = row; k = col; rs = RS - 1 - t; }
        } else if (first) {
            c = (int)(j % d.Cp); rs = (int)((j / d.Cp) % RS); k = (int)(j / ((long long)d.Cp * RS));
        } else {
            k = (int)(j % d.Kp); rs = (int)((j / d.Kp) % RS); c = (int)(j / ((long long)d.Kp * RS));
        }
        float v = 0.f;
        if (k < d.K && c < d.Cin) v = d.w[((long long)k * d.Cin + c) * RS + rs] * ((d.gamma ? d.gamma[k] : 1.f) * (d.var ? 1.0f / sqrtf(d.var[k] + d.eps) : 1.f));
        if (first) w_khwc[j] = from_f32<T>(v);
        else w_chwk[j] = from_f32<T>(v);
    }
}

// ---- grouped weights: fp32 [K][Cg][R][S] (x scale[k]) -> slab-dense T [K][R][S][64] and T [C][R][S][64]
// (K == C, groups of Cg channels, 64-channel slabs; entries outside a channel's own group are zero)
template <typename T>
__global__ void weight_prep_grouped_kernel(const float* __restrict__ w, const float* __restrict__ scale, int K, int Cg, int R, int S,
                                           T* __restrict__ w_khwc, T* __restrict__ w_chwk) {
    const int RS = R * S;
    const long long total = (long long)K * RS * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(idx % 64);                 // slab-local channel on the contracted side
        const int rs = (int)((idx / 64) % RS);
        const int row = (int)(idx / (64LL * RS));      // k for w_khwc, c for w_chwk
        const int slab = row / 64;
        const int row_l = row % 64;
        const bool same = (row_l / Cg) == (l / Cg);
        if (w_khwc) {
            float v = 0.f;
            if (same) { v = w[((long long)row * Cg + (l % Cg)) * RS + rs]; if (scale) v *= scale[row]; }
            w_khwc[idx] = from_f32<T>(v);
        }
        if (w_chwk) {
            float v = 0.f;
            const int k = slab * 64 + l;
            if (same) { v = w[((long long)k * Cg + (row_l % Cg)) * RS + rs]; if (scale) v *= scale[k]; }
            w_chwk[idx] = from_f32<T>(v);
        }
    }
}

// ---- wgrad finalize.  Pass A: grid (ceil(Cin*R*S / 256), K): each thread folds the nsplit partial slabs of ONE weight
// (fixed order -> bitwise reproducible; consecutive threads read consecutive slab addresses), writes dw in torch layout and
// contributes w*dw_raw to a per-channel dot product (one atomic per workgroup).  Pass B (K threads): BN-eval parameter
// gradients / conv-bias gradient from the dot products and the column sums.
__global__ __launch_bounds__(256) void wgrad_finalize_a_kernel(const float* __restrict__ dw_khwc, const float* __restrict__ w,
                                                               const float* scale, int Cin, int RS, int Cp, int Cg_slab,
                                                               float* __restrict__ dw, float* __restrict__ dot, int accumulate,
                                                               int nsplit, long long slab_stride) {
    // Cg_slab == 0: dense weights, raw row layout [rs][Cp];  Cg_slab > 0: grouped slab-dense, raw row [rs][64], Cin == Cg
    const int k = blockIdx.y;
    const int per = Cin * RS;
    // threads walk the SLAB's own [rs][c] order: coalesced reads of the nsplit partials (the bulk of the traffic);
    // the single write per weight lands RS-strided in torch's [c][rs] order
    float contrib = 0.f;
    // grid-stride over the row: with a dot product wanted the launcher starts ONE workgroup per row, which then owns dot[k] (several
    // workgroups per row added their shares with a float atomic: the BN gradient of an eval-mode layer did not repeat bit for bit)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const int rs = i / Cin, c = i - rs * Cin;
        const int j = c * RS + rs;
        const long long src = Cg_slab ? ((long long)k * RS + rs) * 64 + ((k % 64) / Cg_slab) * Cg_slab + c
                                      : ((long long)k * RS + rs) * Cp + c;
        // small layers are split many ways (up to M/64 slices): keep 8 partial loads in flight instead of one
        float a8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] = 0.f;
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += dw_khwc[(z + u) * slab_stride + src];
        }
        for (; z < nsplit; ++z) a8[0] += dw_khwc[z * slab_stride + src];
        const float raw = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        const long long o = (long long)k * per + j;
        if (dot) contrib += w[o] * raw;
        const float val = (scale ? scale[k] : 1.f) * raw;
        dw[o] = accumulate ? dw[o] + val : val;
    }
    if (dot) {
        __shared__ float red[4];
        contrib = wave_sum(contrib);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
        __syncthreads();
        if (threadIdx.x == 0) dot[k] += (red[0] + red[1]) + (red[2] + red[3]);       // gridDim.x == 1: the only writer
    }
}

__global__ void wgrad_finalize_b_kernel(const float* __restrict__ dot, const float* scale, const float* rstd, const float* mean,
                                        const float* gsum, float* dbias, float* dgamma, float* dbeta, int K, int accumulate) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float gs = gsum ? gsum[k] : 0.f;
    if (dgamma) {
        const float v = rstd[k] * (dot[k] - mean[k] * gs);
        dgamma[k] = accumulate ? dgamma[k] + v : v;
    }
    if (dbeta) dbeta[k] = accumulate ? dbeta[k] + gs : gs;
    if (dbias) {
        const float v = (scale ? scale[k] : 1.f) * gs;
        dbias[k] = accumulate ? dbias[k] + v : v;
    }
}

// ---- column sums: out[c] += sum_m g[m][c].  Thread owns 8 channels over a strided row set, the
// workgroup folds its threads through LDS and issues ONE atomic per channel. ---------------------
// PARTIAL: workgroup b stores its sums to out[b * 2*C + c] (the partial-row layout of cs_conv2d_dgrad) instead of adding them
// to out[c]: ~512 atomics per address serialise at the memory side (~50 us on their own, whatever the streaming rate).
template <typename T, bool PARTIAL>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, long long M, int C, float* __restrict__ out,
                                                     int rows_per_block) {
    const int CG = C / 8;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    __shared__ float fold[256][8];
    for (int cg0 = 0; cg0 < CG; cg0 += 256) {
        const int width = (CG - cg0) < 256 ? (CG - cg0) : 256;
        const int rpar = 256 / width;
        const int cg = cg0 + (int)(threadIdx.x % width);
        const int rr = threadIdx.x / width;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (rr < rpar) {
            // eight independent 16-byte loads in flight per thread (one dependent load per iteration streamed at 1.4 TB/s)
            long long r = r0 + rr;
            for (; r + 7LL * rpar < r1; r += 8LL * rpar) {
                float v[8][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) load8<T>(g + (r + (long long)u * rpar) * C + cg * 8, v[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] += ((v[0][e] + v[1][e]) + (v[2][e] + v[3][e])) + ((v[4][e] + v[5][e]) + (v[6][e] + v[7][e]));
            }
            for (; r < r1; r += rpar) {
                float v[8];
                load8<T>(g + r * C + cg * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) fold[threadIdx.x][e] = acc[e];
        __syncthreads();
        // one thread per channel column (8 * width of them), each folds rpar values in row order (the same sums as when `width` lanes
        // walked 8 columns each -- 8 lanes x 256 dependent LDS reads at the end of every workgroup of a 64-channel tensor)
        for (int col = threadIdx.x; col < width * 8; col += 256) {
            const int cgl = col >> 3, e = col & 7;
            float t = 0.f;
            for (int r = 0; r < rpar; ++r) t += fold[r * width + cgl][e];
            const int ch = (cg0 + cgl) * 8 + e;
            if constexpr (PARTIAL) out[(long long)blockIdx.x * 2 * C + ch] = t;
            else atomicAdd(out + ch, t);
        }
    }
}

}  // namespace

static inline int grid_for(long long total, int block) {
    long long b = (total + block - 1) / block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int cs_nchw_to_nhwc(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, void* stream) {
    CS_CHECK_ARG(x && y, "nchw_to_nhwc: NULL tensor");
    CS_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nchw_to_nhwc: bad extents");
    CS_CHECK_ARG(Cp % 8 == 0 || (dtype == CS_F32 && Cp == 4), "nchw_to_nhwc: Cp must be a multiple of 8 (or 4 for fp32)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int groups = Cp / 8 > 0 ? Cp / 8 : 1;
    const long long total = (long long)N * H * W * groups;
    if (dtype == CS_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, x, (float*)y, N, C, H * W, Cp);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, x, (bf16_t*)y, N, C, H * W, Cp);
    else
        CS_CHECK_ARG(false, "nchw_to_nhwc: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_nhwc_to_nchw(const void* y, int dtype, float* x, int N, int C, int H, int W, int Cp, void* stream) {
    CS_CHECK_ARG(x && y, "nhwc_to_nchw: NULL tensor");
    CS_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nhwc_to_nchw: bad extents");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * H * W * ((C + 7) / 8);
    if (dtype == CS_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const float*)y, x, N, C, H * W, Cp);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)y, x, N, C, H * W, Cp);
    else
        CS_CHECK_ARG(false, "nhwc_to_nchw: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          const float* conv_bias, float* scale, float* shift, float* rstd, int C, void* stream) {
    CS_CHECK_ARG(mean && var && C > 0, "bn_fold: NULL statistics");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, var, eps, conv_bias, scale, shift, rstd, C);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_weight_prep(const float* w, const float* scale, int dtype, int K, int Cin, int R, int S, int Cp, int Kp,
                              void* w_khwc, void* w_chwk, void* stream) {
    CS_CHECK_ARG(w && (w_khwc || w_chwk), "weight_prep: NULL tensor");
    CS_CHECK_ARG(K > 0 && Cin > 0 && R > 0 && S > 0 && Cp >= Cin && Kp >= K, "weight_prep: bad extents");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long t1 = (long long)Kp * R * S * Cp, t2 = (long long)Cp * R * S * Kp;
    const long long total = t1 > t2 ? t1 : t2;
    const size_t tile_lds = (size_t)kWpTK * R * S * (kWpTC + 2) * sizeof(bf16_t);
    static const int untiled = cs_env_int_("CELLSEG_WPREP_UNTILED", 0);      // A/B experiments only
    if (!untiled && dtype == CS_BF16 && (R * S == 9 || R * S == 1) && Cp % 2 == 0 && Kp % 2 == 0 && total >= (1 << 21)) {   // (smaller tensors: too few tiles to fill the chip, 10 vs 20 us)
        const dim3 tgrid((unsigned)((Cp + kWpTC - 1) / kWpTC), (unsigned)((Kp + kWpTK - 1) / kWpTK));
        if (R * S == 9)
            hipLaunchKernelGGL(weight_prep_tiled_kernel<9>, tgrid, dim3(256), tile_lds, st, w, scale, K, Cin, Cp, Kp, (bf16_t*)w_khwc, (bf16_t*)w_chwk);
        else
            hipLaunchKernelGGL(weight_prep_tiled_kernel<1>, tgrid, dim3(256), tile_lds, st, w, scale, K, Cin, Cp, Kp, (bf16_t*)w_khwc, (bf16_t*)w_chwk);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (dtype == CS_F32)
        hipLaunchKernelGGL(weight_prep_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, w, scale, K, Cin, R, S, Cp, Kp,
                           (float*)w_khwc, (float*)w_chwk);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, w, scale, K, Cin, R, S, Cp, Kp,
                           (bf16_t*)w_khwc, (bf16_t*)w_chwk);
    else
        CS_CHECK_ARG(false, "weight_prep: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// Batched finalize for n <= 8 layers of identical geometry: ONE launch, one workgroup per (output channel k, layer).  The workgroup
//   * folds the deferred column sums of its channel (gsum given as per-workgroup partial rows: grows[i] > 0), or reads the vector;
//   * sums the split-K slabs of row k, writes dw[k][c][rs] (R*S > 1: the slab layout [rs][c] is turned through LDS so both the reads
//     and the writes are contiguous) and accumulates dot[k] = sum w * raw in registers;
//   * writes dgamma[k] = rstd * (dot - mean * gsum), dbeta[k] = gsum.
// Round 1 did this in three launches per group (fold, finalize_a with one atomicAdd per workgroup into a zeroed dot[], finalize_b):
// 63 launches and 0.64 ms per ResNet-50 step.  Tables travel BY VALUE (static indices only: a runtime index would spill them).
struct FinalizeTables {
    const float* raw[8]; const float* w[8]; const float* scale[8]; const float* rstd[8]; const float* mean[8];
    const float* gsum[8]; float* dw[8]; float* dgamma[8]; float* dbeta[8];
    int grows[8];
    int has_scale, want_bn;
};

template <int NT> __device__ __forceinline__ float block_sum(float v, float* red /* [NT / 64] */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

constexpr int kFinCT = 128;
// NT threads per workgroup: 1024 when the launch has fewer than 2048 rows (K x layers) -- a 64-channel layer would otherwise occupy
// 64 workgroups of 4 waves on a 256-CU chip (33.7 us for layer1's three 3x3 convolutions).
template <int NT>
__global__ __launch_bounds__(NT) void wgrad_finalize_fused_kernel(FinalizeTables t, int Cin, int RS, int Cp, int nsplit, long long slab_stride,
                                                                  int gstride) {
    extern __shared__ float tile[];          // R*S > 1: [RS][kFinCT + 1];  R*S == 1: [NT]
    __shared__ float red[NT / 64];
    const int k = blockIdx.x, z = blockIdx.y;
    const float* raw_s = nullptr; const float* w_s = nullptr; const float* sc_s = nullptr; const float* rstd_s = nullptr;
    const float* mean_s = nullptr; const float* gsum_s = nullptr; float* dw_s = nullptr; float* dg_s = nullptr; float* db_s = nullptr;
    int grows = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (z == i) {
            raw_s = t.raw[i]; w_s = t.w[i]; sc_s = t.scale[i]; rstd_s = t.rstd[i]; mean_s = t.mean[i]; gsum_s = t.gsum[i];
            dw_s = t.dw[i]; dg_s = t.dgamma[i]; db_s = t.dbeta[i]; grows = t.grows[i];
        }
    const float* __restrict__ raw_p = raw_s;
    const float* __restrict__ w = w_s;
    float* __restrict__ dw = dw_s;
    const float sc = t.has_scale ? sc_s[k] : 1.f;
    const int want_bn = t.want_bn;

    float gs = 0.f;
    if (want_bn) {
        if (grows > 0) {
            float a = 0.f;
            for (int r = threadIdx.x; r < grows; r += NT) a += gsum_s[(long long)r * gstride + k];
            gs = block_sum<NT>(a, red);
        } else {
            gs = gsum_s[k];
        }
    }

    float contrib = 0.f;
    if (RS == 1) {
        // CT channel lanes x SL split lanes (few channels, many slabs: 64 x 64 layers are split ~500 ways)
        int CT = NT;
        while (CT > 1 && (CT >> 1) >= Cin) CT >>= 1;
        const int SL = NT / CT;
        const int cl = threadIdx.x % CT, sl = threadIdx.x / CT;
        for (int c0 = 0; c0 < Cin; c0 += CT) {
            const int c = c0 + cl;
            float raw = 0.f;
            if (c < Cin) {
                const float* src = raw_p + (long long)k * Cp + c;
                // eight slabs in flight per thread, the tail predicated instead of walked one load at a time (6 splits were 4 + 1 + 1:
                // three memory round trips)
                for (int s = sl; s < nsplit; s += 8 * SL) {
                    float a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = (s + u * SL < nsplit) ? src[(long long)(s + u * SL) * slab_stride] : 0.f;
                    raw += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
                }
            }
            if (SL > 1) {
                __syncthreads();
                tile[threadIdx.x] = raw;
                __syncthreads();
                if (sl == 0)
                    for (int j = 1; j < SL; ++j) raw += tile[j * CT + cl];
            }
            if (sl == 0 && c < Cin) {
                const long long o = (long long)k * Cin + c;
                if (want_bn) contrib += w[o] * raw;
                dw[o] = sc * raw;
            }
        }
    } else {
        for (int c0 = 0; c0 < Cin; c0 += kFinCT) {
            const int cw = (Cin - c0) < kFinCT ? (Cin - c0) : kFinCT;       // channels of this tile
            __syncthreads();
            for (int idx = threadIdx.x; idx < RS * kFinCT; idx += NT) {
                const int rs = idx / kFinCT, cl = idx - rs * kFinCT;
                float raw = 0.f;
                if (cl < cw) {
                    const float* src = raw_p + ((long long)k * RS + rs) * Cp + c0 + cl;
                    for (int s = 0; s < nsplit; s += 8) {
                        float a[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) a[u] = (s + u < nsplit) ? src[(long long)(s + u) * slab_stride] : 0.f;
                        raw += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
                    }
                }
                tile[rs * (kFinCT + 1) + cl] = raw;
            }
            __syncthreads();
            const long long obase = ((long long)k * Cin + c0) * RS;
            for (int idx = threadIdx.x; idx < cw * RS; idx += NT) {
                const int cl = idx / RS, rs = idx - cl * RS;
                const float raw = tile[rs * (kFinCT + 1) + cl];
                if (want_bn) contrib += w[obase + idx] * raw;
                dw[obase + idx] = sc * raw;
            }
        }
    }
    if (want_bn) {
        const float dot = block_sum<NT>(contrib, red);
        if (threadIdx.x == 0) {
            dg_s[k] = rstd_s[k] * (dot - mean_s[k] * gs);
            db_s[k] = gs;
        }
    }
}

static int finalize_common(const float* raw, int nsplit, long long slab_stride, const float* w, const float* scale, const float* rstd,
                           const float* mean, const float* gsum, int K, int Cin, int RS, int Cp, int Cg_slab, float* dw, float* dbias,
                           float* dgamma, float* dbeta, float* dot_ws, int accumulate, hipStream_t st) {
    float* dot = dgamma ? dot_ws : nullptr;      // zeroed by the caller
    const int per = Cin * RS;
    hipLaunchKernelGGL(wgrad_finalize_a_kernel, dim3(dot ? 1 : (per + 255) / 256, K), dim3(256), 0, st, raw, w, scale, Cin, RS, Cp, Cg_slab, dw, dot,
                       accumulate, nsplit, slab_stride);
    CS_LAUNCH_CHECK();
    if (dgamma || dbeta || dbias) {
        hipLaunchKernelGGL(wgrad_finalize_b_kernel, dim3((K + 255) / 256), dim3(256), 0, st, dot, scale, rstd, mean, gsum, dbias, dgamma,
                           dbeta, K, accumulate);
        CS_LAUNCH_CHECK();
    }
    return CS_OK;
}

extern "C" int cs_wgrad_finalize(const float* dw_khwc, int nsplit, int Kp, const float* w, const float* scale, const float* rstd,
                                 const float* mean, const float* gsum, int K, int Cin, int R, int S, int Cp, float* dw,
                                 float* dbias, float* dgamma, float* dbeta, float* dot_ws, int accumulate, void* stream) {
    CS_CHECK_ARG(nsplit >= 1 && Kp >= K, "wgrad_finalize: bad nsplit / Kp");
    CS_CHECK_ARG(dw_khwc && dw, "wgrad_finalize: NULL tensor");
    CS_CHECK_ARG(K > 0 && Cin > 0 && R > 0 && S > 0 && Cp >= Cin, "wgrad_finalize: bad extents");
    CS_CHECK_ARG(!dgamma || (w && rstd && mean && gsum && dot_ws), "wgrad_finalize: dgamma needs w, rstd, mean, gsum, dot_ws[K]");
    CS_CHECK_ARG(!(dbeta || dbias) || gsum, "wgrad_finalize: dbeta/dbias need gsum");
    return finalize_common(dw_khwc, nsplit, (long long)Kp * R * S * Cp, w, scale, rstd, mean, gsum, K, Cin, R * S, Cp, 0, dw, dbias, dgamma,
                           dbeta, dot_ws, accumulate, reinterpret_cast<hipStream_t>(stream));
}

static int colsum_rows_per_block(long long M) {
    long long rows = (M + 511) / 512;
    return (int)(rows < 64 ? 64 : rows);
}

// Bit plane of a bf16 NHWC tensor x[M][C] (C a multiple of 32) in the CHANNEL-BLOCK-MAJOR layout of the convolution epilogues: dword
// (c / 32) * M + m, bit c % 32 = (x[m][c] > 0) -- the ReLU-mask plane of the data-gradient epilogues for tensors no convolution epilogue
// produced (a train-mode BN + ReLU output, a concatenation of two such tensors).  One lane = 32 channels of one pixel in, one dword out;
// consecutive lanes take consecutive PIXELS of one channel block: contiguous 256-byte stores, 64-byte reads C * 2 bytes apart.
__global__ __launch_bounds__(256) void positive_bits_kernel(const uint4* __restrict__ x, unsigned* __restrict__ bits, long long M, int CB) {
    const long long n32 = M * CB;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n32; i += (long long)gridDim.x * 256) {
        const long long blk = i / M, m = i - blk * M;
        const uint4* src = x + (m * CB + blk) * 4;
        unsigned out = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = src[q];
            const unsigned wds[4] = {v.x, v.y, v.z, v.w};
            unsigned b = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = wds[e] & 0xffffu, hi = wds[e] >> 16;
                // > 0: sign clear and magnitude non-zero (NaN patterns count as positive exactly like `x > 0` does not -- a NaN
                // never reaches here: the tensors are ReLU outputs)
                b |= ((lo != 0u && !(lo & 0x8000u)) ? 1u : 0u) << (2 * e);
                b |= ((hi != 0u && !(hi & 0x8000u)) ? 1u : 0u) << (2 * e + 1);
            }
            out |= b << (8 * q);
        }
        bits[i] = out;
    }
}

extern "C" int cs_positive_bits(const void* x, int dtype, long long n_pixels, int C, uint8_t* bits, void* stream) {
    CS_CHECK_ARG(x && bits && n_pixels > 0 && C > 0 && C % 32 == 0, "positive_bits: need pixels > 0 and a channel count that is a positive multiple of 32");
    CS_CHECK_ARG(dtype == CS_BF16, "positive_bits: bf16 tensors only (the bit planes belong to the packed bf16 kernels)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long n32 = n_pixels * (C / 32);
    long long blocks = (n32 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(positive_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)x, (unsigned*)bits, n_pixels, C / 32);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_colsum_partial_rows(long long M) {
    if (M <= 0) return 0;
    const int rows = colsum_rows_per_block(M);
    return (int)((M + rows - 1) / rows);
}

extern "C" int cs_colsum_partial(const void* g, int dtype, long long M, int C, float* partial, void* stream) {
    CS_CHECK_ARG(g && partial, "colsum_partial: NULL tensor");
    CS_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "colsum_partial: C must be a positive multiple of 8");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rows = colsum_rows_per_block(M);
    const int blocks = (int)((M + rows - 1) / rows);
    if (dtype == CS_F32) hipLaunchKernelGGL((colsum_kernel<float, true>), dim3(blocks), dim3(256), 0, st, (const float*)g, M, C, partial, rows);
    else if (dtype == CS_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, M, C, partial, rows);
    else CS_CHECK_ARG(false, "colsum_partial: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_colsum(const void* g, int dtype, long long M, int C, float* out, void* stream) {
    CS_CHECK_ARG(g && out, "colsum: NULL tensor");
    CS_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "colsum: C must be a positive multiple of 8");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // ~512 workgroups (2 per CU), at least 64 rows each: <= 512 atomics per channel (1024 workgroups doubled the atomic
    // contention on the few dozen addresses and ran 50 % slower)
    long long rows = (M + 511) / 512;
    if (rows < 64) rows = 64;
    const int blocks = (int)((M + rows - 1) / rows);
    if (dtype == CS_F32)
        hipLaunchKernelGGL((colsum_kernel<float, false>), dim3(blocks), dim3(256), 0, st, (const float*)g, M, C, out, (int)rows);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL((colsum_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, M, C, out, (int)rows);
    else
        CS_CHECK_ARG(false, "colsum: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_weight_prep_grouped(const float* w, const float* scale, int dtype, int K, int Cg, int R, int S, void* w_khwc,
                                      void* w_chwk, void* stream) {
    CS_CHECK_ARG(w && (w_khwc || w_chwk), "weight_prep_grouped: NULL tensor");
    CS_CHECK_ARG(K > 0 && K % 64 == 0 && Cg > 0 && 64 % Cg == 0 && R > 0 && S > 0, "weight_prep_grouped: need K % 64 == 0 and Cg | 64");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)K * R * S * 64;
    if (dtype == CS_F32)
        hipLaunchKernelGGL(weight_prep_grouped_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, w, scale, K, Cg, R, S,
                           (float*)w_khwc, (float*)w_chwk);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(weight_prep_grouped_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, w, scale, K, Cg, R, S,
                           (bf16_t*)w_khwc, (bf16_t*)w_chwk);
    else
        CS_CHECK_ARG(false, "weight_prep_grouped: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_wgrad_finalize_grouped(const float* dw_slab, int nsplit, const float* w, const float* scale, const float* rstd,
                                         const float* mean, const float* gsum, int K, int Cg, int R, int S, float* dw, float* dgamma,
                                         float* dbeta, float* dot_ws, void* stream) {
    CS_CHECK_ARG(nsplit >= 1, "wgrad_finalize_grouped: bad nsplit");
    CS_CHECK_ARG(dw_slab && dw && K > 0 && Cg > 0 && 64 % Cg == 0, "wgrad_finalize_grouped: bad arguments");
    CS_CHECK_ARG(!dgamma || (w && rstd && mean && gsum && dot_ws), "wgrad_finalize_grouped: dgamma needs w, rstd, mean, gsum, dot_ws[K]");
    CS_CHECK_ARG(!dbeta || gsum, "wgrad_finalize_grouped: dbeta needs gsum");
    return finalize_common(dw_slab, nsplit, (long long)K * R * S * 64, w, scale, rstd, mean, gsum, K, Cg, R * S, 64, Cg, dw, nullptr, dgamma,
                           dbeta, dot_ws, 0, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int cs_stage_conv_bn(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                const float* conv_bias, int dtype, int K, int Cin, int R, int S, int Cp, int Kp, void* w_khwc,
                                void* w_chwk, float* scale, float* shift, float* rstd, void* stream) {
    CS_CHECK_ARG(w && mean && var && scale && shift && rstd && (w_khwc || w_chwk), "stage_conv_bn: NULL tensor");
    CS_CHECK_ARG(K > 0 && Cin > 0 && R > 0 && S > 0 && Cp >= Cin && Kp >= K, "stage_conv_bn: bad extents");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (w_khwc ? (long long)Kp * R * S * Cp : 0) + (w_chwk ? (long long)Cp * R * S * Kp : 0);
    if (dtype == CS_F32)
        hipLaunchKernelGGL(stage_conv_bn_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, w, gamma, beta, mean, var, eps,
                           conv_bias, K, Cin, R, S, Cp, Kp, (float*)w_khwc, (float*)w_chwk, scale, shift, rstd);
    else if (dtype == CS_BF16)
        hipLaunchKernelGGL(stage_conv_bn_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, w, gamma, beta, mean, var, eps,
                           conv_bias, K, Cin, R, S, Cp, Kp, (bf16_t*)w_khwc, (bf16_t*)w_chwk, scale, shift, rstd);
    else
        CS_CHECK_ARG(false, "stage_conv_bn: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_stage_conv_bn_blocks(int K, int Cin, int R, int S, int Cp, int Kp, int want_fwd, int want_bwd) {
    if (K <= 0 || Cin <= 0 || R <= 0 || S <= 0 || Cp < Cin || Kp < K) return 0;
    const long long total = (want_fwd ? (long long)Kp * R * S * Cp : 0) + (want_bwd ? (long long)Cp * R * S * Kp : 0);
    long long b = (total + kStageElemsPerBlock - 1) / kStageElemsPerBlock;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

extern "C" int cs_stage_conv_bn_multi(const CsStageDesc* desc, int n, int total_blocks, int dtype, void* stream) {
    CS_CHECK_ARG(desc && n >= 1 && total_blocks >= n, "stage_conv_bn_multi: need a device descriptor table and >= 1 workgroup per layer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CS_F32) hipLaunchKernelGGL(stage_conv_bn_multi_kernel<float>, dim3(total_blocks), dim3(256), 0, st, desc, n);
    else if (dtype == CS_BF16) hipLaunchKernelGGL(stage_conv_bn_multi_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, desc, n);
    else CS_CHECK_ARG(false, "stage_conv_bn_multi: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_wgrad_finalize_batched(const float* const* tables /* HOST: 9 consecutive tables of n pointers */, const int* gsum_rows,
                                         int gsum_stride, int n_items, int nsplit, int Kp, int K, int Cin, int R, int S, int Cp, int want_bn,
                                         void* stream) {
    CS_CHECK_ARG(tables && n_items >= 1 && n_items <= 8 && nsplit >= 1 && K > 0 && Cin > 0 && R > 0 && S > 0 && Cp >= Cin && Kp >= K,
                 "wgrad_finalize_batched: bad arguments (1..8 items, HOST pointer tables)");
    CS_CHECK_ARG(R * S <= 64, "wgrad_finalize_batched: filters of more than 64 taps are not served");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FinalizeTables t{};
    for (int i = 0; i < n_items; ++i) {
        t.raw[i] = tables[0 * n_items + i]; t.w[i] = tables[1 * n_items + i]; t.scale[i] = tables[2 * n_items + i];
        t.rstd[i] = tables[3 * n_items + i]; t.mean[i] = tables[4 * n_items + i]; t.gsum[i] = tables[5 * n_items + i];
        t.dw[i] = const_cast<float*>(tables[6 * n_items + i]); t.dgamma[i] = const_cast<float*>(tables[7 * n_items + i]);
        t.dbeta[i] = const_cast<float*>(tables[8 * n_items + i]);
        t.grows[i] = gsum_rows ? gsum_rows[i] : 0;
        CS_CHECK_ARG(t.raw[i] && t.dw[i], "wgrad_finalize_batched: NULL slab / dw");
        CS_CHECK_ARG(!want_bn || (t.w[i] && t.rstd[i] && t.mean[i] && t.gsum[i] && t.dgamma[i] && t.dbeta[i]),
                     "wgrad_finalize_batched: want_bn needs w, rstd, mean, gsum, dgamma, dbeta");
        CS_CHECK_ARG(t.grows[i] == 0 || gsum_stride >= K, "wgrad_finalize_batched: partial rows need their row stride");
    }
    t.has_scale = tables[2 * n_items] != nullptr;
    t.want_bn = want_bn;
    const bool wide = (long long)K * n_items < 2048;
    const size_t lds = R * S > 1 ? (size_t)R * S * (kFinCT + 1) * sizeof(float) : (wide ? 1024 : 256) * sizeof(float);
    if (wide)
        hipLaunchKernelGGL(wgrad_finalize_fused_kernel<1024>, dim3((unsigned)K, (unsigned)n_items), dim3(1024), lds, st, t, Cin, R * S, Cp, nsplit,
                           (long long)Kp * R * S * Cp, gsum_stride);
    else
        hipLaunchKernelGGL(wgrad_finalize_fused_kernel<256>, dim3((unsigned)K, (unsigned)n_items), dim3(256), lds, st, t, Cin, R * S, Cp, nsplit,
                           (long long)Kp * R * S * Cp, gsum_stride);
    CS_LAUNCH_CHECK();
    return CS_OK;
}
