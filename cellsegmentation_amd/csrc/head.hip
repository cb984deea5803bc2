// Heads and losses (tiny, fp32): Linear (model/resnet.py:126,137,140,150,360), CrossEntropy*gamma
// (train/train.py:34,80), softmax prob of class 1 (inference.py:24-27), MSE / weighted MSE
// (train/losses.py:5-29, metrics/metrics.py:23-33), Dice (train/losses.py:44-62,
// metrics/metrics.py:36-53).
#include "cs_common.h"

namespace {

// one wave per output element y[m][n]
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, float* __restrict__ pre,
                                                         int M, int N, int K, int act) {
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (long long)M * N) return;
    const int m = (int)(wave / N), n = (int)(wave % N);
    const float* xr = x + (long long)m * K;
    const float* wr = w + (long long)n * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += xr[k] * wr[k];
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc + (b ? b[n] : 0.f);
        if (pre) pre[(long long)m * N + n] = v;
        if (act == CS_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == CS_ACT_SILU) v = v / (1.f + expf(-v));
        else if (act == CS_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        y[(long long)m * N + n] = v;
    }
}

// SHORT contraction (K <= 64: the second Linear of a squeeze-excitation block, 6-58 inputs): one THREAD per output, k ascending with
// one fused multiply-add per step -- the summation order of the tiled kernel, so both give the same bits.  The 64 x 64 LDS tile
// spends ~5 us of fixed latencies (tile loads, transposed LDS stores, barrier, LDS reads) on a product of 64 x 24 x 576.
__global__ __launch_bounds__(256) void linear_fwd_shortk_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* __restrict__ y, float* __restrict__ pre,
                                                                int M, int N, int K, int act) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long long)M * N) return;
    const int m = (int)(o / N), n = (int)(o - (long long)m * N);
    const float* xr = x + (long long)m * K;
    const float* wr = w + (long long)n * K;
    const float bias = b ? b[n] : 0.f;
    float acc = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        float xv[8], wv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { xv[q] = xr[k + q]; wv[q] = wr[k + q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(xv[q], wv[q], acc);
    }
    for (; k < K; ++k) acc = fmaf(xr[k], wr[k], acc);
    float v = acc + bias;
    if (pre) pre[o] = v;
    if (act == CS_ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (act == CS_ACT_SILU) v = v / (1.f + expf(-v));
    else if (act == CS_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    y[o] = v;
}

// upstream gradient through the output activation; `y` is the stored OUTPUT for ReLU / sigmoid and the stored
// PRE-activation for SiLU
__device__ __forceinline__ float lin_act_grad(float g, float yv, int act) {
    if (act == CS_ACT_RELU) return yv > 0.f ? g : 0.f;
    if (act == CS_ACT_SIGMOID) return g * yv * (1.f - yv);
    if (act == CS_ACT_SILU) {
        const float sg = 1.f / (1.f + expf(-yv));
        return g * sg * (1.f + yv * (1.f - sg));
    }
    return g;
}

// dx[m][k] = sum_n g[m][n] w[n][k]: one wave per output, lanes stride over n (N reaches 2304 in the SE blocks: a
// thread-per-output loop was a 2304-long serial chain on a 24-workgroup grid)
__device__ __forceinline__ void linear_dx_body(unsigned block, const float* __restrict__ dy, const float* __restrict__ y,
                                               const float* __restrict__ w, float* __restrict__ dx, int M, int N, int K, int act) {
    const long long wave = ((long long)block * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (long long)M * K) return;
    const int m = (int)(wave / K), k = (int)(wave % K);
    float acc = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float g = lin_act_grad(dy[(long long)m * N + n], act ? y[(long long)m * N + n] : 0.f, act);
        acc += g * w[(long long)n * K + k];
    }
    acc = wave_sum(acc);
    if (lane == 0) dx[wave] = acc;
}
__global__ __launch_bounds__(256) void linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ w, float* __restrict__ dx, int M, int N,
                                                        int K, int act) {
    linear_dx_body(blockIdx.x, dy, y, w, dx, M, N, K, act);
}

// dw[n][k] = sum_m g[m][n] x[m][k];  db[n] = sum_m g[m][n] (k == 0 thread)
__device__ __forceinline__ void linear_dw_body(unsigned block, const float* __restrict__ dy, const float* __restrict__ y,
                                               const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db, int M, int N,
                                               int K, int act, int accumulate) {
    const long long idx = (long long)block * 256 + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx % K);
    float acc = 0.f, accb = 0.f;
#pragma unroll 8
    for (int m = 0; m < M; ++m) {
        const float g = lin_act_grad(dy[(long long)m * N + n], act ? y[(long long)m * N + n] : 0.f, act);
        acc += g * x[(long long)m * K + k];
        accb += g;
    }
    dw[idx] = accumulate ? dw[idx] + acc : acc;
    if (db && k == 0) db[n] = accumulate ? db[n] + accb : accb;
}
__global__ __launch_bounds__(256) void linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ x, float* __restrict__ dw,
                                                        float* __restrict__ db, int M, int N, int K, int act,
                                                        int accumulate) {
    linear_dw_body(blockIdx.x, dy, y, x, dw, db, M, N, K, act, accumulate);
}

// ---- Round 3: the three Linear products as 64 x 64 LDS tiles (4 x 4 outputs per thread).  The one-output-per-thread / per-wave
// kernels above cost 9-28 us per call on the squeeze-excitation layers of EfficientNet-B3 (53 calls of each per step, 2.6 ms): the
// weight gradient re-evaluated the activation derivative (an exp) N*K*M times instead of N*M, the forward ran one WAVE per output.
//   MODE 0 forward  C[m][n] = sum_k x[m][k] w[n][k] (+ b[n], activation; optional pre-activation copy)
//   MODE 1 dx       C[m][k] = sum_n g[m][n] w[n][k]
//   MODE 2 dw       C[n][k] = sum_m g[m][n] x[m][k], db[n] = sum_m g[m][n]
// with g = dy * act'(y).  Both operands sit in LDS "contraction-major" ([l][64 + 1]) so the inner loop is two conflict-free
// 16-byte reads per 16 multiply-adds.
template <int MODE>
__device__ __forceinline__ void linear_tile_body(float (*At)[68], float (*Bt)[68], unsigned bx, unsigned by, const float* __restrict__ a0,
                                                 const float* __restrict__ a1, const float* __restrict__ b0, const float* __restrict__ bias,
                                                 float* __restrict__ out, float* __restrict__ out2, int M, int N, int K, int act, int accumulate,
                                                 int lb = 0, int le = -1) {
    // I x J outputs, contraction length L (or its slice [lb, le), lb a multiple of 64: the split-M weight gradient)
    const int I = MODE == 2 ? N : M, J = MODE == 0 ? N : K, L = le >= 0 ? le : (MODE == 0 ? K : (MODE == 1 ? N : M));
    const int i0 = (int)bx * 64, j0 = (int)by * 64;
    const int ti = threadIdx.x & 15, tj = threadIdx.x >> 4;
    float acc[4][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
    for (int l0 = lb; l0 < L; l0 += 64) {
        __syncthreads();
        // Loads first, LDS stores second: written as one loop (load, store, load, store ...) hipcc waits for every load before the
        // store that follows it -- 16 memory round trips per 64-deep step, 6.5 us per step whatever the problem size (a 64 x 24 x 576
        // squeeze-excitation product took 11.5 us, 64 x 128 x 576 18.8 us; the vendor library 4.6-5.2 us).
        float ra[16], rb[16], rg[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = (int)threadIdx.x + 256 * it;
            const int r = e >> 6, c = e & 63;              // c runs along the contiguous global dimension of each operand
            rg[it] = 0.f;
            if constexpr (MODE == 0) {                     // x[m][k] -> At[k][m];  w[n][k] -> Bt[k][n]
                const int m = i0 + r, k = l0 + c, n = j0 + r;
                ra[it] = (m < M && k < K) ? a0[(long long)m * K + k] : 0.f;
                rb[it] = (n < N && k < K) ? b0[(long long)n * K + k] : 0.f;
            } else if constexpr (MODE == 1) {              // g[m][n] -> At[n][m];  w[n][k] -> Bt[n][k]
                const int m = i0 + r, n = l0 + c;
                const bool ok = m < M && n < N;
                ra[it] = ok ? a0[(long long)m * N + n] : 0.f;
                if (act) rg[it] = ok ? a1[(long long)m * N + n] : 0.f;
                const int n2 = l0 + r, k = j0 + c;
                rb[it] = (n2 < N && k < K) ? b0[(long long)n2 * K + k] : 0.f;
            } else {                                       // g[m][n] -> At[m][n];  x[m][k] -> Bt[m][k]
                const int m = l0 + r, n = i0 + c, k = j0 + c;
                const bool ok = m < M && n < N;
                ra[it] = ok ? a0[(long long)m * N + n] : 0.f;
                if (act) rg[it] = ok ? a1[(long long)m * N + n] : 0.f;
                rb[it] = (m < M && k < K) ? b0[(long long)m * K + k] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = (int)threadIdx.x + 256 * it;
            const int r = e >> 6, c = e & 63;
            if constexpr (MODE == 0) {
                At[c][r] = ra[it];
                Bt[c][r] = rb[it];
            } else if constexpr (MODE == 1) {
                At[c][r] = lin_act_grad(ra[it], rg[it], act);        // (zero gradient in, zero out: the padding stays zero)
                Bt[r][c] = rb[it];
            } else {
                At[r][c] = lin_act_grad(ra[it], rg[it], act);
                Bt[r][c] = rb[it];
            }
        }
        __syncthreads();
        // Eight contraction steps per batch: all 16 LDS reads of a batch are issued before the first multiply-add.  Written one step per
        // iteration hipcc emitted read, read, s_waitcnt lgkmcnt(0), 8 packed FMAs -- a full LDS latency per step with ONE wave per SIMD to
        // hide it: ~150 cycles per step, 5-6 us per 64-deep tile whatever the problem size.  (Rows past the contraction's end are
        // zeros in LDS: the count is rounded up to a batch.)
        const int lrem = L - l0 < 64 ? L - l0 : 64;
        for (int lq = 0; lq < lrem; lq += 8) {
            float4 av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                av[q] = *reinterpret_cast<const float4*>(&At[lq + q][4 * ti]);
                bv[q] = *reinterpret_cast<const float4*>(&Bt[lq + q][4 * tj]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float a4[4] = {av[q].x, av[q].y, av[q].z, av[q].w}, b4[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    accb[u] += a4[u];
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] += a4[u] * b4[v];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + 4 * ti + u;
        if (i >= I) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = j0 + 4 * tj + v;
            if (j >= J) continue;
            const long long o = (long long)i * J + j;
            float val = acc[u][v];
            if constexpr (MODE == 0) {
                val += bias ? bias[j] : 0.f;
                if (out2) out2[o] = val;
                if (act == CS_ACT_RELU) val = val > 0.f ? val : 0.f;
                else if (act == CS_ACT_SILU) val = val / (1.f + expf(-val));
                else if (act == CS_ACT_SIGMOID) val = 1.f / (1.f + expf(-val));
                out[o] = val;
            } else if constexpr (MODE == 1) {
                out[o] = val;
            } else {
                out[o] = accumulate ? out[o] + val : val;
            }
        }
        if constexpr (MODE == 2) {
            if (out2 && by == 0 && tj == 0) out2[i] = accumulate ? out2[i] + accb[u] : accb[u];
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void linear_tiled_kernel(const float* __restrict__ a0, const float* __restrict__ a1, const float* __restrict__ b0,
                                                           const float* __restrict__ bias, float* __restrict__ out, float* __restrict__ out2,
                                                           int M, int N, int K, int act, int accumulate) {
    __shared__ __attribute__((aligned(16))) float At[64][68];
    __shared__ __attribute__((aligned(16))) float Bt[64][68];
    linear_tile_body<MODE>(At, Bt, blockIdx.x, blockIdx.y, a0, a1, b0, bias, out, out2, M, N, K, act, accumulate);
}

// Weight gradient of a Linear with a LONG batch axis (round 5): dw[n][k] = sum_m g[m][n] x[m][k] contracts over the M rows -- 64 in the
// bags of BASELINE's configs, but 8 192 - 40 960 at the reference's own operating point (train_tile.py -b 40960: the squeeze-excitation
// layers of an EfficientNet encoder then see one row per TILE).  A handful of 64 x 64 output tiles walking thousands of rows one 64-row step
// after the other took 1.8 - 3.3 ms per launch (76 of the 93 ms of an EfficientNet-B0 step at batch 8 192).  Here blockIdx.z cuts the rows
// into slices of `rows_per_split` (a multiple of 64); slice s leaves its partial dw / db in ws[s] and linear_splitm_fold_kernel adds the
// slices in order (bitwise reproducible).
__global__ __launch_bounds__(256) void linear_dw_splitm_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                               float* __restrict__ ws, float* __restrict__ wsb, int M, int N, int K, int act,
                                                               int rows_per_split) {
    __shared__ __attribute__((aligned(16))) float At[64][68];
    __shared__ __attribute__((aligned(16))) float Bt[64][68];
    const int s = blockIdx.z;
    const int lb = s * rows_per_split;
    int le = lb + rows_per_split;
    if (le > M) le = M;
    linear_tile_body<2>(At, Bt, blockIdx.x, blockIdx.y, dy, y, x, nullptr, ws + (long long)s * N * K, wsb + (long long)s * N, M, N, K, act, 0, lb, le);
}
__global__ __launch_bounds__(256) void linear_splitm_fold_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, int S, long long NK, int N,
                                                                 float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NK) {
        float t = accumulate ? dw[i] : 0.f;
        for (int s = 0; s < S; ++s) t += ws[(long long)s * NK + i];
        dw[i] = t;
    }
    if (db && i < N) {
        float t = accumulate ? db[i] : 0.f;
        for (int s = 0; s < S; ++s) t += wsb[(long long)s * N + i];
        db[i] = t;
    }
}

// ---- Round 5: direct forms of the two backward products for the 64-row squeeze-excitation layers.  The activation gradient g = dy * act'(y)
// of ONE row (dx) or ONE column (dw) is staged in LDS once per workgroup; the products then run as plain loops with coalesced loads,
// contraction index ascending with one fused multiply-add per step (the tiled kernel's summation order: same bits).
// dx[m][k] = sum_n g[m][n] w[n][k] for N <= 128: workgroup = (row m, 256 consecutive k)
__device__ __forceinline__ void linear_dx_shortn_body(unsigned block, float* gs /* [128] */, const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ w, float* __restrict__ dx, int M, int N, int K, int act) {
    const unsigned kb = (unsigned)((K + 255) / 256);
    const int m = (int)(block / kb), k = (int)(block % kb) * 256 + (int)threadIdx.x;
    if ((int)threadIdx.x < N) {
        const long long o = (long long)m * N + threadIdx.x;
        gs[threadIdx.x] = lin_act_grad(dy[o], act ? y[o] : 0.f, act);
    }
    __syncthreads();
    if (k >= K) return;
    float acc = 0.f;
    int n = 0;
    for (; n + 8 <= N; n += 8) {
        float wv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = w[(long long)(n + q) * K + k];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(gs[n + q], wv[q], acc);
    }
    for (; n < N; ++n) acc = fmaf(gs[n], w[(long long)n * K + k], acc);
    dx[(long long)m * K + k] = acc;
}
// dw[n][k] = sum_m g[m][n] x[m][k], db[n] = sum_m g[m][n] for M <= 128: workgroup = NPB output rows n x KT consecutive k (KT = 64 when
// K <= 64: four rows per workgroup, else 256: one)
__device__ __forceinline__ void linear_dw_shortm_body(unsigned block, float* gs /* [4][128] */, const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db, int M, int N, int K,
                                                      int act, int accumulate) {
    const int KT = K <= 64 ? 64 : 256, NPB = 256 / KT;
    const unsigned kb = (unsigned)((K + KT - 1) / KT);
    const int nl = (int)threadIdx.x / KT, kl = (int)threadIdx.x % KT;
    const int n0 = (int)(block / kb) * NPB, k = (int)(block % kb) * KT + kl;
    // column n0 + j of g, j < NPB: M * NPB values, thread t -> (j = t / M, m = t % M) while t < NPB * M (<= 512: two rounds)
    for (int t = threadIdx.x; t < NPB * M; t += 256) {
        const int j = t / M, m = t - j * M;
        const int n = n0 + j;
        float g = 0.f;
        if (n < N) {
            const long long o = (long long)m * N + n;
            g = lin_act_grad(dy[o], act ? y[o] : 0.f, act);
        }
        gs[j * 128 + m] = g;
    }
    __syncthreads();
    const int n = n0 + nl;
    if (n >= N) return;
    const float* g1 = gs + nl * 128;
    if (k < K) {
        float acc = 0.f;
        int m = 0;
        for (; m + 8 <= M; m += 8) {
            float xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = x[(long long)(m + q) * K + k];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fmaf(g1[m + q], xv[q], acc);
        }
        for (; m < M; ++m) acc = fmaf(g1[m], x[(long long)m * K + k], acc);
        const long long o = (long long)n * K + k;
        dw[o] = accumulate ? dw[o] + acc : acc;
    }
    if (db && block % kb == 0 && kl == 0) {
        float t = 0.f;
        for (int m = 0; m < M; ++m) t += g1[m];
        db[n] = accumulate ? db[n] + t : t;
    }
}

// dx and dw of one Linear in ONE launch: the two products are independent, but as two launches they ran one after the other -- two chains
// of latencies on a handful of workgroups each (the squeeze-excitation layers of EfficientNet: 104 such launches, 10-15 us each, per
// step).  Workgroups [0, nx) compute dx, the rest dw; each side keeps its own kernel shape (64 x 64 tiles or one wave / thread per output).
// modes: dx 0 = one wave per output, 1 = 64 x 64 tiles, 2 = short contraction (N <= 128);
//        dw 0 = one thread per output, 1 = tiles, 2 = short contraction (M <= 128);  -1 = that product is not wanted
// (Tried for the long contraction with K <= 64 outputs per row -- the second Linear's input gradient, N = 288 .. 2304: one workgroup
// per row, 64 k lanes x 4 slices of n.  64 workgroups walking N / 4 dependent load batches: 14 - 58 us against 8 - 17 us of one wave per
// output, whose 3712 waves hide each other's latencies.  Removed.)
template <int DXM, int DWM>
__global__ __launch_bounds__(256) void linear_bwd_dual_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                              const float* __restrict__ y, float* __restrict__ dx, float* __restrict__ dw,
                                                              float* __restrict__ db, int M, int N, int K, int act, int accumulate, unsigned nx,
                                                              unsigned dx_gx, unsigned dw_gx) {
    constexpr bool TILES = DXM == 1 || DWM == 1;
    __shared__ __attribute__((aligned(16))) float At[TILES ? 64 : 1][68];
    __shared__ __attribute__((aligned(16))) float Bt[TILES ? 64 : 1][68];
    __shared__ float gs[(DXM == 2 || DWM == 2) ? 512 : 1];
    unsigned b = blockIdx.x;
    if (b < nx) {
        if constexpr (DXM == 1) linear_tile_body<1>(At, Bt, b % dx_gx, b / dx_gx, dy, y, w, nullptr, dx, nullptr, M, N, K, act, 0);
        else if constexpr (DXM == 2) linear_dx_shortn_body(b, gs, dy, y, w, dx, M, N, K, act);
        else if constexpr (DXM == 0) linear_dx_body(b, dy, y, w, dx, M, N, K, act);
    } else {
        b -= nx;
        if constexpr (DWM == 1) linear_tile_body<2>(At, Bt, b % dw_gx, b / dw_gx, dy, y, x, nullptr, dw, db, M, N, K, act, accumulate);
        else if constexpr (DWM == 2) linear_dw_shortm_body(b, gs, dy, y, x, dw, db, M, N, K, act, accumulate);
        else if constexpr (DWM == 0) linear_dw_body(b, dy, y, x, dw, db, M, N, K, act, accumulate);
    }
}

// rows handled one per thread (C is 2 or 7 here); block partial sums -> one atomic per block
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         float gamma, float* __restrict__ loss, float* __restrict__ dlogits,
                                                         int M, int C) {
    // grid-stride over the rows: at most kLossBlocks workgroups contribute to the exact accumulator (its additions are exact for up to
    // 2^12 contributors, cs_common.h), whatever M is
    float li = 0.f;
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
        const float* r = logits + m * C;
        float mx = r[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(r[c] - mx);
        const float lse = mx + logf(se);
        const int lab = (int)labels[m];
        li += (lse - r[lab]) * (gamma / (float)M);
        if (dlogits) {
            const float gs = gamma / (float)M;
            for (int c = 0; c < C; ++c) {
                const float pr = expf(r[c] - lse);
                dlogits[(long long)m * C + c] = gs * (pr - (c == lab ? 1.f : 0.f));
            }
        }
    }
    __shared__ float red[4];
    li = wave_sum(li);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = li;
    __syncthreads();
    // the workgroups' sums meet in an exact accumulator behind the result (loss[2 ..]: any arrival order, same bits); loss_finish_kernel
    // rounds the total into loss[0]
    if (threadIdx.x == 0) ex_add(loss + 2, 1, 0, 0, (double)((red[0] + red[1]) + (red[2] + red[3])));
}

__global__ void softmax_prob1_kernel(const float* __restrict__ logits, float* __restrict__ p1, int M, int C) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* r = logits + (long long)m * C;
    float mx = r[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(r[c] - mx);
    p1[m] = expf(r[1] - mx) / se;
}

// inference.py:72-76,118-119: np.argmax(F.softmax(logits, 1), axis=1) -- the arg max is taken over the fp32 PROBABILITIES (two logits
// closer than the rounding of exp / the division tie as probabilities and the first index wins), not over the logits.
__global__ void softmax_argmax_kernel(const float* __restrict__ logits, long long* __restrict__ idx, int M, int C) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* r = logits + (long long)m * C;
    float mx = r[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(r[c] - mx);
    float best = expf(r[0] - mx) / se;
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float pc = expf(r[c] - mx) / se;
        if (pc > best) { best = pc; bi = c; }
    }
    idx[m] = bi;
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ x, const float* __restrict__ t, int weighted,
                                                  float inv, float* __restrict__ loss, float* __restrict__ dx, int M) {
    float li = 0.f;
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {      // (as softmax_ce_kernel)
        const float d = x[m] - t[m];
        float w = 1.f;
        if (weighted) w = t[m] >= 20.f ? logf(t[m]) : t[m];
        li += w * d * d * inv;
        if (dx) dx[m] = 2.f * w * d * inv;
    }
    __shared__ float red[4];
    li = wave_sum(li);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = li;
    __syncthreads();
    // the workgroups' sums meet in an exact accumulator behind the result (loss[2 ..]: any arrival order, same bits); loss_finish_kernel
    // rounds the total into loss[0]
    if (threadIdx.x == 0) ex_add(loss + 2, 1, 0, 0, (double)((red[0] + red[1]) + (red[2] + red[3])));
}


__global__ void loss_finish_kernel(float* __restrict__ loss) { loss[0] = (float)ex_read(loss + 2, 1, 0, 0); }

// ---- Dice (train/losses.py:44-62, metrics/metrics.py:36-53): per-sample a=sum p*t, b=sum p^2, c=sum t^2
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ p, const float* __restrict__ t, long long HW,
                                                        double* __restrict__ sums) {
    const int n = blockIdx.y;
    const float* pp = p + (long long)n * HW;
    const float* tt = t + (long long)n * HW;
    float a = 0.f, b = 0.f, c = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const float x = pp[i], y = tt[i];
        a += x * y; b += x * x; c += y * y;
    }
    __shared__ float red[3][4];
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double v = (double)red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        ex_add(sums, 3 * gridDim.y, 0, n * 3 + threadIdx.x, v);          // exact accumulator: any block order, same bits
    }
}

__global__ void dice_loss_kernel(const double* __restrict__ sums, int N, float eps, int mean, float* __restrict__ loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        const float a = (float)ex_read(sums, 3 * N, 0, n * 3), b = (float)ex_read(sums, 3 * N, 0, n * 3 + 1), c = (float)ex_read(sums, 3 * N, 0, n * 3 + 2);
        acc += 1.f - (2.f * a + eps) / (b + c + eps);
    }
    loss[0] = mean ? acc / (float)N : acc;
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                       const double* __restrict__ sums, long long HW, float eps, float scale,
                                                       float* __restrict__ dp) {
    const int n = blockIdx.y;
    const int C3 = 3 * gridDim.y;
    const float a = (float)ex_read(sums, C3, 0, n * 3), b = (float)ex_read(sums, C3, 0, n * 3 + 1), c = (float)ex_read(sums, C3, 0, n * 3 + 2);
    const float num = 2.f * a + eps, den = b + c + eps;
    const float k = -scale / (den * den);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const long long o = (long long)n * HW + i;
        dp[o] = k * (2.f * t[o] * den - num * 2.f * p[o]);
    }
}

// softmax over the channel axis of NCHW logits, channel `ch` only (F.softmax(out)[:, 1], train/train.py:189)
__global__ __launch_bounds__(256) void softmax_ch_fwd_kernel(const float* __restrict__ logits, float* __restrict__ pc, int C, int ch,
                                                             long long HW, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW, i = idx - n * HW;
        const float* r = logits + n * C * HW + i;
        float mx = r[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c * HW]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(r[c * HW] - mx);
        pc[idx] = expf(r[ch * HW] - mx) / se;
    }
}

__global__ __launch_bounds__(256) void softmax_ch_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ dpc,
                                                             float* __restrict__ dlogits, int C, int ch, long long HW,
                                                             long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW, i = idx - n * HW;
        const float* r = logits + n * C * HW + i;
        float mx = r[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c * HW]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(r[c * HW] - mx);
        const float pch = expf(r[ch * HW] - mx) / se;
        const float g = dpc[idx];
        for (int c = 0; c < C; ++c) {
            const float pcv = expf(r[c * HW] - mx) / se;
            dlogits[n * C * HW + c * HW + i] = g * pch * ((c == ch ? 1.f : 0.f) - pcv);
        }
    }
}

}  // namespace

// The 64 x 64 tiles pay when there are enough of them to spread over the chip or the contraction is short; a skinny product with a long
// contraction (squeeze-excitation fc1: 64 x C -> 64 x C/24, one or two tiles walking C = 2304 in 36 serial steps) stays on the
// one-wave-per-output kernels, which spread the contraction over thousands of waves (first attempt without this rule: EfficientNet-B3
// step 29 -> 34 ms).
static bool lin_use_tiles(int I, int J, int L) {
    const long long tiles = (long long)((I + 63) / 64) * ((J + 63) / 64);
    return tiles >= 8 || L <= 128;
}

extern "C" int cs_linear_fwd(const float* x, const float* w, const float* b, float* y, float* preact, int M, int N, int K, int act,
                             void* stream) {
    CS_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0, "linear_fwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long waves = (long long)M * N;
    static const int direct = cs_env_int_("CELLSEG_LINEAR_DIRECT", 1);       // A/B flavour only (0: the tiles everywhere they used to run)
    if (direct && waves >= 1024 && K <= 64 && waves <= (1LL << 22)) {
        hipLaunchKernelGGL(linear_fwd_shortk_kernel, dim3((unsigned)((waves + 255) / 256)), dim3(256), 0, st, x, w, b, y, preact, M, N, K, act);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (waves >= 1024 && lin_use_tiles(M, N, K)) {
        hipLaunchKernelGGL(linear_tiled_kernel<0>, dim3((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, st, x, nullptr, w, b, y,
                           preact, M, N, K, act, 0);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, w, b, y, preact, M, N, K, act);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// row slices of the split-M weight gradient: none up to 512 rows; else enough slices for ~1024 workgroups over the output tiles, each of
// >= 256 rows (a multiple of 64), at most 128 slices (the fold reads slices x outputs floats: 57 us per launch with 128 slices of a
// 1152 x 48 gradient)
static int lin_dw_rows_per_split(int M, int N, int K) {
    if (M <= 512) return M;
    const long long tiles = (long long)((N + 63) / 64) * ((K + 63) / 64);
    long long s = 1024 / (tiles > 0 ? tiles : 1);
    if (s < 2) s = 2;
    if (s > 128) s = 128;
    long long rps = ((M + s - 1) / s + 63) / 64 * 64;
    if (rps < 256) rps = 256;
    return (int)rps;
}
extern "C" size_t cs_linear_bwd_workspace(int M, int N, int K) {
    if (M <= 512 || N <= 0 || K <= 0) return 0;
    const int rps = lin_dw_rows_per_split(M, N, K);
    const int S = (M + rps - 1) / rps;
    return (size_t)S * ((size_t)N * K + (size_t)N) * sizeof(float);
}

extern "C" int cs_linear_bwd(const float* x, const float* w, const float* dy, const float* y, int act, float* dx, float* dw,
                             float* db, int M, int N, int K, int accumulate, float* workspace, void* stream) {
    CS_CHECK_ARG(dy && M > 0 && N > 0 && K > 0, "linear_bwd: bad arguments");
    CS_CHECK_ARG(act == CS_ACT_NONE || y, "linear_bwd: an output activation needs y (SiLU: the pre-activation)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dw && M > 512) {
        // long batch axis: dx on its own, dw over row slices + an ordered fold (linear_dw_splitm_kernel)
        CS_CHECK_ARG(x && workspace, "linear_bwd: dw over more than 512 rows needs x and a workspace of cs_linear_bwd_workspace() bytes");
        if (dx) {
            const int rc = cs_linear_bwd(x, w, dy, y, act, dx, nullptr, nullptr, M, N, K, 0, nullptr, stream);
            if (rc != CS_OK) return rc;
        }
        const int rps = lin_dw_rows_per_split(M, N, K);
        const int S = (M + rps - 1) / rps;
        float* wsb = workspace + (size_t)S * N * K;
        hipLaunchKernelGGL(linear_dw_splitm_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64), (unsigned)S), dim3(256), 0, st, dy, y, x,
                           workspace, wsb, M, N, K, act, rps);
        CS_LAUNCH_CHECK();
        const long long NK = (long long)N * K;
        hipLaunchKernelGGL(linear_splitm_fold_kernel, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, st, workspace, wsb, S, NK, N, dw, db, accumulate);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    CS_CHECK_ARG(!dx || w, "linear_bwd: dx needs w");
    CS_CHECK_ARG(!dw || x, "linear_bwd: dw needs x");
    // kernel shape of each product (-1: not wanted)
    static const int direct = cs_env_int_("CELLSEG_LINEAR_DIRECT", 1);       // A/B flavour only
    int dxm = -1, dwm = -1;
    unsigned nx = 0, nw = 0;
    const unsigned dx_gx = (unsigned)((M + 63) / 64), dw_gx = (unsigned)((N + 63) / 64), gy = (unsigned)((K + 63) / 64);
    if (dx) {
        if (direct && N <= 128) { dxm = 2; nx = (unsigned)M * (unsigned)((K + 255) / 256); }
        else if ((long long)M * K >= 1024 && lin_use_tiles(M, K, N)) { dxm = 1; nx = dx_gx * gy; }
        else { dxm = 0; nx = (unsigned)(((long long)M * K + 3) / 4); }
    }
    if (dw) {
        if (direct && M <= 128) {
            const int KT = K <= 64 ? 64 : 256, NPB = 256 / KT;
            dwm = 2; nw = (unsigned)((N + NPB - 1) / NPB) * (unsigned)((K + KT - 1) / KT);
        } else if ((long long)N * K >= 1024 && lin_use_tiles(N, K, M)) { dwm = 1; nw = dw_gx * gy; }
        else { dwm = 0; nw = (unsigned)(((long long)N * K + 255) / 256); }
    }
#define CS_LIN_DUAL(A_, B_)                                                                                                            \
    case (A_ + 1) * 4 + (B_ + 1):                                                                                                        \
        hipLaunchKernelGGL((linear_bwd_dual_kernel<A_, B_>), dim3(nx + nw), dim3(256), 0, st, x, w, dy, y, dx, dw, db, M, N, K, act, accumulate, nx, \
                           dx_gx, dw_gx);                                                                                                \
        break
    switch ((dxm + 1) * 4 + (dwm + 1)) {
        CS_LIN_DUAL(0, 0); CS_LIN_DUAL(0, 1); CS_LIN_DUAL(0, 2); CS_LIN_DUAL(1, 0); CS_LIN_DUAL(1, 1); CS_LIN_DUAL(1, 2);
        CS_LIN_DUAL(2, 0); CS_LIN_DUAL(2, 1); CS_LIN_DUAL(2, 2);
        CS_LIN_DUAL(0, -1); CS_LIN_DUAL(1, -1); CS_LIN_DUAL(2, -1);
        CS_LIN_DUAL(-1, 0); CS_LIN_DUAL(-1, 1); CS_LIN_DUAL(-1, 2);
        default: return CS_OK;         // neither product wanted
    }
#undef CS_LIN_DUAL
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// floats behind a loss value: [0] the result, [1] unused, [2 .. 15] an exact accumulator of one channel (7 eight-byte words)
constexpr int kLossWords = 16;
constexpr int kLossBlocks = 2048;       // workgroups of a loss launch: the exact accumulator takes <= 2^12 contributions
extern "C" int cs_loss_words(void) { return kLossWords; }

extern "C" int cs_softmax_ce(const float* logits, const int64_t* labels, float gamma, float* loss, float* dlogits, int M, int C,
                             void* stream) {
    CS_CHECK_ARG(logits && labels && loss && M > 0 && C > 1, "softmax_ce: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float) * kLossWords, st) != hipSuccess) { cs_set_error_("softmax_ce: memset failed"); return CS_ERR_LAUNCH; }
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((M + 255) / 256 < kLossBlocks ? (M + 255) / 256 : kLossBlocks), dim3(256), 0, st, logits, labels, gamma, loss, dlogits, M, C);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, loss);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_softmax_prob1(const float* logits, float* p1, int M, int C, void* stream) {
    CS_CHECK_ARG(logits && p1 && M > 0 && C > 1, "softmax_prob1: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(softmax_prob1_kernel, dim3((M + 255) / 256), dim3(256), 0, st, logits, p1, M, C);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_softmax_argmax(const float* logits, int64_t* idx, int M, int C, void* stream) {
    CS_CHECK_ARG(logits && idx && M > 0 && C > 0, "softmax_argmax: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3((M + 255) / 256), dim3(256), 0, st, logits, reinterpret_cast<long long*>(idx), M, C);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_mse(const float* x, const float* t, int weighted, int mean, float* loss, float* dx, int M, void* stream) {
    CS_CHECK_ARG(x && t && loss && M > 0, "mse: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float) * kLossWords, st) != hipSuccess) { cs_set_error_("mse: memset failed"); return CS_ERR_LAUNCH; }
    const float inv = mean ? 1.f / (float)M : 1.f;
    hipLaunchKernelGGL(mse_kernel, dim3((M + 255) / 256 < kLossBlocks ? (M + 255) / 256 : kLossBlocks), dim3(256), 0, st, x, t, weighted, inv, loss, dx, M);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, loss);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_dice_fwd(const float* p, const float* t, int N, long long HW, float eps, int mean, double* sums, float* loss,
                           void* stream) {
    CS_CHECK_ARG(p && t && sums && loss && N > 0 && HW > 0, "dice_fwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * (size_t)ex_words(3 * N), st) != hipSuccess) { cs_set_error_("dice_fwd: memset failed"); return CS_ERR_LAUNCH; }
    long long bx = (HW + 256 * 8 - 1) / (256 * 8);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(dice_sums_kernel, dim3((unsigned)bx, N), dim3(256), 0, st, p, t, HW, sums);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_loss_kernel, dim3(1), dim3(64), 0, st, sums, N, eps, mean, loss);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_dice_bwd(const float* p, const float* t, const double* sums, int N, long long HW, float eps, int mean, float* dp,
                           void* stream) {
    CS_CHECK_ARG(p && t && sums && dp && N > 0 && HW > 0, "dice_bwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long long bx = (HW + 256 * 4 - 1) / (256 * 4);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)bx, N), dim3(256), 0, st, p, t, sums, HW, eps, mean ? 1.f / (float)N : 1.f, dp);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_softmax_channel_fwd(const float* logits, float* pc, int N, int C, long long HW, int ch, void* stream) {
    CS_CHECK_ARG(logits && pc && N > 0 && C > 1 && HW > 0 && ch >= 0 && ch < C, "softmax_channel_fwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * HW;
    long long b = (total + 255) / 256; if (b > 8192) b = 8192;
    hipLaunchKernelGGL(softmax_ch_fwd_kernel, dim3((unsigned)b), dim3(256), 0, st, logits, pc, C, ch, HW, total);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_softmax_channel_bwd(const float* logits, const float* dpc, float* dlogits, int N, int C, long long HW, int ch,
                                      void* stream) {
    CS_CHECK_ARG(logits && dpc && dlogits && N > 0 && C > 1 && HW > 0 && ch >= 0 && ch < C, "softmax_channel_bwd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * HW;
    long long b = (total + 255) / 256; if (b > 8192) b = 8192;
    hipLaunchKernelGGL(softmax_ch_bwd_kernel, dim3((unsigned)b), dim3(256), 0, st, logits, dpc, dlogits, C, ch, HW, total);
    CS_LAUNCH_CHECK();
    return CS_OK;
}
