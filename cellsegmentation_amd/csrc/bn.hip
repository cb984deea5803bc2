// BatchNorm in TRAIN mode (batch statistics) for NHWC rows [M][C] -- used by the image-mode and
// segment-mode trunks (model.train(): model/resnet.py:21,52,54,56,112,184,199) and by the
// BatchNorm1d layers of the image heads (resnet.py:134,138,144,148; M = batch, C = features).
//
//   stats   : per-channel sum / sum of squares in fp64 (also produced by the conv epilogue)
//   finalize: mean, rstd = 1/sqrt(biased var + eps); running stats updated with momentum and the
//             UNBIASED variance, exactly like nn.BatchNorm2d
//   apply   : y = act( gamma*(z-mean)*rstd + beta + residual )
//   backward: reduce  s0 = sum g, s1 = sum g*xhat ;  dz = gamma*rstd*( g - s0/M - xhat*s1/M )
// All HBM-bound: 8 channels (16 or 32 bytes) per thread, rows strided over the workgroup.
#include "cs_common.h"

namespace {

// Two per-thread partial sums for 8 channels -> one exact (order-independent, cs_common.h: ex_add) contribution per channel per workgroup.
// Threads are laid out tid = rr*width + (cg - cg0); rows rr < rpar hold valid partials.
// partial != NULL: the workgroup's two sums go to partial[(2*blockIdx.x + {0,1}) * C + c] with plain stores instead (folded by
// bn_partial_fold_kernel): ~1000 fp64 atomics per address serialise at the memory side and cost more than the streaming pass.
__device__ __forceinline__ void block_fold_atomic(const float (&a)[8], const float (&b)[8], int width, int rpar, int cg, bool live,
                                                  double* acc, int C, double* partial = nullptr) {
    __shared__ float fold[2][256][8];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        fold[0][threadIdx.x][e] = live ? a[e] : 0.f;
        fold[1][threadIdx.x][e] = live ? b[e] : 0.f;
    }
    __syncthreads();
    // one thread per (sum, channel) COLUMN -- 2 * 8 * width of them: every lane folds rpar values.  (The first version let `width`
    // threads fold 16 columns each: with 18 channel groups that is 18 lanes walking 224 dependent LDS reads, ~9 us at the end of every
    // workgroup of a 26-us launch.)
    const int cg0 = cg - (int)(threadIdx.x % width);
    const int ncols = width * 8;
    for (int col = threadIdx.x; col < 2 * ncols; col += 256) {
        const int a_ = col >= ncols ? 1 : 0;
        const int c_ = col - a_ * ncols;
        const int cgl = c_ >> 3, e = c_ & 7;
        double t = 0.0;
        for (int r = 0; r < rpar; ++r) t += (double)fold[a_][r * width + cgl][e];
        const int ch = (cg0 + cgl) * 8 + e;
        if (partial) partial[(2LL * blockIdx.x + a_) * C + ch] = t;
        else ex_add(acc, C, a_, ch, t);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ z, long long M, int C, double* __restrict__ stats,
                                                       int rows_per_block, double* __restrict__ partial, int cw) {
    const int CG = C / 8;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    {
        // grid = (row blocks, channel chunks of <= cw 8-channel groups), red_split below
        const int cg0 = blockIdx.y * cw;
        const int width = (CG - cg0) < cw ? (CG - cg0) : cw;
        const int rpar = 256 / width;
        const int cg = cg0 + (int)(threadIdx.x % width);
        const int rr = threadIdx.x / width;
        const bool live = rr < rpar;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        if (live) {
            long long r = r0 + rr;
            for (; r + 3LL * rpar < r1; r += 4LL * rpar) {            // four independent 16-byte loads in flight per thread
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) load8<T>(z + (r + (long long)u * rpar) * C + cg * 8, v[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
                    s2[e] += (v[0][e] * v[0][e] + v[1][e] * v[1][e]) + (v[2][e] * v[2][e] + v[3][e] * v[3][e]);
                }
            }
            for (; r < r1; r += rpar) {
                float v[8];
                load8<T>(z + r * C + cg * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
        }
        block_fold_atomic(s1, s2, width, rpar, cg, live, stats, C, partial);
    }
}

// batch mean / 1/sqrt(var + eps) of one channel from its fp64 sums: ONE definition for cs_bn_finalize and for the apply pass that
// derives them itself (cs_bn_apply_stats), so both give the same bits
// (the sums and the mean / variance are fp64; 1/sqrt(var + eps) is formed in fp32 like ATen's batch_norm does for fp32 / bf16 inputs --
// fp64 divisions and a fp64 square root per channel and THREAD cost the fused apply pass 3 us per launch)
__device__ __forceinline__ void bn_moments(double s0, double s1, long long M, float eps, float& mean_f, float& rstd_f, double& var_out) {
    const double inv_m = 1.0 / (double)M;
    const double mean = s0 * inv_m;
    double var = fma(s1, inv_m, -mean * mean);
    if (var < 0) var = 0;
    mean_f = (float)mean;
    rstd_f = 1.0f / sqrtf((float)var + eps);
    var_out = var;
}
__device__ __forceinline__ void bn_finalize_channel(const double* __restrict__ stats, int c, int C, long long M, float eps, float momentum,
                                                    float* running_mean, float* running_var, float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out) {
    float mean, rstd;
    double var;
    bn_moments(ex_read(stats, C, 0, c), ex_read(stats, C, 1, c), M, eps, mean, rstd, var);
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, long long M, float eps, float momentum,
                                   float* running_mean, float* running_var, float* __restrict__ mean_out,
                                   float* __restrict__ rstd_out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_finalize_channel(stats, c, C, M, eps, momentum, running_mean, running_var, mean_out, rstd_out);
}

// Row-strided element-wise passes (round 3): a thread keeps ONE 8-channel group and walks rows r0 + rr, r0 + rr + rpar, ... of its
// workgroup's row block, so every per-channel constant is loaded and combined once per thread -- round 2's flat index walk re-read
// 4-6 parameter vectors (and, in the backward apply, 16 fp64 sums) per 16 bytes of data: instruction-bound at 1.8-2.3 TB/s on the
// EfficientNet-B3 step.  Two rows (4-6 independent 16-byte loads) are in flight per thread.
// fin.stats != NULL (cs_bn_apply_stats): the launch also IS the finalize -- every thread derives mean / rstd of its 8 channels from the
// fp64 sums (bn_moments), workgroup 0 writes them out and updates the running statistics: one launch less per train-mode BatchNorm
// (78 per EfficientNet-B3 step, 61 per segmentation step).
struct BnFinalizeArgs {
    const double* stats; long long M; float eps, momentum; float* running_mean; float* running_var; float* mean_out; float* rstd_out;
};
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ residual,
                                                       int act, T* __restrict__ y, long long M, int C, int rows_per_block, int cw, BnFinalizeArgs fin) {
    // grid = (row blocks, channel chunks of <= cw 8-channel groups).  Round 5: with fin.stats every THREAD used to derive mean / rstd of
    // its 8 channels from the sums itself -- 16 fp64 loads per thread, every workgroup pulling the whole table through L2 (110 KB per
    // workgroup of a 2304-channel tensor once the sums became three-limb exact accumulators: +14 us per launch).  Now a workgroup owns
    // a chunk of <= 32 groups, derives the chunk's constants ONCE into LDS, and its threads read them from there.
    const int CG = C / 8;
    __shared__ float s_mu[2048], s_rs[2048];
    {
        const int cg0 = blockIdx.y * cw;
        const int width = (CG - cg0) < cw ? (CG - cg0) : cw;
        if (fin.stats) {
            for (int i = threadIdx.x; i < width * 8; i += 256) {
                const int c = cg0 * 8 + i;
                double var;
                bn_moments(ex_read(fin.stats, C, 0, c), ex_read(fin.stats, C, 1, c), fin.M, fin.eps, s_mu[i], s_rs[i], var);
                if (blockIdx.x == 0)
                    bn_finalize_channel(fin.stats, c, C, fin.M, fin.eps, fin.momentum, fin.running_mean, fin.running_var, fin.mean_out, fin.rstd_out);
            }
            __syncthreads();
        }
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    {
        const int cg0 = blockIdx.y * cw;
        const int width = (CG - cg0) < cw ? (CG - cg0) : cw;
        const int rpar = 256 / width;
        const int cgl = (int)(threadIdx.x % width);
        const int cg = cg0 + cgl;
        const int rr = threadIdx.x / width;
        if (rr >= rpar) return;
        float mu[8], rs[8], gm[8], bt[8];
        if (fin.stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = s_mu[cgl * 8 + e]; rs[e] = s_rs[cgl * 8 + e]; }
        } else {
            load8p(mean + cg * 8, 0.f, mu);
            load8p(rstd + cg * 8, 1.f, rs);
        }
        load8p(gamma ? gamma + cg * 8 : nullptr, 1.f, gm);
        load8p(beta ? beta + cg * 8 : nullptr, 0.f, bt);
        auto finish = [&](float (&v)[8], const long long off) {
            // (z - mean) * rstd * gamma + beta in the reference's operation order
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] - mu[e]) * rs[e] * gm[e] + bt[e];
            if (residual) {
                float r8[8];
                load8<T>(residual + off, r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
            }
            if (act == CS_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            } else if (act == CS_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
            }
            store8<T>(y + off, v);
        };
        long long r = r0 + rr;
        for (; r + rpar < r1; r += 2LL * rpar) {
            const long long o0 = r * C + cg * 8, o1 = (r + rpar) * C + cg * 8;
            float v0[8], v1[8];
            load8<T>(z + o0, v0);
            load8<T>(z + o1, v1);
            finish(v0, o0);
            finish(v1, o1);
        }
        if (r < r1) {
            const long long o0 = r * C + cg * 8;
            float v0[8];
            load8<T>(z + o0, v0);
            finish(v0, o0);
        }
    }
}

// gradient through the activation that follows the normalisation: ReLU is handled by the consumers' masks
// (engine convention), SiLU needs u = gamma*xhat + beta: d silu(u)/du = sig*(1 + u*(1-sig)).
// eight elements at once with ONE branch on the activation: a branch per element makes every exp -> add -> rcp chain its own basic
// block, and the eight chains run back to back instead of interleaved
__device__ __forceinline__ void act_grad8(float (&g)[8], const float (&xh)[8], const float (&gm)[8], const float (&bt)[8], int act) {
    if (act & CS_BN_BWD_OWN_RELU) {
        // the ReLU that follows THIS normalisation with nothing in between (BatchNorm1d -> ReLU of the image heads, resnet.py:134-136):
        // the mask is the sign of the layer's own output, recomputed in the forward's operation order
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (xh[e] * gm[e] + bt[e]) > 0.f ? g[e] : 0.f;
    }
    if ((act & 0xff) != CS_ACT_SILU) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float u = gm[e] * xh[e] + bt[e];
        const float sg = sigmoid_fast(u);
        g[e] = g[e] * sg * (1.f + u * (1.f - sg));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                            long long M, int C, double* __restrict__ sums,
                                                            int rows_per_block, double* __restrict__ partial, int cw) {
    const int CG = C / 8;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    {
        // grid = (row blocks, channel chunks of <= cw 8-channel groups), red_split below
        const int cg0 = blockIdx.y * cw;
        const int width = (CG - cg0) < cw ? (CG - cg0) : cw;
        const int rpar = 256 / width;
        const int cg = cg0 + (int)(threadIdx.x % width);
        const int rr = threadIdx.x / width;
        const bool live = rr < rpar;
        float s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
        if (live) {
            float mu[8], rs[8], gm[8], bt[8];
            load8p(mean + cg * 8, 0.f, mu);
            load8p(rstd + cg * 8, 1.f, rs);
            load8p(gamma ? gamma + cg * 8 : nullptr, 1.f, gm);
            load8p(beta ? beta + cg * 8 : nullptr, 0.f, bt);
            auto acc = [&](float (&g)[8], const float (&zz)[8]) {
                float xh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xh[e] = (zz[e] - mu[e]) * rs[e];
                act_grad8(g, xh, gm, bt, act);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s0[e] += g[e]; s1[e] += g[e] * xh[e]; }
            };
            long long r = r0 + rr;
            for (; r + rpar < r1; r += 2LL * rpar) {             // four independent 16-byte loads in flight per thread
                float g0[8], z0[8], g1[8], z1[8];
                load8<T>(dy + r * C + cg * 8, g0);
                load8<T>(z + r * C + cg * 8, z0);
                load8<T>(dy + (r + rpar) * C + cg * 8, g1);
                load8<T>(z + (r + rpar) * C + cg * 8, z1);
                acc(g0, z0);
                acc(g1, z1);
            }
            if (r < r1) {
                float g0[8], z0[8];
                load8<T>(dy + r * C + cg * 8, g0);
                load8<T>(z + r * C + cg * 8, z0);
                acc(g0, z0);
            }
        }
        block_fold_atomic(s0, s1, width, rpar, cg, live, sums, C, partial);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                           const double* __restrict__ sums,
                                                           long long M, int C, T* __restrict__ dz, float* dgamma, float* dbeta,
                                                           int rows_per_block, int cw) {
    // grid = (row blocks, channel chunks of <= cw groups): the chunk's two totals per channel are read ONCE per workgroup (bn_apply_kernel)
    const int CG = C / 8;
    const float invM = 1.f / (float)M;
    __shared__ float s_k0[2048], s_k1[2048];
    const bool frozen = (act & CS_BN_BWD_FROZEN) != 0;      // running statistics: mean / rstd do not depend on the batch
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    {
        const int cg0 = blockIdx.y * cw;
        const int width = (CG - cg0) < cw ? (CG - cg0) : cw;
        for (int i = threadIdx.x; i < width * 8; i += 256) {
            const int c = cg0 * 8 + i;
            const float t0 = (float)ex_read(sums, C, 0, c), t1 = (float)ex_read(sums, C, 1, c);
            s_k0[i] = frozen ? 0.f : t0 * invM;
            s_k1[i] = frozen ? 0.f : t1 * invM;
            if (blockIdx.x == 0) {
                if (dbeta) dbeta[c] = t0;
                if (dgamma) dgamma[c] = t1;
            }
        }
        __syncthreads();
        const int rpar = 256 / width;
        const int cgl = (int)(threadIdx.x % width);
        const int cg = cg0 + cgl;
        const int rr = threadIdx.x / width;
        if (rr >= rpar) return;
        float mu[8], rs[8], gm[8], bt[8], k0[8], k1[8], gr[8];
        load8p(mean + cg * 8, 0.f, mu);
        load8p(rstd + cg * 8, 1.f, rs);
        load8p(gamma ? gamma + cg * 8 : nullptr, 1.f, gm);
        load8p(beta ? beta + cg * 8 : nullptr, 0.f, bt);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            k0[e] = s_k0[cgl * 8 + e];
            k1[e] = s_k1[cgl * 8 + e];
            gr[e] = gm[e] * rs[e];
        }
        auto finish = [&](float (&g)[8], const float (&zz)[8], const long long off) {
            float o[8], xh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[e] = (zz[e] - mu[e]) * rs[e];
            act_grad8(g, xh, gm, bt, act);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gr[e] * (g[e] - k0[e] - xh[e] * k1[e]);
            store8<T>(dz + off, o);
        };
        long long r = r0 + rr;
        for (; r + rpar < r1; r += 2LL * rpar) {
            const long long o0 = r * C + cg * 8, o1 = (r + rpar) * C + cg * 8;
            float g0[8], z0[8], g1[8], z1[8];
            load8<T>(dy + o0, g0);
            load8<T>(z + o0, z0);
            load8<T>(dy + o1, g1);
            load8<T>(z + o1, z1);
            finish(g0, z0, o0);
            finish(g1, z1, o1);
        }
        if (r < r1) {
            const long long o0 = r * C + cg * 8;
            float g0[8], z0[8];
            load8<T>(dy + o0, g0);
            load8<T>(z + o0, z0);
            finish(g0, z0, o0);
        }
    }
}

// out[j] += sum_b partial[b][j] for the 2*C columns of the per-workgroup partial sums.  Workgroup = 64 columns x 4 row lanes,
// gridDim.y row chunks -> <= 32 fp64 atomics per address (a column-per-thread walk over ~1000 rows was latency-bound: 38 us)
__global__ __launch_bounds__(256) void bn_partial_fold_kernel(const double* __restrict__ partial, int blocks, int cols, double* __restrict__ out) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int per = (blocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per;
    int b1 = b0 + per;
    if (b1 > blocks) b1 = blocks;
    double a0 = 0.0, a1 = 0.0;
    if (j < cols) {
        int b = b0 + rl;
        for (; b + 4 < b1; b += 8) { a0 += partial[(long long)b * cols + j]; a1 += partial[(long long)(b + 4) * cols + j]; }
        for (; b < b1; b += 4) a0 += partial[(long long)b * cols + j];
    }
    __shared__ double part[4][64];
    part[rl][threadIdx.x & 63] = a0 + a1;
    __syncthreads();
    if (rl == 0 && j < cols && b0 < blocks) {
        const int C = cols >> 1;
        ex_add(out, C, j >= C ? 1 : 0, j >= C ? j - C : j, (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
    }
}

// A reduction adds its per-workgroup sums with fp64 atomics (exact accumulators: 6 per channel and row block) instead of writing
// partial rows for a second launch to fold -- one launch less per small BatchNorm -- as long as there are at most 512 row blocks AND
// at most 300 k atomics in the launch (~100 G atomics/s: 3 us; 38 x 38 x 288 channels with 511 row blocks = 880 k spent 10 of 34 us
// in them, 6400 x 1392 with 100 row blocks = 835 k as long as in its 36 MB of loads).
constexpr int kBnAtomicBlocks = 512;
constexpr long long kBnAtomicBudget = 300000;
inline bool red_wants_partials(int C, long long blocks) { return blocks > kBnAtomicBlocks || blocks * (long long)C * 6 > kBnAtomicBudget; }

inline int fold_partial(const double* partial, int blocks, int C, double* out, hipStream_t st) {
    int chunks = blocks / 16;
    if (chunks < 1) chunks = 1;
    if (chunks > 32) chunks = 32;
    hipLaunchKernelGGL(bn_partial_fold_kernel, dim3((2 * C + 63) / 64, chunks), dim3(256), 0, st, partial, blocks, 2 * C, out);
    return 0;
}

// Decomposition of the reductions (bn_stats_kernel, bn_bwd_reduce_kernel): channel chunks of <= 16 / 32 / 64 8-channel groups (balanced),
// rows per workgroup for ~CELLSEG_BN_BLOCKS workgroups in all, never fewer than 8 row steps per thread.  A thread owns one 8-channel group
// and every rpar-th row of its block (rpar = 256 / chunk width).  Round 5: until then a workgroup covered ALL channels of its rows --
// 174 live threads, one row each, for EfficientNet's 6400 x 1392 tensors, 800 row blocks of 8 rows whose 2 x 1392 partial sums (as
// many bytes as the tensor) went through a workspace and a fold launch: 19 us for a pass that moves 36 MB.  With chunks the same
// tensor is 100 row blocks x 6 chunks with 8 rows in flight per chunk: 14 us.
struct RedSplit { int chunks, cw, rpb; };
inline RedSplit red_split(long long M, int C) {
    static const int target = cs_env_int_("CELLSEG_BN_BLOCKS", 1024);     // A/B experiments only
    const int CG = C / 8 > 0 ? C / 8 : 1;
    RedSplit s;
    // chunk width (measured on EfficientNet-B3's tensors, tools/bn_microbench.py): one chunk while it is at most 64 groups wide and the
    // rows alone give >= 800 workgroups; 16-group chunks for tensors so small that the 8-step floor leaves < 450 workgroups whatever the
    // width (fewer row blocks = fewer contributions per channel: most of those then fit the atomics budget, one launch); 32 otherwise
    static const int forced_w = cs_env_int_("CELLSEG_BN_CW", 0);           // A/B experiments only
    int max_w = 32;
    if (CG <= 64 && M >= 800LL * 8 * (256 / CG)) max_w = 64;
    else if (M * CG < 450LL * 2048) max_w = 16;
    if (forced_w > 0) max_w = forced_w;
    s.chunks = (CG + max_w - 1) / max_w;
    s.cw = (CG + s.chunks - 1) / s.chunks;
    s.chunks = (CG + s.cw - 1) / s.cw;
    const int rpar = 256 / s.cw;
    long long row_blocks = target / s.chunks;
    if (row_blocks < 1) row_blocks = 1;
    long long r = (M + row_blocks - 1) / row_blocks;
    const long long floor_rows = 8LL * rpar;
    if (r < floor_rows) r = floor_rows;
    s.rpb = (int)r;
    return s;
}

// row block of the element-wise passes: ~CELLSEG_EW_BLOCKS (2048: 8 workgroups per compute unit) blocks, at least 16 rows each
inline int ew_rows_per_block(long long M, int C) {
    static const int target = cs_env_int_("CELLSEG_EW_BLOCKS", 2048);       // A/B experiments only
    const int CG = C / 8 > 0 ? C / 8 : 1;
    const int rpar = 256 / (CG < 256 ? CG : 256);
    long long r = (M + target - 1) / target;
    const long long floor_rows = 4LL * (rpar > 0 ? rpar : 1);              // >= 2 two-row steps per thread
    if (r < floor_rows) r = floor_rows;
    return (int)r;
}

// Decomposition of the element-wise passes (bn_apply_kernel, bn_bwd_apply_kernel): channel chunks of max_w 8-channel groups,
// and rows per workgroup for ~CELLSEG_EW_BLOCKS workgroups in all with at least `steps` two-row steps per thread.
struct EwSplit { int chunks, cw, rpb; };
inline EwSplit ew_split(long long M, int C, int max_w, int steps) {
    static const int target = cs_env_int_("CELLSEG_EW_BLOCKS", 2048);       // A/B experiments only
    const int CG = C / 8 > 0 ? C / 8 : 1;
    EwSplit s;
    s.cw = CG < max_w ? CG : max_w;                  // (the last chunk may be narrower: a workgroup derives its own width)
    s.chunks = (CG + s.cw - 1) / s.cw;
    const int rpar = 256 / s.cw > 0 ? 256 / s.cw : 1;
    long long row_blocks = target / s.chunks;
    if (row_blocks < 1) row_blocks = 1;
    long long r = (M + row_blocks - 1) / row_blocks;
    const long long floor_rows = 2LL * steps * rpar;
    if (r < floor_rows) r = floor_rows;
    s.rpb = (int)r;
    return s;
}

}  // namespace

#define CS_DISPATCH_T(dtype, CALL_F32, CALL_BF16, NAME)      \
    if (dtype == CS_F32) { CALL_F32; }                       \
    else if (dtype == CS_BF16) { CALL_BF16; }                \
    else { cs_set_error_(NAME ": bad dtype"); return CS_ERR_INVALID_ARG; }

// One thread per (sum, channel): the exact accumulator's totals as doubles (tests, tools; the kernels read the limbs themselves).
__global__ void bn_accum_read_kernel(const double* __restrict__ acc, int C, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * C) out[i] = ex_read(acc, C, i >= C ? 1 : 0, i >= C ? i - C : i);
}

extern "C" size_t cs_bn_accum_words(int C) { return C > 0 ? (size_t)ex_words(C) : 0; }

extern "C" int cs_bn_accum_read(const double* accum, int C, double* out, void* stream) {
    CS_CHECK_ARG(accum && out && C > 0, "bn_accum_read: bad arguments");
    hipLaunchKernelGGL(bn_accum_read_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), accum, C, out);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_bn_partial_fold(const double* partial, int rows, int C, double* stats, void* stream) {
    CS_CHECK_ARG(partial && stats && rows > 0 && C > 0, "bn_partial_fold: bad arguments");
    fold_partial(partial, rows, C, stats, reinterpret_cast<hipStream_t>(stream));
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" size_t cs_bn_partial_workspace(long long M, int C) {
    if (M <= 0 || C <= 0) return 0;
    const int rpb = red_split(M, C).rpb;
    const long long blocks = (M + rpb - 1) / rpb;
    // 0 = no workspace wanted: with this few workgroups cs_bn_stats / cs_bn_bwd_reduce add their sums with fp64 atomics (the ONE place
    // the rule lives: callers allocate what this function says)
    if (!red_wants_partials(C, blocks)) return 0;
    return (size_t)blocks * 2 * (size_t)C * sizeof(double);
}

extern "C" int cs_bn_stats(const void* z, int dtype, long long M, int C, double* stats, double* workspace, void* stream) {
    CS_CHECK_ARG(z && stats && M > 0 && C > 0 && C % 8 == 0, "bn_stats: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const RedSplit sp = red_split(M, C);
    const int rpb = sp.rpb;
    const int blocks = (int)((M + rpb - 1) / rpb);
    if (!red_wants_partials(C, blocks)) workspace = nullptr;      // few contributions: fp64 atomics straight into `stats`, no fold launch
    const dim3 grid((unsigned)blocks, (unsigned)sp.chunks);
    CS_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)z, M, C, stats, rpb, workspace, sp.cw),
                  hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)z, M, C, stats, rpb, workspace, sp.cw),
                  "bn_stats");
    CS_LAUNCH_CHECK();
    if (workspace) {
        fold_partial(workspace, blocks, C, stats, st);
        CS_LAUNCH_CHECK();
    }
    return CS_OK;
}

extern "C" int cs_bn_finalize(const double* stats, long long M, float eps, float momentum, float* running_mean,
                              float* running_var, float* mean_out, float* rstd_out, int C, void* stream) {
    CS_CHECK_ARG(stats && mean_out && rstd_out && M > 0 && C > 0, "bn_finalize: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, M, eps, momentum, running_mean,
                       running_var, mean_out, rstd_out, C);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_bn_apply(const void* z, int dtype, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, const void* residual, int act, void* y, long long M, int C, void* stream) {
    CS_CHECK_ARG(z && y && mean && rstd && M > 0 && C > 0 && C % 8 == 0, "bn_apply: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const EwSplit sp = ew_split(M, C, 256, 2);
    const dim3 grid((unsigned)((M + sp.rpb - 1) / sp.rpb), (unsigned)sp.chunks);
    const BnFinalizeArgs fin{};
    CS_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)z, mean, rstd, gamma, beta,
                                     (const float*)residual, act, (float*)y, M, C, sp.rpb, sp.cw, fin),
                  hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)z, mean, rstd, gamma, beta,
                                     (const bf16_t*)residual, act, (bf16_t*)y, M, C, sp.rpb, sp.cw, fin),
                  "bn_apply");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_bn_apply_stats(const void* z, int dtype, const double* stats, float eps, float momentum, float* running_mean,
                                 float* running_var, const float* gamma, const float* beta, const void* residual, int act, void* y,
                                 float* mean_out, float* rstd_out, long long M, int C, void* stream) {
    CS_CHECK_ARG(z && y && stats && mean_out && rstd_out && M > 0 && C > 0 && C % 8 == 0, "bn_apply_stats: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // (chunks of <= 32 channel groups: a workgroup reads the accumulator words of its own 256 channels only; wide chunks get twice the rows)
    const EwSplit sp = ew_split(M, C, 32, C > 64 ? 4 : 2);
    const dim3 grid((unsigned)((M + sp.rpb - 1) / sp.rpb), (unsigned)sp.chunks);
    const BnFinalizeArgs fin{stats, M, eps, momentum, running_mean, running_var, mean_out, rstd_out};
    CS_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)z, nullptr, nullptr, gamma, beta,
                                     (const float*)residual, act, (float*)y, M, C, sp.rpb, sp.cw, fin),
                  hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)z, nullptr, nullptr, gamma, beta,
                                     (const bf16_t*)residual, act, (bf16_t*)y, M, C, sp.rpb, sp.cw, fin),
                  "bn_apply_stats");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_bn_bwd_reduce(const void* dy, const void* z, int dtype, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, int act, long long M, int C, double* sums, double* workspace,
                                void* stream) {
    CS_CHECK_ARG(dy && z && mean && rstd && sums && M > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const RedSplit sp = red_split(M, C);
    const int rpb = sp.rpb;
    const int blocks = (int)((M + rpb - 1) / rpb);
    if (!red_wants_partials(C, blocks)) workspace = nullptr;      // (as in cs_bn_stats)
    const dim3 grid((unsigned)blocks, (unsigned)sp.chunks);
    CS_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)z, mean,
                                     rstd, gamma, beta, act, M, C, sums, rpb, workspace, sp.cw),
                  hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)z,
                                     mean, rstd, gamma, beta, act, M, C, sums, rpb, workspace, sp.cw),
                  "bn_bwd_reduce");
    CS_LAUNCH_CHECK();
    if (workspace) {
        fold_partial(workspace, blocks, C, sums, st);
        CS_LAUNCH_CHECK();
    }
    return CS_OK;
}

extern "C" int cs_bn_bwd_apply(const void* dy, const void* z, int dtype, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int act, const double* sums, long long M, int C, void* dz,
                               float* dgamma, float* dbeta, void* stream) {
    CS_CHECK_ARG(dy && z && mean && rstd && sums && dz && M > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const EwSplit sp = ew_split(M, C, 32, C > 64 ? 4 : 2);
    const dim3 grid((unsigned)((M + sp.rpb - 1) / sp.rpb), (unsigned)sp.chunks);
    CS_DISPATCH_T(dtype,
                  hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)z, mean,
                                     rstd, gamma, beta, act, sums, M, C, (float*)dz, dgamma, dbeta, sp.rpb, sp.cw),
                  hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)z, mean,
                                     rstd, gamma, beta, act, sums, M, C, (bf16_t*)dz, dgamma, dbeta, sp.rpb, sp.cw),
                  "bn_bwd_apply");
    CS_LAUNCH_CHECK();
    return CS_OK;
}
