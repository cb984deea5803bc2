// Adaptive top-k instance selection on device -- bit-exact restatement of inference.py:31-43:
//   order = np.lexsort((probs, groups));  index[i] = groups[i] != groups[(i + k_i) % T];
//   selected = order[index]
// groups (tileIDX) is non-decreasing, so lexsort = a stable ascending sort by prob inside each
// group's run.  One workgroup sorts one run in LDS (bitonic network over 64-bit keys
// {orderable(prob), position-in-run}: the low word makes the network stable, i.e. ties keep
// original order like numpy's lexsort).  Then a 3-kernel stream compaction (count / scan / write)
// evaluates the reference's wrap-around predicate literally and emits order[index].
#include "cs_common.h"

namespace {

constexpr int kMaxRun = 8192;   // longest run the LDS kernel sorts (64 KiB of u64 keys); longer runs: run_sort_global_kernel

__device__ __forceinline__ uint32_t orderable(float f) {
    if (f != f) return 0xffffffffu;          // NaN sorts last (numpy)
    if (f == 0.f) f = 0.f;                   // -0.0 == +0.0 must tie
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void run_sort_kernel(const float* __restrict__ probs, const int64_t* __restrict__ seg_off,
                                                       int64_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);
    const int64_t beg = seg_off[blockIdx.x];
    const int64_t n64 = seg_off[blockIdx.x + 1] - beg;
    if (n64 <= 0 || n64 > kMaxRun) return;       // long runs belong to run_sort_global_kernel
    const int n = (int)n64;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += blockDim.x)
        keys[i] = i < n ? (((unsigned long long)orderable(probs[beg + i]) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[beg + i] = beg + (int64_t)(keys[i] & 0xffffffffull);
}

// Runs longer than kMaxRun (the reference has no limit: inference.py:34-41 sorts whatever the dataset holds, e.g. whole-slide
// tile grids): the same stable sort IN PLACE in global memory, one workgroup per run.  The 64-bit keys live in the run's own
// slice of `order` (8 bytes per tile, like the result), so no extra workspace; the network is the all-ascending form of the
// bitonic sorter (first step of each merge compares with the MIRROR position i ^ (k-1), the rest with i ^ j), in which the
// virtual +inf padding of a non-power-of-two run never moves and is simply skipped.  Every access is an agent-scope atomic
// (L2-served: another wave's store must not be shadowed by a stale L1 line), stages are separated by workgroup barriers.
__global__ __launch_bounds__(1024) void run_sort_global_kernel(const float* __restrict__ probs, const int64_t* __restrict__ seg_off,
                                                               int64_t* __restrict__ order) {
    const int64_t beg = seg_off[blockIdx.x];
    const int64_t n = seg_off[blockIdx.x + 1] - beg;
    if (n <= kMaxRun) return;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(order + beg);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
        __hip_atomic_store(keys + i, ((unsigned long long)orderable(probs[beg + i]) << 32) | (unsigned long long)(unsigned)i, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int64_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    auto pass = [&](int64_t mask) {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const int64_t p = i ^ mask;
            if (p > i && p < n) {
                const unsigned long long a = __hip_atomic_load(keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long b = __hip_atomic_load(keys + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a > b) {
                    __hip_atomic_store(keys + i, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(keys + p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    };
    for (int64_t k = 2; k <= np2; k <<= 1) {
        pass(k - 1);
        for (int64_t j = k >> 2; j > 0; j >>= 1) pass(j);
    }
    // keys -> tile indices (each element is rewritten by the thread that reads it)
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long kv = __hip_atomic_load(keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        order[beg + i] = beg + (int64_t)(kv & 0xffffffffull);
    }
}

int launch_run_sort(const float* probs, const int64_t* seg_offsets, int n_groups, int max_run, int64_t* order, hipStream_t st) {
    const int lds_run = max_run < kMaxRun ? max_run : kMaxRun;
    int np2 = 1;
    while (np2 < lds_run) np2 <<= 1;
    hipLaunchKernelGGL(run_sort_kernel, dim3(n_groups), dim3(256), (size_t)np2 * 8, st, probs, seg_offsets, order);
    CS_LAUNCH_CHECK();
    if (max_run > kMaxRun) {
        hipLaunchKernelGGL(run_sort_global_kernel, dim3(n_groups), dim3(1024), 0, st, probs, seg_offsets, order);
        CS_LAUNCH_CHECK();
    }
    return CS_OK;
}

// Selection predicate over SORTED positions.  kpt != NULL: the reference's wrap-around top-k test (inference.py:38-41);
// kpt == NULL: probability of the tile at sorted position i above `thr` (train_seg.py:243 `prob > threshold`, evaluate.py:17).
struct SelPred {
    const int32_t* groups;
    const int32_t* kpt;
    const float* probs;
    const int64_t* order;
    float thr;
};
__device__ __forceinline__ int sel_flag(const SelPred& q, long long i, long long T) {
    if (q.kpt) {
        const long long j = (i + (long long)q.kpt[i]) % T;
        return q.groups[i] != q.groups[j];
    }
    return q.probs[q.order[i]] > q.thr;
}

constexpr int kItems = 2048;   // positions per block in the compaction passes

__global__ __launch_bounds__(256) void sel_count_kernel(SelPred q, long long T, int32_t* __restrict__ block_cnt) {
    const long long base = (long long)blockIdx.x * kItems;
    int c = 0;
    for (int t = threadIdx.x; t < kItems; t += 256) {
        const long long i = base + t;
        if (i < T) c += sel_flag(q, i, T);
    }
    __shared__ int red[4];
    c = (int)wave_sum((float)c);   // <= 2048: exact in fp32
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// single workgroup exclusive scan of block counts (nb <= a few thousand)
__global__ __launch_bounds__(256) void sel_scan_kernel(int32_t* __restrict__ block_cnt, int nb, int64_t* __restrict__ out_count) {
    __shared__ long long carry;
    __shared__ int part[256];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? block_cnt[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        // Hillis-Steele inclusive scan in LDS
        for (int off = 1; off < 256; off <<= 1) {
            const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const long long excl = carry + part[threadIdx.x] - v;
        if (i < nb) block_cnt[i] = (int32_t)excl;
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_count = carry;
}

__global__ __launch_bounds__(256) void sel_write_kernel(SelPred q, long long T, const int32_t* __restrict__ block_off,
                                                        const int64_t* __restrict__ order, int64_t* __restrict__ out_idx) {
    // each thread owns 8 CONSECUTIVE positions so output order == position order
    const long long base = (long long)blockIdx.x * kItems + (long long)threadIdx.x * 8;
    int f[8], c = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long long i = base + e;
        f[e] = (i < T) ? sel_flag(q, i, T) : 0;
        c += f[e];
    }
    __shared__ int part[256];
    part[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    long long w = (long long)block_off[blockIdx.x] + part[threadIdx.x] - c;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (f[e]) out_idx[w++] = order[base + e];
}

}  // namespace

extern "C" size_t cs_segmented_topk_workspace(long long T) {
    const long long nb = (T + kItems - 1) / kItems;
    size_t bytes = (size_t)T * sizeof(int64_t);            // order
    bytes += ((size_t)nb * sizeof(int32_t) + 15) & ~(size_t)15;
    return bytes + 16;
}

extern "C" int cs_segmented_topk(const float* probs, const int32_t* groups, const int32_t* k_per_tile,
                                 const int64_t* seg_offsets, int n_groups, int max_run, long long T, int64_t* out_idx,
                                 int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    CS_CHECK_ARG(probs && groups && k_per_tile && seg_offsets && out_idx && out_count && workspace, "segmented_topk: NULL argument");
    CS_CHECK_ARG(T > 0 && n_groups > 0 && max_run > 0, "segmented_topk: empty input");
    CS_CHECK_ARG(workspace_bytes >= cs_segmented_topk_workspace(T), "segmented_topk: workspace too small");
    CS_CHECK_ARG(max_run < (1 << 30), "segmented_topk: run length out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t* order = reinterpret_cast<int64_t*>(workspace);
    int32_t* block_cnt = reinterpret_cast<int32_t*>(order + T);
    const int rc = launch_run_sort(probs, seg_offsets, n_groups, max_run, order, st);
    if (rc != CS_OK) return rc;
    const int nb = (int)((T + kItems - 1) / kItems);
    const SelPred q{groups, k_per_tile, nullptr, nullptr, 0.f};
    hipLaunchKernelGGL(sel_count_kernel, dim3(nb), dim3(256), 0, st, q, T, block_cnt);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(256), 0, st, block_cnt, nb, out_count);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_write_kernel, dim3(nb), dim3(256), 0, st, q, T, block_cnt, order, out_idx);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// ---- the steps either side of the top-k --------------------------------------------------------------
namespace {

// evaluate.py:8-27: pred = prob > threshold at every sorted position; real = 1 on the union of the per-image intervals
// [end_g - c_g*tiles_per_pos, end_g) -- the host passes, per group, the smallest interval start over this and all later groups
// (a count larger than an image's own run spills into the previous image exactly as the reference's slice assignment does).
__global__ __launch_bounds__(256) void evaluate_tile_kernel(const float* __restrict__ probs, const int64_t* __restrict__ order,
                                                            const int32_t* __restrict__ groups, const int64_t* __restrict__ pos_from,
                                                            float thr, long long T, unsigned long long* __restrict__ counts) {
    unsigned long long neq = 0, fp = 0, fn = 0, r1 = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < T; i += (long long)gridDim.x * 256) {
        const int pred = probs[order[i]] > thr;
        const int real = i >= pos_from[groups[i]];
        neq += pred != real;
        fp += pred && !real;
        fn += !pred && real;
        r1 += real;
    }
    // < 2^24 per thread in any realistic run, but keep it exact: integer wave reduction through shuffles
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        neq += __shfl_xor(neq, off, 64); fp += __shfl_xor(fp, off, 64); fn += __shfl_xor(fn, off, 64); r1 += __shfl_xor(r1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + 0, neq); atomicAdd(counts + 1, fp); atomicAdd(counts + 2, fn); atomicAdd(counts + 3, r1);
    }
}

// utils/image_processing.py:92-98: every selected tile paints a tile_size x tile_size block of ones into its image's mask
// (clipped at the borders like the numpy slice assignment).  One workgroup per selected tile.
__global__ __launch_bounds__(256) void paint_tiles_kernel(const int64_t* __restrict__ sel, long long n_sel, const int32_t* __restrict__ groups,
                                                          const int32_t* __restrict__ xy, int tile, int H, int W, uint8_t* __restrict__ masks) {
    const long long s = blockIdx.x;
    if (s >= n_sel) return;
    const long long t = sel[s];
    const int x0 = xy[2 * t], y0 = xy[2 * t + 1];            // (row, column) of the tile's upper-left corner
    uint8_t* m = masks + (long long)groups[t] * H * W;
    for (int e = threadIdx.x; e < tile * tile; e += 256) {
        const int r = x0 + e / tile, c = y0 + e % tile;
        if (r >= 0 && r < H && c >= 0 && c < W) m[(long long)r * W + c] = 1;
    }
}

// dataset/dataset.py:178-199 after the shuffle: drop the first n_excess entries whose label == flag, keep everything else in
// order.  One workgroup walks the array in chunks of 2048 carrying both running counts (arrays here are <= a few 100k entries).
__global__ __launch_bounds__(256) void prune_kernel(const int32_t* __restrict__ label, long long n, int flag, long long n_excess,
                                                    int64_t* __restrict__ kept, int64_t* __restrict__ kept_count) {
    __shared__ int part_f[256], part_k[256];
    __shared__ long long carry_f, carry_k;
    if (threadIdx.x == 0) { carry_f = 0; carry_k = 0; }
    __syncthreads();
    for (long long base = 0; base < n; base += 2048) {
        int isf[8], cf = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long i = base + (long long)threadIdx.x * 8 + e;
            isf[e] = (i < n) && (label[i] == flag);
            cf += isf[e];
        }
        part_f[threadIdx.x] = cf;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = threadIdx.x >= off ? part_f[threadIdx.x - off] : 0;
            __syncthreads();
            part_f[threadIdx.x] += add;
            __syncthreads();
        }
        long long rank = carry_f + part_f[threadIdx.x] - cf;      // flagged entries before this thread's first position
        int keep[8], ck = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long i = base + (long long)threadIdx.x * 8 + e;
            keep[e] = (i < n) && !(isf[e] && rank < n_excess);
            rank += isf[e];
            ck += keep[e];
        }
        part_k[threadIdx.x] = ck;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = threadIdx.x >= off ? part_k[threadIdx.x - off] : 0;
            __syncthreads();
            part_k[threadIdx.x] += add;
            __syncthreads();
        }
        long long w = carry_k + part_k[threadIdx.x] - ck;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (keep[e]) kept[w++] = base + (long long)threadIdx.x * 8 + e;
        __syncthreads();
        if (threadIdx.x == 255) { carry_f += part_f[255]; carry_k += part_k[255]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *kept_count = carry_k;
}

}  // namespace

extern "C" int cs_segmented_order(const float* probs, const int64_t* seg_offsets, int n_groups, int max_run, long long T,
                                  int64_t* order, void* stream) {
    CS_CHECK_ARG(probs && seg_offsets && order && T > 0 && n_groups > 0 && max_run > 0, "segmented_order: bad arguments");
    CS_CHECK_ARG(max_run < (1 << 30), "segmented_order: run length out of range");
    return launch_run_sort(probs, seg_offsets, n_groups, max_run, order, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int cs_threshold_select(const float* probs, const int64_t* order, long long T, float threshold, int64_t* out_idx,
                                   int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    CS_CHECK_ARG(probs && order && out_idx && out_count && workspace && T > 0, "threshold_select: bad arguments");
    CS_CHECK_ARG(workspace_bytes >= cs_segmented_topk_workspace(T), "threshold_select: workspace too small (cs_segmented_topk_workspace)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t* block_cnt = reinterpret_cast<int32_t*>(workspace);
    const int nb = (int)((T + kItems - 1) / kItems);
    const SelPred q{nullptr, nullptr, probs, order, threshold};
    hipLaunchKernelGGL(sel_count_kernel, dim3(nb), dim3(256), 0, st, q, T, block_cnt);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(256), 0, st, block_cnt, nb, out_count);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_write_kernel, dim3(nb), dim3(256), 0, st, q, T, block_cnt, order, out_idx);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_evaluate_tile_counts(const float* probs, const int64_t* order, const int32_t* groups, const int64_t* pos_from,
                                       float threshold, long long T, unsigned long long* counts4, void* stream) {
    CS_CHECK_ARG(probs && order && groups && pos_from && counts4 && T > 0, "evaluate_tile_counts: bad arguments");
    long long nb = (T + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(evaluate_tile_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), probs, order, groups,
                       pos_from, threshold, T, counts4);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_paint_tile_masks(const int64_t* selected, long long n_selected, const int32_t* groups, const int32_t* tile_xy,
                                   int tile_size, int H, int W, uint8_t* masks, void* stream) {
    CS_CHECK_ARG(groups && tile_xy && masks && tile_size > 0 && H > 0 && W > 0 && n_selected >= 0, "paint_tile_masks: bad arguments");
    if (n_selected == 0) return CS_OK;
    CS_CHECK_ARG(selected != nullptr, "paint_tile_masks: NULL selection");
    hipLaunchKernelGGL(paint_tiles_kernel, dim3((unsigned)n_selected), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), selected, n_selected,
                       groups, tile_xy, tile_size, H, W, masks);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_prune_excess(const int32_t* labels, long long n, int flag, long long n_excess, int64_t* kept, int64_t* kept_count,
                               void* stream) {
    CS_CHECK_ARG(labels && kept && kept_count && n > 0 && n_excess >= 0, "prune_excess: bad arguments");
    hipLaunchKernelGGL(prune_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), labels, n, flag, n_excess, kept, kept_count);
    CS_LAUNCH_CHECK();
    return CS_OK;
}
