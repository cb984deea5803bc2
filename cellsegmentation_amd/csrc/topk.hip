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

constexpr int kMaxRun = 8192;   // 64 KiB of u64 keys

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
    const int n = (int)(seg_off[blockIdx.x + 1] - beg);
    if (n <= 0) return;
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

__device__ __forceinline__ int sel_flag(const int32_t* groups, const int32_t* kpt, long long i, long long T) {
    const long long j = (i + (long long)kpt[i]) % T;
    return groups[i] != groups[j];
}

constexpr int kItems = 2048;   // positions per block in the compaction passes

__global__ __launch_bounds__(256) void sel_count_kernel(const int32_t* groups, const int32_t* kpt, long long T,
                                                        int32_t* __restrict__ block_cnt) {
    const long long base = (long long)blockIdx.x * kItems;
    int c = 0;
    for (int t = threadIdx.x; t < kItems; t += 256) {
        const long long i = base + t;
        if (i < T) c += sel_flag(groups, kpt, i, T);
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

__global__ __launch_bounds__(256) void sel_write_kernel(const int32_t* groups, const int32_t* kpt, long long T,
                                                        const int32_t* __restrict__ block_off, const int64_t* __restrict__ order,
                                                        int64_t* __restrict__ out_idx) {
    // each thread owns 8 CONSECUTIVE positions so output order == position order
    const long long base = (long long)blockIdx.x * kItems + (long long)threadIdx.x * 8;
    int f[8], c = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long long i = base + e;
        f[e] = (i < T) ? sel_flag(groups, kpt, i, T) : 0;
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
    if (max_run > kMaxRun) {
        cs_set_error_("segmented_topk: a group has more than 8192 tiles (LDS sort limit)");
        return CS_ERR_UNSUPPORTED;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t* order = reinterpret_cast<int64_t*>(workspace);
    int32_t* block_cnt = reinterpret_cast<int32_t*>(order + T);
    int np2 = 1;
    while (np2 < max_run) np2 <<= 1;
    hipLaunchKernelGGL(run_sort_kernel, dim3(n_groups), dim3(256), (size_t)np2 * 8, st, probs, seg_offsets, order);
    CS_LAUNCH_CHECK();
    const int nb = (int)((T + kItems - 1) / kItems);
    hipLaunchKernelGGL(sel_count_kernel, dim3(nb), dim3(256), 0, st, groups, k_per_tile, T, block_cnt);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(256), 0, st, block_cnt, nb, out_count);
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_write_kernel, dim3(nb), dim3(256), 0, st, groups, k_per_tile, T, block_cnt, order, out_idx);
    CS_LAUNCH_CHECK();
    return CS_OK;
}
