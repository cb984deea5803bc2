// Implicit-GEMM convolution for gfx950 (CDNA4): forward + data-gradient share one kernel, the
// weight gradient has its own.  Replaces the ATen conv2d / conv2d_backward dispatched by
// model/resnet.py:20,23,51,53,55,111,183,198,164 (and their autograd backward).
//
// GEMM view (fwd / dgrad):  D[m][o] = sum_q A[m][q] * B[o][q]
//   m  = destination pixel (n, dy, dx)            -> rows of the NHWC destination tensor
//   o  = destination channel
//   q  = (tap, source-channel chunk), 16 bytes of K per chunk
//   A  = gathered on the fly from the NHWC source tensor (im2col never materialised)
//   B  = weights stored [o][tap][c] (K contiguous)
// The source coordinate of tap (kh,kw) for destination (dy,dx) is
//   t = d*mul + off0 + sgn*k;  valid iff t % div == 0 and 0 <= t/div < extent
// fwd:   mul=stride, off0=-pad, sgn=+1, div=1        (source = x, dest = y)
// dgrad: mul=1,      off0=+pad, sgn=-1, div=stride   (source = dy, dest = dx)
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2x2), K-step = 8 chunks (128 B per
// row), register-staged double-buffered LDS, XOR-swizzled 16-byte slots so ds_read_b128 of an
// MFMA operand column is bank-conflict free.  MFMA: v_mfma_f32_32x32x16_bf16 (bf16) or the exact
// v_mfma_f32_32x32x2_f32 (fp32 parity mode); one 16-byte chunk per lane feeds 1 resp. 4 MFMAs.
// Epilogue: accumulators -> LDS (fp32) -> 8 channels per thread, fully vectorised
// scale/shift/residual/activation/mask + store + optional per-channel statistics.
#include "cs_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

struct IgemmParams {
    const void* src;
    const void* wgt;
    void* dst;
    const float* scale;
    const float* shift;
    const void* residual;
    const void* mask;
    // one bit per stored element instead of a 16-bit mask operand, CHANNEL-BLOCK-MAJOR (round 5; conv_v2.hip TileOffs): byte
    // ((c / 32) * bits_M + pixel) * 4 + (c / 8) % 4 holds which of the 8 outputs c .. c + 7 of that pixel are > 0 (written next to a ReLU
    // output); bits_in plays the role of `mask` in a data gradient (1 byte per 16).  bits_M = pixels of the whole destination tensor.
    unsigned char* bits_out;
    const unsigned char* bits_in;
    long long bits_M;
    float* slab;      // nullable fp32 [M tiles][2][NOUT]: per-workgroup column sums / sums of squares of the
                      // stored output (no atomics; cs_slab_reduce folds the rows afterwards)
    double* stat_atomic;   // with slab != NULL: add the workgroup's sums to this fp64 [2][NOUT] instead of storing its partial row (launches
                           // with few pixel tiles: <= 512 atomics per address, and the slab_reduce launch -- one per train-mode-BN convolution,
                           // 20-53 per step on the ResNet-18 counter / EfficientNet-B3 / the segmentation encoder -- disappears)
    int SH, SW, SC;   // source extents, stored channels
    int DH, DW;       // destination spatial extents
    int NOUT;         // destination channels
    int R, S;
    int mul, div, off0, off0x, sgn;   // off0 = y offset, off0x = x offset
    int mulx;         // x stride of the source walk (== mul except for the pixel-paired stem: rows stride 2, pixel pairs stride 1)
    int act;
    // weight-tap walk: filter tap of loop tap (kh,kw) = (wk0y + wkstep*kh, wk0x + wkstep*kw) in an S_full-wide filter
    int wk0y, wk0x, wkstep, S_full;
    // destination sub-grid (strided data-gradient parity classes): pixel (n, a*dst_step+dst_oy, b*dst_step+dst_ox) of a
    // DHF x DWF tensor; dst_step == 1 means rows map to pixels one-to-one
    int dst_step, dst_oy, dst_ox, DHF, DWF;
    // grouped convolution as slab-dense GEMM: the 64-wide destination-channel tile n0 contracts only over source
    // channels [n0, n0+64) (block-diagonal weights inside the slab); SCc is then 8 while SC stays the pixel stride
    int cslab;
    long long M;      // N*DH*DW
    long long src_pixels;   // N*SH*SW
    // strided data gradient: up to 4 destination parity classes in ONE launch; workgroup -> class by block0, the class's
    // geometry replaces the fields of the same name (LDS-DMA kernel only)
    int ncls;
    struct ClassOv { int DH, DW, off0, off0x, R, S, wk0y, wk0x, Qtot, dst_oy, dst_ox, block0, row0; long long M; } cls[4];
    int Qtot;         // R*S*SCc (taps walked by this launch)
    int wrow_chunks;  // chunks per weight row (full filter)
    int SCc;          // SC / chunk
};

__device__ __forceinline__ int swz(int row, int c) { return row * 8 + (c ^ ((row >> 1) & 7)); }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
        union { uint4 u; bf16x8 v; } ca, cb;
        ca.u = a; cb.u = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca.v, cb.v, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// ---- hand-issued LDS-DMA -------------------------------------------------------------------------------------
// `buffer_load_dwordx4 ... offen lds` written as inline asm ON PURPOSE.  Through the builtin, hipcc (ROCm 7.2) treats every
// DMA as a pending LDS store that may alias any later LDS read and puts `s_waitcnt vmcnt(0)` in front of the first ds_read
// of each K-step (and of every __syncthreads()): the stage requested a moment earlier is drained before the current one is
// multiplied, i.e. NO copy/compute overlap inside a workgroup (seen in the .s of both DMA kernels; MFMA busy 23 %).  As asm
// the DMA is invisible to that pass; ordering is ours: counted `s_waitcnt vmcnt(N)` + `dma_barrier()` before a stage is read.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xffffu);      // stride 0: raw buffer, offsets are bytes
    r.z = (int)bytes;                                 // range check: offsets >= bytes read as zero
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
// 64 lanes x 16 bytes -> 1 KiB of LDS at `lds_base` (wave-uniform) + 16*lane; `voff` = per-lane byte offset into the buffer
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned lds_base, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}
// the same with a scalar byte offset added to every lane's address (not part of the range check: `voff` alone decides
// in / out of range) and a compile-time LDS displacement folded into the M0 write
template <int LDS_IMM>
__device__ __forceinline__ void dma16s(const i32x4& rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LDS_IMM) : "memory", "scc");
}
// workgroup barrier that neither drains the VM queue nor lets the compiler move LDS accesses across it
__device__ __forceinline__ void dma_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Epilogue operands (residual / add, mask) of one thread, fetched BEFORE the K loop so their latency hides behind it
// (bf16 only: one 16-byte chunk = the thread's 8 channels).  Same row / channel-group ownership as igemm_epilogue.
template <int BM, int BN> struct EpiRegs {
    static constexpr int ITERS = (BM / 2) * BN / 8 / 256;
    uint4 res[2 * ITERS];
    uint4 msk[2 * ITERS];
    unsigned mbits[2 * ITERS];
};

template <typename T, int BM, int BN>
__device__ __forceinline__ void epi_prefetch(const IgemmParams& p, long long m0, int n0, EpiRegs<BM, BN>& e) {
    static_assert(sizeof(T) == 2, "epilogue prefetch is a bf16 path");
    constexpr int CG = BN / 8, HROWS = BM / 2, ITERS = EpiRegs<BM, BN>::ITERS;
    const int tid = threadIdx.x;
    const int o = n0 + (tid % CG) * 8;
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
    const T* __restrict__ msk = reinterpret_cast<const T*>(p.mask);
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const long long m = m0 + half * HROWS + it * (256 / CG) + tid / CG;
            const bool ok = m < p.M && o < p.NOUT;
            const long long off = m * p.NOUT + o;          // identity row -> pixel mapping only (dst_step == 1)
            e.res[half * ITERS + it] = (res && ok) ? *reinterpret_cast<const uint4*>(res + off) : make_uint4(0, 0, 0, 0);
            e.msk[half * ITERS + it] = (msk && ok) ? *reinterpret_cast<const uint4*>(msk + off) : make_uint4(0, 0, 0, 0);
            e.mbits[half * ITERS + it] = (p.bits_in && ok) ? (unsigned)p.bits_in[((long long)(o >> 5) * p.bits_M + m) * 4 + ((o >> 3) & 3)] : 0u;
        }
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& a, float (&v)[8]) {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}

template <typename T, int BM, int BN, bool PF = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x16 (&acc)[BM / 64][BN / 64], unsigned char* smem_raw,
                                               long long m0, int n0, long long slab_row, const EpiRegs<BM, BN>* pf = nullptr) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    // ---- epilogue: accumulators -> LDS fp32 [BM][BN] (aliases the staging buffers) ----
    // Two passes over the tile rows (wm = 0 half, then wm = 1 half) so the fp32 staging image is only
    // (BM/2) x BN: halves the epilogue's LDS footprint -> more workgroups per CU on the 1-K-step (HBM-bound) convs.
    float* Cs = reinterpret_cast<float*>(smem_raw);
    constexpr int CG = BN / 8;              // 8-channel groups per tile row
    constexpr int HROWS = BM / 2;
    constexpr int ITERS = HROWS * BN / 8 / 256;
    const int cg = tid % CG;
    const int o = n0 + cg * 8;
    const bool ook = o < p.NOUT;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = (p.scale && ook) ? p.scale[o + e] : 1.f;
        sh[e] = (p.shift && ook) ? p.shift[o + e] : 0.f;
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    T* __restrict__ dst = reinterpret_cast<T*>(p.dst);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
    const T* __restrict__ msk = reinterpret_cast<const T*>(p.mask);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const int col = wn * (BN / 2) + j * 32 + l31;
                        Cs[row * BN + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        // element offset of this thread's first row of the half; identity row->pixel mapping advances by a constant per iteration
        const long long row0 = m0 + half * HROWS + tid / CG;
        const long long off0 = row0 * p.NOUT + o;
        const long long off_step = (long long)(256 / CG) * p.NOUT;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int hrow = it * (256 / CG) + tid / CG;
            const long long m = row0 + it * (256 / CG);
            if (m < p.M && ook) {
                float v[8];
                const float4 c0 = *reinterpret_cast<const float4*>(Cs + hrow * BN + cg * 8);
                const float4 c1 = *reinterpret_cast<const float4*>(Cs + hrow * BN + cg * 8 + 4);
                v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
                v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
                long long off = off0 + it * off_step;
                long long pix = m;                       // destination pixel (row-major over the whole tensor): the bit planes' index
                if (p.dst_step != 1) {
                    const long long img = m / ((long long)p.DH * p.DW);
                    const int rem = (int)(m - img * (long long)p.DH * p.DW);
                    const int a = rem / p.DW;
                    const int b = rem - a * p.DW;
                    pix = (img * p.DHF + (long long)a * p.dst_step + p.dst_oy) * p.DWF + (long long)b * p.dst_step + p.dst_ox;
                    off = pix * p.NOUT + o;
                }
                const long long boff = ((long long)(o >> 5) * p.bits_M + pix) * 4 + ((o >> 3) & 3);     // this thread's byte of a bit plane
                if (p.scale) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += sh[e];
                }
                if (res) {
                    float r8[8];
                    if constexpr (PF) unpack_bf16x8(pf->res[half * ITERS + it], r8);
                    else load8<T>(res + off, r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                }
                if (p.act == CS_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                } else if (p.act == CS_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
                }
                if (p.bits_in) {
                    unsigned mb;
                    if constexpr (PF) mb = pf->mbits[half * ITERS + it];
                    else mb = p.bits_in[boff];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
                } else if (msk) {
                    float k8[8];
                    if constexpr (PF) unpack_bf16x8(pf->msk[half * ITERS + it], k8);
                    else load8<T>(msk + off, k8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = k8[e] > 0.f ? v[e] : 0.f;
                }
                unsigned mb = 0;
                if constexpr (sizeof(T) == 2) {
                    // pack once; the bits come from the packed words (integer compare of a sign-magnitude half shifted to the top)
                    uint4 o;
                    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                    *reinterpret_cast<uint4*>(dst + off) = o;
                    if (p.bits_out) {
                        const unsigned wds[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            mb |= ((int)(wds[i] << 16) > 0 ? 1u : 0u) << (2 * i);
                            mb |= ((int)(wds[i] & 0xffff0000u) > 0 ? 1u : 0u) << (2 * i + 1);
                        }
                    }
                } else {
                    store8<T>(dst + off, v);
                    if (p.bits_out) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) mb |= (v[e] > 0.f ? 1u : 0u) << e;
                    }
                }
                if (p.bits_out) {
                    // four neighbouring lanes own four consecutive bytes of the same row (NOUT % 32 == 0 keeps the quad inside
                    // one row and makes this branch quad-uniform): one aligned dword store instead of four byte stores
                    // (DPP row shifts: lane i takes lane i+1 / i+2 of its 16-lane row; a ds_bpermute shuffle costs an LDS round trip)
                    mb |= (unsigned)__builtin_amdgcn_mov_dpp((int)mb, 0x101, 0xf, 0xf, true) << 8;      // row_shl:1
                    mb |= (unsigned)__builtin_amdgcn_mov_dpp((int)mb, 0x102, 0xf, 0xf, true) << 16;     // row_shl:2
                    if ((tid & 3) == 0) *reinterpret_cast<unsigned*>(p.bits_out + boff) = mb;
                }
                if (p.slab) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // statistics are those of the STORED (rounded) values
                        const float w = to_f32<T>(from_f32<T>(v[e]));
                        s1[e] += w;
                        s2[e] += w * w;
                    }
                }
            }
        }
    }
    if (p.slab) {
        // fold the 256/CG threads that own the same 8 channels through LDS (Cs is dead now)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);        // [256][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
        __syncthreads();
        // one thread per (sum, channel) column -- 16 * CG = 2 * BN of them -- folding its 256 / CG values in thread order (the same sums
        // as when CG lanes walked 16 columns each: 8-16 lanes x 256-512 dependent LDS reads at the end of EVERY one-shot workgroup, on
        // the train-mode-BN convolutions of EfficientNet and of the segmentation encoder)
        float* row = p.slab + slab_row * 2 * p.NOUT;
        for (int col = tid; col < CG * 16; col += 256) {
            const int cgl = col >> 4, j = col & 15;
            const int oc = n0 + cgl * 8;
            if (oc < p.NOUT) {
                float t = 0.f;
                for (int r = cgl; r < 256; r += CG) t += red[r * 16 + j];
                const int o_ = (j < 8 ? 0 : p.NOUT) + oc + (j & 7);
                if (p.stat_atomic) ex_add(p.stat_atomic, p.NOUT, j < 8 ? 0 : 1, oc + (j & 7), (double)t);      // exact: any arrival order, same bits
                else row[o_] = t;
            }
        }
    }
}

// MODE 0: 1x1, stride 1, no padding  -> plain GEMM rows (no tap logic at all)
// MODE 1: any filter with div == 1     -> per-row tap-validity bitmask + base offset, 1 test + 1 add per load
// MODE 2: div > 1 (strided data-gradient): general coordinate arithmetic per load
template <typename T, int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
    constexpr int CE = Elem<T>::kChunk;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 8;  // uint4 per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* smem = reinterpret_cast<uint4*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    // XCD-aware tile order (1-D grid): workgroup ids go round-robin over the 8 XCDs, so id = 8*slot + xcd.  All N tiles of
    // one M tile run back-to-back on ONE XCD: the second N tile finds the source rows in that XCD's L2 and the two
    // half-row output writes meet there before they go to HBM.
    const int n_ntiles = (p.NOUT + BN - 1) / BN;
    const unsigned slot = blockIdx.x >> 3;
    const long long mtile = (long long)(slot / n_ntiles) * 8 + (blockIdx.x & 7);
    const long long m0 = mtile * BM;
    const int n0 = (int)(slot % n_ntiles) * BN;
    if (m0 >= p.M) return;     // grid is padded to 8 M tiles per round

    // ---- per-thread gather bookkeeping: fixed rows, walking (kh,kw,cc) ----
    const int lc = tid & 7;    // chunk column inside the K-step
    const int lr = tid >> 3;   // 0..31
    // MODE 0/1: rbase = element offset of the row's tap-(0,0) source pixel, vmask bit t = tap t is in bounds
    // MODE 2  : rbase = pixel index of (n,0,0) or -1; ty/tx = destination coordinate terms
    long long rbase[AI];
    unsigned long long vmask[AI];
    int ty[AI], tx[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const long long m = m0 + lr + 32 * i;
        rbase[i] = -1; vmask[i] = 0; ty[i] = 0; tx[i] = 0;
        if (m < p.M) {
            if constexpr (MODE == 0) {
                rbase[i] = m * p.SC;
                vmask[i] = 1;
            } else {
                const long long img = m / ((long long)p.DH * p.DW);
                const int rem = (int)(m - img * (long long)p.DH * p.DW);
                const int dy = rem / p.DW;
                const int dx = rem - dy * p.DW;
                const int y0 = dy * p.mul + p.off0;
                const int x0 = dx * p.mulx + p.off0x;
                if constexpr (MODE == 1) {
                    rbase[i] = ((img * p.SH + y0) * (long long)p.SW + x0) * p.SC;
                    unsigned long long mk = 0;
                    for (int a = 0; a < p.R; ++a) {
                        const int y = y0 + p.sgn * a;
                        if (y < 0 || y >= p.SH) continue;
                        for (int b = 0; b < p.S; ++b) {
                            const int x = x0 + p.sgn * b;
                            if (x >= 0 && x < p.SW) mk |= 1ull << (a * p.S + b);
                        }
                    }
                    vmask[i] = mk;
                } else {
                    rbase[i] = img * (long long)p.SH * p.SW;
                    ty[i] = y0; tx[i] = x0;
                }
            }
        }
    }
    const int slab0 = p.cslab ? (n0 / 64) * (64 / CE) : 0;      // first source chunk of this tile's channel slab
    // walking position of this thread's chunk column in K space
    int q = lc;
    int tap = q / p.SCc;
    int cc = q - tap * p.SCc;
    int kh = tap / p.S;
    int kw = tap - kh * p.S;

    const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);
    const long long wrow_elems = (long long)p.wrow_chunks * CE;

    uint4 ra[AI], rb[BI];
    auto wq = [&]() -> int {      // chunk index inside a weight row for the current (kh,kw,cc)
        if constexpr (MODE == 0) return q;
        return ((p.wk0y + p.wkstep * kh) * p.S_full + p.wk0x + p.wkstep * kw) * p.SCc + cc;
    };
    auto gload = [&]() {
        const bool qok = q < p.Qtot;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (qok && vmask[i]) v = *reinterpret_cast<const uint4*>(src + rbase[i] + q * CE);
                ra[i] = v;
            }
        } else if constexpr (MODE == 1) {
            const int t = kh * p.S + kw;
            const long long toff = ((long long)(p.sgn * kh) * p.SW + p.sgn * kw) * p.SC + (cc + slab0) * CE;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (qok && ((vmask[i] >> t) & 1ull)) v = *reinterpret_cast<const uint4*>(src + rbase[i] + toff);
                ra[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (qok && rbase[i] >= 0) {
                    int y = ty[i] + p.sgn * kh;
                    int x = tx[i] + p.sgn * kw;
                    bool ok = (y >= 0) && (x >= 0) && (y % p.div == 0) && (x % p.div == 0);
                    y /= p.div; x /= p.div;
                    ok = ok && (y < p.SH) && (x < p.SW);
                    if (ok) {
                        const long long pix = rbase[i] + (long long)y * p.SW + x;
                        v = *reinterpret_cast<const uint4*>(src + pix * p.SC + cc * CE);
                    }
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int o = n0 + lr + 32 * j;
            if (qok && o < p.NOUT) v = *reinterpret_cast<const uint4*>(wgt + (long long)o * wrow_elems + (long long)wq() * CE);
            rb[j] = v;
        }
    };
    auto advance = [&]() {
        q += 8;
        if constexpr (MODE != 0) {
            cc += 8;
            while (cc >= p.SCc) {
                cc -= p.SCc;
                ++kw;
                if (kw == p.S) { kw = 0; ++kh; }
            }
        }
    };
    auto lstore = [&](int buf) {
        uint4* As = smem + buf * STAGE;
        uint4* Bs = As + BM * 8;
#pragma unroll
        for (int i = 0; i < AI; ++i) As[swz(lr + 32 * i, lc)] = ra[i];
#pragma unroll
        for (int j = 0; j < BI; ++j) Bs[swz(lr + 32 * j, lc)] = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.Qtot + 7) / 8;
    gload();
    advance();
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = (ks + 1) < nk;
        if (more) { gload(); advance(); }
        const uint4* As = smem + (ks & 1) * STAGE;
        const uint4* Bs = As + BM * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = 2 * kk + hh;
            uint4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[swz(wm * (BM / 2) + i * 32 + l31, c)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[swz(wn * (BN / 2) + j * 32 + l31, c)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
        }
        if (more) lstore((ks + 1) & 1);
        __syncthreads();
    }

    igemm_epilogue<T, BM, BN>(p, acc, smem_raw, m0, n0, mtile);
}


// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (default): global -> LDS with `buffer_load_dwordx4 ... lds`, no staging VGPRs and
// no ds_write pass (the register-staged version spent ~416 LDS-store cycles per 512 MFMA cycles).
//  * one wave-instruction fills 8 tile rows x 128 B = 1 KiB of contiguous LDS (lane-linear
//    destination); the XOR swizzle is therefore applied to the per-lane SOURCE chunk, the MFMA
//    operand reads use the same involution (guide rule 21);
//  * padding taps, M/N tails and K tails are lanes whose buffer offset is out of range: the
//    hardware range check returns zeros, which the DMA writes to LDS -- no branches, no masks;
//  * needs every operand < 2 GiB (32-bit buffer offsets); larger tensors use the register path.
// ---------------------------------------------------------------------------------------------
//  * PF (bf16, > 1 K-step, identity destination mapping): the residual / mask operands of the epilogue are fetched into
//    registers before the K loop.  Multi-K-step launches hold 2 LDS stages = 2 workgroups per CU, so the register budget
//    is 256 per wave anyway; without this the fat epilogue of the wide-output layers (2 operand reads + 1 write per
//    output element) ran strictly after the K loop with nothing else in flight.
//  * UNI (MODE 0 always; MODE 1 when the tap is the same for every lane of a K-step, i.e. C % 64 == 0, and R*S <= 32): the K walk
//    lives in scalar registers.  Weights and MODE-0 pixels are fetched with a lane-constant voffset + a per-K-step scalar
//    soffset (zero VALU per DMA); MODE-1 pixels add the tap offset and test one bit of a 32-bit validity mask (4 VALU per DMA).
//    The lane-by-lane walk it replaces cost ~8 VALU + 6 SALU per MFMA (PMC), more issue slots than the MFMAs themselves.
template <typename T, int BM, int BN, int MODE, bool PF = false, bool UNI = false>
__global__ __launch_bounds__(256, PF ? 2 : 4) void igemm_dma_kernel(IgemmParams p_in, unsigned src_bytes, unsigned wgt_bytes) {
    IgemmParams p = p_in;
    unsigned bid = blockIdx.x;
    int slab_row0 = 0;
    if (p_in.ncls) {
        // wave-uniform selection with static indices only (a runtime index into the by-value argument would go through scratch)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < p_in.ncls && blockIdx.x >= (unsigned)p_in.cls[i].block0) {
                p.DH = p_in.cls[i].DH; p.DW = p_in.cls[i].DW; p.off0 = p_in.cls[i].off0; p.off0x = p_in.cls[i].off0x;
                p.R = p_in.cls[i].R; p.S = p_in.cls[i].S; p.wk0y = p_in.cls[i].wk0y; p.wk0x = p_in.cls[i].wk0x;
                p.Qtot = p_in.cls[i].Qtot; p.dst_oy = p_in.cls[i].dst_oy; p.dst_ox = p_in.cls[i].dst_ox; p.M = p_in.cls[i].M;
                bid = blockIdx.x - (unsigned)p_in.cls[i].block0;
                slab_row0 = p_in.cls[i].row0;
            }
    }
    constexpr int ES = (int)sizeof(T);
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 8;  // uint4 per stage
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* smem = reinterpret_cast<uint4*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    // XCD-aware tile order (1-D grid): workgroup ids go round-robin over the 8 XCDs, so id = 8*slot + xcd.  All N tiles of
    // one M tile run back-to-back on ONE XCD: the second N tile finds the source rows in that XCD's L2 and the two
    // half-row output writes meet there before they go to HBM.
    const int n_ntiles = (p.NOUT + BN - 1) / BN;
    const unsigned slot = bid >> 3;
    const long long mtile = (long long)(slot / n_ntiles) * 8 + (bid & 7);
    const long long m0 = mtile * BM;
    const int n0 = (int)(slot % n_ntiles) * BN;
    if (m0 >= p.M) return;     // grid is padded to 8 M tiles per round

    const i32x4 rsrc_a = dma_rsrc(p.src, src_bytes);
    const i32x4 rsrc_b = dma_rsrc(p.wgt, wgt_bytes);
    const unsigned smem_base = lds_addr(smem_raw);

    const int lr = tid >> 3;                       // tile row of DMA instruction 0 (+32 per instruction)
    const int lc = (tid & 7) ^ ((lr >> 1) & 7);    // LOGICAL chunk column this lane fetches into physical slot tid&7
    int rbase[AI];                                 // byte offset of the row's tap-(0,0) pixel (MODE 0/1) / pixel index (MODE 2)
    unsigned long long vmask[AI];
    int ty[AI], tx[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const long long m = m0 + lr + 32 * i;
        rbase[i] = 0; vmask[i] = 0; ty[i] = 0; tx[i] = 0;
        if (m < p.M) {
            if constexpr (MODE == 0) {
                rbase[i] = (int)(m * p.SC * ES);
                vmask[i] = 1;
            } else {
                const long long img = m / ((long long)p.DH * p.DW);
                const int rem = (int)(m - img * (long long)p.DH * p.DW);
                const int dy = rem / p.DW;
                const int dx = rem - dy * p.DW;
                const int y0 = dy * p.mul + p.off0;
                const int x0 = dx * p.mulx + p.off0x;
                if constexpr (MODE == 1) {
                    rbase[i] = (int)(((img * p.SH + y0) * (long long)p.SW + x0) * p.SC * ES);
                    unsigned long long mk = 0;
                    for (int a = 0; a < p.R; ++a) {
                        const int y = y0 + p.sgn * a;
                        if (y < 0 || y >= p.SH) continue;
                        for (int b = 0; b < p.S; ++b) {
                            const int x = x0 + p.sgn * b;
                            if (x >= 0 && x < p.SW) mk |= 1ull << (a * p.S + b);
                        }
                    }
                    vmask[i] = mk;
                } else {
                    rbase[i] = (int)(img * (long long)p.SH * p.SW);
                    vmask[i] = 1;
                    ty[i] = y0; tx[i] = x0;
                }
            }
        }
    }
    const int slab0 = p.cslab ? (n0 / 64) * (4 * ES) : 0;      // 64 channels = 4*sizeof(T) chunks of 16 bytes
    int q = lc;
    int tap = q / p.SCc;
    int cc = q - tap * p.SCc;
    int kh = tap / p.S;
    int kw = tap - kh * p.S;
    const unsigned wrow_bytes = (unsigned)p.wrow_chunks * 16u;
    unsigned bbase[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int o = n0 + lr + 32 * j;
        bbase[j] = o < p.NOUT ? (unsigned)o * wrow_bytes : OOB;
    }

    auto issue = [&](int buf) {
        const unsigned As = smem_base + (unsigned)buf * (STAGE * 16);
        const unsigned Bs = As + BM * 128;
        const bool qok = q < p.Qtot;
        unsigned va[AI];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < AI; ++i) va[i] = (qok && vmask[i]) ? (unsigned)(rbase[i] + q * 16) : OOB;
        } else if constexpr (MODE == 1) {
            const int t = kh * p.S + kw;
            const int toff = ((p.sgn * kh) * p.SW + p.sgn * kw) * p.SC * ES + (cc + slab0) * 16;
#pragma unroll
            for (int i = 0; i < AI; ++i) va[i] = (qok && ((vmask[i] >> t) & 1ull)) ? (unsigned)(rbase[i] + toff) : OOB;
        } else {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                unsigned v = OOB;
                if (qok && vmask[i]) {
                    int y = ty[i] + p.sgn * kh;
                    int x = tx[i] + p.sgn * kw;
                    bool ok = (y >= 0) && (x >= 0) && (y % p.div == 0) && (x % p.div == 0);
                    y /= p.div; x /= p.div;
                    ok = ok && (y < p.SH) && (x < p.SW);
                    if (ok) v = (unsigned)(((rbase[i] + y * p.SW + x) * p.SC) * ES + cc * 16);
                }
                va[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < AI; ++i)
            dma16(rsrc_a, As + (8 * wave + 32 * i) * 128, va[i]);
        int wqv = q;
        if constexpr (MODE != 0) wqv = ((p.wk0y + p.wkstep * kh) * p.S_full + p.wk0x + p.wkstep * kw) * p.SCc + cc;
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const unsigned vb = (qok && bbase[j] != OOB) ? bbase[j] + (unsigned)wqv * 16u : OOB;
            dma16(rsrc_b, Bs + (8 * wave + 32 * j) * 128, vb);
        }
        // advance this lane's K position by one K-step (8 chunks)
        q += 8;
        if constexpr (MODE != 0) {
            cc += 8;
            while (cc >= p.SCc) {
                cc -= p.SCc;
                ++kw;
                if (kw == p.S) { kw = 0; ++kh; }
            }
        }
    };

    // ---- UNI: lane constants + scalar K walk -------------------------------------------------------------------------
    unsigned voffA[AI], voffB[BI], vm32[AI];
    int u_cc = 0, u_kh = 0, u_kw = 0, u_ks = 0;       // wave-uniform (scalar registers)
    if constexpr (UNI) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            vm32[i] = (unsigned)vmask[i];
            if constexpr (MODE == 0) voffA[i] = vmask[i] ? (unsigned)rbase[i] + (unsigned)lc * 16u : OOB;
            else voffA[i] = (unsigned)rbase[i] + (unsigned)(lc + slab0) * 16u;      // may be "negative": the tap offset is added per lane
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) voffB[j] = bbase[j] != OOB ? bbase[j] + (unsigned)lc * 16u : OOB;
    }
    auto issue_uni = [&](int buf) {
        const unsigned As = smem_base + (unsigned)buf * (STAGE * 16) + (unsigned)wave * 1024u;
        const unsigned Bs = As + BM * 128;
        const bool tail = (u_ks + 1) * 8 > p.Qtot;             // only MODE 0 with C % 64 != 0: some lanes' chunks do not exist
        const unsigned lane_dead = (tail && (u_ks * 8 + lc) >= p.Qtot) ? OOB : 0u;
        if constexpr (MODE == 0) {
            const unsigned soff = (unsigned)u_ks * 128u;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const unsigned va = voffA[i] | lane_dead;
                if (i == 0) dma16s<0>(rsrc_a, As, va, soff);
                else if (i == 1) dma16s<4096>(rsrc_a, As, va, soff);
                else if (i == 2) dma16s<8192>(rsrc_a, As, va, soff);
                else dma16s<12288>(rsrc_a, As, va, soff);
            }
#pragma unroll
            for (int j = 0; j < BI; ++j) {
                const unsigned vb = voffB[j] | lane_dead;
                if (j == 0) dma16s<0>(rsrc_b, Bs, vb, soff);
                else if (j == 1) dma16s<4096>(rsrc_b, Bs, vb, soff);
                else if (j == 2) dma16s<8192>(rsrc_b, Bs, vb, soff);
                else dma16s<12288>(rsrc_b, Bs, vb, soff);
            }
        } else {
            const unsigned toff = (unsigned)(((p.sgn * u_kh) * p.SW + p.sgn * u_kw) * p.SC * ES + u_cc * 16);
            const unsigned bit = 1u << (u_kh * p.S + u_kw);
            const unsigned soffB = (unsigned)(((p.wk0y + p.wkstep * u_kh) * p.S_full + p.wk0x + p.wkstep * u_kw) * p.SCc + u_cc) * 16u;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const unsigned va = (vm32[i] & bit) ? voffA[i] + toff : OOB;
                if (i == 0) dma16s<0>(rsrc_a, As, va, 0u);
                else if (i == 1) dma16s<4096>(rsrc_a, As, va, 0u);
                else if (i == 2) dma16s<8192>(rsrc_a, As, va, 0u);
                else dma16s<12288>(rsrc_a, As, va, 0u);
            }
#pragma unroll
            for (int j = 0; j < BI; ++j) {
                if (j == 0) dma16s<0>(rsrc_b, Bs, voffB[j], soffB);
                else if (j == 1) dma16s<4096>(rsrc_b, Bs, voffB[j], soffB);
                else if (j == 2) dma16s<8192>(rsrc_b, Bs, voffB[j], soffB);
                else dma16s<12288>(rsrc_b, Bs, voffB[j], soffB);
            }
            u_cc += 8;
            if (u_cc >= p.SCc) {
                u_cc = 0;
                if (++u_kw == p.S) { u_kw = 0; ++u_kh; }
            }
        }
        ++u_ks;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    EpiRegs<BM, BN> er;
    if constexpr (PF) epi_prefetch<T, BM, BN>(p, m0, n0, er);
    const int nk = (p.Qtot + 7) / 8;
    // both stages are requested up front (vmcnt retires in issue order: AI+BI outstanding = stage 0 has landed), then
    // stage ks+1 is re-requested into the buffer iteration ks-1 has finished reading
    auto issue_any = [&](int buf) {
        if constexpr (UNI) issue_uni(buf);
        else issue(buf);
    };
    issue_any(0);
    if (nk > 1) issue_any(1);
    for (int ks = 0; ks < nk; ++ks) {
        if (ks == 0 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile ks have landed
        dma_barrier();                                      // ... and everybody else's; also frees the other buffer
        if (ks >= 1 && ks + 1 < nk) issue_any((ks + 1) & 1);
        const uint4* As = smem + (ks & 1) * STAGE;
        const uint4* Bs = As + BM * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = 2 * kk + hh;
            uint4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[swz(wm * (BM / 2) + i * 32 + l31, c)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[swz(wn * (BN / 2) + j * 32 + l31, c)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // a launch with no K-step still has its (all-zero) first stage in flight
    __syncthreads();
    igemm_epilogue<T, BM, BN, PF>(p, acc, smem_raw, m0, n0, slab_row0 + mtile, &er);
}


// ---------------------------------------------------------------------------------------------
// Streaming variant for the HBM-bound pure-GEMM convolutions with a short contraction (1x1, C_in <= 128):
// a PERSISTENT workgroup owns one destination-channel tile, keeps its weights resident in LDS, and walks
// pixel tiles with the NEXT tile's LDS-DMA always in flight while the current tile is multiplied and
// its epilogue (residual/mask loads, stores) drains -- the one-shot kernel had a single short burst
// of loads per workgroup and then nothing outstanding (2.6 TB/s algorithmic); this keeps the memory
// system busy.  LDS: [B: NK stages][A: 2 x NK stages][epilogue staging (BM/2 x BN fp32)].
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int NK>
__global__ __launch_bounds__(256) void igemm_stream_kernel(IgemmParams p, unsigned src_bytes, unsigned wgt_bytes, int n_mtiles) {
    constexpr int ES = (int)sizeof(T);
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int B_BYTES = NK * BN * 128;
    constexpr int A_BYTES = NK * BM * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* Bs_base = smem_raw;
    unsigned char* As_base = smem_raw + B_BYTES;
    unsigned char* epi = As_base + 2 * A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, wgt_bytes, 0x00020000);
    const int lr = tid >> 3;
    const int lc = (tid & 7) ^ ((lr >> 1) & 7);
    const unsigned wrow_bytes = (unsigned)p.wrow_chunks * 16u;
    const unsigned row_bytes = (unsigned)p.SC * ES;

    // weights: once per workgroup
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int q = ks * 8 + lc;
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int o = n0 + lr + 32 * j;
            const unsigned vb = (o < p.NOUT && q < p.Qtot) ? (unsigned)o * wrow_bytes + (unsigned)q * 16u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(Bs_base + ks * BN * 128 + (8 * wave + 32 * j) * 128),
                                                     16, (int)vb, 0, 0, 0);
        }
    }
    auto issue_a = [&](int tile, int buf) {
        const long long m0 = (long long)tile * BM;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int q = ks * 8 + lc;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const long long m = m0 + lr + 32 * i;
                const unsigned va = (m < p.M && q < p.Qtot) ? (unsigned)(m * row_bytes) + (unsigned)q * 16u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsrc_a, (__attribute__((address_space(3))) void*)(As_base + buf * A_BYTES + ks * BM * 128 + (8 * wave + 32 * i) * 128), 16,
                    (int)va, 0, 0, 0);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < n_mtiles) issue_a(tile, 0);
    for (int it = 0; tile < n_mtiles; tile += gridDim.x, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nxt = tile + gridDim.x;
        if (nxt < n_mtiles) issue_a(nxt, (it + 1) & 1);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const uint4* As = reinterpret_cast<const uint4*>(As_base + (it & 1) * A_BYTES + ks * BM * 128);
            const uint4* Bs = reinterpret_cast<const uint4*>(Bs_base + ks * BN * 128);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int c = 2 * kk + hh;
                uint4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[swz(wm * (BM / 2) + i * 32 + l31, c)];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[swz(wn * (BN / 2) + j * 32 + l31, c)];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
            }
        }
        igemm_epilogue<T, BM, BN>(p, acc, epi, (long long)tile * BM, n0, tile);
        // the epilogue's last LDS reads (statistics fold) must finish before the next tile's first staging write
        __syncthreads();
    }
}

// Fold the per-workgroup partial rows: out[c] += sum_r slab[r][c] for c < ncols (row stride = stride).
// Workgroup = 64 columns x 4 row lanes, gridDim.y row chunks.  Every result repeats bit for bit (round 5): `stats` is an exact,
// order-independent accumulator (ex_add); a float `colsum` target is added to by ONE workgroup per column when the grid has one row
// chunk, and with several chunks each writes its sum into the dead second half of partial row blockIdx.y (`chunk_dst`: the sum-of-
// squares columns, which a column-sum fold never reads) for colsum_chunk_fold_kernel to add in chunk order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int rows, int stride, int ncols,
                                                          int nout, float* colsum, double* stats, float* chunk_dst) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per;
    int r1 = r0 + per;
    if (r1 > rows) r1 = rows;
    double acc = 0.0;
    if (c < ncols) {
        // four rows in flight per thread (a single dependent load per step: up to 22 memory latencies per launch)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            const float v0 = slab[(long long)r * stride + c], v1 = slab[(long long)(r + 4) * stride + c];
            const float v2 = slab[(long long)(r + 8) * stride + c], v3 = slab[(long long)(r + 12) * stride + c];
            a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
        }
        for (; r < r1; r += 4) a0 += (double)slab[(long long)r * stride + c];
        acc = (a0 + a1) + (a2 + a3);
    }
    __shared__ double red[4][64];
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (chunk_dst && c < nout) chunk_dst[(long long)blockIdx.y * stride + nout + c] = (float)t;
        else if (colsum && c < nout) colsum[c] += (float)t;            // gridDim.y == 1: the only writer of this column
        if (stats) ex_add(stats, nout, c >= nout ? 1 : 0, c >= nout ? c - nout : c, t);
    }
}

__global__ __launch_bounds__(256) void colsum_chunk_fold_kernel(const float* __restrict__ slab, int chunks, int stride, int nout,
                                                                float* __restrict__ colsum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nout) return;
    double t = 0.0;
    for (int y = 0; y < chunks; ++y) t += (double)slab[(long long)y * stride + nout + c];
    colsum[c] += (float)t;
}

// out[c] += sum over the partial rows, deterministic (see slab_reduce_kernel); `colsum` and `stats` are alternatives.
int launch_slab_reduce(float* slab, int rows, int nout, float* colsum, double* stats, hipStream_t st) {
    const int ncols = stats ? 2 * nout : nout;
    int chunks = rows / 64;
    if (chunks < 1) chunks = 1;
    if (chunks > 32) chunks = 32;
    const bool two_phase = colsum && !stats && chunks > 1;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((ncols + 63) / 64, chunks), dim3(256), 0, st, slab, rows, 2 * nout, ncols, nout, colsum, stats,
                       two_phase ? slab : (float*)nullptr);
    if (two_phase)
        hipLaunchKernelGGL(colsum_chunk_fold_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, slab, chunks, 2 * nout, nout, colsum);
    return 0;
}

// name of the kernel instantiation this thread launched last (cs_last_conv_variant): bench.py / tools/check_bench_vs_profile.py
// key the roofline on it instead of re-deriving the dispatcher's choice
template <typename T> const char* tname() { return sizeof(T) == 2 ? "bf16" : "f32"; }
void note_variant(const char* fmt, ...) {
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    cs_set_variant_(buf);
}

int g_stream_enabled = 0;   // persistent streaming kernel for short-K pure-GEMM convs: opt-in (cs_set_igemm_path(3)); measured
                            // 5-25 % SLOWER than the one-shot kernel on MI355X (its vmcnt(0) also drains the previous tile's stores)
const bool g_merge_classes = !cs_env_flag_("CELLSEG_NO_MERGE");   // A/B experiments only
const bool g_uni_walk = !cs_env_flag_("CELLSEG_NO_UNI");   // A/B experiments only
const bool g_epi_prefetch = !cs_env_flag_("CELLSEG_NO_EPI_PREFETCH");   // A/B experiments only
int g_igemm_path = 0;   // 0 = LDS-DMA when operands < 2 GiB, 1 = always register-staged (A/B testing)

int igemm_mode(const IgemmParams& p) {
    if (p.div > 1) return 2;
    if (p.R == 1 && p.S == 1 && p.mul == 1 && p.mulx == 1 && p.off0 == 0 && p.off0x == 0 && p.wkstep == 1 && p.dst_step == 1 && !p.cslab) return 0;
    return p.R * p.S <= 64 ? 1 : 2;
}

template <typename T, int BM, int BN>
int launch_igemm(const IgemmParams& p, hipStream_t st) {
    constexpr size_t one_stage = (size_t)(BM + BN) * 8 * 16;
    constexpr size_t epi_bytes = (size_t)BM * BN * 2;          // (BM/2) x BN fp32, two passes
    constexpr size_t red_bytes = 256 * 16 * 4;                 // statistics fold
    const int nk_host = (p.Qtot + 7) / 8;
    size_t lds = (nk_host <= 1 ? 1 : 2) * one_stage;           // a single K-step needs a single stage
    if (lds < epi_bytes) lds = epi_bytes;
    if (lds < red_bytes) lds = red_bytes;
    const unsigned n_mt = (unsigned)((p.M + BM - 1) / BM), n_nt = (unsigned)((p.NOUT + BN - 1) / BN);
    dim3 grid(((n_mt + 7) / 8) * 8 * n_nt, 1, 1);             // see the tile-order note in the kernels
    if (p.ncls) {
        // all parity classes of a strided data gradient in one launch (caller checked: DMA-able, tap-walking mode for every class)
        IgemmParams q = p;
        unsigned b0 = 0;
        int r0 = 0, max_nk = 0;
        for (int i = 0; i < q.ncls; ++i) {
            const unsigned mt = (unsigned)((q.cls[i].M + BM - 1) / BM);
            q.cls[i].block0 = (int)b0;
            q.cls[i].row0 = r0;
            b0 += ((mt + 7) / 8) * 8 * n_nt;
            r0 += (int)mt;
            const int nk_c = (q.cls[i].Qtot + 7) / 8;
            if (nk_c > max_nk) max_nk = nk_c;
        }
        size_t clds = (max_nk <= 1 ? 1 : 2) * one_stage;
        if (clds < epi_bytes) clds = epi_bytes;
        if (clds < red_bytes) clds = red_bytes;
        const bool uni = g_uni_walk && q.SCc % 8 == 0;       // every class has R*S <= 4 taps here
        const unsigned long long sb = (unsigned long long)q.src_pixels * q.SC * sizeof(T);
        const unsigned long long wb = (unsigned long long)q.NOUT * q.wrow_chunks * 16ull;
        if (uni) { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,true>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 1, false, true>), dim3(b0), dim3(256), clds, st, q, (unsigned)sb, (unsigned)wb); }
        else { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,false>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 1, false, false>), dim3(b0), dim3(256), clds, st, q, (unsigned)sb, (unsigned)wb); }
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    const unsigned long long src_bytes = (unsigned long long)p.src_pixels * p.SC * sizeof(T);
    const unsigned long long wgt_bytes = (unsigned long long)p.NOUT * p.wrow_chunks * 16ull;
    const bool dma = g_igemm_path == 0 && src_bytes < 0x80000000ull && wgt_bytes < 0x80000000ull;
    const int mode = igemm_mode(p);
    if (dma && mode == 0 && nk_host <= 2 && g_stream_enabled) {
        // persistent streaming kernel: ~2 workgroups per CU in total, each pinned to one N tile
        const int n_mtiles = (int)((p.M + BM - 1) / BM);
        const int n_ntiles = (p.NOUT + BN - 1) / BN;
        int gx = (512 + n_ntiles - 1) / n_ntiles;
        if (gx > n_mtiles) gx = n_mtiles;
        if (gx < 1) gx = 1;
        dim3 sgrid(gx, n_ntiles, 1);
        const size_t slds = (size_t)nk_host * (BN + 2 * BM) * 128 + ((size_t)BM * BN * 2 > 16384 ? (size_t)BM * BN * 2 : 16384);
        auto raise = [&](const void* fn) -> int {
            if (slds <= 65536) return CS_OK;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess) {
                cs_set_error_("igemm_stream: cannot raise the dynamic LDS limit");
                return CS_ERR_LAUNCH;
            }
            return CS_OK;
        };
        if (nk_host <= 1) {
            if (raise(reinterpret_cast<const void*>(&igemm_stream_kernel<T, BM, BN, 1>)) != CS_OK) return CS_ERR_LAUNCH;
            { note_variant("igemm_stream_kernel<%s,%d,%d,1>", tname<T>(), BM, BN); hipLaunchKernelGGL((igemm_stream_kernel<T, BM, BN, 1>), sgrid, dim3(256), slds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes, n_mtiles); }
        } else {
            if (raise(reinterpret_cast<const void*>(&igemm_stream_kernel<T, BM, BN, 2>)) != CS_OK) return CS_ERR_LAUNCH;
            { note_variant("igemm_stream_kernel<%s,%d,%d,2>", tname<T>(), BM, BN); hipLaunchKernelGGL((igemm_stream_kernel<T, BM, BN, 2>), sgrid, dim3(256), slds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes, n_mtiles); }
        }
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    // scalar K walk: every 1x1 launch; tap-walking launches whose tap is wave-uniform (whole 64-element K-steps per tap)
    const bool uni1 = g_uni_walk && mode == 1 && p.SCc % 8 == 0 && p.R * p.S <= 32;
    const bool uni0 = g_uni_walk && mode == 0;
    if constexpr (sizeof(T) == 2) {
        if (dma && mode != 2 && nk_host > 1 && (p.residual || p.mask || p.bits_in) && p.dst_step == 1 && g_epi_prefetch) {
            if (mode == 0 && uni0) { note_variant("igemm_dma_kernel<%s,%d,%d,%d,true,true>", tname<T>(), BM, BN, 0); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 0, true, true>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); }
            else if (mode == 0) { note_variant("igemm_dma_kernel<%s,%d,%d,%d,true,false>", tname<T>(), BM, BN, 0); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 0, true>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); }
            else if (uni1) { note_variant("igemm_dma_kernel<%s,%d,%d,%d,true,true>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 1, true, true>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); }
            else { note_variant("igemm_dma_kernel<%s,%d,%d,%d,true,false>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 1, true>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); }
            CS_LAUNCH_CHECK();
            return CS_OK;
        }
    }
    if (dma && ((mode == 0 && uni0) || uni1)) {
        if (mode == 0) { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,true>", tname<T>(), BM, BN, 0); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 0, false, true>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); }
        else { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,true>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 1, false, true>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); }
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    if (dma) {
        switch (mode) {
            case 0: { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,false>", tname<T>(), BM, BN, 0); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 0>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); } break;
            case 1: { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,false>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 1>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); } break;
            default: { note_variant("igemm_dma_kernel<%s,%d,%d,%d,false,false>", tname<T>(), BM, BN, 2); hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 2>), grid, dim3(256), lds, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes); } break;
        }
    } else {
        switch (mode) {
            case 0: { note_variant("igemm_kernel<%s,%d,%d,%d>", tname<T>(), BM, BN, 0); hipLaunchKernelGGL((igemm_kernel<T, BM, BN, 0>), grid, dim3(256), lds, st, p); } break;
            case 1: { note_variant("igemm_kernel<%s,%d,%d,%d>", tname<T>(), BM, BN, 1); hipLaunchKernelGGL((igemm_kernel<T, BM, BN, 1>), grid, dim3(256), lds, st, p); } break;
            default: { note_variant("igemm_kernel<%s,%d,%d,%d>", tname<T>(), BM, BN, 2); hipLaunchKernelGGL((igemm_kernel<T, BM, BN, 2>), grid, dim3(256), lds, st, p); } break;
        }
    }
    CS_LAUNCH_CHECK();
    return CS_OK;
}

// Tile choice: wide-N tiles when there are enough output channels; shrink BM when the grid
// would not fill the 256 CUs.  Returns BM*1000+BN.
int igemm_tile(long long M, int NOUT) {
    static const int forced = cs_env_int_("CELLSEG_TILE", 0);   // experiments only
    if (forced) return (NOUT <= 64 && forced % 1000 == 128) ? forced - 64 : forced;
    static const int thr = cs_env_int_("CELLSEG_TILE_THR", 1536);   // experiments only (A/B: 384..3072 within 1 %, 1536 best)
    const long long mt128 = (M + 127) / 128;
    // 64-wide tiles where they cut the padded width by a fifth or more (144 or 192 output channels: 192 instead of 256 columns): the
    // EfficientNet expansions at 150 x 150 / 75 x 75 ran their second 128-wide N tile 12 % / 50 % full -- and a workgroup's epilogue costs
    // the same however full its tile is (24 -> 144 @150x150: 244 us at 128 wide)
    static const int narrow = cs_env_int_("CELLSEG_TILE_NARROW", 1);      // A/B experiments only (2 = off)
    const int p128 = (NOUT + 127) / 128 * 128, p64 = (NOUT + 63) / 64 * 64;
    if (NOUT > 64 && !(narrow == 1 && p64 * 5 <= p128 * 4)) {
        const long long blocks = mt128 * ((NOUT + 127) / 128);
        return blocks >= thr ? 128128 : 64128;
    }
    return mt128 >= thr ? 128064 : 64064;
}

template <typename T>
int dispatch_igemm(const IgemmParams& p_in, float* colsum, double* stats, hipStream_t st) {
    int rc;
    IgemmParams p = p_in;
    int tile = igemm_tile(p.M, p.NOUT);
    {
        // train-mode BN statistics of a launch with few pixel tiles: fp64 atomics from the epilogue, no fold launch
        static const int atomic_rows = cs_env_int_("CELLSEG_STATS_ATOMIC_ROWS", 512);      // A/B experiments only (1 = never)
        const long long rows_ = (p.M + tile / 1000 - 1) / (tile / 1000);
        if (p.slab && stats && !colsum && !p.cslab && !p.ncls && rows_ <= atomic_rows) p.stat_atomic = stats;
    }
    if (p.cslab) tile = (p.M + 127) / 128 >= 384 ? 128064 : 64064;   // one 64-channel slab per N tile
    switch (tile) {
        case 128128: rc = launch_igemm<T, 128, 128>(p, st); break;
        case 64128: rc = launch_igemm<T, 64, 128>(p, st); break;
        case 128064: rc = launch_igemm<T, 128, 64>(p, st); break;
        default: rc = launch_igemm<T, 64, 64>(p, st); break;
    }
    if (rc != CS_OK || !p.slab || (!colsum && !stats) || p.stat_atomic) return rc;      // no colsum/stats target: the partial rows are the result
    const int bm = tile / 1000;
    int rows = (int)((p.M + bm - 1) / bm);
    if (p.ncls) {
        rows = 0;
        for (int i = 0; i < p.ncls; ++i) rows += (int)((p.cls[i].M + bm - 1) / bm);
    }
    CS_CHECK_ARG(!(colsum && stats), "igemm: column sums and statistics are alternatives");
    launch_slab_reduce(p.slab, rows, p.NOUT, colsum, stats, st);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

int check_geom(const CsConvGeom* g, int dtype) {
    CS_CHECK_ARG(g != nullptr, "conv: geometry is NULL");
    CS_CHECK_ARG(dtype == CS_F32 || dtype == CS_BF16, "conv: dtype must be CS_F32 or CS_BF16");
    const int ce = dtype == CS_F32 ? 4 : 8;
    CS_CHECK_ARG(g->N > 0 && g->H > 0 && g->W > 0 && g->C > 0 && g->K > 0, "conv: non-positive extent");
    CS_CHECK_ARG(g->R > 0 && g->S > 0 && g->stride > 0 && g->pad >= 0, "conv: bad kernel/stride/pad");
    CS_CHECK_ARG(g->C % ce == 0, "conv: stored input channels must be a multiple of the 16-byte chunk");
    CS_CHECK_ARG(g->K % 8 == 0, "conv: stored output channels must be a multiple of 8");
    CS_CHECK_ARG(g->P == (g->H + 2 * g->pad - g->R) / g->stride + 1, "conv: P does not match H/R/stride/pad");
    CS_CHECK_ARG(g->Q == (g->W + 2 * g->pad - g->S) / g->stride + 1, "conv: Q does not match W/S/stride/pad");
    return CS_OK;
}

}  // namespace

extern "C" int cs_igemm_tile(long long M, int n_out) { return igemm_tile(M, n_out); }
#ifdef CS_AB_SWITCHES
extern "C" int cs_set_igemm_path(int path) {
    // 0 = LDS-DMA (default), 1 = register-staged everywhere, 3 = LDS-DMA + persistent streaming kernel for short-K 1x1
    const int old = g_igemm_path == 1 ? 1 : (g_stream_enabled ? 3 : 0);
    g_igemm_path = path == 1 ? 1 : 0;
    g_stream_enabled = path == 3 ? 1 : 0;
    return old;
}
#endif

extern "C" size_t cs_conv2d_stats_workspace(long long M, int n_out) {
    // one partial row per M tile; sized for the smallest tile height (64) so every dispatch variant fits
    return (size_t)((M + 63) / 64 + 4) * 2 * (size_t)n_out * sizeof(float);
}

static int conv2d_fwd_impl(const CsConvGeom* g, int dtype, const void* x, const void* w_khwc, const float* scale,
                           const float* shift, const void* residual, int act, void* y, double* stats, void* workspace,
                           unsigned char* relu_bits, void* stream);

extern "C" int cs_conv2d_fwd(const CsConvGeom* g, int dtype, const void* x, const void* w_khwc, const float* scale,
                             const float* shift, const void* residual, int act, void* y, double* stats, void* workspace,
                             void* stream) {
    return conv2d_fwd_impl(g, dtype, x, w_khwc, scale, shift, residual, act, y, stats, workspace, nullptr, stream);
}

extern "C" int cs_conv2d_fwd_bits(const CsConvGeom* g, int dtype, const void* x, const void* w_khwc, const float* scale,
                                  const float* shift, const void* residual, int act, void* y, uint8_t* positive_bits, void* stream) {
    CS_CHECK_ARG(positive_bits != nullptr, "conv2d_fwd_bits: NULL bit tensor");
    return conv2d_fwd_impl(g, dtype, x, w_khwc, scale, shift, residual, act, y, nullptr, nullptr, positive_bits, stream);
}

static int conv2d_fwd_impl(const CsConvGeom* g, int dtype, const void* x, const void* w_khwc, const float* scale,
                           const float* shift, const void* residual, int act, void* y, double* stats, void* workspace,
                           unsigned char* relu_bits, void* stream) {
    const int slab = (g && g->groups > 1) ? 1 : 0;
    int rc = check_geom(g, dtype);
    if (rc != CS_OK) return rc;
    if (slab) CS_CHECK_ARG(g->C % 64 == 0 && g->K == g->C, "grouped conv: width must be a multiple of 64 and C == K");
    CS_CHECK_ARG(x && w_khwc && y, "conv2d_fwd: NULL tensor");
    CS_CHECK_ARG(!stats || workspace, "conv2d_fwd: stats need a workspace of cs_conv2d_stats_workspace() bytes");
    IgemmParams p{};
    const int ce = dtype == CS_F32 ? 4 : 8;
    p.src = x; p.wgt = w_khwc; p.dst = y;
    p.scale = scale; p.shift = shift; p.residual = residual; p.mask = nullptr;
    CS_CHECK_ARG(!relu_bits || g->K % 32 == 0, "conv2d_fwd_bits: stored output channels must be a multiple of 32");
    p.bits_out = relu_bits; p.bits_in = nullptr;
    p.bits_M = (long long)g->N * g->P * g->Q;
    p.slab = stats ? reinterpret_cast<float*>(workspace) : nullptr;
    p.SH = g->H; p.SW = g->W; p.SC = g->C;
    p.DH = g->P; p.DW = g->Q; p.NOUT = g->K;
    p.R = g->R; p.S = g->S;
    p.mul = g->stride; p.mulx = g->stride; p.div = 1; p.off0 = -g->pad; p.off0x = -g->pad; p.sgn = 1;
    p.wk0y = 0; p.wk0x = 0; p.wkstep = 1; p.S_full = g->S;
    p.dst_step = 1; p.dst_oy = 0; p.dst_ox = 0; p.DHF = g->P; p.DWF = g->Q;
    p.act = act;
    p.M = (long long)g->N * g->P * g->Q;
    p.src_pixels = (long long)g->N * g->H * g->W;
    p.SCc = g->C / ce;
    p.cslab = slab;
    if (slab) p.SCc = 64 / ce;
    p.Qtot = g->R * g->S * p.SCc;
    p.wrow_chunks = p.Qtot;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == CS_F32 ? dispatch_igemm<float>(p, nullptr, stats, st) : dispatch_igemm<bf16_t>(p, nullptr, stats, st);
}

// can the (up to) four parity classes of a stride-2 data gradient go out as ONE launch (conv2d_dgrad_impl)?  Then its partial column-sum
// rows are numbered across the classes (IgemmParams::ClassOv::row0) and may be left unfolded like a stride-1 launch's.
static bool strided_classes_merge(const CsConvGeom* g, int slab) {
    if (g->stride != 2 || g->R * g->S > 64 || slab || g_igemm_path != 0 || !g_merge_classes) return false;
    const unsigned long long sbytes = (unsigned long long)g->N * g->P * g->Q * g->K * 4ull;      // (fp32: the larger of the two dtypes)
    const unsigned long long wbytes = (unsigned long long)g->C * g->R * g->S * (unsigned long long)g->K * 4ull;
    return sbytes < 0x80000000ull && wbytes < 0x80000000ull;
}

extern "C" int cs_conv2d_dgrad_partial_rows(const CsConvGeom* g) {
    if (!g) return 0;
    if (g->stride == 1) {
        const long long M = (long long)g->N * g->H * g->W;
        const int bm = igemm_tile(M, g->C) / 1000;
        return (int)((M + bm - 1) / bm);
    }
    if (g->groups > 1 || !strided_classes_merge(g, 0)) return 0;          // not deferrable: the caller passes a `colsum` target
    long long max_m = 0, ms[4];
    int n = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int dh = (g->H - py + 1) / 2, dw = (g->W - px + 1) / 2;
            if (dh <= 0 || dw <= 0) continue;
            ms[n] = (long long)g->N * dh * dw;
            if (ms[n] > max_m) max_m = ms[n];
            ++n;
        }
    const int bm = igemm_tile(max_m, g->C) / 1000;
    long long rows = 0;
    for (int i = 0; i < n; ++i) rows += (ms[i] + bm - 1) / bm;
    return (int)rows;
}

extern "C" int cs_fold_partial_rows(const float* partial, int rows, int n_out, float* out, void* stream) {
    CS_CHECK_ARG(partial && out && rows > 0 && n_out > 0, "fold_partial_rows: bad arguments");
    // (the fold may use the dead sum-of-squares half of the first partial rows as scratch: `partial` is written)
    launch_slab_reduce(const_cast<float*>(partial), rows, n_out, out, nullptr, reinterpret_cast<hipStream_t>(stream));
    CS_LAUNCH_CHECK();
    return CS_OK;
}


static int conv2d_dgrad_impl(const CsConvGeom* g, int dtype, const void* dy, const void* w_chwk, const void* add,
                             const void* mask, const unsigned char* mask_bits, void* dx, float* colsum, void* workspace, void* stream);

extern "C" int cs_conv2d_dgrad(const CsConvGeom* g, int dtype, const void* dy, const void* w_chwk, const void* add,
                               const void* mask, void* dx, float* colsum, void* workspace, void* stream) {
    return conv2d_dgrad_impl(g, dtype, dy, w_chwk, add, mask, nullptr, dx, colsum, workspace, stream);
}

extern "C" int cs_conv2d_dgrad_bits(const CsConvGeom* g, int dtype, const void* dy, const void* w_chwk, const void* add,
                                    const uint8_t* mask_bits, void* dx, float* colsum, void* workspace, void* stream) {
    CS_CHECK_ARG(mask_bits != nullptr, "conv2d_dgrad_bits: NULL bit tensor");
    return conv2d_dgrad_impl(g, dtype, dy, w_chwk, add, nullptr, mask_bits, dx, colsum, workspace, stream);
}

static int conv2d_dgrad_impl(const CsConvGeom* g, int dtype, const void* dy, const void* w_chwk, const void* add,
                             const void* mask, const unsigned char* mask_bits, void* dx, float* colsum, void* workspace, void* stream) {
    const int slab = (g && g->groups > 1) ? 1 : 0;
    int rc = check_geom(g, dtype);
    if (rc != CS_OK) return rc;
    if (slab) CS_CHECK_ARG(g->C % 64 == 0 && g->K == g->C, "grouped conv: width must be a multiple of 64 and C == K");
    CS_CHECK_ARG(dy && w_chwk && dx, "conv2d_dgrad: NULL tensor");
    CS_CHECK_ARG(!colsum || workspace, "conv2d_dgrad: colsum needs a workspace of cs_conv2d_stats_workspace() bytes");
    const int ce = dtype == CS_F32 ? 4 : 8;
    CS_CHECK_ARG(g->K % ce == 0, "conv2d_dgrad: stored K must be a chunk multiple");
    CS_CHECK_ARG(g->C % 8 == 0, "conv2d_dgrad: stored C must be a multiple of 8");
    IgemmParams p{};
    p.src = dy; p.wgt = w_chwk; p.dst = dx;
    p.scale = nullptr; p.shift = nullptr; p.residual = add; p.mask = mask;
    p.bits_out = nullptr; p.bits_in = mask_bits;
    p.bits_M = (long long)g->N * g->H * g->W;
    CS_CHECK_ARG(!mask_bits || g->C % 32 == 0, "conv2d_dgrad_bits: stored input channels must be a multiple of 32");
    const bool deferred = !colsum && workspace;
    CS_CHECK_ARG(!deferred || (!slab && (g->stride == 1 || strided_classes_merge(g, slab))),
                 "conv2d_dgrad: deferred column sums need an ungrouped launch of stride 1, or of stride 2 whose parity classes merge "
                 "(cs_conv2d_dgrad_partial_rows > 0)");
    p.slab = (colsum || deferred) ? reinterpret_cast<float*>(workspace) : nullptr;
    p.SH = g->P; p.SW = g->Q; p.SC = g->K;
    p.DH = g->H; p.DW = g->W; p.NOUT = g->C;
    p.R = g->R; p.S = g->S;
    p.mul = 1; p.mulx = 1; p.div = g->stride; p.off0 = g->pad; p.off0x = g->pad; p.sgn = -1;
    p.wk0y = 0; p.wk0x = 0; p.wkstep = 1; p.S_full = g->S;
    p.dst_step = 1; p.dst_oy = 0; p.dst_ox = 0; p.DHF = g->H; p.DWF = g->W;
    p.act = CS_ACT_NONE;
    p.M = (long long)g->N * g->H * g->W;
    p.src_pixels = (long long)g->N * g->P * g->Q;
    p.SCc = g->K / ce;
    p.cslab = slab;
    if (slab) p.SCc = 64 / ce;
    p.Qtot = g->R * g->S * p.SCc;
    p.wrow_chunks = p.Qtot;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (g->stride == 1 || g->R * g->S > 64)
        return dtype == CS_F32 ? dispatch_igemm<float>(p, colsum, nullptr, st) : dispatch_igemm<bf16_t>(p, colsum, nullptr, st);
    // Strided data-gradient = stride^2 independent stride-1 problems, one per residue class of the
    // destination coordinate: class (py,px) owns pixels (a*s+py, b*s+px) and only the filter taps
    // kh == (py+pad) mod s (same for kw), for which the source row is a + (py+pad-kh)/s.  Classes without
    // taps still run (zero K-steps) so that `add`/`mask` are applied and dx is fully written.
    const int sd = g->stride;
    {
        // stride 2: the (up to) four classes go out as ONE launch when the LDS-DMA kernel can take them -- four quarter-size
        // launches (one of them with 4 taps, one with 1) never filled the chip and each paid its own ramp and column-sum fold
        const unsigned long long esz = dtype == CS_F32 ? 4 : 2;
        const unsigned long long sbytes = (unsigned long long)p.src_pixels * p.SC * esz, wbytes = (unsigned long long)p.NOUT * p.wrow_chunks * 16ull;
        IgemmParams q = p;
        int n = 0;
        bool ok = sd == 2 && g_igemm_path == 0 && sbytes < 0x80000000ull && wbytes < 0x80000000ull && !slab && g_merge_classes;
        long long max_m = 0;
        for (int py = 0; ok && py < sd; ++py)
            for (int px = 0; px < sd; ++px) {
                const int kh0 = (py + g->pad) % sd, kw0 = (px + g->pad) % sd;
                const int nj = kh0 < g->R ? (g->R - kh0 + sd - 1) / sd : 0;
                const int ni = kw0 < g->S ? (g->S - kw0 + sd - 1) / sd : 0;
                const int dh = (g->H - py + sd - 1) / sd, dw = (g->W - px + sd - 1) / sd;
                if (dh <= 0 || dw <= 0) continue;
                IgemmParams::ClassOv& c = q.cls[n++];
                c.DH = dh; c.DW = dw;
                c.off0 = (py + g->pad - kh0) / sd; c.off0x = (px + g->pad - kw0) / sd;
                c.R = nj; c.S = ni;
                if (nj == 0 || ni == 0) { c.R = 0; c.S = 1; }
                c.wk0y = kh0; c.wk0x = kw0;
                c.Qtot = c.R * c.S * p.SCc;
                c.dst_oy = py; c.dst_ox = px;
                c.M = (long long)g->N * dh * dw;
                c.block0 = 0; c.row0 = 0;
                if (c.M > max_m) max_m = c.M;
            }
        if (ok && n > 0) {
            q.ncls = n;
            q.div = 1; q.mul = 1; q.mulx = 1; q.sgn = -1; q.wkstep = sd; q.dst_step = sd;
            q.R = 1; q.S = 2;                       // any tap-walking shape: the per-class values replace them in the kernel
            q.Qtot = q.cls[0].Qtot;
            q.M = max_m;                            // tile choice
            return dtype == CS_F32 ? dispatch_igemm<float>(q, colsum, nullptr, st) : dispatch_igemm<bf16_t>(q, colsum, nullptr, st);
        }
    }
    for (int py = 0; py < sd; ++py) {
        for (int px = 0; px < sd; ++px) {
            IgemmParams c = p;
            const int kh0 = (py + g->pad) % sd, kw0 = (px + g->pad) % sd;
            const int nj = kh0 < g->R ? (g->R - kh0 + sd - 1) / sd : 0;
            const int ni = kw0 < g->S ? (g->S - kw0 + sd - 1) / sd : 0;
            c.DH = (g->H - py + sd - 1) / sd;
            c.DW = (g->W - px + sd - 1) / sd;
            if (c.DH <= 0 || c.DW <= 0) continue;
            c.div = 1; c.mul = 1; c.mulx = 1; c.sgn = -1;
            c.off0 = (py + g->pad - kh0) / sd;
            c.off0x = (px + g->pad - kw0) / sd;
            c.R = nj; c.S = ni;
            if (nj == 0 || ni == 0) { c.R = 0; c.S = 1; }
            c.wk0y = kh0; c.wk0x = kw0; c.wkstep = sd;
            c.Qtot = c.R * c.S * c.SCc;
            c.dst_step = sd; c.dst_oy = py; c.dst_ox = px;
            c.M = (long long)g->N * c.DH * c.DW;
            const int rc2 = dtype == CS_F32 ? dispatch_igemm<float>(c, colsum, nullptr, st) : dispatch_igemm<bf16_t>(c, colsum, nullptr, st);
            if (rc2 != CS_OK) return rc2;
        }
    }
    return CS_OK;
}

// =============================================================================================
// Weight gradient:  dW[k][q] += sum_m dy[m][k] * xcol[m][q]      (GEMM M=K_out, N=taps*C, K=pixels)
// Both operands are pixel-major in memory, i.e. "transposed" for the MFMA (a lane needs
// consecutive pixels of ONE channel).  LDS keeps the natural [pixel][channel] image; operands are
// fetched either element-wise (f32: one ds_read_b32 per MFMA operand; bf16 safe path: ds_read_u16)
// or, for bf16, with the gfx950 transposing read ds_read_b64_tr_b16.
// Split-K over pixel ranges (gridDim.z); partial tiles are combined with fp32 atomics.
// =============================================================================================
namespace {

struct WgradParams {
    const void* x;    // source activations NHWC [N][H][W][C]
    const void* dy;   // [M][KO]
    float* dw;        // [KO][QE] fp32, QE = R*S*C
    int H, W, C;
    int P, Q;
    int KO;
    int R, S, stride, pad;
    int stride_x, pad_x;   // == stride, pad except for the pixel-paired stem
    long long M;
    int QE;           // R*S*Cq (elements of one dW row)
    int Cq;           // channels per tap in dW's K space: C, or 64 for slab-dense grouped convolution
    int slab;         // grouped: output-channel tile k0 (BM == 64) reads source channels [k0, k0+64)
    int SCc;          // C / chunk
    long long m_per_split;   // multiple of the K-step
    long long slab_stride;   // elements between split-K slabs (KO*QE)
    // batched launch over n <= 8 identical-geometry layers: blockIdx.z = item * nsplit + slice; the pointer tables travel BY VALUE
    // in the kernel arguments (a device-memory table would need a blocking host-to-device copy every step)
    const void* x_tab[8];
    const void* dy_tab[8];
    float* dw_tab[8];
    int n_items;
    int nsplit;
    int nkt, tiles, total_z, per_xcd;   // output-channel tiles, output tiles per slice, slices x layers, work items per XCD
};

template <typename T, int BM, int BN, bool TR>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    constexpr int CE = Elem<T>::kChunk;
    constexpr int BKP = 32;                         // pixels per K-step
    // LDS row strides (elements).  bf16: +64 B per row so the 4 pixel rows of one transposing read fall on
    // 4 disjoint 16-bank groups (256-B rows put them on the SAME banks: 60 % of LDS cycles were conflicts).
    constexpr int LDA = sizeof(T) == 2 ? BM + 32 : BM;
    constexpr int LDB = sizeof(T) == 2 ? BN + 32 : BN;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int ACPR = BM / CE;            // A chunks per pixel row
    constexpr int BCPR = BN / CE;
    constexpr int AI = BKP * ACPR / 256;     // chunks per thread
    constexpr int BI = BKP * BCPR / 256;
    constexpr int ASTAGE = BKP * LDA;        // elements
    constexpr int BSTAGE = BKP * LDB;
    static_assert(AI >= 1 && BI >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    // XCD-aware order (1-D grid, workgroup id = 8*slot + xcd): all output tiles of one (layer, pixel slice) run back-to-back on
    // ONE XCD, so the slice's dy / x rows -- which every one of those tiles reads -- come out of that XCD's L2 instead of being
    // fetched once per XCD (measured 2.2-2.8x the algorithmic HBM bytes with the z-slowest 3-D grid).
    // The (slice-major) list of slice x tile work items is cut into 8 contiguous runs, one per XCD.
    const unsigned work = (blockIdx.x & 7) * (unsigned)p.per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= (unsigned)p.per_xcd || work >= (unsigned)p.total_z * (unsigned)p.tiles) return;
    const int z = (int)(work / (unsigned)p.tiles);
    const int tile = (int)(work % (unsigned)p.tiles);
    const int k0 = (tile % p.nkt) * BM;      // output-channel tile origin
    const int q0 = (tile / p.nkt) * BN;      // K-space (tap,c) element origin
    const int item = p.n_items ? z / p.nsplit : 0;
    const int slice = p.n_items ? z % p.nsplit : z;
    const long long mbeg = (long long)slice * p.m_per_split;
    long long mend = mbeg + p.m_per_split;
    if (mend > p.M) mend = p.M;
    // (slices are non-empty by construction: nsplit = ceil(M / m_per_split); an empty one would still write zeros)

    // static indices only: a runtime index into a by-value argument array would push the whole struct to scratch
    const void* xsel = p.x;
    const void* gsel = p.dy;
    float* dsel = p.dw;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (p.n_items && item == i) { xsel = p.x_tab[i]; gsel = p.dy_tab[i]; dsel = p.dw_tab[i]; }
    const T* __restrict__ xs = reinterpret_cast<const T*>(xsel);
    const T* __restrict__ gs = reinterpret_cast<const T*>(gsel);
    float* __restrict__ dw_base = dsel;

    // A operand (dy): chunk column fixed per thread
    const int ac = tid % ACPR;
    const int ar = tid / ACPR;               // + (256/ACPR)*i
    const int a_ch = k0 + ac * CE;
    const bool a_ok = a_ch < p.KO;
    // B operand (im2col of x): chunk column fixed per thread -> fixed tap and channel chunk
    const int bc = tid % BCPR;
    const int br = tid / BCPR;
    const int bq = q0 + bc * CE;             // element index in K space
    const bool b_ok = bq < p.QE;
    int b_kh = 0, b_kw = 0, b_c = 0;
    if (b_ok) {
        const int tap = bq / p.Cq;
        b_c = bq - tap * p.Cq + (p.slab ? k0 : 0);
        b_kh = tap / p.S;
        b_kw = tap - b_kh * p.S;
    }
    // walking (n, oy, ox) for each of this thread's B rows
    long long bn_[BI];
    int boy[BI], box[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const long long m = mbeg + br + (256 / BCPR) * i;
        const long long img = m / ((long long)p.P * p.Q);
        const int rem = (int)(m - img * (long long)p.P * p.Q);
        bn_[i] = img;
        boy[i] = rem / p.Q;
        box[i] = rem - boy[i] * p.Q;
    }

    uint4 ra[AI], rb[BI];
    long long mstep = mbeg;
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const long long m = mstep + ar + (256 / ACPR) * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_ok && m < mend) v = *reinterpret_cast<const uint4*>(gs + m * p.KO + a_ch);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const long long m = mstep + br + (256 / BCPR) * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b_ok && m < mend) {
                const int iy = boy[i] * p.stride - p.pad + b_kh;
                const int ix = box[i] * p.stride_x - p.pad_x + b_kw;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const long long pix = (bn_[i] * p.H + iy) * (long long)p.W + ix;
                    v = *reinterpret_cast<const uint4*>(xs + pix * p.C + b_c);
                }
            }
            rb[i] = v;
        }
    };
    auto advance = [&]() {
        mstep += BKP;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            box[i] += BKP;
            while (box[i] >= p.Q) {
                box[i] -= p.Q;
                if (++boy[i] == p.P) { boy[i] = 0; ++bn_[i]; }
            }
        }
    };
    auto lstore = [&](int buf) {
        T* As = smem + buf * (ASTAGE + BSTAGE);
        T* Bs = As + ASTAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<uint4*>(As + (ar + (256 / ACPR) * i) * LDA + ac * CE) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<uint4*>(Bs + (br + (256 / BCPR) * i) * LDB + bc * CE) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (int)((mend - mbeg + BKP - 1) / BKP);
    gload();
    advance();
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = (ks + 1) < nk;
        if (more) { gload(); advance(); }
        const T* As = smem + (ks & 1) * (ASTAGE + BSTAGE);
        const T* Bs = As + ASTAGE;
        if constexpr (sizeof(T) == 4) {
            // f32: v_mfma_f32_32x32x2_f32 wants A[i][k=hh], B[k=hh][j]: one dword each
#pragma unroll
            for (int s = 0; s < BKP / 2; ++s) {
                const int kr = 2 * s + hh;
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = reinterpret_cast<const float*>(As)[kr * LDA + wm * (BM / 2) + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = reinterpret_cast<const float*>(Bs)[kr * LDB + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < BKP / 16; ++s) {
                union Frag { bf16x8 v; unsigned short h[8]; s16x4 q[2]; };
                Frag a[TM], b[TN];
                if constexpr (TR) {
                    // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(channel) block, delivered
                    // column-major: lane i16 gets the 4 k-values of channel i16.
                    const int g16 = lane >> 4;          // 0..3 ; (g16 & 1) selects channels 0-15 / 16-31
                    const int i16 = lane & 15;
                    const int qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kr = 16 * s + 8 * hh + 4 * u + qq;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const unsigned short* ap = reinterpret_cast<const unsigned short*>(As) + kr * LDA +
                                                       wm * (BM / 2) + i * 32 + 16 * (g16 & 1) + 4 * pp;
                            a[i].q[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3)))*)(ap));
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const unsigned short* bp = reinterpret_cast<const unsigned short*>(Bs) + kr * LDB +
                                                       wn * (BN / 2) + j * 32 + 16 * (g16 & 1) + 4 * pp;
                            b[j].q[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3)))*)(bp));
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int kr = 16 * s + 8 * hh + e;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            a[i].h[e] = reinterpret_cast<const unsigned short*>(As)[kr * LDA + wm * (BM / 2) + i * 32 + l31];
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            b[j].h[e] = reinterpret_cast<const unsigned short*>(Bs)[kr * LDB + wn * (BN / 2) + j * 32 + l31];
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
            }
        }
        if (more) lstore((ks + 1) & 1);
        __syncthreads();
    }

    // ---- split-K partials: slice z writes its tile into slab z with PLAIN stores (32 consecutive q per half-wave =
    // 128-B segments).  fp32 atomics run at ~1.3 TB/s chip-wide and were the floor of this kernel (~50 us per launch);
    // plain stores are ~5x faster, need no zero-fill, and the fold over slabs (cs_wgrad_finalize) is deterministic.
    float* slab = dw_base + (long long)slice * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int qe = q0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = k0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (ko < p.KO && qe < p.QE) slab[(long long)ko * p.QE + qe] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA weight gradient (bf16, default): the [pixel][channel] tiles of dy and of the im2col view of x go global -> LDS with
// `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass: the register-staged kernel above spends ~200 LDS-store
// cycles per 256 MFMA cycles), through a ring of NST stages of 32 pixels with NST-1 stages in flight (counted vmcnt).
//  * one wave-instruction fills 1 KiB of LDS = 4 rows of 256 B (8 rows of 128 B for the 64-channel dy tile); rows cannot be
//    padded, so the conflict-free image for the transposing reads comes from an XOR on the 64-byte chunk group
//    (256-B rows: group ^= row & 3; 128-B rows: group ^= (row >> 1) & 1), applied to the per-lane SOURCE chunk;
//  * the lane's logical chunk -- hence its tap and channel -- is the same for every instruction and K-step (row & 3 and
//    (row >> 1) & 1 are lane constants), only the pixel walks;
//  * padding taps, pixel tails and channel / K-space tails are lanes with an out-of-range buffer offset (zeros).
// Needs both operands < 2 GiB; grouped (slab) and fp32 weight gradients stay on the register-staged kernel.
// ---------------------------------------------------------------------------------------------
template <int BM, int NST, bool PLAIN, int BKP>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(WgradParams p, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int BN = 128;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_ROW_B = BM * 2, B_ROW_B = BN * 2;          // bytes per pixel row
    constexpr int A_STAGE = BKP * A_ROW_B, B_STAGE = BKP * B_ROW_B;
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int A_RPI = 1024 / A_ROW_B, B_RPI = 1024 / B_ROW_B;     // rows per wave-instruction (4 or 8)
    constexpr int A_I = BKP / A_RPI / 4, B_I = BKP / B_RPI / 4;       // instructions per wave per stage
    static_assert(A_I >= 1 && B_I >= 1, "stage too small for 4 waves");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    const unsigned work = (blockIdx.x & 7) * (unsigned)p.per_xcd + (blockIdx.x >> 3);      // see wgrad_kernel
    if ((blockIdx.x >> 3) >= (unsigned)p.per_xcd || work >= (unsigned)p.total_z * (unsigned)p.tiles) return;
    const int z = (int)(work / (unsigned)p.tiles);
    const int tile = (int)(work % (unsigned)p.tiles);
    const int k0 = (tile % p.nkt) * BM;
    const int q0 = (tile / p.nkt) * BN;
    const int item = p.n_items ? z / p.nsplit : 0;
    const int slice = p.n_items ? z % p.nsplit : z;
    const long long mbeg = (long long)slice * p.m_per_split;
    long long mend = mbeg + p.m_per_split;
    if (mend > p.M) mend = p.M;

    const void* xsel = p.x;
    const void* gsel = p.dy;
    float* dsel = p.dw;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (p.n_items && item == i) { xsel = p.x_tab[i]; gsel = p.dy_tab[i]; dsel = p.dw_tab[i]; }
    const i32x4 rsrc_x = dma_rsrc(xsel, x_bytes);
    const i32x4 rsrc_g = dma_rsrc(gsel, dy_bytes);
    const unsigned smem_base = lds_addr(smem_raw);

    // ---- A (dy) lanes: row-in-instruction and logical channel chunk
    constexpr int A_CPR = A_ROW_B / 16;                        // chunks per row (16 or 8)
    const int a_rl = lane / A_CPR;                             // row inside one instruction
    const int a_slot = lane % A_CPR;
    const int a_swz = A_ROW_B == 256 ? ((a_rl & 3) << 2) : (((a_rl >> 1) & 1) << 2);
    const int a_ch = k0 + (a_slot ^ a_swz) * 8;
    const unsigned a_col = a_ch < p.KO ? (unsigned)a_ch * 2u : OOB;
    const unsigned g_row_b = (unsigned)p.KO * 2u;
    // ---- B (im2col of x) lanes
    const int b_rl = lane >> 4;
    const int b_slot = lane & 15;
    const int bq = q0 + (b_slot ^ ((b_rl & 3) << 2)) * 8;
    const bool b_ok = bq < p.QE;
    int b_kh = 0, b_kw = 0, b_c = 0;
    if (b_ok) {
        const int tap = bq / p.Cq;
        b_c = bq - tap * p.Cq;
        b_kh = tap / p.S;
        b_kw = tap - b_kh * p.S;
    }
    // Pixel walk.  The stage's first pixel (img0, oy0, ox0) is wave-uniform and advances in scalar registers; a lane's row is
    // that pixel + a lane constant < 32, folded back into (img, oy, ox) with two multiply-high divisions (exact: the
    // dividends stay below Q + 64 resp. P + 8) -- the per-row divergent carry loops of the register-staged kernel cost more
    // issue cycles per K-step than its 8 MFMAs.  PLAIN (1x1, stride 1, no padding): source pixel == destination pixel.
    const unsigned magicQ = 0xffffffffu / (unsigned)p.Q + 1u, magicP = 0xffffffffu / (unsigned)p.P + 1u;
    unsigned img0, oy0, ox0;
    {
        const long long img = mbeg / ((long long)p.P * p.Q);
        const int rem = (int)(mbeg - img * (long long)p.P * p.Q);
        img0 = (unsigned)img;
        oy0 = (unsigned)(rem / p.Q);
        ox0 = (unsigned)(rem - (int)oy0 * p.Q);
    }
    unsigned mstep = (unsigned)mbeg;                          // M * KO * 2 < 2 GiB: 32-bit pixel arithmetic throughout
    const unsigned mend32 = (unsigned)mend;
    const unsigned x_row_b = (unsigned)p.C * 2u;
    const unsigned b_col = (unsigned)b_c * 2u;
    const int b_dy = b_kh - p.pad, b_dx = b_kw - p.pad_x;

    auto issue = [&](int buf) {
        const unsigned As = smem_base + (unsigned)buf * STAGE;
        const unsigned Bs = As + A_STAGE;
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
            const int r = (wave * A_I + i) * A_RPI;            // first row of this instruction
            const unsigned m = mstep + (unsigned)(r + a_rl);
            const unsigned va = (m < mend32 && a_col != OOB) ? m * g_row_b + a_col : OOB;
            dma16(rsrc_g, As + r * A_ROW_B, va);
        }
#pragma unroll
        for (int i = 0; i < B_I; ++i) {
            const int r = (wave * B_I + i) * B_RPI;
            const unsigned m = mstep + (unsigned)(r + b_rl);
            unsigned vb = OOB;
            if constexpr (PLAIN) {
                if (b_ok && m < mend32) vb = m * x_row_b + b_col;
            } else {
                const unsigned t = ox0 + (unsigned)(r + b_rl);
                const unsigned w = __umulhi(t, magicQ);
                const unsigned ox = t - w * (unsigned)p.Q;
                const unsigned u = oy0 + w;
                const unsigned w2 = __umulhi(u, magicP);
                const unsigned oy = u - w2 * (unsigned)p.P;
                const int iy = (int)oy * p.stride + b_dy;
                const int ix = (int)ox * p.stride_x + b_dx;
                if (b_ok && m < mend32 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    vb = (((img0 + w2) * (unsigned)p.H + (unsigned)iy) * (unsigned)p.W + (unsigned)ix) * x_row_b + b_col;
            }
            dma16(rsrc_x, Bs + r * B_ROW_B, vb);
        }
        mstep += BKP;
        if constexpr (!PLAIN) {
            ox0 += BKP;
            const unsigned w = __umulhi(ox0, magicQ);
            ox0 -= w * (unsigned)p.Q;
            oy0 += w;
            const unsigned w2 = __umulhi(oy0, magicP);
            oy0 -= w2 * (unsigned)p.P;
            img0 += w2;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing-read lane constants: per 16-lane group a 4(pixel) x 16(channel) block, lane i16 gets the 4 pixels of channel i16
    const int g16 = lane >> 4, i16 = lane & 15;
    const int qq = i16 >> 2, pp = i16 & 3;
    // byte offset inside a row of this lane's 8-byte piece, swizzle included (row & 3 == qq: kr below is a multiple of 4)
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = wm * (BM / 2) + i * 32 + 16 * (g16 & 1) + 4 * pp;
        const int sw = A_ROW_B == 256 ? (qq << 2) : (((qq >> 1) & 1) << 2);
        a_off[i] = (((ch >> 3) ^ sw) << 4) + (ch & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = wn * (BN / 2) + j * 32 + 16 * (g16 & 1) + 4 * pp;
        b_off[j] = (((ch >> 3) ^ (qq << 2)) << 4) + (ch & 7) * 2;
    }

    constexpr int D = NST - 1;                                 // stages in flight
    constexpr int PER = A_I + B_I;                             // DMA instructions per wave per stage
    const int nk = (int)((mend - mbeg + BKP - 1) / BKP);
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (s0 < nk) issue(s0);
    for (int ks = 0; ks < nk; ++ks) {
        // stage ks has landed once at most min(D-1, nk-1-ks) younger stages are outstanding (vmcnt retires in issue order)
        const int younger = (nk - 1 - ks) < (D - 1) ? (nk - 1 - ks) : (D - 1);
        if (younger >= 2 && D >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (younger == 1 && D >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dma_barrier();                                         // everybody's pieces landed; everybody is done with stage ks-1
        if (ks + D < nk) issue((ks + D) % NST);
        const unsigned char* As = smem_raw + (size_t)(ks % NST) * STAGE;
        const unsigned char* Bs = As + A_STAGE;
#pragma unroll
        for (int s = 0; s < BKP / 16; ++s) {
            union Frag { bf16x8 v; s16x4 q[2]; };
            Frag a[TM], b[TN];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kr = 16 * s + 8 * hh + 4 * u + qq;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i].q[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(As + kr * A_ROW_B + a_off[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j].q[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + kr * B_ROW_B + b_off[j]));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
        }
    }

    float* slab = dsel + (long long)slice * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int qe = q0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = k0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (ko < p.KO && qe < p.QE) slab[(long long)ko * p.QE + qe] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient with SPECIALISED waves (round 3): a workgroup is 4 consumer waves (MFMA + transposing reads, the 2 x 2
// layout of wgrad_dma_kernel) and 4 loader waves that do nothing but issue the LDS-DMA of the stages ahead.  Why: one 1-KiB DMA
// piece costs the issuing wave ~100 cycles of instruction issue (MI355X_MICROARCH "LDS-DMA piece issue cost"), a stage of 32 pixels
// is 4 pieces but only 8 MFMA (256 cycles) per wave -- in wgrad_dma_kernel every wave spends more time issuing its pieces than
// multiplying, and an in-order wave cannot do both (450-650 TFLOP/s on the deep 1x1 layers).  Here the pieces are issued by other
// waves, on the same SIMDs, while the consumers multiply; one s_barrier per stage hands a landed stage over and frees the oldest.
//   loader   ks:  wait (own pieces of stage ks landed) | barrier | issue stage ks + D into the slot of stage ks - 1
//   consumer ks:  barrier | multiply stage ks
// ---------------------------------------------------------------------------------------------
template <int BM, int NST, bool PLAIN, int BKP>
__global__ __launch_bounds__(512) void wgrad_spec_kernel(WgradParams p, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int BN = 128;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_ROW_B = BM * 2, B_ROW_B = BN * 2;          // bytes per pixel row
    constexpr int A_STAGE = BKP * A_ROW_B, B_STAGE = BKP * B_ROW_B;
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int A_RPI = 1024 / A_ROW_B, B_RPI = 1024 / B_ROW_B;     // rows per wave-instruction (4 or 8)
    constexpr int A_I = BKP / A_RPI / 4, B_I = BKP / B_RPI / 4;       // instructions per wave per stage
    static_assert(A_I >= 1 && B_I >= 1, "stage too small for 4 waves");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;                                // index inside the role
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    const unsigned work = (blockIdx.x & 7) * (unsigned)p.per_xcd + (blockIdx.x >> 3);      // see wgrad_kernel
    if ((blockIdx.x >> 3) >= (unsigned)p.per_xcd || work >= (unsigned)p.total_z * (unsigned)p.tiles) return;
    const int z = (int)(work / (unsigned)p.tiles);
    const int tile = (int)(work % (unsigned)p.tiles);
    const int k0 = (tile % p.nkt) * BM;
    const int q0 = (tile / p.nkt) * BN;
    const int item = p.n_items ? z / p.nsplit : 0;
    const int slice = p.n_items ? z % p.nsplit : z;
    const long long mbeg = (long long)slice * p.m_per_split;
    long long mend = mbeg + p.m_per_split;
    if (mend > p.M) mend = p.M;

    const void* xsel = p.x;
    const void* gsel = p.dy;
    float* dsel = p.dw;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (p.n_items && item == i) { xsel = p.x_tab[i]; gsel = p.dy_tab[i]; dsel = p.dw_tab[i]; }
    const i32x4 rsrc_x = dma_rsrc(xsel, x_bytes);
    const i32x4 rsrc_g = dma_rsrc(gsel, dy_bytes);
    const unsigned smem_base = lds_addr(smem_raw);

    // ---- A (dy) lanes: row-in-instruction and logical channel chunk
    constexpr int A_CPR = A_ROW_B / 16;                        // chunks per row (16 or 8)
    const int a_rl = lane / A_CPR;                             // row inside one instruction
    const int a_slot = lane % A_CPR;
    const int a_swz = A_ROW_B == 256 ? ((a_rl & 3) << 2) : (((a_rl >> 1) & 1) << 2);
    const int a_ch = k0 + (a_slot ^ a_swz) * 8;
    const unsigned a_col = a_ch < p.KO ? (unsigned)a_ch * 2u : OOB;
    const unsigned g_row_b = (unsigned)p.KO * 2u;
    // ---- B (im2col of x) lanes
    const int b_rl = lane >> 4;
    const int b_slot = lane & 15;
    const int bq = q0 + (b_slot ^ ((b_rl & 3) << 2)) * 8;
    const bool b_ok = bq < p.QE;
    int b_kh = 0, b_kw = 0, b_c = 0;
    if (b_ok) {
        const int tap = bq / p.Cq;
        b_c = bq - tap * p.Cq;
        b_kh = tap / p.S;
        b_kw = tap - b_kh * p.S;
    }
    // Pixel walk.  The stage's first pixel (img0, oy0, ox0) is wave-uniform and advances in scalar registers; a lane's row is
    // that pixel + a lane constant < 32, folded back into (img, oy, ox) with two multiply-high divisions (exact: the
    // dividends stay below Q + 64 resp. P + 8) -- the per-row divergent carry loops of the register-staged kernel cost more
    // issue cycles per K-step than its 8 MFMAs.  PLAIN (1x1, stride 1, no padding): source pixel == destination pixel.
    const unsigned magicQ = 0xffffffffu / (unsigned)p.Q + 1u, magicP = 0xffffffffu / (unsigned)p.P + 1u;
    unsigned img0, oy0, ox0;
    {
        const long long img = mbeg / ((long long)p.P * p.Q);
        const int rem = (int)(mbeg - img * (long long)p.P * p.Q);
        img0 = (unsigned)img;
        oy0 = (unsigned)(rem / p.Q);
        ox0 = (unsigned)(rem - (int)oy0 * p.Q);
    }
    unsigned mstep = (unsigned)mbeg;                          // M * KO * 2 < 2 GiB: 32-bit pixel arithmetic throughout
    const unsigned mend32 = (unsigned)mend;
    const unsigned x_row_b = (unsigned)p.C * 2u;
    const unsigned b_col = (unsigned)b_c * 2u;
    const int b_dy = b_kh - p.pad, b_dx = b_kw - p.pad_x;

    auto issue = [&](int buf) {
        const unsigned As = smem_base + (unsigned)buf * STAGE;
        const unsigned Bs = As + A_STAGE;
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
            const int r = (wave * A_I + i) * A_RPI;            // first row of this instruction
            const unsigned m = mstep + (unsigned)(r + a_rl);
            const unsigned va = (m < mend32 && a_col != OOB) ? m * g_row_b + a_col : OOB;
            dma16(rsrc_g, As + r * A_ROW_B, va);
        }
#pragma unroll
        for (int i = 0; i < B_I; ++i) {
            const int r = (wave * B_I + i) * B_RPI;
            const unsigned m = mstep + (unsigned)(r + b_rl);
            unsigned vb = OOB;
            if constexpr (PLAIN) {
                if (b_ok && m < mend32) vb = m * x_row_b + b_col;
            } else {
                const unsigned t = ox0 + (unsigned)(r + b_rl);
                const unsigned w = __umulhi(t, magicQ);
                const unsigned ox = t - w * (unsigned)p.Q;
                const unsigned u = oy0 + w;
                const unsigned w2 = __umulhi(u, magicP);
                const unsigned oy = u - w2 * (unsigned)p.P;
                const int iy = (int)oy * p.stride + b_dy;
                const int ix = (int)ox * p.stride_x + b_dx;
                if (b_ok && m < mend32 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    vb = (((img0 + w2) * (unsigned)p.H + (unsigned)iy) * (unsigned)p.W + (unsigned)ix) * x_row_b + b_col;
            }
            dma16(rsrc_x, Bs + r * B_ROW_B, vb);
        }
        mstep += BKP;
        if constexpr (!PLAIN) {
            ox0 += BKP;
            const unsigned w = __umulhi(ox0, magicQ);
            ox0 -= w * (unsigned)p.Q;
            oy0 += w;
            const unsigned w2 = __umulhi(oy0, magicP);
            oy0 -= w2 * (unsigned)p.P;
            img0 += w2;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing-read lane constants: per 16-lane group a 4(pixel) x 16(channel) block, lane i16 gets the 4 pixels of channel i16
    const int g16 = lane >> 4, i16 = lane & 15;
    const int qq = i16 >> 2, pp = i16 & 3;
    // byte offset inside a row of this lane's 8-byte piece, swizzle included (row & 3 == qq: kr below is a multiple of 4)
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = wm * (BM / 2) + i * 32 + 16 * (g16 & 1) + 4 * pp;
        const int sw = A_ROW_B == 256 ? (qq << 2) : (((qq >> 1) & 1) << 2);
        a_off[i] = (((ch >> 3) ^ sw) << 4) + (ch & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = wn * (BN / 2) + j * 32 + 16 * (g16 & 1) + 4 * pp;
        b_off[j] = (((ch >> 3) ^ (qq << 2)) << 4) + (ch & 7) * 2;
    }

    constexpr int D = NST - 1;                                 // stages in flight
    constexpr int PER = A_I + B_I;                             // DMA instructions per loader wave per stage
    const int nk = (int)((mend - mbeg + BKP - 1) / BKP);
    if (loader) {
#pragma unroll
        for (int s0 = 0; s0 < D; ++s0)
            if (s0 < nk) issue(s0);
        for (int ks = 0; ks < nk; ++ks) {
            // stage ks has landed once at most min(D-1, nk-1-ks) younger stages are outstanding (vmcnt retires in issue order)
            const int younger = (nk - 1 - ks) < (D - 1) ? (nk - 1 - ks) : (D - 1);
            if (younger >= 2 && D >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (younger == 1 && D >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dma_barrier();                                     // every loader's pieces of stage ks landed; the consumers are done with stage ks-1
            if (ks + D < nk) issue((ks + D) % NST);
        }
        return;
    }
    for (int ks = 0; ks < nk; ++ks) {
        dma_barrier();
        const unsigned char* As = smem_raw + (size_t)(ks % NST) * STAGE;
        const unsigned char* Bs = As + A_STAGE;
#pragma unroll
        for (int s = 0; s < BKP / 16; ++s) {
            union Frag { bf16x8 v; s16x4 q[2]; };
            Frag a[TM], b[TN];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kr = 16 * s + 8 * hh + 4 * u + qq;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i].q[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(As + kr * A_ROW_B + a_off[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j].q[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + kr * B_ROW_B + b_off[j]));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
        }
    }

    float* slab = dsel + (long long)slice * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int qe = q0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = k0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (ko < p.KO && qe < p.QE) slab[(long long)ko * p.QE + qe] = acc[i][j][r];
            }
        }
}

// number of split-K slices for KO x QE outputs over M pixels with a BM x 128 tile: ~512 workgroups (2 per CU);
// every slice costs one extra write + read of the whole dW in fp32, so no more than needed to fill the chip
int wgrad_splits(long long M, int KO, int QE, int BM, int n_items = 1) {
    const int tiles = cs_ceil_div(KO, BM) * cs_ceil_div(QE, 128) * (n_items > 1 ? n_items : 1);
    static const int target = cs_env_int_("CELLSEG_WGRAD_BLOCKS", 512);   // experiments only
    long long want = (target + tiles - 1) / tiles;
    const long long max_split = (M + 63) / 64;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    long long per = (M + want - 1) / want;
    per = ((per + 31) / 32) * 32;
    return (int)((M + per - 1) / per);
}

const int g_wgrad_nst = cs_env_int_("CELLSEG_WGRAD_NST", 3);   // A/B experiments only
const bool g_wgrad_dma = !cs_env_flag_("CELLSEG_WGRAD_REG");   // A/B experiments only

template <typename T, int BM, int BN, bool TR>
int launch_wgrad(WgradParams p, hipStream_t st, int n_items = 1) {
    constexpr int BKP = 32;
    constexpr int PADE = sizeof(T) == 2 ? 32 : 0;
    const int nsplit = wgrad_splits(p.M, p.KO, p.QE, BM, n_items);
    p.nsplit = nsplit;
    long long per = (p.M + nsplit - 1) / nsplit;
    per = ((per + BKP - 1) / BKP) * BKP;
    p.m_per_split = per;
    p.slab_stride = (long long)p.KO * p.QE;
    constexpr size_t lds = 2ull * BKP * (BM + BN + 2 * PADE) * sizeof(T);
    p.nkt = cs_ceil_div(p.KO, BM);
    p.tiles = p.nkt * cs_ceil_div(p.QE, BN);
    p.total_z = nsplit * (n_items > 1 ? n_items : 1);
    p.per_xcd = (p.total_z * p.tiles + 7) / 8;
    dim3 grid((unsigned)(p.per_xcd * 8), 1, 1);
    if constexpr (sizeof(T) == 2 && TR && BN == 128) {
        const unsigned long long x_bytes = (unsigned long long)(p.M / ((long long)p.P * p.Q)) * p.H * p.W * p.C * 2ull;
        const unsigned long long g_bytes = (unsigned long long)p.M * p.KO * 2ull;
        if (g_wgrad_dma && !p.slab && x_bytes < 0x80000000ull && g_bytes < 0x80000000ull) {
            const bool plain = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
#define CS_WGRAD_DMA(NST_, BKP_) \
    do { \
        constexpr size_t stage_ = (size_t)BKP_ * (BM + 128) * 2; \
        note_variant("wgrad_dma_kernel<%d,%d,%s,%d>", BM, NST_, plain ? "true" : "false", BKP_); \
        if (plain) hipLaunchKernelGGL((wgrad_dma_kernel<BM, NST_, true, BKP_>), grid, dim3(256), NST_ * stage_, st, p, (unsigned)x_bytes, (unsigned)g_bytes); \
        else hipLaunchKernelGGL((wgrad_dma_kernel<BM, NST_, false, BKP_>), grid, dim3(256), NST_ * stage_, st, p, (unsigned)x_bytes, (unsigned)g_bytes); \
    } while (0)
            // loader / consumer specialisation pays on the deep layers (tools/wgrad_ab.sh, 4 layers per launch: 256 -> 1024 @19x19 61 -> 52 us,
            // 2048 -> 512 @10x10 65 -> 56, strided 512 -> 1024 139 -> 107, 1024 -> 512 110 -> 93) and on the pixel-paired stem (106 -> 85 us);
            // the HBM-bound early layers (64 -> 256 @75x75: 146 vs 153 us) keep the four-wave kernel.  In the step: 1x1 family 1.113 -> 1.070 ms,
            // stem 0.106 -> 0.085.  CELLSEG_WGRAD_SPEC = 1 forces the specialised kernel, 2 the four-wave one (A/B).
            static const int spec_knob = cs_env_int_("CELLSEG_WGRAD_SPEC", 0);
            const bool deep = BM == 128 && p.KO >= 256 && p.QE >= 256, stem = !plain && p.C <= 8;
            if (spec_knob == 1 || (spec_knob == 0 && (deep || stem))) {
                constexpr size_t stage_ = (size_t)32 * (BM + 128) * 2;
                note_variant("wgrad_spec_kernel<%d,%d,%s,%d>", BM, 4, plain ? "true" : "false", 32);
                if (plain) hipLaunchKernelGGL((wgrad_spec_kernel<BM, 4, true, 32>), grid, dim3(512), 4 * stage_, st, p, (unsigned)x_bytes, (unsigned)g_bytes);
                else hipLaunchKernelGGL((wgrad_spec_kernel<BM, 4, false, 32>), grid, dim3(512), 4 * stage_, st, p, (unsigned)x_bytes, (unsigned)g_bytes);
                CS_LAUNCH_CHECK();
                return CS_OK;
            }
            if (g_wgrad_nst == 2) CS_WGRAD_DMA(2, 32);
            else if (g_wgrad_nst == 4) CS_WGRAD_DMA(4, 32);
            else if (g_wgrad_nst == 64) CS_WGRAD_DMA(2, 64);
            else CS_WGRAD_DMA(3, 32);
#undef CS_WGRAD_DMA
            CS_LAUNCH_CHECK();
            return CS_OK;
        }
    }
    note_variant("wgrad_kernel<%s,%d,%d,%s>", tname<T>(), BM, BN, TR ? "true" : "false");
    hipLaunchKernelGGL((wgrad_kernel<T, BM, BN, TR>), grid, dim3(256), lds, st, p);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

}  // namespace

extern "C" int cs_conv2d_wgrad_splits(const CsConvGeom* g, int grouped) {
    if (!g) return 0;
    const bool wide = g->K > 64 && !grouped;
    const int cq = grouped ? 64 : g->C;
    return wgrad_splits((long long)g->N * g->P * g->Q, g->K, g->R * g->S * cq, wide ? 128 : 64);
}

extern "C" int cs_conv2d_wgrad_batched_splits(const CsConvGeom* g, int dtype, int n_items) {
    if (!g || n_items < 1) return 0;
    const int v2 = cs_wgrad2_splits_(g, dtype, n_items);         // bf16 stride-1 3x3 with C, K multiples of 64: wgrad_v2.hip
    if (v2 > 0) return v2;
    return wgrad_splits((long long)g->N * g->P * g->Q, g->K, g->R * g->S * g->C, g->K > 64 ? 128 : 64, n_items);
}

extern "C" int cs_conv2d_wgrad2_supported(const CsConvGeom* g, int dtype) { return g && cs_wgrad2_splits_(g, dtype, 1) > 0 ? 1 : 0; }

extern "C" int cs_conv2d_wgrad_batched(const CsConvGeom* g, int dtype, const void* const* x_tab, const void* const* dy_tab,
                                       float* const* dw_tab, int n_items, int use_tr_read, void* stream) {
    int rc = check_geom(g, dtype);
    if (rc != CS_OK) return rc;
    CS_CHECK_ARG(x_tab && dy_tab && dw_tab && n_items >= 1 && n_items <= 8, "conv2d_wgrad_batched: 1..8 items, HOST pointer arrays");
    if (cs_wgrad2_splits_(g, dtype, n_items) > 0) return cs_wgrad2_launch_(g, dtype, x_tab, dy_tab, dw_tab, n_items, stream);
    WgradParams p{};
    const int ce = dtype == CS_F32 ? 4 : 8;
    for (int i = 0; i < n_items; ++i) { p.x_tab[i] = x_tab[i]; p.dy_tab[i] = dy_tab[i]; p.dw_tab[i] = dw_tab[i]; }
    p.n_items = n_items;
    p.H = g->H; p.W = g->W; p.C = g->C;
    p.P = g->P; p.Q = g->Q; p.KO = g->K;
    p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.stride_x = g->stride; p.pad_x = g->pad;
    p.M = (long long)g->N * g->P * g->Q;
    p.Cq = g->C; p.slab = 0;
    p.QE = g->R * g->S * p.Cq;
    p.SCc = g->C / ce;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool wide = g->K > 64;
    if (dtype == CS_F32) return wide ? launch_wgrad<float, 128, 128, false>(p, st, n_items) : launch_wgrad<float, 64, 128, false>(p, st, n_items);
    if (use_tr_read) return wide ? launch_wgrad<bf16_t, 128, 128, true>(p, st, n_items) : launch_wgrad<bf16_t, 64, 128, true>(p, st, n_items);
    return wide ? launch_wgrad<bf16_t, 128, 128, false>(p, st, n_items) : launch_wgrad<bf16_t, 64, 128, false>(p, st, n_items);
}

extern "C" int cs_conv2d_wgrad(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_khwc,
                               int use_tr_read, void* stream) {
    const int slab = (g && g->groups > 1) ? 1 : 0;
    int rc = check_geom(g, dtype);
    if (rc != CS_OK) return rc;
    if (slab) CS_CHECK_ARG(g->C % 64 == 0 && g->K == g->C, "grouped conv: width must be a multiple of 64 and C == K");
    CS_CHECK_ARG(x && dy && dw_khwc, "conv2d_wgrad: NULL tensor");
    WgradParams p{};
    const int ce = dtype == CS_F32 ? 4 : 8;
    p.x = x; p.dy = dy; p.dw = dw_khwc;
    p.H = g->H; p.W = g->W; p.C = g->C;
    p.P = g->P; p.Q = g->Q; p.KO = g->K;
    p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.stride_x = g->stride; p.pad_x = g->pad;
    p.M = (long long)g->N * g->P * g->Q;
    p.Cq = slab ? 64 : g->C;
    p.slab = slab;
    p.QE = g->R * g->S * p.Cq;
    p.SCc = g->C / ce;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool wide = g->K > 64 && !slab;      // slab mode: one 64-channel slab per M tile
    if (dtype == CS_F32) {
        if (wide) return launch_wgrad<float, 128, 128, false>(p, st);
        return launch_wgrad<float, 64, 128, false>(p, st);
    }
    if (use_tr_read) {
        if (wide) return launch_wgrad<bf16_t, 128, 128, true>(p, st);
        return launch_wgrad<bf16_t, 64, 128, true>(p, st);
    }
    if (wide) return launch_wgrad<bf16_t, 128, 128, false>(p, st);
    return launch_wgrad<bf16_t, 64, 128, false>(p, st);
}

// =============================================================================================
// Stem on a pixel-PAIRED image (7x7 / stride 2 / pad 3, 3 input channels: model/resnet.py:111).
// With the 3 channels padded to one 16-byte chunk per pixel the implicit GEMM walks 49 chunks of which 147/392 elements are
// real: 2.7x the MFMA work and DMA count of the arithmetic.  Storing TWO neighbouring pixels x 4 channels per chunk turns the
// layer into a 7x4-tap convolution over [N][H][ceil(W/2)][8] with row stride 2, pair stride 1, pad (3, 2): 28 chunks, 147/224
// real.  Output pixel px reads pairs px-2 .. px+1 = pixels 2px-4 .. 2px+3; tap (kh, kw') element e holds filter column
// kw = 2*kw' - 1 + e/4 (kw = -1: zero) and channel e % 4 (channel 3: zero).  Same igemm / weight-gradient kernels, two
// extra parameters (x stride / x pad); three tiny kernels convert the image, the staged weights and the raw gradient.
// =============================================================================================
namespace {

template <typename T>
__global__ __launch_bounds__(256) void stem_pair_input_kernel(const T* __restrict__ x, int N, int H, int W, int Wh, T* __restrict__ out) {
    // one thread per output chunk: out[n][h][j][0..7] = {x[n][h][2j][0..3], x[n][h][2j+1][0..3]}
    const long long total = (long long)N * H * Wh;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % Wh);
        const long long nh = idx / Wh;
        const T* src = x + (nh * W + 2 * j) * 8;
        T v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = src[e];
            v[4 + e] = (2 * j + 1 < W) ? src[8 + e] : from_f32<T>(0.f);
        }
        v[3] = from_f32<T>(0.f); v[7] = from_f32<T>(0.f);
        T* dst = out + idx * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = v[e];
    }
}

// the same chunks straight from the fp32 NCHW image the drivers hand over (train_tile.py:264 `input.to(device)`): the NHWC8 intermediate
// (91 MB written, 91 MB read per bag of 64 tiles) is never made.  Consecutive threads take consecutive pairs: 8 contiguous bytes per
// thread and channel plane.
template <typename T>
__global__ __launch_bounds__(256) void stem_pair_from_nchw_kernel(const float* __restrict__ x, int N, int H, int W, int Wh, T* __restrict__ out) {
    const long long total = (long long)N * H * Wh, plane = (long long)H * W;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % Wh);
        const long long nh = idx / Wh;
        const long long n = nh / H;
        const int h = (int)(nh - n * H);
        const float* src = x + (n * 3) * plane + (long long)h * W + 2 * j;
        const bool two = 2 * j + 1 < W;
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = src[c * plane];
            v[4 + c] = two ? src[c * plane + 1] : 0.f;
        }
        v[3] = 0.f; v[7] = 0.f;
        store8<T>(out + idx * 8, v);
    }
}

template <typename T>
__global__ void stem_pair_weights_kernel(const T* __restrict__ w /*[K][7][7][8]*/, int K, T* __restrict__ out /*[K][7][4][8]*/) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * 7 * 4 * 8) return;
    const int e = idx & 7, kwp = (idx >> 3) & 3, kh = (idx >> 5) % 7, k = idx / (32 * 7);
    const int kw = 2 * kwp - 1 + (e >> 2), c = e & 3;
    out[idx] = (kw >= 0 && kw < 7 && c < 3) ? w[((k * 7 + kh) * 7 + kw) * 8 + c] : from_f32<T>(0.f);
}

// raw paired slabs [nsplit][K][7][4][8] -> one slab in the ordinary [K][7][7][8] layout (splits summed, padding lanes dropped)
__global__ void stem_unpair_slabs_kernel(const float* __restrict__ slabs, int nsplit, int K, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * 49 * 8) return;
    const int c = idx & 7, kw = (idx >> 3) % 7, kh = (idx / 56) % 7, k = idx / 392;
    float v = 0.f;
    if (c < 3) {
        const int kwp = (kw + 1) >> 1, e = 4 * ((kw + 1) & 1) + c;
        const float* src = slabs + ((long long)(k * 7 + kh) * 4 + kwp) * 8 + e;
        const long long stride = (long long)K * 224;
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.f;
        int s = 0;
        for (; s + 8 <= nsplit; s += 8) {          // the stem is split ~256 ways: keep 8 loads in flight per thread
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += src[(s + u) * stride];
        }
        for (; s < nsplit; ++s) a[0] += src[s * stride];
        v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    out[idx] = v;
}

}  // namespace

extern "C" int cs_stem_pair_input(const void* x_nhwc8, int dtype, int N, int H, int W, void* x_pair, void* stream) {
    CS_CHECK_ARG(x_nhwc8 && x_pair && N > 0 && H > 0 && W > 0, "stem_pair_input: bad arguments");
    const int Wh = (W + 1) / 2;
    const long long total = (long long)N * H * Wh;
    long long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CS_F32) hipLaunchKernelGGL(stem_pair_input_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)x_nhwc8, N, H, W, Wh, (float*)x_pair);
    else if (dtype == CS_BF16) hipLaunchKernelGGL(stem_pair_input_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)x_nhwc8, N, H, W, Wh, (bf16_t*)x_pair);
    else CS_CHECK_ARG(false, "stem_pair_input: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_stem_pair_from_nchw(const float* x_nchw, int dtype, int N, int H, int W, void* x_pair, void* stream) {
    CS_CHECK_ARG(x_nchw && x_pair && N > 0 && H > 0 && W > 0, "stem_pair_from_nchw: bad arguments");
    CS_CHECK_ARG(dtype == CS_BF16, "stem_pair_from_nchw: bf16 pairs only (the fp32 stem takes 4-channel NHWC pixels)");
    const int Wh = (W + 1) / 2;
    const long long total = (long long)N * H * Wh;
    long long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(stem_pair_from_nchw_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x_nchw, N, H, W, Wh,
                       (bf16_t*)x_pair);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_stem_pair_weights(const void* w_khwc, int dtype, int K, void* w_pair, void* stream) {
    CS_CHECK_ARG(w_khwc && w_pair && K > 0, "stem_pair_weights: bad arguments");
    const int total = K * 7 * 4 * 8;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CS_F32) hipLaunchKernelGGL(stem_pair_weights_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)w_khwc, K, (float*)w_pair);
    else if (dtype == CS_BF16) hipLaunchKernelGGL(stem_pair_weights_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)w_khwc, K, (bf16_t*)w_pair);
    else CS_CHECK_ARG(false, "stem_pair_weights: bad dtype");
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_stem_fwd(int N, int H, int W, int K, int dtype, const void* x_pair, const void* w_pair, const float* scale,
                           const float* shift, int act, void* y, double* stats, void* workspace, void* stream) {
    CS_CHECK_ARG(x_pair && w_pair && y && N > 0 && H >= 7 && W >= 7 && K > 0 && K % 8 == 0, "stem_fwd: bad arguments");
    CS_CHECK_ARG(dtype == CS_F32 || dtype == CS_BF16, "stem_fwd: bad dtype");
    CS_CHECK_ARG(!stats || workspace, "stem_fwd: stats need a workspace of cs_conv2d_stats_workspace() bytes");
    const int ce = dtype == CS_F32 ? 4 : 8;
    const int P = (H + 6 - 7) / 2 + 1, Q = (W + 6 - 7) / 2 + 1, Wh = (W + 1) / 2;
    IgemmParams p{};
    p.src = x_pair; p.wgt = w_pair; p.dst = y;
    p.scale = scale; p.shift = shift; p.residual = nullptr; p.mask = nullptr;
    p.slab = stats ? reinterpret_cast<float*>(workspace) : nullptr;
    p.SH = H; p.SW = Wh; p.SC = 8;
    p.DH = P; p.DW = Q; p.NOUT = K;
    p.R = 7; p.S = 4;
    p.mul = 2; p.mulx = 1; p.div = 1; p.off0 = -3; p.off0x = -2; p.sgn = 1;
    p.wk0y = 0; p.wk0x = 0; p.wkstep = 1; p.S_full = 4;
    p.dst_step = 1; p.dst_oy = 0; p.dst_ox = 0; p.DHF = P; p.DWF = Q;
    p.act = act;
    p.M = (long long)N * P * Q;
    p.src_pixels = (long long)N * H * Wh;
    p.SCc = 8 / ce;
    p.cslab = 0;
    p.Qtot = 7 * 4 * p.SCc;
    p.wrow_chunks = p.Qtot;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == CS_F32 ? dispatch_igemm<float>(p, nullptr, stats, st) : dispatch_igemm<bf16_t>(p, nullptr, stats, st);
}

static void stem_wgrad_params(WgradParams& p, int N, int H, int W, int K, int dtype) {
    const int ce = dtype == CS_F32 ? 4 : 8;
    p.H = H; p.W = (W + 1) / 2; p.C = 8;
    p.P = (H + 6 - 7) / 2 + 1; p.Q = (W + 6 - 7) / 2 + 1; p.KO = K;
    p.R = 7; p.S = 4; p.stride = 2; p.pad = 3; p.stride_x = 1; p.pad_x = 2;
    p.M = (long long)N * p.P * p.Q;
    p.Cq = 8; p.slab = 0;
    p.QE = 7 * 4 * 8;
    p.SCc = 8 / ce;
}

extern "C" int cs_stem_wgrad_splits(int N, int H, int W, int K) {
    if (N <= 0 || H < 7 || W < 7 || K <= 0) return 0;
    const long long M = (long long)N * ((H + 6 - 7) / 2 + 1) * ((W + 6 - 7) / 2 + 1);
    return wgrad_splits(M, K, 224, K > 64 ? 128 : 64);
}

extern "C" int cs_stem_wgrad(int N, int H, int W, int K, int dtype, const void* x_pair, const void* dy, float* dw_pair_slabs,
                             int use_tr_read, void* stream) {
    CS_CHECK_ARG(x_pair && dy && dw_pair_slabs && N > 0 && H >= 7 && W >= 7 && K > 0 && K % 8 == 0, "stem_wgrad: bad arguments");
    CS_CHECK_ARG(dtype == CS_F32 || dtype == CS_BF16, "stem_wgrad: bad dtype");
    WgradParams p{};
    p.x = x_pair; p.dy = dy; p.dw = dw_pair_slabs;
    stem_wgrad_params(p, N, H, W, K, dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool wide = K > 64;
    if (dtype == CS_F32) return wide ? launch_wgrad<float, 128, 128, false>(p, st) : launch_wgrad<float, 64, 128, false>(p, st);
    if (use_tr_read) return wide ? launch_wgrad<bf16_t, 128, 128, true>(p, st) : launch_wgrad<bf16_t, 64, 128, true>(p, st);
    return wide ? launch_wgrad<bf16_t, 128, 128, false>(p, st) : launch_wgrad<bf16_t, 64, 128, false>(p, st);
}

extern "C" int cs_stem_unpair_slabs(const float* dw_pair_slabs, int nsplit, int K, float* dw_khwc, void* stream) {
    CS_CHECK_ARG(dw_pair_slabs && dw_khwc && nsplit >= 1 && K > 0, "stem_unpair_slabs: bad arguments");
    const int total = K * 392;
    hipLaunchKernelGGL(stem_unpair_slabs_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dw_pair_slabs, nsplit,
                       K, dw_khwc);
    CS_LAUNCH_CHECK();
    return CS_OK;
}
