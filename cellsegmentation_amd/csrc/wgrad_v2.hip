// Second-generation bf16 weight gradient of the stride-1, pad-1 3x3 convolutions (model/resnet.py:51-53 conv2 of every block and
// its autograd backward), C and K multiples of 64.  Same raw output as cs_conv2d_wgrad_batched: split-K fp32 slabs
// [nsplit][K][3][3][C] that cs_wgrad_finalize_batched folds.
//
// Why a second kernel (conv_igemm.hip's wgrad_dma_kernel: 530-670 TFLOP/s, 11.5 VALU + 7.5 SALU per MFMA): that kernel tiles
// K x (tap, channel) and re-gathers the source pixels for every tap column block -- per 8 MFMA 32 VALU of LDS addressing plus the
// per-stage gather arithmetic.  Here the contraction runs over PADDED-LINEAR POSITIONS D = (n * Hp + y) * Wp + x (conv_v2.hip's
// coordinates: one shared zero slot between image rows and one zero row between images):
//     dW[k][r][s][c] = sum_D  dy_pad[D][k] * x_pad[D + r * Wp + s][c]
// so a tap is a constant ROW SHIFT of the same staged rows: one workgroup stages, per run of BL positions, BL rows of dy (one
// 64-channel chunk) and BL + 2 * Wp + 2 rows of x (one 64-channel chunk) ONCE and multiplies all nine taps out of them.
//   * 4 waves as 2 (k) x 2 (c); a wave owns 32 k x 32 c x 9 taps = 144 accumulator registers;
//   * both MFMA operands are "transposed" w.r.t. the [position][channel] rows in LDS: ds_read_b64_tr_b16.  Rows are 128 bytes with
//     the 16-byte chunk index XOR-ed by 4 * bit 1 of the row -- the 32 eight-byte pieces a half-wave reads (4 rows x 4 chunks x 2)
//     then cover all 64 banks once.  Adding 4 or 16 rows keeps bits 0-1 of a row, so the per-lane address of (tap, lane) is
//     computed ONCE per workgroup and every read of the loop is that register + an immediate: no VALU in the loop;
//   * main loop on owned registers (conv_v2.hip): v[82:91] addresses, v[92:99] dy fragments (2 sets), v[100:111] x fragments
//     (3 in flight), v[112:255] accumulators; every MFMA waits for exactly its own fragment (lgkmcnt 4, or 6 while the next
//     step's dy fragment is in flight);
//   * two LDS stages; the DMA of stage s+1 is issued right after the barrier that opens stage s (vmcnt(0) at the next barrier is
//     exact: nothing else is ever in flight);
//   * positions outside an image (pad slots, the run past the last image) are out-of-range buffer offsets: zeros, no branches.
#include "cs_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <utility>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

struct W2Params {
    const void* x_tab[8];
    const void* dy_tab[8];
    float* dw_tab[8];
    int N, H, W, C, K;
    int Wp, Hp;
    unsigned mg_wp, sh_wp, mg_hp, sh_hp;
    unsigned n_stages;            // ceil(N * Hp * Wp / BL)
    unsigned stages_per_split;
    int nsplit, n_items, n_kt, n_ct;
    unsigned x_bytes, dy_bytes, slab_bytes;      // one split's slab: K * 9 * C * 4
    unsigned per_xcd, total_work;
};

__device__ __forceinline__ unsigned udivm(unsigned n, unsigned mg, unsigned sh) { return __umulhi(n, mg) >> sh; }
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p; }

constexpr int W_AD = 82, W_FA = 92, W_FB = 100, W_ACC = 112;

template <int REG> __device__ __forceinline__ void vset(unsigned v) { asm volatile("v_mov_b32 v[%c0], %1" ::"i"(REG), "v"(v)); }
template <int REG> __device__ __forceinline__ void vzero() { asm volatile("v_mov_b32 v[%c0], 0" ::"i"(REG)); }
template <int... Rs> __device__ __forceinline__ void vzero_all(std::integer_sequence<int, Rs...>) { (vzero<W_ACC + Rs>(), ...); }
template <int REG> __device__ __forceinline__ void vadd_s(unsigned s) { asm volatile("v_add_u32 v[%c0], %1, v[%c0]" ::"i"(REG), "s"(s)); }
template <int... Rs> __device__ __forceinline__ void vadd_all(unsigned s, std::integer_sequence<int, Rs...>) { (vadd_s<W_AD + Rs>(s), ...); }
template <int REG> __device__ __forceinline__ float vget() {
    float x;
    asm volatile("v_mov_b32 %0, v[%c1]" : "=v"(x) : "i"(REG));
    return x;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }

// the two transposing reads of one 32-channel x 16-position fragment: positions +0..3 and +4..7 of the lane's half (u = 0, 1)
template <int DST, int ADDR, unsigned OFF> __device__ __forceinline__ void frag_rd() {
    asm volatile("ds_read_b64_tr_b16 v[%c0:%c1], v[%c4] offset:%c5\n\t"
                 "ds_read_b64_tr_b16 v[%c2:%c3], v[%c4] offset:%c6"
                 ::"i"(DST), "i"(DST + 1), "i"(DST + 2), "i"(DST + 3), "i"(ADDR), "i"(OFF), "i"(OFF + 512u));
}
template <int ACC, int A, int B> __device__ __forceinline__ void mfma() {
    asm volatile("v_mfma_f32_32x32x16_bf16 v[%c0:%c1], v[%c2:%c3], v[%c4:%c5], v[%c0:%c1]" ::"i"(ACC), "i"(ACC + 15), "i"(A), "i"(A + 3), "i"(B),
                 "i"(B + 3));
}

// flattened (position step j, tap t) index I of one stage: the whole stage is one straight line of code
template <int NJ, int I> __device__ __forceinline__ void stage_item() {
    constexpr int TOTAL = 9 * NJ;
    constexpr int j = I / 9, t = I % 9;
    constexpr int younger = (TOTAL - 1 - I) < 2 ? (TOTAL - 1 - I) : 2;            // x fragments issued after this one
    constexpr bool a_inflight = (t >= 3 && t <= 5) && (j + 1 < NJ);            // the next step's dy fragment, issued after MFMA(9j+2)
    wait_lgkm<2 * younger + (a_inflight ? 2 : 0)>();
    mfma<W_ACC + 16 * t, W_FA + 4 * (j & 1), W_FB + 4 * (I % 3)>();
    if constexpr (I + 3 < TOTAL) {
        constexpr int j3 = (I + 3) / 9, t3 = (I + 3) % 9;
        frag_rd<W_FB + 4 * (I % 3), W_AD + 1 + t3, (unsigned)j3 * 2048u>();
    }
    if constexpr (t == 2 && j + 1 < NJ) frag_rd<W_FA + 4 * ((j + 1) & 1), W_AD, (unsigned)(j + 1) * 2048u>();
}
template <int NJ, int... Is> __device__ __forceinline__ void stage_items(std::integer_sequence<int, Is...>) { (stage_item<NJ, Is>(), ...); }

template <int BL, int XR>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(41))) void wgrad2_kernel(W2Params p) {
    static_assert(BL % 16 == 0 && XR % 8 == 0 && XR >= BL, "stage shape");
    constexpr unsigned DYB = BL * 128u, XB = XR * 128u, STAGE = DYB + XB;
    constexpr int NJ = BL / 16;
    constexpr int NP_DY = BL / 8;                                  // 1 KiB DMA pieces (8 rows) of the dy region; the x region has XR / 8
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    // work list: (item, split) major, the (k tile, c tile) pairs of one position run adjacent, cut into 8 runs (one per XCD)
    const unsigned w = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= p.per_xcd || w >= p.total_work) return;
    const unsigned n_tiles = (unsigned)(p.n_kt * p.n_ct);
    const unsigned tile = w % n_tiles, z = w / n_tiles;
    const int kt = (int)(tile % (unsigned)p.n_kt), ct = (int)(tile / (unsigned)p.n_kt);
    const int item = (int)(z / (unsigned)p.nsplit), split = (int)(z % (unsigned)p.nsplit);
    const void* xsel = p.x_tab[0];
    const void* gsel = p.dy_tab[0];
    float* dsel = p.dw_tab[0];
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (item == i) { xsel = p.x_tab[i]; gsel = p.dy_tab[i]; dsel = p.dw_tab[i]; }
    const i32x4 rsrc_x = make_rsrc(xsel, p.x_bytes);
    const i32x4 rsrc_g = make_rsrc(gsel, p.dy_bytes);
    const unsigned smem_base = lds_off(smem);
    const unsigned s_beg = (unsigned)split * p.stages_per_split;
    unsigned s_end = s_beg + p.stages_per_split;
    if (s_end > p.n_stages) s_end = p.n_stages;

    // ---- DMA of one stage: position piece q = wave + 4i = positions d0 + 8q .. 8q+7.  Row 8q+j of the x region and row 8q+j of the
    // dy region are the SAME position, so one decode (two multiply-high divisions) serves both pieces.
    const unsigned xc_bytes = (unsigned)p.C * 2u, gc_bytes = (unsigned)p.K * 2u;
    auto issue = [&](unsigned s, unsigned buf) {
        const unsigned d0 = s * (unsigned)BL;
#pragma unroll
        for (int i = 0; i < (XR / 8 + 3) / 4; ++i) {
            const int q = wave + 4 * i;
            if (q < XR / 8) {
                const unsigned row = (unsigned)q * 8u + (unsigned)(lane >> 3);
                const unsigned pos = d0 + row;
                const unsigned rq = udivm(pos, p.mg_wp, p.sh_wp);
                const unsigned col = pos - rq * (unsigned)p.Wp;
                const unsigned n = udivm(rq, p.mg_hp, p.sh_hp);
                const unsigned ry = rq - n * (unsigned)p.Hp;
                const unsigned chunk = ((unsigned)(lane & 7) ^ (((row >> 1) & 1u) << 2)) * 16u;
                const bool in_n = n < (unsigned)p.N;
                const unsigned pix = (n * (unsigned)p.H + ry) * (unsigned)p.W + col;               // as a dy pixel; the x pixel is one row and one column back
                const unsigned vx = (in_n && col >= 1u && ry >= 1u) ? (pix - (unsigned)p.W - 1u) * xc_bytes + (unsigned)ct * 128u + chunk : OOB;
                const unsigned dstx = __builtin_amdgcn_readfirstlane(smem_base + buf * STAGE + DYB + (unsigned)q * 1024u);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dstx), "v"(vx), "s"(rsrc_x) : "memory");
                if (q < NP_DY) {
                    const unsigned vg = (in_n && col < (unsigned)p.W && ry < (unsigned)p.H) ? pix * gc_bytes + (unsigned)kt * 128u + chunk : OOB;
                    const unsigned dstg = __builtin_amdgcn_readfirstlane(smem_base + buf * STAGE + (unsigned)q * 1024u);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dstg), "v"(vg), "s"(rsrc_g) : "memory");
                }
            }
        }
    };

    // ---- operand addresses (stage 0), once: piece = 4 channels x 8 bytes of row (base row + 8 * hh + (lane & 15) / 4)
    {
        const unsigned i16 = (unsigned)lane & 15u, g16 = (unsigned)lane >> 4;
        const unsigned qq = i16 >> 2, pp = i16 & 3u;
        const unsigned cin = 2u * (g16 & 1u) + (pp >> 1), sub = pp & 1u;
        auto addr = [&](unsigned region, unsigned row0, unsigned chunk0) {
            const unsigned row = row0 + 8u * (unsigned)hh + qq;
            return smem_base + region + row * 128u + (((chunk0 + cin) ^ (((row >> 1) & 1u) << 2)) << 4) + sub * 8u;
        };
        asm volatile("; v[82:255] are owned by the main loop" ::: "v82", "v255");
        vset<W_AD>(addr(0u, 0u, (unsigned)wk * 4u));
        const unsigned wp = (unsigned)p.Wp, cb = (unsigned)wc * 4u;
        vset<W_AD + 1>(addr(DYB, 0u, cb));          vset<W_AD + 2>(addr(DYB, 1u, cb));          vset<W_AD + 3>(addr(DYB, 2u, cb));
        vset<W_AD + 4>(addr(DYB, wp, cb));          vset<W_AD + 5>(addr(DYB, wp + 1u, cb));     vset<W_AD + 6>(addr(DYB, wp + 2u, cb));
        vset<W_AD + 7>(addr(DYB, 2u * wp, cb));     vset<W_AD + 8>(addr(DYB, 2u * wp + 1u, cb)); vset<W_AD + 9>(addr(DYB, 2u * wp + 2u, cb));
    }
    vzero_all(std::make_integer_sequence<int, 144>{});

    issue(s_beg, 0u);
    unsigned buf = 0;
    for (unsigned s = s_beg; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage s (nothing else is ever in flight)
        asm volatile("s_barrier" ::: "memory");                // everybody's pieces; everybody is done with the other buffer
        if (s + 1 < s_end) issue(s + 1, buf ^ 1u);
        // cold start of the stage: the dy fragment of step 0 and the first three x fragments
        frag_rd<W_FA, W_AD, 0u>();
        frag_rd<W_FB, W_AD + 1, 0u>();
        frag_rd<W_FB + 4, W_AD + 2, 0u>();
        frag_rd<W_FB + 8, W_AD + 3, 0u>();
        stage_items<NJ>(std::make_integer_sequence<int, 9 * NJ>{});
        // the operand addresses follow the stage
        vadd_all(buf ? 0u - STAGE : STAGE, std::make_integer_sequence<int, 10>{});
        buf ^= 1u;
    }

    // ---- this split's slab [K][9][C] fp32: register r of tap t = row k0 + (r & 3) + 8 * (r >> 2) + 4 * hh, column c0 + l31
    float* slab = dsel + (size_t)split * (p.slab_bytes / 4u);
    __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(slab, 0, p.slab_bytes, 0x00020000);
    const unsigned k0 = (unsigned)(kt * 64 + wk * 32), c0 = (unsigned)(ct * 64 + wc * 32);
    const unsigned lane_off = (((k0 + 4u * (unsigned)hh) * 9u) * (unsigned)p.C + c0 + (unsigned)l31) * 4u;
    const unsigned row_bytes = 9u * (unsigned)p.C * 4u, tap_bytes = (unsigned)p.C * 4u;
    [&]<int... Is>(std::integer_sequence<int, Is...>) {
        ((__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vget<W_ACC + Is>()), r_out, lane_off,
                                                (unsigned)(((Is % 16) & 3) + 8 * ((Is % 16) >> 2)) * row_bytes + (unsigned)(Is / 16) * tap_bytes, 0)),
         ...);
    }(std::make_integer_sequence<int, 144>{});
}

void magic(unsigned d, unsigned& mg, unsigned& sh) {
    if (d <= 1) { mg = 0; sh = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mg = (unsigned)(((1ull << (31 + l)) / d) + 1ull);
    sh = l - 1;
}

struct W2Plan { int bl, xr; W2Params p; };

const bool g_wgrad2_off = cs_env_flag_("CELLSEG_NO_WGRAD2");     // A/B experiments only

bool plan(const CsConvGeom* g, int dtype, int n_items, W2Plan& pl) {
    if (g_wgrad2_off || !g || dtype != CS_BF16) return false;
    if (g->groups > 1 || g->R != 3 || g->S != 3 || g->stride != 1 || g->pad != 1) return false;
    if (g->C % 64 || g->K % 64 || g->P != g->H || g->Q != g->W || n_items < 1 || n_items > 8) return false;
    const int Wp = g->W + 1, Hp = g->H + 1;
    if (Wp == 2 || Hp == 2) return false;           // (the magic division needs a divisor >= 2; 1x1 images are not worth it anyway)
    int bl, xr;
    if (Wp <= 11) { bl = 128; xr = 152; }
    else if (Wp <= 23) { bl = 128; xr = 176; }
    else if (Wp <= 39) { bl = 64; xr = 144; }
    else if (Wp <= 79) { bl = 64; xr = 224; }
    else if (Wp <= 159) { bl = 64; xr = 384; }       // the 150 x 150 (and 128 x 128) decoder layers: 112 KiB of LDS, one workgroup per compute unit
    else return false;
    const unsigned long long D = (unsigned long long)g->N * Hp * Wp;
    const unsigned long long xb = (unsigned long long)g->N * g->H * g->W * g->C * 2ull, gb = (unsigned long long)g->N * g->H * g->W * g->K * 2ull;
    const unsigned long long sb = (unsigned long long)g->K * 9ull * g->C * 4ull;
    if (D + 4096 >= 0x7fffffffull || xb >= 0x80000000ull || gb >= 0x80000000ull || sb >= 0x80000000ull) return false;
    W2Params& p = pl.p;
    p = W2Params{};
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.K = g->K;
    p.Wp = Wp; p.Hp = Hp;
    magic((unsigned)Wp, p.mg_wp, p.sh_wp);
    magic((unsigned)Hp, p.mg_hp, p.sh_hp);
    p.n_stages = (unsigned)((D + bl - 1) / bl);
    p.n_kt = g->K / 64; p.n_ct = g->C / 64;
    p.n_items = n_items;
    // split the positions until ~CELLSEG_WGRAD2_BLOCKS workgroups exist; every split costs one write + read of |dW| in fp32 and
    // a split should hold at least 4 stages (its first loads and its 144 stores per wave are not overlapped with anything)
    // (measured, bench.py: 256 / 384 / 512 / 768 / 1024 workgroups -> 0.643 / 0.572 / 0.596 / 0.575 / 0.578 ms per step for the family)
    static const int target = cs_env_int_("CELLSEG_WGRAD2_BLOCKS", 384);
    const long long tiles = (long long)p.n_kt * p.n_ct * n_items;
    long long want = (target + tiles - 1) / tiles;
    const long long max_split = p.n_stages / 4 > 0 ? p.n_stages / 4 : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    p.stages_per_split = (unsigned)((p.n_stages + want - 1) / want);
    p.nsplit = (int)((p.n_stages + p.stages_per_split - 1) / p.stages_per_split);
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)gb; p.slab_bytes = (unsigned)sb;
    p.total_work = (unsigned)(tiles * p.nsplit);
    p.per_xcd = (p.total_work + 7u) / 8u;
    pl.bl = bl; pl.xr = xr;
    return true;
}

template <int BL, int XR>
int launch_t(const W2Params& p, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(BL + XR) * 128;
    if (!cs_allow_dynamic_lds_(reinterpret_cast<const void*>(wgrad2_kernel<BL, XR>), lds, lds > 81920 ? 163840 : 81920)) return CS_ERR_LAUNCH;
    char name[48];
    snprintf(name, sizeof(name), "wgrad2_kernel<%d,%d>", BL, XR);
    cs_set_variant_(name);
    hipLaunchKernelGGL((wgrad2_kernel<BL, XR>), dim3(p.per_xcd * 8u), dim3(256), lds, st, p);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

}  // namespace

// number of split-K slabs the second-generation kernel writes for this geometry, 0 = not served (ask the first-generation path)
int cs_wgrad2_splits_(const CsConvGeom* g, int dtype, int n_items) {
    W2Plan pl;
    return plan(g, dtype, n_items, pl) ? pl.p.nsplit : 0;
}

int cs_wgrad2_launch_(const CsConvGeom* g, int dtype, const void* const* x_tab, const void* const* dy_tab, float* const* dw_tab, int n_items,
                      void* stream) {
    W2Plan pl;
    if (!plan(g, dtype, n_items, pl)) {
        cs_set_error_("wgrad2: geometry not served");
        return CS_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < 8; ++i) {
        const int j = i < n_items ? i : 0;
        pl.p.x_tab[i] = x_tab[j]; pl.p.dy_tab[i] = dy_tab[j]; pl.p.dw_tab[i] = dw_tab[j];
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (pl.bl == 128 && pl.xr == 152) return launch_t<128, 152>(pl.p, st);
    if (pl.bl == 128) return launch_t<128, 176>(pl.p, st);
    if (pl.xr == 144) return launch_t<64, 144>(pl.p, st);
    if (pl.xr == 384) return launch_t<64, 384>(pl.p, st);
    return launch_t<64, 224>(pl.p, st);
}
