// Second-generation bf16 convolution kernels for gfx950 (CDNA4): forward and data gradient of the stride-1 RxS convolutions
// and of the 1x1 convolutions that carry the ResNet / ResNeXt trunk (model/resnet.py:20,23,51,53,55 and their autograd
// backward).  The first generation (conv_igemm.hip) stays for fp32 parity mode and for every shape this file declines.
//
// What is different from conv_igemm.hip, and why (profiles/round1_notes.md: MFMA pipe 25 % busy, 15.7 VALU+SALU per MFMA,
// one barrier per 64-deep K-step, two LDS passes in the epilogue):
//  * HALO staging.  The workgroup's pixel tile is a run of BM consecutive destination pixels.  For one 64-channel chunk the
//    source rows that ALL R*S taps of that run touch are fetched ONCE into LDS ("padded-linear" coordinates: one shared run of
//    zero slots between image rows and between images, so a tap is a constant row shift and padding needs no test), instead
//    of once per tap: 1.2-1.6x the tile instead of 9x, and ONE barrier per R*S taps.
//  * Weights never touch LDS.  They are staged in MFMA-fragment order ("packed": one wave-instruction = 1 KiB contiguous =
//    the 32 x 16 operand of one MFMA), each wave streams the fragments of ITS 32 output channels straight into registers
//    two tap-steps ahead (no sharing between waves -> no barrier, no ds_write, no ds_read for half of the operands).
//  * Waves split the OUTPUT CHANNELS (1 x 4), every wave multiplies the whole pixel tile: 128 px x 32 ch per wave = 4 MFMA
//    32x32x16 per 16-deep step fed by 4 ds_read_b128 + 1/4 buffer_load.
//  * The accumulator is kept TRANSPOSED (channel on the register index, pixel on the lane): after bf16 packing and one
//    v_permlane32_swap per dword pair every lane owns 16 contiguous bytes of its pixel's row.  The epilogue exchanges that
//    arrangement for a row arrangement (4 consecutive lanes = 64 contiguous bytes of one pixel) through a wave-private LDS
//    scratch -- no workgroup barrier -- because 64 lanes x 64 different rows is address-rate-bound (struct Epilogue).
//  * 1x1 convolutions: a persistent ring kernel (conv2_ring_kernel) instead of one-shot workgroups.
//  * All global->LDS and global->register traffic of the main loop is inline asm with hand-counted s_waitcnt vmcnt(N)
//    (hipcc drains the queue at every use otherwise, see conv_igemm.hip).
#include "cs_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <utility>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // native vector: a 128-bit register operand of inline asm
constexpr unsigned OOB = 0x80000000u;

struct C2Params {
    const void* src;               // NHWC bf16 [NS][SH][SW][SC]
    const void* wpk;               // packed weights (cs_pack_conv_weights)
    void* dst;                     // NHWC bf16 [NS][DH][DW][NOUT]
    const float* shift;            // per destination channel, nullable
    const void* residual;          // destination-shaped bf16 (forward: residual, data gradient: add), nullable
    unsigned char* bits_out;       // nullable: one bit per stored element (> 0)
    const unsigned char* bits_in;  // nullable: one bit per destination element, 0 -> the element is stored as 0
    float* slab;                   // nullable: per-workgroup partial column sums [rows][2][NOUT]
    int act;                       // CS_ACT_NONE / CS_ACT_RELU
    int SH, SW, SC, NS;
    int DH, DW, NOUT;
    int Wp, Hp, PW, PH;            // padded-linear geometry of the source: Wp = SW + PW, Hp = SH + PH
    int LW;                        // LDS rows between two filter rows: Wp (linear pixel runs) or TW + S - 1 (2-D tiles)
    int TH, TWl, tiles_x, tiles_y; // 2-D tiles (wide images): TH x (1 << TWl) destination pixels, tiles per image
    unsigned n_tiles;              // 2-D: NS * tiles_y * tiles_x
    int stride;                    // 1x1 kernel only
    int gather, GH, GWp;           // ring forward of the pixel-paired stem (cs_stem_fwd_packed): LDS row = 32 gathered 16-byte (kh, pair) slots
    unsigned tap_sh[9];            // wide kernel: LDS row shift of every tap of a stage (3x3: r * Wp + s; 1x1 on two planes: 0, BM)
    int lin;                       // wide kernel, 1x1: the tile is a run of consecutive pixels, a stage holds TWO 64-channel planes
    unsigned chunk_bytes;          // wide kernel: channel bytes a stage advances by (128; 256 with two planes)
    int add_stride, AH, AW;        // ring data gradient: the add operand is a COMPACT [NS][AH][AW][NOUT] tensor that only exists at destination
                                   // pixels (add_stride * y, add_stride * x) -- the gradient a strided 1x1 shortcut sends to the block input
    unsigned mg_dw, sh_dw, mg_dh, sh_dh, mg_wp, sh_wp, mg_hp, sh_hp;
    int NCC;                       // 64-channel chunks of the contraction
    int n_ntiles;
    unsigned M;                    // NS*DH*DW
    unsigned src_bytes;
    unsigned pix_bytes;            // SC * 2
    unsigned wpk_bytes;            // whole packed weight tensor (the range check of a raw buffer covers voffset + soffset)
#ifdef CS_DEBUG_V2
    unsigned long long* dbg;       // diagnostic build only: s_memtime stamps [workgroup][wave][6]
#endif
};

// n / d for n < 2^31, d >= 2: q = umulhi(n, mg) >> sh with mg = floor(2^(31+l) / d) + 1, sh = l - 1, l = ceil(log2 d)
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

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }
// the same where compiler-visible LDS reads of DMA-written bytes follow: they must not be hoisted above the wait
template <int N> __device__ __forceinline__ void wait_vm_mem() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// one 16-row LDS block = two 1 KiB LDS-DMA pieces: lanes = 4 chunk columns x 16 rows; the second piece takes the other
// 64 bytes of every row (scalar offset; out-of-range lanes carry 0x80000000 and stay out of range with any soffset < 2 GiB)
__device__ __forceinline__ void dma_block(const i32x4& rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    unsigned tmp;
    // (no "memory" clobber on purpose: the pieces land in the stage nobody reads until the next raw_barrier(), so the compiler
    // may keep moving the current stage's ds_reads across this statement)
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_add_u32 %0, %4, 64\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_add_u32 m0, %1, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %0 offen lds"
        : "=&s"(tmp)
        : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
        : "scc");
}

// the same with an independent offset per piece (gathered operand rows: each 16-byte column of a row comes from another source pixel)
__device__ __forceinline__ void dma_block2(const i32x4& rsrc, unsigned lds_dst, unsigned voff_a, unsigned voff_b) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
        "s_add_u32 m0, %0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen lds"
        ::"s"(lds_dst), "v"(voff_a), "v"(voff_b), "s"(rsrc)
        : "scc");
}

// ---- the main loop owns v[100:255]: the compiler's own allocation is capped at v0..v99 (amdgpu_num_vgpr counts in units of
// two on gfx950) and NEVER sees a value of the loop.  Why: hipcc may copy, split or rotate an asm OUTPUT between an
// asm-issued load and our own s_waitcnt (seen: v_mov_b64 of a weight buffer in flight at the loop header), and an
// `asm volatile` MFMA is a scheduling barrier for the compiler's own ds_reads (seen: every LDS read serialised in front of
// its MFMA).  A register the compiler never sees can be neither copied nor rescheduled: every instruction of the loop is
// its own asm statement with compile-time register numbers, in program order.
//   v[100:103] address temporaries        v[104:111] operand-row addresses (two sets of 4: this tap / next tap)
//   v[112:143] pixel fragments (two sets of 4 x b128)   v[144:207] accumulators (4 tiles x 16)
//   v[208:255] weight fragments (three tap-steps in flight x 4 x b128)
// Audit after every edit: tools/audit_asm.py (no scratch, no compiler instruction naming v100 or above).
constexpr int R_TMP = 100, R_AD = 104, R_AF = 112, R_ACC = 144, R_B = 208;

__device__ __forceinline__ void own_registers() {
    // makes the kernel descriptor allocate all 256 registers (inline-asm TEXT is not scanned for register use)
    asm volatile("; v[100:255] are owned by the main loop" ::: "v100", "v255");
}
template <int REG> __device__ __forceinline__ void vzero() { asm volatile("v_mov_b32 v[%c0], 0" ::"i"(REG)); }
template <int... Rs> __device__ __forceinline__ void vzero_seq(std::integer_sequence<int, Rs...>) { (vzero<R_ACC + Rs>(), ...); }
template <int BASE, int... Rs> __device__ __forceinline__ void vzero_at(std::integer_sequence<int, Rs...>) { (vzero<BASE + Rs>(), ...); }

template <int J, int RB = R_B>
__device__ __forceinline__ void bload4(const i32x4& rsrc, unsigned voff, unsigned soff) {
    // the four 1 KiB weight fragments of one tap-step (64 contraction channels) of this wave's 32 output channels
    asm volatile("buffer_load_dwordx4 v[%c3:%c4], %0, %1, %2 offen\n\t"
                 "buffer_load_dwordx4 v[%c5:%c6], %0, %1, %2 offen offset:1024\n\t"
                 "buffer_load_dwordx4 v[%c7:%c8], %0, %1, %2 offen offset:2048\n\t"
                 "buffer_load_dwordx4 v[%c9:%c10], %0, %1, %2 offen offset:3072"
                 ::"v"(voff), "s"(rsrc), "s"(soff), "i"(RB + 16 * J), "i"(RB + 16 * J + 3), "i"(RB + 16 * J + 4), "i"(RB + 16 * J + 7),
                 "i"(RB + 16 * J + 8), "i"(RB + 16 * J + 11), "i"(RB + 16 * J + 12), "i"(RB + 16 * J + 15));
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
template <int DST, int ADDR, unsigned OFF> __device__ __forceinline__ void lds_rd() {
    asm volatile("ds_read_b128 v[%c0:%c1], v[%c2] offset:%c3" ::"i"(DST), "i"(DST + 3), "i"(ADDR), "i"(OFF));
}
// acc (channel x pixel) += weight fragment x pixel fragment
template <int ACC, int B, int A> __device__ __forceinline__ void mfma() {
    asm volatile("v_mfma_f32_32x32x16_bf16 v[%c0:%c1], v[%c2:%c3], v[%c4:%c5], v[%c0:%c1]" ::"i"(ACC), "i"(ACC + 15), "i"(B), "i"(B + 3), "i"(A),
                 "i"(A + 3));
}
// the LAST MFMA of an accumulator tile writes a compiler value instead of the owned registers: the epilogue of tile i then runs
// while the MFMAs of tiles i+1.. are still executing, with 16 instead of 64 result registers live and no copy-out.  hipcc does
// not know this statement is an MFMA: the result -> VALU wait states are inside the string.
template <int ACC, int B, int A> __device__ __forceinline__ f32x16 mfma_out() {
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, v[%c1:%c2], v[%c3:%c4], v[%c5:%c6]\n\ts_nop 15\n\ts_nop 3"
                 : "=v"(d)
                 : "i"(B), "i"(B + 3), "i"(A), "i"(A + 3), "i"(ACC), "i"(ACC + 15));
    return d;
}
// operand-row addresses of one tap for the 4 pixel tiles -> address set S (see row_addr; hhb = chunk-column bit of the lane half)
template <int S, int RAD = R_AD, int RTMP = R_TMP>
__device__ __forceinline__ void addr4(unsigned q0, unsigned q1, unsigned q2, unsigned q3, unsigned sh, unsigned hhb, unsigned cf0) {
#define CS_ADDR1(Q, A)                                             \
    "v_add_lshl_u32 v[%c[" A "]], %[" Q "], %[sh], 4\n\t"           \
    "v_and_b32 v[%c[t]], 0xffffff00, v[%c[" A "]]\n\t"              \
    "v_and_or_b32 v[%c[" A "]], v[%c[" A "]], %[cf0], %[hhb]\n\t"   \
    "v_lshl_or_b32 v[%c[" A "]], v[%c[t]], 3, v[%c[" A "]]\n\t"
    asm volatile(CS_ADDR1("q0", "a0") CS_ADDR1("q1", "a1") CS_ADDR1("q2", "a2") CS_ADDR1("q3", "a3")
                 ::[q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3), [sh] "s"(sh), [hhb] "v"(hhb), [cf0] "s"(cf0),
                 [a0] "i"(RAD + 4 * S), [a1] "i"(RAD + 4 * S + 1), [a2] "i"(RAD + 4 * S + 2), [a3] "i"(RAD + 4 * S + 3), [t] "i"(RTMP));
#undef CS_ADDR1
}
// one accumulator register into a compiler value (after the loop)
template <int REG> __device__ __forceinline__ float vget() {
    float x;
    asm volatile("v_mov_b32 %0, v[%c1]" : "=v"(x) : "i"(REG));
    return x;
}
template <int I, int... Rs> __device__ __forceinline__ void acc_tile(f32x16& out, std::integer_sequence<int, Rs...>) {
    ((out[Rs] = vget<R_ACC + 16 * I + Rs>()), ...);
}
template <int BASE, int... Rs> __device__ __forceinline__ void dbg_tile(f32x16& out, std::integer_sequence<int, Rs...>) {
    ((out[Rs] = vget<BASE + Rs>()), ...);
}

// One tap-step of a TM x (32 px x 32 ch) wave tile: 4*TM MFMA and 4*TM ds_read_b128, software-pipelined over the two fragment
// sets (set 0: 16-deep steps 0 and 2, set 1: steps 1 and 3).  COLD = first tap after a barrier (own addresses + first reads);
// otherwise the tap is entered with its first 2*TM reads in flight.  AHEAD = leave with the next tap's first reads in flight
// (addresses from sh_next unless SAME, stage offset NSTG); the last tap before a barrier has no read-ahead (the next stage is
// only visible after the barrier).  Every MFMA waits for exactly its own fragment: 2*TM - 1 younger reads in steady state.
template <int TM, int I, int... Is> struct TileLoop {
    template <typename F> static __device__ __forceinline__ void run(F&& f) {
        f.template operator()<I>();
        if constexpr (sizeof...(Is) > 0) TileLoop<TM, Is...>::run(f);
    }
};
template <int TM, typename F> __device__ __forceinline__ void for_tiles(F&& f) {
    if constexpr (TM == 2) TileLoop<2, 0, 1>::run(f);
    else if constexpr (TM == 3) TileLoop<3, 0, 1, 2>::run(f);
    else TileLoop<4, 0, 1, 2, 3>::run(f);
}

struct NoTile { template <int I> __device__ __forceinline__ void operator()(const f32x16&) const {} };

template <int TM, int CUR, int S, unsigned STG, unsigned NSTG, bool COLD, bool AHEAD, bool SAME, bool FINAL = false, typename OnTile = NoTile>
__device__ __forceinline__ void tap_mfma(unsigned q0, unsigned q1, unsigned q2, unsigned q3, unsigned sh_cur, unsigned sh_next, unsigned hhb,
                                         unsigned cf0, OnTile&& on_tile = NoTile{}) {
    constexpr int A = R_AD + 4 * S, N = SAME ? A : R_AD + 4 * (1 - S);
    constexpr int F0 = R_AF, F1 = R_AF + 16;
    constexpr int B = R_B + 16 * CUR;
    constexpr int W = 2 * TM - 1;
    static_assert(!(FINAL && AHEAD), "the final tap of a workgroup reads nothing ahead");
    if constexpr (COLD) {
        addr4<S>(q0, q1, q2, q3, sh_cur, hhb, cf0);
        for_tiles<TM>([&]<int i>() { lds_rd<F0 + 4 * i, A + i, STG>(); });
        for_tiles<TM>([&]<int i>() { lds_rd<F1 + 4 * i, A + i, STG + 512>(); });
    }
    if constexpr (AHEAD && !SAME) addr4<1 - S>(q0, q1, q2, q3, sh_next, hhb, cf0);
    for_tiles<TM>([&]<int i>() { wait_lgkm<W>(); mfma<R_ACC + 16 * i, B + 0, F0 + 4 * i>(); lds_rd<F0 + 4 * i, A + i, STG + 1024>(); });
    for_tiles<TM>([&]<int i>() { wait_lgkm<W>(); mfma<R_ACC + 16 * i, B + 4, F1 + 4 * i>(); lds_rd<F1 + 4 * i, A + i, STG + 1536>(); });
    if constexpr (AHEAD) {
        for_tiles<TM>([&]<int i>() { wait_lgkm<W>(); mfma<R_ACC + 16 * i, B + 8, F0 + 4 * i>(); lds_rd<F0 + 4 * i, N + i, NSTG>(); });
        for_tiles<TM>([&]<int i>() { wait_lgkm<W>(); mfma<R_ACC + 16 * i, B + 12, F1 + 4 * i>(); lds_rd<F1 + 4 * i, N + i, NSTG + 512>(); });
    } else {
        for_tiles<TM>([&]<int i>() { wait_lgkm<W - i>(); mfma<R_ACC + 16 * i, B + 8, F0 + 4 * i>(); });
        if constexpr (FINAL) {
            for_tiles<TM>([&]<int i>() {
                wait_lgkm<TM - 1 - i>();
                const f32x16 d = mfma_out<R_ACC + 16 * i, B + 12, F1 + 4 * i>();
                on_tile.template operator()<i>(d);
            });
        } else {
            for_tiles<TM>([&]<int i>() { wait_lgkm<TM - 1 - i>(); mfma<R_ACC + 16 * i, B + 12, F1 + 4 * i>(); });
        }
    }
}

// LDS image of one stage: 16-row blocks of 2 KiB, inside a block chunk-major: (row q, 16-byte chunk c) at
// (q >> 4) * 2048 + c * 256 + (q & 15) * 16.  16 lanes reading 16 consecutive rows of one chunk cover one 256-byte bank row:
// conflict-free without a row-dependent swizzle, and one DMA piece (4 chunks x 16 rows) reads 16 x 64 contiguous bytes.
__device__ __forceinline__ unsigned row_addr(unsigned q) {
    const unsigned t4 = q << 4;
    return ((t4 & 0xffffff00u) << 3) | (t4 & 0xf0u);
}

__device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
}
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
    // lanes 32-63 of a <-> lanes 0-31 of b
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// ---- epilogue of one wave: TM pixel tiles (32 pixels each, pixel on the lane) x 32 channels (on the register index) ----
// accumulator tile i, register r: pixel m0w + 32*i + (lane & 31), channel n_w + 8*(r >> 2) + 4*(lane >> 5) + (r & 3)
// * global memory is touched in ROW arrangement only: lane L handles, for q = 0,1, the 16-byte piece (L & 3) of pixel
//   16q + (L >> 2) of the tile -- four consecutive lanes cover the wave's 64 contiguous bytes of one pixel.  Stamped, the
//   accumulator arrangement (one lane = one pixel, 64 lanes = 64 rows 2*NOUT bytes apart) cost 160-250 cycles PER INSTRUCTION
//   for its 16-byte-per-row requests and made the 1x1 layers address-rate-bound at ~3.8 TB/s.  The exchange between the two
//   arrangements goes through a wave-private 32 x 80-byte LDS scratch (80: the 16-byte writes of 8 consecutive pixel lanes fall
//   into distinct banks), 2 ds_write_b128 + 2 ds_read_b128 per direction, no workgroup barrier;
// * its global operands are fetched BEFORE the main loop (prefetch()): stamped, the epilogue spent ~2/3 of its 5-7 k cycles
//   waiting for the shift vector and for one residual / mask load per tile, each a full memory latency with nothing in flight;
// * every global access is a raw-buffer access with a 32-bit offset: rows past the end carry an out-of-range offset and are
//   dropped by the range check (no exec masking, no 64-bit address arithmetic);
// * tile i is finished while the last MFMAs of tiles i+1.. execute (tap_mfma FINAL).
// PRE_RES: fetch the residual / add operand before the main loop too (32 registers live across the loop: only where the loop
// state is small -- the multi-chunk halo kernels would spill, and their layers have no such operand in practice).
// DG: data-gradient flavour (add operand, mask bits in, column sums out) instead of the forward one (shift, residual, ReLU,
// sign bits out): two instantiations instead of runtime branches keep the epilogue's register footprint under the cap.
constexpr unsigned EPI_ROW = 80u, EPI_WAVE = 32u * EPI_ROW, EPI_LDS = 4u * EPI_WAVE;
using lds_u32x4 = __attribute__((address_space(3))) u32x4;
__device__ __forceinline__ void lds_put(unsigned a, const uint4& v) { *(lds_u32x4*)(unsigned long)(a) = __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ uint4 lds_get(unsigned a) { return __builtin_bit_cast(uint4, *(lds_u32x4*)(unsigned long)(a)); }
template <int CTRL> __device__ __forceinline__ unsigned quad(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// bit k of the result = (bf16 element k of the 16 bytes is > 0); both halves of a dword are tested at once: bit 15 of
// ((h & 0x7fff) + 0x7fff) says "magnitude non-zero", and-not h clears it for negative values
__device__ __forceinline__ unsigned pos_bits8(const uint4& o) {
    const unsigned wds[4] = {o.x, o.y, o.z, o.w};
    unsigned t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a = ((wds[i] & 0x7fff7fffu) + 0x7fff7fffu) & ~wds[i];
        t |= ((a >> 15) & 0x10001u) << (2 * i);
    }
    return (t | (t >> 15)) & 0xffu;
}
// the same for values that went through ReLU (no negative element): > 0 <=> the bf16 pattern is >= 1, one packed-u16 minimum per dword
// (asm: hipcc turns the builtin minimum into a compare + select per half)
__device__ __forceinline__ unsigned nz_bits8(const uint4& o) {
    const unsigned wds[4] = {o.x, o.y, o.z, o.w};
    const unsigned ones = 0x00010001u;
    unsigned u = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned t;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(wds[i]), "s"(ones));
        u |= t << (2 * i);
    }
    return (u | (u >> 15)) & 0xffu;
}
// Row-arrangement offsets of one pixel tile, computed ONCE per tile: the byte offset of (32-pixel tile i, half q) is rb + (2i+q) * 16
// rows, valid while its row index stays below M -- 3 instructions per access instead of a multiply-add chain per (i, q).
struct TileOffs {
    static constexpr bool kPreset = false;
    static constexpr unsigned NONE = 0x7fffff00u;      // M < 2^31 - 512: NONE + 16 * 7 neither wraps nor passes the test
    unsigned rb, mr, bb, br;
    // `piece`: which 16 bytes of the wave's 64 this lane handles; bits: lanes with (lane & 3) == q hold the dword of pixel q.
    // Bit planes are CHANNEL-BLOCK-MAJOR (round 5): dword (n_w / 32) * M + m holds the 32 channels [n_w, n_w + 32) of pixel m, so the 32
    // dwords a wave stores (loads) for a 32-pixel tile are 128 contiguous bytes.  In the pixel-major layout of rounds 1-4 they were
    // 32 dwords NOUT / 8 bytes apart: 32 sectors per instruction, 3-4 % of every forward launch that writes a plane (round 4 notes).
    __device__ __forceinline__ void set(unsigned base, bool ok, unsigned nout, unsigned n_w, unsigned piece, unsigned M) {
        const unsigned lane = threadIdx.x & 63;
        const bool live = ok && base != 0xffffffffu;
        const unsigned m = base + (lane >> 2);
        mr = live ? m : NONE;
        rb = (m * nout + n_w + 8u * piece) * 2u;
        const unsigned mbit = m + 16u * (lane & 1u);
        br = (live && !(lane & 2u)) ? mbit : NONE;
        bb = ((n_w >> 5) * M + mbit) * 4u;
    }
    __device__ __forceinline__ unsigned row(unsigned M, unsigned nout, int i, int q) const {
        const unsigned d = (unsigned)(2 * i + q);
        return mr + 16u * d < M ? rb + d * (32u * nout) : OOB;
    }
    __device__ __forceinline__ unsigned bit(unsigned M, unsigned nout, int i) const {
        return br + 32u * (unsigned)i < M ? bb + (unsigned)i * 128u : OOB;
    }
};
// The same offsets for a 2-D pixel tile (wide images, round 3): local pixel j of the tile is (y0 + (j >> TWl), x0 + (j & (TW - 1))); every
// (32-pixel tile, half) has its own offset, computed once before the main loop (3 * TM registers instead of 4).
template <int TM> struct TileOffs2D {
    static constexpr bool kPreset = true;
    unsigned ro[TM][2], bo[TM];
    __device__ __forceinline__ void set2d(unsigned n, unsigned y0, unsigned x0, unsigned dh, unsigned dw, unsigned twl, unsigned j0, bool alive,
                                          unsigned nout, unsigned n_w, unsigned piece, unsigned M) {
        const unsigned lane = threadIdx.x & 63, tw1 = (1u << twl) - 1u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned j = j0 + 32u * i + 16u * q + (lane >> 2);
                const unsigned y = y0 + (j >> twl), x = x0 + (j & tw1);
                const unsigned m = (n * dh + y) * dw + x;
                ro[i][q] = (alive && y < dh && x < dw) ? (m * nout + n_w + 8u * piece) * 2u : OOB;
            }
            const unsigned jb = j0 + 32u * i + (lane >> 2) + 16u * (lane & 1u);
            const unsigned yb = y0 + (jb >> twl), xb = x0 + (jb & tw1);
            const unsigned mb = (n * dh + yb) * dw + xb;
            bo[i] = (alive && !(lane & 2u) && yb < dh && xb < dw) ? ((n_w >> 5) * M + mb) * 4u : OOB;       // channel-block-major plane
        }
    }
    __device__ __forceinline__ unsigned row(unsigned, unsigned, int i, int q) const { return ro[i][q]; }
    __device__ __forceinline__ unsigned bit(unsigned, unsigned, int i) const { return bo[i]; }
};
// the 16 bytes with element k kept where bit k of b is set
__device__ __forceinline__ uint4 keep_bits8(const uint4& o, unsigned b) {
    unsigned wds[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned m = (b >> (2 * i));
        wds[i] &= (m & 1u) * 0xffffu + (m & 2u) * 0x7fff8000u;
    }
    return make_uint4(wds[0], wds[1], wds[2], wds[3]);
}

template <int TM, bool PRE_RES, bool DG, typename TO = TileOffs> struct Epilogue {
    const C2Params& p;
    unsigned m0w, slab_row;
    int n_w;
    bool alive;
    unsigned scr;              // LDS byte address of this wave's exchange scratch (EPI_WAVE bytes)
    float sh[16];
    uint4 rr[TM][2];           // residual / add in row arrangement
    unsigned mb[TM];           // mask words: lanes with piece 0 / 1 hold the dword of pixel q = 0 / 1
    float s1[8];
#ifdef CS_DEBUG_V2
    unsigned long long e_acc[4] = {0, 0, 0, 0}, e_t = 0;
#define CS_ETICK(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); e_acc[k] += t_ - e_t; e_t = t_; } while (0)
#define CS_ESTART() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e_t))
#else
#define CS_ETICK(k)
#define CS_ESTART()
#endif
    __amdgpu_buffer_rsrc_t r_dst, r_res, r_bin, r_bout;

    __device__ __forceinline__ Epilogue(const C2Params& p_, unsigned m0w_, int n_w_, unsigned slab_row_, bool alive_, unsigned scr_)
        : p(p_), m0w(m0w_), slab_row(slab_row_), n_w(n_w_), alive(alive_), scr(scr_) {}

    TO to;                     // row arrangement: lane -> (pixel 16q + (lane >> 2) of tile i, piece lane & 3)
    __device__ __forceinline__ unsigned row_lds(int q) const {       // row arrangement: the lane's slot in the scratch
        const int lane = threadIdx.x & 63;
        return scr + (unsigned)(16 * q + (lane >> 2)) * EPI_ROW + (unsigned)(lane & 3) * 16u;
    }
    __device__ __forceinline__ unsigned acc_lds(int j) const {       // accumulator arrangement: piece 2j + hh of pixel l31
        const int lane = threadIdx.x & 63;
        return scr + (unsigned)(lane & 31) * EPI_ROW + (unsigned)(2 * j + (lane >> 5)) * 16u;
    }

    // byte offset of the residual / add operand's row for (32-pixel tile i, half q): the destination row itself, or -- compact strided
    // operand of a data gradient (add_stride = 2, see RingEpilogue::add_off; the wide kernel's 1x1 form serves such layers too) -- the
    // row of pixel (y / 2, x / 2) when both coordinates are even, nothing otherwise
    __device__ __forceinline__ unsigned res_off(int i, int q) const {
        if constexpr (TO::kPreset) return to.row(p.M, (unsigned)p.NOUT, i, q);
        else {
            if (p.add_stride <= 1) return to.row(p.M, (unsigned)p.NOUT, i, q);
            const unsigned lane = threadIdx.x & 63;
            const unsigned m = to.mr + 16u * (unsigned)(2 * i + q);
            const unsigned yall = udivm(m, p.mg_dw, p.sh_dw);
            const unsigned x = m - yall * (unsigned)p.DW;
            const unsigned n = udivm(yall, p.mg_dh, p.sh_dh);
            const unsigned y = yall - n * (unsigned)p.DH;
            const unsigned mc = (n * (unsigned)p.AH + (y >> 1)) * (unsigned)p.AW + (x >> 1);
            return (m < p.M && !((x | y) & 1u)) ? (mc * (unsigned)p.NOUT + (unsigned)n_w + 8u * (lane & 3u)) * 2u : OOB;
        }
    }

    __device__ __forceinline__ void prefetch() {
        const int lane = threadIdx.x & 63;
        const int hh = lane >> 5;
        const unsigned out_bytes = p.M * (unsigned)p.NOUT * 2u;
        if constexpr (!TO::kPreset) to.set(m0w, alive, (unsigned)p.NOUT, (unsigned)n_w, (unsigned)(lane & 3), p.M);
        r_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, out_bytes, 0x00020000);
        r_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0,
                                                  !p.residual ? 0u : p.add_stride == 2 ? (unsigned)(p.NS * p.AH * p.AW * p.NOUT) * 2u : out_bytes, 0x00020000);
        r_bin = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.bits_in), 0, p.bits_in ? out_bytes >> 4 : 0u, 0x00020000);
        r_bout = __builtin_amdgcn_make_buffer_rsrc(p.bits_out, 0, p.bits_out ? out_bytes >> 4 : 0u, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!DG && p.shift && alive) s = *reinterpret_cast<const float4*>(p.shift + n_w + 8 * g + 4 * hh);
            sh[4 * g] = s.x; sh[4 * g + 1] = s.y; sh[4 * g + 2] = s.z; sh[4 * g + 3] = s.w;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // a NULL operand has a zero-sized buffer: the loads return zeros without touching memory
            if constexpr (PRE_RES) {
                rr[i][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, res_off(i, 0), 0, 0));
                rr[i][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, res_off(i, 1), 0, 0));
            }
            if constexpr (DG) mb[i] = __builtin_amdgcn_raw_buffer_load_b32(r_bin, to.bit(p.M, (unsigned)p.NOUT, i), 0, 0);
        }
    }

    template <int I> __device__ __forceinline__ void operator()(const f32x16& acc) {
        const int lane = threadIdx.x & 63;
        const unsigned piece = (unsigned)(lane & 3);
        CS_ESTART();
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = DG ? acc[r] : acc[r] + sh[r];
        if (p.residual) {
            uint4 x0, x1;
            if constexpr (PRE_RES) { x0 = rr[I][0]; x1 = rr[I][1]; }
            else {
                x0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, res_off(I, 0), 0, 0));
                x1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, res_off(I, 1), 0, 0));
            }
            lds_put(row_lds(0), x0);
            lds_put(row_lds(1), x1);
            __builtin_amdgcn_wave_barrier();
            uint4 xa = lds_get(acc_lds(0)), xb = lds_get(acc_lds(1));
            __builtin_amdgcn_wave_barrier();
            // stored arrangement -> accumulator arrangement (the swap is an involution)
            swap32(xa.x, xa.z); swap32(xa.y, xa.w);
            swap32(xb.x, xb.z); swap32(xb.y, xb.w);
            const unsigned rw[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float lo, hi;
                unpack2(rw[k], lo, hi);
                v[2 * k] += lo;
                v[2 * k + 1] += hi;
            }
        }
        unsigned pk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        if (!DG && p.act == CS_ACT_RELU) {          // on the packed pairs: 8 instructions instead of 32 (cs_common.h relu_bf16x2)
#pragma unroll
            for (int k = 0; k < 8; ++k) pk[k] = relu_bf16x2(pk[k]);
        }
        // groups (0,1) and (2,3): after the swaps lanes 0-31 hold channels 16j .. 16j+7, lanes 32-63 channels 16j+8 .. 16j+15
        swap32(pk[0], pk[2]); swap32(pk[1], pk[3]);
        swap32(pk[4], pk[6]); swap32(pk[5], pk[7]);
        lds_put(acc_lds(0), make_uint4(pk[0], pk[1], pk[2], pk[3]));
        lds_put(acc_lds(1), make_uint4(pk[4], pk[5], pk[6], pk[7]));
        __builtin_amdgcn_wave_barrier();
        uint4 o0 = lds_get(row_lds(0)), o1 = lds_get(row_lds(1));
        __builtin_amdgcn_wave_barrier();
        CS_ETICK(0);
        if constexpr (DG) {
            if (p.bits_in) {
                const unsigned w0 = quad<0x00>(mb[I]), w1 = quad<0x55>(mb[I]);       // broadcast lane 0 / lane 1 of the quad
                o0 = keep_bits8(o0, w0 >> (8u * piece));
                o1 = keep_bits8(o1, w1 >> (8u * piece));
            }
            if constexpr (I == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s1[k] = 0.f;
            }
            if (p.slab) {                      // statistics are those of the STORED values; rows that are not stored count as zeros
                if (to.row(p.M, (unsigned)p.NOUT, I, 0) == OOB) o0 = make_uint4(0u, 0u, 0u, 0u);
                if (to.row(p.M, (unsigned)p.NOUT, I, 1) == OOB) o1 = make_uint4(0u, 0u, 0u, 0u);
                const unsigned ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float lo, hi;
                    unpack2(ow[k], lo, hi);
                    s1[2 * (k & 3)] += lo; s1[2 * (k & 3) + 1] += hi;
                }
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), r_dst, to.row(p.M, (unsigned)p.NOUT, I, 0), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), r_dst, to.row(p.M, (unsigned)p.NOUT, I, 1), 0, 0);
        CS_ETICK(1);
        if (!DG && p.bits_out) {
            // byte j of a pixel's dword = channels 8j .. 8j+7 = piece j; OR over the quad, then lane (piece q) stores pixel q's dword
            const bool relu = p.act == CS_ACT_RELU;
            unsigned w0 = (relu ? nz_bits8(o0) : pos_bits8(o0)) << (8u * piece), w1 = (relu ? nz_bits8(o1) : pos_bits8(o1)) << (8u * piece);
            w0 |= quad<0xb1>(w0); w1 |= quad<0xb1>(w1);       // [1,0,3,2]
            w0 |= quad<0x4e>(w0); w1 |= quad<0x4e>(w1);       // [2,3,0,1]
            __builtin_amdgcn_raw_buffer_store_b32((lane & 1) ? w1 : w0, r_bout, to.bit(p.M, (unsigned)p.NOUT, I), 0, 0);
        }
        CS_ETICK(2);
        CS_ETICK(3);
    }

    __device__ __forceinline__ void finish() {
        if (!DG || !p.slab || !alive) return;
        const int lane = threadIdx.x & 63;
        // fold the 16 lanes that share a piece; lanes 0-3 keep the 8 channels of piece 0-3
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int off = 32; off >= 4; off >>= 1) s1[k] += __shfl_xor(s1[k], off, 64);
        }
        if (lane < 4) {
            float* dst = p.slab + (size_t)slab_row * 2 * p.NOUT + n_w + 8 * lane;
            *reinterpret_cast<float4*>(dst) = make_float4(s1[0], s1[1], s1[2], s1[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
        }
    }
};

// =================================================================================================
// Halo kernel: R x S taps (NTAP = R*S, a multiple of 3; S = SK), stride 1.  Workgroup = 4 waves as WM x WN;
// wave tile = (32*TM pixels) x 32 channels; NBW = 16-row LDS blocks each wave fetches per 64-channel chunk.
// =================================================================================================
// T2D (round 3): the pixel tile is TH x TW destination pixels of ONE image instead of a run of consecutive pixels.  A run's window is
// BM + 2 * Wp + 2 rows (+ Wp more where it crosses into the next image): 2.2x the tile at 75 x 75, 5.8x at 150 x 150 -- the decoder's two
// widest layers (resnet.py:162-163) did not fit two stages.  An 8 x 16 tile's window is 10 x 18 = 180 rows for 128 pixels at any image
// width; LDS row of window position (ly, lx) = ly * LW + lx, LW = TW + S - 1, so a tap is still a constant row shift.
template <int NTAP, int SK, int TM, int WM, int WN, int NBW, bool DG, bool T2D = false>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(50))) void conv2_halo_kernel(C2Params p) {
    static_assert(WM * WN == 4 && NTAP % 3 == 0 && NBW <= 8, "layout");
    constexpr int BM = WM * TM * 32, BN = WN * 32;
    constexpr unsigned STAGE = NBW * 4 * 2048;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
#ifdef CS_DEBUG_V2
    unsigned long long t_stamp[6];
#define CS_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_stamp[i]))
    CS_STAMP(0);
#else
#define CS_STAMP(i)
#endif

    // XCD-aware tile order: id = 8*slot + xcd, all N tiles of one M tile back to back on one XCD (conv_igemm.hip)
    const unsigned bid = blockIdx.x;
    const unsigned slot = bid >> 3;
    const unsigned mtile = (slot / (unsigned)p.n_ntiles) * 8u + (bid & 7u);
    const unsigned m0 = mtile * BM;
    if (T2D ? mtile >= p.n_tiles : m0 >= p.M) return;
    const int n0 = (int)(slot % (unsigned)p.n_ntiles) * BN;
    const int n_w = n0 + wn * 32;
    const bool alive = n_w < p.NOUT;

    using TOffs = std::conditional_t<T2D, TileOffs2D<TM>, TileOffs>;
    Epilogue<TM, (NBW == 8), DG, TOffs> epi(p, m0 + (unsigned)(wm * TM * 32), n_w, mtile * WM + wm, alive, 0u);
    unsigned qb[TM];
    unsigned voff[NBW];
    if constexpr (T2D) {
        const unsigned per_img = (unsigned)(p.tiles_y * p.tiles_x);
        const unsigned n = mtile / per_img, tr = mtile - n * per_img;
        const unsigned ty = tr / (unsigned)p.tiles_x, tx = tr - ty * (unsigned)p.tiles_x;
        const unsigned y0 = ty * (unsigned)p.TH, x0 = tx << p.TWl;
        const unsigned j0 = (unsigned)(wm * TM * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned j = j0 + 32u * i + (unsigned)l31;
            qb[i] = (j >> p.TWl) * (unsigned)p.LW + (j & ((1u << p.TWl) - 1u));
        }
        // this wave's LDS blocks: b = wave + 4j; lane -> window position 16b + (lane & 15) = (ly, lx), source pixel (y0 - PH + ly, x0 - PW + lx)
        const unsigned win_rows = (unsigned)(p.TH + (NTAP / SK) - 1) * (unsigned)p.LW;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const unsigned l = 16u * (unsigned)(wave + 4 * j) + (unsigned)(lane & 15);
            const unsigned ly = udivm(l, p.mg_wp, p.sh_wp);                  // / LW
            const unsigned lx = l - ly * (unsigned)p.LW;
            const int sy = (int)(y0 + ly) - p.PH, sx = (int)(x0 + lx) - p.PW;
            const bool real = l < win_rows && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;
            const unsigned pix = (n * (unsigned)p.SH + (unsigned)sy) * (unsigned)p.SW + (unsigned)sx;
            voff[j] = real ? pix * p.pix_bytes + (unsigned)(lane >> 4) * 16u : OOB;
        }
        epi.to.set2d(n, y0, x0, (unsigned)p.DH, (unsigned)p.DW, (unsigned)p.TWl, j0, alive, (unsigned)p.NOUT, (unsigned)n_w, (unsigned)(lane & 3), p.M);
    } else {
        auto pos0 = [&](unsigned m) -> unsigned {      // padded-linear position of tap offset (0,0) of destination pixel m
            const unsigned yall = udivm(m, p.mg_dw, p.sh_dw);
            const unsigned x = m - yall * (unsigned)p.DW;
            const unsigned n = udivm(yall, p.mg_dh, p.sh_dh);
            const unsigned y = yall - n * (unsigned)p.DH;
            return (n * (unsigned)p.Hp + y) * (unsigned)p.Wp + x;
        };
        const unsigned Lmin = pos0(m0);
        const unsigned m0w = m0 + (unsigned)(wm * TM * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            unsigned m = m0w + 32u * i + (unsigned)l31;
            if (m > p.M - 1) m = p.M - 1;
            qb[i] = pos0(m) - Lmin;
        }
        // this wave's LDS blocks: b = wave + 4j; lane -> (row 16b + (lane & 15), chunk column lane >> 4 (+4 for the second piece))
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const unsigned L = Lmin + 16u * (unsigned)(wave + 4 * j) + (unsigned)(lane & 15);
            const unsigned row = udivm(L, p.mg_wp, p.sh_wp);
            const unsigned col = L - row * (unsigned)p.Wp;
            const unsigned n = udivm(row, p.mg_hp, p.sh_hp);
            const unsigned ry = row - n * (unsigned)p.Hp;
            const bool real = col >= (unsigned)p.PW && ry >= (unsigned)p.PH && n < (unsigned)p.NS;
            const unsigned pix = (n * (unsigned)p.SH + (ry - (unsigned)p.PH)) * (unsigned)p.SW + (col - (unsigned)p.PW);
            voff[j] = real ? pix * p.pix_bytes + (unsigned)(lane >> 4) * 16u : OOB;
        }
    }
    const i32x4 rsrc_a = make_rsrc(p.src, p.src_bytes);
    const i32x4 rsrc_b = make_rsrc(p.wpk, p.wpk_bytes);
    const unsigned bvoff = alive ? (unsigned)lane * 16u : OOB;
    unsigned wsoff = (unsigned)(n_w >> 5) * (unsigned)p.NCC * (unsigned)(NTAP * 4096);
    const unsigned smem_base = lds_off(smem);

    static_assert(TM >= 2 && TM <= 4, "the register map of the main loop holds up to 4 pixel tiles per wave");
    epi.prefetch();
    CS_STAMP(4);
    own_registers();
    vzero_seq(std::make_integer_sequence<int, 16 * TM>{});
    const unsigned hhb = (unsigned)hh * 256u;
    const unsigned cf0 = 0xf0u;

    // ---- prologue: chunk 0 of A, weights of steps 0 and 1
#pragma unroll
    for (int j = 0; j < NBW; ++j) dma_block(rsrc_a, smem_base + (unsigned)(wave + 4 * j) * 2048u, voff[j], 0u);
    bload4<0>(rsrc_b, bvoff, wsoff); wsoff += 4096u;
    bload4<1>(rsrc_b, bvoff, wsoff); wsoff += 4096u;
    wait_vm<8>();
    raw_barrier();
    CS_STAMP(1);

    unsigned nxt_soff = 128u;                    // channel offset (bytes) of the chunk being fetched

    // one tap-step: ODD = the chunk being multiplied sits in stage 1 (and the next one goes to stage 0); HN = a next chunk exists
    auto tap = [&]<int T, bool ODD, bool HN>() {
        constexpr int CUR = T % 3, NX2 = (T + 2) % 3;
        constexpr bool has1 = HN || (T + 1 < NTAP);
        constexpr bool has2 = HN || (T + 2 < NTAP);
        // (the last tap of a chunk leaves none of the next chunk's pieces in flight: the barrier that follows hands the stage over.
        // With NBW <= 5 the last piece is issued at tap 4 and the terms vanish by themselves; NBW = 7 issues its last piece at tap 6.)
        constexpr bool LAST = T == NTAP - 1;
        constexpr int D0 = (HN && !LAST && T < NBW) ? 2 : 0;
        constexpr int D1 = (HN && !LAST && T >= 1 && T - 1 < NBW) ? 2 : 0;
        constexpr int D2 = (HN && !LAST && T >= 2 && T - 2 < NBW) ? 2 : 0;
        constexpr unsigned CUR_STAGE = ODD ? STAGE : 0u, NXT_STAGE = ODD ? 0u : STAGE;
        if constexpr (has2) {
            bload4<NX2>(rsrc_b, bvoff, wsoff);
            wsoff += 4096u;
        }
        if constexpr (D0) dma_block(rsrc_a, smem_base + NXT_STAGE + (unsigned)(wave + 4 * T) * 2048u, voff[T < NBW ? T : 0], nxt_soff);
        wait_vm<(has1 ? 4 : 0) + (has2 ? 4 : 0) + D0 + D1 + D2>();
        const unsigned sh_cur = (unsigned)(T / SK) * (unsigned)p.LW + (unsigned)(T % SK);
        const unsigned sh_next = (unsigned)((T + 1) / SK) * (unsigned)p.LW + (unsigned)((T + 1) % SK);
        // the epilogue's exchange scratch: the stage nobody reads or fills during the last chunk (single-stage launches: the
        // bytes behind the stage)
        if constexpr (T == NTAP - 1 && !HN) epi.scr = smem_base + NXT_STAGE + (unsigned)wave * EPI_WAVE;
        tap_mfma<TM, CUR, T & 1, CUR_STAGE, CUR_STAGE, T == 0, T != NTAP - 1, false, (T == NTAP - 1 && !HN)>(qb[0], qb[1], qb[2 % TM], qb[3 % TM], sh_cur, sh_next,
                                                                                                          hhb, cf0, epi);
    };
    auto chunk = [&]<bool ODD, bool HN>() {
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) { (tap.template operator()<Ts, ODD, HN>(), ...); }(std::make_integer_sequence<int, NTAP>{});
    };

    if constexpr (NBW == 8) {
        chunk.template operator()<false, false>();         // single-chunk launches only (one 64 KiB stage)
    } else {
        for (int cc = 0; cc < p.NCC; cc += 2) {
            if (cc) { raw_barrier(); nxt_soff += 128u; }      // every wave's pieces of this chunk landed (its own tap waits), the previous one is read out
            if (cc + 1 < p.NCC) chunk.template operator()<false, true>();
            else chunk.template operator()<false, false>();
            if (cc + 1 < p.NCC) {
                raw_barrier();
                nxt_soff += 128u;
                if (cc + 2 < p.NCC) chunk.template operator()<true, true>();
                else chunk.template operator()<true, false>();
            }
        }
    }
    CS_STAMP(2);
    epi.finish();
#ifdef CS_DEBUG_V2
    CS_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)");
    CS_STAMP(5);
    if (p.dbg && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 6;
        o[0] = t_stamp[0]; o[1] = t_stamp[4]; o[2] = t_stamp[1]; o[3] = t_stamp[2]; o[4] = t_stamp[3]; o[5] = t_stamp[5];
    }
#endif
}


// =================================================================================================
// Wide kernel (round 4): the halo structure at ONE wave per SIMD with all 512 registers of a lane.
//   * workgroup tile = (32 * TM) pixels x 128 channels, TM = 4 / 6 / 8; the four waves still split the channels (1 x 4) and stream
//     the packed weights of their own 32 channels from L2, so a wave multiplies 32 * TM pixels per 1 KiB weight fragment instead
//     of 96-128: per 16-deep step TM MFMA are fed by TM ds_read_b128 + ONE buffer_load_dwordx4.
//   * accumulators live in AGPRs a[128 : 128 + 16 * TM): 128 registers at TM = 8, which no 256-register wave can hold beside its
//     operands.  (hipcc uses LOW AGPRs as spill space for its own VGPRs when it runs short; the audit checks that it never names
//     a[128] or above.)
//   * the weight stream is a ring of 18 fragment slots at 16-deep granularity (72 VGPRs): a slot is refilled the moment its MFMAs
//     are issued, so 17 steps (4.25 taps; the halo kernel: 2 taps) of weights are in flight -- at one wave per SIMD nobody else
//     covers an L2 round trip.  36 steps per chunk = 2 x 18: slot numbers are compile-time constants in every chunk.
//   * ONE chunk body (9 taps, 36 steps) looped at run time: the LDS stage of a chunk is part of the operand-row addresses, loads
//     past the end are issued out of range (they count in vmcnt, move nothing), so every s_waitcnt immediate is the same in
//     every chunk -- a quarter of the halo kernel's code per pixel tile.
// Register map: v[100:103] temporaries, v[104:119] operand-row addresses (two sets of 8), v[120:183] pixel fragments (two sets
// of 8 x b128), v[184:255] weight ring; the compiler is capped at v0..v99 as in the other kernels of this file.
// =================================================================================================
constexpr int W_TMP = 100, W_AD = 104, W_AF = 120, W_B = 184, W_ACC = 128, W_NSLOT = 18;

template <int N, typename F> __device__ __forceinline__ void for_n(F&& f) {
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (f.template operator()<Is>(), ...); }(std::make_integer_sequence<int, N>{});
}
template <int REG> __device__ __forceinline__ void azero() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(REG)); }
template <int REG> __device__ __forceinline__ float aget() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(REG));
    return x;
}
template <int I> __device__ __forceinline__ f32x16 acc_read_w() {
    f32x16 d;
    [&]<int... Rs>(std::integer_sequence<int, Rs...>) { ((d[Rs] = aget<W_ACC + 16 * I + Rs>()), ...); }(std::make_integer_sequence<int, 16>{});
    return d;
}
template <int ACC, int B, int A> __device__ __forceinline__ void mfma_w() {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" ::"i"(ACC), "i"(ACC + 15), "i"(B), "i"(B + 3), "i"(A),
                 "i"(A + 3));
}
// the 1 KiB weight fragment of one 16-deep step of this wave's 32 output channels -> ring slot SLOT
template <int SLOT> __device__ __forceinline__ void bload1(const i32x4& rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 v[%c3:%c4], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(soff), "i"(W_B + 4 * SLOT), "i"(W_B + 4 * SLOT + 3));
}
// operand-row address of pixel-tile register q shifted by `sh` rows -> v[A]; hhs = chunk-column bit of the lane half + LDS stage base
template <int A> __device__ __forceinline__ void addr1_w(unsigned q, unsigned sh, unsigned hhs, unsigned cf0) {
    asm volatile("v_add_lshl_u32 v[%c[a]], %[q], %[sh], 4\n\t"
                 "v_and_b32 v[%c[t]], 0xffffff00, v[%c[a]]\n\t"
                 "v_and_or_b32 v[%c[a]], v[%c[a]], %[cf0], %[hhs]\n\t"
                 "v_lshl_add_u32 v[%c[a]], v[%c[t]], 3, v[%c[a]]"
                 ::[q] "v"(q), [sh] "s"(sh), [hhs] "v"(hhs), [cf0] "s"(cf0), [a] "i"(A), [t] "i"(W_TMP));
}

// One tap (four 16-deep steps) of a TM x (32 px x 32 ch) wave tile.  T = tap of the chunk; step n = 4 * T + s uses weight slot n % 18
// and fragment set s & 1; every MFMA is followed by the read that refills its fragment registers for step s + 2 (same tap) or for
// steps 0 / 1 of the next tap (address set 1 - S).  The last tap of a chunk reads nothing ahead (the next stage is only visible
// behind the barrier), the first starts cold.  NBW: 16-row LDS blocks this wave fetches per chunk, block j at step 2 * j.
template <int TM, int NBW, int T, int NTAP, int NSLOT, int NB = 0, typename Pre>
__device__ __forceinline__ void tap_wide(const unsigned (&qb)[TM], unsigned sh_cur, unsigned sh_next, unsigned hhs, unsigned cf0, Pre&& pre_step) {
    constexpr int S = T & 1;
    constexpr int A = W_AD + 8 * S, NA = W_AD + 8 * (1 - S);
    constexpr int F0 = W_AF, F1 = W_AF + 32;
    constexpr int W = 2 * TM - 1 < 15 ? 2 * TM - 1 : 15;
    constexpr bool COLD = T == 0, AHEAD = T != NTAP - 1;
    pre_step.template operator()<NB + 4 * T + 0>();
    if constexpr (COLD) {
        for_n<TM>([&]<int i>() { addr1_w<A + i>(qb[i], sh_cur, hhs, cf0); });
        for_n<TM>([&]<int i>() { lds_rd<F0 + 4 * i, A + i, 0u>(); });
        for_n<TM>([&]<int i>() { lds_rd<F1 + 4 * i, A + i, 512u>(); });
    }
    constexpr int B0 = W_B + 4 * ((NB + 4 * T + 0) % NSLOT), B1 = W_B + 4 * ((NB + 4 * T + 1) % NSLOT), B2 = W_B + 4 * ((NB + 4 * T + 2) % NSLOT),
                  B3 = W_B + 4 * ((NB + 4 * T + 3) % NSLOT);
    // (the next tap's addresses -- 4 VALU per pixel tile -- ride in every other MFMA gap of steps 0 and 1: as one block in front of
    // the tap they would leave the matrix pipe idle for ~130 cycles)
    static_assert(TM % 2 == 0, "address interleave");
    for_n<TM>([&]<int i>() {
        wait_lgkm<W>(); mfma_w<W_ACC + 16 * i, B0, F0 + 4 * i>(); lds_rd<F0 + 4 * i, A + i, 1024u>();
        if constexpr (AHEAD && i % 2 == 0) addr1_w<NA + i / 2>(qb[i / 2], sh_next, hhs, cf0);
    });
    pre_step.template operator()<NB + 4 * T + 1>();
    for_n<TM>([&]<int i>() {
        wait_lgkm<W>(); mfma_w<W_ACC + 16 * i, B1, F1 + 4 * i>(); lds_rd<F1 + 4 * i, A + i, 1536u>();
        if constexpr (AHEAD && i % 2 == 0) addr1_w<NA + TM / 2 + i / 2>(qb[TM / 2 + i / 2], sh_next, hhs, cf0);
    });
    pre_step.template operator()<NB + 4 * T + 2>();
    if constexpr (AHEAD) {
        for_n<TM>([&]<int i>() { wait_lgkm<W>(); mfma_w<W_ACC + 16 * i, B2, F0 + 4 * i>(); lds_rd<F0 + 4 * i, NA + i, 0u>(); });
        pre_step.template operator()<NB + 4 * T + 3>();
        for_n<TM>([&]<int i>() { wait_lgkm<W>(); mfma_w<W_ACC + 16 * i, B3, F1 + 4 * i>(); lds_rd<F1 + 4 * i, NA + i, 512u>(); });
    } else {
        for_n<TM>([&]<int i>() { wait_lgkm<(2 * TM - 1 - i < 15 ? 2 * TM - 1 - i : 15)>(); mfma_w<W_ACC + 16 * i, B2, F0 + 4 * i>(); });
        pre_step.template operator()<NB + 4 * T + 3>();
        for_n<TM>([&]<int i>() { wait_lgkm<TM - 1 - i>(); mfma_w<W_ACC + 16 * i, B3, F1 + 4 * i>(); });
    }
}

// NTAP = 9: the 3x3 stride-1 convolutions (a stage = the halo window of one 64-channel chunk).  NTAP = 2 (round 4): the
// DEEP 1x1 convolutions -- no halo, a stage holds two 64-channel planes of the tile's own pixel rows and the "taps" are the planes
// (row shift 0 / BM), so a barrier comes every 8 * TM MFMA; the weight ring has 8 slots there (8 steps per stage) and all DMA pieces of
// a stage are issued in front of its first step.
template <int TM, int NBW, bool DG, int NTAP = 9>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(50))) void conv2_wide_kernel(C2Params p) {
    static_assert(TM >= 2 && TM <= 8 && NBW >= 1 && NBW <= 9, "register map / DMA schedule");
    constexpr int NSTEP = 4 * NTAP;
    // weight ring: 18 slots over the 36 steps of a 3x3 stage; 16 slots over TWO 8-step stages of the 1x1 form (the loop body is
    // unrolled twice so that slot numbers stay compile-time constants)
    constexpr int NSLOT = NTAP == 9 ? W_NSLOT : 16;
    static_assert(NSLOT <= W_NSLOT && (NTAP == 9 ? NSTEP % NSLOT == 0 : NSLOT == 2 * NSTEP), "weight ring");
    static_assert(NTAP != 9 || 2 * (NBW - 1) < NSTEP - (NSLOT - 1), "every DMA piece of a stage must be older than the weights of its last step");
    static_assert(NTAP == 9 || (NTAP == 2 && NBW <= 8), "tap tables exist for 3x3 and for the two-plane 1x1 (one DMA block per step)");
    constexpr int BM = TM * 32, BN = 128;
    constexpr unsigned STAGE = NBW * 4 * 2048;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;

    // XCD-aware tile order: id = 8*slot + xcd, all channel tiles of one pixel tile back to back on one XCD
    const unsigned bid = blockIdx.x;
    const unsigned slot = bid >> 3;
    const unsigned mtile = (slot / (unsigned)p.n_ntiles) * 8u + (bid & 7u);
    const unsigned m0 = mtile * BM;
    if (m0 >= p.M) return;
    const int n0 = (int)(slot % (unsigned)p.n_ntiles) * BN;
    const int n_w = n0 + wave * 32;
    const bool alive = n_w < p.NOUT;

    Epilogue<TM, false, DG, TileOffs> epi(p, m0, n_w, mtile, alive, 0u);
    unsigned qb[TM];
    unsigned voff[NBW];
    if constexpr (NTAP == 2) {
        // 1x1: LDS row r of plane pl = pixel m0 + r, channels [128 * stage + 64 * pl, + 64); block b = wave + 4j holds rows 16b .. 16b + 15
        // of the stage's 2 * BM rows
#pragma unroll
        for (int i = 0; i < TM; ++i) qb[i] = 32u * i + (unsigned)l31;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const unsigned r2 = 16u * (unsigned)(wave + 4 * j) + (unsigned)(lane & 15);
            const unsigned pl = r2 >= (unsigned)BM ? 1u : 0u, r = r2 - pl * (unsigned)BM;
            const unsigned m = m0 + r;
            voff[j] = (r2 < 2u * BM && m < p.M) ? m * p.pix_bytes + pl * 128u + (unsigned)(lane >> 4) * 16u : OOB;
        }
    } else {
        auto pos0 = [&](unsigned m) -> unsigned {      // padded-linear position of tap offset (0,0) of destination pixel m
            const unsigned yall = udivm(m, p.mg_dw, p.sh_dw);
            const unsigned x = m - yall * (unsigned)p.DW;
            const unsigned n = udivm(yall, p.mg_dh, p.sh_dh);
            const unsigned y = yall - n * (unsigned)p.DH;
            return (n * (unsigned)p.Hp + y) * (unsigned)p.Wp + x;
        };
        const unsigned Lmin = pos0(m0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            unsigned m = m0 + 32u * i + (unsigned)l31;
            if (m > p.M - 1) m = p.M - 1;
            qb[i] = pos0(m) - Lmin;
        }
        // this wave's LDS blocks: b = wave + 4j; lane -> (row 16b + (lane & 15), chunk column lane >> 4 (+4 for the second piece))
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const unsigned L = Lmin + 16u * (unsigned)(wave + 4 * j) + (unsigned)(lane & 15);
            const unsigned row = udivm(L, p.mg_wp, p.sh_wp);
            const unsigned col = L - row * (unsigned)p.Wp;
            const unsigned n = udivm(row, p.mg_hp, p.sh_hp);
            const unsigned ry = row - n * (unsigned)p.Hp;
            const bool real = col >= (unsigned)p.PW && ry >= (unsigned)p.PH && n < (unsigned)p.NS;
            const unsigned pix = (n * (unsigned)p.SH + (ry - (unsigned)p.PH)) * (unsigned)p.SW + (col - (unsigned)p.PW);
            voff[j] = real ? pix * p.pix_bytes + (unsigned)(lane >> 4) * 16u : OOB;
        }
    }
    const i32x4 rsrc_a = make_rsrc(p.src, p.src_bytes);
    const i32x4 rsrc_b = make_rsrc(p.wpk, p.wpk_bytes);
    const unsigned bvoff = alive ? (unsigned)lane * 16u : OOB;
    // (NCC = 64-channel chunks of the contraction; a stage multiplies NTAP * 64 deep: 9 taps of one chunk, or two chunks)
    unsigned wsoff = (unsigned)(n_w >> 5) * (unsigned)p.NCC * (unsigned)((NTAP == 9 ? 9 : 1) * 4096);
    const unsigned smem_base = lds_off(smem);
    const int NSTAGE = NTAP == 9 ? p.NCC : p.NCC / 2;

#ifdef CS_DEBUG_V2
    // diagnostic build: shader-clock stamps around the phases + the constant 100 MHz counter at both ends (in-kernel clock =
    // delta s_memtime / delta s_memrealtime x 100 MHz); marker 3 = wide kernel (tools/stamp_probe.py)
    unsigned long long w_t[4], w_rt[2];
#define CS_WSTAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w_t[i]))
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w_rt[0]));
    CS_WSTAMP(0);
#else
#define CS_WSTAMP(i)
#endif
    epi.prefetch();
    asm volatile("; v[100:255] and a[128:255] are owned by the main loop" ::: "v100", "v255", "a128", "a255");
    for_n<16 * TM>([&]<int r>() { azero<W_ACC + r>(); });
    const unsigned cf0 = 0xf0u;

    if constexpr (NTAP == 9) {
        // ---- prologue: chunk 0 of the operand rows, weights of steps 0 .. 16
#pragma unroll
        for (int j = 0; j < NBW; ++j) dma_block(rsrc_a, smem_base + (unsigned)(wave + 4 * j) * 2048u, voff[j], 0u);
        for_n<NSLOT - 1>([&]<int k>() { bload1<k>(rsrc_b, bvoff, wsoff); wsoff += 1024u; });
        wait_vm<NSLOT - 1>();
        raw_barrier();
        CS_WSTAMP(1);

        for (int cc = 0; cc < NSTAGE; ++cc) {
            const unsigned cur = (cc & 1) ? STAGE : 0u, nxt = STAGE - cur;
            const unsigned hhs = (unsigned)hh * 256u + cur;
            const bool more = cc + 1 < NSTAGE;
            const unsigned nxt_soff = (unsigned)(cc + 1) * p.chunk_bytes;
            // in front of the MFMAs of step n: refill the slot step n - 1 has just released with the weights of step n + 17, send this
            // step's share of the next chunk's rows, then wait until the weights of step n have landed (17 younger loads + the DMA
            // pieces issued since then stay in flight)
            auto pre_step = [&]<int n>() {
                bload1<(n + NSLOT - 1) % NSLOT>(rsrc_b, bvoff, wsoff);
                wsoff += 1024u;
                if constexpr (n % 2 == 0 && n / 2 < NBW) {
                    constexpr int j = n / 2;
                    dma_block(rsrc_a, smem_base + nxt + (unsigned)(wave + 4 * j) * 2048u, more ? voff[j] : OOB, nxt_soff);
                }
                constexpr int lo = n - (NSLOT - 1) > 0 ? n - (NSLOT - 1) : 0;
                constexpr int first = (lo + 1) / 2, last = n / 2 < NBW - 1 ? n / 2 : NBW - 1;       // blocks j with lo <= 2j <= n
                constexpr int pieces = last >= first ? 2 * (last - first + 1) : 0;
                wait_vm<NSLOT - 1 + pieces>();
            };
            for_n<NTAP>([&]<int T>() {
                tap_wide<TM, NBW, T, NTAP, NSLOT>(qb, p.tap_sh[T], p.tap_sh[T + 1 < NTAP ? T + 1 : T], hhs, cf0, pre_step);
            });
            // every wave's pieces of the next chunk have landed (the wait of step 35 is younger than all of them) and this stage is read out
            raw_barrier();
            if (!more) epi.scr = smem_base + nxt + (unsigned)wave * EPI_WAVE;
        }
    } else {
        // ---- 1x1 on two-plane stages.  THREE LDS stages: the rows of stage s + 2 are sent during stage s, one 16-row block per step
        // (a burst of 2 * NBW DMA instructions in front of a stage kept the matrix pipe idle for ~2000 cycles: stamps, round 4), and
        // the weights of step g + 15 in front of step g.  In issue order, behind the weights of step g there are always 15 younger
        // loads and the 2 * NBW blocks of two stages' worth of steps: vmcnt(15 + 4 * NBW) -- except at a stage's last step, which
        // must also see the rows of the NEXT stage (sent during the previous one) land before the barrier hands them over: behind
        // the last of those only 16 - NBW loads and this stage's 2 * NBW pieces are younger.  The prologue replays the issue
        // pattern of two virtual stages (-2: rows of stage 0, -1: rows of stage 1) so that the immediates hold from step 0 on.
        for_n<16>([&]<int k>() {
            if constexpr (k >= 1) { bload1<k - 1>(rsrc_b, bvoff, wsoff); wsoff += 1024u; }
            if constexpr (k % 8 < NBW) {
                constexpr int j = k % 8, st = k / 8;
                dma_block(rsrc_a, smem_base + (unsigned)st * STAGE + (unsigned)(wave + 4 * j) * 2048u, (st < NSTAGE) ? voff[j] : OOB, (unsigned)st * p.chunk_bytes);
            }
        });
        wait_vm<16 + NBW>();
        raw_barrier();
        CS_WSTAMP(1);
        unsigned cur = 0u, dst = 2u * STAGE;             // LDS stage being multiplied / being filled (two stages ahead)
        auto stage = [&]<int NB>(int cc) {
            const unsigned hhs = (unsigned)hh * 256u + cur;
            const bool more2 = cc + 2 < NSTAGE;
            const unsigned soff2 = (unsigned)(cc + 2) * p.chunk_bytes;
            auto pre_step = [&]<int n>() {
                constexpr int l = n - NB;
                bload1<(n + NSLOT - 1) % NSLOT>(rsrc_b, bvoff, wsoff);
                wsoff += 1024u;
                if constexpr (l < NBW) dma_block(rsrc_a, smem_base + dst + (unsigned)(wave + 4 * l) * 2048u, more2 ? voff[l] : OOB, soff2);
                wait_vm<(l == 7 ? 16 + NBW : 15 + 4 * NBW)>();
            };
            for_n<NTAP>([&]<int T>() {
                tap_wide<TM, NBW, T, NTAP, NSLOT, NB>(qb, p.tap_sh[T], p.tap_sh[T + 1 < NTAP ? T + 1 : T], hhs, cf0, pre_step);
            });
            raw_barrier();
            cur = cur == 2u * STAGE ? 0u : cur + STAGE;
            dst = dst == 2u * STAGE ? 0u : dst + STAGE;
        };
        for (int cc = 0; cc < NSTAGE; cc += 2) {        // (the plan admits an even number of stages only)
            stage.template operator()<0>(cc);
            stage.template operator()<8>(cc + 1);
        }
        wait_vm<0>();                                    // the out-of-range tail pieces write zeros: gone before the scratch is used
        raw_barrier();
        epi.scr = smem_base + (unsigned)wave * EPI_WAVE;
    }
    // MFMA results -> VALU reads: the last MFMA needs its 16 passes (no hardware interlock on this path)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7");
    CS_WSTAMP(2);
    for_n<TM>([&]<int i>() {
        const f32x16 d = acc_read_w<i>();
        epi.template operator()<i>(d);
    });
    epi.finish();
    wait_vm<0>();                     // the out-of-range tail loads are gone before the registers / LDS are released
#ifdef CS_DEBUG_V2
    CS_WSTAMP(3);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w_rt[1]));
    if (p.dbg && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 6;
        o[0] = 3; o[1] = w_t[1] - w_t[0]; o[2] = w_t[2] - w_t[1]; o[3] = w_t[3] - w_t[2]; o[4] = w_t[3] - w_t[0]; o[5] = w_rt[1] - w_rt[0];
    }
#endif
}

// =================================================================================================
// Ring kernel: 1x1 convolutions of ANY contraction depth as a persistent, software-pipelined stream.  Stamped, a one-shot
// workgroup (load the whole pixel tile, one barrier, multiply, epilogue -- round 2's first 1x1 kernel) spends most of its life with nothing in flight (first-load latency, then an epilogue behind
// which no load is queued); two such workgroups per compute unit reach 3.4-4.3 TB/s on the layers that are HBM-bound, and the
// first-generation kernel that served the deep contractions (C > 256) sat at ~2 TB/s / 15 % MFMA.  Here a workgroup keeps ONE
// output-channel tile and walks (pixel tile, 64-channel chunk) steps over a ring of three 16 KiB LDS slots:
//   step s:  weights of step s+2 -> registers | wait: own pieces of step s landed | barrier | DMA of step s+2 -> slot (s+2)%3
//            | 4*TM MFMA of step s | on the last chunk of a pixel tile: epilogue, accumulators back to zero
// so two steps of operands (per wave 8 KiB) are always in flight -- through every epilogue too.
//   * EVERY load inside the loop is inline asm with hand-counted s_waitcnt: a load the compiler can see makes it insert a wait
//     that counts only its own operations, which drains our younger DMA / weight loads as well (this is what made the first
//     persistent attempt slower than the one-shot kernel).  The epilogue's residual / add operand therefore arrives by LDS-DMA
//     (row arrangement, armed one pixel tile ahead during the previous epilogue), its mask words land in owned registers
//     v[96:99]; stores are compiler-issued (they cause no waits) but their NUMBER per epilogue is fixed (operands that are
//     absent use zero-sized buffers), because every later vmcnt immediate has to count them: E_OPS per epilogue.
//   * step s needs the weights B(s) and the pieces DMA(s); issued after DMA(s): [E(s-2)] B(s+1) DMA(s+1) [E(s-1)] B(s+2)
//     -> vmcnt(12 + E_OPS * (epilogues among steps s-1, s-2)).  Loads past the last step are issued with out-of-range
//     offsets (they count, move nothing) so the immediates hold to the end.
//   * the residual of the tile that finishes at step s was armed by E(s - NCC): for NCC >= 3 the wait above covers it, for
//     NCC = 1, 2 the epilogue waits for vmcnt(E_OPS - 5 + 8 * NCC) (5 operations per 32-pixel tile: 2 stores, 1 bit store or
//     mask load, 2 DMA).
//   * the exchange between accumulator and row arrangement uses the residual's own 2 KiB of LDS per 32-pixel tile, rows of
//     64 bytes, 16-byte slot = piece ^ ((row >> 2) & 3): conflict-free for both arrangements, and the DMA (which can only write
//     lane-contiguous LDS) applies the swizzle on the global side.
// LDS: 3 x 16 KiB + 4 waves x TM x 2 KiB = 80 KiB (TM = 4) -> two workgroups per compute unit.
// =================================================================================================
constexpr int R_MB = 96;

template <int TM, bool DG> struct RingEpilogue {
    static constexpr int E_OPS = 5 * TM + (DG ? 2 : 0);
    const C2Params& p;
    unsigned m0w = 0, m0w_next = 0xffffffffu, slab_row = 0;
    int n_w;
    bool alive;
    unsigned region;           // LDS byte address of this wave's TM x 2 KiB
    float sh[16];
    float s1[8];
    i32x4 q_res, q_bin;
    __amdgpu_buffer_rsrc_t r_dst, r_bout, r_slab;
#ifdef CS_DEBUG_V2
    unsigned long long e_acc[4] = {0, 0, 0, 0}, e_t = 0;
#endif

    __device__ __forceinline__ RingEpilogue(const C2Params& p_, int n_w_, bool alive_, unsigned region_) : p(p_), n_w(n_w_), alive(alive_), region(region_) {}

    // row arrangement of this kernel: lane L <-> LDS slot L of a 1 KiB half tile = row 16q + (L >> 2), slot L & 3, which holds
    // piece (L & 3) ^ ((row >> 2) & 3) = (L & 3) ^ ((L >> 4) & 3) of that pixel's 64 bytes
    __device__ __forceinline__ unsigned piece() const {
        const unsigned lane = threadIdx.x & 63;
        return (lane & 3u) ^ ((lane >> 4) & 3u);
    }
    TileOffs cur, nxt;         // offsets of the pixel tile being finished / of the one whose operands are being armed
    __device__ __forceinline__ void set_tile(unsigned base, unsigned base_next) {
        m0w = base; m0w_next = base_next;
        cur.set(base, alive, (unsigned)p.NOUT, (unsigned)n_w, piece(), p.M);
        nxt.set(base_next, alive, (unsigned)p.NOUT, (unsigned)n_w, piece(), p.M);
    }
    __device__ __forceinline__ unsigned row_lds(int i, int q) const { return region + (unsigned)i * 2048u + (unsigned)q * 1024u + (threadIdx.x & 63u) * 16u; }
    __device__ __forceinline__ unsigned acc_lds(int i, int j) const {
        const unsigned lane = threadIdx.x & 63, l31 = lane & 31u;
        return region + (unsigned)i * 2048u + l31 * 64u + (((2u * j + (lane >> 5)) ^ ((l31 >> 2) & 3u)) * 16u);
    }

    // residual / add rows (and mask words) of 32-pixel tile I of the pixel tile that starts at `base`: fly from now on
    // byte offset of the add operand's row for (32-pixel tile i, half q): the destination row itself, or -- compact strided operand --
    // the row of pixel (y / 2, x / 2) when both coordinates are even, nothing otherwise
    __device__ __forceinline__ unsigned add_off(const TileOffs& t, int i, int q) const {
        if (p.add_stride == 1) return t.row(p.M, (unsigned)p.NOUT, i, q);
        const unsigned m = t.mr + 16u * (unsigned)(2 * i + q);
        const unsigned yall = udivm(m, p.mg_dw, p.sh_dw);
        const unsigned x = m - yall * (unsigned)p.DW;
        const unsigned n = udivm(yall, p.mg_dh, p.sh_dh);
        const unsigned y = yall - n * (unsigned)p.DH;
        const unsigned mc = (n * (unsigned)p.AH + (y >> 1)) * (unsigned)p.AW + (x >> 1);
        return (m < p.M && !((x | y) & 1u)) ? (mc * (unsigned)p.NOUT + (unsigned)n_w + 8u * piece()) * 2u : OOB;
    }
    template <int I> __device__ __forceinline__ void arm(const TileOffs& t) {
        const unsigned v0 = add_off(t, I, 0), v1 = add_off(t, I, 1);
        const unsigned lds = __builtin_amdgcn_readfirstlane(region + (unsigned)I * 2048u);
        // (round 4: the `nt` cache policy on these two loads -- the operand is read once -- measured equal on layer1 and 25-35 % slower
        // on layer2-4, where the residual is still in the Infinity Cache from the launch that wrote it: default policy)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
            "s_add_u32 m0, %0, 0x400\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %2, %3, 0 offen lds"
            ::"s"(lds), "v"(v0), "v"(v1), "s"(q_res)
            : "memory", "scc");
        if constexpr (DG) {
            const unsigned b = t.bit(p.M, (unsigned)p.NOUT, I);
            asm volatile("buffer_load_dword v[%c2], %0, %1, 0 offen" ::"v"(b), "s"(q_bin), "i"(R_MB + I) : "memory");
        }
    }
    template <int... Is> __device__ __forceinline__ void arm_all(const TileOffs& t, std::integer_sequence<int, Is...>) { (arm<Is>(t), ...); }

    __device__ __forceinline__ void prefetch(unsigned first_base) {
        const int lane = threadIdx.x & 63;
        const int hh = lane >> 5;
        const unsigned out_bytes = p.M * (unsigned)p.NOUT * 2u;
        r_dst = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, out_bytes, 0x00020000);
        r_bout = __builtin_amdgcn_make_buffer_rsrc(p.bits_out, 0, p.bits_out ? out_bytes >> 4 : 0u, 0x00020000);
        r_slab = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, p.slab ? 0x7ffffff0u : 0u, 0x00020000);
        q_res = make_rsrc(p.residual, !p.residual ? 0u : p.add_stride == 1 ? out_bytes : (unsigned)(p.NS * p.AH * p.AW * p.NOUT) * 2u);
        q_bin = make_rsrc(p.bits_in, p.bits_in ? out_bytes >> 4 : 0u);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!DG && p.shift && alive) s = *reinterpret_cast<const float4*>(p.shift + n_w + 8 * g + 4 * hh);
            sh[4 * g] = s.x; sh[4 * g + 1] = s.y; sh[4 * g + 2] = s.z; sh[4 * g + 3] = s.w;
        }
        // the only compiler-visible loads of the kernel: complete before the loop, so hipcc places no vmcnt wait inside it
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(sh[r]));
        TileOffs first;
        first.set(first_base, alive, (unsigned)p.NOUT, (unsigned)n_w, piece(), p.M);
        arm_all(first, std::make_integer_sequence<int, TM>{});
    }

    template <int I> __device__ __forceinline__ void operator()(const f32x16& acc) {
        const int lane = threadIdx.x & 63;
        const unsigned pc = piece();
        CS_ESTART();
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = DG ? acc[r] : acc[r] + sh[r];
        if (p.residual) {
            uint4 xa = lds_get(acc_lds(I, 0)), xb = lds_get(acc_lds(I, 1));
            __builtin_amdgcn_wave_barrier();
            swap32(xa.x, xa.z); swap32(xa.y, xa.w);
            swap32(xb.x, xb.z); swap32(xb.y, xb.w);
            const unsigned rw[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float lo, hi;
                unpack2(rw[k], lo, hi);
                v[2 * k] += lo;
                v[2 * k + 1] += hi;
            }
        }
        unsigned pk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        if (!DG && p.act == CS_ACT_RELU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) pk[k] = relu_bf16x2(pk[k]);
        }
        swap32(pk[0], pk[2]); swap32(pk[1], pk[3]);
        swap32(pk[4], pk[6]); swap32(pk[5], pk[7]);
        CS_ETICK(0);
        lds_put(acc_lds(I, 0), make_uint4(pk[0], pk[1], pk[2], pk[3]));
        lds_put(acc_lds(I, 1), make_uint4(pk[4], pk[5], pk[6], pk[7]));
        __builtin_amdgcn_wave_barrier();
        uint4 o0 = lds_get(row_lds(I, 0)), o1 = lds_get(row_lds(I, 1));
        __builtin_amdgcn_wave_barrier();
        asm volatile("" : "+v"(o0.x), "+v"(o1.w));
        CS_ETICK(1);
        if constexpr (DG) {
            if (p.bits_in) {
                unsigned mbw;
                asm volatile("v_mov_b32 %0, v[%c1]" : "=v"(mbw) : "i"(R_MB + I));
                const unsigned w0 = quad<0x00>(mbw), w1 = quad<0x55>(mbw);
                o0 = keep_bits8(o0, w0 >> (8u * pc));
                o1 = keep_bits8(o1, w1 >> (8u * pc));
            }
            if constexpr (I == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s1[k] = 0.f;
            }
            if (p.slab) {
                const unsigned ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float lo, hi;
                    unpack2(ow[k], lo, hi);
                    s1[2 * (k & 3)] += lo; s1[2 * (k & 3) + 1] += hi;
                }
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), r_dst, cur.row(p.M, (unsigned)p.NOUT, I, 0), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), r_dst, cur.row(p.M, (unsigned)p.NOUT, I, 1), 0, 0);
        if constexpr (!DG) {
            // always issued (a NULL bit plane has a zero-sized buffer): the vmcnt arithmetic of the loop counts it
            unsigned w0 = 0u, w1 = 0u;
            if (p.bits_out) {
                const bool relu = p.act == CS_ACT_RELU;
                w0 = (relu ? nz_bits8(o0) : pos_bits8(o0)) << (8u * pc); w1 = (relu ? nz_bits8(o1) : pos_bits8(o1)) << (8u * pc);
                w0 |= quad<0xb1>(w0); w1 |= quad<0xb1>(w1);
                w0 |= quad<0x4e>(w0); w1 |= quad<0x4e>(w1);
            }
            __builtin_amdgcn_raw_buffer_store_b32((lane & 1) ? w1 : w0, r_bout, cur.bit(p.M, (unsigned)p.NOUT, I), 0, 0);
        }
        CS_ETICK(2);
        arm<I>(nxt);
        CS_ETICK(3);
    }

    __device__ __forceinline__ void finish() {
        if constexpr (DG) {
            const int lane = threadIdx.x & 63;
            // lanes sharing a piece: any (lane >> 2) & 3, and slot = piece ^ g for g = lane >> 4
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = s1[k];
                a += __shfl_xor(a, 4, 64);
                a += __shfl_xor(a, 8, 64);
                float t = a;
#pragma unroll
                for (int g = 1; g < 4; ++g) t += __shfl(a, ((lane & 3) ^ g) | (g << 4), 64);
                s1[k] = t;
            }
            // lanes 0-3 hold piece = lane; always two stores (the loop's vmcnt arithmetic counts them)
            const unsigned off = (lane < 4 && alive && p.slab) ? ((slab_row * 2u * (unsigned)p.NOUT + (unsigned)n_w + 8u * (unsigned)lane) * 4u) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(s1[0], s1[1], s1[2], s1[3])), r_slab, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(s1[4], s1[5], s1[6], s1[7])), r_slab, off, 16, 0);
        }
    }
};

// GA: gathered operand rows (the pixel-paired stem, cs_stem_fwd_packed); a template flag because its per-tile state costs registers
template <int TM, int WM, int WN, bool DG, bool GA = false>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(48))) void conv2_ring_kernel(C2Params p, int n_groups) {
    static_assert(WM * WN == 4 && WM * TM <= 4 && TM >= 2, "a ring slot holds up to 128 pixel rows");
    // BM < 128 (1 x 4 waves of 2 or 3 pixel tiles, round 3): the slot keeps its 16 KiB stride, its tail is never read; the DMA
    // blocks past the tile are issued out of range (they count in vmcnt, and write their zeros into that tail)
    constexpr int BM = WM * TM * 32, BN = WN * 32;
    constexpr unsigned SLOT = 16384u, NSLOT = 3u, SLOT_ROWS = 128u;
    using Epi = RingEpilogue<TM, DG>;
    constexpr int E = Epi::E_OPS;
    static_assert(12 + 2 * E <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
    // workgroup -> (output-channel tile, pixel-tile group): id = 8*slot + xcd; the channel tiles of one group sit on one XCD
    const unsigned bid = blockIdx.x;
    const unsigned gslot = bid >> 3;
    const unsigned group = (gslot / (unsigned)p.n_ntiles) * 8u + (bid & 7u);
    const int n0 = (int)(gslot % (unsigned)p.n_ntiles) * BN;
    const int n_w = n0 + wn * 32;
    const bool alive = n_w < p.NOUT;
    if (group * BM >= p.M) return;
    const int NCC = p.NCC;

    unsigned qb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qb[i] = (unsigned)((wm * TM + (i < TM ? i : 0)) * 32 + l31);
    const i32x4 rsrc_a = make_rsrc(p.src, p.src_bytes);
    const i32x4 rsrc_b = make_rsrc(p.wpk, p.wpk_bytes);
    const unsigned smem_base = lds_off(smem);
    const unsigned hhb = (unsigned)hh * 256u;
    const unsigned cf0 = 0xf0u;
    const unsigned w_base = (unsigned)(n_w >> 5) * (unsigned)NCC * 4096u;

    // this wave's two 16-row blocks (wave, wave + 4) of pixel tile mt: byte offset of (row, chunk column) or out of range
    // gathered rows (the pixel-paired stem: 7 x 4 taps of 16 bytes, stride 2 rows / 1 pair, pad 3 rows / 2 pairs): per block the byte
    // base of the row's image (~0: no such row) and its first tap's source row / pair; slot (kh, pw) adds (kh * GWp + pw) * 16 bytes
    int f_gy[2] = {0, 0}, f_gx[2] = {0, 0};
    auto tile_voff = [&](unsigned mt, unsigned (&vo)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned m = mt * BM + 16u * (unsigned)(wave + 4 * j) + (unsigned)(lane & 15);
            if (BM < 128 && 16 * (wave + 4 * j) >= BM) { vo[j] = OOB; continue; }
            if constexpr (GA) {
                const unsigned yall = udivm(m, p.mg_dw, p.sh_dw);
                const unsigned x = m - yall * (unsigned)p.DW;
                const unsigned n = udivm(yall, p.mg_dh, p.sh_dh);
                const unsigned y = yall - n * (unsigned)p.DH;
                vo[j] = (m < p.M && mt != 0xffffffffu) ? n * (unsigned)(p.GH * p.GWp) * 16u : 0xffffffffu;
                f_gy[j] = 2 * (int)y - 3;
                f_gx[j] = (int)x - 2;
                continue;
            }
            unsigned pix = m;
            if (p.stride > 1) {
                const unsigned yall = udivm(m, p.mg_dw, p.sh_dw);
                const unsigned x = m - yall * (unsigned)p.DW;
                const unsigned n = udivm(yall, p.mg_dh, p.sh_dh);
                const unsigned y = yall - n * (unsigned)p.DH;
                pix = (n * (unsigned)p.SH + y * (unsigned)p.stride) * (unsigned)p.SW + x * (unsigned)p.stride;
            }
            vo[j] = (m < p.M && mt != 0xffffffffu) ? pix * p.pix_bytes + (unsigned)(lane >> 4) * 16u : OOB;
        }
    };

    Epi epi(p, n_w, alive, smem_base + NSLOT * SLOT + (unsigned)wave * (unsigned)(TM * 2048));
    unsigned mt = group;                                    // the pixel tile being multiplied
    epi.prefetch(mt * BM + (unsigned)(wm * TM * 32));
    asm volatile("; v[96:255] are owned by the main loop" ::: "v96", "v255");
    vzero_seq(std::make_integer_sequence<int, 16 * TM>{});

    // fetch cursor = step s+2: pixel tile f_mt (~0: past the end), chunk f_c, ring slot f_slot
    unsigned f_mt = mt, f_slot = 0;
    int f_c = 0;
    unsigned f_vo[2];
    tile_voff(f_mt, f_vo);
    auto advance_fetch = [&]() {
        f_slot = f_slot == NSLOT - 1 ? 0u : f_slot + 1u;
        if (++f_c == NCC) {
            f_c = 0;
            if (f_mt != 0xffffffffu) {
                f_mt += (unsigned)n_groups;
                if (f_mt * BM >= p.M) f_mt = 0xffffffffu;
            }
            tile_voff(f_mt, f_vo);
        }
    };
    auto issue_b = [&]<int J>() {
        const unsigned bv = (alive && f_mt != 0xffffffffu) ? (unsigned)lane * 16u : OOB;
        bload4<J>(rsrc_b, bv, w_base + (unsigned)f_c * 4096u);
    };
    auto issue_a = [&]() {
        const unsigned dst = smem_base + f_slot * SLOT;
        if constexpr (GA) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                auto slot_off = [&](int sl) -> unsigned {
                    const int kh = sl >> 2, pw = sl & 3;
                    const int iy = f_gy[j] + kh, px = f_gx[j] + pw;
                    const bool ok = f_vo[j] != 0xffffffffu && sl < 28 && (unsigned)iy < (unsigned)p.GH && (unsigned)px < (unsigned)p.GWp;
                    return ok ? f_vo[j] + (unsigned)(iy * p.GWp + px) * 16u : OOB;
                };
                const int s0 = f_c * 8 + (lane >> 4);
                dma_block2(rsrc_a, dst + (unsigned)(wave + 4 * j) * 2048u, slot_off(s0), slot_off(s0 + 4));
            }
            return;
        }
        dma_block(rsrc_a, dst + (unsigned)wave * 2048u, f_vo[0], (unsigned)f_c * 128u);
        dma_block(rsrc_a, dst + (unsigned)(wave + 4) * 2048u, f_vo[1], (unsigned)f_c * 128u);
    };
    // prologue = steps -2 and -1 of the schedule
    issue_b.template operator()<0>(); issue_a(); advance_fetch();
    issue_b.template operator()<1>(); issue_a(); advance_fetch();
    // (Round 3, second experiment: a 512-thread variant with four LOADER waves issuing the operand-row DMA five steps ahead over a
    // six-slot ring (one workgroup per compute unit), the consumers waiting for their own weight loads only -- vmcnt retires in order
    // per wave, so a wave that also streams weights cannot keep more than two steps of rows in flight.  Correct on all 52 packed
    // tests, and slower or equal everywhere: 1024 -> 256 @19x19 26.0 -> 29.9 us, 2048 -> 512 @10x10 (one workgroup per compute unit
    // either way) 25.9 -> 26.4 us.  PMC on that launch: per 64-deep step a consumer wave is parked ~660 cycles (barrier, first LDS
    // fragment, weight wait), issues for ~590 and sits behind its own MFMAs for ~590 (512 cycles of MFMA busy): the rows were never
    // what it waited for.  Removed; profiles/round3_notes.md.)
    // (Round 3 tried issuing the loads of step s+2 BETWEEN the MFMAs of step s instead of in front of them -- B0 A0 B1 A1 B2 B3 after
    // MFMA 0..5, vmcnt(8 + ...) -- to take the ~1000 cycles of issue that round 2's stamps showed off the critical path of the deep
    // contractions.  All tests passed and nothing moved: 2048 -> 512 @10x10 26.4 vs 26.0 us, 1024 -> 256 @19x19 28.5 vs 28.6 us.  The
    // stamps' own lgkmcnt(0) had serialised what they measured; removed again.)
    int c = 0;                        // chunk of the step being multiplied
    unsigned slot = 0;
    bool fin1 = false, fin2 = false;  // steps s-1 / s-2 ended with an epilogue
#ifdef CS_DEBUG_V2
    unsigned long long r_acc[5] = {0, 0, 0, 0, 0}, r_a, r_b;
    unsigned r_tiles = 0, r_steps = 0;
#define CS_RTICK(k) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_b)); r_acc[k] += r_b - r_a; r_a = r_b; } while (0)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_a));
#else
#define CS_RTICK(k)
#endif
    auto step = [&]<int CUR>() -> bool {
        issue_b.template operator()<(CUR + 2) % 3>();
        const int nfin = (int)fin1 + (int)fin2;
        if (nfin == 0) wait_vm<12>();
        else if (nfin == 1) wait_vm<12 + E>();
        else wait_vm<12 + 2 * E>();
        CS_RTICK(0);
        raw_barrier();
        CS_RTICK(1);
        issue_a();
        advance_fetch();
        CS_RTICK(2);
        const unsigned sh_cur = slot * SLOT_ROWS;          // the slot, in LDS rows
        const bool fin = c == NCC - 1;
        bool more = true;
        if (!fin) {
            tap_mfma<TM, CUR, 0, 0u, 0u, true, false, true, false>(qb[0], qb[1], qb[2], qb[3], sh_cur, 0u, hhb, cf0);
            ++c;
            CS_RTICK(3);
        } else {
            const unsigned mt_next = mt + (unsigned)n_groups;
            const bool has_next = mt_next * BM < p.M;
            epi.set_tile(mt * BM + (unsigned)(wm * TM * 32), has_next ? mt_next * BM + (unsigned)(wm * TM * 32) : 0xffffffffu);
            epi.slab_row = mt * WM + wm;
            auto on_tile = [&]<int i>(const f32x16& d) {
                // the operands E(s - NCC) armed for this 32-pixel tile (NCC >= 3: older than what the step's own wait covered)
                if (NCC == 1) wait_vm_mem<(E - 5 + 8 < 63 ? E - 5 + 8 : 63)>();
                else if (NCC == 2) wait_vm_mem<(E - 5 + 16 < 63 ? E - 5 + 16 : 63)>();
                epi.template operator()<i>(d);
            };
            tap_mfma<TM, CUR, 0, 0u, 0u, true, false, true, true>(qb[0], qb[1], qb[2], qb[3], sh_cur, 0u, hhb, cf0, on_tile);
            epi.finish();
            vzero_seq(std::make_integer_sequence<int, 16 * TM>{});
            c = 0;
            mt = mt_next;
            more = has_next;
            CS_RTICK(4);
#ifdef CS_DEBUG_V2
            ++r_tiles;
#endif
        }
#ifdef CS_DEBUG_V2
        ++r_steps;
#endif
        slot = slot == NSLOT - 1 ? 0u : slot + 1u;
        fin2 = fin1;
        fin1 = fin;
        return more;
    };
    for (;;) {
        if (!step.template operator()<0>()) break;
        if (!step.template operator()<1>()) break;
        if (!step.template operator()<2>()) break;
    }
    wait_vm<0>();                     // the out-of-range tail loads (LDS-DMA among them) are gone before the LDS is released
#ifdef CS_DEBUG_V2
    if (p.dbg && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 6;
        o[0] = 2; o[1] = r_acc[0]; o[2] = r_acc[1]; o[3] = r_acc[2] + r_acc[3]; o[4] = r_acc[4]; o[5] = ((unsigned long long)r_steps << 32) | r_tiles;
        if (p.act >= 200) { o[3] = r_acc[2]; o[1] = r_acc[3]; }
        if (p.act >= 300) { o[1] = epi.e_acc[0]; o[2] = epi.e_acc[1]; o[3] = epi.e_acc[2]; o[4] = epi.e_acc[3]; }
    }
#endif
}

// ---- weights [ROWS][taps][COLS] bf16 (ROWS = destination channels, COLS = contraction channels, both staged layouts of
// cs_weight_prep have this shape) -> MFMA-fragment order [row tile 32][chunk 64][tap][k16][lane][8]; flip = taps mirrored
// (data gradient).  One thread per 16 bytes.
__global__ __launch_bounds__(256) void pack_weights_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, int rows, int cols, int R, int S,
                                                           int flip, long long total) {
    const int ntap = R * S, ncc = cols / 64;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        long long u = idx >> 6;
        const int k16 = (int)(u & 3); u >>= 2;
        const int t = (int)(u % ntap); u /= ntap;
        const int cc = (int)(u % ncc);
        const int rt = (int)(u / ncc);
        const int row = rt * 32 + (lane & 31);
        const int col = cc * 64 + k16 * 16 + 8 * (lane >> 5);
        const int ts = flip ? ntap - 1 - t : t;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rows) v = w[(((long long)row * ntap + ts) * cols + col) >> 3];
        out[idx] = v;
    }
}

void magic(unsigned d, unsigned& mg, unsigned& sh) {
    int l = 0;
    while ((1u << l) < d) ++l;
    const unsigned long long two_p = 1ull << (31 + l);
    mg = (unsigned)(two_p / d + 1ull);
    sh = (unsigned)(l - 1);
}

// geometry of the packed-operand launch; cfg: 0 = not served, 1 = 128 px x 128 ch (1x4 waves), 2 = 256 px x 64 ch (2x2 waves),
// 3 = wide kernel (one wave per SIMD), (32 * tm) px x 128 ch
struct C2Plan {
    int cfg, nbw, ncc, rows;       // rows = partial column-sum rows
    int tm;                        // pixel tiles (32 px) per wave: 1 x 4 configurations only
    int t2d;                       // halo kernel on 8 x 16-pixel tiles (wide images)
    C2Params p;
};

// Pixel-tile height of the wide kernel (one workgroup per compute unit: 256 run at once).  The busiest compute unit sets the time:
// rounds-of-residency x tile height, the taller tile on ties (fewer weight passes).  0 = do not use the wide kernel.
// CELLSEG_WIDE (A/B flavour): 1 = never, 4 / 6 / 8 = force that height wherever the window fits.
int pick_wide_tm(long long M, int n_ntiles, int ncc) {
    static const int knob = cs_env_int_("CELLSEG_WIDE", 0);
    if (knob == 1) return 0;
    if (knob == 4 || knob == 6 || knob == 8) return knob;
    if (ncc < 2) return 0;
    int best = 0;
    const long long cus = cs_device_cus_();            // one workgroup per compute unit: `cus` run at once
    long long best_cost = 1ll << 60, best_tiles = 0;
    for (int tm = 8; tm >= 4; tm -= 2) {
        const long long tiles = ((M + 32 * tm - 1) / (32 * tm)) * n_ntiles;
        const long long cost = ((tiles + cus - 1) / cus) * tm;
        if (cost < best_cost) { best_cost = cost; best = tm; best_tiles = tiles; }
    }
    // Measured (tools/wide_ab.sh, profiles/round4_notes.md): with every tile resident at once -- one round -- the wide kernel is ahead
    // of two 4-wave halo workgroups per compute unit (ResNet-50 layer3 34.2 -> 31.0 us, layer4 41.8 -> 33.9 us, decoder 1024 -> 512
    // 106 -> 96 us); with a second round its serial prologue and epilogue are paid twice with nothing to hide them behind
    // (layer2: 32.7 -> 40.2 us), and the halo kernel stays.
    return best_tiles <= cus ? best : 0;
}

// Pixel-tile height of the 1 x 4 halo configuration.  Workgroups do not run in lock-step rounds, so a shorter tile only pays
// where the whole grid is resident at once (<= 512 workgroups on 256 compute units, two each) and the busiest compute unit's
// share shrinks: ResNet-50 layer3 (M = 23104, N = 256) is 362 workgroups of 128 px -- 106 compute units hold two -- but 482
// of 96 px, each 3/4 of the work: measured 34.0 -> 29.3 us forward.  Elsewhere the shorter tile loses: layer2 (722 -> 963
// workgroups: the same total work plus 241 more prologues, 31.9 -> 33.3 us) and layer4 (200 workgroups of 128 px, one per compute
// unit; 64-px tiles re-read the 4.7 MB of weights twice as often and run into the L2 -> LDS feed, 33.4 -> 37.9 us).
int pick_tm(long long M, int n_ntiles) {
    static const int forced = cs_env_int_("CELLSEG_TM", 0);   // experiments only
    if (forced >= 2 && forced <= 4) return forced;
    const long long wg4 = ((M + 127) / 128) * n_ntiles, wg3 = ((M + 95) / 96) * n_ntiles;
    return (wg4 > 256 && wg4 <= 512 && wg3 <= 512) ? 3 : 4;
}

// The same decision for the deep 1x1 convolutions (wide kernel on two-plane stages, cfg 8).  CELLSEG_WIDE1 (A/B flavour): 1 = never,
// 4 / 6 / 8 = force that height.  Only where the ring kernel is chain-bound: >= 8 chunks (512 contraction channels), all tiles resident
// at once; the short contractions into wide outputs stay on the ring kernel (HBM-bound, many pixel tiles per workgroup).
int pick_wide1_tm(long long M, int n_ntiles, int ncc) {
    static const int knob = cs_env_int_("CELLSEG_WIDE1", 0);
    if (knob == 1) return 0;
    if (knob == 4 || knob == 6) return knob;
    if (ncc < 8) return 0;
    int best = 0;
    const long long cus = cs_device_cus_();
    long long best_cost = 1ll << 60, best_tiles = 0;
    for (int tm = 6; tm >= 4; tm -= 2) {               // (three LDS stages: 3 x 8 KiB x tm per wave column -- 144 KiB at tm = 6)
        const long long tiles = ((M + 32 * tm - 1) / (32 * tm)) * n_ntiles;
        const long long cost = ((tiles + cus - 1) / cus) * tm;
        if (cost < best_cost) { best_cost = cost; best = tm; best_tiles = tiles; }
    }
    return best_tiles <= cus ? best : 0;
}

const bool g_v2_off = cs_env_flag_("CELLSEG_NO_V2");     // A/B flavour only

bool plan_halo(const CsConvGeom* g, int dgrad, C2Plan& pl) {
    if (g_v2_off) return false;
    if (g->groups > 1) return false;
    if (g->R != 3 || g->S != 3 || g->stride != 1 || g->pad < 0 || g->pad > 2) return false;
    const int SC = dgrad ? g->K : g->C, NOUT = dgrad ? g->C : g->K;
    const int SH = dgrad ? g->P : g->H, SW = dgrad ? g->Q : g->W;
    const int DH = dgrad ? g->H : g->P, DW = dgrad ? g->W : g->Q;
    if (SC % 64 || NOUT % 64 || DH < 2 || DW < 2 || SH < 1 || SW < 1) return false;
    const int PWH = dgrad ? g->R - 1 - g->pad : g->pad;
    if (PWH < 0) return false;
    const long long M = (long long)g->N * DH * DW;
    const unsigned long long src_bytes = (unsigned long long)g->N * SH * SW * SC * 2ull;
    if (M >= (1ll << 31) - 512 || src_bytes >= 0x80000000ull || (unsigned long long)M * NOUT * 2ull >= (1ull << 40)) return false;
    const int Wp = SW + PWH, Hp = SH + PWH;
    if (Wp < 2 || Hp < 2 || Wp < DW || Hp < DH) return false;
    if ((long long)g->N * Hp * Wp + 4096 >= (1ll << 31)) return false;
    const int ncc = SC / 64;
    auto window_blocks = [&](int BM) {
        const long long span = (BM - 1) + (long long)cs_ceil_div(BM - 1, DW) * (Wp - DW) + (long long)cs_ceil_div(BM - 1, (long long)DH * DW) * ((long long)(Hp - DH) * Wp);
        const long long rows = span + (long long)(g->R - 1) * Wp + (g->S - 1) + 1;
        return (int)((rows + 15) / 16);
    };
    int cfg = 0, nbw = 0, tm = 4;
    if (NOUT % 128 == 0) {
        // wide kernel first: window blocks per wave rounded up to an instantiated count (TM 4: 3 / 5, TM 6: 5 / 7, TM 8: 7 / 9)
        const int wtm = pick_wide_tm(M, NOUT / 128, ncc);
        if (wtm) {
            const int nb = (window_blocks(32 * wtm) + 3) / 4;
            const int lo = wtm == 4 ? 3 : wtm == 6 ? 5 : 7;
            if (nb <= lo + 2) { cfg = 3; tm = wtm; nbw = nb <= lo ? lo : lo + 2; }
        }
    }
    if (!cfg && NOUT % 128 == 0) {
        tm = pick_tm(M, NOUT / 128);
        for (; tm <= 4; ++tm) {              // a shorter tile has a smaller window; fall back to taller ones only if it somehow does not fit
            const int nb = (window_blocks(32 * tm) + 3) / 4;
            if (ncc == 1 ? nb <= 8 : nb <= 5) { cfg = 1; nbw = nb < 3 ? 3 : nb; break; }
        }
    } else if (NOUT == 64 && ncc == 1) {
        const int nb = (window_blocks(256) + 3) / 4;
        if (nb <= 8) { cfg = 2; nbw = 8; }
    }
    // wide images: 8 x 16-pixel tiles (see conv2_halo_kernel T2D); 1 x 4 waves of 4 pixel tiles, 180 window rows = 3 blocks per wave
    int t2d = 0;
    if (!cfg && NOUT % 128 == 0 && (8 + g->R - 1) * (16 + g->S - 1) <= 192) { cfg = 1; nbw = 3; tm = 4; t2d = 1; }
    if (!cfg) return false;
    if (cfg == 1 && ncc == 1 && nbw > 5) nbw = 8;
    C2Params& p = pl.p;
    p = C2Params{};
    p.SH = SH; p.SW = SW; p.SC = SC; p.NS = g->N;
    p.DH = DH; p.DW = DW; p.NOUT = NOUT;
    p.Wp = Wp; p.Hp = Hp; p.PW = PWH; p.PH = PWH;
    p.stride = 1;
    p.LW = t2d ? 16 + g->S - 1 : Wp;
    for (int t = 0; t < 9; ++t) p.tap_sh[t] = (unsigned)((t / 3) * p.LW + t % 3);
    p.chunk_bytes = 128u;
    magic((unsigned)DW, p.mg_dw, p.sh_dw);
    magic((unsigned)DH, p.mg_dh, p.sh_dh);
    magic((unsigned)p.LW, p.mg_wp, p.sh_wp);
    magic((unsigned)Hp, p.mg_hp, p.sh_hp);
    if (t2d) {
        p.TH = 8; p.TWl = 4;
        p.tiles_x = cs_ceil_div(DW, 16); p.tiles_y = cs_ceil_div(DH, 8);
        const long long nt = (long long)g->N * p.tiles_x * p.tiles_y;
        if (nt >= (1ll << 28)) return false;
        p.n_tiles = (unsigned)nt;
    }
    p.NCC = ncc;
    p.M = (unsigned)M;
    p.src_bytes = (unsigned)src_bytes;
    p.pix_bytes = (unsigned)SC * 2u;
    const unsigned long long wbytes = (unsigned long long)cs_ceil_div(NOUT, 32) * 32 * g->R * g->S * (unsigned long long)SC * 2ull;
    if (wbytes >= 0x80000000ull) return false;
    p.wpk_bytes = (unsigned)wbytes;
    const int BM = cfg != 2 ? 32 * tm : 256, BN = cfg != 2 ? 128 : 64;
    p.n_ntiles = cs_ceil_div(NOUT, BN);
    pl.cfg = cfg; pl.nbw = nbw; pl.ncc = ncc; pl.tm = cfg != 2 ? tm : 4;
    pl.t2d = t2d;
    pl.rows = t2d ? (int)p.n_tiles : cs_ceil_div(M, BM) * (cfg != 2 ? 1 : 2);
    return true;
}
// group count of a ring launch (a multiple of 8: one group per XCD slot): minimises rounds-of-residency x pixel tiles per workgroup;
// 512 workgroups are resident at once.  *cost_out: that product, in pixel tiles (the unit of pick_ring_tm's comparison).
unsigned ring_groups(unsigned n_mt, unsigned n_ntiles, unsigned long long* cost_out) {
    const unsigned cap = (n_mt + 7u) / 8u * 8u;
    unsigned best = 8u;
    unsigned long long best_cost = ~0ull, best_tiles = 0;
    for (unsigned g = 8u; g <= cap; g += 8u) {
        const unsigned long long rounds = ((unsigned long long)g * n_ntiles + 511ull) / 512ull;
        const unsigned long long cost = rounds * ((n_mt + g - 1u) / g) * 16ull + rounds;      // + a little for every extra round's ramp-up
        if (cost <= best_cost) { best_cost = cost; best = g; best_tiles = rounds * ((n_mt + g - 1u) / g); }
    }
    if (cost_out) *cost_out = best_tiles;
    return best;
}

// Pixel-tile height of the 1 x 4 ring configuration (32 * TM pixels per workgroup tile; CELLSEG_RING_TM forces 2 / 3 / 4).  Measured per
// shape (tools/ring_tm_sweep.sh, profiles/round3_notes.md): a deep contraction is a serial chain of NCC steps whose length hardly
// depends on the tile height, so shorter tiles do NOT buy back an under-filled grid (2048 -> 512 @10x10, 200 tiles of 128 px on 512
// resident workgroups: 26.0 us, 400 tiles of 64 px: 32.3 us).  The one case that wins is pick_tm's: all 128-px tiles resident but more
// than one per compute unit somewhere, and the 96-px tiles still all resident (1024 -> 256 @19x19: 28.6 -> 26.0 us).
int pick_ring_tm(long long M, int n_ntiles, int ncc) {
    static const int forced = cs_env_int_("CELLSEG_RING_TM", 0);
    if (forced >= 2 && forced <= 4) return forced;
    (void)ncc;
    const long long wg4 = ((M + 127) / 128) * n_ntiles, wg3 = ((M + 95) / 96) * n_ntiles;
    return (wg4 > 256 && wg4 <= 512 && wg3 <= 512) ? 3 : 4;
}

bool plan_gemm(const CsConvGeom* g, int dgrad, C2Plan& pl) {
    if (g_v2_off) return false;
    if (g->groups > 1) return false;
    if (g->R != 1 || g->S != 1 || g->pad != 0 || g->stride < 1) return false;
    // a STRIDED 1x1 data gradient is served in compact form: dx_compact[n][y][x] = W^T dy[n][y][x] on the P x Q grid (the values of
    // the destination pixels (stride * y, stride * x); every other destination pixel is zero and never materialised -- the consumer
    // adds the compact tensor through cs_conv2d_dgrad_packed's add_stride)
    const bool compact = dgrad && g->stride != 1;
    if (compact && g->stride != 2) return false;
    const int SC = dgrad ? g->K : g->C, NOUT = dgrad ? g->C : g->K;
    const int SH = dgrad ? g->P : g->H, SW = dgrad ? g->Q : g->W;
    const int DH = dgrad ? (compact ? g->P : g->H) : g->P, DW = dgrad ? (compact ? g->Q : g->W) : g->Q;
    if (SC % 64 || NOUT % 64 || DH < 1 || DW < 1) return false;
    if (!dgrad && g->stride > 1 && (DH < 2 || DW < 2)) return false;
    const int ncc = SC / 64;
    // (isolated, tools/conv_microbench.py shows the first-generation kernel ahead on >= 16 chunks -- 24.5 vs 29.1 us on 1024 -> 256 at
    // 19 x 19 -- but inside the training step the ring kernel wins on the family: 3.37 vs 3.43 ms per step; CELLSEG_RING_MAX_NCC
    // declines deeper contractions for A/B runs)
    static const int max_ncc = cs_env_int_("CELLSEG_RING_MAX_NCC", 1 << 20);
    if (ncc > max_ncc) return false;
    const long long M = (long long)g->N * DH * DW;
    const unsigned long long src_bytes = (unsigned long long)g->N * SH * SW * SC * 2ull;
    if (M >= (1ll << 31) - 512 || src_bytes >= 0x80000000ull) return false;
    int cfg = 0;
    if (NOUT % 128 == 0) cfg = 6;                       // ring kernel, 128 px x 128 ch: 1 x 4 waves of 4 tiles
    else if (NOUT == 64) cfg = 7;                       // ring kernel, 128 px x 64 ch: 2 x 2 waves of 2 tiles
    if (!cfg) return false;
    C2Params& p = pl.p;
    p = C2Params{};
    p.SH = SH; p.SW = SW; p.SC = SC; p.NS = g->N;
    p.DH = DH; p.DW = DW; p.NOUT = NOUT;
    p.stride = dgrad ? 1 : g->stride;
    p.add_stride = 1;
    if (DW >= 2 && DH >= 2) {               // (destination pixel -> (n, y, x): the strided forward gather and the strided add operand)
        magic((unsigned)DW, p.mg_dw, p.sh_dw);
        magic((unsigned)DH, p.mg_dh, p.sh_dh);
    }
    p.NCC = ncc;
    p.M = (unsigned)M;
    p.src_bytes = (unsigned)src_bytes;
    p.pix_bytes = (unsigned)SC * 2u;
    const unsigned long long wbytes = (unsigned long long)cs_ceil_div(NOUT, 32) * 32 * (unsigned long long)SC * 2ull;
    if (wbytes >= 0x80000000ull) return false;
    p.wpk_bytes = (unsigned)wbytes;
    const int BN = cfg == 6 ? 128 : 64;
    p.n_ntiles = cs_ceil_div(NOUT, BN);
    pl.cfg = cfg; pl.nbw = 0; pl.ncc = ncc;
    pl.tm = cfg == 6 ? pick_ring_tm(M, p.n_ntiles, ncc) : 2;
    pl.rows = cfg == 6 ? cs_ceil_div(M, 32 * pl.tm) : cs_ceil_div(M, 128) * 2;
    pl.t2d = 0;
    if (cfg == 6 && p.stride == 1 && !compact && ncc % 4 == 0) {      // (two chunks per stage, two stages per loop body)
        const int wtm = pick_wide1_tm(M, p.n_ntiles, ncc);
        if (wtm) {
            pl.cfg = 8; pl.tm = wtm; pl.nbw = wtm;          // 2 planes x 32 * tm rows = 4 * tm blocks of 16 rows, tm per wave
            pl.rows = cs_ceil_div(M, 32 * wtm);
            p.lin = 1; p.chunk_bytes = 256u;
            p.tap_sh[0] = 0u; p.tap_sh[1] = 32u * (unsigned)wtm;
        }
    }
    return true;
}

// dynamic LDS beyond 64 KiB has to be allowed per (device, kernel): cs_api.cpp keeps the table
template <typename F> bool allow_lds(F fn, size_t bytes) {
    return cs_allow_dynamic_lds_(reinterpret_cast<const void*>(fn), bytes, bytes > 81920 ? 163840 : 81920) != 0;
}

template <int TM, int WM, int WN, bool DG>
int launch_ring_t(const C2Params& p, hipStream_t st) {
    // persistent: each workgroup keeps one channel tile and walks every n_groups-th pixel tile.  The group count (a multiple of
    // 8: one group per XCD slot) minimises rounds-of-residency x pixel tiles per workgroup; 512 workgroups are resident at once
    const unsigned n_mt = (unsigned)cs_ceil_div(p.M, 32 * TM * WM);
    const unsigned best = ring_groups(n_mt, (unsigned)p.n_ntiles, nullptr);
    const size_t lds = 3 * 16384 + 4 * (size_t)TM * 2048;
    if constexpr (TM == 2 && !DG) {
        if (p.gather) {
            if (!allow_lds(conv2_ring_kernel<TM, WM, WN, DG, true>, lds)) return CS_ERR_LAUNCH;
            cs_set_variant_("conv2_ring_kernel<2,2,2,false,true>");
            hipLaunchKernelGGL((conv2_ring_kernel<TM, WM, WN, DG, true>), dim3(best * (unsigned)p.n_ntiles), dim3(256), lds, st, p, (int)best);
            CS_LAUNCH_CHECK();
            return CS_OK;
        }
    }
    if (p.gather) {
        cs_set_error_("conv2: gathered operand rows are served for 64 output channels only");
        return CS_ERR_UNSUPPORTED;
    }
    if (!allow_lds(conv2_ring_kernel<TM, WM, WN, DG>, lds)) return CS_ERR_LAUNCH;
    char name[64];
    snprintf(name, sizeof(name), "conv2_ring_kernel<%d,%d,%d,%s,false>", TM, WM, WN, DG ? "true" : "false");
    cs_set_variant_(name);
    hipLaunchKernelGGL((conv2_ring_kernel<TM, WM, WN, DG>), dim3(best * (unsigned)p.n_ntiles), dim3(256), lds, st, p, (int)best);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

template <int TM, int NBW, bool DG, int NTAP = 9> int launch_wide(const C2Params& p, hipStream_t st);

template <bool DG>
int launch_gemm(const C2Plan& pl, hipStream_t st) {
    const C2Params& p = pl.p;
    if (pl.cfg == 8) {
        if (pl.tm == 4) return launch_wide<4, 4, DG, 2>(p, st);
        return launch_wide<6, 6, DG, 2>(p, st);
    }
    if (pl.cfg == 6) {
        if (pl.tm == 2) return launch_ring_t<2, 1, 4, DG>(p, st);
        if (pl.tm == 3) return launch_ring_t<3, 1, 4, DG>(p, st);
        return launch_ring_t<4, 1, 4, DG>(p, st);
    }
    return launch_ring_t<2, 2, 2, DG>(p, st);
}

template <int TM, int NBW, bool DG, bool T2D = false>
int launch_cfg1(const C2Params& p, hipStream_t st, int two_stage) {
    const unsigned n_mt = T2D ? p.n_tiles : (unsigned)cs_ceil_div(p.M, 32 * TM);
    dim3 grid(((n_mt + 7) / 8) * 8 * (unsigned)p.n_ntiles);
    // two stages: the epilogue borrows the idle one; one stage: its exchange scratch sits behind it
    const size_t lds = two_stage ? (size_t)NBW * 8192 * 2 : (size_t)NBW * 8192 + EPI_LDS;
    auto fn = conv2_halo_kernel<9, 3, TM, 1, 4, NBW, DG, T2D>;
    if (!allow_lds(fn, lds)) return CS_ERR_LAUNCH;
    char name[64];
    snprintf(name, sizeof(name), "conv2_halo_kernel<9,3,%d,1,4,%d,%s%s>", TM, NBW, DG ? "true" : "false", T2D ? ",true" : ",false");
    cs_set_variant_(name);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, p);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

template <int TM, int NBW, bool DG, int NTAP>
int launch_wide(const C2Params& p, hipStream_t st) {
    const unsigned n_mt = (unsigned)cs_ceil_div(p.M, 32 * TM);
    dim3 grid(((n_mt + 7) / 8) * 8 * (unsigned)p.n_ntiles);
    const size_t lds = (size_t)NBW * 8192 * (NTAP == 9 ? 2 : 3);      // two stages (1x1 form: three); the epilogue's exchange scratch borrows an idle one
    auto fn = conv2_wide_kernel<TM, NBW, DG, NTAP>;
    if (!allow_lds(fn, lds)) return CS_ERR_LAUNCH;
    char name[64];
    snprintf(name, sizeof(name), "conv2_wide_kernel<%d,%d,%s,%d>", TM, NBW, DG ? "true" : "false", NTAP);
    cs_set_variant_(name);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, p);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

template <bool DG>
int launch_halo(const C2Plan& pl, hipStream_t st) {
    const C2Params& p = pl.p;
    if (pl.cfg == 3) {
        switch (pl.tm * 16 + pl.nbw) {
            case 4 * 16 + 3: return launch_wide<4, 3, DG>(p, st);
            case 4 * 16 + 5: return launch_wide<4, 5, DG>(p, st);
            case 6 * 16 + 5: return launch_wide<6, 5, DG>(p, st);
            case 6 * 16 + 7: return launch_wide<6, 7, DG>(p, st);
            case 8 * 16 + 7: return launch_wide<8, 7, DG>(p, st);
            case 8 * 16 + 9: return launch_wide<8, 9, DG>(p, st);
        }
        cs_set_error_("conv2: no wide-kernel instantiation for this plan");
        return CS_ERR_UNSUPPORTED;
    }
    if (pl.cfg == 2) {
        const unsigned n_mt = (unsigned)cs_ceil_div(p.M, 256);
        dim3 grid(((n_mt + 7) / 8) * 8 * (unsigned)p.n_ntiles);
        cs_set_variant_(DG ? "conv2_halo_kernel<9,3,4,2,2,8,true,false>" : "conv2_halo_kernel<9,3,4,2,2,8,false,false>");
        if (!allow_lds(conv2_halo_kernel<9, 3, 4, 2, 2, 8, DG>, 8 * 8192 + EPI_LDS)) return CS_ERR_LAUNCH;
        hipLaunchKernelGGL((conv2_halo_kernel<9, 3, 4, 2, 2, 8, DG>), grid, dim3(256), 8 * 8192 + EPI_LDS, st, p);
        CS_LAUNCH_CHECK();
        return CS_OK;
    }
    const int two = pl.ncc > 1;
    if (pl.t2d) return launch_cfg1<4, 3, DG, true>(p, st, two);
#define CS_HALO_TM(TM_)                                         \
    switch (pl.nbw) {                                           \
        case 3: return launch_cfg1<TM_, 3, DG>(p, st, two);     \
        case 4: return launch_cfg1<TM_, 4, DG>(p, st, two);     \
        case 5: return launch_cfg1<TM_, 5, DG>(p, st, two);     \
        default: return launch_cfg1<TM_, 8, DG>(p, st, 0);      \
    }
    if (pl.tm == 2) { CS_HALO_TM(2) }
    if (pl.tm == 3) { CS_HALO_TM(3) }
    CS_HALO_TM(4)
#undef CS_HALO_TM
}

#ifdef CS_DEBUG_V2
unsigned long long* g_dbg_buf = nullptr;
#endif

}  // namespace

#ifdef CS_DEBUG_V2
extern "C" int cs_debug_set_stamp_buffer(void* p) { g_dbg_buf = reinterpret_cast<unsigned long long*>(p); return CS_OK; }
#endif

// ---------------------------------------------------------------------------------------------------------------------
static bool plan_any(const CsConvGeom* g, int dgrad, C2Plan& pl) { return plan_halo(g, dgrad, pl) || plan_gemm(g, dgrad, pl); }
static int launch_any(const C2Plan& pl, hipStream_t st, bool dg) {
    if (dg) return pl.cfg <= 3 ? launch_halo<true>(pl, st) : launch_gemm<true>(pl, st);
    return pl.cfg <= 3 ? launch_halo<false>(pl, st) : launch_gemm<false>(pl, st);
}

extern "C" int cs_conv2d_packed_supported(const CsConvGeom* g, int dgrad) {
    if (!g) return 0;
    C2Plan pl;
    return plan_any(g, dgrad, pl) ? 1 : 0;
}

extern "C" size_t cs_conv2d_packed_weight_bytes(const CsConvGeom* g, int dgrad) {
    if (!g) return 0;
    const int rows = dgrad ? g->C : g->K, cols = dgrad ? g->K : g->C;
    return (size_t)cs_ceil_div(rows, 32) * 32 * g->R * g->S * (size_t)cols * 2;
}

extern "C" int cs_pack_conv_weights(const CsConvGeom* g, int dgrad, const void* w_staged, void* w_packed, void* stream) {
    CS_CHECK_ARG(g && w_staged && w_packed, "pack_conv_weights: NULL argument");
    const int rows = dgrad ? g->C : g->K, cols = dgrad ? g->K : g->C;
    CS_CHECK_ARG(cols % 64 == 0 && rows % 8 == 0, "pack_conv_weights: contraction channels must be a multiple of 64");
    const long long total = (long long)cs_ceil_div(rows, 32) * (cols / 64) * g->R * g->S * 4 * 64;
    long long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint4*>(w_staged),
                       reinterpret_cast<uint4*>(w_packed), rows, cols, g->R, g->S, dgrad ? 1 : 0, total);
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_conv2d_packed_partial_rows(const CsConvGeom* g, int dgrad) {
    C2Plan pl;
    if (!g || !plan_any(g, dgrad, pl)) return 0;
    return pl.rows;
}

extern "C" int cs_conv2d_fwd_packed(const CsConvGeom* g, const void* x, const void* w_packed, const float* shift, const void* residual, int act,
                                    void* y, uint8_t* positive_bits, void* stream) {
    CS_CHECK_ARG(g && x && w_packed && y, "conv2d_fwd_packed: NULL tensor");
#ifndef CS_DEBUG_V2
    CS_CHECK_ARG(act == CS_ACT_NONE || act == CS_ACT_RELU, "conv2d_fwd_packed: activation must be none or ReLU");
#endif
    C2Plan pl;
    if (!plan_any(g, 0, pl)) {
        cs_set_error_("conv2d_fwd_packed: geometry not served by the packed-operand kernel (ask cs_conv2d_packed_supported first)");
        return CS_ERR_UNSUPPORTED;
    }
    pl.p.src = x; pl.p.wpk = w_packed; pl.p.dst = y;
    pl.p.shift = shift; pl.p.residual = residual; pl.p.act = act;
    pl.p.bits_out = positive_bits;
#ifdef CS_DEBUG_V2
    pl.p.dbg = g_dbg_buf;
#endif
    return launch_any(pl, reinterpret_cast<hipStream_t>(stream), false);
}

extern "C" int cs_conv2d_dgrad_packed(const CsConvGeom* g, const void* dy, const void* w_packed, const void* add, int add_stride,
                                      const uint8_t* mask_bits, void* dx, float* partial_rows, void* stream) {
    CS_CHECK_ARG(g && dy && w_packed && dx, "conv2d_dgrad_packed: NULL tensor");
    CS_CHECK_ARG(add_stride == 1 || (add_stride == 2 && add), "conv2d_dgrad_packed: add_stride is 1, or 2 with a compact add operand");
    C2Plan pl;
    if (!plan_any(g, 1, pl)) {
        cs_set_error_("conv2d_dgrad_packed: geometry not served by the packed-operand kernel (ask cs_conv2d_packed_supported first)");
        return CS_ERR_UNSUPPORTED;
    }
    if (g->stride != 1)
        CS_CHECK_ARG(!add && !mask_bits && !partial_rows,
                     "conv2d_dgrad_packed: a strided 1x1 data gradient is written in compact [N][P][Q][C] form, without add / mask / column sums");
    if (add_stride == 2) {
        // (the SAME plan as cs_conv2d_packed_partial_rows reported: the caller sized the column-sum rows by it)
        CS_CHECK_ARG(pl.cfg >= 6 && pl.p.DH >= 2 && pl.p.DW >= 2, "conv2d_dgrad_packed: a strided add operand is served by the 1x1 kernels only");
        pl.p.add_stride = 2;
        pl.p.AH = (pl.p.DH + 1) / 2;
        pl.p.AW = (pl.p.DW + 1) / 2;
    }
    pl.p.src = dy; pl.p.wpk = w_packed; pl.p.dst = dx;
    pl.p.residual = add; pl.p.act = CS_ACT_NONE;
    pl.p.bits_in = mask_bits;
    pl.p.slab = partial_rows;
#ifdef CS_DEBUG_V2
    pl.p.dbg = g_dbg_buf;
#endif
    return launch_any(pl, reinterpret_cast<hipStream_t>(stream), true);
}

// Forward of the pixel-paired 7x7 / stride-2 stem (model/resnet.py:171, csrc/conv_igemm.hip cs_stem_*) on the ring kernel: the LDS row of
// an output pixel is GATHERED -- 32 slots of 16 bytes, slot (kh, pw) = pair (x - 2 + pw) of source row (2y - 3 + kh), slots 28..31 zero --
// and multiplied by the packed [K][256] weights (w_pair padded with one zero filter row, cs_pack_conv_weights of the 1x1 geometry
// C = 256).  x_pair: [N][H][(W+1)/2][8] bf16 (cs_stem_pair_input); y: [N][P][Q][K], P = (H - 1) / 2 + 1.
extern "C" int cs_stem_fwd_packed(int N, int H, int W, int K, const void* x_pair, const void* w_packed, const float* shift, int act, void* y,
                                  uint8_t* positive_bits, void* stream) {
    CS_CHECK_ARG(x_pair && w_packed && y && N > 0 && H >= 7 && W >= 7, "stem_fwd_packed: bad arguments");
    CS_CHECK_ARG(K == 64, "stem_fwd_packed: 64 output channels (the stem of every ResNet / ResNeXt of model/resnet.py)");
    CS_CHECK_ARG(act == CS_ACT_NONE || act == CS_ACT_RELU, "stem_fwd_packed: activation must be none or ReLU");
    const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1, Wh = (W + 1) / 2;
    const long long M = (long long)N * P * Q;
    const unsigned long long src_bytes = (unsigned long long)N * H * Wh * 16ull;
    CS_CHECK_ARG(P >= 2 && Q >= 2 && M < (1ll << 31) - 512 && src_bytes < 0x80000000ull, "stem_fwd_packed: extents out of range");
    C2Plan pl;
    C2Params& p = pl.p;
    p = C2Params{};
    p.SH = H; p.SW = Wh; p.SC = 256; p.NS = N;
    p.DH = P; p.DW = Q; p.NOUT = K;
    p.stride = 1; p.add_stride = 1;
    p.gather = 1; p.GH = H; p.GWp = Wh;
    magic((unsigned)Q, p.mg_dw, p.sh_dw);
    magic((unsigned)P, p.mg_dh, p.sh_dh);
    p.NCC = 4;
    p.M = (unsigned)M;
    p.src_bytes = (unsigned)src_bytes;
    p.pix_bytes = 16u;
    p.wpk_bytes = (unsigned)(cs_ceil_div(K, 32) * 32 * 256 * 2);
    pl.cfg = K == 64 ? 7 : 6;
    p.n_ntiles = cs_ceil_div(K, K == 64 ? 64 : 128);
    pl.nbw = 0; pl.ncc = 4; pl.rows = 0;
    p.src = x_pair; p.wpk = w_packed; p.dst = y;
    p.shift = shift; p.act = act; p.bits_out = positive_bits;
    return launch_gemm<false>(pl, reinterpret_cast<hipStream_t>(stream));
}
