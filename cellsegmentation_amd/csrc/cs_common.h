// Shared device/host helpers for libcellseg_hip.so (gfx950 / CDNA4 only).
//
// Storage convention for every activation tensor on the hot path:
//   NHWC, element type T in {float, bf16}, channel count padded to a multiple of one
//   16-byte "chunk" (4 floats or 8 bf16), so every (pixel, channel-chunk) is one aligned
//   16-byte global access.  fp32 is the parity mode (exact-f32 MFMA), bf16 the throughput mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../../include/cellseg_hip.h"

typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kChunk = 4;   // elements per 16-byte chunk
    static constexpr int kDtype = CS_F32;
};
template <> struct Elem<bf16_t> {
    static constexpr int kChunk = 8;
    static constexpr int kDtype = CS_BF16;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) {
    return __uint_as_float(lo16 << 16);
}

// 8 consecutive elements -> 8 floats (p must be 16-byte aligned for bf16, 32 for the pair of float4s
// is NOT required: two independent 16-byte loads).
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // ONE v_cvt_pk_bf16_f32 (RNE, NaN stays NaN).  Written as asm: from two plain casts hipcc (ROCm 7.2) emits the instruction once
    // PER VALUE with a dummy second source and glues the halves with a shift and an SDWA or -- four instructions per pair, 32 per
    // 32 x 32 accumulator tile in every convolution epilogue (seen in the .s of conv_v2.hip, round 4).
    uint32_t u;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(lo), "v"(hi));
    return u;
}
// ReLU of two packed bf16 values: the sign bit of a bf16 is the sign bit of its int16 pattern, so max(x, 0) as signed 16-bit integers
// clears negative values (and -0.0) and keeps positive ones -- the same result as rounding max(x, 0.f), one instruction per pair.
// NaN: a NaN with the sign bit CLEAR passes through (as torch.relu propagates it), one with the sign bit SET becomes 0 -- sign-dependent,
// unlike the fp32 form `v > 0 ? v : 0` (every NaN -> 0) of rounds 1-3.  The arithmetic of this library produces the canonical quiet
// NaN 0x7fc0 (sign clear: v_cvt_pk_bf16_f32 of an fp32 NaN from MFMA / FMA inputs), so a diverging run still shows NaN downstream;
// the batch-statistics accumulators carry their own sticky non-finite flag (ex_add).
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t x) {
    uint32_t u;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(u) : "v"(x));
    return u;
}

template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
}

// 8 consecutive fp32 per-channel parameters (32-byte aligned: channel offsets are multiples of 8) as two 16-byte loads;
// NULL -> fill value.  Scalar p[c] loads inside an 8-wide loop made the BN kernels instruction-bound (0.6 TB/s).
__device__ __forceinline__ void load8p(const float* p, float fill, float (&v)[8]) {
    if (p) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fill;
    }
}

// SiLU / sigmoid with the hardware reciprocal (1 ulp) instead of an IEEE division: `a / b` compiles to v_div_scale x2, v_rcp, four
// FMAs, v_div_fmas, v_div_fixup -- 53 VALU instructions per element made the BatchNorm backward passes of the EfficientNet path
// instruction-bound at 2.7 TB/s (round 3).  The result is rounded to bf16 (or feeds a gradient that is), 2e-7 relative is noise.
__device__ __forceinline__ float sigmoid_fast(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float silu_fast(float u) { return u * sigmoid_fast(u); }

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// ---- exact, order-independent accumulation (round 5: deterministic batch statistics) --------------------------------------------
// The per-channel sums of a train-mode BatchNorm (sum z / sum z^2, sum g / sum g*xhat) are added up from hundreds of workgroups.  With
// plain floating-point atomics the result depends on the order the workgroups arrive in: two runs of the same step gave different
// bits.  Here every contribution t (|t| < 2^40) is split into three LIMBS that are multiples of 2^0, 2^-40 and
// 2^-80 and smaller than 2^40 times their unit: l2 = trunc(t), l1 = trunc((t - l2) 2^40) 2^-40, l0 = rint((t - l2 - l1) 2^80) 2^-80.
// Each limb is accumulated with an fp64 atomic add of its own: up to 2^12 contributors keep every partial sum of a limb a multiple
// of its unit below 2^53 units, i.e. EXACTLY representable -- every addition is exact, so the total is the same in any order, and no
// bit of a contribution at or above 2^-80 is lost.  (A first version with three int64 limbs was as exact but cost the consumers
// -- every thread of the apply passes reads 16 totals -- 48 emulated int64 -> fp64 conversions: C4 3.35 -> 3.06 k tiles/s.)
// Accumulator of a C-channel tensor: cs_bn_accum_words(C) = 6 C + 1 zero-initialised 8-byte words,
//   word (2 * limb + which) * C + c   limb 0..2 (units 2^-80, 2^-40, 2^0) of sum `which` (0: first sum, 1: second sum) of channel c
//   word 6 C                          sticky flag: a contribution was NaN / infinite / >= 2^40 -> every total reads as NaN
constexpr int kExLimbs = 3;
__host__ __device__ inline long long ex_words(int C) { return 6LL * C + 1; }
__device__ __forceinline__ void ex_add(void* acc, int C, int which, int c, double t) {
    double* w = reinterpret_cast<double*>(acc);
    if (!(fabs(t) < 0x1p40)) {                      // NaN, infinity or out of range
        atomicOr(reinterpret_cast<unsigned long long*>(w + 6LL * C), 1ull);
        return;
    }
    const double l2 = trunc(t);
    const double r1 = t - l2;                       // exact: the fraction of t
    const double l1 = trunc(r1 * 0x1p40) * 0x1p-40;
    const double r0 = r1 - l1;                      // exact: the bits below 2^-40
    const double l0 = rint(r0 * 0x1p80) * 0x1p-80;
    double* p = w + (long long)which * C + c;
    if (l0 != 0.0) atomicAdd(p, l0);
    if (l1 != 0.0) atomicAdd(p + 2LL * C, l1);
    if (l2 != 0.0) atomicAdd(p + 4LL * C, l2);
}
__device__ __forceinline__ double ex_read(const void* acc, int C, int which, int c) {
    const double* w = reinterpret_cast<const double*>(acc);
    const double* p = w + (long long)which * C + c;
    const double v = p[4LL * C] + (p[2LL * C] + p[0]);
    return reinterpret_cast<const unsigned long long*>(w)[6LL * C] ? __longlong_as_double(0x7ff8000000000000LL) : v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- host side error plumbing (cs_api.cpp owns the storage) ----
extern "C" void cs_set_error_(const char* msg);
extern "C" void cs_set_variant_(const char* name);
// wgrad_v2.hip: split-K slabs of the second-generation 3x3 weight gradient (0 = geometry not served) and its launch
int cs_wgrad2_splits_(const CsConvGeom* g, int dtype, int n_items);
int cs_wgrad2_launch_(const CsConvGeom* g, int dtype, const void* const* x_tab, const void* const* dy_tab, float* const* dw_tab, int n_items,
                      void* stream);
#define CS_CHECK_ARG(cond, msg)                         \
    do {                                                \
        if (!(cond)) {                                  \
            cs_set_error_(msg);                         \
            return CS_ERR_INVALID_ARG;                  \
        }                                               \
    } while (0)
#define CS_LAUNCH_CHECK()                               \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) {                        \
            cs_set_error_(hipGetErrorString(e__));      \
            return CS_ERR_LAUNCH;                       \
        }                                               \
    } while (0)

static inline int cs_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Dynamic LDS beyond 64 KiB has to be allowed per kernel AND per device (hipFuncSetAttribute applies to the device that is current
// at the call).  cs_api.cpp keeps the (device, function) table under a mutex; `limit` is the byte count to allow (<= 160 KiB).
extern "C" int cs_allow_dynamic_lds_(const void* fn, size_t bytes, size_t limit);
// compute units of the current device (cs_api.cpp; 256 when the query fails)
extern "C" int cs_device_cus_(void);

// A/B knobs.  The production library (`make`) has none: every knob is its default, nothing reads the environment and
// cs_set_igemm_path is not exported.  `make AB=1` (-DCS_AB_SWITCHES -> libcellseg_hip_ab.so, loaded by the forced-mode tests and
// the tools/ sweeps through CELLSEG_LIB_FLAVOUR=ab) reads them: a positive integer, anything else (unset, 0, negative,
// non-numeric) -> `dflt`.
#ifdef CS_AB_SWITCHES
static inline int cs_env_int_(const char* name, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || v <= 0 || v > (1 << 24)) return dflt;
    return (int)v;
}
#else
static inline int cs_env_int_(const char*, int dflt) { return dflt; }
#endif
// "<NAME>=<positive integer>" set (A/B flavour only)
static inline bool cs_env_flag_(const char* name) { return cs_env_int_(name, 0) != 0; }
