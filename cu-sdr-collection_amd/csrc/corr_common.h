// corr_common.h — pieces shared by the generic and the fast E/P/L correlator kernels.
#pragma once
#include <type_traits>

#include "gc_internal.h"

// NOTE: the correlator translation units are compiled with -ffp-contract=off (build.py explains why): write
// wanted FMAs as fmaf() / fma().

namespace gcorr {

constexpr int kWG = 256;
constexpr int kSPL = 8;  // samples per lane-chunk

enum Mode { I8_IQ = 0, I8_QI, I16_IQ, I16_QI, I8_REAL, I16_REAL };

constexpr int kInlineBlocks = 16;
constexpr int kGuard = 16;  // zero entries on both sides of the generic kernel's staged f16 tables

template <int ARMS>
struct ArmPitch {  // f16 values per interleaved table entry
  static constexpr int v = ARMS == 1 ? 1 : ARMS == 2 ? 2 : 4;
};
inline int gc_arm_pitch(int arms) { return arms <= 1 ? 1 : arms == 2 ? 2 : 4; }

constexpr int kLaneWaves = 16;  // wavefronts per workgroup of the lane kernel (corr_lane.hip)

// Every kLaneReseedSteps steps of a lane (64 samples each) the lane kernel re-seeds its carrier phasor from the exact float64
// phase and takes the rounding drift out of its 32.32 ramp (the ramp's 64-sample increment is rounded to 2^-32 chip; what
// kLaneReseedSteps of them miss is a block-uniform integer, added back in one instruction).
constexpr int kLaneReseedSteps = 128;

// Near-tie window of the lane kernel in 2^-32-chip units: 16 ulp of the largest ramp value (the reference's
// float64 rounding of a + i*d and its two-sided colon), one unit per ramp step of a lane since the last drift correction
// (the 32.32 increment is rounded to 2^-33 chip) plus one per correction (each is rounded too), and 2 units of slack.
// Host (gc_mark_tie_free) and device use the same formula.
__host__ __device__ inline unsigned int gc_tie_window_units(double max_ramp, int lane_steps) {
  const int drift = lane_steps <= kLaneReseedSteps ? lane_steps : kLaneReseedSteps + lane_steps / kLaneReseedSteps + 1;
  return 2u + (unsigned int)drift + (unsigned int)(max_ramp * (16.0 * 2.220446049250313e-16 * 4294967296.0));
}

// The same for a derived arm (corr_lane.hip): its position is the base ramp's fraction times m6 (6), so the base ramp's rounding
// (the 2 + drift units above) counts m6 times, in units of 2^-32 of ITS entries; the float64 term is already in those units.
__host__ __device__ inline unsigned int gc_tie_window_units6(double max_ramp, int lane_steps, double m6) {
  const int drift = lane_steps <= kLaneReseedSteps ? lane_steps : kLaneReseedSteps + lane_steps / kLaneReseedSteps + 1;
  return (unsigned int)(m6 * (double)(4 + drift)) + 8u + (unsigned int)(max_ramp * (16.0 * 2.220446049250313e-16 * 4294967296.0));
}

struct TaggedSlot {
  double value;
  unsigned int tag;
  unsigned int zero;
};

// Second kernel argument of the fast kernel: never named in device code (it is read through the
// kernel-argument segment pointer with scalar loads), only copied there by the launch.
struct InlineBlocks {
  gc_block b[kInlineBlocks];
};

struct DevLoopArgs;

struct KArgs {
  const uint8_t* if_base;
  const gc_block* blocks;
  const DevChannel* chans;
  double* out;      // [nblocks][GC_OUT_STRIDE] when splits == 1
  double* partial;  // [nblocks][splits][GC_OUT_STRIDE] when splits > 1
  double fs;
  double inv_fs;  // 1 / fs
  int64_t nblocks;
  int splits;
  int xcd_swizzle;
  int red_off;  // byte offset of the reduction scratch in dynamic LDS
  int bpw;      // blocks per workgroup (fast kernel)
  int stride;   // descriptor stride between consecutive blocks of one workgroup
  // Closed loop (fast kernel only): results go to host-mapped TAGGED slots — one 16-byte record
  // {double value, uint32 tag, uint32 0} per output, written by ONE store instruction per workgroup — so
  // the host polls the tags instead of paying a stream-synchronise wake-up, and no fence / atomic /
  // counter is needed on the device.  nullptr = off.
  TaggedSlot* tagged;       // [nblocks][splits][GC_OUT_STRIDE]
  unsigned int notify_tag;
  // Up to kInlineBlocks descriptors travel in the kernel-argument segment (no PCIe read of the
  // host-mapped descriptor buffer at the start of every workgroup).
  int use_inline;
  const struct DevLoopArgs* devloop;  // device-side loop closure (devloop.h); nullptr otherwise
  int wide;      // fast kernel variant with 4 waves per workgroup and int8-pair LDS tables
  long long total_wg;  // xcd_swizzle outside the device loop: workgroups of the launch that have work (the grid is that rounded up to 8)
  int share_el;  // every block has el_spacing*R*M == 1/2: early and late taps share one ramp
  int derived;   // lane kernel: three-arm channels whose third arm is derived from the second (DevChannel::derived)
  int rho_off;   // lane kernel: byte offset in dynamic LDS of the waves' carrier-step tables (kLaneReseedSteps float2 per wave)
};

// t = k0 - G / 2^64 ;  ceil(t + x) for x = xi + xf/2^64  is  k0 + xi + (xf > G)
struct Fx {
  int k0;
  unsigned long long G;
};

__device__ __forceinline__ unsigned long long frac_to_u64(double g) {
  // g in [0,1) -> floor(g * 2^64), exact for doubles with <= 64 fractional bits
  const double gh = g * 4294967296.0;
  const unsigned int hi = (unsigned int)gh;
  const double gl = (gh - (double)hi) * 4294967296.0;
  const unsigned int lo = (unsigned int)gl;
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ Fx to_fx(double t) {
  const double kf = ceil(t);
  Fx r;
  r.k0 = (int)kf;
  r.G = frac_to_u64(kf - t);
  return r;
}

template <int MODE>
__device__ __forceinline__ void load_chunk(const uint8_t* __restrict__ base, long long q,
                                           float (&a)[kSPL], float (&b)[kSPL]) {
  if constexpr (MODE == I8_IQ || MODE == I8_QI) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + 16 * q);
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < kSPL; ++j) {
      const unsigned int word = w[j >> 1];
      const int sh = (j & 1) * 16;
      const float x0 = (float)(int)(signed char)(word >> sh);
      const float x1 = (float)(int)(signed char)(word >> (sh + 8));
      a[j] = (MODE == I8_IQ) ? x0 : x1;
      b[j] = (MODE == I8_IQ) ? x1 : x0;
    }
  } else if constexpr (MODE == I16_IQ || MODE == I16_QI) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(base + 32 * q);
    const uint4 v1 = *reinterpret_cast<const uint4*>(base + 32 * q + 16);
    const unsigned int w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < kSPL; ++j) {
      const float x0 = (float)(int)(short)(w[j] & 0xffffu);
      const float x1 = (float)(int)(short)(w[j] >> 16);
      a[j] = (MODE == I16_IQ) ? x0 : x1;
      b[j] = (MODE == I16_IQ) ? x1 : x0;
    }
  } else if constexpr (MODE == I8_REAL) {
    const uint2 v = *reinterpret_cast<const uint2*>(base + 8 * q);
    const unsigned int w[2] = {v.x, v.y};
#pragma unroll
    for (int j = 0; j < kSPL; ++j) {
      a[j] = (float)(int)(signed char)(w[j >> 2] >> ((j & 3) * 8));
      b[j] = 0.0f;
    }
  } else {
    const uint4 v = *reinterpret_cast<const uint4*>(base + 16 * q);
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < kSPL; ++j) {
      a[j] = (float)(int)(short)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
      b[j] = 0.0f;
    }
  }
}

template <int MODE>
struct Fmt {
  static constexpr int bps = (MODE == I8_IQ || MODE == I8_QI || MODE == I16_REAL) ? 2 : (MODE == I8_REAL) ? 1 : 4;
  static constexpr bool swap = (MODE == I8_QI || MODE == I16_QI);
};

template <int MODE, int SPL>
__device__ __forceinline__ void load_words(const uint8_t* __restrict__ base, long long q,
                                           unsigned int (&w)[SPL * Fmt<MODE>::bps / 4]) {
  constexpr int NW = SPL * Fmt<MODE>::bps / 4;
  const uint8_t* p = base + (long long)(SPL * Fmt<MODE>::bps) * q;
  if constexpr (NW == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    w[0] = v.x; w[1] = v.y;
  } else {
#pragma unroll
    for (int k = 0; k < NW / 4; ++k) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * k);
      w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
  }
}

// Zero the samples of an edge chunk that lie outside [0, N): sample j is valid iff 0 <= i0+j < N.
template <int MODE, int SPL>
__device__ __forceinline__ void mask_words(unsigned int (&w)[SPL * Fmt<MODE>::bps / 4], int i0, int N) {
  constexpr int bits = 8 * Fmt<MODE>::bps;
  constexpr int per_word = 32 / bits;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    const bool valid = (unsigned int)(i0 + j) < (unsigned int)N;
    const unsigned int m = (bits == 32) ? 0xffffffffu : (((1u << bits) - 1u) << ((j % per_word) * bits));
    if (!valid) w[j / per_word] &= ~m;
  }
}

#define GC_CVT_SDWA(sel)                                                                                  \
  {                                                                                                       \
    float r;                                                                                              \
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel : "=v"(r) : "v"(word)); \
    return r;                                                                                             \
  }
template <int B>
__device__ __forceinline__ float cvt_byte(unsigned int word) {
  if constexpr (B == 0) GC_CVT_SDWA("BYTE_0")
  else if constexpr (B == 1) GC_CVT_SDWA("BYTE_1")
  else if constexpr (B == 2) GC_CVT_SDWA("BYTE_2")
  else GC_CVT_SDWA("BYTE_3")
}
template <int H>
__device__ __forceinline__ float cvt_half(unsigned int word) {
  if constexpr (H == 0) GC_CVT_SDWA("WORD_0")
  else GC_CVT_SDWA("WORD_1")
}

// Sample J of the chunk as floats (a, b) = (I, Q) after the layout's swap; b = 0 for real data.
template <int MODE, int J, int NW>
__device__ __forceinline__ void sample_ab(const unsigned int (&w)[NW], float& a, float& b) {
  float x0, x1;
  if constexpr (MODE == I8_IQ || MODE == I8_QI) {
    x0 = cvt_byte<(J & 1) * 2>(w[J >> 1]);
    x1 = cvt_byte<(J & 1) * 2 + 1>(w[J >> 1]);
  } else if constexpr (MODE == I16_IQ || MODE == I16_QI) {
    x0 = cvt_half<0>(w[J]);
    x1 = cvt_half<1>(w[J]);
  } else if constexpr (MODE == I8_REAL) {
    x0 = cvt_byte<J & 3>(w[J >> 2]);
    x1 = 0.0f;
  } else {
    x0 = cvt_half<J & 1>(w[J >> 1]);
    x1 = 0.0f;
  }
  a = Fmt<MODE>::swap ? x1 : x0;
  b = Fmt<MODE>::swap ? x0 : x1;
}

template <int J, int SPL, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < SPL) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, SPL>(f);
  }
}

// Wavefront sum with DPP row shifts / broadcasts (no LDS traffic); the total lands in lane 63.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value,
                                                      decltype(row_mask)::value, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
  v += dpp(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
  v += dpp(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
  v += dpp(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31
  return v;
}

// Derived BOC(6,1) arm of the lane kernel: 1 = its f32 table image carries a third column, the base arm times (-1)^entry (16-byte
// entries, one VALU instruction per tap less); 0 = 8-byte entries {arm 0, arm 1} and the entry's parity added to the sign word
// (half the LDS bytes per tap).  Host image (gc_sync_channels) and kernel must agree.
#ifndef GC_LANE_PN
#define GC_LANE_PN 1
#endif

// K wavefront sums at once, transposing as they go: on return the lane with wave_transpose_slot(lane) == c (c < K) holds the
// 64-lane total of v[c].  A step over one lane bit pairs the values (2i, 2i + 1): the lanes with the bit clear keep value 2i and
// add their partner's copy of it, the lanes with the bit set do the same for value 2i + 1 - one add per PAIR where K separate
// trees take one per value and step, and no readlane / select to bring the totals to "their" lanes afterwards.  The first two steps
// (most pairs) go over lane bits 2 and 3: masked row shifts, bank_mask picks the lanes whose partner lies above / below, so the two
// masked adds write disjoint lanes of one register - two instructions per pair and no select.  Then bits 0 and 1 (quad_perm, two
// selects + one add per pair), bit 4 (ds_swizzle), bit 5 (ds_bpermute).  K = 18 (three arms): ~45 cross-lane and select
// instructions against 108 DPP adds + 36 readlanes + 36 selects.  Every lane of the wave must be active.
__device__ __forceinline__ int wave_transpose_slot(int lane) {  // bits taken in the order 2, 3, 0, 1, 4
  return ((lane >> 2) & 3) | ((lane & 3) << 2) | (lane & 16);
}
namespace wts {
template <int CTRL>
__device__ __forceinline__ float quad(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// x on the lanes whose mask bit is clear, y on the others; the mask is a wave-uniform constant (an SGPR pair, no compare per select)
__device__ __forceinline__ float pick(float x, float y, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(mask));
  return r;
}
// x + partner's x on the lanes whose bit is clear, y + partner's y on the lanes whose bit is set (S = 4 or 8 lanes apart, same row);
// s_nop: the two wait states a DPP read wants after the VALU write of its source (the assembler cannot see into the block)
template <int S>
__device__ __forceinline__ float row_pair(float x, float y) {
  float r;
  if constexpr (S == 4)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(r) : "v"(x), "v"(y));
  else
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(r) : "v"(x), "v"(y));
  return r;
}
}  // namespace wts

template <int K>
__device__ __forceinline__ float wave_transpose_sum(const float (&v)[K], int lane) {
  static_assert(K >= 1 && K <= 32, "one value per lane of half a wave at most");
  constexpr int N1 = (K + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
  constexpr unsigned long long kBit0 = 0xaaaaaaaaaaaaaaaaull, kBit1 = 0xccccccccccccccccull, kBit4 = 0xffff0000ffff0000ull;
  float a1[N1], a2[N2], a3[N3], a4[N4];
#pragma unroll
  for (int i = 0; i < N1; ++i) a1[i] = wts::row_pair<4>(v[2 * i], v[2 * i + 1 < K ? 2 * i + 1 : 2 * i]);     // lane bit 2
#pragma unroll
  for (int i = 0; i < N2; ++i) a2[i] = wts::row_pair<8>(a1[2 * i], a1[2 * i + 1 < N1 ? 2 * i + 1 : 2 * i]);  // lane bit 3
#pragma unroll
  for (int i = 0; i < N3; ++i) {  // lane bit 0: quad_perm [1, 0, 3, 2]
    if (2 * i + 1 < N2)
      a3[i] = wts::pick(a2[2 * i], a2[2 * i + 1], kBit0) + wts::quad<0xB1>(wts::pick(a2[2 * i + 1], a2[2 * i], kBit0));
    else
      a3[i] = a2[2 * i] + wts::quad<0xB1>(a2[2 * i]);
  }
#pragma unroll
  for (int i = 0; i < N4; ++i) {  // lane bit 1: quad_perm [2, 3, 0, 1]
    if (2 * i + 1 < N3)
      a4[i] = wts::pick(a3[2 * i], a3[2 * i + 1], kBit1) + wts::quad<0x4E>(wts::pick(a3[2 * i + 1], a3[2 * i], kBit1));
    else
      a4[i] = a3[2 * i] + wts::quad<0x4E>(a3[2 * i]);
  }
  float r;  // lane bit 4: the other half of the 32-lane group (ds_swizzle, bit mode: and 0x1f, or 0, xor 0x10)
  if constexpr (N4 == 2)
    r = wts::pick(a4[0], a4[1], kBit4) + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(wts::pick(a4[1], a4[0], kBit4)), 0x401F));
  else
    r = a4[0] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a4[0]), 0x401F));
  return r + __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(r)));  // lane bit 5
}

__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ unsigned int rl_u(unsigned int v, int lane) {
  return (unsigned int)__builtin_amdgcn_readlane((int)v, lane);
}


// Descriptor fetch: the device/host-mapped list, or (closed loop, <= kInlineBlocks blocks) the copy that
// travels in the kernel-argument segment.  The segment is read through its constant-address-space
// pointer so the loads stay scalar (s_load) and the per-block quantities stay in SGPRs; indexing p.inl as
// a by-value array, or going through a generic pointer, drags everything into VGPRs (measured: 179 VGPRs,
// -20 % throughput).
constexpr size_t kInlineOffset = (sizeof(KArgs) + 7) / 8 * 8;  // second explicit kernel argument

__device__ __forceinline__ gc_block load_block(const KArgs& p, long long lb) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (p.use_inline) {
    typedef const __attribute__((address_space(4))) char* cptr4;
    typedef const __attribute__((address_space(4))) unsigned long long* qptr4;
    qptr4 src = (qptr4)((cptr4)__builtin_amdgcn_kernarg_segment_ptr() + kInlineOffset + lb * sizeof(gc_block));
    union {
      gc_block b;
      unsigned long long q[sizeof(gc_block) / 8];
    } u;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(gc_block) / 8); ++i) u.q[i] = src[i];
    return u.b;
  }
#endif
  return p.blocks[lb];
}


}  // namespace gcorr

// corr_lane.hip
int gc_launch_devloop_lane(gc_context* ctx, const gcorr::KArgs& a, unsigned int grid, int max_arms, bool share_el, int waves);
int gc_launch_correlator_lane(gc_context* ctx, const gcorr::KArgs& a, const gcorr::InlineBlocks& ib, unsigned int grid,
                              int max_arms, bool share_el);
// corr_multi.hip
int gc_multi_waves(const gc_context* ctx, int max_arms, long long nblocks, int period, int kt, bool share_el);
int gc_launch_correlator_multi(gc_context* ctx, const gcorr::KArgs& a, unsigned int grid, int max_arms, int kt, bool share_el, int waves);
// corr_cboc.hip
int gc_cboc_waves(const gc_context* ctx);
// the hybrid kernel takes a periodic replay list of `nblocks` blocks (channel pattern period `period`) of the scope just validated:
// every channel a three-arm channel with a derived six-fold arm, base ramp with <= 2 transitions per 16-sample chunk (scope_kt6), int8
// I/Q or Q/I record, tables + 8 KB of running sums per wave fit a CU, and the launch at least two rounds (of waves x CUs epochs), at
// least two thirds full
bool gc_cboc_takes(const gc_context* ctx, long long nblocks, int period);
int gc_launch_correlator_cboc(gc_context* ctx, const gcorr::KArgs& a, unsigned int grid, int waves);
// corr_fast.hip
bool gc_fast_prefers_wide();  // compiled with the prefix-sum variant
int gc_launch_devloop(gc_context* ctx, const gcorr::KArgs& a, unsigned int grid, bool spl16, bool share_el);
int gc_launch_correlator_fast(gc_context* ctx, const gcorr::KArgs& a, const gcorr::InlineBlocks& ib, unsigned int grid,
                              int max_arms, bool spl16);
