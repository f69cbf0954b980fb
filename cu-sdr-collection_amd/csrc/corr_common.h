// corr_common.h — pieces shared by the generic and the fast E/P/L correlator kernels.
#pragma once
#include "gc_internal.h"

namespace gcorr {

constexpr int kWG = 256;
constexpr int kSPL = 8;  // samples per lane-chunk

enum Mode { I8_IQ = 0, I8_QI, I16_IQ, I16_QI, I8_REAL, I16_REAL };

constexpr int kInlineBlocks = 16;

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

struct KArgs {
  const uint8_t* if_base;
  const gc_block* blocks;
  const DevChannel* chans;
  double* out;      // [nblocks][GC_OUT_STRIDE] when splits == 1
  double* partial;  // [nblocks][splits][GC_OUT_STRIDE] when splits > 1
  double fs;
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
  int wide;      // fast kernel variant with 4 waves per workgroup and int8-pair LDS tables
  int share_el;  // every block has el_spacing*R*M == 1/2: early and late taps share their step mask
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

__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ unsigned int rl_u(unsigned int v, int lane) {
  return (unsigned int)__builtin_amdgcn_readlane((int)v, lane);
}


}  // namespace gcorr

// corr_fast.hip
int gc_launch_correlator_fast(gc_context* ctx, const gcorr::KArgs& a, const gcorr::InlineBlocks& ib, unsigned int grid,
                              int max_arms, bool spl16);
