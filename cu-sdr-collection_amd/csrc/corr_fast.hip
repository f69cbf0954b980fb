// corr_fast.hip — fast path of the E/P/L correlator for "low-rate" replicas: the table index
// advances by less than one entry over a lane-chunk of 8 samples (7*step*R*M < 1: GPS L1 C/A 17.6
// samples/chip, GLONASS 23.5, B1I / E1 BOC(1,1) / B1C BOC(1,1) 8.8, L2C 7.8).  Same arithmetic
// contract as corr_kernel.hip (tracking.m:247-300), ~3x fewer VALU instructions per sample:
//
//   * a lane-chunk sees at most ONE table transition per tap, so the replica over the chunk is
//     c1 + dc*step(j - u): the six sums become c1*T + dc*S_x with T = sum_j y_j shared by all taps
//     and arms and S_x = sum_j step(j - u_x)*y_j; step() is ONE full-rate VALU op
//     (v_sub_f32 ... clamp), no compare / select / LDS gather per sample;
//   * LDS holds {c[k], c[k+1]-c[k]} as float2, one ds_read_b64 per tap and arm per chunk;
//   * int8 samples are converted with v_cvt_f32_ubyteN after xor 0x80 and the +128 offset is
//     folded into the addend of the carrier FMAs (no extra instruction);
//   * the carrier base rotation is applied Horner-style to the accumulators (acc = acc*conj(rho) + U)
//     and once more at the end with the exact per-thread phase;
//   * the transition position u = g/(step*R*M) is a float quotient; chunks where any u is within
//     4e-6 of an integer (a sample within ~1e-7 chip of a chip edge, where float32 and the reference's
//     float64 rounding could disagree) take the exact double-precision path.
#include "corr_common.h"

using namespace gcorr;

namespace {

constexpr float kBig = 8388608.0f;  // 2^23: clamp(kBig*(j-u)) is exactly 0 or 1 outside the tie band
constexpr float kTieTol = 4e-6f;

template <int MODE>
__device__ __forceinline__ void load_words(const uint8_t* __restrict__ base, long long q, unsigned int (&w)[8]) {
  if constexpr (MODE == I8_IQ || MODE == I8_QI || MODE == I16_REAL) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + 16 * q);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else if constexpr (MODE == I16_IQ || MODE == I16_QI) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(base + 32 * q);
    const uint4 v1 = *reinterpret_cast<const uint4*>(base + 32 * q + 16);
    w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
  } else {
    const uint2 v = *reinterpret_cast<const uint2*>(base + 8 * q);
    w[0] = v.x; w[1] = v.y;
  }
}

// Zero the samples of an edge chunk that lie outside [0, N): sample j is valid iff 0 <= i0+j < N.
template <int MODE>
__device__ __forceinline__ void mask_words(unsigned int (&w)[8], int i0, int N) {
  constexpr int bits = (MODE == I8_IQ || MODE == I8_QI || MODE == I16_REAL) ? 16 : (MODE == I8_REAL) ? 8 : 32;
  constexpr int per_word = 32 / bits;
#pragma unroll
  for (int j = 0; j < kSPL; ++j) {
    const bool valid = (unsigned int)(i0 + j) < (unsigned int)N;
    const unsigned int m = (bits == 32) ? 0xffffffffu : (((1u << bits) - 1u) << ((j % per_word) * bits));
    if (!valid) w[j / per_word] &= ~m;
  }
}

// Sample j of the chunk as floats (a, b) = (first, second) component in file order, offset by
// OFS (128 for the xor-0x80 int8 path, 0 otherwise).
template <int MODE>
__device__ __forceinline__ void sample_ab(const unsigned int (&w)[8], int j, float& a, float& b) {
  if constexpr (MODE == I8_IQ || MODE == I8_QI) {
    const unsigned int word = w[j >> 1] ^ 0x80808080u;
    const int sh = (j & 1) * 16;
    a = (float)((word >> sh) & 0xffu);        // v_cvt_f32_ubyteN
    b = (float)((word >> (sh + 8)) & 0xffu);
  } else if constexpr (MODE == I16_IQ || MODE == I16_QI) {
    a = (float)(int)(short)(w[j] & 0xffffu);
    b = (float)(int)(short)(w[j] >> 16);
  } else if constexpr (MODE == I8_REAL) {
    const unsigned int word = w[j >> 2] ^ 0x80808080u;
    a = (float)((word >> ((j & 3) * 8)) & 0xffu);
    b = 128.0f;
  } else {
    a = (float)(int)(short)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
    b = 0.0f;
  }
}

template <int ARMS, int MODE>
__global__ __launch_bounds__(kWG) void corr_epl_fast_kernel(const KArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool kSwap = (MODE == I8_QI || MODE == I16_QI);
  constexpr float kOfs = (MODE == I8_IQ || MODE == I8_QI || MODE == I8_REAL) ? 128.0f : 0.0f;

  long long wg = blockIdx.x;
  if (p.xcd_swizzle) {
    const long long per = (long long)gridDim.x >> 3;
    wg = (wg & 7) * per + (wg >> 3);
  }
  const long long lb = wg / p.splits;
  const int split = (int)(wg - lb * p.splits);
  const gc_block blk = p.blocks[lb];
  const DevChannel* __restrict__ chn = p.chans + blk.channel;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int arms_here = chn->arms;

  // ---- stage {c[k], c[k+1]-c[k]} for k = -1 .. nent (c[-1] := c[0], c[>=nent] := 0) ---------------
  float2* tab2[ARMS];
#pragma unroll
  for (int a = 0; a < ARMS; ++a) {
    const int aa = (a < arms_here) ? a : 0;
    tab2[a] = reinterpret_cast<float2*>(smem + 8 * (size_t)chn->lds_off[aa]);
    if (a < arms_here) {
      const int off = blk.table_offset[a];
      const int n = min(chn->stage_len[a], chn->nent[a] - off);
      const int8_t* __restrict__ src = chn->tab[a] + off;
      for (int i = tid; i < n + 3; i += kWG) {
        const int k = i - 1;  // table index of .x
        const float c0 = (k < 0) ? (float)src[0] : (k < n) ? (float)src[k] : 0.0f;
        const float c1 = (k + 1 < n) ? (float)src[k + 1] : 0.0f;
        tab2[a][i] = make_float2(c0, c1 - c0);
      }
    }
  }
  float* red = reinterpret_cast<float*>(smem + p.red_off);
  __syncthreads();

  // ---- per-block uniform quantities (see corr_kernel.hip for the reference line citations) --------
  const double R = chn->index_scale;
  const double M = chn->mult[0];
  const double rem = blk.rem_code_phase;
  const double step = blk.code_phase_step;
  const double d = blk.el_spacing;
  const int N = blk.blksize;
  const long long s0 = blk.first_sample;
  const double aE = (rem - d) * R;
  const double aL = (rem + d) * R;
  const double aP = rem * R;
  const double sp = step * R;
  const double tau = blk.carr_freq / p.fs;
  const double bP = __dmul_rn(__dadd_rn(__dmul_rn((double)(N - 1), step), rem), R);
  const double bE = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(N - 1), step), rem), -d), R);
  const double bL = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(N - 1), step), rem), d), R);
  const float uk = (float)(1.0 / (sp * M) * 2.3283064365386963e-10);  // g_hi (2^-32 units) -> u
  const float ukB = uk * kBig;

  // lanes 0..7: delta^j = exp(-i*2*pi*j*tau) with the +128 offset terms; lane 8: chunk stride
  float myC, myS;
  unsigned int myJlo, myJhi;
  int myJint;
  {
    const int j = (lane < 8) ? lane : kSPL * kWG;
    const double x = (double)j * tau;
    double sn, cs;
    sincospi(2.0 * (x - floor(x)), &sn, &cs);
    myC = (float)cs;
    myS = (float)sn;
    const double y = (double)j * (sp * M);
    const double yi = floor(y);
    const unsigned long long jf = frac_to_u64(y - yi);
    myJint = (int)yi;
    myJlo = (unsigned int)jf;
    myJhi = (unsigned int)(jf >> 32);
  }
  float C[kSPL], S[kSPL], KR[kSPL], KI[kSPL], KJ[kSPL];
#pragma unroll
  for (int j = 0; j < kSPL; ++j) {
    C[j] = rl_f(myC, j);
    S[j] = rl_f(myS, j);
    // y = (a + i b)(C - i S) with (a, b) = (ua - 128, ub - 128):  constants of the folded offset
    KR[j] = -kOfs * (C[j] + S[j]);
    KI[j] = -kOfs * (C[j] - S[j]);
    KJ[j] = (float)j * kBig;
  }
  const float rotC = rl_f(myC, 8), rotS = rl_f(myS, 8);
  const unsigned long long Df = ((unsigned long long)rl_u(myJhi, 8) << 32) | rl_u(myJlo, 8);
  const int Di = __builtin_amdgcn_readlane(myJint, 8);

  const long long q0 = s0 >> 3;
  const long long q1 = (s0 + N - 1) >> 3;
  const int nchunks = (int)(q1 - q0 + 1);
  const int cps = (nchunks + p.splits - 1) / p.splits;
  const int cbeg = split * cps;
  const int cend = min(nchunks, cbeg + cps);

  float accr[ARMS][3], acci[ARMS][3];
#pragma unroll
  for (int a = 0; a < ARMS; ++a)
#pragma unroll
    for (int x = 0; x < 3; ++x) accr[a][x] = acci[a][x] = 0.0f;

  int c = cbeg + tid;
  float wc = 1.0f, ws = 0.0f;
  if (c < cend) {
    int i0 = (int)((q0 + c) * kSPL - s0);
    Fx fx[3];
    const double isp = __dmul_rn((double)i0, sp);
    fx[0] = to_fx(__dmul_rn(__dadd_rn(aE, isp), M));
    fx[1] = to_fx(__dmul_rn(__dadd_rn(aP, isp), M));
    fx[2] = to_fx(__dmul_rn(__dadd_rn(aL, isp), M));
    const uint8_t* __restrict__ base = p.if_base;

    for (; c < cend; c += kWG) {
      unsigned int w[8];
      load_words<MODE>(base, q0 + c, w);
      if ((i0 < 0) | (i0 + kSPL > N)) mask_words<MODE>(w, i0, N);

      // transition positions and the near-tie filter
      float u[3], uB[3];
      bool suspect = false;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const float gh = (float)(unsigned int)(fx[x].G >> 32);
        u[x] = gh * uk;
        uB[x] = gh * ukB;
        suspect |= fabsf(u[x] - rintf(u[x])) < kTieTol;
      }

      float Ur[ARMS][3], Ui[ARMS][3];
      if (__any(suspect)) {
        // ---- exact path (~1e-4 of wave-chunks): float64 index per sample, as the reference ------
#pragma unroll
        for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
          for (int x = 0; x < 3; ++x) Ur[ar][x] = Ui[ar][x] = 0.0f;
#pragma unroll
        for (int j = 0; j < kSPL; ++j) {
          float a, b;
          sample_ab<MODE>(w, j, a, b);
          if (kSwap) { const float t = a; a = b; b = t; }
          const float yr = fmaf(a, C[j], fmaf(b, S[j], KR[j]));
          const float yi = fmaf(b, C[j], fmaf(a, -S[j], KI[j]));
          const int i = i0 + j;
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            const double ax = (x == 0) ? aE : (x == 1) ? aP : aL;
            const double bx = (x == 0) ? bE : (x == 1) ? bP : bL;
            double t;
            if (2 * i < N - 1)
              t = __dadd_rn(ax, __dmul_rn((double)i, sp));
            else if (2 * i > N - 1)
              t = __dadd_rn(bx, -__dmul_rn((double)(N - 1 - i), sp));
            else
              t = __dadd_rn(ax, bx) / 2.0;
            int k = (int)ceil(__dmul_rn(t, M));
            k = max(-1, min(k, 0x3fffffff));
#pragma unroll
            for (int ar = 0; ar < ARMS; ++ar) {
              const int kk = min(k, chn->stage_len[(ar < arms_here) ? ar : 0] + 1);
              const float cf = tab2[ar][kk + 1].x;
              Ur[ar][x] = fmaf(cf, yr, Ur[ar][x]);
              Ui[ar][x] = fmaf(cf, yi, Ui[ar][x]);
            }
          }
        }
      } else {
        // ---- fast path ------------------------------------------------------------------------------
        float Tr = 0.f, Ti = 0.f, Sr[3] = {0.f, 0.f, 0.f}, Si[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kSPL; ++j) {
          float a, b;
          sample_ab<MODE>(w, j, a, b);
          if (kSwap) { const float t = a; a = b; b = t; }
          const float yr = fmaf(a, C[j], fmaf(b, S[j], KR[j]));
          const float yi = fmaf(b, C[j], fmaf(a, -S[j], KI[j]));
          Tr += yr;
          Ti += yi;
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            const float s = __builtin_amdgcn_fmed3f(KJ[j] - uB[x], 0.0f, 1.0f);  // clamp: 1 iff j > u
            Sr[x] = fmaf(s, yr, Sr[x]);
            Si[x] = fmaf(s, yi, Si[x]);
          }
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) {
#pragma unroll
          for (int ar = 0; ar < ARMS; ++ar) {
            const float2 cd = tab2[ar][fx[x].k0 + 1];
            Ur[ar][x] = fmaf(cd.x, Tr, cd.y * Sr[x]);
            Ui[ar][x] = fmaf(cd.x, Ti, cd.y * Si[x]);
          }
        }
      }
      // Horner step: acc = acc * conj(rho) + U, rho = delta^2048 = rotC - i rotS
#pragma unroll
      for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const float nr = fmaf(accr[ar][x], rotC, fmaf(-acci[ar][x], rotS, Ur[ar][x]));
          const float ni = fmaf(accr[ar][x], rotS, fmaf(acci[ar][x], rotC, Ui[ar][x]));
          accr[ar][x] = nr;
          acci[ar][x] = ni;
        }
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const unsigned long long g = fx[x].G;
        fx[x].k0 += Di + (g < Df ? 1 : 0);
        fx[x].G = g - Df;
      }
      i0 += kSPL * kWG;
    }
    // exact carrier phase at the first sample of this thread's LAST chunk
    const int i_last = i0 - kSPL * kWG;
    const double ph = blk.rem_carr_phase * 0.15915494309189535 + (double)i_last * tau;
    sincospif(2.0f * (float)(ph - floor(ph)), &ws, &wc);
  }

  // ---- rotate into the absolute frame, reduce across the wave and the 4 waves -------------------------
  const int wave = tid >> 6;
#pragma unroll
  for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      float vr = wc * accr[ar][x] + ws * acci[ar][x];
      float vi = wc * acci[ar][x] - ws * accr[ar][x];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        vr += __shfl_down(vr, off, 64);
        vi += __shfl_down(vi, off, 64);
      }
      if (lane == 0) {
        red[(wave * ARMS + ar) * 6 + 2 * x] = vr;
        red[(wave * ARMS + ar) * 6 + 2 * x + 1] = vi;
      }
    }
  __syncthreads();
  if (tid < ARMS * 6) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < kWG / 64; ++wv) s += (double)red[wv * ARMS * 6 + tid];
    if (tid >= arms_here * 6) s = 0.0;
    if (p.splits == 1)
      p.out[lb * GC_OUT_STRIDE + tid] = s;
    else
      p.partial[(lb * p.splits + split) * GC_OUT_STRIDE + tid] = s;
  }
}

template <int ARMS>
int launch_fast_mode(gc_context* ctx, const KArgs& a, dim3 grid, size_t smem) {
  int mode;
  if (ctx->if_dtype == GC_I8)
    mode = ctx->if_layout == GC_IQ ? I8_IQ : ctx->if_layout == GC_QI ? I8_QI : I8_REAL;
  else
    mode = ctx->if_layout == GC_IQ ? I16_IQ : ctx->if_layout == GC_QI ? I16_QI : I16_REAL;
  switch (mode) {
    case I8_IQ: hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, I8_IQ>), grid, dim3(kWG), smem, ctx->stream, a); break;
    case I8_QI: hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, I8_QI>), grid, dim3(kWG), smem, ctx->stream, a); break;
    case I16_IQ: hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, I16_IQ>), grid, dim3(kWG), smem, ctx->stream, a); break;
    case I16_QI: hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, I16_QI>), grid, dim3(kWG), smem, ctx->stream, a); break;
    case I8_REAL: hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, I8_REAL>), grid, dim3(kWG), smem, ctx->stream, a); break;
    default: hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, I16_REAL>), grid, dim3(kWG), smem, ctx->stream, a); break;
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace

int gc_launch_correlator_fast(gc_context* ctx, const KArgs& a, unsigned int grid, int max_arms) {
  // float2 tables: 8 bytes per staged entry (+3 pad entries per arm) — lds_off is in entries here
  const size_t smem = (size_t)a.red_off + kWG / 64 * GC_OUT_STRIDE * sizeof(float);
  switch (max_arms) {
    case 1: return launch_fast_mode<1>(ctx, a, dim3(grid), smem);
    case 2: return launch_fast_mode<2>(ctx, a, dim3(grid), smem);
    default: return launch_fast_mode<3>(ctx, a, dim3(grid), smem);
  }
}
