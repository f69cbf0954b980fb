// corr_kernel.hip — Early/Prompt/Late integrate-and-dump correlator for gfx950 (MI355X), generic variant:
// any chipping rate (GPS L5, BDS B2a/B3I, Galileo E5 at 10.23 Mcps take several table transitions per
// 8-sample lane-chunk, which rules out the transition-mask kernel of corr_fast.hip).
//
// Replaces the vector expressions of GPS/GPS_L1CA/include/tracking.m:247-300 (and the R-scaled /
// multi-arm variants GAL_E1C/include/tracking.m:236-303, GPS_L5C/include/tracking.m:255-326):
//   T2  code-replica index ramps   tcode = a : step : b ; idx = ceil(tcode)+1      (:252-270)
//   T3  carrier replica            exp(-1i*((carrFreq*2*pi)*(n/fs) + remCarrPhase)) (:280-287)
//   T4  mix + six sums per arm                                                      (:291-300)
//
// Design (wave64, 256-thread workgroups, no MFMA — elementwise multiply + reduce, VALU-issue bound):
//   * raw int8/int16 IF samples are read straight from HBM as 16-byte vectors: one lane-chunk
//     = 8 consecutive samples, chunk grid aligned to absolute sample index so every load is
//     16-B aligned and fully coalesced (1 KiB per wave-instruction), next chunk prefetched;
//   * the padded code tables of all arms are staged once per workgroup in LDS as INTERLEAVED f16
//     ({arm0, arm1} per entry): one ds_read per tap and sample serves every arm, and the f16 value
//     feeds v_fma_mix_f32 directly — no int->float conversion in the loop.  16 zero guard entries on
//     both sides let the masked samples of a block's first/last chunk index without clamps;
//   * code phase is a 64-bit fixed-point fraction per lane (exact double-precision base per thread,
//     2^-64-chip increments).  The per-sample edge decision is ONE v_sub_co_u32 on the high words
//     (borrow -> table index via v_subb), and the same difference feeds a running unsigned min/max:
//     only if some difference lies within a few 2^-32 chip of zero can the high-word decision (or
//     the reference's float64 rounding, fl(a + fl(i*d)) and MATLAB's two-sided colon) disagree, and
//     then the whole wave-chunk is redone by the exact float64 per-sample path;
//   * carrier: per-block table delta^j = exp(-i*2*pi*j*f/fs), j = 0..7, held in SGPRs, an exact
//     double-precision phase base per thread reduced to one turn before the float sincos, and a
//     per-iteration rotation by delta^2048;
//   * 6*ARMS float accumulators per lane, wavefront shuffle reduction, LDS cross-wave combine in
//     double, one store per output.
#include <cstdlib>

#include "corr_common.h"

using namespace gcorr;

namespace {

template <int ARMS, int MODE>
__global__ __launch_bounds__(kWG) void corr_epl_kernel(const KArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int AP = ArmPitch<ARMS>::v;  // f16 values per staged entry
  constexpr int NW = kSPL * Fmt<MODE>::bps / 4;
  constexpr bool kReal = (MODE == I8_REAL || MODE == I16_REAL);
  _Float16* tabh = reinterpret_cast<_Float16*>(smem);  // [kGuard + maxn + kGuard][AP]

  // ---- which block / which split -------------------------------------------------------
  long long wg = blockIdx.x;
  if (p.xcd_swizzle) {
    // Workgroup b is dispatched to XCD b % 8.  Give every XCD one contiguous range of the
    // descriptor list so that neighbouring descriptors (the channels of one epoch, which read
    // the same IF window) share an L2.
    const long long total = gridDim.x;
    const long long per = total >> 3;  // host guarantees total % 8 == 0 when swizzling
    wg = (wg & 7) * per + (wg >> 3);
  }
  // With bpw > 1 (periodic replay lists) a workgroup walks bpw consecutive epochs of ONE channel and
  // re-stages the tables only when channel or table offsets change: for 10 230-chip codes the two
  // staged tables (41 KB) are more bytes than one block's IF samples (36 KB).
  const long long wq = wg / p.splits;
  const int split = (int)(wg - wq * p.splits);
  const long long grp = wq / p.stride;
  const int cslot = (int)(wq - grp * p.stride);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  int staged_channel = -1;
  int staged_off[ARMS];
#pragma unroll
  for (int a = 0; a < ARMS; ++a) staged_off[a] = -1;

  for (int bi = 0; bi < p.bpw; ++bi) {
  const long long lb = (grp * p.bpw + bi) * p.stride + cslot;
  if (lb >= p.nblocks) break;
  const gc_block blk = p.blocks[lb];
  const DevChannel* __restrict__ chn = p.chans + blk.channel;

  // ---- stage the code tables into LDS --------------------------------------------------
  int nent[ARMS];
  int maxn = 0;
  const int arms_here = chn->arms;
  bool restage = blk.channel != staged_channel;
  bool plain = chn->tabh != nullptr && chn->tabh_ap == AP;  // pre-interleaved copy usable as is
#pragma unroll
  for (int a = 0; a < ARMS; ++a) {
    nent[a] = 0;
    if (a < arms_here) {
      const int off = blk.table_offset[a];
      nent[a] = min(chn->stage_len[a], chn->nent[a] - off);
      maxn = max(maxn, nent[a]);
      restage |= off != staged_off[a];
      plain &= off == 0 && chn->stage_len[a] == chn->nent[a];
    }
  }
  __syncthreads();  // the previous block's table reads and reduction scratch are done
  if (restage) {
    if (plain) {
      const uint4* __restrict__ src = reinterpret_cast<const uint4*>(chn->tabh);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      const int n16 = chn->tabh_bytes >> 4;
#pragma unroll 4
      for (int i = tid; i < n16; i += kWG) dst[i] = src[i];
    } else {
      const int total = maxn + 2 * kGuard;
      for (int i = tid; i < total; i += kWG) {
        const int e = i - kGuard;
#pragma unroll
        for (int a = 0; a < AP; ++a) {
          float v = 0.0f;
          if (a < ARMS && a < arms_here && e >= 0 && e < nent[a < ARMS ? a : 0])
            v = (float)chn->tab[a < ARMS ? a : 0][blk.table_offset[a < ARMS ? a : 0] + e];
          tabh[i * AP + a] = (_Float16)v;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < ARMS; ++a) staged_off[a] = (a < arms_here) ? blk.table_offset[a] : -1;
  }
  staged_channel = blk.channel;
  float* red = reinterpret_cast<float*>(smem + p.red_off);
  __syncthreads();

  // ---- per-block uniform quantities ----------------------------------------------------
  const double R = chn->index_scale;
  const double M = chn->mult[0];
  const double rem = blk.rem_code_phase;
  const double step = blk.code_phase_step;
  const double d = blk.el_spacing;
  const int N = blk.blksize;
  const long long s0 = blk.first_sample;
  // colon() arguments exactly as the reference writes them (tracking.m:252-268;
  // GAL_E1C tracking.m:236-262 for R = 2); x*1.0 is exact so R = 1 needs no special case.
  const double aE = (rem - d) * R;
  const double aL = (rem + d) * R;
  const double aP = rem * R;
  const double sp = step * R;
  const double tau = blk.carr_freq / p.fs;  // carrier turns per sample
  // colon() end points b = ((N-1)*step + rem -/+ d) * R, evaluated in the reference's order
  const double bP = __dmul_rn(__dadd_rn(__dmul_rn((double)(N - 1), step), rem), R);
  const double bE = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(N - 1), step), rem), -d), R);
  const double bL = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(N - 1), step), rem), d), R);
  // near-tie window in 2^-32 chip units: 16 ulp of the largest ramp value, at least 2 units
  const unsigned int tie_e =
      2u + (unsigned int)((fabs(aE) + fabs(aL) + (double)N * fabs(sp) + 1.0) * fabs(M) * (16.0 * 2.220446049250313e-16 * 4294967296.0));
  // the masked samples of an edge chunk index at most 7 ramp steps outside the table: inside the guard?
  const bool guard_ok = 7.0 * fabs(sp * M) + 2.0 < (double)kGuard;

  // lanes 0..7: delta^j and the fixed-point ramp increments j*sp*M; lane 8: chunk-stride terms
  float myC, myS;
  unsigned int myJlo, myJhi;
  int myJint;
  {
    const int j = (lane < 8) ? lane : kSPL * kWG;
    const double x = (double)j * tau;
    const double fr = x - floor(x);
    sincospif(2.0f * (float)fr, &myS, &myC);  // range reduction in double above, sincos in float
    const double y = (double)j * (sp * M);
    const double yi = floor(y);
    const unsigned long long jf = frac_to_u64(y - yi);
    myJint = (int)yi;
    myJlo = (unsigned int)jf;
    myJhi = (unsigned int)(jf >> 32);
  }
  float C[kSPL], S[kSPL];
  unsigned int Jfh[kSPL];
  int Ji[kSPL];
#pragma unroll
  for (int j = 0; j < kSPL; ++j) {
    C[j] = rl_f(myC, j);
    S[j] = rl_f(myS, j);
    Jfh[j] = rl_u(myJhi, j);
    Ji[j] = __builtin_amdgcn_readlane(myJint, j);
  }
  const float rotC = rl_f(myC, 8), rotS = rl_f(myS, 8);
  const unsigned long long Df = ((unsigned long long)rl_u(myJhi, 8) << 32) | rl_u(myJlo, 8);
  const int Di = __builtin_amdgcn_readlane(myJint, 8);

  // ---- chunk range of this split -------------------------------------------------------
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
  if (c < cend) {
    int i0 = (int)((q0 + c) * kSPL - s0);  // block-relative index of the chunk's first sample
    // exact double-precision bases (the reference's a + k*d, then *M for a BOC(6,1)-only channel)
    Fx fx[3];
    const double isp = __dmul_rn((double)i0, sp);
    fx[0] = to_fx(__dmul_rn(__dadd_rn(aE, isp), M));
    fx[1] = to_fx(__dmul_rn(__dadd_rn(aP, isp), M));
    fx[2] = to_fx(__dmul_rn(__dadd_rn(aL, isp), M));
    float wc, ws;
    {
      const double ph = blk.rem_carr_phase * 0.15915494309189535 + (double)i0 * tau;
      const float f = (float)(ph - floor(ph));
      sincospif(2.0f * f, &ws, &wc);
    }
    const uint8_t* __restrict__ base = p.if_base;
    unsigned int w[NW];
    load_words<MODE, kSPL>(base, q0 + c, w);

    while (true) {
      const int cn = c + kWG;
      unsigned int wn[NW];
      if (cn < cend) load_words<MODE, kSPL>(base, q0 + cn, wn);

      const bool edge = (i0 < 0) | (i0 + kSPL > N);
      float sr[ARMS][3], si[ARMS][3];
#pragma unroll
      for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
        for (int x = 0; x < 3; ++x) sr[ar][x] = si[ar][x] = 0.0f;

      bool slow = !guard_ok && __any(edge) != 0;
      if (!slow) {
        // ---- lean path ---------------------------------------------------------------------
        if (__builtin_expect(edge, 0)) mask_words<MODE, kSPL>(w, i0, N);
        unsigned int gh[3], kb[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          gh[x] = (unsigned int)(fx[x].G >> 32);
          kb[x] = (unsigned int)fx[x].k0;
        }
        unsigned int dmin = 0xffffffffu, dmax = 0u;
        static_for<0, kSPL>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          float a, b;
          sample_ab<MODE, j, NW>(w, a, b);
          const float yr = kReal ? a * C[j] : fmaf(a, C[j], b * S[j]);
          const float yi = kReal ? -a * S[j] : fmaf(b, C[j], -a * S[j]);
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            // ceil(t + j*sp*M) = k0 + Ji[j] + (Jf[j] > G), decided on the high words (which differ unless the
            // chunk is flagged below): the borrow of G_hi - Jf_hi[j] IS that bit, added to k0 by v_addc.
            // (The C form compiles to cndmask + shifts + add3: 7 instructions instead of 2.)
            unsigned int df, t;
            asm("v_subrev_co_u32_e32 %0, vcc, %2, %3\n\tv_addc_co_u32_e32 %1, vcc, 0, %4, vcc"
                : "=&v"(df), "=v"(t)
                : "s"(Jfh[j]), "v"(gh[x]), "v"(kb[x])
                : "vcc");
            const unsigned int k = t + (unsigned int)Ji[j];
            dmin = min(dmin, df);
            dmax = max(dmax, df);
            if constexpr (AP == 1) {
              const float cf = (float)tabh[kGuard + (int)k];
              sr[0][x] = fmaf(cf, yr, sr[0][x]);
              si[0][x] = fmaf(cf, yi, si[0][x]);
            } else if constexpr (AP == 2) {
              typedef _Float16 h2 __attribute__((ext_vector_type(2)));
              const h2 e = reinterpret_cast<const h2*>(tabh)[kGuard + (int)k];
#pragma unroll
              for (int ar = 0; ar < ARMS; ++ar) {
                const float cf = (float)e[ar];
                sr[ar][x] = fmaf(cf, yr, sr[ar][x]);
                si[ar][x] = fmaf(cf, yi, si[ar][x]);
              }
            } else {
              typedef _Float16 h4 __attribute__((ext_vector_type(4)));
              const h4 e = reinterpret_cast<const h4*>(tabh)[kGuard + (int)k];
#pragma unroll
              for (int ar = 0; ar < ARMS; ++ar) {
                const float cf = (float)e[ar];
                sr[ar][x] = fmaf(cf, yr, sr[ar][x]);
                si[ar][x] = fmaf(cf, yi, si[ar][x]);
              }
            }
          }
        });
        // Near-tie test.  Sample j of tap x lies within e chips of a table edge iff (Jf[j] - G_x) mod 2^64 is
        // within e*2^64 of zero.  That is not measure-zero: with remCodePhase = 0 and the nominal code rate
        // (every channel's first block, tracking.m:163-165) 1.023e6/18e6 is rational and samples 3000k land
        // exactly on edges.
        const bool suspect = (dmin <= tie_e) | (dmax >= 0u - tie_e);
        slow = __any(suspect) != 0;
      }
      if (slow) {
        // ---- exact path: MATLAB colon element i (tracking.m:252-270) in float64 for every sample of the
        // chunk — forwards from a for the first half, backwards from the end point b for the second, mean
        // of both in the exact middle.  A rolled loop that re-reads the samples from memory (L1-resident)
        // so that the rare path does not set the kernel's register budget.
#pragma unroll
        for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
          for (int x = 0; x < 3; ++x) sr[ar][x] = si[ar][x] = 0.0f;
        const uint8_t* sp8 = base + (long long)(kSPL * Fmt<MODE>::bps) * (q0 + c);
        float cr = 1.0f, ci = 0.0f;  // delta^j = cr - i*ci
#pragma unroll 1
        for (int j = 0; j < kSPL; ++j) {
          const int i = i0 + j;
          float x0, x1 = 0.0f;
          if constexpr (Fmt<MODE>::bps == 2 && !kReal) {
            x0 = (float)(signed char)sp8[2 * j];
            x1 = (float)(signed char)sp8[2 * j + 1];
          } else if constexpr (Fmt<MODE>::bps == 4) {
            x0 = (float)reinterpret_cast<const short*>(sp8)[2 * j];
            x1 = (float)reinterpret_cast<const short*>(sp8)[2 * j + 1];
          } else if constexpr (Fmt<MODE>::bps == 1) {
            x0 = (float)(signed char)sp8[j];
          } else {
            x0 = (float)reinterpret_cast<const short*>(sp8)[j];
          }
          float a = Fmt<MODE>::swap ? x1 : x0, b = Fmt<MODE>::swap ? x0 : x1;
          if ((unsigned int)i >= (unsigned int)N) a = b = 0.0f;  // edge chunk
          const float yr = a * cr + b * ci;
          const float yi = b * cr - a * ci;
          const float ncr = cr * C[1] - ci * S[1], nci = cr * S[1] + ci * C[1];
          cr = ncr;
          ci = nci;
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
            const double kc = fmin(fmax(ceil(__dmul_rn(t, M)), (double)-kGuard), (double)(maxn + kGuard - 1));
            const int kk = (int)kc;  // out-of-table indices belong to masked samples only
#pragma unroll
            for (int ar = 0; ar < ARMS; ++ar) {
              const float cf = (float)tabh[(kGuard + kk) * AP + ar];
              sr[ar][x] = fmaf(cf, yr, sr[ar][x]);
              si[ar][x] = fmaf(cf, yi, si[ar][x]);
            }
          }
        }
      }
      // rotate the chunk sums from the lane frame by w = exp(-i*theta0) and accumulate
#pragma unroll
      for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          accr[ar][x] += wc * sr[ar][x] + ws * si[ar][x];
          acci[ar][x] += wc * si[ar][x] - ws * sr[ar][x];
        }
      if (cn >= cend) break;
      // advance this thread by kWG chunks
      const float nwc = wc * rotC - ws * rotS;
      const float nws = wc * rotS + ws * rotC;
      wc = nwc;
      ws = nws;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const unsigned long long g = fx[x].G;
        fx[x].k0 += Di + (g < Df ? 1 : 0);
        fx[x].G = g - Df;
      }
      i0 += kSPL * kWG;
      c = cn;
#pragma unroll
      for (int k = 0; k < NW; ++k) w[k] = wn[k];
    }
  }

  // ---- reduce: wavefront shuffles, then LDS across the 4 waves in double ----------------
  const int wave = tid >> 6;
#pragma unroll
  for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      float vr = accr[ar][x], vi = acci[ar][x];
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
    for (int w = 0; w < kWG / 64; ++w) s += (double)red[w * ARMS * 6 + tid];
    if (tid >= arms_here * 6) s = 0.0;
    if (p.splits == 1)
      p.out[lb * GC_OUT_STRIDE + tid] = s;
    else
      p.partial[(lb * p.splits + split) * GC_OUT_STRIDE + tid] = s;
  }
  }  // bpw loop
}

// ---- exact reference kernel for channels whose arms use DIFFERENT ramp multipliers ---------------------
// (BDS B1C wide-band: data BOC(1,1), pilot BOC(1,1) and pilot BOC(6,1) read through ceil(6*t),
// BDS/B1C/include/WB_tracking.m:285-317; the 122 762-entry BOC(6,1) table does not fit LDS next to the
// other two).  One thread per sample (grid-stride), every index from the reference's float64 colon element
// rule, tables read through L2: simple and exact rather than fast — this is one signal of twelve.
template <int MODE>
__global__ __launch_bounds__(kWG) void corr_epl_mixed_kernel(const KArgs p) {
  __shared__ double red[kWG / 64][GC_OUT_STRIDE];
  const long long lb = blockIdx.x / p.splits;
  const int split = (int)(blockIdx.x - lb * p.splits);
  const gc_block blk = p.blocks[lb];
  const DevChannel* __restrict__ chn = p.chans + blk.channel;
  const int arms = chn->arms;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double R = chn->index_scale, rem = blk.rem_code_phase, step = blk.code_phase_step, d = blk.el_spacing;
  const int N = blk.blksize;
  const double sp = step * R;
  const double a3[3] = {(rem - d) * R, rem * R, (rem + d) * R};
  const double nm1s = __dmul_rn((double)(N - 1), step);
  const double b3[3] = {__dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), -d), R), __dmul_rn(__dadd_rn(nm1s, rem), R),
                        __dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), d), R)};
  const double tau = blk.carr_freq / p.fs;
  const double ph0 = blk.rem_carr_phase * 0.15915494309189535;
  const int per = (N + p.splits - 1) / p.splits;
  const int i_beg = split * per, i_end = min(N, i_beg + per);
  float acc[GC_OUT_STRIDE];
#pragma unroll
  for (int v = 0; v < GC_OUT_STRIDE; ++v) acc[v] = 0.f;
  constexpr int bps = (MODE == I8_IQ || MODE == I8_QI || MODE == I16_REAL) ? 2 : (MODE == I8_REAL) ? 1 : 4;
  for (int i = i_beg + tid; i < i_end; i += kWG) {
    const uint8_t* s = p.if_base + (size_t)(blk.first_sample + i) * bps;
    float a, b;
    if (MODE == I8_IQ || MODE == I8_QI) {
      a = (float)(signed char)s[0];
      b = (float)(signed char)s[1];
    } else if (MODE == I16_IQ || MODE == I16_QI) {
      a = (float)((const short*)s)[0];
      b = (float)((const short*)s)[1];
    } else if (MODE == I8_REAL) {
      a = (float)(signed char)s[0];
      b = 0.f;
    } else {
      a = (float)((const short*)s)[0];
      b = 0.f;
    }
    if (MODE == I8_QI || MODE == I16_QI) {
      const float t = a;
      a = b;
      b = t;
    }
    const double ph = ph0 + (double)i * tau;
    float sn, cs;
    sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
    const float xr = a * cs + b * sn, xi = b * cs - a * sn;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double t;
      if (2 * i < N - 1)
        t = __dadd_rn(a3[x], __dmul_rn((double)i, sp));
      else if (2 * i > N - 1)
        t = __dadd_rn(b3[x], -__dmul_rn((double)(N - 1 - i), sp));
      else
        t = __dadd_rn(a3[x], b3[x]) / 2.0;
      for (int ar = 0; ar < arms; ++ar) {
        const int k = (int)ceil(__dmul_rn(t, chn->mult[ar])) + blk.table_offset[ar];
        const float c = (float)chn->tab[ar][min(max(k, 0), chn->nent[ar] - 1)];
        acc[ar * 6 + 2 * x] = fmaf(c, xr, acc[ar * 6 + 2 * x]);
        acc[ar * 6 + 2 * x + 1] = fmaf(c, xi, acc[ar * 6 + 2 * x + 1]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < GC_OUT_STRIDE; ++v) {
    float x = acc[v];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) red[wave][v] = (double)x;
  }
  __syncthreads();
  if (tid < GC_OUT_STRIDE) {
    double sum = 0.0;
    for (int w = 0; w < kWG / 64; ++w) sum += red[w][tid];
    if (p.splits == 1)
      p.out[lb * GC_OUT_STRIDE + tid] = sum;
    else
      p.partial[(lb * p.splits + split) * GC_OUT_STRIDE + tid] = sum;
  }
}

__global__ void combine_partials_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                        long long nblocks, int splits) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks * GC_OUT_STRIDE) return;
  const long long lb = i / GC_OUT_STRIDE;
  const int v = (int)(i - lb * GC_OUT_STRIDE);
  double s = 0.0;
  for (int k = 0; k < splits; ++k) s += partial[(lb * splits + k) * GC_OUT_STRIDE + v];
  out[i] = s;
}

template <typename K>
void launch_generic(gc_context* ctx, K kernel, const KArgs& a, dim3 grid, size_t smem) {
  if (smem > 64 * 1024)  // above the default dynamic-LDS limit (gfx950 has 160 KiB per workgroup)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kernel, grid, dim3(kWG), smem, ctx->stream, a);
}

template <int ARMS>
int launch_mode(gc_context* ctx, KArgs a, dim3 grid) {
  // interleaved f16 tables with guards, then the cross-wave reduction scratch
  const size_t tab_bytes = ((size_t)(ctx->max_stage_len + 2 * kGuard) * ArmPitch<ARMS>::v * 2 + 15) / 16 * 16;
  const size_t smem = tab_bytes + kWG / 64 * GC_OUT_STRIDE * sizeof(float);
  if (smem > 160 * 1024) {
    gc_set_error("code tables need %zu bytes of LDS (> 160 KiB); set a window with gc_set_code_window", smem);
    return GC_E_UNSUPPORTED;
  }
  a.red_off = (int)tab_bytes;
  int mode;
  if (ctx->if_dtype == GC_I8)
    mode = ctx->if_layout == GC_IQ ? I8_IQ : ctx->if_layout == GC_QI ? I8_QI : I8_REAL;
  else
    mode = ctx->if_layout == GC_IQ ? I16_IQ : ctx->if_layout == GC_QI ? I16_QI : I16_REAL;
  switch (mode) {
    case I8_IQ: launch_generic(ctx, corr_epl_kernel<ARMS, I8_IQ>, a, grid, smem); break;
    case I8_QI: launch_generic(ctx, corr_epl_kernel<ARMS, I8_QI>, a, grid, smem); break;
    case I16_IQ: launch_generic(ctx, corr_epl_kernel<ARMS, I16_IQ>, a, grid, smem); break;
    case I16_QI: launch_generic(ctx, corr_epl_kernel<ARMS, I16_QI>, a, grid, smem); break;
    case I8_REAL: launch_generic(ctx, corr_epl_kernel<ARMS, I8_REAL>, a, grid, smem); break;
    default: launch_generic(ctx, corr_epl_kernel<ARMS, I16_REAL>, a, grid, smem); break;
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace

int gc_launch_correlator(gc_context* ctx, const gc_block* d_blocks, int64_t nblocks, int splits,
                         double* d_out, double* d_partial, int max_arms, int fast, int period, unsigned int notify_tag,
                         bool share_el) {
  if (nblocks <= 0) return GC_OK;
  KArgs a;
  a.if_base = ctx->d_if;
  a.blocks = d_blocks;
  a.chans = ctx->d_channels;
  a.out = d_out;
  a.partial = d_partial;
  a.fs = ctx->fs;
  a.nblocks = nblocks;
  a.splits = splits;
  a.red_off = ctx->max_lds_bytes;
  a.bpw = 1;
  a.stride = 1;
  InlineBlocks ib;
  a.tagged = nullptr;
  a.notify_tag = 0;
  a.use_inline = 0;
  a.share_el = share_el ? 1 : 0;
  a.wide = 0;
  if (fast && notify_tag != 0 && ctx->h_tagged_pinned) {
    // closed loop: d_blocks is the host-mapped descriptor buffer (readable by the host right here)
    a.tagged = reinterpret_cast<TaggedSlot*>(ctx->h_tagged_pinned);
    a.notify_tag = notify_tag;
    if (nblocks <= kInlineBlocks) {
      a.use_inline = 1;
      for (int64_t i = 0; i < nblocks; ++i) ib.b[i] = d_blocks[i];
    }
  }
  long long total = (long long)nblocks * splits;
  int want_bpw = 8;
  if (const char* e = std::getenv("GC_REPLAY_BPW")) want_bpw = std::max(1, std::atoi(e));
  const bool wide_tables = fast > 0 && gc_fast_table_mode(ctx) == 1;
  const bool big_list = nblocks >= 64 * (long long)period * ctx->compute_units;
  if (want_bpw > 1 && fast >= 0 && splits == 1 && period > 0 && (big_list || wide_tables || fast == 0)) {
    // periodic list (all table offsets zero): a workgroup stages its channel's table once and walks
    // several consecutive epochs of that channel — 8 for big lists; the WIDE variant needs at least one
    // block per wave, so 4 even for short lists
    a.bpw = big_list ? std::max(want_bpw, wide_tables ? 4 : 1) : (fast == 0 ? std::min(want_bpw, 8) : 4);
    a.stride = period;
    total = ((nblocks + (long long)a.bpw * period - 1) / ((long long)a.bpw * period)) * period;
  }
  a.xcd_swizzle = (fast >= 0 && total % 8 == 0 && total >= 64) ? 1 : 0;
  if (total > 0x7fffffffLL) {
    gc_set_error("too many workgroups (%lld)", total);
    return GC_E_INVALID;
  }
  dim3 grid((unsigned int)total);
  int rc;
  if (fast > 0 && gc_fast_table_mode(ctx) == 1) {
    // WIDE fast kernel: 8-sample chunks, four waves per workgroup
    a.wide = 1;
    fast = 1;
    a.share_el = 0;
    if (a.bpw == 1) {
      if (splits % 4 != 0 && splits != 1) {
        gc_set_error("internal: WIDE correlator launch needs splits %% 4 == 0 (got %d)", splits);
        return GC_E_INVALID;
      }
      if (splits == 1) {
        // unrelated blocks cannot share a staged table: one block per workgroup, split four ways in-kernel
        // is not supported -> take the generic kernel for such lists
        a.wide = 0;
        fast = 0;
      } else {
        total = (total + 3) / 4;
        grid = dim3((unsigned int)total);
      }
    }
    a.xcd_swizzle = (a.wide && total % 8 == 0 && total >= 64) ? 1 : 0;
  }
  if (fast < 0) {
    // mixed ramp multipliers: exact per-sample kernel
    int mode;
    if (ctx->if_dtype == GC_I8)
      mode = ctx->if_layout == GC_IQ ? I8_IQ : ctx->if_layout == GC_QI ? I8_QI : I8_REAL;
    else
      mode = ctx->if_layout == GC_IQ ? I16_IQ : ctx->if_layout == GC_QI ? I16_QI : I16_REAL;
    switch (mode) {
      case I8_IQ: hipLaunchKernelGGL((corr_epl_mixed_kernel<I8_IQ>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I8_QI: hipLaunchKernelGGL((corr_epl_mixed_kernel<I8_QI>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I16_IQ: hipLaunchKernelGGL((corr_epl_mixed_kernel<I16_IQ>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I16_QI: hipLaunchKernelGGL((corr_epl_mixed_kernel<I16_QI>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I8_REAL: hipLaunchKernelGGL((corr_epl_mixed_kernel<I8_REAL>), grid, dim3(kWG), 0, ctx->stream, a); break;
      default: hipLaunchKernelGGL((corr_epl_mixed_kernel<I16_REAL>), grid, dim3(kWG), 0, ctx->stream, a); break;
    }
    rc = (hipGetLastError() == hipSuccess) ? GC_OK : GC_E_HIP;
  } else if (fast) {
    a.red_off = (a.wide ? 2 : 8) * ctx->max_lds_bytes;  // float2 {c, dc} tables: 8 bytes per staged entry (int8 pairs: 2)
    rc = gc_launch_correlator_fast(ctx, a, ib, (unsigned int)total, max_arms, fast == 2);
  } else {
    switch (max_arms) {
      case 1: rc = launch_mode<1>(ctx, a, grid); break;
      case 2: rc = launch_mode<2>(ctx, a, grid); break;
      default: rc = launch_mode<3>(ctx, a, grid); break;
    }
  }
  if (rc != GC_OK) return rc;
  if (splits > 1 && d_out != nullptr) {
    const long long n = nblocks * GC_OUT_STRIDE;
    hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_partial, d_out, (long long)nblocks, splits);
    GC_HIP(hipGetLastError());
  }
  return GC_OK;
}
