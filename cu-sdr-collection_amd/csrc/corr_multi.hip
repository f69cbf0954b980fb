// corr_multi.hip — the transition / prefix-sum formulation of corr_fast.hip carried to SEVERAL table transitions per
// lane-chunk: replicas whose table index advances by up to KT entries over a 16-sample chunk ((16-1)*step*R*M < KT).
//   KT = 2: Galileo E1 B+C and BDS B1C BOC(1,1) tables, BDS B1I at 18 Msps (0.114 entries per sample; corr_fast.hip
//           takes them with 8-sample chunks and one transition), GPS L2C CM at 8 Msps
//   KT = 4: the 10.23-Mcps codes at 50 Msps (GPS L5, BDS B2a: 0.2046 chips per sample, 3.07 per chunk) — so far
//           the lane kernel's (corr_lane.hip: 12 fma per sample for six tap-arms, 24.8 VALU per channel-sample)
// Same arithmetic contract as corr_kernel.hip / corr_fast.hip (tracking.m:247-300; GPS_L5C/include/tracking.m:255-326,
// GAL_E1C/include/tracking.m:236-303 for the two-arm packages).
//
//   * per sample only what does not depend on the replica: two sign-extending converts, four fused multiply-adds for
//     the running sums P_j = y_0 + .. + y_j (y = carrier-wiped sample), parked in LDS with ds_write_addtid_b32;
//   * per TRANSITION, not per sample, everything else: the ramp t = k0 - G crosses the integers k0 + n at the sample
//     positions u_n = (G + n) / (step*R*M); with m_n = floor(u_n) clamped to the chunk, the samples (m_{n-1}, m_n]
//     read table entry k0 + n, so a tap's sum over the chunk is  sum_n c[k0 + n] * (P[m_n] - P[m_{n-1}])  with
//     P[m_{-1}] = 0 and P[m_KT] = T: one ds_read2_b32 per transition and ramp, two multiply-adds per transition,
//     tap and arm.  Early and late taps half a table entry either side of prompt share one ramp (entries shifted
//     by one), data and pilot arms share every position and every segment sum;
//   * the arms' tables are interleaved in LDS as int8 (ARMS bytes per entry: 20.5 KB for two 10 232-entry tables)
//     and shared by the four wavefronts of a workgroup; a workgroup walks several consecutive epochs of ONE channel of
//     a periodic replay list (the closed loops keep their kernels: they are latency-bound);
//   * chunks in which any transition position is within 4e-6 samples of an integer (where float32 and the reference's
//     float64 rounding could disagree about a sample's table entry) take the exact float64 per-sample path, as in
//     corr_fast.hip; blocks the host proved tie-free (gc_mark_tie_free) skip the test.
#include <cstdlib>

#include "corr_common.h"

using namespace gcorr;

namespace {

constexpr float kTieTolM = 4e-6f;  // |u_n| < 20: float32 resolution 1.9e-6, accumulated error of u_0 + n * ustep <= 2.7e-6
constexpr int kMW = 64;
constexpr int kMSPL = 16;
constexpr int kMaxLds = 160 * 1024;
constexpr int kGLO = 8;   // zero guard entries below entry 0 (a block's first chunk starts up to 15 samples early: k0 >= -1 - KT)
constexpr int kGHI = 8;   // ... and above the last staged entry (a lane's last chunk runs up to KT + 1 entries past the block's end)
#ifndef GC_MULTI_SCHED_GROUP
#define GC_MULTI_SCHED_GROUP 4
#endif

template <int OFF_RE, int OFF_IM>
__device__ __forceinline__ void store_prefix_m(float tr, float ti, unsigned int lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"
               :
               : "v"(tr), "v"(ti), "s"(lds_base), "n"(OFF_RE), "n"(OFF_IM)
               : "memory", "m0");
}

template <int B>
__device__ __forceinline__ float cvt_sbyte(unsigned int word) {
  return cvt_byte<B>(word);
}

// ARMS in {1, 2}; MODE in {I8_IQ, I8_QI, I16_IQ, I16_QI}; KT in {2, 4}; SHARE_EL: every block has 2*earlyLateSpc*R*M == 1 (host-checked);
// NWV wavefronts per workgroup share the staged tables: 16 where tables + 16 x 8 KB of running sums fit the CU's 160 KB (one
// workgroup = four waves per SIMD: the sample loop is one dependent chain of 32 multiply-adds per component, and what hides its
// latency is other waves), 8 for the longest tables, 4 for short lists (more, smaller workgroups: less of a tail)
template <int ARMS, int MODE, int KT, bool SHARE_EL, int NWV>
__global__ __launch_bounds__(NWV* kMW) void corr_epl_multi_kernel(const KArgs p) {
  constexpr int kMWaves = NWV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SPL = kMSPL;
  constexpr int NW = SPL * Fmt<MODE>::bps / 4;
  constexpr int kShift = 4;

  long long wg = blockIdx.x;
  if (p.xcd_swizzle) {
    const long long per = (long long)gridDim.x >> 3;  // the host rounds the grid up to a multiple of 8 when swizzling
    wg = (wg & 7) * per + (wg >> 3);
    if (wg >= p.total_wg) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long grp = wg / p.stride;
  const int cslot = (int)(wg - grp * p.stride);

  // ---- stage the channel's tables, arms interleaved: byte (k + kGLO) * ARMS + arm = c_arm[k], zeros in the guards -------------
  {
    const long long lb0 = min(grp * p.bpw * p.stride + cslot, (long long)p.nblocks - 1);
    const gc_block blk0 = p.blocks[lb0];
    const DevChannel* __restrict__ chn0 = p.chans + blk0.channel;
    int nmax = 0;
#pragma unroll
    for (int a = 0; a < ARMS; ++a)
      if (a < chn0->arms) nmax = max(nmax, chn0->nent[a]);
    const int entries = nmax + kGLO + kGHI;
    for (int e = threadIdx.x; e < entries; e += kMWaves * kMW) {
      const int k = e - kGLO;
#pragma unroll
      for (int a = 0; a < ARMS; ++a) {
        signed char v = 0;
        if (a < chn0->arms && k >= 0 && k < chn0->nent[a]) v = chn0->tab[a][k];
        reinterpret_cast<signed char*>(smem)[e * ARMS + a] = v;
      }
    }
    __syncthreads();
  }
  // the arms' values of entry k (k = -kGLO .. n + kGHI - 1) as one word: byte a = arm a
  auto table_word = [&](int k) -> unsigned int {
    if constexpr (ARMS == 1) return (unsigned int)smem[k + kGLO];
    else return (unsigned int)reinterpret_cast<const unsigned short*>(smem)[k + kGLO];
  };
  float* pfx = reinterpret_cast<float*>(smem + p.red_off) + wave * SPL * 2 * kMW;
  const unsigned int pfx_m0 = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)pfx);

  for (int bi = wave; bi < p.bpw; bi += kMWaves) {
  const long long lb = (grp * p.bpw + bi) * p.stride + cslot;
  if (lb >= p.nblocks) break;
  const gc_block blk = p.blocks[lb];
  const DevChannel* __restrict__ chn = p.chans + blk.channel;
  const int arms_here = chn->arms;

  // ---- per-block uniform quantities (corr_fast.hip; corr_kernel.hip has the reference line citations) --------
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
  const double tau = blk.carr_freq * p.inv_fs;
  const double spM = sp * M;
  double rspM = __builtin_amdgcn_rcp(spM);
  rspM = fma(rspM, fma(-spM, rspM, 1.0), rspM);
  rspM = fma(rspM, fma(-spM, rspM, 1.0), rspM);
  const float uk = (float)(rspM * 2.3283064365386963e-10);  // g_hi (2^-32 units) -> u_0
  const float ustep = (float)rspM;                           // samples between two transitions
  const bool tie_free = __builtin_amdgcn_readfirstlane((int)(blk.reserved & 1)) != 0;

  // lanes 0..SPL-1: delta^j = exp(-i*2*pi*j*tau); lane SPL: the chunk stride SPL*64 samples
  float myC, myS;
  unsigned int myJlo, myJhi;
  int myJint;
  {
    const int j = (lane < SPL) ? lane : SPL * kMW;
    const double x = (double)j * tau;
    sincospif(2.0f * (float)(x - floor(x)), &myS, &myC);
    const double y = (double)j * (sp * M);
    const double yi = floor(y);
    const unsigned long long jf = frac_to_u64(y - yi);
    myJint = (int)yi;
    myJlo = (unsigned int)jf;
    myJhi = (unsigned int)(jf >> 32);
  }
  float C[SPL], S[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    C[j] = rl_f(myC, j);
    S[j] = rl_f(myS, j);
  }
  const float rotC = rl_f(myC, SPL), rotS = rl_f(myS, SPL);
  const unsigned long long Df = ((unsigned long long)rl_u(myJhi, SPL) << 32) | rl_u(myJlo, SPL);
  const int Di = __builtin_amdgcn_readlane(myJint, SPL);

  const long long q0 = s0 >> kShift;
  const long long q1 = (s0 + N - 1) >> kShift;
  const int nchunks = (int)(q1 - q0 + 1);
  const int cbeg = 0, cend = nchunks;

  float accr[ARMS][3], acci[ARMS][3];
#pragma unroll
  for (int a = 0; a < ARMS; ++a)
#pragma unroll
    for (int x = 0; x < 3; ++x) accr[a][x] = acci[a][x] = 0.0f;

  const int iters = (cend - cbeg + kMW - 1) / kMW;
  const int c0 = cbeg + lane;
  float wc = 1.0f, ws = 0.0f;
  if (iters > 0) {
    constexpr int CB = SPL * Fmt<MODE>::bps;
    const int i00 = (int)((q0 + c0) * SPL - s0);
    constexpr bool SHARE = SHARE_EL;
    constexpr int NS = SHARE ? 2 : 3;
    Fx fx[NS];
    const double isp = __dmul_rn((double)i00, sp);
    fx[0] = to_fx(__dmul_rn(__dadd_rn(aE, isp), M));
    fx[1] = to_fx(__dmul_rn(__dadd_rn(aP, isp), M));
    if constexpr (!SHARE) fx[2] = to_fx(__dmul_rn(__dadd_rn(aL, isp), M));
    const uint8_t* __restrict__ base = p.if_base;
    // SHARE instantiation, but this block's spacing is not half an entry (host-checked: cannot happen): every chunk goes exact
    const bool share_broken = SHARE_EL && __builtin_amdgcn_readfirstlane(2.0 * d * R * M == 1.0 ? 0 : 1) != 0;
    unsigned int glo[NS], ghi[NS];
    int kk[NS];
#pragma unroll
    for (int sx = 0; sx < NS; ++sx) {
      glo[sx] = (unsigned int)fx[sx].G;
      ghi[sx] = (unsigned int)(fx[sx].G >> 32);
      kk[sx] = fx[sx].k0;
    }
    const unsigned int Dlo = (unsigned int)Df, Dhi = (unsigned int)(Df >> 32);
    const uint8_t* __restrict__ bs = base + (long long)CB * (q0 + cbeg);
    const unsigned int voff = (unsigned int)lane * CB;
    const unsigned int voff_last = min(voff, (unsigned int)(cend - 1 - cbeg - (iters - 1) * kMW) * CB);

    auto load_k = [&](const int k, unsigned int (&w)[NW]) {
      const uint8_t* __restrict__ pk = bs + (size_t)k * (size_t)(kMW * CB);
      const unsigned int off = (k == iters - 1) ? voff_last : voff;  // idle lanes of the last iteration re-read the last chunk
      load_words<MODE, SPL>(pk + off, 0, w);
    };

    auto process = [&](unsigned int (&w)[NW], const int k) {
      const bool last = (k == iters - 1);
      if ((k == 0) | last) {
        int kq = k;
        asm volatile("" : "+v"(kq));  // opaque: nothing of this rare branch is to be precomputed outside the loop
        const int i0 = i00 + kq * (SPL * kMW);
        if (last && c0 + kq * kMW >= cend) {  // idle lane: no samples, and table indices that exist
#pragma unroll
          for (int q = 0; q < NW; ++q) w[q] = 0u;
#pragma unroll
          for (int sx = 0; sx < NS; ++sx) kk[sx] = 0;
        }
        if ((i0 < 0) | (i0 + SPL > N)) mask_words<MODE, SPL>(w, i0, N);
      }

      // transition positions u_n = u_0 + n / (step*R*M) and the near-tie filter
      float un[NS][KT];
#pragma unroll
      for (int sx = 0; sx < NS; ++sx) {
        float gh;
        asm("v_cvt_f32_u32_e32 %0, %1" : "=v"(gh) : "v"(ghi[sx]));
        const float u0 = gh * uk;
#pragma unroll
        for (int n = 0; n < KT; ++n) un[sx][n] = (n == 0) ? u0 : fmaf((float)n, ustep, u0);
      }
      bool exact = share_broken;
      if (!tie_free) {
        bool suspect = false;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx)
#pragma unroll
          for (int n = 0; n < KT; ++n) suspect |= fabsf(un[sx][n] - rintf(un[sx][n])) < kTieTolM;
        exact |= __any(suspect) != 0;
      }

      float Ur[ARMS][3], Ui[ARMS][3];
      if (exact) {
        // ---- exact path: float64 index per sample, as the reference (rolled loop, samples re-read from memory) ------
#pragma unroll
        for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
          for (int x = 0; x < 3; ++x) Ur[ar][x] = Ui[ar][x] = 0.0f;
        {
          int kq = k, Nq = N;
          asm volatile("" : "+v"(kq), "+v"(Nq));
          const int c = c0 + kq * kMW;
          const bool act = c < cend;
          const int i0 = i00 + kq * (SPL * kMW);
          const double bP = __dmul_rn(__dadd_rn(__dmul_rn((double)(Nq - 1), step), rem), R);
          const double bE = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(Nq - 1), step), rem), -d), R);
          const double bL = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(Nq - 1), step), rem), d), R);
          const uint8_t* sp8 = base + (long long)CB * (q0 + min(c, cend - 1));
          int nmax = 0;
#pragma unroll
          for (int a = 0; a < ARMS; ++a)
            if (a < arms_here) nmax = max(nmax, chn->nent[a]);
          float cr = 1.0f, ci = 0.0f;  // delta^j = cr - i*ci
#pragma unroll 1
          for (int j = 0; j < SPL; ++j) {
            const int i = i0 + j;
            float x0, x1;
            if constexpr (Fmt<MODE>::bps == 2) {
              x0 = (float)(signed char)sp8[2 * j];
              x1 = (float)(signed char)sp8[2 * j + 1];
            } else {
              x0 = (float)reinterpret_cast<const short*>(sp8)[2 * j];
              x1 = (float)reinterpret_cast<const short*>(sp8)[2 * j + 1];
            }
            float a = Fmt<MODE>::swap ? x1 : x0, b = Fmt<MODE>::swap ? x0 : x1;
            if ((unsigned int)i >= (unsigned int)Nq || !act) a = b = 0.0f;
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
              if (2 * i < Nq - 1)
                t = __dadd_rn(ax, __dmul_rn((double)i, sp));
              else if (2 * i > Nq - 1)
                t = __dadd_rn(bx, -__dmul_rn((double)(Nq - 1 - i), sp));
              else
                t = __dadd_rn(ax, bx) / 2.0;
              int kx = (int)ceil(__dmul_rn(t, M));
              kx = max(-kGLO, min(kx, nmax + kGHI - 1));
              const unsigned int e = table_word(kx);
#pragma unroll
              for (int ar = 0; ar < ARMS; ++ar) {
                const float cf = (float)(signed char)(e >> (8 * ar));
                Ur[ar][x] = fmaf(cf, yr, Ur[ar][x]);
                Ui[ar][x] = fmaf(cf, yi, Ui[ar][x]);
              }
            }
          }
        }
      } else {
        // ---- running sums to LDS as they are formed ----------------------------------------------------------------
        float Tr = 0.f, Ti = 0.f;
        static_for<0, SPL>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (j % GC_MULTI_SCHED_GROUP == 0 && j != 0) __builtin_amdgcn_sched_barrier(0);
          float a, b;
          sample_ab<MODE, j, NW>(w, a, b);
          Tr = fmaf(a, C[j], fmaf(b, S[j], Tr));
          Ti = fmaf(b, C[j], fmaf(-a, S[j], Ti));
          store_prefix_m<(2 * j) * kMW * 4, (2 * j + 1) * kMW * 4>(Tr, Ti, pfx_m0);
        });
        // ---- per ramp: the segment sums D_n = P[m_n] - P[m_{n-1}], n = 0 .. KT ------------------------------------------
        float Dr[NS][KT + 1], Di_[NS][KT + 1];
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) {
          float pr = 0.f, pi = 0.f;
#pragma unroll
          for (int n = 0; n < KT; ++n) {
            const int m = min((int)un[sx][n], SPL - 1);  // samples .. m read entries <= k0 + n (P[SPL-1] = T: nothing after)
            const float qr = pfx[(2 * m) * kMW + lane];
            const float qi = pfx[(2 * m + 1) * kMW + lane];
            Dr[sx][n] = qr - pr;
            Di_[sx][n] = qi - pi;
            pr = qr;
            pi = qi;
          }
          Dr[sx][KT] = Tr - pr;
          Di_[sx][KT] = Ti - pi;
        }
        // ---- per tap and arm: sum_n c[k0 + n] * D_n --------------------------------------------------------------------
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) {
          constexpr int NE = KT + 1;
          const bool with_late = SHARE && sx == 0;
          float cv[ARMS][NE + 1];
#pragma unroll
          for (int n = 0; n < NE + 1; ++n) {
            if (n == NE && !with_late) break;
            const unsigned int e = table_word(kk[sx] + n);
            cv[0][n] = cvt_sbyte<0>(e);
            if constexpr (ARMS == 2) cv[1][n] = cvt_sbyte<1>(e);
          }
          const int x = (sx == 0) ? 0 : (sx == 1) ? 1 : 2;
#pragma unroll
          for (int ar = 0; ar < ARMS; ++ar) {
            float ur = cv[ar][0] * Dr[sx][0], ui = cv[ar][0] * Di_[sx][0];
#pragma unroll
            for (int n = 1; n < NE; ++n) {
              ur = fmaf(cv[ar][n], Dr[sx][n], ur);
              ui = fmaf(cv[ar][n], Di_[sx][n], ui);
            }
            Ur[ar][x] = ur;
            Ui[ar][x] = ui;
            if (with_late) {  // late = the same ramp one entry on
              float lr = cv[ar][1] * Dr[sx][0], li = cv[ar][1] * Di_[sx][0];
#pragma unroll
              for (int n = 1; n < NE; ++n) {
                lr = fmaf(cv[ar][n + 1], Dr[sx][n], lr);
                li = fmaf(cv[ar][n + 1], Di_[sx][n], li);
              }
              Ur[ar][2] = lr;
              Ui[ar][2] = li;
            }
          }
        }
      }
      // Horner step: acc = acc * conj(rho) + U, rho = delta^(SPL*64) = rotC - i rotS
#pragma unroll
      for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const float nr = fmaf(accr[ar][x], rotC, fmaf(-acci[ar][x], rotS, Ur[ar][x]));
          const float ni = fmaf(accr[ar][x], rotS, fmaf(acci[ar][x], rotC, Ui[ar][x]));
          accr[ar][x] = nr;
          acci[ar][x] = ni;
        }
      // next chunk: t += 64*SPL*step*R*M, exactly
#pragma unroll
      for (int sx = 0; sx < NS; ++sx)
        asm("v_sub_co_u32_e32 %0, vcc, %0, %3\n\tv_subb_co_u32_e32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32_e32 %2, vcc, %2, %5, vcc"
            : "+v"(glo[sx]), "+v"(ghi[sx]), "+v"(kk[sx])
            : "v"(Dlo), "v"(Dhi), "v"(Di)
            : "vcc");
    };

    unsigned int wa[NW], wb[NW];
    load_k(0, wa);
    for (int k = 0;; k += 2) {
      if (k + 1 < iters) load_k(k + 1, wb);
      process(wa, k);
      if (k + 1 >= iters) break;
      if (k + 2 < iters) load_k(k + 2, wa);
      process(wb, k + 1);
      if (k + 2 >= iters) break;
    }
    const double ph = blk.rem_carr_phase * 0.15915494309189535 + (double)(i00 + (iters - 1) * (SPL * kMW)) * tau;
    sincospif(2.0f * (float)(ph - floor(ph)), &ws, &wc);
  }

  // ---- rotate into the absolute frame and reduce across the wavefront (DPP) ------------------------
  double* o = p.out + lb * GC_OUT_STRIDE;
  float tot[ARMS * 6];
#pragma unroll
  for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      tot[ar * 6 + 2 * x] = wave_sum_lane63(wc * accr[ar][x] + ws * acci[ar][x]);
      tot[ar * 6 + 2 * x + 1] = wave_sum_lane63(wc * acci[ar][x] - ws * accr[ar][x]);
    }
  if (lane == 63) {
#pragma unroll
    for (int v = 0; v < ARMS * 6; ++v) o[v] = (v < arms_here * 6) ? (double)tot[v] : 0.0;
    for (int v = ARMS * 6; v < GC_OUT_STRIDE; ++v) o[v] = 0.0;
  }
  }  // bpw loop
}

template <int ARMS, int MODE, int KT, bool SHARE, int NWV>
void launch_multi_one(gc_context* ctx, const KArgs& a, dim3 grid, size_t smem) {
  const void* fn = reinterpret_cast<const void*>(corr_epl_multi_kernel<ARMS, MODE, KT, SHARE, NWV>);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((corr_epl_multi_kernel<ARMS, MODE, KT, SHARE, NWV>), grid, dim3(NWV * kMW), smem, ctx->stream, a);
}

template <int ARMS, int MODE, int NWV>
void launch_multi_mode(gc_context* ctx, const KArgs& a, dim3 grid, size_t smem, int kt, bool share) {
  if (kt <= 2) {
    if (share) launch_multi_one<ARMS, MODE, 2, true, NWV>(ctx, a, grid, smem);
    else launch_multi_one<ARMS, MODE, 2, false, NWV>(ctx, a, grid, smem);
  } else {
    if (share) launch_multi_one<ARMS, MODE, 4, true, NWV>(ctx, a, grid, smem);
    else launch_multi_one<ARMS, MODE, 4, false, NWV>(ctx, a, grid, smem);
  }
}

template <int NWV>
void launch_multi_waves(gc_context* ctx, const KArgs& a, dim3 grid, size_t smem, int max_arms, int kt, bool share) {
  const bool qi = ctx->if_layout == GC_QI;
  if (ctx->if_dtype == GC_I16) {  // int16 I/Q records: 64 bytes per lane-chunk (instantiated for four-wave workgroups: 136-151 VGPRs, three workgroups per CU)
    if constexpr (NWV == 4) {
      if (max_arms <= 1) {
        if (qi) launch_multi_mode<1, I16_QI, NWV>(ctx, a, grid, smem, kt, share);
        else launch_multi_mode<1, I16_IQ, NWV>(ctx, a, grid, smem, kt, share);
      } else {
        if (qi) launch_multi_mode<2, I16_QI, NWV>(ctx, a, grid, smem, kt, share);
        else launch_multi_mode<2, I16_IQ, NWV>(ctx, a, grid, smem, kt, share);
      }
    }
    return;
  }
  if (max_arms <= 1) {
    if (qi) launch_multi_mode<1, I8_QI, NWV>(ctx, a, grid, smem, kt, share);
    else launch_multi_mode<1, I8_IQ, NWV>(ctx, a, grid, smem, kt, share);
  } else {
    if (qi) launch_multi_mode<2, I8_QI, NWV>(ctx, a, grid, smem, kt, share);
    else launch_multi_mode<2, I8_IQ, NWV>(ctx, a, grid, smem, kt, share);
  }
}

}  // namespace

// LDS bytes of the interleaved int8 tables of a launch whose longest table has `max_entries` entries
int gc_multi_table_bytes(int max_entries, int arms) { return ((max_entries + kGLO + kGHI) * (arms <= 1 ? 1 : 2) + 15) / 16 * 16; }

// Wavefronts per workgroup for a launch: the most of {16, 12, 8, 4} whose LDS (tables + 8 KB of running sums per wave) fits the CU
// and that still leaves the list >= 2 workgroups per CU (GC_MULTI_WAVES overrides); 0 = the tables do not fit at all.
int gc_multi_waves(const gc_context* ctx, int max_arms, long long nblocks, int period, int kt, bool share_el) {
  const int tb = gc_multi_table_bytes(ctx->max_stage_len, max_arms);
  int forced = 0;
  if (const char* e = GC_TUNE_ENV("GC_MULTI_WAVES")) forced = std::atoi(e);
  if (ctx->if_dtype == GC_I16) return tb + 4 * kMSPL * 2 * kMW * (int)sizeof(float) <= kMaxLds ? 4 : 0;  // the int16 instantiations: 4 waves
  // two transitions per chunk (short tables: Galileo E1, BDS B1I): three four-wave workgroups per CU measured 3 % ahead of one
  // sixteen-wave workgroup (e1x8: 1.40 against 1.44 ms); four transitions (GPS L5 at 50 Msps): the other way round (3.95 / 4.10 ms)
  if (forced == 0 && kt <= 2 && 3 * (tb + 4 * kMSPL * 2 * kMW * (int)sizeof(float)) <= kMaxLds) return 4;
  for (int w : {16, 12, 8, 4}) {
    if (tb + w * kMSPL * 2 * kMW * (int)sizeof(float) > kMaxLds) continue;
    if (w == 16 && max_arms == 2 && kt == 4 && !share_el) continue;  // three ramps x four transitions x two arms: 135 VGPRs, over the 128 a 1024-thread workgroup gets
    if (forced == w) return w;
    if (forced == 0 && (w == 4 || nblocks / ((long long)w * std::max(1, period)) * period >= 2LL * ctx->compute_units)) return w;
  }
  return 0;
}

// Periodic replay lists of int8 I/Q (Q/I) records, one or two arms with one ramp multiplier, at most `kt` (2 or 4) table
// transitions per 16-sample chunk; a.bpw = a multiple of `waves` (gc_multi_waves), a.stride = the list's period, a.splits == 1.
int gc_launch_correlator_multi(gc_context* ctx, const KArgs& a_in, unsigned int grid, int max_arms, int kt, bool share_el, int waves) {
  KArgs a = a_in;
  a.red_off = gc_multi_table_bytes(ctx->max_stage_len, max_arms);
  const size_t smem = (size_t)a.red_off + (size_t)waves * kMSPL * 2 * kMW * sizeof(float);
  if (waves == 16) launch_multi_waves<16>(ctx, a, dim3(grid), smem, max_arms, kt, share_el);
  else if (waves == 12) launch_multi_waves<12>(ctx, a, dim3(grid), smem, max_arms, kt, share_el);
  else if (waves == 8) launch_multi_waves<8>(ctx, a, dim3(grid), smem, max_arms, kt, share_el);
  else launch_multi_waves<4>(ctx, a, dim3(grid), smem, max_arms, kt, share_el);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
