// corr_cboc.hip — the hybrid correlator for channels that carry a BOC(6,1) arm next to its BOC(1,1) arm: Galileo E1-C
// CBOC(6,1,1/11) as BASELINE config 3 words it, BDS B1C wide-band (BDS/B1C/include/WB_tracking.m:285-317: tcode2 = ceil(tcode) + 1
// for B1CData / pilotBOC11, ceil(tcode * 6) + 1 for pilotBOC61; :338-369: the 18 sums).  Three arms x three taps.
//
// Mapping as corr_multi.hip (lane = 16 consecutive samples, aligned to the absolute sample index; a wave walks 64 chunks at a
// time), and the same transition / prefix-sum formulation for the two BOC(1,1) arms: at 18 Msps a half-chip table advances
// 0.114 entries per sample, a chunk crosses at most two entries per tap (KT = 2), and a tap's sum is
//     c[k0] P[m_0] + c[k0+1] (P[m_1] - P[m_0]) + c[k0+2] (T - P[m_1])          P = running sums of the carrier-wiped samples y
// with nothing per sample and tap.  The BOC(6,1) arm crosses ~11 entries per chunk: for it transitions cost more than samples.
// But its table is the BOC(1,1) table times a sign that flips every sixth of an entry (gc_channel_is_derived checks exactly that:
// entry k6 = arm-1 entry p = (k6 + 5) / 6 times (-1)^(p + k6)), and away from ties ceil(ceil(6t) / 6) == ceil(t).  With the ramp
// written t = p - g, 0 < g < 1, the sign is (-1)^(p + floor(6g)) - and floor(6g) is odd exactly where frac(3g) >= 1/2: bit 31 of
// three times the ramp's 32-bit fraction word.  So per sample and tap: one integer add (the word's chain), one v_and_or (the bit
// as +-1.0f), two multiply-adds into the running sums P6 of the SIGN-MODULATED samples - and the arm's sum is the same three-term
// expression over P6 with the coefficients c[k0+n] (-1)^(k0+n).  No table at six times the rate, no gather, no second ramp.
//   per sample:      2 converts + 4 (y) + 2 (P) + 3 x 4 (P6 of early / prompt / late)                       = 20 VALU, 8 LDS stores
//   per chunk:       positions, 12 LDS reads, 27 coefficient ops, 54 tap multiply-adds, 36 Horner, ramp advance  ~ 230 VALU
//   (lane kernel, corr_lane.hip DER: 47.4 VALU per sample for the same 18 sums)
// The running sums of all four streams are parked in LDS with ds_write_addtid_b32 (8 floats per sample and lane: 32 KB per wave),
// so a workgroup is NWV <= 4 waves next to the two interleaved int8 tables.
//
// Exactness: as corr_multi.hip for the BOC(1,1) positions (float32 positions, chunks with a position within 4e-6 samples of an
// integer take the float64 per-sample path); for the six-fold edges an integer test on the fraction words (band kTie6 units of
// 2^-32 sub-entry, inside the band gc_mark_tie_free searches with).  gc_block::reserved bit 0 = the host proved the block free
// of both kinds (no test at all), bit 1 = free of six-fold near-edges (the per-sample integer test is skipped).
#include <cstdlib>

#include "corr_common.h"

using namespace gcorr;

namespace {

constexpr float kTieTolM = 4e-6f;   // corr_multi.hip
constexpr unsigned int kTie6 = 160u;  // 6 x (2 units of the truncated words + 16 chain steps) + the reference's float64 rounding + slack
constexpr int kW = 64;
constexpr int kSPL = 16;
constexpr int kMaxLds = 160 * 1024;
constexpr int kGLO = 8;
constexpr int kGHI = 8;
constexpr int kRow = 2;                                // floats per sample and lane: one stream's running sum, re and im
constexpr int kWaveLds = kSPL * kRow * kW * 4;         // 8 KB: the rows of ONE stream, used by P, then by the P6 of each tap in turn

template <int OFF_RE, int OFF_IM>
__device__ __forceinline__ void park2(float tr, float ti, unsigned int lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"
               :
               : "v"(tr), "v"(ti), "s"(lds_base), "n"(OFF_RE), "n"(OFF_IM)
               : "memory", "m0");
}

// four samples' running sums (re, im of rows J .. J + 3) in one go: ONE m0 set-up for eight stores
template <int J>
__device__ __forceinline__ void park8(const float (&tr)[4], const float (&ti)[4], unsigned int lds_base) {
  asm volatile(
      "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
      "ds_write_addtid_b32 %0 offset:%9\n\tds_write_addtid_b32 %1 offset:%10\n\t"
      "ds_write_addtid_b32 %2 offset:%11\n\tds_write_addtid_b32 %3 offset:%12\n\t"
      "ds_write_addtid_b32 %4 offset:%13\n\tds_write_addtid_b32 %5 offset:%14\n\t"
      "ds_write_addtid_b32 %6 offset:%15\n\tds_write_addtid_b32 %7 offset:%16"
      :
      : "v"(tr[0]), "v"(ti[0]), "v"(tr[1]), "v"(ti[1]), "v"(tr[2]), "v"(ti[2]), "v"(tr[3]), "v"(ti[3]), "s"(lds_base),
        "n"((kRow * J) * kW * 4), "n"((kRow * J + 1) * kW * 4), "n"((kRow * (J + 1)) * kW * 4), "n"((kRow * (J + 1) + 1) * kW * 4),
        "n"((kRow * (J + 2)) * kW * 4), "n"((kRow * (J + 2) + 1) * kW * 4), "n"((kRow * (J + 3)) * kW * 4), "n"((kRow * (J + 3) + 1) * kW * 4)
      : "memory", "m0");
}

// a block-uniform float64 into scalar registers (the compiler computes the block's constants with vector instructions and keeps them in
// VGPRs - two dozen registers of a kernel that has 128 at four waves per SIMD)
__device__ __forceinline__ double uni64(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// MODE in {I8_IQ, I8_QI}; NWV wavefronts per workgroup share the staged tables of one channel
template <int MODE, int NWV>
__global__ __launch_bounds__(NWV* kW) void corr_epl_cboc_kernel(const KArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SPL = kSPL;
  constexpr int NW = SPL * Fmt<MODE>::bps / 4;
  constexpr int kShift = 4;
  constexpr int NS = 3;

  long long wg = blockIdx.x;
  if (p.xcd_swizzle) {
    const long long per = (long long)gridDim.x >> 3;
    wg = (wg & 7) * per + (wg >> 3);
    if (wg >= p.total_wg) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long grp = wg / p.stride;
  const int cslot = (int)(wg - grp * p.stride);

  // ---- stage arms 0 and 1 interleaved: bytes (k + kGLO) * 2 + {0, 1}, zeros in the guards -------------------------------------
  int nmax;
  {
    const long long lb0 = min(grp * p.bpw * p.stride + cslot, (long long)p.nblocks - 1);
    const gc_block blk0 = p.blocks[lb0];
    const DevChannel* __restrict__ chn0 = p.chans + blk0.channel;
    nmax = max(chn0->nent[0], chn0->nent[1]);
    const int entries = nmax + kGLO + kGHI;
    for (int e = threadIdx.x; e < entries; e += NWV * kW) {
      const int k = e - kGLO;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        signed char v = 0;
        if (k >= 0 && k < chn0->nent[a]) v = chn0->tab[a][k];
        reinterpret_cast<signed char*>(smem)[e * 2 + a] = v;
      }
    }
    __syncthreads();
  }
  auto table_word = [&](int k) -> unsigned int { return (unsigned int)reinterpret_cast<const unsigned short*>(smem)[k + kGLO]; };
  float* pfx = reinterpret_cast<float*>(smem + p.red_off) + wave * (kWaveLds / 4);
  const unsigned int pfx_m0 = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)pfx);

  for (int bi = wave; bi < p.bpw; bi += NWV) {
    const long long lb = (grp * p.bpw + bi) * p.stride + cslot;
    if (lb >= p.nblocks) break;
    const gc_block blk = p.blocks[lb];
    const DevChannel* __restrict__ chn = p.chans + blk.channel;

    // ---- per-block uniform quantities (corr_multi.hip / corr_kernel.hip have the reference line citations) --------
    const double R = uni64(chn->index_scale);
    const double M = uni64(chn->mult[0]);
    const double M6 = uni64(chn->mult[2]);
    const double rem = uni64(blk.rem_code_phase);
    const double step = uni64(blk.code_phase_step);
    const double d = uni64(blk.el_spacing);
    const int N = __builtin_amdgcn_readfirstlane(blk.blksize);
    const long long s0 = blk.first_sample;
    const double aE = uni64((rem - d) * R);
    const double aL = uni64((rem + d) * R);
    const double aP = uni64(rem * R);
    const double sp = uni64(step * R);
    const double tau = uni64(blk.carr_freq * p.inv_fs);
    const double spM = uni64(sp * M);
    double rspM = __builtin_amdgcn_rcp(spM);
    rspM = fma(rspM, fma(-spM, rspM, 1.0), rspM);
    rspM = fma(rspM, fma(-spM, rspM, 1.0), rspM);
    const float uk = (float)(rspM * 2.3283064365386963e-10);
    const float ustep = (float)rspM;
    const int flags = __builtin_amdgcn_readfirstlane((int)blk.reserved);
    const bool tie_free = (flags & 1) != 0;
    const bool tie6_free = (flags & 2) != 0;
    // three times the per-sample advance of a ramp's fraction word, negated: the sign word's chain (spM < 1: KT <= 2)
    const unsigned int negD3 = __builtin_amdgcn_readfirstlane(0u - 3u * (unsigned int)((spM - floor(spM)) * 4294967296.0));

    float myC, myS;
    unsigned int myJlo, myJhi;
    int myJint;
    {
      const int j = (lane < SPL) ? lane : SPL * kW;
      const double x = (double)j * tau;
      sincospif(2.0f * (float)(x - floor(x)), &myS, &myC);
      const double y = (double)j * spM;
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
    const int cend = (int)(q1 - q0 + 1);

    float accr[3][3], acci[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int x = 0; x < 3; ++x) accr[a][x] = acci[a][x] = 0.0f;

    const int iters = (cend + kW - 1) / kW;
    const int c0 = lane;
    float wc = 1.0f, ws = 0.0f;
    if (iters > 0) {
      constexpr int CB = SPL * Fmt<MODE>::bps;
      const int i00 = (int)((q0 + c0) * SPL - s0);
      Fx fx[NS];
      const double isp = __dmul_rn((double)i00, sp);
      fx[0] = to_fx(__dmul_rn(__dadd_rn(aE, isp), M));
      fx[1] = to_fx(__dmul_rn(__dadd_rn(aP, isp), M));
      fx[2] = to_fx(__dmul_rn(__dadd_rn(aL, isp), M));
      const uint8_t* __restrict__ base = p.if_base;
      unsigned int glo[NS], ghi[NS];
      int kk[NS];
#pragma unroll
      for (int sx = 0; sx < NS; ++sx) {
        glo[sx] = (unsigned int)fx[sx].G;
        ghi[sx] = (unsigned int)(fx[sx].G >> 32);
        kk[sx] = fx[sx].k0;
      }
      const unsigned int Dlo = (unsigned int)Df, Dhi = (unsigned int)(Df >> 32);
      const uint8_t* __restrict__ bs = base + (long long)CB * q0;
      const unsigned int voff = (unsigned int)lane * CB;
      const unsigned int voff_last = min(voff, (unsigned int)(cend - 1 - (iters - 1) * kW) * CB);

      auto load_k = [&](const int k, unsigned int (&w)[NW]) {
        const uint8_t* __restrict__ pk = bs + (size_t)k * (size_t)(kW * CB);
        const unsigned int off = (k == iters - 1) ? voff_last : voff;
        load_words<MODE, SPL>(pk + off, 0, w);
      };

      auto process = [&](unsigned int (&w)[NW], const int k) {
        const bool last = (k == iters - 1);
        if ((k == 0) | last) {
          int kq = k;
          asm volatile("" : "+v"(kq));
          const int i0 = i00 + kq * (SPL * kW);
          if (last && c0 + kq * kW >= cend) {
#pragma unroll
            for (int q = 0; q < NW; ++q) w[q] = 0u;
#pragma unroll
            for (int sx = 0; sx < NS; ++sx) kk[sx] = 0;
          }
          if ((i0 < 0) | (i0 + SPL > N)) mask_words<MODE, SPL>(w, i0, N);
        }

        // BOC(1,1) transition positions u_n = (G + n) / (step*R*M), n = 0, 1, and the sign words of the six-fold arm
        float un[NS][2];
        unsigned int w6[NS];
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) {
          float gh;
          asm("v_cvt_f32_u32_e32 %0, %1" : "=v"(gh) : "v"(ghi[sx]));
          un[sx][0] = gh * uk;
          un[sx][1] = un[sx][0] + ustep;
          w6[sx] = ghi[sx] + (ghi[sx] << 1);
        }
        bool exact = false;
        if (!tie_free) {
          bool suspect = false;
#pragma unroll
          for (int sx = 0; sx < NS; ++sx)
#pragma unroll
            for (int n = 0; n < 2; ++n) suspect |= fabsf(un[sx][n] - rintf(un[sx][n])) < kTieTolM;
          if (!tie6_free) {
#pragma unroll
            for (int sx = 0; sx < NS; ++sx) {
              unsigned int wq = w6[sx];
#pragma unroll
              for (int j = 0; j < SPL; ++j) {
                suspect |= ((wq << 1) + kTie6) <= 2u * kTie6;   // 6g within kTie6 units of an integer
                wq += negD3;
              }
            }
          }
          exact = __any(suspect) != 0;
        }

        // Horner step of one sum: acc = acc * conj(rho) + U, rho = delta^(SPL*64) = rotC - i rotS
        auto horner = [&](int ar, int x, float ur, float ui) __attribute__((always_inline)) {
          const float nr = fmaf(accr[ar][x], rotC, fmaf(-acci[ar][x], rotS, ur));
          const float ni = fmaf(accr[ar][x], rotS, fmaf(acci[ar][x], rotC, ui));
          accr[ar][x] = nr;
          acci[ar][x] = ni;
        };
        if (exact) {
          float Ur[3][3], Ui[3][3];
          // ---- exact path: the reference's float64 index per sample and arm (rolled loop, samples re-read from memory) ------
#pragma unroll
          for (int ar = 0; ar < 3; ++ar)
#pragma unroll
            for (int x = 0; x < 3; ++x) Ur[ar][x] = Ui[ar][x] = 0.0f;
          int kq = k, Nq = N;
          asm volatile("" : "+v"(kq), "+v"(Nq));
          const int c = c0 + kq * kW;
          const bool act = c < cend;
          const int i0 = i00 + kq * (SPL * kW);
          const double nm1s = __dmul_rn((double)(Nq - 1), step);
          const double bP = __dmul_rn(__dadd_rn(nm1s, rem), R);
          const double bE = __dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), -d), R);
          const double bL = __dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), d), R);
          const uint8_t* sp8 = base + (long long)CB * (q0 + min(c, cend - 1));
          float cr = 1.0f, ci = 0.0f;
#pragma unroll 1
          for (int j = 0; j < SPL; ++j) {
            const int i = i0 + j;
            const float x0 = (float)(signed char)sp8[2 * j];
            const float x1 = (float)(signed char)sp8[2 * j + 1];
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
              const float cd = (float)(signed char)e, cp = (float)(signed char)(e >> 8);
              Ur[0][x] = fmaf(cd, yr, Ur[0][x]);
              Ui[0][x] = fmaf(cd, yi, Ui[0][x]);
              Ur[1][x] = fmaf(cp, yr, Ur[1][x]);
              Ui[1][x] = fmaf(cp, yi, Ui[1][x]);
              // arm 2: entry k6 = ceil(6t) of the six-fold table = arm-1 entry (k6 + 5) / 6 times (-1)^(that + k6)
              const int k6 = (int)fmin(fmax(ceil(__dmul_rn(t, M6)), 0.0), (double)(6 * (nmax - 2) + 1));
              const int pidx = (int)(((float)(k6 + 5) + 0.5f) * 0.16666667f);
              const float c1 = (float)(signed char)(table_word(pidx) >> 8);
              const float c6 = ((pidx + k6) & 1) ? -c1 : c1;
              Ur[2][x] = fmaf(c6, yr, Ur[2][x]);
              Ui[2][x] = fmaf(c6, yi, Ui[2][x]);
            }
          }
#pragma unroll
          for (int ar = 0; ar < 3; ++ar)
#pragma unroll
            for (int x = 0; x < 3; ++x) horner(ar, x, Ur[ar][x], Ui[ar][x]);
        } else {
          // ---- phase A: the carrier-wiped samples y - KEPT IN REGISTERS for the three sign phases below - and their running sums
          // P, parked as they are formed (row j = sample j: {re, im} x 64 lanes)
          float yr[SPL], yi[SPL];
          float Tr = 0.f, Ti = 0.f;
          static_for<0, SPL / 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g != 0) __builtin_amdgcn_sched_barrier(0);
            float pr[4], pi[4];
            static_for<0, 4>([&](auto rc) {
              constexpr int r = decltype(rc)::value, j = 4 * g + r;
              float a, b;
              sample_ab<MODE, j, NW>(w, a, b);
              yr[j] = fmaf(a, C[j], b * S[j]);
              yi[j] = fmaf(-a, S[j], b * C[j]);
              Tr += yr[j];
              Ti += yi[j];
              pr[r] = Tr;
              pi[r] = Ti;
            });
            park8<4 * g>(pr, pi, pfx_m0);
          });
          // every tap's two transition rows of P, read BEFORE the rows are reused (a wave's LDS operations complete in issue order)
          const float* prow[NS][2];
          float Q[NS][2][2];
#pragma unroll
          for (int sx = 0; sx < NS; ++sx)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              prow[sx][n] = pfx + (kRow * min((int)un[sx][n], SPL - 1)) * kW + lane;
              Q[sx][n][0] = prow[sx][n][0];
              Q[sx][n][1] = prow[sx][n][kW];
            }
          // ---- phases E, P, L: one tap's sign-modulated running sums P6 at a time through the SAME rows (8 KB per wave instead of
          // the 32 KB all four streams took side by side: twelve to sixteen waves per CU instead of four).  A tap's two rows of P6 are
          // read as soon as its phase has parked them and USED one phase later (the next tap's sample loop runs while the reads are
          // on their way: a wave's LDS operations complete in issue order, so the next phase's stores cannot overtake them)
          float q6[NS][2][2], t6r[NS], t6i[NS];
          auto tap_sums = [&](auto sc) {
            constexpr int sx = decltype(sc)::value;
            float cd[3], cp[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
              const unsigned int e = table_word(kk[sx] + n);
              cd[n] = cvt_byte<0>(e);
              cp[n] = cvt_byte<1>(e);
            }
            // sum_n c_n (Q_n - Q_{n-1}) = (c_0 - c_1) Q_0 + (c_1 - c_2) Q_1 + c_2 T
            const float ed0 = cd[0] - cd[1], ed1 = cd[1] - cd[2];
            const float ep0 = cp[0] - cp[1], ep1 = cp[1] - cp[2];
            horner(0, sx, fmaf(ed0, Q[sx][0][0], fmaf(ed1, Q[sx][1][0], cd[2] * Tr)), fmaf(ed0, Q[sx][0][1], fmaf(ed1, Q[sx][1][1], cd[2] * Ti)));
            horner(1, sx, fmaf(ep0, Q[sx][0][0], fmaf(ep1, Q[sx][1][0], cp[2] * Tr)), fmaf(ep0, Q[sx][0][1], fmaf(ep1, Q[sx][1][1], cp[2] * Ti)));
            // six-fold arm: coefficients c_n (-1)^(k0 + n): differences (-1)^k0 (c_0 + c_1), -(-1)^k0 (c_1 + c_2), last (-1)^k0 c_2
            const float sig = __uint_as_float(((unsigned int)kk[sx] << 31) | 0x3f800000u);
            const float e60 = sig * (cp[0] + cp[1]);
            const float e61 = -sig * (cp[1] + cp[2]);
            const float c62 = sig * cp[2];
            horner(2, sx, fmaf(e60, q6[sx][0][0], fmaf(e61, q6[sx][1][0], c62 * t6r[sx])), fmaf(e60, q6[sx][0][1], fmaf(e61, q6[sx][1][1], c62 * t6i[sx])));
          };
          static_for<0, NS>([&](auto sc) {
            constexpr int sx = decltype(sc)::value;
            float ar = 0.f, ai = 0.f;
            unsigned int wq = w6[sx];
            static_for<0, SPL / 4>([&](auto gc) {
              constexpr int g = decltype(gc)::value;
              float pr[4], pi[4];
              static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value, j = 4 * g + r;
                const float sg = __uint_as_float((wq & 0x80000000u) | 0x3f800000u);
                ar = fmaf(sg, yr[j], ar);
                ai = fmaf(sg, yi[j], ai);
                pr[r] = ar;
                pi[r] = ai;
                wq += negD3;
              });
              park8<4 * g>(pr, pi, pfx_m0);
            });
            t6r[sx] = ar;
            t6i[sx] = ai;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              q6[sx][n][0] = prow[sx][n][0];
              q6[sx][n][1] = prow[sx][n][kW];
            }
            if constexpr (sx > 0) tap_sums(std::integral_constant<int, sx - 1>{});
          });
          tap_sums(std::integral_constant<int, NS - 1>{});
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
      const double ph = blk.rem_carr_phase * 0.15915494309189535 + (double)(i00 + (iters - 1) * (SPL * kW)) * tau;
      sincospif(2.0f * (float)(ph - floor(ph)), &ws, &wc);
    }

    // ---- rotate into the absolute frame and reduce across the wavefront (DPP) ------------------------
    double* o = p.out + lb * GC_OUT_STRIDE;
    float tot[18];
#pragma unroll
    for (int ar = 0; ar < 3; ++ar)
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        tot[ar * 6 + 2 * x] = wave_sum_lane63(wc * accr[ar][x] + ws * acci[ar][x]);
        tot[ar * 6 + 2 * x + 1] = wave_sum_lane63(wc * acci[ar][x] - ws * accr[ar][x]);
      }
    if (lane == 63) {
#pragma unroll
      for (int v = 0; v < 18; ++v) o[v] = (double)tot[v];
      for (int v = 18; v < GC_OUT_STRIDE; ++v) o[v] = 0.0;
    }
  }  // bpw loop
}

template <int MODE, int NWV>
void launch_cboc_one(gc_context* ctx, const KArgs& a, dim3 grid, size_t smem) {
  const void* fn = reinterpret_cast<const void*>(corr_epl_cboc_kernel<MODE, NWV>);
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((corr_epl_cboc_kernel<MODE, NWV>), grid, dim3(NWV * kW), smem, ctx->stream, a);
}

template <int NWV>
void launch_cboc_waves(gc_context* ctx, const KArgs& a, dim3 grid, size_t smem) {
  if (ctx->if_layout == GC_QI) launch_cboc_one<I8_QI, NWV>(ctx, a, grid, smem);
  else launch_cboc_one<I8_IQ, NWV>(ctx, a, grid, smem);
}

}  // namespace

// Wavefronts per workgroup: the most of {4, 3, 2, 1} whose LDS (two interleaved int8 tables + 32 KB of running sums per wave)
// fits a CU; 0 = not even one (GC_CBOC_WAVES overrides)
int gc_cboc_waves(const gc_context* ctx) {
  const int tb = gc_multi_table_bytes(ctx->max_stage_len, 2);
  int forced = 0;
  if (const char* e = GC_TUNE_ENV("GC_CBOC_WAVES")) forced = std::atoi(e);
  for (int w : {16, 12, 8, 6, 4, 2, 1}) {
    if (tb + w * kWaveLds > kMaxLds) continue;
    if (forced == 0 || forced == w) return w;
  }
  return 0;
}

bool gc_cboc_takes(const gc_context* ctx, long long nblocks, int period) {
  if (!(ctx->scope_kt6 >= 1 && period > 0 && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL)) return false;
  const long long waves = gc_cboc_waves(ctx);
  if (waves <= 0) return false;
  // A wave takes one epoch and a workgroup fills a CU, so the launch runs in rounds of waves x CUs epochs and a part-filled last round
  // costs a whole one; a single round is as long as its slowest wave (a channel's first block sits on exact chip edges and takes the
  // float64 path chunk after chunk: ~0.2 ms more).  Measured on config 3's shape (eight channels, 1 - 10 s: lane kernel 97 ns per block;
  // the hybrid 0.47 ms for one round, 0.255 ms per round from two on): ahead from two rounds at least two thirds full on (3 s: 0.52
  // against 0.57 ms), up to 58 % behind below (1.5 s = 0.72 rounds: 0.47 / 0.30; 2.1 s = 1.02 rounds: 0.51 / 0.41).
  const long long cus = ctx->compute_units;
  const long long wgs = ((nblocks / period + waves - 1) / waves) * period;
  const long long rounds = (wgs + cus - 1) / cus;
  return rounds >= 2 && 3 * nblocks >= 2 * rounds * cus * waves;
}

// Periodic replay lists of int8 I/Q (Q/I) records whose channels are three-arm channels with a derived six-fold arm
// (gc_channel_is_derived), base ramp with at most two table transitions per 16-sample chunk; a.bpw = a multiple of `waves`,
// a.stride = the list's period, a.splits == 1.
int gc_launch_correlator_cboc(gc_context* ctx, const KArgs& a_in, unsigned int grid, int waves) {
  KArgs a = a_in;
  a.red_off = gc_multi_table_bytes(ctx->max_stage_len, 2);
  const size_t smem = (size_t)a.red_off + (size_t)waves * kWaveLds;
  if (waves == 16) launch_cboc_waves<16>(ctx, a, dim3(grid), smem);
  else if (waves == 8) launch_cboc_waves<8>(ctx, a, dim3(grid), smem);
  else if (waves == 12) launch_cboc_waves<12>(ctx, a, dim3(grid), smem);
  else if (waves == 6) launch_cboc_waves<6>(ctx, a, dim3(grid), smem);
  else if (waves == 4) launch_cboc_waves<4>(ctx, a, dim3(grid), smem);
  else if (waves == 2) launch_cboc_waves<2>(ctx, a, dim3(grid), smem);
  else launch_cboc_waves<1>(ctx, a, dim3(grid), smem);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
