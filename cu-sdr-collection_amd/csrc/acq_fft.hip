// acq_fft.hip — the transforms of the FFT-based parallel code-phase search (acquisition.m:151-200): plans, pass kernels, launch_pass.
// (The searches themselves: acq_coarse.hip, acq_shift.hip; fine-frequency stages: acq_fine.hip; all on gfx950.)
//
// Reference per PRN, bin b, hop h (acquisition.m:167-191):
//     results(b,:) += abs(ifft( fft( exp(-1i*f_b*phasePoints) .* x[h*spc : (h+2)*spc) ) .* conj(fft([code zeros]))))
// What is done differently (same arithmetic contract, float32 transforms):
//   * the signal spectra depend on (b, h) only, so they are computed ONCE (nbins*H transforms) and
//     reused by every PRN — the reference recomputes them for each of the 32 PRNs;
//   * N = 2*spc (36 000 at the default front end) is not a power of two: a four-step
//     (N = N1 x N2) mixed-radix {5,4,3,2} Stockham FFT, each pass a tile of short vectors
//     transformed in LDS by one workgroup, twiddles from a float64-computed table;
//   * int8 -> float conversion, carrier mixing, the product with the conjugated code spectrum,
//     the twiddles, abs() and the non-coherent sum over hops are fused into the passes;
//   * the peak pick reproduces max(max(.)) first-occurrence semantics with exact float compares.
//
// This file: the transform plans, the pass kernels (run-time fft_pass_kernel, per-shape fft_pass_ct, the fused kernel of the tuning
// build), launch_pass / forward.  The searches that use them: acq_coarse.hip, acq_shift.hip.
#include "acq_internal.h"

using namespace gcacq;

namespace {
// |z| of one output of an inverse transform (acquisition.m:187 abs(ifft(..))): v_sqrt_f32 as the hardware rounds it (1 ulp).  sqrtf()
// expands to the instruction plus a denormal pre-scale and two correction steps - 15 VALU instructions per element, half of the
// issue cycles of a columns pass - to move a float32 sum of squares that is itself ~1e-6 relative from the float64 reference by half
// an ulp (GC_ACQ_IEEE_SQRT=1 at build time: the correctly rounded one).
#ifndef GC_ACQ_IEEE_SQRT
#define GC_ACQ_IEEE_SQRT 0
#endif
__device__ __forceinline__ float cabs_f(float x, float y) {
  const float s = x * x + y * y;
  return GC_ACQ_IEEE_SQRT ? sqrtf(s) : __builtin_amdgcn_sqrtf(s);
}

// Radices a stage can take: 2, 3, 4, 5 directly, the others as two nested butterflies with compile-time inner twiddles
// (butterfly<R> below).  A pass spends most of its time between stages (LDS round trip, barrier, index arithmetic), so
// the plan is the factorisation with the FEWEST stages; among those the one whose largest radix is smallest (registers).
// Largest radix compiled into the pass kernel.  Measured (default L1 C/A search, MI355X): stages with radices up to 20
// halve the stage count of the 180- and 200-point passes but need 162 VGPRs (3 waves per SIMD instead of the 4 the
// tile's LDS allows) and the search gets 10 % SLOWER; up to 8 stays at 112 VGPRs and is 1 % faster than {5,4,3,2}.
#ifndef GC_FFT_MAXR
#define GC_FFT_MAXR 8
#endif
constexpr int kRadixSet[] = {20, 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};

int max_radix() {
  static const int m = [] {
    const char* e = GC_TUNE_ENV("GC_ACQ_MAX_RADIX");  // tuning: largest radix a stage may take (2 .. 20)
    return std::min(GC_FFT_MAXR, e ? std::max(5, std::atoi(e)) : 20);
  }();
  return m;
}

bool factor_rec(int r, int depth, int maxr, int* cur, int* best, int* best_n, int* best_max) {
  if (r == 1) {
    if (depth < *best_n || (depth == *best_n && maxr < *best_max)) {
      *best_n = depth;
      *best_max = maxr;
      for (int i = 0; i < depth; ++i) best[i] = cur[i];
    }
    return true;
  }
  if (depth >= kMaxRadices || depth + 1 > *best_n) return false;
  bool any = false;
  for (int c : kRadixSet) {
    if (r % c || c > max_radix()) continue;
    if (depth > 0 && c > cur[depth - 1]) continue;  // non-increasing: each multiset once
    cur[depth] = c;
    any |= factor_rec(r / c, depth + 1, std::max(maxr, c), cur, best, best_n, best_max);
  }
  return any;
}

bool factor(int len, SubPlan* sp) {
  sp->len = len;
  sp->nrad = 0;
  if (len == 1) return true;
  int cur[kMaxRadices], best[kMaxRadices], best_n = kMaxRadices + 1, best_max = 1 << 30;
  const bool simple = GC_TUNE_ENV("GC_ACQ_SIMPLE_RADIX") != nullptr;  // tuning: radices 5, 4, 3, 2 only
  if (simple) {
    int r = len;
    for (int c : {5, 4, 3, 2})
      while (r % c == 0) {
        if (sp->nrad >= kMaxRadices) return false;
        sp->rad[sp->nrad++] = c;
        r /= c;
      }
    return r == 1;
  }
  factor_rec(len, 0, 1, cur, best, &best_n, &best_max);
  if (best_n > kMaxRadices) return false;
  sp->nrad = best_n;
  for (int i = 0; i < best_n; ++i) sp->rad[i] = best[i];
  return true;
}
}  // namespace
namespace gcacq {
bool make_plan(int n, Plan* pl) {
  int best = 1;
  for (int d = 1; (long long)d * d <= n; ++d)
    if (n % d == 0) best = d;
  // sizes whose most square split is not the fastest one: GPS L2C's 320 000 points as 320 x 1 000 instead of 512 x 625 - the columns pass
  // reads 64-byte row segments (tiles of 8 columns in the same LDS) instead of 40-byte ones, 1.18 -> 1.00 ms per PRN; the rows pass has
  // a stage more, 0.83 -> 0.97 ms; the search 65.6 -> 62.5 ms.  GC_ACQ_PLAN_N1=<n>:<n1> tries another split (run-time pass kernels)
  static const int kSplit[][2] = {{320000, 320}};
  for (const auto& k : kSplit)
    if (n == k[0]) best = k[1];
  if (const char* e = GC_TUNE_ENV("GC_ACQ_PLAN_N1")) {
    int en = 0, e1 = 0;
    if (std::sscanf(e, "%d:%d", &en, &e1) == 2 && en == n && e1 > 0 && n % e1 == 0) best = e1;
  }
  pl->n = n;
  pl->n1 = best;
  pl->n2 = n / best;
  return factor(pl->n1, &pl->p1) && factor(pl->n2, &pl->p2) && pl->n2 <= kMaxPassLen;
}
}  // namespace gcacq
namespace {
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// floor(i / d) for 0 <= i < 2^21 through the float reciprocal of d (exact: (i + 0.5) / d is at least 0.5 / d away from
// every integer, far more than the float rounding of the product) - the runtime divisors of the index arithmetic would
// otherwise cost a software division each
__device__ __forceinline__ int fdiv_small(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

// Radix-R DFT of (already twiddled) inputs, sign = +1: exp(-i..) (forward), -1: inverse.  Radix 2 and 4 need no
// multiplications, 3 and 5 the classical real-constant forms (the generic R x R complex product they replace was the
// passes' VALU bound).
__device__ __forceinline__ float2 mul_mi(float2 a, float s) { return make_float2(s * a.y, -s * a.x); }  // a * (-i*s)
// cos / sin of 2*pi*m/R at compile time (Taylor series on an argument reduced to [-pi, pi]; double, rounded once to float)
constexpr double cx_angle(int m, int R) {
  const double t = 6.283185307179586476925286766559 * (double)(m % R) / (double)R;
  return t > 3.14159265358979323846 ? t - 6.283185307179586476925286766559 : t;
}
constexpr double cx_cos(int m, int R) {
  const double x = cx_angle(m, R);
  double term = 1.0, sum = 1.0;
  for (int n = 1; n < 20; ++n) {
    term *= -x * x / (double)((2 * n - 1) * (2 * n));
    sum += term;
  }
  return sum;
}
constexpr double cx_sin(int m, int R) {
  const double x = cx_angle(m, R);
  double term = x, sum = x;
  for (int n = 1; n < 20; ++n) {
    term *= -x * x / (double)((2 * n) * (2 * n + 1));
    sum += term;
  }
  return sum;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// composite radices R = A * B: B inner butterflies of size A, compile-time twiddles W_R^(q2*k1), A butterflies of size B
template <int R> struct RadixSplit { static constexpr int a = 0, b = 0; };
template <> struct RadixSplit<6> { static constexpr int a = 3, b = 2; };
template <> struct RadixSplit<8> { static constexpr int a = 4, b = 2; };
// (9 and 10 are always there: the fused search kernel below uses them whatever the planner's limit is)
template <> struct RadixSplit<9> { static constexpr int a = 3, b = 3; };
template <> struct RadixSplit<10> { static constexpr int a = 5, b = 2; };
#if GC_FFT_MAXR >= 12
template <> struct RadixSplit<12> { static constexpr int a = 4, b = 3; };
#endif
#if GC_FFT_MAXR >= 15
template <> struct RadixSplit<15> { static constexpr int a = 5, b = 3; };
#endif
#if GC_FFT_MAXR >= 16
template <> struct RadixSplit<16> { static constexpr int a = 4, b = 4; };
#endif
#if GC_FFT_MAXR >= 20
template <> struct RadixSplit<20> { static constexpr int a = 5, b = 4; };
#endif

template <int R>
__device__ __forceinline__ void butterfly(const float2 (&v)[R], float s, float2 (&o)[R]) {
  if constexpr (RadixSplit<R>::a != 0) {
    // X[k1 + A*k2] = sum_q2 W_B^(q2*k2) * W_R^(q2*k1) * sum_q1 v[q1*B + q2] * W_A^(q1*k1)
    constexpr int A = RadixSplit<R>::a, B = RadixSplit<R>::b;
    float2 t[B][A];
    static_for<0, B>([&](auto q2c) __attribute__((always_inline)) {
      constexpr int q2 = decltype(q2c)::value;
      float2 in[A], out[A];
#pragma unroll
      for (int q1 = 0; q1 < A; ++q1) in[q1] = v[q1 * B + q2];
      butterfly<A>(in, s, out);
      static_for<0, A>([&](auto k1c) __attribute__((always_inline)) {
        constexpr int k1 = decltype(k1c)::value;
        constexpr int m = (q2 * k1) % R;
        if constexpr (m == 0) {
          t[q2][k1] = out[k1];
        } else if constexpr ((4 * m) % R == 0) {  // quarter turns: W = (-i*s)^(4m/R)
          constexpr int qt = 4 * m / R;
          if constexpr (qt == 1) t[q2][k1] = mul_mi(out[k1], s);
          else if constexpr (qt == 2) t[q2][k1] = make_float2(-out[k1].x, -out[k1].y);
          else t[q2][k1] = mul_mi(out[k1], -s);
        } else {
          constexpr float c = (float)cx_cos(m, R), sn = (float)cx_sin(m, R);
          const float wy = -s * sn;  // table convention: exp(-i..) for s = +1
          t[q2][k1] = make_float2(out[k1].x * c - out[k1].y * wy, out[k1].x * wy + out[k1].y * c);
        }
      });
    });
    static_for<0, A>([&](auto k1c) __attribute__((always_inline)) {
      constexpr int k1 = decltype(k1c)::value;
      float2 in[B], out[B];
#pragma unroll
      for (int q2 = 0; q2 < B; ++q2) in[q2] = t[q2][k1];
      butterfly<B>(in, s, out);
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) o[k1 + A * k2] = out[k2];
    });
  } else if constexpr (R == 2) {
    o[0] = make_float2(v[0].x + v[1].x, v[0].y + v[1].y);
    o[1] = make_float2(v[0].x - v[1].x, v[0].y - v[1].y);
  } else if constexpr (R == 4) {
    const float2 t0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 t2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
    const float2 t3 = mul_mi(make_float2(v[1].x - v[3].x, v[1].y - v[3].y), s);
    o[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
    o[1] = make_float2(t1.x + t3.x, t1.y + t3.y);
    o[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
    o[3] = make_float2(t1.x - t3.x, t1.y - t3.y);
  } else if constexpr (R == 3) {
    const float2 t = make_float2(v[1].x + v[2].x, v[1].y + v[2].y);
    const float2 d = make_float2(v[1].x - v[2].x, v[1].y - v[2].y);
    const float2 m = make_float2(fmaf(-0.5f, t.x, v[0].x), fmaf(-0.5f, t.y, v[0].y));
    const float2 n = mul_mi(make_float2(0.8660254037844386f * d.x, 0.8660254037844386f * d.y), s);
    o[0] = make_float2(v[0].x + t.x, v[0].y + t.y);
    o[1] = make_float2(m.x + n.x, m.y + n.y);
    o[2] = make_float2(m.x - n.x, m.y - n.y);
  } else {
    static_assert(R == 5, "radices 2, 3, 4, 5 and their pairwise products up to 20");
    constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749473f, s1 = 0.9510565162951535f, s2 = 0.5877852522924731f;
    const float2 a1 = make_float2(v[1].x + v[4].x, v[1].y + v[4].y), a2 = make_float2(v[2].x + v[3].x, v[2].y + v[3].y);
    const float2 b1 = make_float2(v[1].x - v[4].x, v[1].y - v[4].y), b2 = make_float2(v[2].x - v[3].x, v[2].y - v[3].y);
    const float2 m1 = make_float2(fmaf(c2, a2.x, fmaf(c1, a1.x, v[0].x)), fmaf(c2, a2.y, fmaf(c1, a1.y, v[0].y)));
    const float2 m2 = make_float2(fmaf(c1, a2.x, fmaf(c2, a1.x, v[0].x)), fmaf(c1, a2.y, fmaf(c2, a1.y, v[0].y)));
    const float2 n1 = mul_mi(make_float2(fmaf(s2, b2.x, s1 * b1.x), fmaf(s2, b2.y, s1 * b1.y)), s);
    const float2 n2 = mul_mi(make_float2(fmaf(-s1, b2.x, s2 * b1.x), fmaf(-s1, b2.y, s2 * b1.y)), s);
    o[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
    o[1] = make_float2(m1.x + n1.x, m1.y + n1.y);
    o[4] = make_float2(m1.x - n1.x, m1.y - n1.y);
    o[2] = make_float2(m2.x + n2.x, m2.y + n2.y);
    o[3] = make_float2(m2.x - n2.x, m2.y - n2.y);
  }
}

// One radix-R Stockham stage of the tile in LDS, R a compile-time constant: the R inputs live in registers.
template <int R>
__device__ __forceinline__ void fft_stage(const float2* __restrict__ tw, int n, const float2* src, float2* dst, float2* twl,
                                          int L, int C, int ns, int tid, float sign) {
  const int lr = L / R;
  const int tws = n / (ns * R);  // table stride for W_{ns*R}
  // this stage's twiddles W_{ns*R}^{k*q} (k < ns, 0 < q < R) from the global table into LDS once per tile: the butterflies'
  // own lookups were scattered 8-byte global loads, the dominant cost of the pass
  if (ns > 1) {
    for (int i = tid; i < ns * (R - 1); i += kFftThreads) {
      const int k = i / (R - 1), q = i % (R - 1) + 1;
      float2 w = tw[k * q * tws];  // k*q*tws < ns*R*tws = n
      w.y *= sign;
      twl[i] = w;
    }
    __syncthreads();
  }
  const float inv_lr = 1.0f / (float)lr, inv_ns = 1.0f / (float)ns;
  for (int idx = tid; idx < lr * C; idx += kFftThreads) {
    const int c = fdiv_small(idx, inv_lr);
    const int j = idx - __mul24(c, lr);
    const int k = j - __mul24(fdiv_small(j, inv_ns), ns);
    const int cL = __mul24(c, L);
    float2 vq[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      float2 x = src[cL + j + q * lr];
      if (k != 0 && q != 0) x = cmul(x, twl[k * (R - 1) + q - 1]);
      vq[q] = x;
    }
    const int obase = cL + (j - k) * R + k;
    float2 oq[R];
    butterfly<R>(vq, sign, oq);
#pragma unroll
    for (int q = 0; q < R; ++q) dst[obase + q * ns] = oq[q];
  }
}

// One workgroup: `cols` vectors of length L, Stockham autosort in LDS (ping-pong), one output
// element group (j, column) per thread per stage.
__global__ __launch_bounds__(kFftThreads) void fft_pass_kernel(const PassArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int L = a.len, C = a.cols;
  float2* buf0 = lds;
  float2* buf1 = lds + (size_t)L * C;
  float2* twl = lds + (size_t)2 * L * C;  // [L] stage twiddles
  const int tiles = (a.nvec + C - 1) / C;
  const int tile = blockIdx.x % tiles;
  const int HG = (a.post == POST_ABS_ACC && a.hop_groups > 1) ? a.hop_groups : 1;
  const long long bb = blockIdx.x / tiles;
  const long long batch = bb / HG;
  const int hg = (int)(bb - batch * HG);
  const int v0 = tile * C;
  const int tid = threadIdx.x;
  const int nel = L * C;
  const float sign = a.inverse ? -1.0f : 1.0f;  // table holds exp(-i..): conjugate for the inverse
  const float inv_L = 1.0f / (float)L, inv_C = 1.0f / (float)C;

  const int reps = (a.post == POST_ABS_ACC) ? a.nhops / HG : 1;
  // POST_ABS_ACC keeps its accumulators in registers across the hop loop
  float accv[kFftSlots];
#pragma unroll
  for (int k = 0; k < kFftSlots; ++k) accv[k] = 0.f;

  for (int rep = 0; rep < reps; ++rep) {
    const long long tb = (a.post == POST_ABS_ACC) ? batch * a.nhops + (long long)hg * reps + rep : batch;
    // ---- load tile (coalesced along whichever index is contiguous in memory) ------------------------
    for (int idx = tid; idx < nel; idx += kFftThreads) {
      int e, c;
      if (a.estride == 1) {
        c = fdiv_small(idx, inv_L);
        e = idx - c * L;
      } else {
        e = fdiv_small(idx, inv_C);
        c = idx - e * C;
      }
      const int v = v0 + c;
      float2 val = make_float2(0.f, 0.f);
      if (v < a.nvec) {
        const int pos = __mul24(e, a.estride) + __mul24(v, a.vstride);  // index within transform (< n <= 2^24)
        if (a.pre == PRE_IF_CARRIER) {
          // x[n] = (I + iQ) * exp(-1i * f_b * n*2*pi/fs)  (acquisition.m:169-181), batch = b*nhops + h
          const int b = (int)(tb / a.nhops), h = (int)(tb % a.nhops);
          int p2 = pos;
          bool live = true;
          if (a.wrap_len > 0 && pos >= a.wrap_len) {
            p2 = pos - a.wrap_len;
            live = p2 < a.spc;
            p2 = live ? p2 : 0;
          }
          const long long s = a.first_sample + (long long)h * a.spc + (long long)p2;
          float xi, xq;
          if (a.if_f32) {
            const float2 z = a.if_f32[s];
            xi = z.x;
            xq = z.y;
          } else {
            xi = (float)a.if_base[2 * s];
            xq = (float)a.if_base[2 * s + 1];
          }
          const double fb = a.f0 - a.fstep * b;
          const double ph = (fb / a.fs) * (double)p2;
          float sn, cs;
          sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
          val = live ? make_float2(xi * cs + xq * sn, xq * cs - xi * sn) : make_float2(0.f, 0.f);
        } else if (a.pre == PRE_CODE) {
          val = (pos < a.spc) ? make_float2((float)a.codes[tb * a.spc + pos], 0.f) : make_float2(0.f, 0.f);
        } else {
          if (a.pre == PRE_MUL_CONJ && (a.shift_bins > 0 || a.shift_q > 0)) {
            // circshift(X, s): Y[k] = X[(k - s) mod n] in natural frequency order; storage position of frequency
            // k = k1 + n1*k2 is k1*n2 + k2
            const int den = a.shift_den > 1 ? a.shift_den : 1, sbin = (int)(tb / a.nhops);
            const long long src = a.shift_q > 0 ? (long long)(sbin % den) * a.nhops + tb % a.nhops : tb / a.shift_bins;
            int sft = a.shift_q > 0 ? (sbin / den) * a.shift_q + a.shift0 : (int)(tb % a.shift_bins);
            if (sft >= a.n) sft -= a.n;
            const int k1 = (int)(pos / a.n2), k2 = (int)(pos % a.n2);
            int k = k1 + a.n1 * k2 - sft;
            if (k < 0) k += a.n;
            val = a.in[src * a.in_batch_stride + (long long)(k % a.n1) * a.n2 + k / a.n1];
          } else {
            val = a.in[tb * a.in_batch_stride + pos];
          }
          if (a.pre == PRE_MUL_CONJ) {
            const float2 o = a.other[pos];
            val = cmul(val, make_float2(o.x, -o.y));
          }
        }
      }
      buf0[__mul24(c, L) + e] = val;
    }
    __syncthreads();

    // ---- Stockham stages ---------------------------------------------------------------------------
    float2* src = buf0;
    float2* dst = buf1;
    int ns = 1;
    for (int s = 0; s < a.nrad; ++s) {
      const int r = a.rad[s];
      switch (r) {
        case 2: fft_stage<2>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
        case 3: fft_stage<3>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
        case 4: fft_stage<4>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
        case 5: fft_stage<5>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#if GC_FFT_MAXR >= 6
        case 6: fft_stage<6>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 8
        case 8: fft_stage<8>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 9
        case 9: fft_stage<9>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 10
        case 10: fft_stage<10>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 12
        case 12: fft_stage<12>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 15
        case 15: fft_stage<15>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 16
        case 16: fft_stage<16>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
#if GC_FFT_MAXR >= 20
        case 20: fft_stage<20>(a.tw, a.n, src, dst, twl, L, C, ns, tid, sign); break;
#endif
        default: break;
      }
      __syncthreads();
      float2* t = src;
      src = dst;
      dst = t;
      ns *= r;
    }

    // POST_TWIDDLE: W_n^(v*e) = W_n^(v*16*(e>>4)) * W_n^(v*(e&15)) from a per-tile LDS table (C * (EH + 16)
    // entries from the global table instead of one scattered 8-byte load per element: that gather was 13 % of a search)
    const int EH = ((L - 1) >> 4) + 1, TW2 = EH + 16;
    float2* tw2 = twl + L;  // its own LDS region (launch_pass sizes it)
    if (a.post == POST_TWIDDLE) {
      const float inv_tw2 = 1.0f / (float)TW2;
      for (int i = tid; i < C * TW2; i += kFftThreads) {
        const int c = fdiv_small(i, inv_tw2), j = i - c * TW2;
        const int v = v0 + c;
        if (v < a.nvec) {
          float2 w = a.tw[j < EH ? __mul24(v, j << 4) : __mul24(v, j - EH)];  // v * e < n for every e < L
          w.y *= sign;
          tw2[i] = w;
        }
      }
      __syncthreads();
    }

    // ---- store -------------------------------------------------------------------------------------------
#pragma unroll
    for (int slot = 0; slot < kFftSlots; ++slot) {
      const int idx = tid + slot * kFftThreads;
      if (idx >= nel) continue;
      int e, c;
      if (a.estride == 1) {
        c = fdiv_small(idx, inv_L);
        e = idx - c * L;
      } else {
        e = fdiv_small(idx, inv_C);
        c = idx - e * C;
      }
      const int v = v0 + c;
      if (v >= a.nvec) continue;
      float2 val = src[__mul24(c, L) + e];
      const int pos = __mul24(e, a.estride) + __mul24(v, a.vstride);
      if (a.post == POST_TWIDDLE) val = cmul(val, cmul(tw2[c * TW2 + (e >> 4)], tw2[c * TW2 + EH + (e & 15)]));
      if (a.post == POST_ABS_ACC) {
        accv[slot] += cabs_f(val.x, val.y);
      } else {
        a.out[tb * a.out_batch_stride + pos] = val;
      }
    }
    __syncthreads();
  }
  if (a.post == POST_ABS_ACC) {
    const float inv_n = 1.0f / (float)a.n;
#pragma unroll
    for (int slot = 0; slot < kFftSlots; ++slot) {
      const int idx = tid + slot * kFftThreads;
      if (idx >= nel) continue;
      int e, c;
      if (a.estride == 1) {
        c = fdiv_small(idx, inv_L);
        e = idx - c * L;
      } else {
        e = fdiv_small(idx, inv_C);
        c = idx - e * C;
      }
      const int v = v0 + c;
      if (v >= a.nvec) continue;
      const int pos = __mul24(e, a.estride) + __mul24(v, a.vstride);
      if (HG > 1) {
        a.acc_part[((long long)hg * a.acc_bins + batch) * a.n + pos] = accv[slot];
      } else {
        float* dstp = a.acc_out + batch * a.n + pos;
        *dstp = (a.acc_add ? *dstp : 0.0f) + accv[slot] * inv_n * (a.acc_scale != 0.0f ? a.acc_scale : 1.0f);
      }
    }
  }
}

// ---- pass kernels generated per shape ------------------------------------------------------------------------------
// fft_pass_kernel above takes every size at run time and pays for it: ~170 VALU instructions per element and pass, most
// of them index arithmetic (run-time divisors, strides, radix dispatch, bounds tests).  fft_pass_ct is the same pass with
// the vector length, the other dimension, the tile width, the radices, the pre/post operation and the direction as
// template parameters: divisions by constants, LDS addresses with immediate offsets, the stage twiddles of ALL stages
// staged once per workgroup (not per stage and hop), no bounds tests (the tile width divides the vector count), sign
// flips folded into the butterflies.  launch_pass picks it for the shapes listed in GC_CT_SHAPES (the FFT sizes of the
// reference's default front ends) and falls back to the generic kernel for everything else; GC_ACQ_GENERIC=1 forces the
// generic kernel.
// LP: pitch of a tile row of dst in LDS (L, or L + 1 in the fused-I/O columns pass); SP > 0: the SOURCE rows carry one pad element after
// every 2^SP (element i at i + (i >> SP), pitch SLP) - what a fused first stage of radix 2^SP leaves (stage_first_ct)
template <int NT, int R, int L, int C, int NS, bool INV, int LP = L, int SP = 0, int SLP = LP>
__device__ __forceinline__ void stage_ct(const float2* __restrict__ src, float2* __restrict__ dst, const float2* __restrict__ twl,
                                         unsigned tid) {
  constexpr unsigned LR = L / R, NB = LR * C, ITERS = (NB + NT - 1) / NT;
  constexpr float sign = INV ? -1.0f : 1.0f;
#pragma unroll
  for (unsigned it = 0; it < ITERS; ++it) {
    const unsigned b = tid + it * NT;
    if ((it + 1) * NT > NB && b >= NB) break;
    const unsigned c = b / LR, j = b - c * LR;
    const unsigned k = NS == 1 ? 0u : j % (unsigned)NS;
    const float2* s = src + c * SLP;
    auto at = [&](unsigned i) -> float2 { return SP > 0 ? s[i + (i >> SP)] : s[i]; };
    float2 vq[R], oq[R];
    vq[0] = at(j);
#pragma unroll
    for (int q = 1; q < R; ++q) {
      float2 x = at(j + q * LR);
      if constexpr (NS > 1) x = cmul(x, twl[(q - 1) * NS + k]);  // [q][k]: the lanes of a wave read consecutive k (k = 0 holds ones)
      vq[q] = x;
    }
    butterfly<R>(vq, sign, oq);
    float2* d = dst + c * LP + (j - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) d[q * NS] = oq[q];
  }
}

// Fused-I/O passes (GC_ACQ_FUSE_IO): the FIRST stage takes its inputs straight from memory (no store of the loaded tile into LDS and
// read back), the LAST stage hands its outputs to the pass's epilogue in registers (no store of the finished tile and read back):
// four LDS accesses per element of a three-stage pass instead of eight.  The passes of the search were LDS-bound: 8.0 / 6.3 LDS
// instructions per element at 2.0-2.5 bank-conflict cycles each (profiles/r03) are ~39 us of LDS time per launch against ~25 us of VALU.
// in(c, e) -> element e of the tile's vector c.  CFAST: consecutive threads take consecutive VECTORS of one butterfly index (the
// blocked intermediate of the columns pass is stored vector-fastest: one contiguous run per wave-load), the tile rows then sit LP = L + 1
// apart so that the radix-R groups the threads write do not pile onto a few banks.
// PAD: a thread's R outputs are followed by one pad element (row pitch LP = L + L / R): with R = 8 the threads' 64-byte groups
// would otherwise start 16 banks apart - two bank groups for 64 lanes, four conflict cycles per store.
template <int NT, int R, int L, int LP, int C, bool INV, bool CFAST, bool PAD, class F>
__device__ __forceinline__ void stage_first_ct(F&& in, float2* __restrict__ dst, unsigned tid) {
  constexpr unsigned LR = L / R, NB = LR * C, ITERS = (NB + NT - 1) / NT;
  constexpr float sign = INV ? -1.0f : 1.0f;
#pragma unroll
  for (unsigned it = 0; it < ITERS; ++it) {
    const unsigned b = tid + it * NT;
    if ((it + 1) * NT > NB && b >= NB) break;
    unsigned c, j;
    if constexpr (CFAST) {
      j = b / C;
      c = b - j * C;
    } else {
      c = b / LR;
      j = b - c * LR;
    }
    float2 vq[R], oq[R];
#pragma unroll
    for (int q = 0; q < R; ++q) vq[q] = in(it, q, c, j + q * LR);
    butterfly<R>(vq, sign, oq);
    float2* d = dst + c * LP + j * (PAD ? R + 1 : R);
#pragma unroll
    for (int q = 0; q < R; ++q) d[q] = oq[q];
  }
}

// f(it, q, c, e) for every input of a first stage, in stage_first_ct's (iteration, q) order: element e of vector c.  A pass that walks
// several hops fetches the NEXT hop's inputs into registers with this right after its first stage has consumed the current ones: the
// loads are in flight during the other stages (their barriers wait for LDS, not for memory) instead of every hop starting with a
// full memory latency in front of its first butterfly.
template <int NT, int R, int L, int C, bool CFAST, class F>
__device__ __forceinline__ void first_each_ct(unsigned tid, F&& f) {
  constexpr unsigned LR = L / R, NB = LR * C, ITERS = (NB + NT - 1) / NT;
#pragma unroll
  for (unsigned it = 0; it < ITERS; ++it) {
    const unsigned b = tid + it * NT;
    if ((it + 1) * NT > NB && b >= NB) break;
    unsigned c, j;
    if constexpr (CFAST) {
      j = b / C;
      c = b - j * C;
    } else {
      c = b / LR;
      j = b - c * LR;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) f(it, q, c, j + q * LR);
  }
}
// uniform base + a 32-bit byte offset per thread, as a buffer load: the base stays in scalar registers (a descriptor built per hop)
// and an address costs one VGPR that does not depend on the hop - with flat loads the compiler keeps a 64-bit address per input
// and adds the hop's stride to each (16 VGPRs and 8 64-bit adds for a radix-8 first stage; the rows pass spilled at six waves per SIMD)
__device__ __forceinline__ float2 ld_off(const float2* __restrict__ base, unsigned byte_off) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  const unsigned long long bu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu), 0, 0x7fffffff, 0x00020000);
  // (bit_cast of the whole vector: element-wise v[0], v[1] came out of this compiler as ONE buffer_load_dword used twice)
  return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}

// the last stage (NS = L / R: k = j): out(it, q, c, e, value) receives element e = j + q * NS of vector c
template <int NT, int R, int L, int LP, int C, bool INV, int SP, class F>
__device__ __forceinline__ void stage_last_ct(const float2* __restrict__ src, const float2* __restrict__ twl, unsigned tid, F&& out) {
  constexpr unsigned LR = L / R, NB = LR * C, ITERS = (NB + NT - 1) / NT;
  constexpr float sign = INV ? -1.0f : 1.0f;
#pragma unroll
  for (unsigned it = 0; it < ITERS; ++it) {
    const unsigned b = tid + it * NT;
    if ((it + 1) * NT > NB && b >= NB) break;
    const unsigned c = b / LR, j = b - c * LR;
    const float2* s = src + c * LP;
    auto at = [&](unsigned i) -> float2 { return SP > 0 ? s[i + (i >> SP)] : s[i]; };
    float2 vq[R], oq[R];
    vq[0] = at(j);
#pragma unroll
    for (int q = 1; q < R; ++q) vq[q] = cmul(at(j + q * LR), twl[(q - 1) * LR + j]);
    butterfly<R>(vq, sign, oq);
#pragma unroll
    for (int q = 0; q < R; ++q) out(it, q, c, j + q * LR, oq[q]);
  }
}

// W_{NS*R}^{k*q} (k < NS, 0 < q < R) of one stage from the global table exp(-2*pi*i*m/N)
template <int NT, int R, int NS, int N, bool INV>
__device__ __forceinline__ void stage_twiddles_ct(const float2* __restrict__ tw, float2* twl, unsigned tid) {
  if constexpr (R > 1 && NS > 1) {
    // stored [q - 1][k] (not [k][q - 1]): a butterfly's lanes have consecutive k, and R - 1 = 4 values of 8 bytes per k put every
    // fourth lane on the same banks - 2.7 / 4.0 conflict cycles per LDS instruction of the rows / columns pass (profiles/r04)
    constexpr unsigned CNT = NS * (R - 1), TWS = N / (NS * R);
    static_assert(N % (NS * R) == 0, "stage size divides the transform size");
    for (unsigned i = tid; i < CNT; i += NT) {
      const unsigned q = i / NS + 1, k = i - (q - 1) * NS;
      float2 w = tw[k * q * TWS];
      if (INV) w.y = -w.y;
      twl[i] = w;
    }
  }
}

// -DGC_ACQ_STAGE_CLOCKS=1 (scripts/acq_stage_clocks.py, a tuning build): wavefront w of every workgroup of the fused columns pass adds the
// shader-clock cycles it spent per hop in [wait for the prefetched tile + first stage | fetch issue + barrier | middle stage | barrier |
// last stage] to g_stage_clk[w * 8 + phase] (and the hops it counted to [w * 8 + 7]); the same for the fused rows pass from slot 64 on.
#ifdef GC_ACQ_STAGE_CLOCKS
__device__ unsigned long long g_stage_clk[128];
#define GC_CLK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); clk_acc[i] += t_ - clk_last; clk_last = t_; } while (0)
#else
#define GC_CLK(i) do { } while (0)
#endif

// Threads per workgroup of a specialised pass.  A stage of radix R has (L / R) * C butterflies, one per thread and iteration; with 256
// threads the 288 radix-5 butterflies of the 180 x 8 columns tile are two iterations for wavefront 0 (the second one for 32 lanes) and
// one for the others, and every barrier waits for wavefront 0: the workgroup's critical path is the SUM over the stages of
// ceil(butterflies / threads).  The passes are latency-bound, so the workgroup takes the smallest thread count up to
// the cap that minimises that sum; wavefronts without a butterfly in a stage skip it (the issue slots spent stay the same).
// The cap, measured per plan over the twelve default searches (-DGC_ACQ_NT_MAX=n applies one cap to every plan): 320 for the short
// vectors of the 36 000- and 24 000-point plans (columns pass of the default L1 C/A search 4 -> 3 iterations: 3.29 -> 3.06 ms sustained,
// L5 / E5a / E5b / B3I -5..-8 %), 512 for the 600 x 600 plan (8 -> 4: B1C 109 -> 104 ms; 320 gives 7 iterations and 137 ms), 256
// elsewhere (375 x 384, 250 x 288, 512 x 625: within the noise or slower with more wavefronts per tile).  With 512 threads the 600 x 600
// plan's columns tile is 5 columns wide (3 000 values, 40-byte tile rows instead of 24: B1C 99.7 -> 90.0 ms; 4 columns 91.4, 6 113, 8 - one
// workgroup per CU - 108; rows tiles of 2 / 4 / 5 rows instead of 3: 101.7 / 97.9 / 90.9 ms).
template <int L>
constexpr int ct_threads_cap() {
#ifdef GC_ACQ_NT_MAX
  return GC_ACQ_NT_MAX > kFftThreads ? GC_ACQ_NT_MAX : kFftThreads;
#else
  return L <= 200 ? 320 : L == 600 ? 512 : kFftThreads;
#endif
}
template <int L, int C, int R0, int R1, int R2, int R3>
constexpr int ct_threads() {
  const int rad[4] = {R0, R1, R2, R3};
  int best = kFftThreads, best_it = 1 << 30;
  for (int nt = kFftThreads; nt <= ct_threads_cap<L>(); nt += 64) {
    int it = 0;
    for (int r : rad)
      if (r > 1) it += ((L / r) * C + nt - 1) / nt;
    if (it < best_it) {
      best_it = it;
      best = nt;
    }
  }
  return best;
}

template <int L, int OTHER, bool CONTIG, int C, int PRE, int POST, bool INV, bool SHIFT, int R0, int R1, int R2, int R3>
#ifndef GC_ACQ_PASS_WAVES
#define GC_ACQ_PASS_WAVES 5
#endif
// second bound: wavefronts per SIMD the register allocation must leave room for - the tiles of the short passes (<= 26 KB of LDS) fit
// six workgroups per CU, and the passes are latency-bound (barriers between stages): the registers must not be what limits them
__global__ __launch_bounds__((ct_threads<L, C, R0, R1, R2, R3>()), (L <= 200 ? GC_ACQ_PASS_WAVES : 1)) void fft_pass_ct(const PassArgs a) {
  constexpr unsigned NT = ct_threads<L, C, R0, R1, R2, R3>();
  static_assert(R0 * R1 * R2 * R3 == L && OTHER % C == 0, "radices multiply to L; whole tiles only");
  static_assert(!SHIFT || (CONTIG && PRE == PRE_MUL_CONJ), "shifted reads belong to the rows pass of the inverse transform");
  constexpr unsigned N = L * OTHER, NEL = L * C, SLOTS = (NEL + NT - 1) / NT, TILES = OTHER / C;
  static_assert(SLOTS <= kFftSlots + 2, "tile too large");
  constexpr unsigned ESTR = CONTIG ? 1 : OTHER, VSTR = CONTIG ? L : 1;
  constexpr int NS1 = R0, NS2 = R0 * R1, NS3 = R0 * R1 * R2;
  constexpr unsigned T1 = R1 > 1 ? NS1 * (R1 - 1) : 0, T2 = R2 > 1 ? NS2 * (R2 - 1) : 0, T3 = R3 > 1 ? NS3 * (R3 - 1) : 0;
  constexpr int NST = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
  constexpr unsigned EH = ((L - 1) >> 4) + 1, TW2 = EH + 16;
#ifndef GC_ACQ_FUSE_IO
#define GC_ACQ_FUSE_IO 1
#endif
  // the two hot passes of the search - rows (product with the code spectrum -> twiddle) and columns (-> |.| summed over the hops) - with
  // their first stage fed from memory and their last stage feeding the epilogue (stage_first_ct / stage_last_ct)
  constexpr bool FUSE = GC_ACQ_FUSE_IO != 0 && NST >= 2 &&
                        ((PRE == PRE_MUL_CONJ && POST == POST_TWIDDLE) || (PRE == PRE_NONE && POST == POST_ABS_ACC));
  constexpr unsigned LP = (FUSE && !CONTIG) ? L + 1 : L;  // row pitch of the tile in LDS
  // fused rows pass with a first stage of radix 8 (or 4): its output rows are padded (stage_first_ct PAD), read back through SP1
  constexpr bool PAD1 = FUSE && CONTIG && (R0 == 8 || R0 == 4);
  constexpr int SP1 = PAD1 ? (R0 == 8 ? 3 : 2) : 0;
  constexpr unsigned LP1 = PAD1 ? L + L / R0 : LP;  // pitch of buf1's rows while they hold the first stage's output
  __shared__ __attribute__((aligned(16))) float2 buf0[C * LP];
  __shared__ __attribute__((aligned(16))) float2 buf1[C * LP1];
  __shared__ float2 twl[T1 + T2 + T3 + 1];
  __shared__ float2 tw2[POST == POST_TWIDDLE ? C * TW2 : 1];
  const unsigned tid = threadIdx.x;
  [[maybe_unused]] const unsigned sden = a.shift_den > 1 ? (unsigned)a.shift_den : 1u;  // (PassArgs::shift_den)
  // Strided (column) passes: a tile row is C consecutive float2 - 64 bytes at C = 8, half of a 128-byte line.  The neighbouring
  // tile reads the other half; consecutive workgroups go to consecutive XCDs, each with an L2 of its own, and both fetched
  // the whole line (rocprofv3 FETCH_SIZE: 328 MB per launch of the inverse columns pass for the 167 MB it reads).  Blocks b
  // and b + 8 of a group of 16 share an XCD and start together: they take neighbouring tiles.
  // Narrower tiles (24 bytes at C = 3: the 600 x 600 plan of BDS B1C; 40 at C = 5) share a line among five: every XCD takes a
  // contiguous run of (batch, tile) - consecutive tiles of a batch run on one XCD at about the same time and find each other's lines
  // in its L2 (B1C columns pass: FETCH_SIZE 1.42 GB per launch for the 0.58 GB it reads with the pairs only).
  unsigned bid = blockIdx.x;
  if constexpr (!CONTIG && (C * 8) % 128 != 0) {
    if (a.no_xcd_pairs == 2) {  // GC_ACQ_XCD_MAP=pairs: the pairing only
      const unsigned g = bid & ~15u;
      if (g + 16 <= gridDim.x) bid = g + ((bid & 7u) << 1) + ((bid >> 3) & 1u);
    } else if (a.no_xcd_pairs == 0) {
      const unsigned n8 = gridDim.x & ~7u;
      if (bid < n8) bid = (bid & 7u) * (n8 >> 3) + (bid >> 3);
    }
  }
  const unsigned tile = bid % TILES;
  const unsigned bb = bid / TILES;
  const unsigned HG = (POST == POST_ABS_ACC && a.hop_groups > 1) ? (unsigned)a.hop_groups : 1u;
  // BQ consecutive batches per workgroup (fused columns pass of a search without hops - the circshift family, Galileo E1: a
  // workgroup that lives for ONE tile of 1 800 values spends its life waiting for its twiddles, then for its tile)
  const unsigned BQ = (POST == POST_ABS_ACC && HG == 1 && a.bins_per_wg > 1) ? (unsigned)a.bins_per_wg : 1u;
  unsigned arm = 0, bbl = bb;  // (PassArgs::arm_batches)
  if constexpr (PRE == PRE_MUL_CONJ && POST == POST_TWIDDLE) {
    if (a.arm_batches > 0) {
      arm = fdiv(bb, a.fd_arm_batches);
      bbl = bb - arm * (unsigned)a.arm_batches;
    }
  }
  [[maybe_unused]] const float2* __restrict__ const other = a.other + (size_t)arm * N;
  // where transform tb of this launch goes in the intermediate
  [[maybe_unused]] auto out_tb = [&](long long tb) -> long long {
    if constexpr (PRE == PRE_MUL_CONJ && POST == POST_TWIDDLE) {
      if (a.arm_batches > 0) {
        const unsigned t = (unsigned)tb, q = fdiv(t, a.fd_nhops);
        return (long long)(q * (unsigned)a.narms_merged + arm) * a.nhops + (t - q * (unsigned)a.nhops);
      }
    }
    return tb;
  };
  const unsigned bgrp = HG == 1u ? bbl : fdiv(bbl, a.fd_hg);
  const unsigned hg = bbl - bgrp * HG, batch = bgrp * BQ + (unsigned)a.batch0;
  const unsigned nq = BQ == 1 ? 1u : min(BQ, (unsigned)a.nbatch_total - bgrp * BQ);
  const unsigned v0 = tile * C;
  constexpr bool RR = SHIFT && POST != POST_ABS_ACC;  // rows pass that may walk several hops of its bin (PassArgs::row_reps)
  const int reps = POST == POST_ABS_ACC ? (HG == 1u ? a.nhops : (int)fdiv((unsigned)a.nhops, a.fd_hg)) : (RR && a.row_reps > 1 && a.shift_q > 0) ? a.row_reps : 1;

  stage_twiddles_ct<NT, R1, NS1, N, INV>(a.tw, twl, tid);
  stage_twiddles_ct<NT, R2, NS2, N, INV>(a.tw, twl + T1, tid);
  stage_twiddles_ct<NT, R3, NS3, N, INV>(a.tw, twl + T1 + T2, tid);
  if constexpr (POST == POST_TWIDDLE) {
    // W_N^(v*e) = W_N^(v*16*(e>>4)) * W_N^(v*(e&15)): C * (EH + 16) table entries per tile
    for (unsigned i = tid; i < C * TW2; i += NT) {
      const unsigned c = i / TW2, j = i - c * TW2;
      const unsigned v = v0 + c;
      float2 w = a.tw[j < EH ? v * (j << 4) : v * (j - EH)];  // v * e < N for every e < L
      if (INV) w.y = -w.y;
      tw2[i] = w;
    }
  }

  float accv[SLOTS];
#pragma unroll
  for (unsigned k = 0; k < SLOTS; ++k) accv[k] = 0.f;

  // the sums of one batch (accv, in the tile's memory order): partial sums of a hop group, the results, or the workgroup's peak candidate
  auto finish = [&](unsigned batch_q, unsigned slot_id) {
    const float inv_n = 1.0f / (float)N, scale = a.acc_scale != 0.0f ? a.acc_scale : 1.0f;
    PeakTrack pk;
#pragma unroll
    for (unsigned slot = 0; slot < SLOTS; ++slot) {
      const unsigned idx = tid + slot * NT;
      if ((slot + 1) * NT > NEL && idx >= NEL) break;
      unsigned pos;
      if constexpr (CONTIG) {
        pos = v0 * L + idx;
      } else {
        const unsigned e = idx / C, c = idx - e * C;
        pos = e * ESTR + (v0 + c) * VSTR;
      }
      if (HG > 1) {
        a.acc_part[((long long)hg * a.acc_bins + batch_q) * N + pos] = accv[slot];
      } else {
        float* dstp = a.acc_out + (long long)(batch_q - a.acc_row0) * N + pos;
        const float v = (a.acc_add ? *dstp : 0.0f) + accv[slot] * inv_n * scale;
        if (a.peak_slots) {  // the finished sums of a PRN feed nothing but its peak keys
          if ((int)pos < a.peak_valid) pk.see(v, batch_q, pos);
        } else {
          *dstp = v;
        }
      }
    }
    if (HG == 1 && a.peak_slots) pk.publish_slot(a.peak_slots + 2 * (size_t)slot_id, a.peak_second ? a.peak_second + slot_id : nullptr);  // slot_id: the (batch, tile) after the XCD mapping, not blockIdx.x
  };

  if constexpr (FUSE) {
    constexpr int RL = R3 > 1 ? R3 : R2 > 1 ? R2 : R1;  // the last stage's radix; its inputs are L / RL apart
    constexpr unsigned LR0 = L / R0, NB0 = LR0 * C, IT0 = (NB0 + NT - 1) / NT;
    constexpr unsigned NBL = (L / RL) * C, ITL = (NBL + NT - 1) / NT;
    const float2* const twl_last = R3 > 1 ? twl + T1 + T2 : R2 > 1 ? twl + T1 : twl;
    [[maybe_unused]] float acc2[POST == POST_ABS_ACC ? ITL : 1][POST == POST_ABS_ACC ? RL : 1];
    // rows pass that walks several hops of one bin: where each of the thread's inputs comes from and the code-spectrum value it is
    // multiplied with do not depend on the hop
    [[maybe_unused]] unsigned fr_src[RR ? IT0 : 1][RR ? R0 : 1];
    [[maybe_unused]] float2 fr_oth[RR ? IT0 : 1][RR ? R0 : 1];
    if constexpr (RR) {
      const long long tb0 = (long long)batch * reps;
      unsigned sft = a.shift_q > 0 ? fdiv(fdiv((unsigned)tb0, a.fd_nhops), a.fd_sden) * (unsigned)a.shift_q + (unsigned)a.shift0 : fmodu((unsigned)tb0, a.fd_shift_bins);
      sft -= sft >= N ? N : 0u;
      const unsigned s2 = sft / OTHER, s1 = sft - s2 * OTHER;
#pragma unroll
      for (unsigned it = 0; it < IT0; ++it) {
        const unsigned b = tid + it * NT;
        if ((it + 1) * NT > NB0 && b >= NB0) break;
        const unsigned c = b / LR0, j = b - c * LR0;
        int k1 = (int)(v0 + c) - (int)s1;
        const int bor = k1 < 0;
        k1 += bor ? OTHER : 0;
#pragma unroll
        for (int q = 0; q < R0; ++q) {
          const unsigned e = j + q * LR0;
          int e2 = (int)e - (int)s2 - bor;
          e2 += e2 < 0 ? L : 0;
          fr_src[RR ? it : 0][RR ? q : 0] = (unsigned)(k1 * L + e2) * 8u;  // byte offset (ld_off)
          fr_oth[RR ? it : 0][RR ? q : 0] = other[(v0 + c) * L + e];
        }
      }
    }
    // ---- the first stage's inputs, one hop ahead (first_fetch_ct) ---------------------------------------------------------------
    auto tb_of = [&](unsigned bq, int rep) -> long long {
      return POST == POST_ABS_ACC ? (long long)bq * a.nhops + (long long)hg * reps + rep
             : RR                 ? (long long)bq * reps + rep
                                  : (long long)bq;
    };
    float2 pre[IT0][R0];
    [[maybe_unused]] float2 poth[(PRE == PRE_MUL_CONJ && !RR) ? IT0 : 1][(PRE == PRE_MUL_CONJ && !RR) ? R0 : 1];
    // columns pass: where the thread's inputs sit in a hop's intermediate (bytes) does not depend on the hop
    [[maybe_unused]] unsigned foff[PRE == PRE_NONE ? IT0 : 1][PRE == PRE_NONE ? R0 : 1];
    if constexpr (PRE == PRE_NONE) {
      const bool blocked = a.in_blocked != 0;
      // blocked: this tile's L x C values vector-fastest (e * C + c: consecutive threads, consecutive addresses)
      first_each_ct<NT, R0, L, C, true>(tid, [&](unsigned it, int q, unsigned c, unsigned e) {
        foff[it][q] = (blocked ? tile * NEL + e * C + c : e * ESTR + (v0 + c) * VSTR) * 8u;
      });
    }
    auto fetch = [&](unsigned bq, int rep) {
      const long long tb = tb_of(bq, rep);
      if constexpr (PRE == PRE_MUL_CONJ) {
        [[maybe_unused]] long long shsrc = 0;
        [[maybe_unused]] unsigned sh1 = 0, sh2 = 0;
        if constexpr (SHIFT) {
          const unsigned tbu = (unsigned)tb, sbin = fdiv(tbu, a.fd_nhops);
          const unsigned sbq = fdiv(sbin, a.fd_sden);
          unsigned sft = a.shift_q > 0 ? sbq * (unsigned)a.shift_q + (unsigned)a.shift0 : fmodu(tbu, a.fd_shift_bins);
          sft -= sft >= N ? N : 0u;
          shsrc = a.shift_q > 0 ? (long long)(sbin - sbq * sden) * a.nhops + (tbu - sbin * (unsigned)a.nhops) : (long long)fdiv(tbu, a.fd_shift_bins);
          sh2 = sft / OTHER;
          sh1 = sft - sh2 * OTHER;
        }
        const float2* __restrict__ src = a.in + (SHIFT ? shsrc : tb) * a.in_batch_stride;
        first_each_ct<NT, R0, L, C, false>(tid, [&](unsigned it, int q, unsigned c, unsigned e) {
          if constexpr (RR) {
            pre[it][q] = ld_off(src, fr_src[it][q]);
          } else {
            const unsigned pos = (v0 + c) * L + e;
            poth[it][q] = other[pos];
            if constexpr (SHIFT) {
              int k1 = (int)(v0 + c) - (int)sh1;
              const int bor = k1 < 0;
              k1 += bor ? OTHER : 0;
              int e2 = (int)e - (int)sh2 - bor;
              e2 += e2 < 0 ? L : 0;
              pre[it][q] = src[k1 * L + e2];
            } else {
              pre[it][q] = src[pos];
            }
          }
        });
      } else {
        const float2* __restrict__ src = a.in + tb * a.in_batch_stride;
        first_each_ct<NT, R0, L, C, true>(tid, [&](unsigned it, int q, unsigned, unsigned) { pre[it][q] = ld_off(src, foff[it][q]); });
      }
    };
    fetch(batch, 0);
    for (unsigned qi = 0; qi < nq; ++qi) {  // (one batch, but for the fused columns pass of a search without hops: PassArgs::bins_per_wg)
    const unsigned batch_q = batch + qi;
    if constexpr (POST == POST_ABS_ACC) {
#pragma unroll
      for (unsigned i = 0; i < ITL; ++i)
#pragma unroll
        for (int q = 0; q < RL; ++q) acc2[i][q] = 0.f;
    }
#ifdef GC_ACQ_STAGE_CLOCKS
    unsigned long long clk_acc[6] = {0, 0, 0, 0, 0, 0}, clk_last = __builtin_readcyclecounter();
#endif
    for (int rep = 0; rep < reps; ++rep) {
      const long long tb = tb_of(batch_q, rep);
      GC_CLK(5);
      // ---- first stage, inputs from registers -------------------------------------------------------------------------
      if constexpr (PRE == PRE_MUL_CONJ) {
        stage_first_ct<NT, R0, L, LP1, C, INV, false, PAD1>(
            [&](unsigned it, int q, unsigned, unsigned) -> float2 {
              const float2 val = pre[it][q];
              float2 o;
              if constexpr (RR) o = fr_oth[it][q];
              else o = poth[it][q];
              return make_float2(val.x * o.x + val.y * o.y, val.y * o.x - val.x * o.y);
            },
            buf1, tid);
      } else {
        stage_first_ct<NT, R0, L, LP1, C, INV, true, false>([&](unsigned it, int q, unsigned, unsigned) -> float2 { return pre[it][q]; }, buf1, tid);
      }
      GC_CLK(0);
      if (rep + 1 < reps) fetch(batch_q, rep + 1);
      else if (qi + 1 < nq) fetch(batch_q + 1, 0);
      __syncthreads();
      GC_CLK(1);
      // ---- middle stages: buf1 -> buf0 (-> buf1) ------------------------------------------------------------------------
      if constexpr (NST >= 3) {
        stage_ct<NT, R1, L, C, NS1, INV, LP, SP1, LP1>(buf1, buf0, twl, tid);
        GC_CLK(2);
        __syncthreads();
        GC_CLK(3);
      }
      if constexpr (NST >= 4) {
        stage_ct<NT, R2, L, C, NS2, INV, LP>(buf0, buf1, twl + T1, tid);
        __syncthreads();
      }
      const float2* lsrc = (NST == 3) ? buf0 : buf1;
      constexpr int SPL_ = NST == 2 ? SP1 : 0;            // two stages: the last one reads the first one's padded rows
      constexpr unsigned LPL = NST == 2 ? LP1 : LP;
      // ---- last stage, outputs to the epilogue in registers ---------------------------------------------------------------
      if constexpr (POST == POST_TWIDDLE) {
        float2* __restrict__ dstp = a.out + out_tb(tb) * a.out_batch_stride;
        const unsigned obl = (unsigned)a.out_blocked;
        stage_last_ct<NT, RL, L, LPL, C, INV, SPL_>(lsrc, twl_last, tid, [&](unsigned, int, unsigned c, unsigned e, float2 val) {
          val = cmul(val, cmul(tw2[c * TW2 + (e >> 4)], tw2[c * TW2 + EH + (e & 15)]));
          unsigned pos = (v0 + c) * L + e;
          if (obl) {
            const unsigned sh = obl - 1u, eb = e >> sh;
            pos = eb * (OTHER << sh) + ((v0 + c) << sh) + (e - (eb << sh));
          }
          dstp[pos] = val;
        });
      } else {
        const bool weighted = a.arm_hops > 0;
        const float wrep = weighted ? a.arm_w[min((int)fdiv((unsigned)rep, a.fd_arm_hops), 3)] : 1.0f;
        stage_last_ct<NT, RL, L, LPL, C, INV, SPL_>(lsrc, twl_last, tid, [&](unsigned it, int q, unsigned, unsigned, float2 val) {
          const float m = cabs_f(val.x, val.y);
          acc2[it][q] = weighted ? fmaf(wrep, m, acc2[it][q]) : acc2[it][q] + m;
        });
      }
      GC_CLK(4);
      // two stages: the last one read buf1, which the next hop's first stage writes
      if constexpr (NST == 2 || NST == 4) __syncthreads();
    }
#ifdef GC_ACQ_STAGE_CLOCKS
    if ((tid & 63u) == 0u) {
      unsigned long long* g = g_stage_clk + (POST == POST_ABS_ACC ? 0 : 64) + (tid >> 6) * 8;
      for (int i = 0; i < 6; ++i) atomicAdd(&g[i], clk_acc[i]);
      atomicAdd(&g[7], (unsigned long long)reps);
    }
#endif
    if constexpr (POST == POST_ABS_ACC) {
      // the sums, held per (iteration, output) of the last stage, through LDS into the order of the tile in memory (once per launch)
      float* fbuf = reinterpret_cast<float*>(buf1);
      __syncthreads();
      {
        constexpr unsigned LRL = L / RL;
#pragma unroll
        for (unsigned it = 0; it < ITL; ++it) {
          const unsigned b = tid + it * NT;
          if ((it + 1) * NT > NBL && b >= NBL) break;
          const unsigned c = b / LRL, j = b - c * LRL;
#pragma unroll
          for (int q = 0; q < RL; ++q) fbuf[c * L + j + q * LRL] = acc2[it][q];
        }
      }
      __syncthreads();
#pragma unroll
      for (unsigned slot = 0; slot < SLOTS; ++slot) {
        const unsigned idx = tid + slot * NT;
        if ((slot + 1) * NT > NEL && idx >= NEL) break;
        const unsigned e = idx / C, c = idx - e * C;
        accv[slot] = fbuf[c * L + e];
      }
      finish(batch_q, (bb * BQ + qi) * TILES + tile);  // (= bid when BQ == 1)
      if (qi + 1 < nq) __syncthreads();                 // the next batch's first stage writes buf1, which held the sums
    }
    }  // batches of the workgroup
  } else {

  // RR: everything of the load that does not depend on the hop - where in the source spectrum each of the thread's values comes
  // from (the rotation by the bin's shift) and the code-spectrum value it is multiplied with - is worked out once
  [[maybe_unused]] unsigned rr_src[RR ? SLOTS : 1];
  [[maybe_unused]] float2 rr_oth[RR ? SLOTS : 1];
  if constexpr (RR) {
    const long long tb0 = (long long)batch * reps;
    unsigned sft = a.shift_q > 0 ? fdiv(fdiv((unsigned)tb0, a.fd_nhops), a.fd_sden) * (unsigned)a.shift_q + (unsigned)a.shift0 : fmodu((unsigned)tb0, a.fd_shift_bins);
    sft -= sft >= N ? N : 0u;
    const unsigned s2 = sft / OTHER, s1 = sft - s2 * OTHER;
#pragma unroll
    for (unsigned slot = 0; slot < SLOTS; ++slot) {
      const unsigned idx = tid + slot * NT;
      if ((slot + 1) * NT > NEL && idx >= NEL) break;
      const unsigned c = idx / L, e = idx - c * L;
      int k1 = (int)(v0 + c) - (int)s1;
      const int bor = k1 < 0;
      k1 += bor ? OTHER : 0;
      int e2 = (int)e - (int)s2 - bor;
      e2 += e2 < 0 ? L : 0;
      rr_src[RR ? slot : 0] = (unsigned)(k1 * L + e2);
      rr_oth[RR ? slot : 0] = other[v0 * L + idx];
    }
  }

  for (int rep = 0; rep < reps; ++rep) {
    const long long tb = POST == POST_ABS_ACC ? (long long)batch * a.nhops + (long long)hg * reps + rep
                         : RR                 ? (long long)batch * reps + rep
                                              : (long long)batch;
    // ---- load --------------------------------------------------------------------------------------------
    [[maybe_unused]] int cb = 0, ch = 0;
    [[maybe_unused]] double fcyc = 0.0;
    [[maybe_unused]] long long shsrc = 0;
    [[maybe_unused]] unsigned sh1 = 0, sh2 = 0;
    if constexpr (SHIFT) {
      const unsigned tbu = (unsigned)tb, sbin = fdiv(tbu, a.fd_nhops);
      const unsigned sbq = fdiv(sbin, a.fd_sden);
          unsigned sft = a.shift_q > 0 ? sbq * (unsigned)a.shift_q + (unsigned)a.shift0 : fmodu(tbu, a.fd_shift_bins);
      sft -= sft >= N ? N : 0u;
      shsrc = a.shift_q > 0 ? (long long)(sbin - sbq * sden) * a.nhops + (tbu - sbin * (unsigned)a.nhops) : (long long)fdiv(tbu, a.fd_shift_bins);
      sh2 = sft / OTHER;
      sh1 = sft - sh2 * OTHER;
    }
    if constexpr (PRE == PRE_IF_CARRIER) {
      cb = (int)fdiv((unsigned)tb, a.fd_nhops);
      ch = (int)((unsigned)tb - (unsigned)cb * (unsigned)a.nhops);
      fcyc = (a.f0 - a.fstep * cb) / a.fs;  // cycles per sample of bin cb (acquisition.m:169-181)
    }
#pragma unroll
    for (unsigned slot = 0; slot < SLOTS; ++slot) {
      const unsigned idx = tid + slot * NT;
      if ((slot + 1) * NT > NEL && idx >= NEL) break;
      unsigned pos, li;
      if constexpr (CONTIG) {
        pos = v0 * L + idx;  // the tile is C whole vectors: one contiguous run of the transform
        li = idx;
      } else {
        const unsigned e = idx / C, c = idx - e * C;
        pos = e * ESTR + (v0 + c) * VSTR;
        if constexpr (PRE == PRE_NONE) {
          if (a.in_blocked) pos = tile * NEL + idx;  // this tile's L x C values, in the order the threads take them
        }
        li = c * L + e;
      }
      float2 val;
      if constexpr (PRE == PRE_IF_CARRIER) {
        const long long s = a.first_sample + (long long)ch * a.spc + (long long)pos;
        float xi, xq;
        if (a.if_f32) {
          const float2 z = a.if_f32[s];
          xi = z.x;
          xq = z.y;
        } else {
          const char2 x = *reinterpret_cast<const char2*>(a.if_base + 2 * s);
          xi = (float)x.x;
          xq = (float)x.y;
        }
        const double ph = fcyc * (double)pos;
        float sn, cs;
        sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
        val = make_float2(xi * cs + xq * sn, xq * cs - xi * sn);
      } else if constexpr (PRE == PRE_CODE) {
        val = (int)pos < a.spc ? make_float2((float)a.codes[tb * a.spc + pos], 0.f) : make_float2(0.f, 0.f);
      } else {
        if constexpr (SHIFT) {
          // Y[k] = X[(k - s) mod N], k = k1 + N1*k2 stored at k1*N2 + k2 (N1 = OTHER rows of N2 = L): row v of Y is row
          // (v - s1) mod N1 of X rotated by s2 (+1 when the row index wrapped), s = s1 + N1*s2
          if constexpr (RR) {
            val = a.in[shsrc * a.in_batch_stride + rr_src[slot]];
          } else {
            const unsigned c = idx / L, e = idx - c * L;
            int k1 = (int)(v0 + c) - (int)sh1;
            const int bor = k1 < 0;
            k1 += bor ? OTHER : 0;
            int e2 = (int)e - (int)sh2 - bor;
            e2 += e2 < 0 ? L : 0;
            val = a.in[shsrc * a.in_batch_stride + k1 * L + e2];
          }
        } else {
          val = a.in[tb * a.in_batch_stride + pos];
        }
        if constexpr (PRE == PRE_MUL_CONJ) {
          float2 o;
          if constexpr (RR) o = rr_oth[slot];
          else o = other[pos];
          val = make_float2(val.x * o.x + val.y * o.y, val.y * o.x - val.x * o.y);
        }
      }
      buf0[li] = val;
    }
    __syncthreads();

    // ---- stages: buf0 -> buf1 -> buf0 -> ... -------------------------------------------------------------
    stage_ct<NT, R0, L, C, 1, INV>(buf0, buf1, twl, tid);
    __syncthreads();
    if constexpr (R1 > 1) {
      stage_ct<NT, R1, L, C, NS1, INV>(buf1, buf0, twl, tid);
      __syncthreads();
    }
    if constexpr (R2 > 1) {
      stage_ct<NT, R2, L, C, NS2, INV>(buf0, buf1, twl + T1, tid);
      __syncthreads();
    }
    if constexpr (R3 > 1) {
      stage_ct<NT, R3, L, C, NS3, INV>(buf1, buf0, twl + T1 + T2, tid);
      __syncthreads();
    }
    const float2* res = (NST & 1) ? buf1 : buf0;

    // ---- store -------------------------------------------------------------------------------------------
#pragma unroll
    for (unsigned slot = 0; slot < SLOTS; ++slot) {
      const unsigned idx = tid + slot * NT;
      if ((slot + 1) * NT > NEL && idx >= NEL) break;
      unsigned pos, li, e, c;
      if constexpr (CONTIG) {
        c = idx / L;
        e = idx - c * L;
        pos = v0 * L + idx;
        if constexpr (POST == POST_TWIDDLE || POST == POST_STORE) {
          if (a.out_blocked) {
            const unsigned sh = (unsigned)a.out_blocked - 1u, eb = e >> sh;
            pos = eb * (OTHER << sh) + ((v0 + c) << sh) + (e - (eb << sh));
          }
        }
        li = idx;
      } else {
        e = idx / C;
        c = idx - e * C;
        pos = e * ESTR + (v0 + c) * VSTR;
        li = c * L + e;
      }
      float2 val = res[li];
      if constexpr (POST == POST_TWIDDLE) val = cmul(val, cmul(tw2[c * TW2 + (e >> 4)], tw2[c * TW2 + EH + (e & 15)]));
      if constexpr (POST == POST_ABS_ACC) {
        accv[slot] += cabs_f(val.x, val.y);
      } else {
        a.out[out_tb(tb) * a.out_batch_stride + pos] = val;
      }
    }
    // the next hop's load overwrites buf0: safe without a barrier when the result sits in buf1 (the barrier after the
    // load orders this hop's reads of buf1 before the next first stage writes it)
    if constexpr ((POST == POST_ABS_ACC || RR) && !(NST & 1)) __syncthreads();
  }
    if constexpr (POST == POST_ABS_ACC) finish(batch, bid);
  }  // !FUSE
}

// ---- the whole inverse transform of a (PRN, bin) in workgroups that never touch memory in between ------------------------
// acquisition.m:183-191 per (PRN, bin): for every hop ifft(fft(sigCarr .* x) .* conj(fft(code))), |.|, summed over the hops.
// The two-pass inverse transform above writes N complex values per (bin, hop) and reads them back: 334 MB per PRN at the
// default search, four orders of magnitude above the search's input.  Here the N-point inverse transform is cut by ONE
// decimation-in-frequency step of radix 4 into four independent transforms of M = N / 4 points,
//     y[4m + r] = IDFT_M( (sum_q P[k' + M q] * i^(q r)) * exp(+2 pi i k' r / N) )[m],     P[k] = X[(k - s) mod N] * conj(C[k]),
// and one workgroup of 1024 threads owns (PRN, bin, r): M = 9 000 points are 72 KB, two such buffers (Stockham ping-pong) fit the
// 160 KB of LDS, so the product, the radix-4 step, the M-point transform, |.| and the sum over the hops (nine float registers
// per thread) never leave the CU; after the last hop the workgroup picks its own peak (the same two 64-bit atomic maxima per
// PRN).  Nothing is written but those keys: no intermediate, no results array, no combine kernel, ONE launch for all PRNs.
// The price is reading the spectra four times (each of the four workgroups of a (PRN, bin) forms all N products): 5.8 MB of
// hop spectra and 288 KB of code spectrum per PRN that live in L2.  The spectra are stored [k1][k2] (k = k1 + N1 k2, rows of N2
// contiguous values): k' + M q is the same row, N2 / 4 columns further, so the reads are runs of N2 / 4 contiguous values; the
// transform wants k' natural, i.e. [k2'][k1] - the first buffer's rows are padded by one element so that those transposed
// stores do not pile onto a few LDS banks.

constexpr int kFusedThreads = 1024;

template <int R, int L, int NS, int SRC_ROW, int MODE, bool INV>  // SRC_ROW > 0: the source buffer's rows of SRC_ROW values are padded by one
__device__ __forceinline__ void stage_fused(const float2* __restrict__ src, float2* __restrict__ dst, const float2* __restrict__ twl,
                                            const float2* __restrict__ ta, const float2* __restrict__ tb, unsigned tid) {
  // MODE 0: no twiddles (NS == 1); 1: table twl[k * (R - 1) + q - 1]; 2: two-level, W^(k q) = ta[(k q) / 100] * tb[(k q) % 100]
  constexpr unsigned LR = L / R, ITERS = (LR + kFusedThreads - 1) / kFusedThreads;
  constexpr float sign = INV ? -1.0f : 1.0f;
#pragma unroll
  for (unsigned it = 0; it < ITERS; ++it) {
    const unsigned j = tid + it * kFusedThreads;
    if ((it + 1) * kFusedThreads > LR && j >= LR) break;
    const unsigned k = NS == 1 ? 0u : j % (unsigned)NS;
    float2 vq[R], oq[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      unsigned pos = j + q * LR;
      if constexpr (SRC_ROW > 0) pos += pos / (unsigned)SRC_ROW;
      float2 x = src[pos];
      if constexpr (MODE == 1) {
        if (q > 0) x = cmul(x, twl[k * (R - 1) + q - 1]);  // row k = 0 holds ones
      } else if constexpr (MODE == 2) {
        if (q > 0) {
          const unsigned m = k * (unsigned)q, hi = m / 100u, lo = m - hi * 100u;
          x = cmul(x, cmul(ta[hi], tb[lo]));
        }
      }
      vq[q] = x;
    }
    butterfly<R>(vq, sign, oq);
    float2* d = dst + (j - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) d[q * NS] = oq[q];
  }
}

template <int N1, int N2, int R0, int R1, int R2, int R3>
__global__ __launch_bounds__(kFusedThreads) void acq_fused_kernel(const FusedArgs a) {
  constexpr int N = N1 * N2, C2 = N2 / 4, M = N1 * C2, PADR = N1 + 1;
  static_assert(N2 % 4 == 0 && R0 * R1 * R2 * R3 == M && M % 100 == 0, "one radix-4 DIF step, then four Stockham stages");
  constexpr int SLOTS = (M + kFusedThreads - 1) / kFusedThreads;
  constexpr int NS1 = R0, NS2 = R0 * R1, NS3 = R0 * R1 * R2, LR3 = M / R3;
  static_assert(NS3 == LR3 && LR3 <= kFusedThreads && R3 <= SLOTS + 1, "the last stage: one butterfly per thread, its outputs the thread's own columns");
  constexpr unsigned T1 = NS1 * (R1 - 1), T2 = NS2 * (R2 - 1), TA = M / 100;
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  float2* const buf0 = reinterpret_cast<float2*>(fsm);   // C2 rows of N1 (+1) values: k' = k1 + N1 k2' at k2' * PADR + k1
  float2* const buf1 = buf0 + C2 * PADR;                 // M
  float2* const tw1 = buf1 + M;                          // stage 2: W_{NS1 R1}^(k q)
  float2* const tw2 = tw1 + T1;                          // stage 3
  float2* const ta = tw2 + T2;                           // stage 4, two-level: W_M^(100 a)
  float2* const tb = ta + TA;                            //                     W_M^b, b < 100
  const unsigned tid = threadIdx.x;
  const unsigned r = blockIdx.x & 3u;
  const unsigned pb = blockIdx.x >> 2;
  const unsigned bin = pb % (unsigned)a.nbins, prn = pb / (unsigned)a.nbins;

  // stage twiddles (inverse transform: conjugates of the table), once per workgroup
  for (unsigned i = tid; i < T1; i += kFusedThreads) {
    const unsigned k = i / (R1 - 1), q = i % (R1 - 1) + 1;
    const float2 w = a.tw[k * q * (N / (NS1 * R1))];
    tw1[i] = make_float2(w.x, -w.y);
  }
  for (unsigned i = tid; i < T2; i += kFusedThreads) {
    const unsigned k = i / (R2 - 1), q = i % (R2 - 1) + 1;
    const float2 w = a.tw[k * q * (N / (NS2 * R2))];
    tw2[i] = make_float2(w.x, -w.y);
  }
  for (unsigned i = tid; i < TA + 100u; i += kFusedThreads) {
    const float2 w = a.tw[(i < TA ? i * 100u : i - TA) * (unsigned)(N / M)];
    ta[i] = make_float2(w.x, -w.y);  // (tb follows ta)
  }
  const unsigned sft = a.shift_q > 0 ? bin * (unsigned)a.shift_q : 0u;
  const unsigned sh2 = sft / (unsigned)N1, sh1 = sft - sh2 * (unsigned)N1;  // s = s1 + N1 s2
  float acc[R3];
#pragma unroll
  for (int q = 0; q < R3; ++q) acc[q] = 0.0f;
  __syncthreads();

  for (int hop = 0; hop < a.nhops; ++hop) {
    const float2* __restrict__ X = a.sig + (size_t)(a.shift_q > 0 ? (unsigned)hop : bin * (unsigned)a.nhops + (unsigned)hop) * N;
    for (int arm = 0; arm < a.narms; ++arm) {
      const float2* __restrict__ C = a.codespec + ((size_t)prn * a.narms + arm) * N;
      // ---- product, radix-4 decimation-in-frequency step, twiddle: buf0[k'] -----------------------------------------
#pragma unroll 3
      for (int sl = 0; sl < SLOTS; ++sl) {
        const unsigned idx = tid + sl * kFusedThreads;
        if (idx >= (unsigned)M) break;
        const unsigned k1 = idx / (unsigned)C2, k2p = idx - k1 * (unsigned)C2;
        // Y[k] = X[(k - s) mod N], k = k1 + N1 k2 stored at k1 N2 + k2: row k1 of Y is row (k1 - s1) mod N1 of X rotated by s2
        // (+ 1 when the row index wrapped)
        int k1s = (int)k1 - (int)sh1;
        const int bor = k1s < 0;
        k1s += bor ? N1 : 0;
        const float2* __restrict__ xrow = X + k1s * N2;
        const float2* __restrict__ crow = C + k1 * N2;
        float2 pq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k2 = (int)k2p + C2 * q;
          int e2 = k2 - (int)sh2 - bor;
          e2 += e2 < 0 ? N2 : 0;
          const float2 x = xrow[e2], o = crow[k2];
          pq[q] = make_float2(x.x * o.x + x.y * o.y, x.y * o.x - x.x * o.y);  // X * conj(C)
        }
        const float2 s02 = make_float2(pq[0].x + pq[2].x, pq[0].y + pq[2].y), d02 = make_float2(pq[0].x - pq[2].x, pq[0].y - pq[2].y);
        const float2 s13 = make_float2(pq[1].x + pq[3].x, pq[1].y + pq[3].y), d13 = make_float2(pq[1].x - pq[3].x, pq[1].y - pq[3].y);
        float2 z;
        if (r == 0u) z = make_float2(s02.x + s13.x, s02.y + s13.y);
        else if (r == 2u) z = make_float2(s02.x - s13.x, s02.y - s13.y);
        else if (r == 1u) z = make_float2(d02.x - d13.y, d02.y + d13.x);   // + i * d13
        else z = make_float2(d02.x + d13.y, d02.y - d13.x);                // - i * d13
        if (r != 0u) {  // the radix-4 step's twiddle exp(+2 pi i k' r / N): k' r < 3 M < N
          const float2 w = a.tw[(k1 + (unsigned)N1 * k2p) * r];
          z = make_float2(z.x * w.x + z.y * w.y, z.y * w.x - z.x * w.y);  // z * conj(w)
        }
        buf0[k2p * PADR + k1] = z;
      }
      __syncthreads();
      // ---- M-point inverse transform: buf0 -> buf1 -> buf0 -> buf1 -> buf0 ---------------------------------------------
      stage_fused<R0, M, 1, N1, 0, true>(buf0, buf1, nullptr, nullptr, nullptr, tid);
      __syncthreads();
      stage_fused<R1, M, NS1, 0, 1, true>(buf1, buf0, tw1, nullptr, nullptr, tid);
      __syncthreads();
      stage_fused<R2, M, NS2, 0, 1, true>(buf0, buf1, tw2, nullptr, nullptr, tid);
      __syncthreads();
      // last stage: thread j < M / R3 turns out y[4m + r] for m = j + (M / R3) q, q < R3 - its own columns in every hop: |.| goes
      // straight from the butterfly's registers into the thread's sums (no store, no barrier: the next hop's products go to
      // buf0, which nobody reads any more, and its first stage writes buf1 only behind the barrier that follows them)
      const float wgt = a.weight[arm] != 0.0f ? a.weight[arm] : 1.0f;
      if (tid < (unsigned)LR3) {
        const unsigned k = tid;  // NS3 = M / R3: k = j
        float2 vq[R3], oq[R3];
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          float2 x = buf1[tid + q * LR3];
          if (q > 0) {
            const unsigned m = k * (unsigned)q, hi = m / 100u, lo = m - hi * 100u;
            x = cmul(x, cmul(ta[hi], tb[lo]));
          }
          vq[q] = x;
        }
        butterfly<R3>(vq, -1.0f, oq);
#pragma unroll
        for (int q = 0; q < R3; ++q) acc[q] = fmaf(wgt, cabs_f(oq[q].x, oq[q].y), acc[q]);
      }
    }
  }
  // ---- this workgroup's peak: largest value, smallest bin, smallest column (acquisition.m:196-198) ------------------------
  unsigned int pm = 0, pbin = 0xffffffffu, pcol = 0xffffffffu;
#pragma unroll
  for (int q = 0; q < R3; ++q) {
    const unsigned m = tid + (unsigned)q * (unsigned)LR3;
    const unsigned c = 4u * m + r;
    if (tid < (unsigned)LR3 && c < (unsigned)a.valid) {
      const unsigned int u = __float_as_uint(acc[q] * a.inv_n);
      if (u > pm) {
        pm = u;
        pcol = c;
      } else if (u == pm) {
        pcol = min(pcol, c);
      }
      pbin = bin;
    }
  }
  __shared__ unsigned int sm[16], sc[16];
  unsigned int wm = pm;
  for (int off = 32; off > 0; off >>= 1) wm = max(wm, (unsigned int)__shfl_xor((int)wm, off, 64));
  unsigned int c = (pm == wm && pbin != 0xffffffffu) ? pcol : 0xffffffffu;
  for (int off = 32; off > 0; off >>= 1) c = min(c, (unsigned int)__shfl_xor((int)c, off, 64));
  const int wave = tid >> 6;
  if ((tid & 63u) == 0u) {
    sm[wave] = wm;
    sc[wave] = c;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kFusedThreads / 64; ++w) {
      if (sm[w] > wm) {
        wm = sm[w];
        c = sc[w];
      } else if (sm[w] == wm) {
        c = min(c, sc[w]);
      }
    }
    if (c != 0xffffffffu) {
      unsigned long long* keys = a.keys + 2 * (size_t)prn;
      const unsigned long long ka = ((unsigned long long)wm << 32) | (unsigned long long)(0xffffffffu - bin);
      const unsigned long long kb = ((unsigned long long)wm << 32) | (unsigned long long)(0xffffffffu - c);
      if (ka > __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&keys[0], ka);
      if (kb > __hip_atomic_load(&keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&keys[1], kb);
    }
  }
}

template <int N1, int N2, int R0, int R1, int R2, int R3>
bool try_fused(gc_context* ctx, const Plan& pl, const FusedArgs& a, int nprn) {
  if (pl.n1 != N1 || pl.n2 != N2) return false;
  constexpr int C2 = N2 / 4, M = N1 * C2;
  constexpr size_t smem = ((size_t)C2 * (N1 + 1) + M + R0 * (R1 - 1) + R0 * R1 * (R2 - 1) + M / 100 + 100) * sizeof(float2);
  static_assert(smem <= 160 * 1024, "two buffers of N / 4 points and the stage tables in 160 KB of LDS");
  auto fn = acq_fused_kernel<N1, N2, R0, R1, R2, R3>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(fn, dim3((unsigned int)(nprn * a.nbins * 4)), dim3(kFusedThreads), smem, ctx->stream, a);
  return true;
}

// Launches fft_pass_ct<...> when `a` describes exactly that instantiation (its tile width C replaces a.cols).
template <int L, int OTHER, bool CONTIG, int C, int PRE, int POST, bool INV, bool SHIFT, int R0, int R1, int R2, int R3>
bool try_ct(gc_context* ctx, const PassArgs& a, long long nbatch_groups) {
  constexpr int rad[4] = {R0, R1, R2, R3};
  constexpr int nst = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
  if (a.len != L || a.nvec != OTHER || a.n != L * OTHER || a.pre != PRE || a.post != POST || (a.inverse != 0) != INV ||
      (a.pre == PRE_MUL_CONJ && (a.shift_bins > 0 || a.shift_q > 0)) != SHIFT || a.nrad != nst || a.wrap_len > 0)
    return false;
  if (SHIFT && (a.n1 != OTHER || a.n2 != L)) return false;
  if (CONTIG ? (a.estride != 1 || a.vstride != L) : (a.estride != OTHER || a.vstride != 1)) return false;
  for (int i = 0; i < nst; ++i)
    if (a.rad[i] != rad[i]) return false;
  PassArgs b = a;
  b.bins_per_wg = 1;
  b.nbatch_total = (int)nbatch_groups;
  constexpr bool fused = GC_ACQ_FUSE_IO != 0 && nst >= 2 && PRE == PRE_NONE && POST == POST_ABS_ACC;
  if (fused && a.hop_groups <= 1 && !GC_TUNE_ENV("GC_ACQ_ONE_BIN")) {
    // several consecutive batches per workgroup while the launch keeps a dozen workgroups per CU (BDS B1C: 200 tiles x 201 bins)
    for (int cand : {4, 2})
      if ((long long)(OTHER / C) * nbatch_groups / cand >= 12LL * ctx->compute_units) {
        b.bins_per_wg = cand;
        break;
      }
    if (const char* e = GC_TUNE_ENV("GC_ACQ_BINS_PER_WG")) b.bins_per_wg = std::max(1, std::atoi(e));
  }
  const long long groups = (nbatch_groups + b.bins_per_wg - 1) / b.bins_per_wg;
  hipLaunchKernelGGL((fft_pass_ct<L, OTHER, CONTIG, C, PRE, POST, INV, SHIFT, R0, R1, R2, R3>),
                     dim3((unsigned int)((OTHER / C) * groups)), dim3(ct_threads<L, C, R0, R1, R2, R3>()), 0, ctx->stream, b);
  return true;
}

// the passes of a search over N = N1 x N2 (columns: length N1, C1 per tile, radices A..; rows: length N2, C2, B..)
#define GC_CT_SHAPE(N1, N2, C1, A0, A1, A2, A3, C2, B0, B1, B2, B3)                                                     \
  (try_ct<N1, N2, false, C1, PRE_IF_CARRIER, POST_TWIDDLE, false, false, A0, A1, A2, A3>(ctx, a, nbatch_groups) ||      \
   try_ct<N1, N2, false, C1, PRE_CODE, POST_TWIDDLE, false, false, A0, A1, A2, A3>(ctx, a, nbatch_groups) ||            \
   try_ct<N2, N1, true, C2, PRE_NONE, POST_STORE, false, false, B0, B1, B2, B3>(ctx, a, nbatch_groups) ||               \
   try_ct<N2, N1, true, C2, PRE_MUL_CONJ, POST_TWIDDLE, true, false, B0, B1, B2, B3>(ctx, a, nbatch_groups) ||          \
   try_ct<N2, N1, true, C2, PRE_MUL_CONJ, POST_TWIDDLE, true, true, B0, B1, B2, B3>(ctx, a, nbatch_groups) ||           \
   try_ct<N1, N2, false, C1, PRE_NONE, POST_ABS_ACC, true, false, A0, A1, A2, A3>(ctx, a, nbatch_groups))
}  // namespace
namespace gcacq {

bool launch_fused(gc_context* ctx, const Plan& pl, const FusedArgs& a, int nprn) {
  return try_fused<180, 200, 10, 10, 10, 9>(ctx, pl, a, nprn) || try_fused<150, 160, 10, 10, 10, 6>(ctx, pl, a, nprn);
}

// tile width C1 of the specialised columns pass for vectors of `len`, `nvec` of them per transform (the shapes of GC_CT_SHAPE below:
// its launch has nvec / C1 workgroups per batch, whatever PassArgs::cols says); 0: no specialised pass
int ct_columns_tile(int len, int nvec) {
  static const int shapes[][3] = {{180, 200, 8}, {150, 160, 8}, {375, 384, 4}, {250, 288, 8}, {600, 600, 5}, {320, 1000, 8}};
  for (const auto& k : shapes)
    if (len == k[0] && nvec == k[1]) return k[2];
  return 0;
}

int launch_pass(gc_context* ctx, PassArgs& a, long long nbatch_groups, bool* used_ct) {
  if (used_ct) *used_ct = false;
  {
    const long long sden = a.shift_den > 1 ? a.shift_den : 1, hgr = a.hop_groups > 1 ? a.hop_groups : 1;
    a.fd_nhops = make_fdiv(a.nhops);
    a.fd_shift_bins = make_fdiv(a.shift_bins);
    a.fd_sden = make_fdiv(sden);
    a.fd_hg = make_fdiv(hgr);
    a.fd_arm_batches = make_fdiv(a.arm_batches);
    a.fd_arm_hops = make_fdiv(a.arm_hops);
    // the largest number any of them divides: a transform index of the launch (batches x hops, plus the first batch's number)
    const long long xmax = (nbatch_groups + a.batch0 + 1) * std::max(1, a.nhops) * std::max<long long>(1, a.row_reps);
    const long long dmax = std::max({(long long)a.nhops, (long long)a.shift_bins, sden, hgr, (long long)a.arm_batches, (long long)a.arm_hops, 1LL});
    if (xmax * dmax >= (1LL << 32)) {
      gc_set_error("acquisition: %lld transforms per launch (divisor %lld) are more than the pass kernels' index arithmetic takes", xmax, dmax);
      return GC_E_UNSUPPORTED;
    }
  }
  const bool generic = GC_TUNE_ENV("GC_ACQ_GENERIC") != nullptr;  // (read per call: the tests switch it)
  const bool no_pairs = GC_TUNE_ENV("GC_ACQ_NO_XCD_PAIRS") != nullptr;
  const char* xmap = GC_TUNE_ENV("GC_ACQ_XCD_MAP");
  a.no_xcd_pairs = no_pairs ? 1 : (xmap && std::strcmp(xmap, "pairs") == 0) ? 2 : 0;
  if (!generic) {
    // N = 36 000: 18 Msps, 1 ms codes (GPS L1 C/A, L5, Galileo E5a/E5b, BDS B2a/B3I: initSettings.m of each package);
    // N = 24 000: GLONASS L1/L2 at 12 Msps
    // N = 144 000: Galileo E1 (4-ms codes at 18 Msps); N = 72 000 / 360 000 / 320 000: the circular-shift searches of BDS B1I
    // (4-ms blocks), BDS B1C (20 ms) and GPS L2C (40 ms at 8 Msps)
    if (GC_CT_SHAPE(180, 200, 8, 6, 6, 5, 1, 6, 8, 5, 5, 1) || GC_CT_SHAPE(150, 160, 8, 6, 5, 5, 1, 6, 8, 5, 4, 1) ||
        GC_CT_SHAPE(375, 384, 4, 5, 5, 5, 3, 5, 8, 8, 6, 1) || GC_CT_SHAPE(250, 288, 8, 5, 5, 5, 2, 5, 8, 6, 6, 1) ||
        GC_CT_SHAPE(600, 600, 5, 6, 5, 5, 4, 3, 6, 5, 5, 4) || GC_CT_SHAPE(320, 1000, 8, 8, 8, 5, 1, 2, 8, 5, 5, 5)) {
      GC_HIP(hipGetLastError());
      if (used_ct) *used_ct = true;
      return GC_OK;
    }
  }
  if (a.batch0 != 0 || a.arm_batches > 0 || a.acc_row0 != 0) {  // fft_pass_kernel numbers its batches from 0 and knows no merged arms: it would transform other rows into other places
    gc_set_error("acquisition: rows / bins in chunks, merged arms and single-row transforms need the specialised pass kernels (length %d x %d)",
                 a.len, a.nvec);
    return GC_E_STATE;
  }
  if (a.in_blocked || a.out_blocked || a.row_reps > 1) {  // handover_block() promised a specialised pair of passes for this plan
    gc_set_error("acquisition: no specialised pass kernel for a blocked hand-over (length %d x %d)", a.len, a.nvec);
    return GC_E_STATE;
  }
  const int tiles = (a.nvec + a.cols - 1) / a.cols;
  const size_t smem = ((size_t)2 * a.len * a.cols + a.len + (size_t)a.cols * (((a.len - 1) >> 4) + 17)) * sizeof(float2);
  hipLaunchKernelGGL(fft_pass_kernel, dim3((unsigned int)(tiles * nbatch_groups)), dim3(kFftThreads), smem, ctx->stream, a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// log2(B) + 1 for the blocked hand-over between the inverse transform's passes (PassArgs::out_blocked), B = the tile width of
// the specialised columns pass of this plan (GC_CT_SHAPE above) where that is a power of two; 0: natural order (generic kernel,
// the 600 x 600 and 512 x 625 plans with tiles of 3 and 5 columns, GC_ACQ_NATURAL_ORDER=1 for A/B runs).
int handover_block(const Plan& pl) {
  const bool off = GC_TUNE_ENV("GC_ACQ_GENERIC") != nullptr || GC_TUNE_ENV("GC_ACQ_NATURAL_ORDER") != nullptr;
  if (off) return 0;
  static const struct { int n1, n2, log2b; } shapes[] = {{180, 200, 3}, {150, 160, 3}, {375, 384, 2}, {250, 288, 3}};
  for (const auto& k : shapes)
    if (pl.n1 == k.n1 && pl.n2 == k.n2) return k.log2b + 1;
  return 0;
}

void fill_sub(PassArgs& a, const SubPlan& sp) {
  a.len = sp.len;
  a.nrad = sp.nrad;
  for (int i = 0; i < sp.nrad; ++i) a.rad[i] = sp.rad[i];
}

// columns per tile: keep 2*L*C*8 bytes <= 64 KiB and L*C <= 8*256 (POST_ABS_ACC register slots)
int choose_cols(int L, int estride) {
  int budget = 2048;
  if (const char* e = GC_TUNE_ENV("GC_ACQ_TILE")) budget = std::max(256, std::atoi(e));  // tuning: elements per tile
  int c = std::max(1, std::min(16, budget / L));
  // strided vectors (the column passes): a tile row is c consecutive float2; whole 64-byte sectors when c is a multiple of 8
  static const int align = [] { const char* e = GC_TUNE_ENV("GC_ACQ_COLS_ALIGN"); return e ? std::atoi(e) : 8; }();
  if (estride != 1 && align > 1 && c >= align) c -= c % align;
  return c;
}

// Forward transform of `nbatch` sequences produced by `pre` into `dst` (layout [k1][k2]).
int forward(gc_context* ctx, AcqScratch* s, PassArgs base, int pre, long long nbatch, float2* dst) {
  const Plan& pl = s->plan;
  PassArgs a = base;
  a.n = pl.n;
  a.tw = s->tw;
  a.inverse = 0;
  // F1: columns (length n1, element stride n2), twiddle, store [k1][n2]
  fill_sub(a, pl.p1);
  a.nvec = pl.n2;
  a.estride = pl.n2;
  a.vstride = 1;
  a.cols = choose_cols(a.len, a.estride);
  a.pre = pre;
  a.post = POST_TWIDDLE;
  a.out = s->tmp;
  a.out_batch_stride = pl.n;
  int rc = launch_pass(ctx, a, nbatch);
  if (rc) return rc;
  // F2: rows (length n2, contiguous)
  fill_sub(a, pl.p2);
  a.nvec = pl.n1;
  a.estride = 1;
  a.vstride = pl.n2;
  a.cols = choose_cols(a.len, a.estride);
  a.pre = PRE_NONE;
  a.post = POST_STORE;
  a.in = s->tmp;
  a.in_batch_stride = pl.n;
  a.out = dst;
  a.out_batch_stride = pl.n;
  return launch_pass(ctx, a, nbatch);
}
}  // namespace gcacq

// Test hook: forward FFT of `nbatch` host sequences of length n (complex64) with the library's
// transform; output in natural frequency order.

extern "C" int gc_debug_fft(gc_context* ctx, int n, int nbatch, const float* in, float* out_natural, int inverse) {
  if (!ctx || n <= 1 || nbatch <= 0 || !in || !out_natural) return GC_E_INVALID;
  GC_HIP(hipSetDevice(ctx->device));
  AcqScratch* s = nullptr;
  int rc = ensure_scratch(ctx, n, nbatch, 1, 1, n / 2, &s);
  if (rc) return rc;
  const Plan& pl = s->plan;
  GC_HIP(hipMemcpy(s->sig, in, (size_t)nbatch * n * sizeof(float2), hipMemcpyHostToDevice));
  PassArgs a;
  std::memset(&a, 0, sizeof a);
  a.n = pl.n;
  a.tw = s->tw;
  a.inverse = inverse;
  fill_sub(a, pl.p1);
  a.nvec = pl.n2;
  a.estride = pl.n2;
  a.vstride = 1;
  a.cols = choose_cols(a.len, a.estride);
  a.pre = PRE_NONE;
  a.post = POST_TWIDDLE;
  a.in = s->sig;
  a.in_batch_stride = pl.n;
  a.out = s->tmp;
  a.out_batch_stride = pl.n;
  rc = launch_pass(ctx, a, nbatch);
  if (rc) return rc;
  fill_sub(a, pl.p2);
  a.nvec = pl.n1;
  a.estride = 1;
  a.vstride = pl.n2;
  a.cols = choose_cols(a.len, a.estride);
  a.post = POST_STORE;
  a.in = s->tmp;
  a.out = s->sig;
  rc = launch_pass(ctx, a, nbatch);
  if (rc) return rc;
  std::vector<float2> h((size_t)nbatch * n);
  GC_HIP(hipMemcpyAsync(h.data(), s->sig, h.size() * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  float2* o = (float2*)out_natural;
  for (int b = 0; b < nbatch; ++b)
    for (int k1 = 0; k1 < pl.n1; ++k1)
      for (int k2 = 0; k2 < pl.n2; ++k2) o[(size_t)b * n + k1 + (size_t)pl.n1 * k2] = h[(size_t)b * n + (size_t)k1 * pl.n2 + k2];
  return GC_OK;
}

#ifdef GC_ACQ_STAGE_CLOCKS
// tuning builds only (not declared in include/gnsscorr.h): the counters of GC_CLK, optionally cleared
extern "C" int gc_debug_acq_stage_clocks(unsigned long long* out128, int reset) {
  if (out128 && hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_stage_clk), sizeof(unsigned long long) * 128) != hipSuccess) return GC_E_HIP;
  if (reset) {
    static const unsigned long long zeros[128] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stage_clk), zeros, sizeof zeros) != hipSuccess) return GC_E_HIP;
  }
  return GC_OK;
}
#endif
