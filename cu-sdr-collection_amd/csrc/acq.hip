// acq.hip — FFT-based parallel code-phase search (acquisition.m:151-200) and the GPS L1 C/A
// fine-frequency stage (acquisition.m:213-254) on gfx950.
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
#include <algorithm>
#include <cmath>
#include <mutex>
#include <type_traits>

#include "acq_guard.h"
#include "gc_internal.h"

namespace {

constexpr int kGuardListCap = 4096;  // cells within gc_acq_tie_eps of a PRN's winner that the guard's slow path re-evaluates at most
constexpr int kMaxRadices = 12;
constexpr int kMaxPassLen = 2048;  // longest vector of a pass: one tile of 2048 complex values (choose_cols), i.e. transforms of up to 2048 x 2048 points
constexpr int kFftThreads = 256;
constexpr int kFftSlots = 8;  // tile elements per thread at most: L*C <= kFftSlots * kFftThreads

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

struct SubPlan {
  int len;
  int nrad;
  int rad[kMaxRadices];
};

struct Plan {
  int n, n1, n2;  // n = n1 * n2; n1 = column length (stride n2), n2 = row length (contiguous)
  SubPlan p1, p2;
};

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

// Division of a small wave-uniform number by a run-time constant of the launch (hops per bin, bins per spectrum, hop groups ...) as one
// multiply-high: q = (x * mul) >> 32 with mul = floor(2^32 / d) + 1 is floor(x / d) whenever x * d < 2^32 (launch_pass checks the
// launch's largest batch number against that).  The pass kernels did these as 64-bit divisions - the compiler's float-reciprocal
// sequences, ~10 of them per fetch: a fifth of the vector instructions of a rows pass that is VALU-bound (BDS B1C, DESIGN.md 4.4 xxv).
struct FDiv {
  unsigned mul, d;
};
inline FDiv make_fdiv(long long d) {
  FDiv f;
  f.d = d > 0 ? (unsigned)d : 0u;
  f.mul = d > 1 ? (unsigned)((1ull << 32) / (unsigned long long)d) + 1u : 0u;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, FDiv f) { return f.d <= 1u ? x : __umulhi(x, f.mul); }
__device__ __forceinline__ unsigned fmodu(unsigned x, FDiv f) { return x - fdiv(x, f) * f.d; }

enum PreOp { PRE_NONE = 0, PRE_IF_CARRIER, PRE_CODE, PRE_MUL_CONJ };
enum PostOp { POST_STORE = 0, POST_TWIDDLE, POST_ABS_ACC };

struct PassArgs {
  // geometry of this pass
  int len;           // vector length L
  int nvec;          // vectors per transform
  int estride;       // element stride (in complex elements)
  int vstride;       // vector stride
  int cols;          // vectors per workgroup tile
  int nrad;
  int rad[kMaxRadices];
  int n;             // full transform size (twiddle table period)
  int inverse;       // 0: exp(-i..), 1: exp(+i..)
  int pre, post;
  long long in_batch_stride;   // elements between transforms of the batch
  long long out_batch_stride;
  const float2* in;
  float2* out;
  const float2* tw;   // exp(-2*pi*i*k/n), k = 0..n-1
  // PRE_IF_CARRIER
  const int8_t* if_base;
  const float2* if_f32;  // the conditioned signal of gc_acq_condition instead of the int8 record (nullptr: the record)
  long long first_sample;
  int spc, nhops;
  double f0, fstep, fs;  // bin frequency f_b = f0 - fstep*b (Hz)
  // PRE_CODE
  const int8_t* codes;  // [batch][spc]
  // PRE_MUL_CONJ
  const float2* other;  // code spectrum (same layout)
  // POST_ABS_ACC: batch index = bin; loops over nhops transforms in*, accumulates |.|/n
  float* acc_out;
  int acc_add;  // POST_ABS_ACC: add to what acc_out already holds (second code arm of the same PRN)
  float acc_scale;  // POST_ABS_ACC: weight of this arm (B1C: sqrt(11/40), sqrt(29/40)); 0 means 1
  // POST_ABS_ACC with few bins: the hops of a bin are split over hop_groups workgroups (otherwise tiles x bins
  // workgroups, ~2 per CU, each walking all the hops); group g's raw sums go to acc_part[g][bin][n] and
  // abs_combine_kernel adds them in group order
  int hop_groups, acc_bins;
  // Hand-over between the inverse transform's two passes in the CONSUMER's tile order: element (row r, column k) of the [OTHER rows][L]
  // intermediate at (k / B) * (rows * B) + r * B + k % B, B = the columns pass's tile width (a power of two) - its tile is then
  // ONE contiguous run instead of `rows` segments of B values (half a 128-byte line each at B = 8).  out_blocked: log2(B) + 1 on
  // the producing rows pass, in_blocked != 0 on the consuming columns pass; 0: natural order.
  int out_blocked, in_blocked;
  // shifted rows pass of the inverse transform: one workgroup walks row_reps consecutive batches (hops of ONE bin: the shift, the
  // source rows, the code-spectrum values and the twiddle tables are the same for all of them); 0 or 1: one batch per workgroup
  int row_reps;
  int bins_per_wg;   // fused columns pass without hop groups: consecutive batches (bins) one workgroup takes, the next one's inputs fetched ahead (0 / 1: one)
  int nbatch_total;  // ... and how many batches the launch has in all
  int no_xcd_pairs;  // 0: every XCD a contiguous run of the strided passes' tiles; 1 (GC_ACQ_NO_XCD_PAIRS): blockIdx -> tile as it comes; 2 (GC_ACQ_XCD_MAP=pairs): neighbours paired (A/B)
  float* acc_part;
  // PRE_MUL_CONJ with circular spectrum shifts (the circshift search family): batch tb reads input transform
  // tb / shift_bins shifted by tb % shift_bins natural-frequency bins; n1, n2 give the [k1][k2] storage order
  int shift_bins, n1, n2;
  // the same with batch tb = bin * nhops + hop (coarse search whose bin spacing is a whole number of FFT bins): reads input
  // transform `hop` shifted by bin * shift_q.  shift_den > 1: the spacing is shift_q / shift_den bins (Galileo E5b: 60 Hz x 2 ms = 3 / 25,
  // Galileo E1: 150 Hz x 8 ms = 6 / 5) - bin b reads transform (b % shift_den) * nhops + hop, one of shift_den x nhops, shifted by
  // (b / shift_den) * shift_q whole bins
  int shift_q;
  int shift_den;
  // rows pass of a data + pilot search with both arms in ONE launch (gc_acquire_coarse_offsets): the launch's first arm_batches batches
  // are arm 0's, the next ones arm 1's ...; arm k multiplies with other + k * n and writes transform (bin * narms + k) * nhops + hop of
  // the intermediate, so that the columns pass sees narms * nhops hops per bin and adds the arms' magnitudes like hops.  0: one arm
  int arm_batches, narms_merged;
  // the columns pass of such a search when the arms have different weights (BDS B1C: sqrt(11/40), sqrt(29/40)): hop r of a bin belongs to
  // arm r / arm_hops and its magnitude counts arm_w[arm] times.  0: every hop counts once
  int arm_hops;
  // the launch's run-time divisors as multiply-high constants (filled by launch_pass)
  FDiv fd_nhops, fd_shift_bins, fd_sden, fd_hg, fd_arm_batches, fd_arm_hops;
  float arm_w[4];
  int shift0;  // whole bins added to every batch's shift (a search around another centre frequency: gc_acquire_coarse_offsets), in [0, n)
  // PRE_IF_CARRIER on a transform longer than the reference's 2*spc (sizes the radix-{2..8} plan cannot take are padded to
  // the next one it can, launch in gc_acquire_coarse_multi): positions wrap_len .. wrap_len + spc - 1 repeat the first spc
  // (mixed) samples, everything behind is zero
  int wrap_len;
  // POST_ABS_ACC without hop groups, last code arm of a PRN: the workgroup's own peak candidate (PeakTrack::publish_slot) over the
  // first peak_valid columns goes to peak_slots[2 * blockIdx.x] - the finished results are not read back by a peak kernel
  // (94 bins x 144 000 columns = 54 MB per PRN in the Galileo E1 search: 45 us of the 190 us a PRN took)
  unsigned long long* peak_slots;
  int peak_valid;
  // ... and, next to each slot, the workgroup's runner-up value (float bits; PeakTrack::m2) for the float64 guard; nullptr: not wanted
  unsigned int* peak_second;
  // first batch of the launch (fft_pass_ct): gc_acq_shift_row recomputes ONE row of a search whose results were not written
  int batch0;
  // POST_ABS_ACC (fft_pass_ct): batch q's sums land at acc_out + (q - acc_row0) * N - the batch search writes PRN k's winning row
  // (batch irow) to slot k of its row buffer with acc_out = slot k, acc_row0 = irow (no pointer formed outside the allocation)
  int acc_row0;
};

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

// The running maximum of a thread / workgroup with MATLAB's first-occurrence rule (acquisition.m:196-198), see the peak kernels below
struct PeakTrack {
  unsigned int m = 0, bin = 0xffffffffu, col = 0xffffffffu;
  // the largest value among all OTHER cells seen (== m when another cell holds the same value): how far the runner-up is from the
  // winner decides whether the float32 ordering can be trusted or the cells go to the float64 guard (acq_guard.h)
  unsigned int m2 = 0;
  __device__ __forceinline__ void see(float v, unsigned int b, unsigned int c) {
    const unsigned int u = __float_as_uint(v);
    if (u > m) {
      m2 = m;
      m = u;
      bin = b;
      col = c;
    } else {
      m2 = max(m2, u);
      if (u == m) {
        bin = min(bin, b);
        col = min(col, c);
      }
    }
  }
  // one pair of atomics per workgroup at most, and none when the workgroup's maximum is below what is already there
  // (every wave of a 4 000-workgroup launch hitting the same two addresses cost 0.37 ms per PRN)
  // the workgroup's candidate in thread 0: {maximum's bits, smallest bin, smallest column among the lanes that hold it}
  __device__ __forceinline__ bool reduce(unsigned long long& ka, unsigned long long& kb, unsigned int* second = nullptr) const {
    __shared__ unsigned int sm[16], sb[16], sc[16], s2[16];  // one entry per wavefront (workgroups of up to 1024 threads)
    unsigned int wm = m;
    for (int off = 32; off > 0; off >>= 1) wm = max(wm, (unsigned int)__shfl_xor((int)wm, off, 64));
    unsigned int b = m == wm ? bin : 0xffffffffu, c = m == wm ? col : 0xffffffffu;
    // the wave's runner-up: every lane's second, every lane's maximum except ONE holder of the wave's (two holders: a tie)
    const unsigned long long holders = __ballot(m == wm);
    unsigned int w2 = (m == wm && __popcll(holders) == 1) ? m2 : (m == wm ? m : max(m, m2));
    for (int off = 32; off > 0; off >>= 1) {
      b = min(b, (unsigned int)__shfl_xor((int)b, off, 64));
      c = min(c, (unsigned int)__shfl_xor((int)c, off, 64));
      w2 = max(w2, (unsigned int)__shfl_xor((int)w2, off, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      sm[wave] = wm;
      sb[wave] = b;
      sc[wave] = c;
      s2[wave] = w2;
    }
    __syncthreads();
    ka = kb = 0;
    if (threadIdx.x != 0) return false;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) {
      if (sm[w] > wm) {
        w2 = max(max(w2, wm), s2[w]);  // the old maximum is now a runner-up
        wm = sm[w];
        b = sb[w];
        c = sc[w];
      } else {
        w2 = max(max(w2, sm[w]), s2[w]);  // (sm[w] == wm: a second holder, w2 becomes wm)
        if (sm[w] == wm) {
          b = min(b, sb[w]);
          c = min(c, sc[w]);
        }
      }
    }
    if (second) *second = w2;
    if (b == 0xffffffffu) return true;  // nothing seen: keys stay 0
    ka = ((unsigned long long)wm << 32) | (unsigned long long)(0xffffffffu - b);
    kb = ((unsigned long long)wm << 32) | (unsigned long long)(0xffffffffu - c);
    return true;
  }
  // one pair of atomics per workgroup at most, and none when the workgroup's maximum is below what is already there
  // (every wave of a 4 000-workgroup launch hitting the same two addresses cost 0.37 ms per PRN)
  // second != nullptr: the PRN's runner-up value (float bits) by the same scheme - whichever of {this workgroup's maximum, the key it
  // displaces} loses goes to *second together with the workgroup's own second
  __device__ __forceinline__ void publish(unsigned long long* keys, unsigned int* second = nullptr) const {
    unsigned long long ka, kb;
    unsigned int w2 = 0;
    if (reduce(ka, kb, &w2) && ka) {
      unsigned int loser = (unsigned int)(ka >> 32);
      if (ka > __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        const unsigned long long old = atomicMax(&keys[0], ka);
        if (ka > old) loser = (unsigned int)(old >> 32);
      }
      if (kb > __hip_atomic_load(&keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&keys[1], kb);
      if (second) {
        w2 = max(w2, loser);
        if (w2 > __hip_atomic_load(second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(second, w2);
      }
    }
  }
  // the same candidate as a plain store into this workgroup's own slot (keys_reduce_kernel picks them up).  Thread 0 of EVERY
  // workgroup stores unconditionally (zeros when it saw nothing): the slot buffer is cleared only when it is allocated
  // second_slot != nullptr: the workgroup's runner-up value (float bits) next to its candidate (keys_reduce_kernel)
  __device__ __forceinline__ void publish_slot(unsigned long long* slot, unsigned int* second_slot = nullptr) const {
    unsigned long long ka, kb;
    unsigned int w2 = 0;
    if (reduce(ka, kb, &w2)) {
      slot[0] = ka;
      slot[1] = kb;
      if (second_slot) *second_slot = w2;
    }
  }
};

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
struct FusedArgs {
  const float2* tw;        // exp(-2 pi i m / N), m < N
  const float2* sig;       // [nsrc][N] signal spectra, [k1][k2]
  const float2* codespec;  // [nprn * narms][N]
  unsigned long long* keys;  // [nprn][2]
  int nbins, nhops, narms;
  int shift_q;             // > 0: bin b reads hop spectrum h shifted by b * shift_q bins; 0: spectrum b * nhops + h as it is
  int valid;               // columns that count for the peak (2 * spc; the transform may be longer)
  float inv_n;
  float weight[4];         // per code arm (B1C: sqrt(11/40), sqrt(29/40)); 0 means 1
};

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

// Peak pick with MATLAB's first-occurrence semantics (acquisition.m:196-198: max(max(results, [], 2)) and max(max(results))):
// the largest value, the smallest bin holding it and the smallest column holding it (not necessarily the same element).
// Positive floats order like their bit patterns, so two 64-bit atomic maxima do it: (bits << 32) | ~bin and
// (bits << 32) | ~column.
// The peak keys of one PRN from the per-workgroup candidates abs_combine_kernel left in `slots` (2 keys per workgroup,
// `per_prn` workgroups per PRN): one workgroup per PRN, launched once after the last PRN.  A thousand workgroups starting
// together and all finding the keys at zero made the two atomics of PeakTrack::publish a 2 000-deep queue on two addresses -
// a third of that kernel's time; plain stores and this one small launch replace them.
__global__ __launch_bounds__(256) void keys_reduce_kernel(const unsigned long long* __restrict__ slots, int per_prn,
                                                          unsigned long long* __restrict__ keys, const unsigned int* __restrict__ sec_slots = nullptr,
                                                          unsigned int* __restrict__ sec_out = nullptr) {
  __shared__ unsigned long long sa[4], sb[4];
  __shared__ unsigned int s2[4], sh[4], top;
  const unsigned long long* mine = slots + (size_t)blockIdx.x * per_prn * 2;
  unsigned long long ka = 0, kb = 0;
  for (int i = threadIdx.x; i < per_prn; i += blockDim.x) {
    ka = max(ka, mine[2 * i]);
    kb = max(kb, mine[2 * i + 1]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    ka = max(ka, (unsigned long long)__shfl_xor((long long)ka, off, 64));
    kb = max(kb, (unsigned long long)__shfl_xor((long long)kb, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    sa[threadIdx.x >> 6] = ka;
    sb[threadIdx.x >> 6] = kb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      ka = max(ka, sa[w]);
      kb = max(kb, sb[w]);
    }
    keys[2 * blockIdx.x] = max(keys[2 * blockIdx.x], ka);
    keys[2 * blockIdx.x + 1] = max(keys[2 * blockIdx.x + 1], kb);
    top = (unsigned int)(ka >> 32);
  }
  if (!sec_slots) return;
  // the PRN's runner-up (float bits): every workgroup's own second, every workgroup's maximum except ONE holder of the PRN's
  __syncthreads();
  const unsigned int* sec = sec_slots + (size_t)blockIdx.x * per_prn;
  const unsigned int m1 = top;
  unsigned int w2 = 0, holders = 0;
  for (int i = threadIdx.x; i < per_prn; i += blockDim.x) {
    const unsigned int mi = (unsigned int)(mine[2 * i] >> 32);
    w2 = max(w2, sec[i]);
    if (mi == m1) ++holders;
    else w2 = max(w2, mi);
  }
  for (int off = 32; off > 0; off >>= 1) {
    w2 = max(w2, (unsigned int)__shfl_xor((int)w2, off, 64));
    holders += (unsigned int)__shfl_xor((int)holders, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s2[threadIdx.x >> 6] = w2;
    sh[threadIdx.x >> 6] = holders;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      w2 = max(w2, s2[w]);
      holders += sh[w];
    }
    if (holders > 1) w2 = m1;
    sec_out[blockIdx.x] = max(sec_out[blockIdx.x], w2);
  }
}

// POST_ABS_ACC with hop groups: results = (add ? results : 0) + (sum over groups, in group order) / n * scale;
// `keys` != nullptr: also the peak pick of the finished results (last code arm of a PRN) over their first `valid` columns
__global__ __launch_bounds__(256) void abs_combine_kernel(const float* __restrict__ part, int groups, int nbins, int n,
                                                          float* __restrict__ out, int add, float inv_n, float scale,
                                                          unsigned long long* keys, int valid, unsigned int* seconds = nullptr) {
  // keys: this launch's slot region (2 keys per workgroup), nullptr: no peak pick
  PeakTrack pk;
  const long long total = (long long)nbins * n;
  // the finished results of a PRN feed nothing but its peak keys: they are not written back (16.7 MB per PRN at the default
  // search); an earlier code arm's sums (keys == nullptr) are what the last arm adds to
  const bool store = keys == nullptr;
  if ((n & 3) == 0) {  // four columns per thread and step: 16-byte loads
    const int n4 = n >> 2;
    for (int bin = blockIdx.y; bin < nbins; bin += gridDim.y)
      for (int c4 = blockIdx.x * blockDim.x + threadIdx.x; c4 < n4; c4 += gridDim.x * blockDim.x) {
        const long long i = (long long)bin * n + 4 * c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < groups; ++g) {
          const float4 t = *reinterpret_cast<const float4*>(part + (long long)g * total + i);
          v.x += t.x;
          v.y += t.y;
          v.z += t.z;
          v.w += t.w;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add) o = *reinterpret_cast<const float4*>(out + i);
        v = make_float4(o.x + v.x * inv_n * scale, o.y + v.y * inv_n * scale, o.z + v.z * inv_n * scale, o.w + v.w * inv_n * scale);
        if (store) *reinterpret_cast<float4*>(out + i) = v;
        const int c = 4 * c4;
        if (c < valid) pk.see(v.x, (unsigned int)bin, (unsigned int)c);
        if (c + 1 < valid) pk.see(v.y, (unsigned int)bin, (unsigned int)(c + 1));
        if (c + 2 < valid) pk.see(v.z, (unsigned int)bin, (unsigned int)(c + 2));
        if (c + 3 < valid) pk.see(v.w, (unsigned int)bin, (unsigned int)(c + 3));
      }
  } else {
    for (int bin = blockIdx.y; bin < nbins; bin += gridDim.y)
      for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const long long i = (long long)bin * n + c;
        float v = 0.0f;
        for (int g = 0; g < groups; ++g) v += part[(long long)g * total + i];
        v = (add ? out[i] : 0.0f) + v * inv_n * scale;
        if (store) out[i] = v;
        if (c < valid) pk.see(v, (unsigned int)bin, (unsigned int)c);
      }
  }
  if (keys) pk.publish_slot(keys + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x), seconds ? seconds + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) : nullptr);
}

// the peak pick alone (results written by the pass kernel itself: no hop groups)
__global__ __launch_bounds__(256) void peak_kernel(const float* __restrict__ r, int nbins, int n, unsigned long long* keys, int valid,
                                                   unsigned int* second = nullptr) {
  PeakTrack pk;
  for (int bin = blockIdx.y; bin < nbins; bin += gridDim.y)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < valid; c += gridDim.x * blockDim.x)
      pk.see(r[(long long)bin * n + c], (unsigned int)bin, (unsigned int)c);
  pk.publish(keys, second);
}

// ---- sigPower inputs: exact integer sums of the first spc samples (acquisition.m:151) -------------------
__global__ void sigpower_kernel(const int8_t* __restrict__ x, long long first, int n, long long* out3) {
  long long si = 0, sq = 0, s2 = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int a = x[2 * (first + i)], b = x[2 * (first + i) + 1];
    si += a;
    sq += b;
    s2 += a * a + b * b;
  }
  atomicAdd((unsigned long long*)&out3[0], (unsigned long long)si);
  atomicAdd((unsigned long long*)&out3[1], (unsigned long long)sq);
  atomicAdd((unsigned long long*)&out3[2], (unsigned long long)s2);
}

// the same sums for the conditioned (complex float) signal: one workgroup, fixed summation order, float64
__global__ __launch_bounds__(1024) void sigpower_f32_kernel(const float2* __restrict__ x, long long first, int n, double* out3) {
  double si = 0.0, sq = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float2 z = x[first + i];
    si += (double)z.x;
    sq += (double)z.y;
    s2 += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
  }
  __shared__ double red[3][1024];
  red[0][threadIdx.x] = si;
  red[1][threadIdx.x] = sq;
  red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 3) out3[threadIdx.x] = red[threadIdx.x][0];
}

// ---- input conditioning (acquisition.m:46-111, row A0) -------------------------------------------------------------------
// filtfilt(b, 1, x) = the signal extended by nfact odd-reflected samples at both ends, filtered forwards with the filter
// starting in the steady state of the first extended sample (for an FIR filter: as if that sample had been there for
// ever), reversed, filtered again the same way, reversed, the extensions dropped.
// Sample i of the IF record as data1 + 1i*data2 (postProcessing.m:88-96): int8 / int16, I/Q, Q/I (GLONASS: tracking.m:227 of its
// packages reads the pair the other way round) or real samples.
__device__ __forceinline__ float2 record_sample(const void* __restrict__ rec, int dtype, int layout, long long i) {
  float a, b = 0.0f;
  if (dtype == GC_I16) {
    const short* x = reinterpret_cast<const short*>(rec);
    if (layout == GC_REAL) {
      a = (float)x[i];
    } else {
      a = (float)x[2 * i];
      b = (float)x[2 * i + 1];
    }
  } else {
    const int8_t* x = reinterpret_cast<const int8_t*>(rec);
    if (layout == GC_REAL) {
      a = (float)x[i];
    } else {
      a = (float)x[2 * i];
      b = (float)x[2 * i + 1];
    }
  }
  return layout == GC_QI ? make_float2(b, a) : make_float2(a, b);
}

// The record's samples [first, first + n) as the complex float signal the searches read with source = CONDITIONED: records that
// are not int8 I/Q (int16 files, postProcessing.m:61-96 dataType; Q/I order; real samples) go through this instead of a kernel
// variant per format in every acquisition pass.
__global__ void record_to_float_kernel(const void* __restrict__ rec, int dtype, int layout, long long first, long long n, float2* __restrict__ out) {
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x)
    out[j] = record_sample(rec, dtype, layout, first + j);
}

__global__ void cond_extend_kernel(const void* __restrict__ x, int dtype, int layout, long long first, long long n, int nfact, float2* __restrict__ xe) {
  const long long ne = n + 2LL * nfact;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < ne; j += (long long)gridDim.x * blockDim.x) {
    auto at = [&](long long i) { return record_sample(x, dtype, layout, first + i); };
    float2 v;
    if (j < nfact) {  // 2*x(1) - x(nfact+1:-1:2)
      const float2 e = at(0), r = at(nfact - j);
      v = make_float2(2.f * e.x - r.x, 2.f * e.y - r.y);
    } else if (j < nfact + n) {
      v = at(j - nfact);
    } else {          // 2*x(end) - x(end-1:-1:end-nfact)
      const float2 e = at(n - 1), r = at(n - 2 - (j - nfact - n));
      v = make_float2(2.f * e.x - r.x, 2.f * e.y - r.y);
    }
    xe[j] = v;
  }
}

// out[m] = sum_k b[k] * in[m - k] (BACK: in[m + k]) with the index clamped to the array: the steady-state start
template <bool BACK>
__global__ __launch_bounds__(256) void cond_fir_kernel(const float2* __restrict__ in, long long ne, const float* __restrict__ b, int nb,
                                                       float2* __restrict__ out) {
  extern __shared__ float2 tile[];  // 256 + nb - 1 inputs
  const long long m0 = (long long)blockIdx.x * 256;
  const int span = 256 + nb - 1;
  for (int i = threadIdx.x; i < span; i += 256) {
    long long j = BACK ? m0 + i : m0 - (nb - 1) + i;
    j = j < 0 ? 0 : (j >= ne ? ne - 1 : j);
    tile[i] = in[j];
  }
  __syncthreads();
  const long long m = m0 + threadIdx.x;
  if (m >= ne) return;
  float sr = 0.f, si = 0.f;
  const float2* t = tile + threadIdx.x + (BACK ? 0 : nb - 1);
  for (int k = 0; k < nb; ++k) {
    const float2 v = BACK ? t[k] : t[-k];
    const float c = b[k];
    sr = fmaf(c, v.x, sr);
    si = fmaf(c, v.y, si);
  }
  out[m] = make_float2(sr, si);
}

// longSignal(index), index = ceil((0:len-1)/newFs*oldFs), index(1) = 1 (acquisition.m:84-91)
__global__ void cond_decimate_kernel(const float2* __restrict__ y, int nfact, double old_fs, double new_fs, long long len,
                                     float2* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
    long long idx = (long long)ceil(__dmul_rn(__ddiv_rn((double)i, new_fs), old_fs));
    if (i == 0) idx = 1;
    out[i] = y[nfact + idx - 1];
  }
}

// ---- fine frequency (acquisition.m:213-238): per-code-period sums of x[n] * code[floor(ts*(n + offset)/tc) mod len] *
// exp(-1i*2*pi*f_bin*n/fs) for every fine bin, several detections per launch (blockIdx.x = code period, .y = detection,
// .z = group of kFineBins bins).  The first version of this kernel evaluated sincos and the float64 code index once per
// (bin, sample) - 21 times the work for the 21 bins of a 500-Hz coarse step - in a launch of its own per detection.  Here
// a sample is read, its code chip looked up and the carrier of the group's middle bin evaluated once; the other bins'
// carriers follow by rotating with exp(-+i*2*pi*fstep*n/fs) (at most kFineBins/2 rotations away from an evaluated
// sincos: ~1e-6 relative, the float32 level of the sums themselves).  Per-thread sums in float64 as before.
struct FineDet {
  long long first;  // absolute index of the detection's first sample
  double f0;        // its first fine bin, Hz
};
constexpr int kFineBins = 24;
constexpr int kFineParts = 8;  // at most this many workgroups per code period (fine_multi_kernel)

template <bool F32>  // F32: the conditioned complex float signal instead of the int8 record
__global__ __launch_bounds__(256) void fine_multi_kernel(const void* __restrict__ xv, const FineDet* __restrict__ det, int spc,
                                                          int ncodes, const int8_t* __restrict__ codes, int code_len, double ts,
                                                          double tc, double fstep, double fs, int nbins, int index_offset,
                                                          float dcr, float dcq, double* __restrict__ out, int parts, size_t part_stride) {
  constexpr int MID = kFineBins / 2;
  // parts > 1: few detections are few workgroups (seven detections x 40 code periods on 256 CUs: 131 us of a 3.3-ms search) - a code
  // period's samples are cut into `parts` runs, one workgroup each, summed in order by fine_parts_kernel
  const int ci = blockIdx.x / parts, part = blockIdx.x - ci * parts, d = blockIdx.y, b0 = blockIdx.z * kFineBins;
  const int run = ((spc + parts - 1) / parts + 255) / 256 * 256, i_lo = part * run, i_hi = min(spc, i_lo + run);
  const int nb = min(kFineBins, nbins - b0);
  const FineDet dd = det[d];
  const int8_t* code = codes + (size_t)d * code_len;
  const double fmid = (dd.f0 - fstep * (double)(b0 + MID)) / fs, fst = fstep / fs;  // cycles per sample
  double sr[kFineBins], si[kFineBins];
#pragma unroll
  for (int k = 0; k < kFineBins; ++k) sr[k] = si[k] = 0.0;
  for (int i = i_lo + threadIdx.x; i < i_hi; i += 256) {
    const long long n = (long long)ci * spc + i;
    // acquisition.m:215-216; tc == 0: the replica is already one entry per sample
    const double cvi = tc > 0.0 ? floor(__ddiv_rn(__dmul_rn(ts, (double)(n + index_offset)), tc)) : (double)(n + index_offset);
    const float c = (float)code[(int)fmod(cvi, (double)code_len)];
    float xr, xq;
    if constexpr (F32) {
      const float2 z = reinterpret_cast<const float2*>(xv)[dd.first + n];
      xr = z.x;
      xq = z.y;
    } else {
      const char2 xs = *reinterpret_cast<const char2*>(reinterpret_cast<const int8_t*>(xv) + 2 * (dd.first + n));
      xr = (float)xs.x;
      xq = (float)xs.y;
    }
    xr -= dcr;
    xq -= dcq;
    const float cr = c * xr, cq = c * xq;
    const double ph = fmid * (double)n, dp = fst * (double)n;
    float sn, cs, sd, cd;
    sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
    sincospif(2.0f * (float)(dp - floor(dp)), &sd, &cd);
    sr[MID] += (double)(cr * cs + cq * sn);
    si[MID] += (double)(cq * cs - cr * sn);
    float wr = cs, wi = sn;  // exp(+i*2*pi*ph_k); bin k+1 is fstep lower: multiply by exp(-i*2*pi*dp)
#pragma unroll
    for (int k = MID + 1; k < kFineBins; ++k) {
      const float tr = wr * cd + wi * sd, ti = wi * cd - wr * sd;
      wr = tr;
      wi = ti;
      if (k < nb) {
        sr[k] += (double)(cr * wr + cq * wi);
        si[k] += (double)(cq * wr - cr * wi);
      }
    }
    wr = cs;
    wi = sn;
#pragma unroll
    for (int k = MID - 1; k >= 0; --k) {
      const float tr = wr * cd - wi * sd, ti = wi * cd + wr * sd;
      wr = tr;
      wi = ti;
      sr[k] += (double)(cr * wr + cq * wi);
      si[k] += (double)(cq * wr - cr * wi);
    }
  }
  __shared__ double red[4][kFineBins][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kFineBins; ++k) {
    double a = sr[k], b = si[k];
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off, 64);
      b += __shfl_down(b, off, 64);
    }
    if (lane == 0) {
      red[wave][k][0] = a;
      red[wave][k][1] = b;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * nb) {
    const int k = threadIdx.x >> 1, q = threadIdx.x & 1;
    out[(size_t)part * part_stride + (((size_t)d * nbins + b0 + k) * ncodes + ci) * 2 + q] = ((red[0][k][q] + red[1][k][q]) + red[2][k][q]) + red[3][k][q];
  }
}

__global__ __launch_bounds__(256) void fine_parts_kernel(const double* __restrict__ part, int parts, size_t n, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double sum = part[i];
  for (int s = 1; s < parts; ++s) sum += part[(size_t)s * n + i];
  out[i] = sum;
}

// GPS L1 C/A fine stage, the part behind the per-code sums (acquisition.m:240-253): for every fine bin the largest |sum of 20 consecutive
// per-code sums| over the 20 navigation-bit-edge hypotheses, then the first bin that holds the largest of those.  One workgroup per
// detection, one thread per bin; every sum is added in the reference's order in float64 with separately rounded operations (what the host
// loop this replaces did: 161 KB of sums per 12 detections came back for 100 000 dependent additions on one core - ~0.1 ms of a 3-ms search).
__global__ __launch_bounds__(64) void fine_l1ca_pick_kernel(const double* __restrict__ sums, int nbins, int ncodes, int* __restrict__ best_bin) {
  __shared__ double pw[64];
  const int d = blockIdx.x, b = threadIdx.x;
  double max_power = 0.0;
  if (b < nbins) {
    const double* hd = sums + ((size_t)d * nbins + b) * ncodes * 2;
    for (int c0 = 0; c0 + 20 <= ncodes && c0 < 20; ++c0) {
      double sr = 0.0, si = 0.0;
      for (int c = c0; c < c0 + 20; ++c) {
        sr = __dadd_rn(sr, hd[2 * c]);
        si = __dadd_rn(si, hd[2 * c + 1]);
      }
      const double pwr = __dsqrt_rn(__dadd_rn(__dmul_rn(sr, sr), __dmul_rn(si, si)));
      max_power = pwr > max_power ? pwr : max_power;   // max(maxPower, comPower), :247
    }
  }
  pw[b] = max_power;
  __syncthreads();
  if (b == 0) {
    double best = -1.0;
    int bb = 0;
    for (int k = 0; k < nbins; ++k)
      if (pw[k] > best) {  // [~, maxFinBin] = max(fineResult): the first maximum, :253
        best = pw[k];
        bb = k;
      }
    best_bin[d] = bb;
  }
}

// One workgroup per row: maximum and its first position (MATLAB's max returns the first maximum).
// Row maxima of a circshift search from the per-workgroup candidates its last pass left (PeakTrack::publish_slot: every workgroup of
// that pass belongs to ONE row; key = value bits << 32 | ~column, so the largest key is the row's maximum at its first column):
// one wave per row.  Replaces writing rows x N sums and reading them back (GPS L2C: 1 GB each way per PRN).
__global__ __launch_bounds__(64) void rowkeys_reduce_kernel(const unsigned long long* __restrict__ slots, int tiles, float* vmax, int* amax) {
  const unsigned long long* mine = slots + (size_t)blockIdx.x * tiles * 2;
  unsigned long long k = 0;
  for (int i = threadIdx.x; i < tiles; i += 64) k = max(k, mine[2 * i + 1]);
  for (int off = 32; off > 0; off >>= 1) k = max(k, (unsigned long long)__shfl_xor((long long)k, off, 64));
  if (threadIdx.x == 0) {
    vmax[blockIdx.x] = __uint_as_float((unsigned int)(k >> 32));
    amax[blockIdx.x] = (int)(0xffffffffu - (unsigned int)(k & 0xffffffffu));
  }
}

__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ r, int ncols, int stride, float* vmax, int* amax) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const float* row = r + (long long)blockIdx.x * stride;
  float best = -1.0f;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < ncols; i += 256) {
    const float v = row[i];
    if (v > best) {
      best = v;
      bi = i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float v = sv[threadIdx.x + off];
      const int i = si[threadIdx.x + off];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) {
        sv[threadIdx.x] = v;
        si[threadIdx.x] = i;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    vmax[blockIdx.x] = sv[0];
    amax[blockIdx.x] = si[0];
  }
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

// tile width C1 of the specialised columns pass for vectors of `len`, `nvec` of them per transform (the shapes of GC_CT_SHAPE below:
// its launch has nvec / C1 workgroups per batch, whatever PassArgs::cols says); 0: no specialised pass
int ct_columns_tile(int len, int nvec) {
  static const int shapes[][3] = {{180, 200, 8}, {150, 160, 8}, {375, 384, 4}, {250, 288, 8}, {600, 600, 5}, {320, 1000, 8}};
  for (const auto& k : shapes)
    if (len == k[0] && nvec == k[1]) return k[2];
  return 0;
}

int launch_pass(gc_context* ctx, PassArgs& a, long long nbatch_groups, bool* used_ct = nullptr) {
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
int choose_cols(int L, int estride = 1) {
  int budget = 2048;
  if (const char* e = GC_TUNE_ENV("GC_ACQ_TILE")) budget = std::max(256, std::atoi(e));  // tuning: elements per tile
  int c = std::max(1, std::min(16, budget / L));
  // strided vectors (the column passes): a tile row is c consecutive float2; whole 64-byte sectors when c is a multiple of 8
  static const int align = [] { const char* e = GC_TUNE_ENV("GC_ACQ_COLS_ALIGN"); return e ? std::atoi(e) : 8; }();
  if (estride != 1 && align > 1 && c >= align) c -= c % align;
  return c;
}

struct AcqScratch {
  int n = 0;
  Plan plan;
  float2* tw = nullptr;       // n
  float2* sig = nullptr;      // nbh * n   signal spectra, layout [k1][k2]
  float2* tmp = nullptr;      // nbh * n   scratch between passes
  float2* codespec = nullptr; // nprn * n
  float* results = nullptr;   // nbins * n: sums of a PRN's earlier code arms (what its last arm adds to).  NOT the finished results of a search:
                              // the last arm of a PRN feeds nothing but its peak keys and is not written back (abs_combine_kernel, fft_pass_ct)
  float* partial = nullptr;   // hop-group sums of the last inverse pass (launch_pass)
  size_t partial_cap = 0;
  // second lane of the PRN loop (gc_acquire_coarse_multi): odd PRNs run on a stream of their own with their own intermediates, so one
  // PRN's columns pass fills the device while the next PRN's rows pass drains (and the other way round)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
  hipStream_t lane_stream[2] = {nullptr, nullptr};  // where the lanes' launches go in the call under way
  float2* tmp2 = nullptr;
  float* results2 = nullptr;
  float* partial2 = nullptr;
  size_t partial2_cap = 0;
  int lane = 0;               // the lane the launches under way belong to (launch_abs_pass picks its partial buffer by it)
  int nlanes = 1;             // lanes of the PRN loop under way: their launches run together, which counts when hop groups are chosen
  // circshift search with the row maxima taken inside the last pass (gc_acq_shift_search): `results` holds nothing then and
  // gc_acq_shift_row transforms the row it is asked for again, with the arms and weights of the search
  bool shift_rows_fused = false;
  int shift_narms = 0;
  double shift_weight[4] = {1.0, 1.0, 1.0, 1.0};
  int8_t* codes = nullptr;    // nprn * spc
  size_t codes_cap = 0;
  long long* sums = nullptr;  // 3 + scratch for argmax
  long long nbh = 0;
  int nprn = 0, nbins = 0;
  // circshift search family (gc_acq_shift_*)
  gc_acq_shift_params shift;  // what `sig` currently holds (n == 0: nothing)
  bool shift_padded = false;  // the block length is no size for the plan: every row has its own carrier, transforms of s->n >= 2*shift.n points
  float* rowmax = nullptr;
  int* rowarg = nullptr;
  // pinned staging for the circshift family's read-backs (row maxima per search, the winning row's n sums per PRN): a copy into the
  // caller's pageable array goes through the runtime's own staging in pieces - 0.15 - 0.3 ms for the 1.4 MB row of a B1C search
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  int shift_rows = 0;
  // gc_acq_shift_search_batch: every PRN's codes, code spectra, row maxima, the winning rows and the picks of one search
  GcBuf b_codes, b_chips, b_codespec, b_rowmax, b_rowarg, b_rows, b_pick;
  int shift_slot_lanes = 1;   // lanes of the batch call under way: launch_abs_pass gives each its own region of row-candidate slots
  unsigned long long* peaks = nullptr;  // per-PRN peak keys of gc_acquire_coarse_multi
  int peaks_cap = 0;
  unsigned long long* slots = nullptr;  // per-workgroup peak candidates of abs_combine_kernel, one region per PRN
  size_t slots_cap = 0;
  unsigned int* sec_slots = nullptr;    // per-workgroup runner-up values (float bits), slots_cap / 2 of them (ensure_slots)
  // the float64 guard (acq_guard.h): per-PRN runner-up, cells, their per-hop values, the slow path's candidate list + count
  GcBuf b_second, b_cells, b_exact, b_list, b_off;
  int guard_ties = 0;          // PRNs of the last search whose runner-up was within gc_acq_tie_eps of the winner (resolved in float64)
  double guard_max_dev = 0.0;  // largest |float32 peak - float64 peak| / float64 peak over the last search's PRNs
  int slots_per_prn = 0;                // workgroups per region in the call under way (0: keys were published directly)
};

// The coarse search's two streams, one pair per device for the whole process (created on first use, never destroyed).  HIP deals
// streams out to a few hardware queues; with a stream pair per context, whether a context's two PRN lanes really ran side by side
// depended on how many streams the process had made before: of six engines in one process the second searched in 3.65 instead of
// 2.77 ms (its lanes one after the other), bench.py's searches ran 20 - 30 % slower than the same searches alone, and streams of
// different priority (the multi.hip remedy) moved the bad case elsewhere and made it worse (5.7 ms).  One pair made back to back and
// used by every context behaves the same for all of them.  (Searches of two contexts on one device at the same time share the
// pair: still correct - every call forks and joins with its own events - and no faster than one after the other.)
struct AcqStreams {
  hipStream_t main = nullptr, lane = nullptr;
};
AcqStreams* acq_streams(int device) {
  static std::mutex mu;
  static AcqStreams pool[64];
  static bool made[64] = {false};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!made[device]) {
    made[device] = true;
    if (hipStreamCreateWithFlags(&pool[device].main, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&pool[device].lane, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      pool[device].main = pool[device].lane = nullptr;
    }
  }
  return pool[device].main && pool[device].lane ? &pool[device] : nullptr;
}

// The two-lane searches' fork / join events, created as a unit: all three exist or none does (a half-made set would leave later calls
// recording and waiting on null events with the lanes never joined into the caller's stream - ADVICE r5).
bool lane_events(AcqScratch* s) {
  if (s->ev_fork && s->ev_join && s->ev_join2) return true;
  hipEvent_t* const evs[3] = {&s->ev_fork, &s->ev_join, &s->ev_join2};
  bool ok = true;
  for (hipEvent_t* e : evs)
    if (!*e && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
      *e = nullptr;
      ok = false;
    }
  if (ok) return true;
  (void)hipGetLastError();
  for (hipEvent_t* e : evs) {
    if (*e) (void)hipEventDestroy(*e);
    *e = nullptr;
  }
  return false;
}

void free_scratch(AcqScratch* s) {
  if (!s) return;
  void* ptrs[] = {s->tw, s->sig, s->tmp, s->codespec, s->results, s->partial, s->codes, s->sums, s->rowmax, s->rowarg, s->peaks, s->slots,
                  s->tmp2, s->results2, s->partial2, s->sec_slots};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (s->stream2) (void)hipStreamDestroy(s->stream2);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->ev_join2) (void)hipEventDestroy(s->ev_join2);
  if (s->pinned) (void)hipHostFree(s->pinned);
  for (GcBuf* b : {&s->b_codes, &s->b_chips, &s->b_codespec, &s->b_rowmax, &s->b_rowarg, &s->b_rows, &s->b_pick, &s->b_second, &s->b_cells, &s->b_exact,
                   &s->b_list, &s->b_off})
    gc_buf_free(*b);
  delete s;
}

// Room for `want` slot keys (two per workgroup) and, next to them, one runner-up value per workgroup (sec_slots).  Both are cleared
// only when they are (re)allocated: every workgroup of a launch stores into its own slot unconditionally.
int ensure_slots(AcqScratch* s, size_t want) {
  if (s->slots_cap >= want) return GC_OK;
  GC_HIP(hipDeviceSynchronize());  // (both lanes of the PRN loop: the buffers are theirs together)
  if (s->slots) (void)hipFree(s->slots);
  if (s->sec_slots) (void)hipFree(s->sec_slots);
  s->slots = nullptr;
  s->sec_slots = nullptr;
  s->slots_cap = 0;
  GC_HIP(hipMalloc((void**)&s->slots, want * sizeof(unsigned long long)));
  GC_HIP(hipMalloc((void**)&s->sec_slots, (want / 2 + 1) * sizeof(unsigned int)));
  GC_HIP(hipMemset(s->slots, 0, want * sizeof(unsigned long long)));
  GC_HIP(hipMemset(s->sec_slots, 0, (want / 2 + 1) * sizeof(unsigned int)));
  GC_HIP(hipDeviceSynchronize());
  s->slots_cap = want;
  return GC_OK;
}

// Last inverse pass (POST_ABS_ACC) over `nbins` bins.  With few bins the launch would have ~2 workgroups per CU, each
// walking all nhops hops of its bin: the hops are then split over hop groups (a divisor of nhops), whose raw sums meet in
// abs_combine_kernel - deterministic, group order fixed.
// bin0 / nbins_total: the launch covers bins bin0 .. bin0 + nbins - 1 of a search of nbins_total (a PRN's bins in chunks, see
// gc_acquire_coarse_multi): no hop groups then, and the chunk's candidates go to their bins' places in the PRN's slot region.
int launch_abs_pass(gc_context* ctx, AcqScratch* s, PassArgs& a, long long nbins, unsigned long long* keys = nullptr, int valid = 0,
                    int ip = 0, int nprn = 1, bool* rows_fused = nullptr, int bin0 = 0, long long nbins_total = 0) {
  if (rows_fused) *rows_fused = false;
  const bool chunked = nbins_total > nbins;
  if (nbins_total < nbins) nbins_total = nbins;
  if (valid <= 0) valid = a.n;
  const int tiles = (a.nvec + a.cols - 1) / a.cols;
  int hg = 1;
  for (int g = 1; g <= a.nhops; ++g)
    if (a.nhops % g == 0 && (long long)tiles * nbins * hg * std::max(1, s->nlanes) < 4LL * ctx->compute_units) hg = g;  // (both lanes' launches run together)
  if (GC_TUNE_ENV("GC_ACQ_NO_HOP_GROUPS") || chunked) hg = 1;
  if (const char* e = GC_TUNE_ENV("GC_ACQ_HOP_GROUPS")) {  // experiments: any divisor of the hop count (a chunk of bins has none)
    const int g = std::atoi(e);
    if (g >= 1 && a.nhops % g == 0 && !chunked) hg = g;
  }
  a.hop_groups = hg;
  dim3 pgrid((unsigned int)std::max(1, std::min((a.n + 1023) / 1024, 64)), (unsigned int)std::min<long long>(nbins, 65535));
  if (const char* e = GC_TUNE_ENV("GC_ACQ_COMBINE_GX")) pgrid.x = (unsigned int)std::max(1, std::atoi(e));
  if (hg == 1) {
    // last arm of a PRN on a specialised pass kernel: every workgroup leaves its own peak candidate (fft_pass_ct), reduced into the
    // keys after the last PRN like the hop-grouped path's; the generic pass kernel writes the results and peak_kernel reads them
    const int c1 = ct_columns_tile(a.len, a.nvec);
    if (rows_fused && c1 > 0 && a.hop_groups <= 1 && !GC_TUNE_ENV("GC_ACQ_GENERIC") && !GC_TUNE_ENV("GC_ACQ_ROWMAX_KERNEL")) {
      // circshift search, last arm: per-workgroup candidates (tiles of one row each) instead of the sums themselves
      const int tiles_ct = a.nvec / c1;
      const size_t want = (size_t)nbins_total * tiles_ct * 2 * (size_t)std::max(1, s->shift_slot_lanes);  // (gc_acq_shift_search_batch: a region per lane)
      if (int rc = ensure_slots(s, want)) return rc;
      unsigned long long* const region = s->slots + (size_t)s->lane * (size_t)nbins_total * tiles_ct * 2 * (s->shift_slot_lanes > 1 ? 1 : 0);
      a.peak_slots = region + (size_t)bin0 * tiles_ct * 2;
      a.peak_valid = valid;
      a.batch0 = bin0;
      bool used_ct = false;
      int rc = launch_pass(ctx, a, nbins, &used_ct);
      a.peak_slots = nullptr;
      a.batch0 = 0;
      if (rc) return rc;
      if (used_ct) {
        hipLaunchKernelGGL(rowkeys_reduce_kernel, dim3((unsigned int)nbins), dim3(64), 0, ctx->stream, region + (size_t)bin0 * tiles_ct * 2, tiles_ct,
                           s->rowmax + bin0, s->rowarg + bin0);
        GC_HIP(hipGetLastError());
        *rows_fused = true;
      }
      return GC_OK;  // (the generic kernel ignored the slots and wrote the sums: the caller runs rowmax_kernel)
    }
    const bool fused_peak = keys && c1 > 0 && !GC_TUNE_ENV("GC_ACQ_GENERIC") && !GC_TUNE_ENV("GC_ACQ_PEAK_KERNEL");
    if (chunked && (c1 == 0 || (keys && !fused_peak))) {
      gc_set_error("acquisition: bins in chunks need the specialised passes and their peak candidates");
      return GC_E_STATE;
    }
    if (fused_peak) {
      const int per = (int)((long long)(a.nvec / c1) * nbins_total);  // the specialised kernel's grid (over all chunks)
      const size_t want = (size_t)nprn * per * 2;
      if (int rc = ensure_slots(s, want)) return rc;
      s->slots_per_prn = per;
      a.peak_slots = s->slots + ((size_t)ip * per + (size_t)bin0 * (a.nvec / c1)) * 2;
      a.peak_second = s->sec_slots + ((size_t)ip * per + (size_t)bin0 * (a.nvec / c1));
      a.peak_valid = valid;
    }
    a.batch0 = bin0;
    bool used_ct = false;
    int rc = launch_pass(ctx, a, nbins, &used_ct);
    a.peak_slots = nullptr;
    a.peak_second = nullptr;
    a.batch0 = 0;
    if (rc || !keys) return rc;
    if (fused_peak && used_ct) return GC_OK;
    if (fused_peak) s->slots_per_prn = 0;  // the generic pass kernel took it after all (tuning knobs): it wrote the results, peak_kernel reads them
    hipLaunchKernelGGL(peak_kernel, pgrid, dim3(256), 0, ctx->stream, a.acc_out, (int)nbins, a.n, keys, valid,
                       s->b_second.p ? (unsigned int*)s->b_second.p + ip : nullptr);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  const size_t need = (size_t)hg * (size_t)nbins * (size_t)a.n;
  float*& part = s->lane ? s->partial2 : s->partial;
  size_t& part_cap = s->lane ? s->partial2_cap : s->partial_cap;
  if (part_cap < need) {
    GC_HIP(hipDeviceSynchronize());
    if (part) (void)hipFree(part);
    part = nullptr;
    part_cap = 0;
    GC_HIP(hipMalloc((void**)&part, need * sizeof(float)));
    part_cap = need;
  }
  a.acc_part = part;
  a.acc_bins = (int)nbins;
  int rc = launch_pass(ctx, a, nbins * hg);
  if (rc) return rc;
  unsigned long long* region = nullptr;
  if (keys) {  // PRN ip of nprn: its own region of candidate slots, reduced into the keys after the last PRN (finish_keys)
    const int per = (int)(pgrid.x * pgrid.y);
    const size_t want = (size_t)nprn * per * 2;
    if (int rc = ensure_slots(s, want)) return rc;
    s->slots_per_prn = per;
    region = s->slots + (size_t)ip * per * 2;
  }
  hipLaunchKernelGGL(abs_combine_kernel, pgrid, dim3(256), 0, ctx->stream, part, hg, (int)nbins, a.n, a.acc_out, a.acc_add,
                     1.0f / (float)a.n, a.acc_scale != 0.0f ? a.acc_scale : 1.0f, region, valid, region ? s->sec_slots + (size_t)ip * (pgrid.x * pgrid.y) : nullptr);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace

void gc_acq_free(gc_context* ctx) {
  free_scratch((AcqScratch*)ctx->acq_scratch);
  ctx->acq_scratch = nullptr;
}

static int ensure_scratch(gc_context* ctx, int n, long long nbh, int nprn, int nbins, int spc, AcqScratch** out) {
  AcqScratch* s = (AcqScratch*)ctx->acq_scratch;
  if (s && s->n == n && s->nbh >= nbh && s->nprn >= nprn && s->nbins >= nbins && s->codes_cap >= (size_t)nprn * spc) {
    *out = s;
    return GC_OK;
  }
  GC_HIP(hipStreamSynchronize(ctx->stream));
  free_scratch(s);
  ctx->acq_scratch = nullptr;
  s = new AcqScratch();
  std::memset(&s->shift, 0, sizeof s->shift);
  if (!make_plan(n, &s->plan)) {
    delete s;
    gc_set_error("acquisition: FFT size %d is not of the form 2^a 3^b 5^c (or its factors are too large)", n);
    return GC_E_UNSUPPORTED;
  }
  s->n = n;
  s->nbh = nbh;
  s->nprn = nprn;
  s->nbins = nbins;
  s->codes_cap = (size_t)nprn * (size_t)std::max(spc, n);
  const size_t ne = (size_t)n;
  if (hipMalloc((void**)&s->tw, ne * sizeof(float2)) != hipSuccess ||
      hipMalloc((void**)&s->sig, (size_t)nbh * ne * sizeof(float2)) != hipSuccess ||
      hipMalloc((void**)&s->tmp, (size_t)nbh * ne * sizeof(float2)) != hipSuccess ||
      hipMalloc((void**)&s->codespec, (size_t)nprn * ne * sizeof(float2)) != hipSuccess ||
      hipMalloc((void**)&s->results, (size_t)nbins * ne * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&s->codes, (size_t)nprn * (size_t)std::max(spc, n)) != hipSuccess ||
      hipMalloc((void**)&s->sums, 16 * sizeof(long long)) != hipSuccess) {
    free_scratch(s);
    gc_set_error("acquisition: device allocation failed");
    return GC_E_NOMEM;
  }
  std::vector<float2> tw(ne);
  for (size_t k = 0; k < ne; ++k) {
    const double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
    tw[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  GC_HIP(hipMemcpy(s->tw, tw.data(), ne * sizeof(float2), hipMemcpyHostToDevice));
  ctx->acq_scratch = s;
  *out = s;
  return GC_OK;
}

// Forward transform of `nbatch` sequences produced by `pre` into `dst` (layout [k1][k2]).
static int forward(gc_context* ctx, AcqScratch* s, PassArgs base, int pre, long long nbatch, float2* dst) {
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

// The guard's slow path for ONE PRN whose runner-up is within eps of its winner: `rerun(ip)` searches the PRN again with the sums of all
// its bins written (s->results, [nbins][n]); every cell at or above `thr` within the first `valid` columns is re-evaluated in float64
// and the reference's rule picks: the largest value, the smallest bin and the smallest column holding it (acquisition.m:196-198).
// More than kGuardListCap such cells (a plateau: a record of zeros, a saturated block - inputs on which the float32 sums are exact
// anyway): the float32 decision stands.
template <class Rerun>
int guard_resolve(gc_context* ctx, AcqScratch* s, const GcExactSetup& ex, int ip, int nbins, int n, int valid, int H, float thr, double f0_row, double fstep,
                  long long first, Rerun rerun, int* bin, int* col, double* val) {
  int rc = rerun(ip);
  if (rc) return rc;
  int* const d_count = (int*)s->b_list.p;
  int2* const d_list = (int2*)((char*)s->b_list.p + 64);
  GC_HIP(hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream));
  rc = gc_collect_cells(ctx->stream, s->results, nbins, (long long)n, valid, thr, d_count, d_list, kGuardListCap);
  if (rc) return rc;
  int count = 0;
  GC_HIP(hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  if (count <= 0 || count > kGuardListCap) return GC_OK;
  std::vector<int2> list((size_t)count);
  GC_HIP(hipMemcpy(list.data(), d_list, list.size() * sizeof(int2), hipMemcpyDeviceToHost));
  std::vector<GcExactCell> cells((size_t)count);
  for (int k = 0; k < count; ++k) {
    GcExactCell& c = cells[(size_t)k];
    c.code = ip;
    c.col = list[(size_t)k].y;
    c.shift = 0;
    c.bin = list[(size_t)k].x;
    c.freq = f0_row - fstep * (double)c.bin;
    c.first = first;
  }
  GC_HIP(hipMemcpyAsync(s->b_cells.p, cells.data(), cells.size() * sizeof(GcExactCell), hipMemcpyHostToDevice, ctx->stream));
  rc = gc_exact_cells(ctx->stream, ex, (const GcExactCell*)s->b_cells.p, count, (double*)s->b_exact.p);
  if (rc) return rc;
  std::vector<double> part((size_t)count * H);
  GC_HIP(hipMemcpyAsync(part.data(), s->b_exact.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  double best = -1.0;
  int bb = 0, bc = 0;
  for (int k = 0; k < count; ++k) {
    double v = 0.0;
    for (int h = 0; h < H; ++h) v += part[(size_t)k * H + h];
    if (v > best) {
      best = v;
      bb = cells[(size_t)k].bin;
      bc = cells[(size_t)k].col;
    } else if (v == best) {
      bb = std::min(bb, cells[(size_t)k].bin);
      bc = std::min(bc, cells[(size_t)k].col);
    }
  }
  *bin = bb;
  *col = bc;
  *val = best;
  return GC_OK;
}

extern "C" int gc_acquire_coarse_multi(gc_context* ctx, const gc_acq_params* p, int nprn, int narms,
                                       const int8_t* sampled_codes, gc_acq_result* out);
extern "C" int gc_acquire_coarse_offsets(gc_context* ctx, const gc_acq_params* p, int nprn, int narms,
                                         const int8_t* sampled_codes, const double* freq_offset, gc_acq_result* out);

extern "C" int gc_acquire_coarse(gc_context* ctx, const gc_acq_params* p, int nprn, const int8_t* sampled_codes,
                                 gc_acq_result* out) {
  return gc_acquire_coarse_multi(ctx, p, nprn, 1, sampled_codes, out);
}

// `narms` sampled codes per PRN (rows prn*narms + arm): results = sum over arms of |ifft(S .* conj(C_arm))|,
// the data+pilot search of GPS_L5C/include/acquisition.m:175-216 (narms = 1: acquisition.m:158-192).
extern "C" int gc_acquire_coarse_multi(gc_context* ctx, const gc_acq_params* p, int nprn, int narms,
                                       const int8_t* sampled_codes, gc_acq_result* out) {
  return gc_acquire_coarse_offsets(ctx, p, nprn, narms, sampled_codes, nullptr, out);
}

extern "C" int gc_acquire_coarse_offsets(gc_context* ctx, const gc_acq_params* p, int nprn, int narms,
                                         const int8_t* sampled_codes, const double* freq_offset, gc_acq_result* out) {
  if (!ctx || !p || nprn <= 0 || narms < 1 || narms > 4 || !sampled_codes || !out) {
    gc_set_error("gc_acquire_coarse: bad arguments");
    return GC_E_INVALID;
  }
  const bool cond = p->source == GC_ACQ_SOURCE_CONDITIONED;
  if (cond) {
    if (ctx->acq_cond_n <= 0) {
      gc_set_error("gc_acquire_coarse: no conditioned signal (call gc_acq_condition first)");
      return GC_E_STATE;
    }
  } else if (!ctx->d_if || ctx->if_dtype != GC_I8 || ctx->if_layout != GC_IQ) {
    gc_set_error("gc_acquire_coarse: needs an int8 I/Q IF buffer");
    return ctx->d_if ? GC_E_UNSUPPORTED : GC_E_STATE;
  }
  const uint64_t avail = cond ? (uint64_t)ctx->acq_cond_n : ctx->if_nsamples;
  const float2* const cond_sig = cond ? (const float2*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p : nullptr;
  GC_HIP(hipSetDevice(ctx->device));
  const double x = p->sampling_freq / (p->code_freq_basis / p->code_length);
  const int spc = (int)std::floor(x + 0.5);                                     // acquisition.m:116
  const int nbins = p->n_bins > 0 ? p->n_bins : (int)std::floor(p->search_band * 2 / p->search_step + 0.5) + 1;  // :124
  const int H = p->non_coh_time;
  if (spc <= 0 || nbins <= 0 || H <= 0 || p->first_sample < 0) return GC_E_INVALID;
  // block and replica lengths: 2*spc and spc in the L1 C/A family; len10PlusXms and samplesXmsLen for a B1C-type search
  const int blk = p->block_len > 0 ? p->block_len : 2 * spc;
  const int cl = p->code_samples > 0 ? p->code_samples : spc;
  if ((p->block_len > 0 || p->code_samples > 0) && (H != 1 || cl > blk)) {
    gc_set_error("gc_acquire_coarse: block_len / code_samples need non_coh_time == 1 and code_samples <= block_len");
    return GC_E_INVALID;
  }
  if ((uint64_t)p->first_sample + (uint64_t)(H - 1) * spc + (uint64_t)blk > avail) {
    gc_set_error("gc_acquire_coarse: needs %lld samples from %lld, buffer holds %llu", (long long)(H - 1) * spc + blk,
                 (long long)p->first_sample, (unsigned long long)avail);
    return GC_E_RANGE;
  }
  // The reference transforms 2*spc points (one code period + one of zeros).  Where the radix-{2..8} plan cannot take that
  // length (2*spc = 32 736 = 2^5*3*11*31 at the common 16.368-Msps front ends, 5 172 = 2^2*3*431 after the A0 resampling),
  // the circular correlation is computed inside a longer transform instead: the 2*spc mixed samples followed by a repeat
  // of their first spc and zeros up to the next size M >= 3*spc the plan takes - for the code of spc samples the first
  // 2*spc lags of that M-point circular correlation ARE the reference's 2*spc-point one, term by term.
  int n = blk;
  bool padded = false;
  {
    Plan probe;
    if (!make_plan(n, &probe) || GC_TUNE_ENV("GC_ACQ_PAD")) {
      padded = true;
      n = 0;
      for (int m = blk + cl; m < blk + cl + (1 << 20); ++m)
        if (make_plan(m, &probe)) {
          n = m;
          break;
        }
      if (n == 0) {
        gc_set_error("acquisition: no transform size at or above %d fits the plan", blk + cl);
        return GC_E_UNSUPPORTED;
      }
    }
  }
  // Both (all) code arms of a PRN in one rows-pass launch and one columns-pass launch when their weights are equal (the data + pilot
  // searches add the arms' magnitudes, GPS_L5C acquisition.m:175-216): the arms' transforms sit next to each other per bin in the
  // intermediate and the columns pass adds them like hops - half the launches, each twice the size, and no sums written by the first
  // arm for the second to read back and add to.  It pays where a bin has few hops - Galileo E1's one: 10.9 -> 9.5 ms - and not where the
  // columns pass already walks 15 - 25 hops per bin and the doubled intermediate needs twice the chunks (L5 6.2 -> 6.4 .. 7.2 ms, E5b 28.5 ->
  // 30.4 .. 33, E5a / B2a +-0): merged up to 4 arm-hops per bin (GC_ACQ_ARMS_MERGE=1: always; GC_ACQ_ARMS_SEPARATE=1: never).
  bool merge_arms = narms > 1 && !GC_TUNE_ENV("GC_ACQ_ARMS_SEPARATE") && !GC_TUNE_ENV("GC_ACQ_FUSED") && !GC_TUNE_ENV("GC_ACQ_GENERIC") &&
                    ((long long)narms * H <= 4 || GC_TUNE_ENV("GC_ACQ_ARMS_MERGE"));
  for (int arm = 1; arm < narms; ++arm) merge_arms = merge_arms && p->arm_weight[arm] == p->arm_weight[0];
  AcqScratch* s = nullptr;
  int rc = ensure_scratch(ctx, n, (long long)nbins * H * (merge_arms ? narms : 1), nprn * narms, nbins, cl, &s);
  if (rc) return rc;
  s->shift.n = 0;  // the signal spectra of a circshift search, if any, are overwritten below
  const Plan& pl = s->plan;

  // sigPower = sqrt(var(x(1:spc)) * spc), var normalised by N-1 (acquisition.m:151)
  GC_HIP(hipMemsetAsync(s->sums, 0, 16 * sizeof(long long), ctx->stream));
  if (cond)
    hipLaunchKernelGGL(sigpower_f32_kernel, dim3(1), dim3(1024), 0, ctx->stream, cond_sig, (long long)p->first_sample, cl, (double*)s->sums);
  else
    hipLaunchKernelGGL(sigpower_kernel, dim3(64), dim3(256), 0, ctx->stream, (const int8_t*)ctx->d_if, (long long)p->first_sample,
                       cl, s->sums);
  long long hs[3];
  GC_HIP(hipMemcpyAsync(hs, s->sums, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipMemcpyAsync(s->codes, sampled_codes, (size_t)nprn * narms * cl, hipMemcpyHostToDevice, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  // from here on the call runs on the device's search streams (acq_streams); the context's own is idle and comes back at every return
  struct RestoreStream {
    gc_context* c;
    hipStream_t own;
    ~RestoreStream() { c->stream = own; }
  } restore_stream{ctx, ctx->stream};
  const char* lane_streams_env = GC_TUNE_ENV("GC_ACQ_LANE_STREAMS");
  AcqStreams* const shared = (lane_streams_env && std::strcmp(lane_streams_env, "own") == 0) ? nullptr : acq_streams(ctx->device);
  if (shared) ctx->stream = shared->main;
  double sum3[3];
  if (cond) std::memcpy(sum3, hs, sizeof sum3);  // the float kernel wrote doubles
  else for (int k = 0; k < 3; ++k) sum3[k] = (double)hs[k];
  const double mr = sum3[0] / cl, mi = sum3[1] / cl;
  const double var = (sum3[2] - cl * (mr * mr + mi * mi)) / (cl - 1);
  const double sig_power = std::sqrt(var * cl);

  // signal spectra for every (bin, hop)
  PassArgs base;
  std::memset(&base, 0, sizeof base);
  base.if_base = (const int8_t*)ctx->d_if;
  base.if_f32 = cond_sig;
  base.first_sample = p->first_sample;
  base.spc = (p->block_len > 0 || p->code_samples > 0) ? cl : spc;  // hop stride and replica length coincide in the L1 C/A family; one hop otherwise
  base.nhops = H;
  base.f0 = p->intermediate_freq + p->search_band;  // coarseFreqBin(1), :169
  base.fstep = p->search_step;
  base.fs = p->sampling_freq;
  // When the bin spacing is a whole number q of FFT bins (500 Hz x 2 ms = 1 at every default front end), the spectrum of
  // bin b is the spectrum of bin 0 moved by b*q positions: x .* exp(-1i*(f0 - b*step)*phasePoints) =
  // (x .* exp(-1i*f0*phasePoints)) .* exp(+2i*pi*b*q*n/N).  H spectra are then computed instead of nbins*H and the
  // inverse transforms read them shifted (5.8 MB that stay in cache instead of 167 MB from HBM per PRN).
  // A spacing of q / den bins (den > 1: Galileo E5b's 60 Hz x 2 ms = 3 / 25, Galileo E1's 150 Hz x 8 ms = 6 / 5) leaves den classes of bins,
  // b % den, each a whole-bin shift of its class's first bin: den x H spectra instead of nbins x H (E5b: 375 for 2 520, 108 MB that the
  // rows passes of all 72 code arms find in the last-level cache instead of 725 MB from HBM each; GC_ACQ_NO_RATIONAL_SHIFT=1: whole bins only)
  const double qd = p->search_step * (double)n / p->sampling_freq;
  long long q = 0;
  int den = 1;
  {
    const int den_max = GC_TUNE_ENV("GC_ACQ_NO_RATIONAL_SHIFT") || GC_TUNE_ENV("GC_ACQ_FUSED") ? 1 : std::min(64, nbins / 2);
    for (int d = 1; d <= den_max && q == 0; ++d) {
      const double qq = qd * d, r = std::floor(qq + 0.5);
      if (r >= 1 && std::fabs(qq - r) <= 1e-12 * qq) {
        q = (long long)r;
        den = d;
      }
    }
  }
  const bool shifted = !padded && q >= 1 && (long long)((nbins - 1) / den) * q < n && GC_TUNE_ENV("GC_ACQ_NO_SHIFT") == nullptr;
  if (!shifted) den = 1;
  // per-row centre frequencies (gc_acquire_coarse_offsets): row ip searches around IF + freq_offset[ip] - the same signal spectra moved
  // by -freq_offset * N / fs bins, which must be whole bins (GLONASS: 562.5 kHz x 2 ms = 1 125)
  std::vector<int> row_shift((size_t)nprn, 0);
  if (freq_offset) {
    for (int ip = 0; ip < nprn; ++ip) {
      const double b = -freq_offset[ip] * (double)n / p->sampling_freq, r = std::floor(b + 0.5);
      if (!shifted || std::fabs(b - r) > 1e-9 * std::max(1.0, std::fabs(b)) || GC_TUNE_ENV("GC_ACQ_FUSED")) {
        gc_set_error("gc_acquire_coarse_offsets: a row's offset of %.3f Hz is not a whole number of the search's FFT bins (%.6f Hz), or the "
                     "search does not run on shifted spectra", freq_offset[ip], p->sampling_freq / n);
        return GC_E_UNSUPPORTED;
      }
      const long long m = (long long)r % n;
      row_shift[(size_t)ip] = (int)(m < 0 ? m + n : m);
    }
  }
  base.wrap_len = padded ? blk : 0;
  rc = forward(ctx, s, base, PRE_IF_CARRIER, shifted ? (long long)den * H : (long long)nbins * H, s->sig);
  if (rc) return rc;
  // code spectra (conj applied at the product)
  base.codes = s->codes;
  rc = forward(ctx, s, base, PRE_CODE, (long long)nprn * narms, s->codespec);
  if (rc) return rc;

  if (s->peaks_cap < nprn) {
    if (s->peaks) (void)hipFree(s->peaks);
    s->peaks = nullptr;
    s->peaks_cap = 0;
    GC_HIP(hipMalloc((void**)&s->peaks, (size_t)nprn * 2 * sizeof(unsigned long long)));
    s->peaks_cap = nprn;
  }
  // per-PRN peak keys {(bits << 32) | ~bin, (bits << 32) | ~column}, read back once after the last PRN
  unsigned long long* const peaks = s->peaks;
  GC_HIP(hipMemsetAsync(peaks, 0, (size_t)nprn * 2 * sizeof(unsigned long long), ctx->stream));
  s->slots_per_prn = 0;
  // the float64 guard's buffers: per-PRN runner-up (cleared like the keys), winner cells, their per-hop values, per-row offsets
  if (gc_buf_reserve(s->b_second, (size_t)nprn * sizeof(unsigned int), false) != hipSuccess ||
      gc_buf_reserve(s->b_cells, (size_t)std::max(nprn, kGuardListCap) * sizeof(GcExactCell), false) != hipSuccess ||
      gc_buf_reserve(s->b_exact, (size_t)std::max(nprn, kGuardListCap) * H * sizeof(double), false) != hipSuccess ||
      gc_buf_reserve(s->b_list, (size_t)kGuardListCap * sizeof(int2) + 64, false) != hipSuccess ||
      gc_buf_reserve(s->b_off, (size_t)nprn * sizeof(double), false) != hipSuccess) {
    (void)hipGetLastError();
    gc_set_error("acquisition: no memory for the guard's buffers");
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemsetAsync(s->b_second.p, 0, (size_t)nprn * sizeof(unsigned int), ctx->stream));
  if (freq_offset) GC_HIP(hipMemcpyAsync(s->b_off.p, freq_offset, (size_t)nprn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));

  // GC_ACQ_FUSED=1: the whole inverse side in one launch where a fused kernel exists for the plan (acq_fused_kernel: N = 36 000 -
  // GPS L1 C/A, L5, Galileo E5a / E5b, BDS B2a / B3I at 18 Msps - and N = 24 000, GLONASS at 12 Msps).  Same results (the parity
  // tests run both), no intermediate in memory - and measured SLOWER than the two passes (6.7 against 5.4 ms for the default
  // search), so the two passes stay the default.
  bool fused = false;
  {
    const char* ev = GC_TUNE_ENV("GC_ACQ_FUSED");
    if (ev && std::atoi(ev) != 0) {
      FusedArgs fa;
      std::memset(&fa, 0, sizeof fa);
      fa.tw = s->tw;
      fa.sig = s->sig;
      fa.codespec = s->codespec;
      fa.keys = peaks;
      fa.nbins = nbins;
      fa.nhops = H;
      fa.narms = narms;
      fa.shift_q = shifted ? (int)q : 0;
      fa.valid = blk;
      fa.inv_n = 1.0f / (float)pl.n;
      for (int arm = 0; arm < narms && arm < 4; ++arm) fa.weight[arm] = (float)p->arm_weight[arm];
      fused = try_fused<180, 200, 10, 10, 10, 9>(ctx, pl, fa, nprn) || try_fused<150, 160, 10, 10, 10, 6>(ctx, pl, fa, nprn);
      if (fused) GC_HIP(hipGetLastError());
    }
  }
  const int hblock = base.wrap_len > 0 ? 0 : handover_block(pl);
  merge_arms = merge_arms && hblock && shifted && ct_columns_tile(pl.p1.len, pl.n2) > 0 && !GC_TUNE_ENV("GC_ACQ_PEAK_KERNEL");
  const int marms = merge_arms ? narms : 1;  // arms per launch
  // shifted spectra on a specialised plan: a workgroup of the rows pass walks several hops of its bin (same rotation, same code
  // spectrum values, same twiddle tables), as long as the launch keeps ~8 workgroups per CU; GC_ACQ_ROW_REPS overrides (a divisor of H)
  int row_reps = 1;
  if (hblock && shifted) {
    const long long wgs = (long long)(pl.n1 / 6 > 0 ? pl.n1 / 6 : 1) * nbins * H;  // tiles of about six rows
    for (int g = 1; g <= H && g <= 8; ++g)
      if (H % g == 0 && wgs / g >= 8LL * ctx->compute_units) row_reps = g;
    if (const char* e = GC_TUNE_ENV("GC_ACQ_ROW_REPS")) {
      const int g = std::atoi(e);
      if (g >= 1 && H % g == 0) row_reps = g;
    }
  }
  // Two lanes: even PRNs on the context's stream, odd PRNs on a second one with intermediates of their own (GC_ACQ_LANES=1: one lane).
  // A PRN is three dependent launches (rows pass, columns pass, combine) of a few thousand workgroups each: alone, every launch
  // ends in a tail of half-empty CUs and starts after a gap; two independent chains fill each other's.
  int lanes = (nprn > 1 && !fused) ? 2 : 1;
  if (const char* e = GC_TUNE_ENV("GC_ACQ_LANES")) lanes = std::max(1, std::min(2, std::atoi(e)));
  if (lanes == 2) {
    const size_t ne = (size_t)pl.n;
    // the lanes' streams: the device's pair (the first lane on the one the call runs on), or - GC_ACQ_LANE_STREAMS=own - the
    // context's stream and a second one of its own
    if (shared) {
      s->lane_stream[0] = shared->main;
      s->lane_stream[1] = shared->lane;
    } else {
      if (!s->stream2 && hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking) != hipSuccess) lanes = 1;
      s->lane_stream[0] = ctx->stream;
      s->lane_stream[1] = s->stream2;
    }
    if (lanes == 2 && !lane_events(s)) lanes = 1;
    if (lanes == 2 && !s->tmp2 &&
        (hipMalloc((void**)&s->tmp2, (size_t)s->nbh * ne * sizeof(float2)) != hipSuccess ||
         hipMalloc((void**)&s->results2, (size_t)s->nbins * ne * sizeof(float)) != hipSuccess)) {
      (void)hipGetLastError();
      if (s->tmp2) (void)hipFree(s->tmp2);
      s->tmp2 = nullptr;
      lanes = 1;  // no room for a second set of intermediates: one lane
    }
  }
  hipStream_t const stream1 = ctx->stream;
  if (lanes == 2) {
    GC_HIP(hipEventRecord(s->ev_fork, stream1));  // spectra, code spectra and the cleared keys are ready
    for (hipStream_t ls : s->lane_stream)
      if (ls != stream1) GC_HIP(hipStreamWaitEvent(ls, s->ev_fork, 0));
  }
  // Bins in chunks (specialised passes only): a PRN's bins are searched in `chunks` parts after one another, the lanes take (PRN, chunk)
  // items in turn - both lanes' intermediates together are then 1 / chunks of lanes x nbins x H x N x 8 bytes: 334 MB at the default
  // L1 C/A size, 302 MB at L5's, 1.45 GB at Galileo E5b's - more than the 256 MB last-level cache in front of HBM holds; in parts that
  // fit (with the signal spectra the rows passes read) L1 C/A 3.10 -> 2.99 ms, L5 6.61 -> 6.38, E5b 34.2 -> 28.6 ms (12 parts of 14 bins;
  // 2 .. 8 parts, which do not fit next to its 108 MB of spectra, gain nothing; 16 parts 30.5 ms), Galileo E1 11.7 -> 11.1 ms in halves
  // (two arms: the 54 MB of sums per lane that the pilot arm adds to count too).  A search that fits anyway stays whole (smaller
  // launches fill the device less well: E5a +8 % in halves).  GC_ACQ_BIN_CHUNKS=n overrides.
  int chunks = 1;
  if (hblock && !fused && ct_columns_tile(pl.p1.len, pl.n2) > 0 && !GC_TUNE_ENV("GC_ACQ_GENERIC") && !GC_TUNE_ENV("GC_ACQ_PEAK_KERNEL")) {
    // the fewest chunks (of at least 8 bins) that bring the lanes' intermediates (+ the sums a second code arm adds to) + the signal
    // spectra under ~235 MB; none if nothing does
    const double hop_bytes = (double)H * (double)pl.n * sizeof(float2);
    const double per_bin = marms * hop_bytes + ((narms > 1 && !merge_arms) ? (double)pl.n * sizeof(float) : 0.0),
                 spectra = (shifted ? (double)den : (double)nbins) * hop_bytes;
    const double room = 236.0 * 1024 * 1024;
    for (int c = 1; c <= nbins / 8; ++c)
      if ((double)lanes * ((nbins + c - 1) / c) * per_bin + spectra <= room) {
        chunks = c;
        break;
      }
    if (GC_TUNE_ENV("GC_ACQ_HOP_GROUPS")) chunks = 1;  // (hop groups are a property of whole searches)
    if (const char* e = GC_TUNE_ENV("GC_ACQ_BIN_CHUNKS")) chunks = std::max(1, std::min(nbins, std::atoi(e)));
  }
  const int chunk_bins = (nbins + chunks - 1) / chunks;
  chunks = (nbins + chunk_bins - 1) / chunk_bins;
  int lane_rc = GC_OK;
  // One (PRN, chunk of bins) item on the lane and stream that s->lane / ctx->stream name.  with_keys: the item's peak candidates go to
  // the PRN's keys (and its runner-up to s->b_second); without, the sums of ALL the item's bins are written to the lane's `results`
  // (the float64 guard's slow path collects its candidate cells from them).
  auto run_item = [&](int ip, int bin0, int cb, bool with_keys) -> int {
    // (a chunk's batches keep their numbers, bin0 * H on: the chunk's first batch sits at the start of the lane's intermediate)
    float2* const tmp = (s->lane ? s->tmp2 : s->tmp) - (size_t)bin0 * H * marms * (size_t)pl.n;
    float* const results = s->lane ? s->results2 : s->results;
    for (int arm = 0; arm < (merge_arms ? 1 : narms); ++arm) {
      // I1: rows of the product S .* conj(Ccode) (length n2, contiguous), inverse, twiddle
      PassArgs a = base;
      a.n = pl.n;
      a.tw = s->tw;
      a.inverse = 1;
      fill_sub(a, pl.p2);
      a.nvec = pl.n1;
      a.estride = 1;
      a.vstride = pl.n2;
      a.cols = choose_cols(a.len, a.estride);
      a.pre = PRE_MUL_CONJ;
      a.post = POST_TWIDDLE;
      a.in = s->sig;
      a.in_batch_stride = pl.n;
      a.shift_q = shifted ? (int)q : 0;
      a.shift_den = den;
      a.shift0 = row_shift[(size_t)ip];
      a.n1 = pl.n1;
      a.n2 = pl.n2;
      a.other = s->codespec + ((size_t)ip * narms + arm) * pl.n;
      a.out = tmp;
      a.out_batch_stride = pl.n;
      a.out_blocked = hblock;  // the intermediate in the columns pass's tile order
      a.row_reps = row_reps;
      a.batch0 = (int)((long long)bin0 * H / row_reps);
      a.arm_batches = merge_arms ? (int)((long long)cb * H / row_reps) : 0;
      a.narms_merged = marms;
      int rc = launch_pass(ctx, a, (long long)marms * cb * H / row_reps);
      a.batch0 = 0;
      a.arm_batches = 0;
      a.nhops = marms * H;  // the columns pass adds the arms of a bin like hops
      if (rc) return rc;
      // I2: columns (length n1, stride n2), inverse, |.|/n accumulated over the hops of each bin
      a.out_blocked = 0;
      a.row_reps = 0;
      a.in_blocked = hblock;
      fill_sub(a, pl.p1);
      a.nvec = pl.n2;
      a.estride = pl.n2;
      a.vstride = 1;
      a.cols = choose_cols(a.len, a.estride);
      a.pre = PRE_NONE;
      a.post = POST_ABS_ACC;
      a.in = tmp;
      a.acc_out = results;
      a.acc_add = arm > 0;
      a.acc_scale = (float)p->arm_weight[arm];  // 0: 1
      rc = launch_abs_pass(ctx, s, a, cb, (with_keys && (merge_arms || arm == narms - 1)) ? peaks + 2 * ip : nullptr, blk, ip, nprn, nullptr, bin0, nbins);
      if (rc) return rc;
    }
    return GC_OK;
  };
  for (int item = 0; item < nprn * chunks && !fused && lane_rc == GC_OK; ++item) {
    const int ip = item / chunks, bin0 = (item % chunks) * chunk_bins, cb = std::min(chunk_bins, nbins - bin0);
    s->lane = lanes == 2 ? (item & 1) : 0;
    s->nlanes = lanes;
    ctx->stream = lanes == 2 ? s->lane_stream[s->lane] : stream1;  // launch_pass / launch_abs_pass launch on the context's stream
    lane_rc = run_item(ip, bin0, cb, true);
  }
  ctx->stream = stream1;
  s->lane = 0;
  s->nlanes = 1;
  if (lanes == 2) {  // the lanes join before the keys are reduced and read back (also on an error: nothing may still run on them)
    hipEvent_t const ej[2] = {s->ev_join, s->ev_join2};
    for (int k = 0; k < 2; ++k)
      if (s->lane_stream[k] != stream1) {
        (void)hipEventRecord(ej[k], s->lane_stream[k]);
        (void)hipStreamWaitEvent(stream1, ej[k], 0);
      }
  }
  if (lane_rc != GC_OK) {
    (void)hipDeviceSynchronize();
    return lane_rc;
  }
  unsigned int* const seconds = (unsigned int*)s->b_second.p;
  if (!fused && s->slots_per_prn) {
    hipLaunchKernelGGL(keys_reduce_kernel, dim3((unsigned int)nprn), dim3(256), 0, ctx->stream, s->slots, s->slots_per_prn, peaks, s->sec_slots, seconds);
    GC_HIP(hipGetLastError());
  }
  // ---- the float64 guard (acq_guard.h; GC_ACQ_NO_GUARD=1 in the tuning build: the float32 values as before) ----------------------
  // Always: the winner's cell of every PRN again in float64, so that peak / peakMetric - the numbers the caller thresholds
  // (acquisition.m:200-206) - carry no float32 transform error.  The cells are decoded from the keys on the device: one read-back.
  const bool guard = !fused && GC_TUNE_ENV("GC_ACQ_NO_GUARD") == nullptr;
  GcExactSetup ex;
  ex.if_i8 = cond ? nullptr : (const int8_t*)ctx->d_if;
  ex.if_f32 = cond_sig;
  ex.blk = blk;
  ex.cl = cl;
  ex.hop_stride = base.spc;
  ex.nhops = H;
  ex.narms = narms;
  for (int arm = 0; arm < narms; ++arm) ex.w[arm] = p->arm_weight[arm] != 0.0 ? p->arm_weight[arm] : 1.0;
  ex.codes = s->codes;
  ex.code_stride = cl;
  ex.fs = p->sampling_freq;
  GcExactCell* const d_cells = (GcExactCell*)s->b_cells.p;
  double* const d_exact = (double*)s->b_exact.p;
  if (guard) {
    rc = gc_exact_cells_from_keys(ctx->stream, peaks, nprn, base.f0, base.fstep, freq_offset ? (const double*)s->b_off.p : nullptr, p->first_sample, d_cells);
    if (!rc) rc = gc_exact_cells(ctx->stream, ex, d_cells, nprn, d_exact);
    if (rc) return rc;
  }
  std::vector<unsigned long long> hpeaks((size_t)nprn * 2);
  std::vector<unsigned int> hsec((size_t)nprn, 0u);
  std::vector<double> hexact((size_t)nprn * H, 0.0);
  GC_HIP(hipMemcpyAsync(hpeaks.data(), peaks, hpeaks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  if (guard) {
    GC_HIP(hipMemcpyAsync(hsec.data(), seconds, hsec.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipMemcpyAsync(hexact.data(), d_exact, hexact.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  GC_HIP(hipStreamSynchronize(ctx->stream));
  const double eps = gc_acq_tie_eps(pl.n);
  s->guard_ties = 0;
  s->guard_max_dev = 0.0;
  for (int ip = 0; ip < nprn; ++ip) {
    int harg[2] = {(int)(0xffffffffu - (unsigned int)(hpeaks[2 * ip] & 0xffffffffu)),
                   (int)(0xffffffffu - (unsigned int)(hpeaks[2 * ip + 1] & 0xffffffffu))};
    const unsigned int bits = (unsigned int)(hpeaks[2 * ip] >> 32);
    float peak32;
    std::memcpy(&peak32, &bits, sizeof peak32);
    double peak = (double)peak32;
    if (guard && hpeaks[2 * ip] != 0) {
      float second32;
      std::memcpy(&second32, &hsec[(size_t)ip], sizeof second32);
      double exact = 0.0;
      for (int h = 0; h < H; ++h) exact += hexact[(size_t)ip * H + h];  // hop order, acquisition.m:186-190
      if (exact > 0.0) s->guard_max_dev = std::max(s->guard_max_dev, std::fabs((double)peak32 - exact) / exact);
      // Near-tie: another cell within eps of the winner.  Which of them is the larger - and so codePhase, the coarse bin, everything
      // the fine stage is then run on - is decided on float64 values of ALL the cells that close, by the reference's rule: the largest
      // value, the first bin and the first column that hold it (acquisition.m:196-198).
      if (peak32 > 0.0f && (double)second32 >= (double)peak32 * (1.0 - eps)) {
        ++s->guard_ties;
        int gbin = harg[0], gcol = harg[1];
        double gval = exact;
        rc = guard_resolve(ctx, s, ex, ip, nbins, pl.n, blk, H, (float)((double)peak32 * (1.0 - eps)), base.f0 + (freq_offset ? freq_offset[ip] : 0.0), base.fstep,
                           p->first_sample, [&](int q) { s->lane = 0; s->nlanes = 1; return run_item(q, 0, nbins, false); }, &gbin, &gcol, &gval);
        if (rc) return rc;
        harg[0] = gbin;
        harg[1] = gcol;
        exact = gval;
      }
      peak = exact;
    }
    out[ip].coarse_bin = harg[0] + 1;   // 1-based like MATLAB
    out[ip].code_phase = harg[1] + 1;
    out[ip].peak = peak;
    out[ip].peak_metric = peak / sig_power / H;  // :200
    out[ip].coarse_freq = p->intermediate_freq + (freq_offset ? freq_offset[ip] : 0.0) + p->search_band - p->search_step * harg[0];
  }
  return GC_OK;
}



// ---- input conditioning (row A0) ----------------------------------------------------------------------------------------
// fir1(order, [w1 w2]): Hamming-windowed ideal band-pass, scaled to unit gain at the centre of the pass band (float64 here)
static std::vector<double> fir1_bandpass(int order, double w1, double w2) {
  const int nb = order + 1;
  const double alpha = 0.5 * order, pi = 3.14159265358979323846;
  std::vector<double> h((size_t)nb);
  auto sinc = [&](double x) { return x == 0.0 ? 1.0 : std::sin(pi * x) / (pi * x); };
  for (int n = 0; n < nb; ++n) {
    const double m = n - alpha;
    h[n] = (w2 * sinc(w2 * m) - w1 * sinc(w1 * m)) * (0.54 - 0.46 * std::cos(2.0 * pi * n / order));
  }
  const double fc = 0.5 * (w1 + w2);
  double g = 0.0;
  for (int n = 0; n < nb; ++n) g += h[n] * std::cos(pi * (n - alpha) * fc);
  for (double& v : h) v /= g;
  return h;
}

extern "C" int gc_acq_condition(gc_context* ctx, const gc_acq_front_params* p, gc_acq_front_result* out) {
  if (!ctx || !p || !out || p->n_samples <= 0 || p->first_sample < 0 || p->fir_order < 2 || p->fir_order > 4096 ||
      !(p->sampling_freq > 0) || !(p->bandwidth > 0)) {
    gc_set_error("gc_acq_condition: bad arguments");
    return GC_E_INVALID;
  }
  if (!ctx->d_if) {
    gc_set_error("gc_acq_condition: no IF record loaded");
    return GC_E_STATE;
  }
  const long long n = p->n_samples;
  const int nb = p->fir_order + 1, nfact = 3 * (nb - 1);  // filtfilt's edge length
  if ((uint64_t)p->first_sample + (uint64_t)n > ctx->if_nsamples || n <= nfact) {
    gc_set_error("gc_acq_condition: %lld samples from %lld: outside the record, or not longer than filtfilt's %d edge samples", n,
                 (long long)p->first_sample, nfact);
    return GC_E_RANGE;
  }
  const double fs = p->sampling_freq, IF = p->intermediate_freq, BW = p->bandwidth;
  const double w1 = (IF - BW / 2) * 2 / fs - p->band_margin, w2 = (IF + BW / 2) * 2 / fs + p->band_margin;  // acquisition.m:60-62, L5 :69
  if (!(w1 > 0.0) || !(w2 < 1.0)) {
    gc_set_error("gc_acq_condition: band edges %g .. %g of the Nyquist frequency (fir1 needs 0 < w < 1)", w1, w2);
    return GC_E_INVALID;
  }
  const std::vector<double> hd = fir1_bandpass(p->fir_order, w1, w2);
  std::vector<float> hf(hd.begin(), hd.end());
  // resampling frequency from the band-pass sampling bounds (:70-89)
  const double fu = IF + BW / 2, fl = IF - BW / 2;
  double nz = std::floor(fu / BW);
  if (nz < 1) nz = 1;
  const double lower = 2 * fu / nz, upper = nz > 1 ? 2 * fl / (nz - 1) : lower;
  const double new_fs = std::ceil((lower + upper) / 2);
  const long long len = (long long)std::floor((double)(n - 1) / fs * new_fs);  // :84
  if (len <= 0) return GC_E_INVALID;
  GC_HIP(hipSetDevice(ctx->device));
  const long long ne = n + 2LL * nfact;
  GcBuf& bsig = ctx->acqbuf[gc_context::ACQ_COND_SIG];
  GcBuf& ba = ctx->acqbuf[gc_context::ACQ_COND_A];
  GcBuf& bb = ctx->acqbuf[gc_context::ACQ_COND_B];
  GcBuf& bt = ctx->acqbuf[gc_context::ACQ_COND_TAPS];
  ctx->acq_cond_n = 0;
  if (gc_buf_reserve(bsig, (size_t)len * sizeof(float2), false) != hipSuccess || gc_buf_reserve(ba, (size_t)ne * sizeof(float2), false) != hipSuccess ||
      gc_buf_reserve(bb, (size_t)ne * sizeof(float2), false) != hipSuccess || gc_buf_reserve(bt, (size_t)nb * sizeof(float), false) != hipSuccess) {
    gc_set_error("gc_acq_condition: device allocation failed");
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemcpyAsync(bt.p, hf.data(), (size_t)nb * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  const unsigned int nblk = (unsigned int)((ne + 255) / 256);
  const size_t smem = (size_t)(256 + nb - 1) * sizeof(float2);
  hipLaunchKernelGGL(cond_extend_kernel, dim3(std::min(nblk, 65535u)), dim3(256), 0, ctx->stream, (const void*)ctx->d_if, ctx->if_dtype,
                     ctx->if_layout, (long long)p->first_sample, n, nfact, (float2*)ba.p);
  hipLaunchKernelGGL(cond_fir_kernel<false>, dim3(nblk), dim3(256), smem, ctx->stream, (const float2*)ba.p, ne, (const float*)bt.p, nb, (float2*)bb.p);
  hipLaunchKernelGGL(cond_fir_kernel<true>, dim3(nblk), dim3(256), smem, ctx->stream, (const float2*)bb.p, ne, (const float*)bt.p, nb, (float2*)ba.p);
  hipLaunchKernelGGL(cond_decimate_kernel, dim3((unsigned int)std::min<long long>((len + 255) / 256, 65535)), dim3(256), 0, ctx->stream,
                     (const float2*)ba.p, nfact, fs, new_fs, len, (float2*)bsig.p);
  GC_HIP(hipGetLastError());
  GC_HIP(hipStreamSynchronize(ctx->stream));  // hf must outlive its copy
  ctx->acq_cond_n = len;
  out->sampling_freq = new_fs;
  out->intermediate_freq = std::fmod(IF, new_fs);  // rem(), :95
  out->n_samples = len;
  return GC_OK;
}

// The searches' other source (gc_acq_params.source = CONDITIONED) filled without the conditioning block: from the record in
// whatever format it has, or from the caller's own complex samples (acquisition(longSignal, settings) takes any complex row).
extern "C" int gc_acq_signal_from_record(gc_context* ctx, int64_t first_sample, int64_t n) {
  if (!ctx || first_sample < 0 || n <= 0) {
    gc_set_error("gc_acq_signal_from_record: bad arguments");
    return GC_E_INVALID;
  }
  if (!ctx->d_if) {
    gc_set_error("gc_acq_signal_from_record: no IF record loaded");
    return GC_E_STATE;
  }
  if ((uint64_t)first_sample + (uint64_t)n > ctx->if_nsamples) {
    gc_set_error("gc_acq_signal_from_record: %lld samples from %lld: outside the record", (long long)n, (long long)first_sample);
    return GC_E_RANGE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  GcBuf& bsig = ctx->acqbuf[gc_context::ACQ_COND_SIG];
  ctx->acq_cond_n = 0;
  if (gc_buf_reserve(bsig, (size_t)n * sizeof(float2), false) != hipSuccess) {
    gc_set_error("gc_acq_signal_from_record: device allocation failed");
    return GC_E_NOMEM;
  }
  hipLaunchKernelGGL(record_to_float_kernel, dim3((unsigned int)std::min<long long>((n + 255) / 256, 65535)), dim3(256), 0, ctx->stream,
                     (const void*)ctx->d_if, ctx->if_dtype, ctx->if_layout, (long long)first_sample, (long long)n, (float2*)bsig.p);
  GC_HIP(hipGetLastError());
  ctx->acq_cond_n = n;
  return GC_OK;
}

extern "C" int gc_acq_set_signal(gc_context* ctx, const float* iq, int64_t n) {
  if (!ctx || !iq || n <= 0) {
    gc_set_error("gc_acq_set_signal: bad arguments");
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  GcBuf& bsig = ctx->acqbuf[gc_context::ACQ_COND_SIG];
  ctx->acq_cond_n = 0;
  if (gc_buf_reserve(bsig, (size_t)n * sizeof(float2), false) != hipSuccess) {
    gc_set_error("gc_acq_set_signal: device allocation failed");
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemcpyAsync(bsig.p, iq, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  ctx->acq_cond_n = n;
  return GC_OK;
}

extern "C" int gc_acq_conditioned(gc_context* ctx, int64_t first, int64_t n, float* dst) {
  if (!ctx || !dst || first < 0 || n <= 0 || first + n > ctx->acq_cond_n) {
    gc_set_error("gc_acq_conditioned: range outside the conditioned signal");
    return GC_E_RANGE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipMemcpy(dst, (const float2*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p + first, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost));
  return GC_OK;
}

// Generic fine-frequency stage: per-code-period complex sums of signal x code x carrier for `nbins`
// carriers f0 - k*fstep over `ncodes` periods from first_sample; the hypothesis search over bit edges / Neuman-
// Hofman / secondary codes / data+pilot combinations is a few hundred flops and stays with the caller.
// `ndet` detections (code d*code_len.., first_sample[d], f0[d]) share one launch and one read-back.
// Queues the per-code-period sums of `ndet` detections on the context's stream and leaves them on the device (*dsums:
// double[ndet][nbins][ncodes][2]); nothing is synchronised: hdet (filled here) and `codes` must stay alive until the caller has.
static int fine_sums_enqueue(gc_context* ctx, const gc_fine_params* p, int ndet, const int8_t* codes, const int64_t* first_sample,
                             const double* f0, std::vector<FineDet>& hdet, const double** dsums) {
  if (!ctx || !p || ndet <= 0 || ndet > 65535 || !codes || !first_sample || !f0 || p->spc <= 0 || p->ncodes <= 0 ||
      p->nbins <= 0 || p->code_len <= 0) {
    gc_set_error("gc_acquire_fine_sums: bad arguments");
    return GC_E_INVALID;
  }
  const bool cond = p->source == GC_ACQ_SOURCE_CONDITIONED;
  if (cond) {
    if (ctx->acq_cond_n <= 0) {
      gc_set_error("gc_acquire_fine_sums: no conditioned signal (call gc_acq_condition first)");
      return GC_E_STATE;
    }
  } else if (!ctx->d_if || ctx->if_dtype != GC_I8 || ctx->if_layout != GC_IQ) {
    return ctx->d_if ? GC_E_UNSUPPORTED : GC_E_STATE;
  }
  const uint64_t avail = cond ? (uint64_t)ctx->acq_cond_n : ctx->if_nsamples;
  hdet.resize((size_t)ndet);
  for (int d = 0; d < ndet; ++d) {
    if (first_sample[d] < 0 || (uint64_t)first_sample[d] + (uint64_t)p->ncodes * p->spc > avail) {
      gc_set_error("gc_acquire_fine_sums: %d code periods from sample %lld exceed the IF buffer", p->ncodes, (long long)first_sample[d]);
      return first_sample[d] < 0 ? GC_E_INVALID : GC_E_RANGE;
    }
    hdet[d].first = first_sample[d];
    hdet[d].f0 = f0[d];
  }
  GC_HIP(hipSetDevice(ctx->device));
  const size_t nout = (size_t)ndet * p->nbins * p->ncodes * 2;
  GcBuf& bcode = ctx->acqbuf[gc_context::ACQ_FINE_CODE];
  GcBuf& bdet = ctx->acqbuf[gc_context::ACQ_FINE_DET];
  GcBuf& bout = ctx->acqbuf[gc_context::ACQ_FINE_OUT];
  if (gc_buf_reserve(bcode, (size_t)ndet * p->code_len, false) != hipSuccess ||
      gc_buf_reserve(bdet, (size_t)ndet * sizeof(FineDet), false) != hipSuccess ||
      gc_buf_reserve(bout, nout * sizeof(double) * (size_t)(kFineParts + 1), false) != hipSuccess) {
    gc_set_error("gc_acquire_fine_sums: device allocation failed");
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemcpyAsync(bcode.p, codes, (size_t)ndet * p->code_len, hipMemcpyHostToDevice, ctx->stream));
  GC_HIP(hipMemcpyAsync(bdet.p, hdet.data(), (size_t)ndet * sizeof(FineDet), hipMemcpyHostToDevice, ctx->stream));
  dim3 grid((unsigned int)p->ncodes, (unsigned int)ndet, (unsigned int)((p->nbins + kFineBins - 1) / kFineBins));
  int parts = 1;  // workgroups per code period: enough of them for four per CU, runs of at least 2 048 samples
  while (parts < kFineParts && (long long)grid.x * grid.y * grid.z * parts < 4LL * ctx->compute_units && p->spc / (2 * parts) >= 2048) parts *= 2;
  if (GC_TUNE_ENV("GC_ACQ_FINE_PARTS")) parts = std::max(1, std::min(kFineParts, std::atoi(GC_TUNE_ENV("GC_ACQ_FINE_PARTS"))));
  grid.x *= (unsigned int)parts;
  double* const dout = (double*)bout.p;
  double* const dpart = parts > 1 ? dout + nout : dout;  // [parts][nout] behind the result
  const double tc = p->code_freq > 0.0 ? 1.0 / p->code_freq : 0.0;  // 0: sampled replica, one entry per sample
  if (cond)
    hipLaunchKernelGGL(fine_multi_kernel<true>, grid, dim3(256), 0, ctx->stream, (const void*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p,
                       (const FineDet*)bdet.p, p->spc, p->ncodes, (const int8_t*)bcode.p, p->code_len, 1.0 / p->sampling_freq,
                       tc, p->fstep, p->sampling_freq, p->nbins, p->index_offset, (float)p->dc_re, (float)p->dc_im, dpart, parts, nout);
  else
    hipLaunchKernelGGL(fine_multi_kernel<false>, grid, dim3(256), 0, ctx->stream, (const void*)ctx->d_if, (const FineDet*)bdet.p, p->spc,
                       p->ncodes, (const int8_t*)bcode.p, p->code_len, 1.0 / p->sampling_freq, tc, p->fstep,
                       p->sampling_freq, p->nbins, p->index_offset, (float)p->dc_re, (float)p->dc_im, dpart, parts, nout);
  GC_HIP(hipGetLastError());
  if (parts > 1) {
    hipLaunchKernelGGL(fine_parts_kernel, dim3((unsigned int)((nout + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)dpart, parts, nout, dout);
    GC_HIP(hipGetLastError());
  }
  *dsums = dout;
  return GC_OK;
}

extern "C" int gc_acquire_fine_sums_batch(gc_context* ctx, const gc_fine_params* p, int ndet, const int8_t* codes,
                                          const int64_t* first_sample, const double* f0, double* out) {
  if (!out) {
    gc_set_error("gc_acquire_fine_sums: bad arguments");
    return GC_E_INVALID;
  }
  std::vector<FineDet> hdet;
  const double* dsums = nullptr;
  const int rc = fine_sums_enqueue(ctx, p, ndet, codes, first_sample, f0, hdet, &dsums);
  if (rc) {
    if (ctx) (void)hipStreamSynchronize(ctx->stream);  // (copies of hdet / codes may be queued)
    return rc;
  }
  const size_t nout = (size_t)ndet * p->nbins * p->ncodes * 2;
  hipError_t e = hipMemcpyAsync(out, dsums, nout * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);  // also keeps hdet / codes alive until the copies are done
  if (e == hipSuccess) e = e2;
  if (e != hipSuccess) {
    gc_set_error("gc_acquire_fine_sums: %s", hipGetErrorString(e));
    return GC_E_HIP;
  }
  return GC_OK;
}

extern "C" int gc_acq_signal_stats(gc_context* ctx, int64_t first_sample, int64_t n, int32_t source, double* mean_re, double* mean_im,
                                   double* var) {
  if (!ctx || first_sample < 0 || n < 2 || n > 0x7fffffff || !mean_re || !mean_im || !var) {
    gc_set_error("gc_acq_signal_stats: bad arguments");
    return GC_E_INVALID;
  }
  const bool cond = source == GC_ACQ_SOURCE_CONDITIONED;
  if (cond) {
    if (ctx->acq_cond_n <= 0) {
      gc_set_error("gc_acq_signal_stats: no conditioned signal (call gc_acq_condition first)");
      return GC_E_STATE;
    }
  } else if (!ctx->d_if || ctx->if_dtype != GC_I8 || ctx->if_layout != GC_IQ) {
    gc_set_error("gc_acq_signal_stats: needs an int8 I/Q IF buffer");
    return ctx->d_if ? GC_E_UNSUPPORTED : GC_E_STATE;
  }
  if ((uint64_t)first_sample + (uint64_t)n > (cond ? (uint64_t)ctx->acq_cond_n : ctx->if_nsamples)) {
    gc_set_error("gc_acq_signal_stats: %lld samples from %lld exceed the signal", (long long)n, (long long)first_sample);
    return GC_E_RANGE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  GcBuf& b = ctx->acqbuf[gc_context::ACQ_FINE_DET];
  if (gc_buf_reserve(b, 64, false) != hipSuccess) {
    gc_set_error("gc_acq_signal_stats: device allocation failed");
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemsetAsync(b.p, 0, 64, ctx->stream));
  if (cond)
    hipLaunchKernelGGL(sigpower_f32_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const float2*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p,
                       (long long)first_sample, (int)n, (double*)b.p);
  else
    hipLaunchKernelGGL(sigpower_kernel, dim3(64), dim3(256), 0, ctx->stream, (const int8_t*)ctx->d_if, (long long)first_sample, (int)n,
                       (long long*)b.p);
  GC_HIP(hipGetLastError());
  long long hs[3];
  GC_HIP(hipMemcpyAsync(hs, b.p, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  double s3[3];
  if (cond) std::memcpy(s3, hs, sizeof s3);
  else for (int k = 0; k < 3; ++k) s3[k] = (double)hs[k];
  const double mr = s3[0] / (double)n, mi = s3[1] / (double)n;
  *mean_re = mr;
  *mean_im = mi;
  *var = (s3[2] - (double)n * (mr * mr + mi * mi)) / (double)(n - 1);
  return GC_OK;
}

extern "C" int gc_acquire_fine_sums(gc_context* ctx, const gc_fine_params* p, const int8_t* code, double* out) {
  if (!p) {
    gc_set_error("gc_acquire_fine_sums: bad arguments");
    return GC_E_INVALID;
  }
  const int64_t first = p->first_sample;
  return gc_acquire_fine_sums_batch(ctx, p, 1, code, &first, &p->f0, out);
}

// ---- circshift search family ------------------------------------------------------------------------------
// GPS_L2C/include/acquisition.m:40-75, BDS/B1I/include/acquisition.m:76-123, BDS/B1C/include/acquisition.m:137-170:
// the signal block is mixed with a handful of carriers and transformed ONCE; Doppler bins are circular shifts of
// that spectrum before the product with the code spectrum and the inverse transform.
extern "C" int gc_acq_shift_prepare(gc_context* ctx, const gc_acq_shift_params* p) {
  if (!ctx || !p || p->n <= 0 || p->n_signals <= 0 || p->n_carriers <= 0 || p->n_bins <= 0 || p->first_sample < 0 ||
      p->n_arms_max < 1 || p->n_arms_max > 4) {
    gc_set_error("gc_acq_shift_prepare: bad arguments");
    return GC_E_INVALID;
  }
  const bool cond = p->source == GC_ACQ_SOURCE_CONDITIONED;
  if (cond) {
    if (ctx->acq_cond_n <= 0) {
      gc_set_error("gc_acq_shift_prepare: no conditioned signal (gc_acq_condition / gc_acq_signal_from_record / gc_acq_set_signal first)");
      return GC_E_STATE;
    }
  } else if (!ctx->d_if || ctx->if_dtype != GC_I8 || ctx->if_layout != GC_IQ) {
    gc_set_error("gc_acq_shift_prepare: needs an int8 I/Q IF buffer (other records: gc_acq_signal_from_record, then source = 1)");
    return ctx->d_if ? GC_E_UNSUPPORTED : GC_E_STATE;
  }
  const uint64_t avail = cond ? (uint64_t)ctx->acq_cond_n : ctx->if_nsamples;
  if ((uint64_t)p->first_sample + (uint64_t)p->n_signals * p->n > avail) {
    gc_set_error("gc_acq_shift_prepare: needs %lld samples from %lld, the signal holds %llu", (long long)p->n_signals * p->n,
                 (long long)p->first_sample, (unsigned long long)avail);
    return GC_E_RANGE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  const int rows = p->n_carriers * p->n_signals * p->n_bins;
  // A block length the radix plan cannot take (16.368-Msps front ends: 2*16 368*... has the factors 11 and 31): the shifted
  // product needs a transform of exactly n points, so every row gets its own carrier instead - circshift(X, b) is the
  // carrier moved down by b*fs/n - and the n-point circular correlation is read off a transform of M >= 2n points fed
  // with the block twice and zeros (for a replica that ends inside the block the first n lags are the same sums).
  int m = p->n;
  bool padded = false;
  {
    Plan probe;
    if (!make_plan(m, &probe) || GC_TUNE_ENV("GC_ACQ_PAD")) {
      padded = true;
      m = 0;
      for (int c = 2 * p->n; c < 2 * p->n + (1 << 20); ++c)
        if (make_plan(c, &probe)) {
          m = c;
          break;
        }
      if (m == 0) {
        gc_set_error("gc_acq_shift_prepare: no transform size at or above %d fits the plan", 2 * p->n);
        return GC_E_UNSUPPORTED;
      }
    }
  }
  AcqScratch* s = nullptr;
  int rc = ensure_scratch(ctx, m, rows, p->n_arms_max, rows, p->n, &s);
  if (rc) return rc;
  if (s->shift_rows < rows) {
    if (s->rowmax) (void)hipFree(s->rowmax);
    if (s->rowarg) (void)hipFree(s->rowarg);
    s->rowmax = nullptr;
    s->rowarg = nullptr;
    s->shift_rows = 0;
    if (hipMalloc((void**)&s->rowmax, sizeof(float) * rows) != hipSuccess || hipMalloc((void**)&s->rowarg, sizeof(int) * rows) != hipSuccess) {
      gc_set_error("gc_acq_shift_prepare: device allocation failed");
      return GC_E_NOMEM;
    }
    s->shift_rows = rows;
  }
  PassArgs base;
  std::memset(&base, 0, sizeof base);
  base.if_base = (const int8_t*)ctx->d_if;
  base.if_f32 = cond ? (const float2*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p : nullptr;
  base.first_sample = p->first_sample;
  base.spc = p->n;              // signal k starts k*n samples later; the carrier phase restarts with every block
  base.nhops = p->n_signals;
  base.f0 = p->carrier_f0;
  base.fstep = -p->carrier_step;  // kernel: f_b = f0 - fstep*b
  base.fs = p->sampling_freq;
  s->shift.n = 0;
  if (!padded) {
    rc = forward(ctx, s, base, PRE_IF_CARRIER, (long long)p->n_carriers * p->n_signals, s->sig);
    if (rc) return rc;
  } else {
    // internal row order: ((carrier * n_bins + bin) * n_signals + signal)
    base.wrap_len = p->n;
    base.fstep = p->sampling_freq / (double)p->n;  // one position of circshift
    for (int i = 0; i < p->n_carriers; ++i) {
      base.f0 = p->carrier_f0 + p->carrier_step * i;
      rc = forward(ctx, s, base, PRE_IF_CARRIER, (long long)p->n_bins * p->n_signals,
                   s->sig + (size_t)i * p->n_bins * p->n_signals * (size_t)m);
      if (rc) return rc;
    }
  }
  s->shift = *p;
  s->shift_padded = padded;
  return GC_OK;
}

// public row ((carrier * n_signals + signal) * n_bins + bin) -> row of the padded mode's internal order
static int shift_internal_row(const gc_acq_shift_params& p, int row) {
  const int bin = row % p.n_bins, cs = row / p.n_bins, signal = cs % p.n_signals, carrier = cs / p.n_signals;
  return (carrier * p.n_bins + bin) * p.n_signals + signal;
}

// device -> caller through the scratch's pinned buffer (grown on demand); GC_ACQ_SHIFT_PAGEABLE=1 or no pinned memory: straight into the
// caller's array.  Synchronises the stream.
static int shift_read_back(gc_context* ctx, AcqScratch* s, void* dst0, const void* src0, size_t bytes0, void* dst1 = nullptr, const void* src1 = nullptr,
                           size_t bytes1 = 0) {
  const size_t total = bytes0 + bytes1;
  if (!GC_TUNE_ENV("GC_ACQ_SHIFT_PAGEABLE")) {
    if (s->pinned_bytes < total) {
      if (s->pinned) (void)hipHostFree(s->pinned);
      s->pinned = nullptr;
      s->pinned_bytes = 0;
      const size_t want = std::max(total, (size_t)1 << 21);
      if (hipHostMalloc(&s->pinned, want, hipHostMallocDefault) == hipSuccess) s->pinned_bytes = want;
      else (void)hipGetLastError();
    }
  }
  if (s->pinned_bytes >= total && !GC_TUNE_ENV("GC_ACQ_SHIFT_PAGEABLE")) {
    char* h = static_cast<char*>(s->pinned);
    GC_HIP(hipMemcpyAsync(h, src0, bytes0, hipMemcpyDeviceToHost, ctx->stream));
    if (bytes1) GC_HIP(hipMemcpyAsync(h + bytes0, src1, bytes1, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(dst0, h, bytes0);
    if (bytes1) std::memcpy(dst1, h + bytes0, bytes1);
    return GC_OK;
  }
  GC_HIP(hipMemcpyAsync(dst0, src0, bytes0, hipMemcpyDeviceToHost, ctx->stream));
  if (bytes1) GC_HIP(hipMemcpyAsync(dst1, src1, bytes1, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

// The inverse side of ONE PRN of a circshift search: rows pass (shifted product with the PRN's code spectra `codespec`, narms x N) and
// columns pass for every chunk of rows, the row maxima into s->rowmax / s->rowarg (specialised passes: *all_fused, the sums
// themselves are not written; otherwise they are in s->results and the caller runs rowmax_kernel).
static int shift_search_passes(gc_context* ctx, AcqScratch* s, int narms, const float2* codespec, const double* arm_weight, bool* all_fused_out,
                               float2* tmpbuf) {
  const gc_acq_shift_params& p = s->shift;
  const Plan& pl = s->plan;
  const int rows = p.n_carriers * p.n_signals * p.n_bins;
  PassArgs base;
  std::memset(&base, 0, sizeof base);
  base.spc = p.n;
  base.nhops = 1;
  int rc = GC_OK;
  // Rows in chunks (specialised passes): the rows pass writes rows x N x 8 bytes that the columns pass reads back - 579 MB per PRN and
  // arm for BDS B1C, 2 GB for GPS L2C, through HBM both ways.  A chunk of rows whose intermediate is <= GC_ACQ_SHIFT_CHUNK_MB goes
  // through both passes (and both arms) before the next one starts, in the same place: the columns pass finds it in the 256 MB
  // last-level cache.  Measured: B1C (600 x 600 plan) 103.9 -> 99.3 ms at 160 MB (100.7 at 96, 113.8 at 48); L2C (512 x 625) 75.3 -> 77.8 /
  // 82.3 / 92.9 ms - its 802 rows of 125 narrow tiles lose more to the additional launches than the cache gives back: chunks for the
  // 600 x 600 plan only (0: all rows at once).
  int chunk_rows = rows;
  if (ct_columns_tile(pl.p1.len, pl.n2) > 0 && !GC_TUNE_ENV("GC_ACQ_GENERIC")) {
    double mb = (pl.n1 == 600 && pl.n2 == 600) ? 160.0 : 0.0;
    if (const char* e = GC_TUNE_ENV("GC_ACQ_SHIFT_CHUNK_MB")) mb = std::atof(e);
    if (mb > 0.0) chunk_rows = std::max(8, std::min(rows, (int)(mb * 1024.0 * 1024.0 / ((double)pl.n * sizeof(float2)))));
  }
  // Both arms of a chunk in one launch pair (PassArgs::arm_batches, as the coarse search does for Galileo E1): a row has ONE transform per
  // arm, so the columns pass walks the arms like hops, weighting each (PassArgs::arm_w).  GC_ACQ_ARMS_SEPARATE=1: arm by arm.
  const bool merge_arms = narms > 1 && ct_columns_tile(pl.p1.len, pl.n2) > 0 && !GC_TUNE_ENV("GC_ACQ_GENERIC") && !GC_TUNE_ENV("GC_ACQ_ARMS_SEPARATE") &&
                          !GC_TUNE_ENV("GC_ACQ_ROWMAX_KERNEL");
  const int marms = merge_arms ? narms : 1;
  if (merge_arms) chunk_rows = std::max(1, std::min(chunk_rows, rows / narms));  // (the intermediate holds `rows` transforms)
  bool all_fused = true;
  for (int r0 = 0; r0 < rows; r0 += chunk_rows)
  for (int arm = 0; arm < (merge_arms ? 1 : narms); ++arm) {  // (separate arms of a chunk after one another: the second one adds to sums the first one just wrote)
    const int rc_rows = std::min(chunk_rows, rows - r0);
    float2* const tmp = tmpbuf - (size_t)r0 * marms * (size_t)pl.n;  // (a chunk's batches keep their numbers; its first one sits at the start of the buffer)
    PassArgs a = base;
    a.n = pl.n;
    a.tw = s->tw;
    a.inverse = 1;
    fill_sub(a, pl.p2);
    a.nvec = pl.n1;
    a.estride = 1;
    a.vstride = pl.n2;
    a.cols = choose_cols(a.len, a.estride);
    a.pre = PRE_MUL_CONJ;
    a.post = POST_TWIDDLE;
    a.in = s->sig;
    a.in_batch_stride = pl.n;
    a.other = codespec + (size_t)arm * pl.n;
    a.out = tmp;
    a.out_batch_stride = pl.n;
    a.shift_bins = s->shift_padded ? 0 : p.n_bins;  // padded: every row is a spectrum of its own
    a.n1 = pl.n1;
    a.n2 = pl.n2;
    a.batch0 = r0;
    a.arm_batches = merge_arms ? rc_rows : 0;
    a.narms_merged = marms;
    rc = launch_pass(ctx, a, (long long)marms * rc_rows);
    a.batch0 = 0;
    a.arm_batches = 0;
    if (rc) return rc;
    if (merge_arms) {
      a.nhops = narms;  // the columns pass adds a row's arms like hops
      a.arm_hops = 1;
      for (int k = 0; k < 4; ++k) a.arm_w[k] = (arm_weight && k < narms) ? (float)arm_weight[k] : 1.0f;
    }
    fill_sub(a, pl.p1);
    a.nvec = pl.n2;
    a.estride = pl.n2;
    a.vstride = 1;
    a.cols = choose_cols(a.len, a.estride);
    a.pre = PRE_NONE;
    a.shift_bins = 0;
    a.post = POST_ABS_ACC;
    a.in = tmp;
    a.acc_out = s->results;
    a.acc_add = arm > 0;
    a.acc_scale = merge_arms ? 1.0f : arm_weight ? (float)arm_weight[arm] : 1.0f;
    bool fused_rows = false;
    rc = launch_abs_pass(ctx, s, a, rc_rows, nullptr, p.n, 0, 1, (merge_arms || arm == narms - 1) ? &fused_rows : nullptr, r0, rows);
    if (rc) return rc;
    if (merge_arms || arm == narms - 1) all_fused = all_fused && fused_rows;
  }
  *all_fused_out = all_fused;
  return GC_OK;
}

extern "C" int gc_acq_shift_search(gc_context* ctx, int narms, const int8_t* codes, const double* arm_weight,
                                   float* row_max, int32_t* row_argmax) {
  AcqScratch* s = ctx ? (AcqScratch*)ctx->acq_scratch : nullptr;
  if (!s || s->shift.n <= 0 || (s->shift_padded ? s->n < 2 * s->shift.n : s->shift.n != s->n)) {
    gc_set_error("gc_acq_shift_search: call gc_acq_shift_prepare first");
    return GC_E_STATE;
  }
  const gc_acq_shift_params& p = s->shift;
  if (narms < 1 || narms > p.n_arms_max || !codes || !row_max || !row_argmax) {
    gc_set_error("gc_acq_shift_search: bad arguments");
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  const Plan& pl = s->plan;
  const int rows = p.n_carriers * p.n_signals * p.n_bins;
  GC_HIP(hipMemcpyAsync(s->codes, codes, (size_t)narms * p.n, hipMemcpyHostToDevice, ctx->stream));
  PassArgs base;
  std::memset(&base, 0, sizeof base);
  base.spc = p.n;
  base.nhops = 1;
  base.codes = s->codes;
  int rc = forward(ctx, s, base, PRE_CODE, narms, s->codespec);
  if (rc) return rc;
  bool all_fused = true;
  rc = shift_search_passes(ctx, s, narms, s->codespec, arm_weight, &all_fused, s->tmp);
  if (rc) return rc;
  s->shift_rows_fused = all_fused;
  s->shift_narms = narms;
  for (int arm = 0; arm < 4; ++arm) s->shift_weight[arm] = (arm_weight && arm < narms) ? arm_weight[arm] : 1.0;
  if (!s->shift_rows_fused) {
    hipLaunchKernelGGL(rowmax_kernel, dim3(rows), dim3(256), 0, ctx->stream, s->results, p.n, pl.n, s->rowmax, s->rowarg);
    GC_HIP(hipGetLastError());
  }
  if (!s->shift_padded) {
    return shift_read_back(ctx, s, row_max, s->rowmax, sizeof(float) * rows, row_argmax, s->rowarg, sizeof(int) * rows);
  }
  std::vector<float> hv((size_t)rows);
  std::vector<int> ha((size_t)rows);
  GC_HIP(hipMemcpyAsync(hv.data(), s->rowmax, sizeof(float) * rows, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipMemcpyAsync(ha.data(), s->rowarg, sizeof(int) * rows, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < rows; ++r) {
    row_max[r] = hv[(size_t)shift_internal_row(p, r)];
    row_argmax[r] = ha[(size_t)shift_internal_row(p, r)];
  }
  return GC_OK;
}

extern "C" int gc_acq_shift_dims(gc_context* ctx, int32_t* n, int32_t* rows, int32_t* n_arms_max) {
  AcqScratch* s = ctx ? (AcqScratch*)ctx->acq_scratch : nullptr;
  if (!s || s->shift.n <= 0) {
    gc_set_error("gc_acq_shift_dims: call gc_acq_shift_prepare first");
    return GC_E_STATE;
  }
  if (n) *n = s->shift.n;
  if (rows) *rows = s->shift.n_carriers * s->shift.n_signals * s->shift.n_bins;
  if (n_arms_max) *n_arms_max = s->shift.n_arms_max;
  return GC_OK;
}

// Row `irow` (internal order) of a circshift search transformed again: one batch per pass, every arm of `codespec` (narms x N) with its
// weight; the row's n sums land at acc_out + irow * N, or with to_slot at acc_out itself (the specialised passes only: launch_pass
// refuses otherwise).
static int shift_row_passes(gc_context* ctx, AcqScratch* s, int irow, int narms, const float2* codespec, const double* weight, float* acc_out,
                            bool to_slot = false) {
  const gc_acq_shift_params& p = s->shift;
  const Plan& pl = s->plan;
  PassArgs base;
  std::memset(&base, 0, sizeof base);
  base.spc = p.n;
  base.nhops = 1;
  for (int arm = 0; arm < narms; ++arm) {
    PassArgs a = base;
    a.n = pl.n;
    a.tw = s->tw;
    a.inverse = 1;
    fill_sub(a, pl.p2);
    a.nvec = pl.n1;
    a.estride = 1;
    a.vstride = pl.n2;
    a.cols = choose_cols(a.len, a.estride);
    a.pre = PRE_MUL_CONJ;
    a.post = POST_TWIDDLE;
    a.in = s->sig;
    a.in_batch_stride = pl.n;
    a.other = codespec + (size_t)arm * pl.n;
    a.out = s->tmp;
    a.out_batch_stride = pl.n;
    a.shift_bins = s->shift_padded ? 0 : p.n_bins;
    a.n1 = pl.n1;
    a.n2 = pl.n2;
    a.batch0 = irow;
    int rc = launch_pass(ctx, a, 1);
    if (rc) return rc;
    fill_sub(a, pl.p1);
    a.nvec = pl.n2;
    a.estride = pl.n2;
    a.vstride = 1;
    a.cols = choose_cols(a.len, a.estride);
    a.pre = PRE_NONE;
    a.shift_bins = 0;
    a.post = POST_ABS_ACC;
    a.in = s->tmp;
    a.acc_out = acc_out;
    a.acc_row0 = to_slot ? irow : 0;
    a.acc_add = arm > 0;
    a.acc_scale = (float)weight[arm];
    a.hop_groups = 1;
    rc = launch_pass(ctx, a, 1);
    if (rc) return rc;
  }
  return GC_OK;
}

extern "C" int gc_acq_shift_row(gc_context* ctx, int row, float* out) {
  AcqScratch* s = ctx ? (AcqScratch*)ctx->acq_scratch : nullptr;
  if (!s || s->shift.n <= 0 || !out || row < 0 || row >= s->shift.n_carriers * s->shift.n_signals * s->shift.n_bins) {
    gc_set_error("gc_acq_shift_row: bad arguments or nothing searched yet");
    return GC_E_INVALID;
  }
  if (s->shift_rows_fused && s->shift_narms < 1) {
    gc_set_error("gc_acq_shift_row: the last search was gc_acq_shift_search_batch (it returns each PRN's pick itself); search one PRN with gc_acq_shift_search first");
    return GC_E_STATE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  const int irow = s->shift_padded ? shift_internal_row(s->shift, row) : row;
  const size_t at = (size_t)irow * (size_t)s->n;
  if (s->shift_rows_fused) {
    // the search kept only the row maxima: this row's inverse transforms again (the code spectra of the search are still in place)
    int rc = shift_row_passes(ctx, s, irow, s->shift_narms, s->codespec, s->shift_weight, s->results);
    if (rc) return rc;
  }
  return shift_read_back(ctx, s, out, s->results + at, sizeof(float) * s->shift.n);
}

// ---- the whole search of a package in one call -------------------------------------------------------------------------------
namespace {
// What shift_pick_kernel leaves per PRN (device-internal; the public gc_acq_shift_pick is filled from it and from the float64 guard)
struct ShiftPickDev {
  int row;          // in: the winning row (-1: none)
  int code_phase;   // 0-based first maximum of the row
  int second_col;   // 0-based first position of the second peak (-1: no second peak asked for or range empty)
  int near_peak;    // cells of the row at or above peak * (1 - eps), the maximum itself included
  int near_second;  // cells of the second-peak range at or above second * (1 - eps)
  float peak, second;
  int pad_;
};

// One workgroup per PRN: the first maximum of the winning row (BDS/B1I acquisition.m:126, GPS_L2C :72) and the largest value of the
// row's first `period` samples outside +-exclude samples of it - the reference's three range cases (B1I :141-156, L2C :77-91;
// 1-based there: e1 = codePhase - exclude, e2 = codePhase + exclude; e1 < 2: e2 .. period + e1; e2 >= period: e2 - period + 1 .. e1;
// else 1 .. e1 and e2 .. period).  period <= 0: no second peak (GC_SHIFT_PICK_GLOBAL).  For the float64 guard: how many cells lie
// within eps (relative) of either value - more than one means the float32 ordering decided something it cannot.
__global__ __launch_bounds__(1024) void shift_pick_kernel(const float* __restrict__ rows, long long row_stride, int n, int exclude, int period, float eps,
                                                         ShiftPickDev* __restrict__ picks) {
  __shared__ float sv[1024];
  __shared__ int si[1024];
  ShiftPickDev& pk = picks[blockIdx.x];
  if (pk.row < 0) return;
  const float* __restrict__ r = rows + (size_t)blockIdx.x * (size_t)row_stride;
  float best = -1.0f;
  int bi = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float v = r[i];
    if (v > best) {
      best = v;
      bi = i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float v = sv[threadIdx.x + off];
      const int i = si[threadIdx.x + off];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) {
        sv[threadIdx.x] = v;
        si[threadIdx.x] = i;
      }
    }
    __syncthreads();
  }
  const float peak = sv[0];
  const int cp = si[0] + 1;  // 1-based, as the reference's ranges
  __syncthreads();
  const int e1 = cp - exclude, e2 = cp + exclude;
  int lo0 = 1, hi0 = 0, lo1 = 1, hi1 = 0;  // 1-based inclusive ranges
  if (period > 0) {
    if (e1 < 2) {
      lo0 = e2;
      hi0 = period + e1;
    } else if (e2 >= period) {
      lo0 = e2 - period + 1;
      hi0 = e1;
    } else {
      lo0 = 1;
      hi0 = e1;
      lo1 = e2;
      hi1 = period;
    }
  }
  float second = -1.0f;
  int sc = 0x7fffffff;
  auto see = [&](int i) {
    const float v = r[i];
    if (v > second || (v == second && i < sc)) {
      second = v;
      sc = i;
    }
  };
  for (int i = lo0 - 1 + (int)threadIdx.x; i < hi0 && i < n; i += 1024)
    if (i >= 0) see(i);
  for (int i = lo1 - 1 + (int)threadIdx.x; i < hi1 && i < n; i += 1024)
    if (i >= 0) see(i);
  sv[threadIdx.x] = second;
  si[threadIdx.x] = sc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float v = sv[threadIdx.x + off];
      const int i = si[threadIdx.x + off];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) {
        sv[threadIdx.x] = v;
        si[threadIdx.x] = i;
      }
    }
    __syncthreads();
  }
  const float sec = sv[0];
  const int sec_col = si[0];
  __syncthreads();
  // the guard's counts
  const float tp = peak * (1.0f - eps), ts2 = sec * (1.0f - eps);
  int np = 0, ns = 0;
  for (int i = threadIdx.x; i < n; i += 1024) np += r[i] >= tp ? 1 : 0;
  if (sec >= 0.0f) {
    for (int i = lo0 - 1 + (int)threadIdx.x; i < hi0 && i < n; i += 1024)
      if (i >= 0) ns += r[i] >= ts2 ? 1 : 0;
    for (int i = lo1 - 1 + (int)threadIdx.x; i < hi1 && i < n; i += 1024)
      if (i >= 0) ns += r[i] >= ts2 ? 1 : 0;
  }
  si[threadIdx.x] = np;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) si[threadIdx.x] += si[threadIdx.x + off];
    __syncthreads();
  }
  np = si[0];
  __syncthreads();
  si[threadIdx.x] = ns;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) si[threadIdx.x] += si[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pk.code_phase = cp - 1;
    pk.peak = peak;
    pk.second = sec >= 0.0f ? sec : 0.0f;
    pk.second_col = sec >= 0.0f ? sec_col : -1;
    pk.near_peak = np;
    pk.near_second = si[0];
  }
}

// Local replicas on the device: out[c][k] = chips[c][index[k]] for k < n_index, 0 up to n - the package's make*Table.m gather
// (code(ceil(ts * k / tc)), an index vector that depends on the rates only) and its zero padding ([table zeros], B1I :86, L2C :44,
// B1C :155-156) without the host forming or sending n bytes per code.
__global__ __launch_bounds__(256) void shift_expand_codes_kernel(const int8_t* __restrict__ chips, int chip_len, const int* __restrict__ index, int n_index, int n,
                                                                 int8_t* __restrict__ out) {
  const int8_t* __restrict__ c = chips + (size_t)blockIdx.y * chip_len;
  int8_t* __restrict__ o = out + (size_t)blockIdx.y * n;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) o[k] = k < n_index ? c[index[k]] : (int8_t)0;
}

// The reference's sequential selection over the (carrier, bin) grid (BDS/B1I acquisition.m:87-122, GPS_L2C :46-66): a value is taken
// only if it EXCEEDS the largest so far, starting from 0, and the last bin of every carrier but the first is not looked at - the
// first position, in scan order, of the largest value, if that is above 0.  v(carrier, bin) = rowmax, or the larger of the two signal
// blocks' (pairs).  Returns the public row index, or -1.
template <class T>
int pick_sequential(const gc_acq_shift_params& p, const T* rmax, bool pairs) {
  T best = 0;
  int row = -1;
  for (int c = 0; c < p.n_carriers; ++c)
    for (int b = 0; b < p.n_bins; ++b) {
      if (c > 0 && b == p.n_bins - 1) continue;
      if (!pairs) {
        const int r = c * p.n_bins + b;
        if (rmax[r] > best) {
          best = rmax[r];
          row = r;
        }
      } else {
        const int r1 = (c * 2 + 0) * p.n_bins + b, r2 = (c * 2 + 1) * p.n_bins + b;
        const T v = std::max(rmax[r1], rmax[r2]);
        if (v > best) {
          best = v;
          row = rmax[r1] > rmax[r2] ? r1 : r2;
        }
      }
    }
  return row;
}
}  // namespace

extern "C" int gc_acq_shift_search_batch(gc_context* ctx, int nprn, int narms, const int8_t* codes, int code_len, const int32_t* sample_index,
                                         int n_index, const double* arm_weight, int rule, int exclude, int period, gc_acq_shift_pick* out) {
  AcqScratch* s = ctx ? (AcqScratch*)ctx->acq_scratch : nullptr;
  if (!s || s->shift.n <= 0 || (s->shift_padded ? s->n < 2 * s->shift.n : s->shift.n != s->n)) {
    gc_set_error("gc_acq_shift_search_batch: call gc_acq_shift_prepare first");
    return GC_E_STATE;
  }
  const gc_acq_shift_params& p = s->shift;
  const bool pairs = rule == GC_SHIFT_PICK_SEQUENTIAL_PAIRS;
  if (nprn < 1 || narms < 1 || narms > p.n_arms_max || !codes || !out || rule < GC_SHIFT_PICK_GLOBAL || rule > GC_SHIFT_PICK_SEQUENTIAL_PAIRS ||
      (pairs && p.n_signals != 2) || (rule == GC_SHIFT_PICK_SEQUENTIAL && p.n_signals != 1) ||
      (rule != GC_SHIFT_PICK_GLOBAL && (exclude < 0 || period < 1 || period > p.n)) ||
      (sample_index && (code_len < 1 || n_index < 1 || n_index > p.n))) {
    gc_set_error("gc_acq_shift_search_batch: bad arguments");
    return GC_E_INVALID;
  }
  if (sample_index)
    for (int k = 0; k < n_index; ++k)
      if (sample_index[k] < 0 || sample_index[k] >= code_len) {
        gc_set_error("gc_acq_shift_search_batch: sample_index[%d] = %d is outside the %d chips of a code", k, (int)sample_index[k], code_len);
        return GC_E_INVALID;
      }
  if (s->shift_padded || ct_columns_tile(s->plan.p1.len, s->plan.n2) == 0 || GC_TUNE_ENV("GC_ACQ_GENERIC") || GC_TUNE_ENV("GC_ACQ_ROWMAX_KERNEL")) {
    gc_set_error("gc_acq_shift_search_batch: this block length has no specialised pass kernels - search PRN by PRN (gc_acq_shift_search / _row)");
    return GC_E_UNSUPPORTED;
  }
  GC_HIP(hipSetDevice(ctx->device));
  const Plan& pl = s->plan;
  const int rows = p.n_carriers * p.n_signals * p.n_bins;
  const size_t N = (size_t)pl.n;
  const bool second = rule != GC_SHIFT_PICK_GLOBAL;
  if (gc_buf_reserve(s->b_codes, (size_t)nprn * narms * p.n, false) != hipSuccess ||
      gc_buf_reserve(s->b_codespec, (size_t)nprn * narms * N * sizeof(float2), false) != hipSuccess ||
      gc_buf_reserve(s->b_rowmax, (size_t)nprn * rows * sizeof(float), false) != hipSuccess ||
      gc_buf_reserve(s->b_rowarg, (size_t)nprn * rows * sizeof(int), false) != hipSuccess ||
      gc_buf_reserve(s->b_pick, (size_t)nprn * sizeof(ShiftPickDev), false) != hipSuccess ||
      gc_buf_reserve(s->b_rows, (size_t)nprn * N * sizeof(float), false) != hipSuccess) {
    (void)hipGetLastError();
    gc_set_error("gc_acq_shift_search_batch: device allocation failed");
    return GC_E_NOMEM;
  }
  // every PRN's codes up in one copy - sampled replicas of n entries, or chip tables and the index vector that samples them all
  // (expanded here) -, their spectra in as few forward launches as the intermediate buffer allows
  if (!sample_index) {
    GC_HIP(hipMemcpyAsync(s->b_codes.p, codes, (size_t)nprn * narms * p.n, hipMemcpyHostToDevice, ctx->stream));
  } else {
    const size_t chips_bytes = (size_t)nprn * narms * code_len, idx_off = (chips_bytes + 15) / 16 * 16;
    if (gc_buf_reserve(s->b_chips, idx_off + (size_t)n_index * sizeof(int), false) != hipSuccess) {
      (void)hipGetLastError();
      gc_set_error("gc_acq_shift_search_batch: device allocation failed");
      return GC_E_NOMEM;
    }
    GC_HIP(hipMemcpyAsync(s->b_chips.p, codes, chips_bytes, hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipMemcpyAsync((char*)s->b_chips.p + idx_off, sample_index, (size_t)n_index * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(shift_expand_codes_kernel, dim3((unsigned int)std::min(64, (p.n + 255) / 256), (unsigned int)(nprn * narms)), dim3(256), 0, ctx->stream,
                       (const int8_t*)s->b_chips.p, code_len, (const int*)((char*)s->b_chips.p + idx_off), n_index, p.n, (int8_t*)s->b_codes.p);
    GC_HIP(hipGetLastError());
  }
  float2* const cspec = (float2*)s->b_codespec.p;
  {
    PassArgs base;
    std::memset(&base, 0, sizeof base);
    base.spc = p.n;
    base.nhops = 1;
    const long long total = (long long)nprn * narms, step = std::max<long long>(1, s->nbh);
    for (long long k0 = 0; k0 < total; k0 += step) {
      base.codes = (const int8_t*)s->b_codes.p + (size_t)k0 * p.n;
      int rc = forward(ctx, s, base, PRE_CODE, std::min(step, total - k0), cspec + (size_t)k0 * N);
      if (rc) return rc;
    }
  }
  // phase 1: every PRN's rows and columns passes, its row maxima into its own slot - nothing comes back in between.  Two lanes where a
  // PRN's intermediate is small (BDS B1I: 62 PRNs x 0.17 ms of launches that each leave a tail of half-empty CUs - 5.8 -> 5.1 ms): even
  // PRNs on one stream of the device's search pair, odd PRNs on the other with an intermediate buffer and candidate slots of their own.
  // With gigabyte intermediates the two lanes only share the memory system they both wait for (GPS L2C 65.0 -> 66.2 ms, BDS B1C
  // 65.1 -> 64.1): one lane there.  GC_ACQ_SHIFT_LANES=1 / 2 overrides.
  int lanes = (nprn > 1 && (size_t)rows * N * sizeof(float2) <= ((size_t)256 << 20)) ? 2 : 1;
  if (const char* e = GC_TUNE_ENV("GC_ACQ_SHIFT_LANES")) lanes = std::max(1, std::min(2, std::atoi(e)));
  if (nprn < 2) lanes = 1;
  AcqStreams* const shared = lanes == 2 ? acq_streams(ctx->device) : nullptr;
  if (!shared) lanes = 1;
  if (lanes == 2 && !lane_events(s)) lanes = 1;
  if (lanes == 2 && !s->tmp2 && hipMalloc((void**)&s->tmp2, (size_t)s->nbh * N * sizeof(float2)) != hipSuccess) {
    (void)hipGetLastError();
    s->tmp2 = nullptr;
    lanes = 1;  // no room for a second intermediate
  }
  hipStream_t const stream1 = ctx->stream;
  hipStream_t lane_stream[2] = {stream1, stream1};
  if (lanes == 2) {
    lane_stream[0] = shared->main;
    lane_stream[1] = shared->lane;
    GC_HIP(hipEventRecord(s->ev_fork, stream1));  // signal spectra (gc_acq_shift_prepare) and code spectra are ready
    for (hipStream_t ls : lane_stream) GC_HIP(hipStreamWaitEvent(ls, s->ev_fork, 0));
  }
  float* const save_max = s->rowmax;
  int* const save_arg = s->rowarg;
  int rc = GC_OK;
  bool fused = true;
  s->shift_slot_lanes = lanes;
  for (int k = 0; k < nprn && rc == GC_OK && fused; ++k) {
    s->lane = lanes == 2 ? (k & 1) : 0;
    ctx->stream = lane_stream[s->lane];
    s->rowmax = (float*)s->b_rowmax.p + (size_t)k * rows;
    s->rowarg = (int*)s->b_rowarg.p + (size_t)k * rows;
    rc = shift_search_passes(ctx, s, narms, cspec + (size_t)k * narms * N, arm_weight, &fused, s->lane ? s->tmp2 : s->tmp);
  }
  ctx->stream = stream1;
  s->lane = 0;
  s->shift_slot_lanes = 1;
  s->rowmax = save_max;
  s->rowarg = save_arg;
  if (lanes == 2) {  // the lanes join the caller's stream (also on an error: nothing may still run on them)
    hipEvent_t const ej[2] = {s->ev_join, s->ev_join2};
    for (int k = 0; k < 2; ++k) {
      (void)hipEventRecord(ej[k], lane_stream[k]);
      (void)hipStreamWaitEvent(stream1, ej[k], 0);
    }
  }
  if (rc) {
    (void)hipDeviceSynchronize();
    return rc;
  }
  if (!fused) {
    gc_set_error("gc_acq_shift_search_batch: the passes did not run on the specialised kernels - search PRN by PRN");
    return GC_E_UNSUPPORTED;
  }
  // no single PRN's search is "the last one" after this call: gc_acq_shift_row has nothing to take a row from until the next
  // gc_acq_shift_search (its code spectra are not the ones in place)
  s->shift_rows_fused = true;
  s->shift_narms = 0;
  std::vector<float> hmax((size_t)nprn * rows);
  std::vector<int> harg((size_t)nprn * rows);
  rc = shift_read_back(ctx, s, hmax.data(), s->b_rowmax.p, sizeof(float) * hmax.size(), harg.data(), s->b_rowarg.p, sizeof(int) * harg.size());
  if (rc) return rc;
  // ---- the float64 guard (acq_guard.h) --------------------------------------------------------------------------------------------
  // Row maxima, first maxima and second peaks come out of float32 transforms; the reference's sequential `>` tests (B1I :98-119,
  // L2C :46-66), `[~, codePhase] = max(corr)` and `max_peak / second > threshold` (B1I :126-166) are float64.  Wherever two candidates
  // are closer than eps the cells that close are evaluated again as float64 correlations at one lag, and peak / second_peak of every
  // PRN always are (the two numbers the caller divides and thresholds).
  const bool guard = GC_TUNE_ENV("GC_ACQ_NO_GUARD") == nullptr;
  const double eps = gc_acq_tie_eps(pl.n);
  const double ones[4] = {1.0, 1.0, 1.0, 1.0};
  const double* const wts = arm_weight ? arm_weight : ones;
  GcExactSetup ex;
  ex.if_i8 = p.source == GC_ACQ_SOURCE_CONDITIONED ? nullptr : (const int8_t*)ctx->d_if;
  ex.if_f32 = p.source == GC_ACQ_SOURCE_CONDITIONED ? (const float2*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p : nullptr;
  ex.blk = p.n;
  ex.cl = sample_index ? n_index : p.n;  // (replica entries beyond the index vector are the zero padding)
  ex.hop_stride = 0;
  ex.nhops = 1;
  ex.narms = narms;
  for (int arm = 0; arm < narms; ++arm) ex.w[arm] = wts[arm];
  ex.codes = (const int8_t*)s->b_codes.p;
  ex.code_stride = p.n;
  ex.fs = p.sampling_freq;
  auto cell_of = [&](int k, int row, int col) {
    GcExactCell c;
    const int carrier = row / (p.n_signals * p.n_bins), sig = (row / p.n_bins) % p.n_signals, bin = row % p.n_bins;
    c.code = k;
    c.col = col;
    c.shift = bin;                                            // circshift(IQfreqDom, bin): the signal times exp(+2i*pi*bin*m/n)
    c.bin = row;
    c.freq = p.carrier_f0 + p.carrier_step * (double)carrier;
    c.first = p.first_sample + (long long)sig * p.n;
    return c;
  };
  s->guard_ties = 0;
  s->guard_max_dev = 0.0;
  const size_t cells_cap = (size_t)std::max(2 * nprn, kGuardListCap);
  if (guard && (gc_buf_reserve(s->b_cells, cells_cap * sizeof(GcExactCell), false) != hipSuccess ||
                gc_buf_reserve(s->b_exact, cells_cap * sizeof(double), false) != hipSuccess ||
                gc_buf_reserve(s->b_list, (size_t)kGuardListCap * sizeof(int2) + 64, false) != hipSuccess ||
                gc_buf_reserve(s->b_rows, (size_t)nprn * N * sizeof(float), false) != hipSuccess)) {
    (void)hipGetLastError();
    gc_set_error("gc_acq_shift_search_batch: device allocation failed");
    return GC_E_NOMEM;
  }
  // float64 values of cells {row, col} of PRN k's results
  auto exact_values = [&](int k, const std::vector<int2>& rc_list, std::vector<double>& vals) -> int {
    std::vector<GcExactCell> cells(rc_list.size());
    for (size_t i = 0; i < rc_list.size(); ++i) cells[i] = cell_of(k, rc_list[i].x, rc_list[i].y);
    GC_HIP(hipMemcpyAsync(s->b_cells.p, cells.data(), cells.size() * sizeof(GcExactCell), hipMemcpyHostToDevice, ctx->stream));
    int rc2 = gc_exact_cells(ctx->stream, ex, (const GcExactCell*)s->b_cells.p, (int)cells.size(), (double*)s->b_exact.p);
    if (rc2) return rc2;
    vals.resize(cells.size());
    GC_HIP(hipMemcpyAsync(vals.data(), s->b_exact.p, vals.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
  };
  // Row `row` of PRN k transformed again into the PRN's slot of b_rows; its cells at or above thr collected (<= kGuardListCap, else
  // *overflow) as {row, col}
  auto row_cells = [&](int k, int row, float thr, std::vector<int2>& list, bool* overflow) -> int {
    int rc2 = shift_row_passes(ctx, s, row, narms, cspec + (size_t)k * narms * N, wts, (float*)s->b_rows.p + (size_t)k * N, /*to_slot=*/true);
    if (rc2) return rc2;
    int* const d_count = (int*)s->b_list.p;
    int2* const d_list = (int2*)((char*)s->b_list.p + 64);
    GC_HIP(hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream));
    rc2 = gc_collect_cells(ctx->stream, (const float*)s->b_rows.p + (size_t)k * N, 1, (long long)N, p.n, thr, d_count, d_list, kGuardListCap);
    if (rc2) return rc2;
    int count = 0;
    GC_HIP(hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    *overflow = count > kGuardListCap;
    list.assign((size_t)std::max(0, std::min(count, kGuardListCap)), make_int2(0, 0));
    if (!list.empty()) GC_HIP(hipMemcpy(list.data(), d_list, list.size() * sizeof(int2), hipMemcpyDeviceToHost));
    for (int2& c : list) c.x = row;  // (the collector numbered the one row it saw 0)
    return GC_OK;
  };

  // the package's selection rule on the row maxima (host: nprn x rows numbers)
  std::vector<double> rmd((size_t)rows);
  std::vector<int> rad((size_t)rows);
  for (int k = 0; k < nprn; ++k) {
    const float* rm = hmax.data() + (size_t)k * rows;
    const int* ra = harg.data() + (size_t)k * rows;
    gc_acq_shift_pick& pk = out[k];
    pk.row = -1;
    pk.code_phase = 0;
    pk.peak = 0.0;
    pk.second_peak = 0.0;
    for (int r = 0; r < rows; ++r) {
      rmd[(size_t)r] = (double)rm[r];
      rad[(size_t)r] = ra[r];
    }
    auto apply_rule = [&]() {
      if (rule == GC_SHIFT_PICK_GLOBAL) {
        // BDS/B1C acquisition.m:193-197: the row of max(max(results,[],2)) (first), the first column holding the global maximum
        int best = 0;
        for (int r = 1; r < rows; ++r)
          if (rmd[(size_t)r] > rmd[(size_t)best]) best = r;
        int col = rad[(size_t)best];
        for (int r = 0; r < rows; ++r)
          if (rmd[(size_t)r] == rmd[(size_t)best] && rad[(size_t)r] < col) col = rad[(size_t)r];
        pk.row = best;
        pk.code_phase = col;
        pk.peak = rmd[(size_t)best];
      } else {
        pk.row = pick_sequential(p, rmd.data(), pairs);
      }
    };
    apply_rule();
    if (!guard || pk.row < 0) continue;
    // rows whose maximum is within eps of the chosen one: which of them the rule takes is decided on their float64 maxima
    const double near = rmd[(size_t)pk.row] * (1.0 - eps);
    std::vector<int> tied;
    for (int r = 0; r < rows; ++r)
      if (rmd[(size_t)r] >= near && rmd[(size_t)r] > 0.0) tied.push_back(r);
    if (tied.size() > 1 && tied.size() <= 64) {
      ++s->guard_ties;
      for (int r : tied) {
        std::vector<int2> list;
        bool overflow = false;
        rc = row_cells(k, r, (float)((double)rm[r] * (1.0 - eps)), list, &overflow);
        if (rc) return rc;
        if (overflow || list.empty()) continue;  // a plateau: the float32 maximum stands for this row
        std::vector<double> vals;
        rc = exact_values(k, list, vals);
        if (rc) return rc;
        double best = -1.0;
        int bc = 0;
        for (size_t i = 0; i < list.size(); ++i)
          if (vals[i] > best || (vals[i] == best && list[i].y < bc)) {
            best = vals[i];
            bc = list[i].y;
          }
        rmd[(size_t)r] = best;
        rad[(size_t)r] = bc;
      }
      apply_rule();
    }
  }
  // phase 2: the winning rows again (their sums were never written), first maximum and second peak on the device, one read-back
  std::vector<ShiftPickDev> dev((size_t)nprn);
  for (int k = 0; k < nprn; ++k) {
    std::memset(&dev[(size_t)k], 0, sizeof(ShiftPickDev));
    dev[(size_t)k].row = out[k].row;
    dev[(size_t)k].second_col = -1;
  }
  if (!second && !guard) return GC_OK;
  if (gc_buf_reserve(s->b_pick, (size_t)nprn * sizeof(ShiftPickDev), false) != hipSuccess) {
    (void)hipGetLastError();
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemcpyAsync(s->b_pick.p, dev.data(), (size_t)nprn * sizeof(ShiftPickDev), hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < nprn; ++k) {
    if (out[k].row < 0) continue;
    const int irow = out[k].row;
    rc = shift_row_passes(ctx, s, irow, narms, cspec + (size_t)k * narms * N, wts, (float*)s->b_rows.p + (size_t)k * N,
                          /*to_slot=*/true);  // row irow lands at b_rows + k * N
    if (rc) return rc;
  }
  hipLaunchKernelGGL(shift_pick_kernel, dim3((unsigned int)nprn), dim3(1024), 0, ctx->stream, (const float*)s->b_rows.p, (long long)N, p.n, exclude,
                     second ? period : 0, (float)eps, (ShiftPickDev*)s->b_pick.p);
  GC_HIP(hipGetLastError());
  rc = shift_read_back(ctx, s, dev.data(), s->b_pick.p, (size_t)nprn * sizeof(ShiftPickDev));
  if (rc) return rc;
  for (int k = 0; k < nprn; ++k) {
    if (out[k].row < 0) continue;
    const ShiftPickDev& d = dev[(size_t)k];
    if (second || !guard) out[k].code_phase = d.code_phase;  // (GC_SHIFT_PICK_GLOBAL took its column from the row maxima; the guard may move it below)
    out[k].peak = (double)d.peak;
    out[k].second_peak = second ? (double)d.second : 0.0;
  }
  if (!guard) return GC_OK;
  // every PRN's peak and second-peak cells in float64 (one launch), then the PRNs whose row holds another cell within eps of either
  {
    std::vector<GcExactCell> cells;
    std::vector<int> owner;
    for (int k = 0; k < nprn; ++k) {
      if (out[k].row < 0) continue;
      cells.push_back(cell_of(k, out[k].row, second ? dev[(size_t)k].code_phase : out[k].code_phase));
      owner.push_back(2 * k);
      if (second && dev[(size_t)k].second_col >= 0) {
        cells.push_back(cell_of(k, out[k].row, dev[(size_t)k].second_col));
        owner.push_back(2 * k + 1);
      }
    }
    if (!cells.empty()) {
      GC_HIP(hipMemcpyAsync(s->b_cells.p, cells.data(), cells.size() * sizeof(GcExactCell), hipMemcpyHostToDevice, ctx->stream));
      rc = gc_exact_cells(ctx->stream, ex, (const GcExactCell*)s->b_cells.p, (int)cells.size(), (double*)s->b_exact.p);
      if (rc) return rc;
      std::vector<double> vals(cells.size());
      GC_HIP(hipMemcpyAsync(vals.data(), s->b_exact.p, vals.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      GC_HIP(hipStreamSynchronize(ctx->stream));
      for (size_t i = 0; i < cells.size(); ++i) {
        const int k = owner[i] / 2;
        double& dst = (owner[i] & 1) ? out[k].second_peak : out[k].peak;
        if (vals[i] > 0.0) s->guard_max_dev = std::max(s->guard_max_dev, std::fabs(dst - vals[i]) / vals[i]);
        dst = vals[i];
      }
    }
  }
  for (int k = 0; k < nprn; ++k) {
    const ShiftPickDev& d = dev[(size_t)k];
    if (out[k].row < 0 || (d.near_peak <= 1 && d.near_second <= 1)) continue;
    ++s->guard_ties;
    // every cell of the winning row that could be the first maximum or the second peak: all those at or above the smaller of the two
    // float32 values less eps (the peak's lobe is among them).  Then the reference's rules on the float64 values.
    const float low = second && d.second_col >= 0 ? std::min(d.peak, d.second) : d.peak;
    std::vector<int2> list;
    bool overflow = false;
    rc = row_cells(k, out[k].row, (float)((double)low * (1.0 - eps)), list, &overflow);
    if (rc) return rc;
    if (overflow || list.empty()) continue;
    std::vector<double> vals;
    rc = exact_values(k, list, vals);
    if (rc) return rc;
    double best = -1.0;
    int bc = 0;
    for (size_t i = 0; i < list.size(); ++i)
      if (vals[i] > best || (vals[i] == best && list[i].y < bc)) {
        best = vals[i];
        bc = list[i].y;
      }
    out[k].code_phase = bc;
    out[k].peak = best;
    if (second) {
      // the reference's three range cases around the (float64) first maximum, 1-based (B1I :141-156, L2C :77-91)
      const int cp = bc + 1, e1 = cp - exclude, e2 = cp + exclude;
      int lo0, hi0, lo1 = 1, hi1 = 0;
      if (e1 < 2) {
        lo0 = e2;
        hi0 = period + e1;
      } else if (e2 >= period) {
        lo0 = e2 - period + 1;
        hi0 = e1;
      } else {
        lo0 = 1;
        hi0 = e1;
        lo1 = e2;
        hi1 = period;
      }
      double sec = -1.0;
      for (size_t i = 0; i < list.size(); ++i) {
        const int c1 = list[i].y + 1;
        if ((c1 >= lo0 && c1 <= hi0) || (c1 >= lo1 && c1 <= hi1)) sec = std::max(sec, vals[i]);
      }
      if (sec >= 0.0) out[k].second_peak = sec;
    }
  }
  return GC_OK;
}

// Test hook: forward FFT of `nbatch` host sequences of length n (complex64) with the library's
// transform; output in natural frequency order.
extern "C" int gc_acq_guard_stats(gc_context* ctx, int32_t* ties, double* max_dev, double* eps) {
  AcqScratch* s = ctx ? (AcqScratch*)ctx->acq_scratch : nullptr;
  if (!s) {
    gc_set_error("gc_acq_guard_stats: nothing searched yet");
    return GC_E_STATE;
  }
  if (ties) *ties = s->guard_ties;
  if (max_dev) *max_dev = s->guard_max_dev;
  if (eps) *eps = gc_acq_tie_eps(s->plan.n);
  return GC_OK;
}

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

extern "C" int gc_acquire_fine_l1ca_batch(gc_context* ctx, const gc_acq_params* p, int ndet, const int8_t* codes,
                                          const int32_t* code_phase, const double* coarse_freq, double* carr_freq) {
  if (!ctx || !p || ndet <= 0 || !codes || !code_phase || !coarse_freq || !carr_freq) {
    gc_set_error("gc_acquire_fine_l1ca: bad arguments");
    return GC_E_INVALID;
  }
  const double x = p->sampling_freq / (p->code_freq_basis / p->code_length);
  const int spc = (int)std::floor(x + 0.5);
  const int ncodes = 40;
  const double fine_step = 25;                                                   // acquisition.m:138
  const int nfine = (int)std::floor(p->search_step / fine_step + 0.5) + 1;      // :140
  gc_fine_params fp;
  std::memset(&fp, 0, sizeof fp);
  fp.sampling_freq = p->sampling_freq;
  fp.code_freq = p->code_freq_basis;
  fp.fstep = fine_step;
  fp.spc = spc;
  fp.ncodes = ncodes;
  fp.nbins = nfine;
  fp.code_len = (int)p->code_length;
  fp.index_offset = 0;                                                           // codeValueIndex over (0 : 40*spc-1), :210
  fp.source = p->source;
  std::vector<int64_t> first((size_t)ndet);
  std::vector<double> f0((size_t)ndet);
  for (int d = 0; d < ndet; ++d) {
    if (code_phase[d] < 1) {
      gc_set_error("gc_acquire_fine_l1ca: bad arguments");
      return GC_E_INVALID;
    }
    first[d] = p->first_sample + code_phase[d] - 1;                              // sig40cm, :221
    f0[d] = coarse_freq[d] + p->search_step / 2;                                 // fineFreqBins(1), :227-228
  }
  // the hypothesis search on the device (fine_l1ca_pick_kernel): one bin index per detection comes back instead of every sum
  // (GC_ACQ_FINE_HOST=1: the sums come back and the host loop below picks, as before)
  if (nfine <= 64 && !GC_TUNE_ENV("GC_ACQ_FINE_HOST")) {
    std::vector<FineDet> hdet;
    const double* dsums = nullptr;
    int rc = fine_sums_enqueue(ctx, &fp, ndet, codes, first.data(), f0.data(), hdet, &dsums);
    GcBuf& bpick = ctx->acqbuf[gc_context::ACQ_FINE_DET];  // (the detections' records were consumed by the sums kernel queued before)
    std::vector<int> best((size_t)ndet, 0);
    hipError_t e = hipSuccess;
    if (rc == GC_OK) {
      int* const dbest = reinterpret_cast<int*>(reinterpret_cast<char*>(bpick.p));
      hipLaunchKernelGGL(fine_l1ca_pick_kernel, dim3((unsigned int)ndet), dim3(64), 0, ctx->stream, dsums, nfine, ncodes, dbest);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(best.data(), dbest, (size_t)ndet * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);  // also keeps hdet / codes alive until their copies are done
    if (rc) return rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) {
      gc_set_error("gc_acquire_fine_l1ca: %s", hipGetErrorString(e));
      return GC_E_HIP;
    }
    for (int d = 0; d < ndet; ++d) {
      double f = f0[d] - fine_step * best[d];
      if (f == 0) f = 1;  // :258-260
      carr_freq[d] = f;
    }
    return GC_OK;
  }
  std::vector<double> h((size_t)ndet * nfine * ncodes * 2);
  const int rc = gc_acquire_fine_sums_batch(ctx, &fp, ndet, codes, first.data(), f0.data(), h.data());
  if (rc) return rc;
  for (int d = 0; d < ndet; ++d) {
    // 20 navigation-bit-edge hypotheses, max |sum of 20 consecutive per-code sums| (:242-249); first max (:253)
    const double* hd = h.data() + (size_t)d * nfine * ncodes * 2;
    double best = -1.0;
    int best_bin = 0;
    for (int b = 0; b < nfine; ++b) {
      double max_power = 0.0;
      for (int c0 = 0; c0 < 20; ++c0) {
        double sr = 0.0, si = 0.0;
        for (int c = c0; c < c0 + 20; ++c) {
          sr += hd[2 * ((size_t)b * ncodes + c)];
          si += hd[2 * ((size_t)b * ncodes + c) + 1];
        }
        max_power = std::max(max_power, std::sqrt(sr * sr + si * si));
      }
      if (max_power > best) {
        best = max_power;
        best_bin = b;
      }
    }
    double f = f0[d] - fine_step * best_bin;
    if (f == 0) f = 1;  // :258-260
    carr_freq[d] = f;
  }
  return GC_OK;
}

extern "C" int gc_acquire_fine_l1ca(gc_context* ctx, const gc_acq_params* p, const int8_t* code, int code_phase,
                                    double coarse_freq, double* carr_freq) {
  const int32_t cp = code_phase;
  return gc_acquire_fine_l1ca_batch(ctx, p, 1, code, &cp, &coarse_freq, carr_freq);
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
