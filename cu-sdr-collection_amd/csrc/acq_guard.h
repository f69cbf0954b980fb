// acq_guard.h - float64 re-evaluation of single cells of an acquisition search (csrc/acq_guard.hip).
//
// The searches (acq_coarse.hip, acq_shift.hip) transform in float32.  The reference decides in float64: `[peakSize, codePhase] = max(max(results))`,
// `peakMetric > acqThreshold` (GPS/GPS_L1CA/include/acquisition.m:196-206), `max_peak / second` (BDS/B1I/include/acquisition.m:
// 141-166, GPS/GPS_L2C/include/acquisition.m:89-112).  Two cells of `results` that lie closer together than the float32 transforms'
// rounding, or a metric that close to the threshold, would be decided by that rounding.  The guard closes the gap without a float64
// transform: one cell of `results` is, term by term, a circular correlation at ONE lag -
//
//     results(b, tau) = sum_hops sum_arms w_arm * | sum_{n < cl} z_h[(n + tau) mod blk] * code_arm[n] |,
//     z_h[m] = x_h[m] * exp(-1i * f_b * phasePoints(m)) * exp(+2i*pi * s * m / blk)
//
// (ifft(fft(a) .* conj(fft(b)))(tau) = sum_n a[n + tau] conj(b[n]); circshift(fft(a), s) = fft(a .* exp(2i*pi*s*m/N)): checked against
// the oracle's FFT-based rows to 4e-16, tests/test_oracle_semantics.py) - cl multiply-adds per hop and arm in float64, microseconds for
// the handful of cells that matter:
//   * always: the winning cell of every PRN, so that peak / peakMetric leave the library with float64 accuracy (~1e-15 instead of ~1e-6)
//     and the caller's threshold test is the reference's;
//   * when the search's runner-up is within GC_ACQ_TIE_EPS of the winner: every cell that close, then the reference's first-occurrence
//     rule on the float64 values.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct gc_context;

// What a cell's value is made of (one search; the cells say where)
struct GcExactSetup {
  const int8_t* if_i8 = nullptr;   // int8 I/Q record ...
  const float2* if_f32 = nullptr;  // ... or the conditioned complex float signal (gc_acq_condition) when non-null
  int blk = 0;                     // circular period: the reference's transform length (2*samplesPerCode, acquisition.m:174; the block of a circshift search)
  int cl = 0;                      // replica samples that are not zero padding (samplesPerCode; samplesXmsLen)
  int hop_stride = 0, nhops = 1;   // non-coherent hops: block h starts hop_stride * h samples later (acquisition.m:177-178)
  int narms = 1;
  double w[4] = {1.0, 1.0, 1.0, 1.0};
  const int8_t* codes = nullptr;   // device, sampled replicas: row (code * narms + arm) * code_stride
  long long code_stride = 0;
  double fs = 0.0;
};

struct GcExactCell {
  int code;         // which PRN's codes
  int col;          // tau, 0-based
  int shift;        // s: whole FFT bins the signal spectrum is moved by (circshift family), any sign; 0 in the carrier-per-bin searches
  int bin;          // carried through for the caller (not used by the kernel)
  double freq;      // f_b in Hz
  long long first;  // first sample of hop 0's block within the record / the conditioned signal
};

// Winner's cell of every PRN from the search's peak keys {(bits << 32) | ~bin, (bits << 32) | ~col}: freq = f0 + off[ip] - fstep * bin
int gc_exact_cells_from_keys(hipStream_t stream, const unsigned long long* keys, int nprn, double f0, double fstep, const double* d_off,
                             long long first, GcExactCell* d_cells);
// partial[cell * nhops + hop] = sum_arm w_arm |...|; the caller adds the hops in hop order (acquisition.m:186-190)
int gc_exact_cells(hipStream_t stream, const GcExactSetup& s, const GcExactCell* d_cells, int ncells, double* d_partial);
// Cells of a [rows][row_stride] float array at or above `thr` within its first `valid` columns: list[k] = {row, col}, *count = how many
// there are (the list holds the first `cap` in no particular order; *count > cap: it overflowed).  *count must be zero before.
int gc_collect_cells(hipStream_t stream, const float* r, int rows, long long row_stride, int valid, float thr, int* d_count, int2* d_list, int cap);

// Relative distance under which two float32 results count as tied (and a collected cell as a candidate): 8 * log2(N) * 2^-24.
// The float32 forward + inverse transforms, product, magnitude and hop sums of a cell stay within ~(2 log2 N + 4) * 2^-24 ~ 2e-6 of the
// float64 value relative to the peak in the worst case and measure <= 2.8e-7 (the twelve default-size searches and every fixture
// scene: scripts/acq_guard_report.py; bench.py prints the run's own figure, acq_f32_vs_f64_peak_max_rel): two cells of a comparison
// stay inside the band even at the bound, with 27x margin on what is measured.  7.6e-6 at N = 36 000, 9e-6 at N = 360 000.  (A wider band
// only costs time: every noise-only PRN whose two largest cells happen to lie that close takes the slow path - 64 * log2 N had one
// such PRN in the default GPS L1 C/A search, +0.25 ms.)
double gc_acq_tie_eps(int n);  // (acq_guard.hip; GC_ACQ_GUARD_EPS=<value> in the tuning build: that band instead - the tests widen it to send every search through the slow path)
