// acq_shift.hip - the circshift search family (GPS L2C, BDS B1I, BDS B1C): ONE signal spectrum per carrier and block, Doppler bins as circular
// shifts of it inside the inverse transform; per-PRN calls and the whole-package batch call with its float64 guard.
// Reference: GPS/GPS_L2C/include/acquisition.m:40-118, BDS/B1I/include/acquisition.m:76-176, BDS/B1C/include/acquisition.m:137-235.
// Split out of acq.hip in round 6 (same code, one translation unit per part of the search; shared declarations: acq_internal.h).
#include "acq_internal.h"

using namespace gcacq;

namespace {

__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ r, int ncols, int stride, float* vmax, int* amax, float* vsecond = nullptr) {
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
  if (!vsecond) return;
  // the row's runner-up for the float64 guard: the largest value at any OTHER column (== the maximum when a second column holds it)
  const float m1 = sv[0];
  const int a1 = si[0];
  __syncthreads();
  float sec = -1.0f;
  for (int i = threadIdx.x; i < ncols; i += 256)
    if (i != a1) sec = fmaxf(sec, row[i]);
  sv[threadIdx.x] = sec;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sv[threadIdx.x] = fmaxf(sv[threadIdx.x], sv[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) vsecond[blockIdx.x] = fminf(sv[0], m1);
}
}  // namespace

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
  s->rowsecond = nullptr;  // (only the batch search tracks the rows' runner-ups)
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
                           size_t bytes1 = 0, void* dst2 = nullptr, const void* src2 = nullptr, size_t bytes2 = 0) {
  const AcqBack back[3] = {{dst0, src0, bytes0}, {dst1, src1, bytes1}, {dst2, src2, bytes2}};
  return acq_read_back(ctx, s, back, 3);
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

// gc_acq_shift_search_batch where the passes WRITE the rows (no per-tile candidates): PRN by PRN inside the call - rows and columns
// passes, row maxima with their runner-ups (rowmax_kernel), the package's rule on the host, rows within eps of the chosen one and the
// cells within eps of the first maximum / second peak re-evaluated in float64 (acq_guard.h) from the rows as they lie in s->results.
// b_codes (replicas, [nprn * narms][p.n]) and cspec (their spectra) are in place.
static int shift_batch_written(gc_context* ctx, AcqScratch* s, int nprn, int narms, const float2* cspec, const double* arm_weight, int rule, int exclude,
                               int period, int code_samples, gc_acq_shift_pick* out) {
  const gc_acq_shift_params& p = s->shift;
  const Plan& pl = s->plan;
  const int rows = p.n_carriers * p.n_signals * p.n_bins;
  const size_t N = (size_t)pl.n;
  const bool pairs = rule == GC_SHIFT_PICK_SEQUENTIAL_PAIRS, second = rule != GC_SHIFT_PICK_GLOBAL;
  const bool guard = GC_TUNE_ENV("GC_ACQ_NO_GUARD") == nullptr;
  const double eps = gc_acq_tie_eps(pl.n);
  const double ones[4] = {1.0, 1.0, 1.0, 1.0};
  const double* const wts = arm_weight ? arm_weight : ones;
  GcExactSetup ex;
  ex.if_i8 = p.source == GC_ACQ_SOURCE_CONDITIONED ? nullptr : (const int8_t*)ctx->d_if;
  ex.if_f32 = p.source == GC_ACQ_SOURCE_CONDITIONED ? (const float2*)ctx->acqbuf[gc_context::ACQ_COND_SIG].p : nullptr;
  ex.blk = p.n;
  ex.cl = code_samples;
  ex.hop_stride = 0;
  ex.nhops = 1;
  ex.narms = narms;
  for (int arm = 0; arm < narms; ++arm) ex.w[arm] = wts[arm];
  ex.codes = (const int8_t*)s->b_codes.p;
  ex.code_stride = p.n;
  ex.fs = p.sampling_freq;
  if (gc_buf_reserve(s->b_cells, (size_t)kGuardListCap * sizeof(GcExactCell), false) != hipSuccess ||
      gc_buf_reserve(s->b_exact, (size_t)kGuardListCap * sizeof(double), false) != hipSuccess ||
      gc_buf_reserve(s->b_list, (size_t)kGuardListCap * sizeof(int2) + 64, false) != hipSuccess ||
      gc_buf_reserve(s->b_rowsec, (size_t)rows * sizeof(float), false) != hipSuccess ||
      gc_buf_reserve(s->b_pick, sizeof(ShiftPickDev), false) != hipSuccess) {
    (void)hipGetLastError();
    gc_set_error("gc_acq_shift_search_batch: device allocation failed");
    return GC_E_NOMEM;
  }
  s->guard_ties = 0;
  s->guard_max_dev = 0.0;
  s->rowsecond = nullptr;
  auto irow_of = [&](int row) { return s->shift_padded ? shift_internal_row(p, row) : row; };
  auto exact_values = [&](int k, const std::vector<int2>& rc_list, std::vector<double>& vals) -> int {
    std::vector<GcExactCell> cells(rc_list.size());
    for (size_t i = 0; i < rc_list.size(); ++i) {
      GcExactCell& c = cells[i];
      const int row = rc_list[i].x;
      c.code = k;
      c.col = rc_list[i].y;
      c.shift = row % p.n_bins;
      c.bin = row;
      c.freq = p.carrier_f0 + p.carrier_step * (double)(row / (p.n_signals * p.n_bins));
      c.first = p.first_sample + (long long)((row / p.n_bins) % p.n_signals) * p.n;
    }
    GC_HIP(hipMemcpyAsync(s->b_cells.p, cells.data(), cells.size() * sizeof(GcExactCell), hipMemcpyHostToDevice, ctx->stream));
    int rc2 = gc_exact_cells(ctx->stream, ex, (const GcExactCell*)s->b_cells.p, (int)cells.size(), (double*)s->b_exact.p);
    if (rc2) return rc2;
    vals.resize(cells.size());
    GC_HIP(hipMemcpyAsync(vals.data(), s->b_exact.p, vals.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    return GC_OK;
  };
  // cells of public row `row` (as it lies in s->results) at or above thr
  auto row_cells = [&](int row, float thr, std::vector<int2>& list, bool* overflow) -> int {
    int* const d_count = (int*)s->b_list.p;
    int2* const d_list = (int2*)((char*)s->b_list.p + 64);
    GC_HIP(hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream));
    int rc2 = gc_collect_cells(ctx->stream, s->results + (size_t)irow_of(row) * N, 1, (long long)N, p.n, thr, d_count, d_list, kGuardListCap);
    if (rc2) return rc2;
    int count = 0;
    GC_HIP(hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
    *overflow = count > kGuardListCap;
    list.assign((size_t)std::max(0, std::min(count, kGuardListCap)), make_int2(0, 0));
    if (!list.empty()) GC_HIP(hipMemcpy(list.data(), d_list, list.size() * sizeof(int2), hipMemcpyDeviceToHost));
    for (int2& c : list) c.x = row;
    return GC_OK;
  };
  std::vector<float> hm((size_t)rows), hs((size_t)rows);
  std::vector<int> ha((size_t)rows);
  std::vector<double> rmd((size_t)rows);
  std::vector<int> rad((size_t)rows);
  for (int k = 0; k < nprn; ++k) {
    gc_acq_shift_pick& pk = out[k];
    pk.row = -1;
    pk.code_phase = 0;
    pk.peak = 0.0;
    pk.second_peak = 0.0;
    bool all_fused = false;
    int rc = shift_search_passes(ctx, s, narms, cspec + (size_t)k * narms * N, arm_weight, &all_fused, s->tmp);
    if (rc) return rc;
    if (all_fused) {
      gc_set_error("gc_acq_shift_search_batch: internal - the written-rows path met fused row candidates");
      return GC_E_STATE;
    }
    hipLaunchKernelGGL(rowmax_kernel, dim3(rows), dim3(256), 0, ctx->stream, s->results, p.n, pl.n, s->rowmax, s->rowarg, (float*)s->b_rowsec.p);
    GC_HIP(hipGetLastError());
    rc = shift_read_back(ctx, s, hm.data(), s->rowmax, sizeof(float) * rows, ha.data(), s->rowarg, sizeof(int) * rows, hs.data(), s->b_rowsec.p, sizeof(float) * rows);
    if (rc) return rc;
    for (int r = 0; r < rows; ++r) {  // public order
      rmd[(size_t)r] = (double)hm[(size_t)irow_of(r)];
      rad[(size_t)r] = ha[(size_t)irow_of(r)];
    }
    auto apply_rule = [&]() {
      if (rule == GC_SHIFT_PICK_GLOBAL) {
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
    if (pk.row < 0) continue;
    if (guard) {
      const double near = rmd[(size_t)pk.row] * (1.0 - eps);
      std::vector<int> tied;
      for (int r = 0; r < rows; ++r)
        if (rmd[(size_t)r] >= near && rmd[(size_t)r] > 0.0) tied.push_back(r);
      if (tied.size() > 1 && tied.size() <= 64) {
        ++s->guard_ties;
        for (int r : tied) {
          std::vector<int2> list;
          bool overflow = false;
          rc = row_cells(r, (float)(rmd[(size_t)r] * (1.0 - eps)), list, &overflow);
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
          rmd[(size_t)r] = best;
          rad[(size_t)r] = bc;
        }
        apply_rule();
      }
    }
    // first maximum and second peak of the chosen row, from the row in memory
    ShiftPickDev d;
    std::memset(&d, 0, sizeof d);
    d.row = pk.row;
    d.second_col = -1;
    GC_HIP(hipMemcpyAsync(s->b_pick.p, &d, sizeof d, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(shift_pick_kernel, dim3(1), dim3(1024), 0, ctx->stream, s->results + (size_t)irow_of(pk.row) * N, (long long)N, p.n, exclude,
                       second ? period : 0, (float)eps, (ShiftPickDev*)s->b_pick.p);
    GC_HIP(hipGetLastError());
    rc = shift_read_back(ctx, s, &d, s->b_pick.p, sizeof d);
    if (rc) return rc;
    if (second) pk.code_phase = d.code_phase;
    pk.peak = (double)d.peak;
    pk.second_peak = second ? (double)d.second : 0.0;
    if (!guard) continue;
    {
      std::vector<int2> two;
      two.push_back(make_int2(pk.row, pk.code_phase));
      if (second && d.second_col >= 0) two.push_back(make_int2(pk.row, d.second_col));
      std::vector<double> vals;
      rc = exact_values(k, two, vals);
      if (rc) return rc;
      if (vals[0] > 0.0) s->guard_max_dev = std::max(s->guard_max_dev, std::fabs(pk.peak - vals[0]) / vals[0]);
      pk.peak = vals[0];
      if (two.size() > 1) pk.second_peak = vals[1];
    }
    if (d.near_peak <= 1 && d.near_second <= 1) continue;
    ++s->guard_ties;
    const float low = second && d.second_col >= 0 ? std::min(d.peak, d.second) : d.peak;
    std::vector<int2> list;
    bool overflow = false;
    rc = row_cells(pk.row, (float)((double)low * (1.0 - eps)), list, &overflow);
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
    pk.code_phase = bc;
    pk.peak = best;
    if (second) {
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
      if (sec >= 0.0) pk.second_peak = sec;
    }
  }
  // no single PRN's search is "the last one" for gc_acq_shift_row after this call
  s->shift_rows_fused = true;
  s->shift_narms = 0;
  return GC_OK;
}

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
  // Block lengths without specialised pass kernels (16.368-Msps front ends: padded transforms; GC_ACQ_GENERIC / GC_ACQ_ROWMAX_KERNEL in the
  // tuning build): the passes write every row's sums, so the PRNs are searched one after the other and each one's rows are picked, and
  // guarded, straight from s->results before the next PRN overwrites them (round 5 answered GC_E_UNSUPPORTED here and left the rules to
  // the caller, on float32 rows)
  const bool written_rows = s->shift_padded || ct_columns_tile(s->plan.p1.len, s->plan.n2) == 0 || GC_TUNE_ENV("GC_ACQ_GENERIC") || GC_TUNE_ENV("GC_ACQ_ROWMAX_KERNEL");
  GC_HIP(hipSetDevice(ctx->device));
  const Plan& pl = s->plan;
  const int rows = p.n_carriers * p.n_signals * p.n_bins;
  const size_t N = (size_t)pl.n;
  const bool second = rule != GC_SHIFT_PICK_GLOBAL;
  if (gc_buf_reserve(s->b_codes, (size_t)nprn * narms * p.n, false) != hipSuccess ||
      gc_buf_reserve(s->b_codespec, (size_t)nprn * narms * N * sizeof(float2), false) != hipSuccess ||
      gc_buf_reserve(s->b_rowmax, (size_t)nprn * rows * sizeof(float), false) != hipSuccess ||
      gc_buf_reserve(s->b_rowarg, (size_t)nprn * rows * sizeof(int), false) != hipSuccess ||
      gc_buf_reserve(s->b_rowsec, (size_t)nprn * rows * sizeof(float), false) != hipSuccess ||
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
  if (written_rows) return shift_batch_written(ctx, s, nprn, narms, cspec, arm_weight, rule, exclude, period, sample_index ? n_index : p.n, out);
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
    s->rowsecond = (float*)s->b_rowsec.p + (size_t)k * rows;
    rc = shift_search_passes(ctx, s, narms, cspec + (size_t)k * narms * N, arm_weight, &fused, s->lane ? s->tmp2 : s->tmp);
  }
  ctx->stream = stream1;
  s->lane = 0;
  s->shift_slot_lanes = 1;
  s->rowmax = save_max;
  s->rowarg = save_arg;
  s->rowsecond = nullptr;
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
  std::vector<float> hmax((size_t)nprn * rows), hsecond((size_t)nprn * rows);
  std::vector<int> harg((size_t)nprn * rows);
  rc = shift_read_back(ctx, s, hmax.data(), s->b_rowmax.p, sizeof(float) * hmax.size(), harg.data(), s->b_rowarg.p, sizeof(int) * harg.size(),
                       hsecond.data(), s->b_rowsec.p, sizeof(float) * hsecond.size());
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
  if (second) {
    GC_HIP(hipMemcpyAsync(s->b_pick.p, dev.data(), (size_t)nprn * sizeof(ShiftPickDev), hipMemcpyHostToDevice, ctx->stream));
    for (int k = 0; k < nprn; ++k) {
      if (out[k].row < 0) continue;
      const int irow = out[k].row;
      rc = shift_row_passes(ctx, s, irow, narms, cspec + (size_t)k * narms * N, wts, (float*)s->b_rows.p + (size_t)k * N,
                            /*to_slot=*/true);  // row irow lands at b_rows + k * N
      if (rc) return rc;
    }
    hipLaunchKernelGGL(shift_pick_kernel, dim3((unsigned int)nprn), dim3(1024), 0, ctx->stream, (const float*)s->b_rows.p, (long long)N, p.n, exclude, period,
                       (float)eps, (ShiftPickDev*)s->b_pick.p);
    GC_HIP(hipGetLastError());
    rc = shift_read_back(ctx, s, dev.data(), s->b_pick.p, (size_t)nprn * sizeof(ShiftPickDev));
    if (rc) return rc;
    for (int k = 0; k < nprn; ++k) {
      if (out[k].row < 0) continue;
      const ShiftPickDev& d = dev[(size_t)k];
      out[k].code_phase = d.code_phase;
      out[k].peak = (double)d.peak;
      out[k].second_peak = (double)d.second;
    }
  } else {
    // GC_SHIFT_PICK_GLOBAL has no second peak and took its column from the row maxima: the winning row is not transformed again -
    // whether another of its cells lies within eps of the maximum is known from the row's runner-up (PeakTrack::m2 through the
    // per-tile slots, rowkeys_reduce_kernel)
    for (int k = 0; k < nprn; ++k) {
      if (out[k].row < 0) continue;
      ShiftPickDev& d = dev[(size_t)k];
      const size_t at = (size_t)k * rows + (size_t)out[k].row;
      d.code_phase = out[k].code_phase;
      d.peak = (float)out[k].peak;
      d.near_peak = (double)hsecond[at] >= (double)hmax[at] * (1.0 - eps) ? 2 : 1;
      d.near_second = 0;
    }
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
