// acq_cond.hip - input conditioning of the acquisition: zero-phase FIR(700) band-pass + band-pass-sampling decimation,
// and the float copy of records that are not int8 I/Q.  Reference: GPS/GPS_L1CA/include/acquisition.m:46-111 (the same block with its own
// bandwidth in ten packages), postProcessing.m:61-96 (dataType / fileType).
// Split out of acq.hip in round 6 (same code, one translation unit per part of the search; shared declarations: acq_internal.h).
#include "acq_internal.h"

using namespace gcacq;

namespace {
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
}  // namespace

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
