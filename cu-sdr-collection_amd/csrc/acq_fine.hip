// acq_fine.hip - fine-frequency stages: per-code-period sums at every fine bin for all detections of a search in one launch, and GPS L1 C/A's
// bin pick on the device.  Reference: GPS/GPS_L1CA/include/acquisition.m:213-260; the per-package hypothesis searches over these sums
// (NH20, secondary codes, split sums) stay with the caller (acq_family.py / the MATLAB drop-ins).
// Split out of acq.hip in round 6 (same code, one translation unit per part of the search; shared declarations: acq_internal.h).
#include "acq_internal.h"

using namespace gcacq;

namespace {
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
}  // namespace

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

extern "C" int gc_acquire_fine_sums(gc_context* ctx, const gc_fine_params* p, const int8_t* code, double* out) {
  if (!p) {
    gc_set_error("gc_acquire_fine_sums: bad arguments");
    return GC_E_INVALID;
  }
  const int64_t first = p->first_sample;
  return gc_acquire_fine_sums_batch(ctx, p, 1, code, &first, &p->f0, out);
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
