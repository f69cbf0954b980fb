// navsync.hip — bit-synchronisation front end of navigation decoding, the first consumer of the correlator output
// (SURVEY.md §8f item 4): GPS/GPS_L1CA/include/NAVdecoding.m:62-76 hard-limits the prompt in-phase stream to +-1 and
// cross-correlates it with the 160-sample TLM preamble pattern (xcorr, non-negative lags).  One thread per lag, the
// pattern in LDS, the sign taken on the fly: out[l] = sum_k sgn(I_P[l + k]) * pattern[k], terms beyond the end dropped.
#include "gc_internal.h"

namespace {
__global__ __launch_bounds__(256) void preamble_xcorr_kernel(const double* __restrict__ ip, long long n, const int8_t* __restrict__ pat,
                                                             int m, float* __restrict__ out) {
  extern __shared__ float spat[];
  for (int k = threadIdx.x; k < m; k += blockDim.x) spat[k] = (float)pat[k];
  __syncthreads();
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  float acc = 0.0f;
  const int kmax = (int)((n - l < (long long)m) ? (n - l) : (long long)m);
  for (int k = 0; k < kmax; ++k) acc += (ip[l + k] > 0.0 ? 1.0f : -1.0f) * spat[k];  // bits(bits > 0) = 1; bits(bits <= 0) = -1
  out[l] = acc;
}
}  // namespace

extern "C" int gc_preamble_xcorr(gc_context* ctx, const double* i_p, int64_t n, const int8_t* pattern, int m, float* out) {
  if (!ctx || !i_p || !pattern || !out || n <= 0 || m <= 0 || m > 8192) {
    gc_set_error("gc_preamble_xcorr: bad arguments");
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  double* d_ip = nullptr;
  int8_t* d_pat = nullptr;
  float* d_out = nullptr;
  hipError_t e = hipMalloc((void**)&d_ip, sizeof(double) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void**)&d_pat, (size_t)m);
  if (e == hipSuccess) e = hipMalloc((void**)&d_out, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ip, i_p, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pat, pattern, (size_t)m, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(preamble_xcorr_kernel, dim3((unsigned int)((n + 255) / 256)), dim3(256), sizeof(float) * (size_t)m, ctx->stream,
                       (const double*)d_ip, (long long)n, (const int8_t*)d_pat, m, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (d_ip) (void)hipFree(d_ip);
  if (d_pat) (void)hipFree(d_pat);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) {
    gc_set_error("gc_preamble_xcorr: %s", hipGetErrorString(e));
    return GC_E_HIP;
  }
  return GC_OK;
}
