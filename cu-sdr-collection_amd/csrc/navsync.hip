// navsync.hip — bit-synchronisation front end of navigation decoding, the first consumer of the correlator output
// (SURVEY.md §8f item 4).  Every package's NAVdecoding.m hard-limits the prompt in-phase stream and cross-correlates it with
// its sync pattern stretched to the stream's rate (xcorr, of which only the non-negative lags are looked at):
//   GPS/GPS_L1CA/include/NAVdecoding.m:69-85   8-bit TLM preamble x 20 ms              (160 samples)
//   GAL/GAL_E1C/include/NAVdecoding.m:79-88    10-symbol sync pattern, one value per 4-ms symbol; bits = (I_P < 0)
//   GAL/GAL_E5a/include/NAVdecoding.m:69-95    12 sync symbols x the 20-chip secondary code  (240)
//   GAL/GAL_E5b/include/NAVdecoding.m:80-100   10 preamble symbols x the 4-chip secondary code (40)
//   BDS/B1I/include/NAVdecoding.m:71-105, BDS/B3I/include/NAVdecoding.m:72-97   11-bit preamble x -NH20 (D1, 220) or x 2 (D2, 22)
//   GLO/GLO_GL1/include/NAVdecoding.m:69-86    30-bit time mark x 10 ms                  (300)
// out[l] = sum_k s(I_P[l + k]) * pattern[k], terms beyond the end dropped (xcorr pads with zeros), with
// s(x) = +1 for x > 0, -1 otherwise (`bits(bits > 0) = 1; bits(bits <= 0) = -1`), or - GC_SYNC_ZERO_IS_PLUS, Galileo E1's
// `1 - 2*(I_P < 0)` - +1 for x >= 0.  A workgroup owns 1024 lags: the signs of its 1024 + m samples go to LDS once as floats,
// the pattern next to them; a thread sums four lags 256 apart (conflict-free reads, the pattern entry is a broadcast).
#include "gc_internal.h"

namespace {
constexpr int kLagsPerBlock = 1024;
constexpr int kMaxPattern = 8192;

__global__ __launch_bounds__(256) void sync_xcorr_kernel(const double* __restrict__ ip, long long n, const int8_t* __restrict__ pat, int m,
                                                         int zero_is_plus, float* __restrict__ out) {
  extern __shared__ float lds[];
  float* spat = lds;          // [m]
  float* sgn = lds + m;       // [kLagsPerBlock + m]
  const long long l0 = (long long)blockIdx.x * kLagsPerBlock;
  for (int k = threadIdx.x; k < m; k += 256) spat[k] = (float)pat[k];
  for (int k = threadIdx.x; k < kLagsPerBlock + m; k += 256) {
    const long long i = l0 + k;
    float s = 0.0f;  // past the end: xcorr's zero padding
    if (i < n) {
      const double x = ip[i];
      s = (zero_is_plus ? (x < 0.0) : !(x > 0.0)) ? -1.0f : 1.0f;
    }
    sgn[k] = s;
  }
  __syncthreads();
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = 0; k < m; ++k) {
    const float p = spat[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fmaf(sgn[threadIdx.x + 256 * j + k], p, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long l = l0 + threadIdx.x + 256 * j;
    if (l < n) out[l] = acc[j];
  }
}
}  // namespace

extern "C" int gc_sync_xcorr(gc_context* ctx, const double* i_p, int64_t n, const int8_t* pattern, int m, int flags, float* out) {
  if (!ctx || !i_p || !pattern || !out || n <= 0 || m <= 0 || m > kMaxPattern || (flags & ~GC_SYNC_ZERO_IS_PLUS)) {
    gc_set_error("gc_sync_xcorr: bad arguments (pattern length 1 .. %d, flags 0 or GC_SYNC_ZERO_IS_PLUS)", kMaxPattern);
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  // the stream, the pattern and the result live in grow-only buffers of the context: a hipFree per call waits for every stream
  // of the device, and a receiver calls this once per channel
  hipError_t e = gc_buf_reserve(ctx->nav[0], sizeof(double) * (size_t)n, false);
  if (e == hipSuccess) e = gc_buf_reserve(ctx->nav[1], (size_t)m, false);
  if (e == hipSuccess) e = gc_buf_reserve(ctx->nav[2], sizeof(float) * (size_t)n, false);
  if (e != hipSuccess) {
    gc_set_error("gc_sync_xcorr: %s", hipGetErrorString(e));
    return GC_E_NOMEM;
  }
  double* d_ip = (double*)ctx->nav[0].p;
  int8_t* d_pat = (int8_t*)ctx->nav[1].p;
  float* d_out = (float*)ctx->nav[2].p;
  e = hipMemcpyAsync(d_ip, i_p, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pat, pattern, (size_t)m, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const size_t smem = sizeof(float) * (size_t)(kLagsPerBlock + 2 * m);
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)sync_xcorr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(sync_xcorr_kernel, dim3((unsigned int)((n + kLagsPerBlock - 1) / kLagsPerBlock)), dim3(256), smem, ctx->stream,
                       (const double*)d_ip, (long long)n, (const int8_t*)d_pat, m, (flags & GC_SYNC_ZERO_IS_PLUS) ? 1 : 0, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    gc_set_error("gc_sync_xcorr: %s", hipGetErrorString(e));
    return GC_E_HIP;
  }
  return GC_OK;
}

extern "C" int gc_preamble_xcorr(gc_context* ctx, const double* i_p, int64_t n, const int8_t* pattern, int m, float* out) {
  return gc_sync_xcorr(ctx, i_p, n, pattern, m, 0, out);
}
