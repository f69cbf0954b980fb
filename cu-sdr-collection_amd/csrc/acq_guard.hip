// acq_guard.hip - single cells of an acquisition search in float64 (see acq_guard.h for what the guard is and why).
// Reference arithmetic restated: GPS/GPS_L1CA/include/acquisition.m:122 (phasePoints), :172 (sigCarr), :177-190 (hop blocks, product
// with the conjugated code spectrum, abs(ifft(.)) added per hop); BDS/B1I/include/acquisition.m:98-119 (bins as circshift).
#include "acq_guard.h"

#include <hip/hip_runtime.h>

#include "gc_internal.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

// exp(-1i * f * phasePoints(m)) * exp(+2i*pi * shift * m / blk), phasePoints(m) = ((m * 2) * pi) * ts as MATLAB evaluates
// (0 : n - 1) * 2 * pi * ts, left to right (acquisition.m:122)
__device__ __forceinline__ void seed_phasor(double f, double ts, int shift, int blk, int m, double& pr, double& pi) {
  const double arg = f * ((((double)m * 2.0) * kPi) * ts);
  double sn, cs;
  sincos(arg, &sn, &cs);
  pr = cs;
  pi = -sn;
  if (shift != 0) {
    const long long r = ((long long)shift * (long long)m) % (long long)blk;  // whole turns taken out exactly
    double s2, c2;
    sincos(2.0 * kPi * (double)r / (double)blk, &s2, &c2);
    const double tr = pr * c2 - pi * s2, ti = pr * s2 + pi * c2;
    pr = tr;
    pi = ti;
  }
}

// One workgroup per (cell, hop): thread t takes a run of consecutive n; the phasor is advanced by one complex multiplication per
// sample and re-seeded from the exact expression every 16 samples and where (n + tau) wraps to the block's start.
__global__ __launch_bounds__(256) void acq_exact_kernel(const GcExactSetup s, const GcExactCell* __restrict__ cells, double* __restrict__ partial) {
  const GcExactCell c = cells[blockIdx.x];
  const int hop = blockIdx.y, tid = threadIdx.x;
  const long long base = c.first + (long long)hop * s.hop_stride;
  const int per = (s.cl + 255) / 256;
  const int n0 = min(s.cl, tid * per), n1 = min(s.cl, n0 + per);
  int shift = c.shift % s.blk;
  if (shift < 0) shift += s.blk;
  const double ts = 1.0 / s.fs;
  double str, sti;  // the per-sample step: exp(-1i * f * 2*pi*ts) * exp(+2i*pi * shift / blk)
  seed_phasor(c.freq, ts, shift, s.blk, 1, str, sti);
  double sr[4] = {0.0, 0.0, 0.0, 0.0}, si[4] = {0.0, 0.0, 0.0, 0.0};
  double pr = 1.0, pi = 0.0;
  int m = (int)(((long long)n0 + (long long)c.col) % (long long)s.blk);
  for (int n = n0; n < n1; ++n) {
    if (((n - n0) & 15) == 0 || m == 0) seed_phasor(c.freq, ts, shift, s.blk, m, pr, pi);
    double xr, xi;
    if (s.if_f32) {
      const float2 v = s.if_f32[base + m];
      xr = (double)v.x;
      xi = (double)v.y;
    } else {
      const int8_t* p = s.if_i8 + 2 * (base + m);
      xr = (double)p[0];
      xi = (double)p[1];
    }
    const double zr = xr * pr - xi * pi, zi = xr * pi + xi * pr;
    for (int arm = 0; arm < s.narms; ++arm) {
      const double cd = (double)s.codes[((long long)c.code * s.narms + arm) * s.code_stride + n];
      sr[arm] += zr * cd;
      si[arm] += zi * cd;
    }
    const double tr = pr * str - pi * sti, ti = pr * sti + pi * str;
    pr = tr;
    pi = ti;
    if (++m == s.blk) m = 0;
  }
  __shared__ double red[8][256];
  for (int arm = 0; arm < 4; ++arm) {
    red[2 * arm][tid] = sr[arm];
    red[2 * arm + 1][tid] = si[arm];
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {  // fixed order: bit-reproducible
    if (tid < off)
      for (int k = 0; k < 2 * s.narms; ++k) red[k][tid] += red[k][tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    double v = 0.0;
    for (int arm = 0; arm < s.narms; ++arm) v += s.w[arm] * hypot(red[2 * arm][0], red[2 * arm + 1][0]);
    partial[(long long)blockIdx.x * s.nhops + hop] = v;
  }
}

__global__ void cells_from_keys_kernel(const unsigned long long* __restrict__ keys, int nprn, double f0, double fstep, const double* __restrict__ off,
                                       long long first, GcExactCell* __restrict__ cells) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;
  if (ip >= nprn) return;
  const unsigned int nb = (unsigned int)(keys[2 * ip] & 0xffffffffull), nc = (unsigned int)(keys[2 * ip + 1] & 0xffffffffull);
  // a PRN whose search saw nothing (keys 0) decodes to bin / col 0xffffffff: cell (0, 0), as the host side reports it
  const int bin = keys[2 * ip] ? (int)(0xffffffffu - nb) : 0, col = keys[2 * ip + 1] ? (int)(0xffffffffu - nc) : 0;
  GcExactCell c;
  c.code = ip;
  c.col = col;
  c.shift = 0;
  c.bin = bin;
  c.freq = f0 + (off ? off[ip] : 0.0) - fstep * (double)bin;
  c.first = first;
  cells[ip] = c;
}

__global__ __launch_bounds__(256) void collect_cells_kernel(const float* __restrict__ r, int rows, long long row_stride, int valid, float thr,
                                                            int* __restrict__ count, int2* __restrict__ list, int cap) {
  for (int row = blockIdx.y; row < rows; row += gridDim.y)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < valid; c += gridDim.x * blockDim.x)
      if (r[(long long)row * row_stride + c] >= thr) {
        const int k = atomicAdd(count, 1);
        if (k < cap) list[k] = make_int2(row, c);
      }
}

}  // namespace

double gc_acq_tie_eps(int n) {
  if (const char* e = GC_TUNE_ENV("GC_ACQ_GUARD_EPS")) {
    const double v = std::atof(e);
    if (v > 0.0 && v < 0.5) return v;
  }
  double l2 = 1.0;
  for (long long m = 2; m < n; m *= 2) l2 += 1.0;
  return 8.0 * l2 / 16777216.0;
}

int gc_exact_cells_from_keys(hipStream_t stream, const unsigned long long* keys, int nprn, double f0, double fstep, const double* d_off,
                             long long first, GcExactCell* d_cells) {
  if (nprn <= 0) return GC_OK;
  hipLaunchKernelGGL(cells_from_keys_kernel, dim3((unsigned int)((nprn + 63) / 64)), dim3(64), 0, stream, keys, nprn, f0, fstep, d_off, first, d_cells);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gc_exact_cells(hipStream_t stream, const GcExactSetup& s, const GcExactCell* d_cells, int ncells, double* d_partial) {
  if (ncells <= 0) return GC_OK;
  if (s.blk <= 0 || s.cl <= 0 || s.cl > s.blk || s.nhops < 1 || s.narms < 1 || s.narms > 4 || !s.codes || (!s.if_i8 && !s.if_f32) || !(s.fs > 0.0)) {
    gc_set_error("acquisition guard: bad cell set-up");
    return GC_E_INVALID;
  }
  hipLaunchKernelGGL(acq_exact_kernel, dim3((unsigned int)ncells, (unsigned int)s.nhops), dim3(256), 0, stream, s, d_cells, d_partial);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gc_collect_cells(hipStream_t stream, const float* r, int rows, long long row_stride, int valid, float thr, int* d_count, int2* d_list, int cap) {
  if (rows <= 0 || valid <= 0) return GC_OK;
  const dim3 grid((unsigned int)std::max(1, std::min((valid + 1023) / 1024, 64)), (unsigned int)std::min(rows, 65535));
  hipLaunchKernelGGL(collect_cells_kernel, grid, dim3(256), 0, stream, r, rows, row_stride, valid, thr, d_count, d_list, cap);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
