// acq_coarse.hip - the carrier-per-bin searches (GPS L1 C/A, L5, Galileo E1 / E5a / E5b, BDS B2a / B3I, GLONASS): gc_acquire_coarse*,
// the peak reductions of their last pass, the search scratch, the float64 guard's slow path.
// Reference: GPS/GPS_L1CA/include/acquisition.m:151-206 (sigPower, the PRN x bin x hop loop, max(max(results)), peakMetric);
// GPS/GPS_L5C/include/acquisition.m:175-216 (data + pilot arms); GLO/GLO_GL1/include/acquisition.m:146-147 (one search per frequency number).
// Split out of acq.hip in round 6; shared declarations: acq_internal.h.
#include "acq_internal.h"

using namespace gcacq;

namespace {
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

// One workgroup per row: maximum and its first position (MATLAB's max returns the first maximum).
// Row maxima of a circshift search from the per-workgroup candidates its last pass left (PeakTrack::publish_slot: every workgroup of
// that pass belongs to ONE row; key = value bits << 32 | ~column, so the largest key is the row's maximum at its first column):
// one wave per row.  Replaces writing rows x N sums and reading them back (GPS L2C: 1 GB each way per PRN).
__global__ __launch_bounds__(64) void rowkeys_reduce_kernel(const unsigned long long* __restrict__ slots, int tiles, float* vmax, int* amax,
                                                            const unsigned int* __restrict__ sec_slots = nullptr, float* vsecond = nullptr) {
  const unsigned long long* mine = slots + (size_t)blockIdx.x * tiles * 2;
  unsigned long long k = 0;
  for (int i = threadIdx.x; i < tiles; i += 64) k = max(k, mine[2 * i + 1]);
  for (int off = 32; off > 0; off >>= 1) k = max(k, (unsigned long long)__shfl_xor((long long)k, off, 64));
  if (threadIdx.x == 0) {
    vmax[blockIdx.x] = __uint_as_float((unsigned int)(k >> 32));
    amax[blockIdx.x] = (int)(0xffffffffu - (unsigned int)(k & 0xffffffffu));
  }
  if (!sec_slots || !vsecond) return;
  // the row's runner-up for the float64 guard: every tile's own second, every tile's maximum except ONE holder of the row's
  const unsigned int m1 = (unsigned int)(k >> 32);
  const unsigned int* sec = sec_slots + (size_t)blockIdx.x * tiles;
  unsigned int w2 = 0, holders = 0;
  for (int i = threadIdx.x; i < tiles; i += 64) {
    const unsigned int mi = (unsigned int)(mine[2 * i + 1] >> 32);
    w2 = max(w2, sec[i]);
    if (mi == m1) ++holders;
    else w2 = max(w2, mi);
  }
  for (int off = 32; off > 0; off >>= 1) {
    w2 = max(w2, (unsigned int)__shfl_xor((int)w2, off, 64));
    holders += (unsigned int)__shfl_xor((int)holders, off, 64);
  }
  if (threadIdx.x == 0) vsecond[blockIdx.x] = __uint_as_float(holders > 1 ? m1 : w2);
}
}  // namespace
namespace gcacq {
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
                   &s->b_list, &s->b_off, &s->b_rowsec})
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

int acq_read_back(gc_context* ctx, AcqScratch* s, const AcqBack* parts, int nparts) {
  size_t total = 0;
  for (int i = 0; i < nparts; ++i) total += (parts[i].bytes + 15) / 16 * 16;
  const bool pageable = GC_TUNE_ENV("GC_ACQ_SHIFT_PAGEABLE") != nullptr;
  if (!pageable && s->pinned_bytes < total) {
    if (s->pinned) (void)hipHostFree(s->pinned);
    s->pinned = nullptr;
    s->pinned_bytes = 0;
    const size_t want = std::max(total, (size_t)1 << 21);
    if (hipHostMalloc(&s->pinned, want, hipHostMallocDefault) == hipSuccess) s->pinned_bytes = want;
    else (void)hipGetLastError();
  }
  if (!pageable && s->pinned_bytes >= total) {
    char* h = static_cast<char*>(s->pinned);
    size_t at = 0;
    for (int i = 0; i < nparts; ++i) {
      if (parts[i].bytes) GC_HIP(hipMemcpyAsync(h + at, parts[i].src, parts[i].bytes, hipMemcpyDeviceToHost, ctx->stream));
      at += (parts[i].bytes + 15) / 16 * 16;
    }
    GC_HIP(hipStreamSynchronize(ctx->stream));
    at = 0;
    for (int i = 0; i < nparts; ++i) {
      if (parts[i].bytes) std::memcpy(parts[i].dst, h + at, parts[i].bytes);
      at += (parts[i].bytes + 15) / 16 * 16;
    }
    return GC_OK;
  }
  for (int i = 0; i < nparts; ++i)
    if (parts[i].bytes) GC_HIP(hipMemcpyAsync(parts[i].dst, parts[i].src, parts[i].bytes, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

// Last inverse pass (POST_ABS_ACC) over `nbins` bins.  With few bins the launch would have ~2 workgroups per CU, each
// walking all nhops hops of its bin: the hops are then split over hop groups (a divisor of nhops), whose raw sums meet in
// abs_combine_kernel - deterministic, group order fixed.
// bin0 / nbins_total: the launch covers bins bin0 .. bin0 + nbins - 1 of a search of nbins_total (a PRN's bins in chunks, see
// gc_acquire_coarse_multi): no hop groups then, and the chunk's candidates go to their bins' places in the PRN's slot region.
int launch_abs_pass(gc_context* ctx, AcqScratch* s, PassArgs& a, long long nbins, unsigned long long* keys, int valid, int ip, int nprn, bool* rows_fused,
                    int bin0, long long nbins_total) {
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
      const size_t lane_off = (size_t)s->lane * (size_t)nbins_total * tiles_ct * (s->shift_slot_lanes > 1 ? 1 : 0);
      unsigned long long* const region = s->slots + lane_off * 2;
      unsigned int* const sec_region = s->sec_slots + lane_off;
      a.peak_slots = region + (size_t)bin0 * tiles_ct * 2;
      a.peak_second = s->rowsecond ? sec_region + (size_t)bin0 * tiles_ct : nullptr;
      a.peak_valid = valid;
      a.batch0 = bin0;
      bool used_ct = false;
      int rc = launch_pass(ctx, a, nbins, &used_ct);
      a.peak_slots = nullptr;
      a.peak_second = nullptr;
      a.batch0 = 0;
      if (rc) return rc;
      if (used_ct) {
        hipLaunchKernelGGL(rowkeys_reduce_kernel, dim3((unsigned int)nbins), dim3(64), 0, ctx->stream, region + (size_t)bin0 * tiles_ct * 2, tiles_ct,
                           s->rowmax + bin0, s->rowarg + bin0, s->rowsecond ? sec_region + (size_t)bin0 * tiles_ct : nullptr,
                           s->rowsecond ? s->rowsecond + bin0 : nullptr);
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
}  // namespace gcacq

void gc_acq_free(gc_context* ctx) {
  free_scratch((AcqScratch*)ctx->acq_scratch);
  ctx->acq_scratch = nullptr;
}
namespace gcacq {
int ensure_scratch(gc_context* ctx, int n, long long nbh, int nprn, int nbins, int spc, AcqScratch** out) {
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
}  // namespace gcacq

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
      fused = launch_fused(ctx, pl, fa, nprn);
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
  {
    const AcqBack back[3] = {{hpeaks.data(), peaks, hpeaks.size() * sizeof(unsigned long long)},
                             {hsec.data(), seconds, guard ? hsec.size() * sizeof(unsigned int) : 0},
                             {hexact.data(), d_exact, guard ? hexact.size() * sizeof(double) : 0}};
    rc = acq_read_back(ctx, s, back, 3);  // keys, runner-ups and the winners' float64 values: one pinned staging buffer, one synchronisation
    if (rc) return rc;
  }
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
