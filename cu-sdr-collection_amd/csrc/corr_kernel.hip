// corr_kernel.hip — correlator launch logic (kernel choice, work decomposition), the exact per-sample kernel
// for channels with mixed ramp multipliers, and the partial-sum combiner.  The two production kernels live in
// corr_fast.hip (low chipping rates: at most one table transition per lane-chunk) and corr_lane.hip (any rate).
#include <cstdlib>

#include "corr_common.h"

using namespace gcorr;

namespace {

// ---- exact reference kernel for channels whose arms use DIFFERENT ramp multipliers ---------------------
// (BDS B1C wide-band: data BOC(1,1), pilot BOC(1,1) and pilot BOC(6,1) read through ceil(6*t),
// BDS/B1C/include/WB_tracking.m:285-317; the 122 762-entry BOC(6,1) table does not fit LDS next to the
// other two).  One thread per sample (grid-stride), every index from the reference's float64 colon element
// rule, tables read through L2: simple and exact rather than fast — this is one signal of twelve.
template <int MODE>
__global__ __launch_bounds__(kWG) void corr_epl_mixed_kernel(const KArgs p) {
  __shared__ double red[kWG / 64][GC_OUT_STRIDE];
  const long long lb = blockIdx.x / p.splits;
  const int split = (int)(blockIdx.x - lb * p.splits);
  const gc_block blk = p.blocks[lb];
  const DevChannel* __restrict__ chn = p.chans + blk.channel;
  const int arms = chn->arms;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double R = chn->index_scale, rem = blk.rem_code_phase, step = blk.code_phase_step, d = blk.el_spacing;
  const int N = blk.blksize;
  const double sp = step * R;
  const double a3[3] = {(rem - d) * R, rem * R, (rem + d) * R};
  const double nm1s = __dmul_rn((double)(N - 1), step);
  const double b3[3] = {__dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), -d), R), __dmul_rn(__dadd_rn(nm1s, rem), R),
                        __dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), d), R)};
  const double tau = blk.carr_freq / p.fs;
  const double ph0 = blk.rem_carr_phase * 0.15915494309189535;
  const int per = (N + p.splits - 1) / p.splits;
  const int i_beg = split * per, i_end = min(N, i_beg + per);
  float acc[GC_OUT_STRIDE];
#pragma unroll
  for (int v = 0; v < GC_OUT_STRIDE; ++v) acc[v] = 0.f;
  constexpr int bps = (MODE == I8_IQ || MODE == I8_QI || MODE == I16_REAL) ? 2 : (MODE == I8_REAL) ? 1 : 4;
  for (int i = i_beg + tid; i < i_end; i += kWG) {
    const uint8_t* s = p.if_base + (size_t)(blk.first_sample + i) * bps;
    float a, b;
    if (MODE == I8_IQ || MODE == I8_QI) {
      a = (float)(signed char)s[0];
      b = (float)(signed char)s[1];
    } else if (MODE == I16_IQ || MODE == I16_QI) {
      a = (float)((const short*)s)[0];
      b = (float)((const short*)s)[1];
    } else if (MODE == I8_REAL) {
      a = (float)(signed char)s[0];
      b = 0.f;
    } else {
      a = (float)((const short*)s)[0];
      b = 0.f;
    }
    if (MODE == I8_QI || MODE == I16_QI) {
      const float t = a;
      a = b;
      b = t;
    }
    const double ph = ph0 + (double)i * tau;
    float sn, cs;
    sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
    const float xr = a * cs + b * sn, xi = b * cs - a * sn;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double t;
      if (2 * i < N - 1)
        t = __dadd_rn(a3[x], __dmul_rn((double)i, sp));
      else if (2 * i > N - 1)
        t = __dadd_rn(b3[x], -__dmul_rn((double)(N - 1 - i), sp));
      else
        t = __dadd_rn(a3[x], b3[x]) / 2.0;
      for (int ar = 0; ar < arms; ++ar) {
        const int k = (int)ceil(__dmul_rn(t, chn->mult[ar])) + blk.table_offset[ar];
        const float c = (float)chn->tab[ar][min(max(k, 0), chn->nent[ar] - 1)];
        acc[ar * 6 + 2 * x] = fmaf(c, xr, acc[ar * 6 + 2 * x]);
        acc[ar * 6 + 2 * x + 1] = fmaf(c, xi, acc[ar * 6 + 2 * x + 1]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < GC_OUT_STRIDE; ++v) {
    float x = acc[v];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) red[wave][v] = (double)x;
  }
  __syncthreads();
  if (tid < GC_OUT_STRIDE) {
    double sum = 0.0;
    for (int w = 0; w < kWG / 64; ++w) sum += red[w][tid];
    if (p.splits == 1)
      p.out[lb * GC_OUT_STRIDE + tid] = sum;
    else
      p.partial[(lb * p.splits + split) * GC_OUT_STRIDE + tid] = sum;
  }
}

__global__ void combine_partials_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                        long long nblocks, int splits) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks * GC_OUT_STRIDE) return;
  const long long lb = i / GC_OUT_STRIDE;
  const int v = (int)(i - lb * GC_OUT_STRIDE);
  double s = 0.0;
  for (int k = 0; k < splits; ++k) s += partial[(lb * splits + k) * GC_OUT_STRIDE + v];
  out[i] = s;
}

}  // namespace

int gc_launch_correlator(gc_context* ctx, const gc_block* d_blocks, int64_t nblocks, int splits,
                         double* d_out, double* d_partial, int max_arms, int fast, int period, unsigned int notify_tag,
                         bool share_el) {
  if (nblocks <= 0) return GC_OK;
  KArgs a;
  a.if_base = ctx->d_if;
  a.blocks = d_blocks;
  a.chans = ctx->d_channels;
  a.out = d_out;
  a.partial = d_partial;
  a.fs = ctx->fs;
  a.inv_fs = 1.0 / ctx->fs;
  a.nblocks = nblocks;
  a.splits = splits;
  a.red_off = ctx->max_lds_bytes;
  a.bpw = 1;
  a.stride = 1;
  InlineBlocks ib;
  a.tagged = nullptr;
  a.notify_tag = 0;
  a.use_inline = 0;
  a.share_el = share_el ? 1 : 0;
  a.wide = 0;
  a.derived = (fast == 0 && ctx->launch_derived) ? 1 : 0;
  a.devloop = nullptr;
  if (fast >= 0 && notify_tag != 0 && ctx->h_tagged_pinned) {
    // closed loop: d_blocks is the host-mapped descriptor buffer (readable by the host right here)
    a.tagged = reinterpret_cast<TaggedSlot*>(ctx->h_tagged_pinned);
    a.notify_tag = notify_tag;
    if (nblocks <= kInlineBlocks) {
      a.use_inline = 1;
      for (int64_t i = 0; i < nblocks; ++i) ib.b[i] = d_blocks[i];
    }
  }
  long long total = (long long)nblocks * splits;
  int want_bpw = 8;
  if (const char* e = GC_TUNE_ENV("GC_REPLAY_BPW")) want_bpw = std::max(1, std::atoi(e));
  // Hybrid kernel for channels with a derived six-fold arm (corr_cboc.hip): periodic replay lists of int8 I/Q records, all channels
  // derived, base ramp with <= 2 transitions per 16-sample chunk.  Round 5's version (all four running-sum streams parked side by side:
  // 4 - 8 waves per CU) measured slower than the lane kernel's derived-arm instantiation; round 6's phased parking at sixteen waves per CU
  // is ahead of it (config 3's shape over 20 s: 2.76 ms against 2.91 - DESIGN.md 4.2c), so it takes these lists.  GC_NO_CBOC=1 (tuning
  // build): the lane kernel as before.
  if (fast == 0 && a.derived && gc_cboc_takes(ctx, nblocks, period) && splits == 1 && notify_tag == 0 && max_arms == 3 && !ctx->force_generic &&
      !GC_TUNE_ENV("GC_NO_CBOC")) {
    const int cwaves = gc_cboc_waves(ctx);
    {
      a.bpw = cwaves * (nblocks >= 64LL * cwaves * ctx->compute_units ? 2 : 1);  // a staged table serves bpw epochs of its channel
      a.stride = period;
      a.wide = 1;
      total = ((nblocks + (long long)a.bpw * period - 1) / ((long long)a.bpw * period)) * period;
      a.xcd_swizzle = 0;
      a.total_wg = 0;
      if (total >= 64) {
        a.xcd_swizzle = 1;
        a.total_wg = total;
        total = (total + 7) / 8 * 8;
      }
      if (total > 0x7fffffffLL) {
        gc_set_error("too many workgroups (%lld)", total);
        return GC_E_INVALID;
      }
      ctx->last_kernel = 5;
      return gc_launch_correlator_cboc(ctx, a, (unsigned int)total, cwaves);
    }
  }
  // Multi-transition kernel (corr_multi.hip): big periodic replay lists whose chunks of 16 samples see up to 2 or 4 table
  // transitions - lists the single-transition kernel takes with 8-sample chunks (fast == 1) or hands to the lane kernel
  // (fast == 0).  GC_NO_MULTI=1 keeps the old choice (A/B), GC_MULTI_MIN = epochs per CU from which it is taken.
  {
    const int multi_min = GC_TUNE_ENV("GC_MULTI_MIN") ? std::max(1, std::atoi(GC_TUNE_ENV("GC_MULTI_MIN"))) : 4;
    const int mwaves = (period > 0 && max_arms <= 2) ? gc_multi_waves(ctx, max_arms, nblocks, period, ctx->scope_kt, ctx->scope_share_lane) : 0;
    const bool multi = (fast == 0 || fast == 1) && ctx->scope_kt >= 2 && period > 0 && splits == 1 && notify_tag == 0 && !a.derived &&
                       ctx->if_layout != GC_REAL && max_arms <= 2 && mwaves > 0 &&
                       // enough work to fill the device: epochs per CU, a block counted by its length in 16 384-sample units (two BDS B1C
                       // channels x 10 s are 2 000 blocks of 180 000 samples)
                       nblocks * std::max<long long>(1, ctx->replay_min_blksize / 16384) >= multi_min * (long long)period * ctx->compute_units &&
                       !GC_TUNE_ENV("GC_NO_MULTI") && !ctx->force_generic;
    if (multi) {
      // blocks per workgroup: a table staged once serves bpw epochs of its channel, but a short list cut into few workgroups ends in
      // a long tail (three Galileo E1 channels x 10 s: 940 workgroups of 8 blocks 0.450 ms, 1 875 of 4 blocks 0.406 ms)
      const int bpw4 = GC_TUNE_ENV("GC_REPLAY_BPW") ? std::max(4, want_bpw) / 4 * 4 : (nblocks / 8 >= 6LL * ctx->compute_units ? 8 : 4);
      a.bpw = mwaves >= 8 ? mwaves * (nblocks / period >= 64LL * mwaves ? 2 : 1) : bpw4;
      a.stride = period;
      a.wide = 1;
      total = ((nblocks + (long long)a.bpw * period - 1) / ((long long)a.bpw * period)) * period;
      a.xcd_swizzle = 0;
      a.total_wg = 0;
      if (total >= 64) {
        a.xcd_swizzle = 1;
        a.total_wg = total;
        total = (total + 7) / 8 * 8;
      }
      if (total > 0x7fffffffLL) {
        gc_set_error("too many workgroups (%lld)", total);
        return GC_E_INVALID;
      }
      ctx->last_kernel = 4;
      return gc_launch_correlator_multi(ctx, a, (unsigned int)total, max_arms, ctx->scope_kt, ctx->scope_share_lane, mwaves);
    }
  }
  const bool must_wide = fast > 0 && gc_fast_table_mode(ctx) == 1;  // tables too large for single-wave workgroups
  // by choice: every wave of the fast kernel parks 4-8 KB of running sums in LDS (corr_fast.hip), and
  // only four waves sharing an int8-pair table keep 16 waves per CU resident (big periodic replay lists, int8 I/Q, <= 2 arms)
  // (measured, scripts/replay_scaling.py: the four-wave float-table kernel wins from 4 epochs per CU on - 12 channels x 2 s: 0.70 of
  // the HBM figure against 0.52 with single-wave workgroups, 3 channels x 10 s: 0.55 against 0.39; the first version waited for 64)
  static const int wide_min = GC_TUNE_ENV("GC_WIDE_MIN") ? std::max(1, std::atoi(GC_TUNE_ENV("GC_WIDE_MIN"))) : 4;
  const bool big_list0 = nblocks >= wide_min * (long long)period * ctx->compute_units;
  const bool choose_wide = fast > 0 && !must_wide && gc_fast_prefers_wide() && period > 0 && splits == 1 && big_list0 && notify_tag == 0 &&
                           ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL && max_arms <= 2 && 2 * ctx->max_lds_bytes + 512 <= 40 * 1024;
  const bool wide_tables = must_wide || choose_wide;
  const bool big_list = choose_wide || nblocks >= 64 * (long long)period * ctx->compute_units;  // tables that MUST be shared: 8 epochs per workgroup only for long lists
  if (fast == 0) {
    // lane kernel: 16 wavefronts per workgroup, one (block, split) item each
    if (splits == 1 && period > 0) {
      // periodic list (all table offsets zero): a workgroup stages its channel's tables once and its waves
      // walk consecutive epochs of that channel
      a.bpw = (nblocks / period >= 256) ? 2 * kLaneWaves : kLaneWaves;
      if (GC_TUNE_ENV("GC_REPLAY_BPW")) a.bpw = std::max(kLaneWaves, want_bpw / kLaneWaves * kLaneWaves);
      a.stride = period;
      total = ((nblocks + (long long)a.bpw * period - 1) / ((long long)a.bpw * period)) * period;
      if (total < 2LL * ctx->compute_units && nblocks > total && !GC_TUNE_ENV("GC_REPLAY_BPW")) {
        // few, long blocks (two BDS B1C channels, 10-ms epochs: 996 blocks would make 32 workgroups): one block per
        // workgroup, split over its 16 waves, fills the device; the table is staged per block instead of per 16-32 blocks
        a.bpw = 1;
        a.stride = 1;
        a.wide = 1;
        total = nblocks;
      }
    } else if (splits == 1) {
      a.wide = 1;  // one block per workgroup, split over its 16 waves in-kernel
      total = nblocks;
    } else {
      if (splits % kLaneWaves != 0) {
        gc_set_error("internal: lane correlator launch needs splits %% %d == 0 (got %d)", kLaneWaves, splits);
        return GC_E_INVALID;
      }
      total = (long long)nblocks * (splits / kLaneWaves);
    }
  } else if (want_bpw > 1 && fast > 0 && splits == 1 && period > 0 && (big_list || wide_tables)) {
    // periodic list (all table offsets zero): a workgroup stages its channel's table once and walks
    // several consecutive epochs of that channel — 8 for big lists; the WIDE variant needs at least one
    // block per wave, so 4 even for short lists
    a.bpw = big_list ? std::max(want_bpw, wide_tables ? 4 : 1) : 4;
    a.stride = period;
    total = ((nblocks + (long long)a.bpw * period - 1) / ((long long)a.bpw * period)) * period;
  }
  if (total > 0x7fffffffLL) {
    gc_set_error("too many workgroups (%lld)", total);
    return GC_E_INVALID;
  }
  dim3 grid((unsigned int)total);
  int rc;
  if (fast > 0 && wide_tables) {
    // WIDE fast kernel: four waves per workgroup, int8-pair tables (8-sample chunks and no early/late sharing unless
    // chosen for the prefix-sum variant, which is instantiated for both chunk sizes)
    a.wide = 1;
    if (!choose_wide) {
      fast = 1;
      a.share_el = 0;
    } else if (max_arms == 1 && a.bpw > 1 && !GC_TUNE_ENV("GC_NO_TABF") &&
               4 * (size_t)ctx->max_lds_bytes + 4 * (size_t)(fast == 2 ? 8192 : 4096) + 64 <= 40 * 1024) {
      a.wide = 2;  // small single-arm table: plain float code values, no conversions in the chunk loop; still 4 workgroups per CU
    }
    if (a.bpw == 1) {
      if (splits % 4 != 0 && splits != 1) {
        gc_set_error("internal: WIDE correlator launch needs splits %% 4 == 0 (got %d)", splits);
        return GC_E_INVALID;
      }
      if (splits == 1) {
        // unrelated blocks cannot share a staged table: the lane kernel takes such lists, one block per
        // workgroup
        a.wide = 1;
        fast = 0;
      } else {
        total = (total + 3) / 4;
        grid = dim3((unsigned int)total);
      }
    }
  }
  ctx->last_kernel = fast < 0 ? -1 : fast == 0 ? 0 : a.wide == 2 ? 3 : a.wide ? 2 : 1;
  if (fast < 0) {
    // mixed ramp multipliers: exact per-sample kernel
    int mode;
    if (ctx->if_dtype == GC_I8)
      mode = ctx->if_layout == GC_IQ ? I8_IQ : ctx->if_layout == GC_QI ? I8_QI : I8_REAL;
    else
      mode = ctx->if_layout == GC_IQ ? I16_IQ : ctx->if_layout == GC_QI ? I16_QI : I16_REAL;
    switch (mode) {
      case I8_IQ: hipLaunchKernelGGL((corr_epl_mixed_kernel<I8_IQ>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I8_QI: hipLaunchKernelGGL((corr_epl_mixed_kernel<I8_QI>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I16_IQ: hipLaunchKernelGGL((corr_epl_mixed_kernel<I16_IQ>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I16_QI: hipLaunchKernelGGL((corr_epl_mixed_kernel<I16_QI>), grid, dim3(kWG), 0, ctx->stream, a); break;
      case I8_REAL: hipLaunchKernelGGL((corr_epl_mixed_kernel<I8_REAL>), grid, dim3(kWG), 0, ctx->stream, a); break;
      default: hipLaunchKernelGGL((corr_epl_mixed_kernel<I16_REAL>), grid, dim3(kWG), 0, ctx->stream, a); break;
    }
    rc = (hipGetLastError() == hipSuccess) ? GC_OK : GC_E_HIP;
  }
  // XCD-aware order of the workgroups (corr_fast.hip / corr_lane.hip: workgroup b runs on XCD b % 8; every XCD gets one contiguous
  // range of the list, so that the channels of one epoch - neighbours in the list, readers of the same IF window - share an L2).
  // Any grid: rounded up to a multiple of 8, the kernels send the workgroups past `total` home.  (It used to need total % 8 == 0:
  // three channels x 20 s = 7 500 workgroups fetched the record three times, 2.13 GB per launch at 5.1 TB/s, HBM-bound.)
  a.xcd_swizzle = 0;
  a.total_wg = 0;
  if (fast >= 0 && total >= 64) {
    a.xcd_swizzle = 1;
    a.total_wg = total;
    total = (total + 7) / 8 * 8;
  }
  if (fast < 0) {
  } else if (fast) {
    a.red_off = (a.wide == 2 ? 4 : a.wide ? 2 : 8) * ctx->max_lds_bytes;  // float2 {c, dc} tables: 8 bytes per staged entry (int8 pairs: 2, plain floats: 4)
    rc = gc_launch_correlator_fast(ctx, a, ib, (unsigned int)total, max_arms, fast == 2);
  } else {
    rc = gc_launch_correlator_lane(ctx, a, ib, (unsigned int)total, max_arms, ctx->scope_share_lane);
  }
  if (rc != GC_OK) return rc;
  if (splits > 1 && d_out != nullptr) {
    const long long n = nblocks * GC_OUT_STRIDE;
    hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_partial, d_out, (long long)nblocks, splits);
    GC_HIP(hipGetLastError());
  }
  return GC_OK;
}
