// track.hip — gc_track: the reference's tracking loop (GPS/GPS_L1CA/include/tracking.m:133-368
// and the 3-state / pilot variants GPS_L5C/include/tracking.m:255-382,
// GAL_E1C/include/tracking.m:236-348) with the correlator (lines 247-300) on the GPU and the
// discriminators + loop filters (lines 302-335) on the host, as BASELINE.json's north_star
// prescribes.  All channels advance in lock step: one correlator launch per epoch covers every
// active channel; descriptors and partial sums live in host-mapped pinned memory so an epoch
// costs one kernel launch and one stream synchronisation.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <thread>

#include "corr_common.h"
#include "devloop.h"

namespace {

struct ChanState {
  bool active = false;
  int64_t pos = 0;
  double code_freq = 0, code_freq_basis = 0, rem_code = 0;
  double carr_freq = 0, carr_basis = 0, rem_carr = 0;
  double old_code_nco = 0, old_code_err = 0;
  double old_carr_nco = 0, old_carr_err = 0;  // 2nd-order PLL
  double d2_carr_err = 0, d_carr_err = 0;      // 3-state PLL
  int epochs = 0;
  bool aborted = false;
  int table_phase = 0;  // GPS L2C CLCodePhase
};

// Common/calcLoopCoef.m:41-45
void calc_loop_coef(double lbw, double zeta, double k, double* tau1, double* tau2) {
  const double wn = lbw * 8 * zeta / (4 * zeta * zeta + 1);
  *tau1 = k / (wn * wn);
  *tau2 = 2.0 * zeta / wn;
}

const double kPi = 3.141592653589793;  // MATLAB pi

// What gc_correlate checks per descriptor (validate_blocks, gnsscorr.hip), for the blocks a tracking loop will cut from one
// channel: every ramp stays inside the reference's [c(end) c c(1)] padding (tracking.m:158,252-270) for ANY code step, because
// (blksize-1)*step + rem < codeLength by construction of blksize (:222): the largest index is ceil((codeLength + spacing)*R*M).
int validate_track_channel(const gc_context* ctx, const gc_track_params* p, const gc_channel_init& in, const char* who) {
  const HostChannel& c = ctx->ch[in.channel];
  if (!(p->sampling_freq > 0) || !(p->code_length > 0) || !(p->el_spacing >= 0) || !(p->int_time > 0) || !(in.code_freq > 0) ||
      !std::isfinite(in.code_freq) || !std::isfinite(in.acquired_freq) || in.code_phase < 0) {
    gc_set_error("%s: channel %d: invalid parameters (samplingFreq %g, codeLength %g, spacing %g, codeFreq %g, acquiredFreq %g)", who,
                 in.channel, p->sampling_freq, p->code_length, p->el_spacing, in.code_freq, in.acquired_freq);
    return GC_E_INVALID;
  }
  for (int a = 0; a < c.arms; ++a) {
    const double rm = c.index_scale * c.mult[a];
    if (!(p->el_spacing * rm < 1.0)) {
      gc_set_error("%s: channel %d arm %d: dllCorrelatorSpacing * index scale * ramp multiplier = %g table entries; the early/late "
                   "ramps must stay within one entry of the prompt ramp ([c(end) c c(1)] has one pad entry either side)", who,
                   in.channel, a, p->el_spacing * rm);
      return GC_E_INVALID;
    }
    int offset_max = 0;
    if (a == 1 && p->table_phase_count > 0) {  // GPS_L2C tracking.m:261: window offset codeLength*(CLCodePhase-1), CLCodePhase <= count
      if (in.table_phase < 0 || in.table_phase > p->table_phase_count) {
        gc_set_error("%s: channel %d: table_phase %d outside 1..%d", who, in.channel, in.table_phase, p->table_phase_count);
        return GC_E_INVALID;
      }
      offset_max = (int)p->code_length * (p->table_phase_count - 1);
    }
    const int stage = (c.window[a] > 0) ? std::min(c.window[a], c.nent[a]) : c.nent[a];
    const int avail = std::min(stage, c.nent[a] - offset_max);
    const double top = std::ceil((p->code_length + p->el_spacing) * rm);
    if (top > avail - 1) {
      gc_set_error("%s: channel %d arm %d: the code ramps reach table index %g but the table (window) holds %d entries - "
                   "settings.codeLength (%g) x index scale x multiplier does not match the table set with gc_set_code", who,
                   in.channel, a, top, avail, p->code_length);
      return GC_E_INVALID;
    }
  }
  return GC_OK;
}

}  // namespace

extern "C" int gc_track(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init,
                        double* out, int32_t* epochs_done) {
  const int rc = gc_track_window(ctx, p, nch, init, out, epochs_done, nullptr);
  if (rc == GC_OK || rc == GC_E_RANGE) gc_fill_cno_host(ctx, p, nch, out, epochs_done);
  return rc;
}

extern "C" int gc_set_cno_output(gc_context* ctx, double* cno, int64_t capacity) {
  if (!ctx || capacity < 0 || (cno && capacity == 0)) {
    gc_set_error("gc_set_cno_output: bad arguments");
    return GC_E_INVALID;
  }
  ctx->cno_out = cno;
  ctx->cno_cap = cno ? capacity : 0;
  return GC_OK;
}

// Calc_CNo_PLD.m (BDS/B2a, BDS/B1C) over the recorded prompt sums with the per-interval bookkeeping of BDS/B2a/include/
// tracking.m:409-432: two-pass mean and N-1 variance like MATLAB's mean / var, sqrt of a negative number complex, abs() of
// the complex quotient.
static void fill_cno_pld(gc_context* ctx, const gc_track_params* p, int nch, const double* out, const int32_t* epochs_done) {
  const int K = p->cno_interval, n_epochs = p->n_epochs, nk = n_epochs / K;
  if ((long long)nch * nk * GC_CNO_NPLD > ctx->cno_cap) return;
  const double T = p->cno_acc_time;
  auto arm = [&](const double* I, const double* Q, double& lin, double& pld) {  // Calc_CNo_PLD.m:47-68
    double zm = 0.0;
    for (int e = 0; e < K; ++e) zm += I[e] * I[e] + Q[e] * Q[e];
    zm /= K;
    double zv = 0.0, wiped = 0.0, sq = 0.0;
    for (int e = 0; e < K; ++e) {
      const double dz = I[e] * I[e] + Q[e] * Q[e] - zm;
      zv += dz * dz;
      wiped += std::fabs(I[e]);  // sum(I_P(I_P>0)) - sum(I_P(I_P<0))
      sq += Q[e];
    }
    zv /= (K - 1);
    const double d = zm * zm - zv;
    double ratio;  // |Pav / (2*Nv)|, Nv = (Zm - Pav)/2
    if (d >= 0.0) {
      const double pav = std::sqrt(d);
      ratio = std::fabs(pav / (zm - pav));
    } else {
      ratio = std::sqrt(-d / (zm * zm - d));
    }
    lin = ratio / T;
    pld = (wiped * wiped - sq * sq) / (wiped * wiped + sq * sq);
  };
  for (int c = 0; c < nch; ++c) {
    const double* rec = out + (size_t)c * GC_TRK_NFIELDS * n_epochs;
    double prev[3] = {0.0, 0.0, 0.0};  // tempCNoValue, tracking.m:192,432
    for (int k = 0; k < nk; ++k) {
      double* o = ctx->cno_out + ((size_t)c * nk + k) * GC_CNO_NPLD;
      for (int j = 0; j < GC_CNO_NPLD; ++j) o[j] = 0.0;
      if ((k + 1) * K > epochs_done[c]) continue;
      const size_t e0 = (size_t)k * K;
      double data = 0.0, pilot = 0.0, cur[3] = {0.0, 0.0, 0.0};
      arm(rec + (size_t)GC_TRK_I_P * n_epochs + e0, rec + (size_t)GC_TRK_Q_P * n_epochs + e0, data, o[3]);
      cur[0] = 10.0 * std::log10(data);
      if (p->cno_mode != GC_CNO_PLD) {
        const double* pi = rec + (size_t)GC_TRK_PILOT_I_P * n_epochs + e0;
        const double* pq = rec + (size_t)GC_TRK_PILOT_Q_P * n_epochs + e0;
        if (p->cno_mode == GC_CNO_PLD_PILOT_SWAPPED)
          arm(pq, pi, pilot, o[4]);
        else
          arm(pi, pq, pilot, o[4]);
        cur[1] = 10.0 * std::log10(pilot);
      }
      cur[2] = 10.0 * std::log10(data + pilot);
      for (int j = 0; j < 3; ++j) {
        o[j] = cur[j] * 0.5 + prev[j] * 0.5;
        prev[j] = cur[j];
      }
    }
  }
}

// CNoVSM over the recorded data-arm prompt sums, interval by interval, exactly as tracking.m:351-358 calls it (two-pass mean and
// N-1 variance like MATLAB's; Common/CNoVSM.m:38-47)
void gc_fill_cno_host(gc_context* ctx, const gc_track_params* p, int nch, const double* out, const int32_t* epochs_done) {
  const int K = p->cno_interval;
  if (!ctx->cno_out || K <= 1) return;
  const int n_epochs = p->n_epochs, nk = n_epochs / K;
  if (p->cno_mode != GC_CNO_VSM) {
    fill_cno_pld(ctx, p, nch, out, epochs_done);
    return;
  }
  if ((long long)nch * nk > ctx->cno_cap) return;
  for (int c = 0; c < nch; ++c) {
    const double* ip = out + ((size_t)c * GC_TRK_NFIELDS + GC_TRK_I_P) * n_epochs;
    const double* qp = out + ((size_t)c * GC_TRK_NFIELDS + GC_TRK_Q_P) * n_epochs;
    for (int k = 0; k < nk; ++k) {
      double v = 0.0;
      if ((k + 1) * K <= epochs_done[c]) {
        double zm = 0.0;
        for (int e = k * K; e < (k + 1) * K; ++e) zm += ip[e] * ip[e] + qp[e] * qp[e];
        zm /= K;
        double zv = 0.0;
        for (int e = k * K; e < (k + 1) * K; ++e) {
          const double dz = ip[e] * ip[e] + qp[e] * qp[e] - zm;
          zv += dz * dz;
        }
        zv /= (K - 1);
        const double d = zm * zm - zv;
        double ratio;
        if (d >= 0.0) {
          const double pav = std::sqrt(d);
          ratio = std::fabs(pav / (zm - pav));
        } else {
          ratio = std::sqrt(-d / (zm * zm - d));
        }
        v = 10.0 * std::log10(ratio / p->cno_acc_time);
      }
      ctx->cno_out[(size_t)c * nk + k] = v;
    }
  }
}

// gc_track over one window of a record (GcTrackResume, gc_internal.h): channel state comes from / goes back to r->state,
// positions there count from the start of the RECORD (the IF buffer holds its samples from r->origin on), and with
// r->pause_at_end the call stops every channel, in lock step, at the first epoch whose block one of them cannot read from
// this window - more of the record follows - instead of ending that channel (tracking.m:241-245 is the END of the file).
int gc_track_window(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init, double* out,
                    int32_t* epochs_done, GcTrackResume* r) {
  if (r) r->paused = false;
  if (!ctx || !p || nch <= 0 || nch > GC_MAX_CHANNELS || !init || !out || !epochs_done || p->n_epochs <= 0 || (r && !r->state)) {
    gc_set_error("gc_track: bad arguments");
    return GC_E_INVALID;
  }
  if (!ctx->d_if) {
    gc_set_error("gc_track: no IF buffer loaded");
    return GC_E_STATE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  ctx->fs = p->sampling_freq;
  // channels that share the device during this call (gc_track_multi): teams are sized for all of them
  const int nch_dev = ctx->concurrent_jobs ? std::max(nch, ctx->concurrent_channels) : nch;
  int rc = gc_sync_channels(ctx);
  if (rc) return rc;

  int max_arms = 1;
  bool any_mixed = false, all_mixed_derived = true, any_three_plain = false;
  gc_scope_reset(ctx);
  for (int c = 0; c < nch; ++c) {
    const int ci = init[c].channel;
    if (ci < 0 || ci >= GC_MAX_CHANNELS || !ctx->ch[ci].configured) {
      gc_set_error("gc_track: channel %d not configured", ci);
      return GC_E_STATE;
    }
    for (int a = 0; a < ctx->ch[ci].arms; ++a)
      if (!ctx->ch[ci].d_tab[a]) {
        gc_set_error("gc_track: channel %d arm %d has no code table", ci, a);
        return GC_E_STATE;
      }
    if ((rc = validate_track_channel(ctx, p, init[c], "gc_track"))) return rc;
    max_arms = std::max(max_arms, ctx->ch[ci].arms);
    if (ctx->ch[ci].arms == 3 && !gc_channel_is_derived(ctx->ch[ci])) any_three_plain = true;
    gc_scope_add(ctx, ci);
    for (int a = 1; a < ctx->ch[ci].arms; ++a)
      if (ctx->ch[ci].mult[a] != ctx->ch[ci].mult[0]) {  // B1C wide-band / E1 CBOC: exact per-sample kernel, or the lane
        any_mixed = true;                                  // kernel's derived-arm instantiation when every such channel allows it
        if (!gc_channel_is_derived(ctx->ch[ci])) all_mixed_derived = false;
      }
  }
  if ((p->pilot_combine == 4 || p->pilot_combine == 5) && max_arms < 3) {
    gc_set_error("gc_track: pilot_combine 4 / 5 need three arms {data, pilot BOC(1,1), pilot BOC(6,1)}");
    return GC_E_INVALID;
  }
  if (p->pilot_combine != 0 && max_arms < 2) {
    gc_set_error("gc_track: pilot_combine requires a pilot arm");
    return GC_E_INVALID;
  }

  // splits: fill the device with one epoch's worth of blocks
  const int approx_chunks = (int)(p->code_length / (p->code_freq_basis / p->sampling_freq) / 8.0) + 1;
  // nominal kernel choice (per-epoch blocks are re-checked below): the fast kernel runs one
  // wavefront per workgroup, the generic one four
  int fast_nominal = (gc_fast_lds_ok(ctx) && !ctx->force_generic) ? 2 : 0;
  for (int c = 0; c < nch && fast_nominal; ++c) {
    gc_block probe;
    std::memset(&probe, 0, sizeof probe);
    probe.channel = init[c].channel;
    probe.code_phase_step = init[c].code_freq * 1.001 / p->sampling_freq;
    fast_nominal = std::min(fast_nominal, gc_block_lowrate_level(ctx, probe));
  }
  if (any_mixed && all_mixed_derived && !any_three_plain && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL)
    fast_nominal = 0;  // derived-arm channels run on the lane kernel
  int splits;
  if (fast_nominal) {
    const int chunks_nominal = approx_chunks * 8 / (fast_nominal == 2 ? 16 : 8);
    splits = (8 * ctx->compute_units + nch_dev - 1) / nch_dev;
    splits = std::max(1, std::min(std::min(splits, 32), std::max(1, chunks_nominal / (2 * 64))));
    if (gc_fast_table_mode(ctx) == 1) splits = std::max(4, std::min(32, (splits / 4) * 4));  // WIDE: 4 waves per workgroup
  } else {
    splits = gc_lane_splits(ctx, nch, approx_chunks * 8, 32);  // lane kernel: one wave per item, 16 items per workgroup
    if (splits == 1) splits = 16;  // the closed loop always goes through per-item records
  }
  if (const char* e = GC_TUNE_ENV("GC_TRACK_SPLITS")) splits = std::max(1, std::min(32, std::atoi(e)));

  // pinned, device-visible descriptor and result buffers
  if (ctx->pinned_cap_blocks < nch * 32) {
    if (ctx->h_blocks_pinned) (void)hipHostFree(ctx->h_blocks_pinned);
    if (ctx->h_out_pinned) (void)hipHostFree(ctx->h_out_pinned);
    ctx->h_blocks_pinned = nullptr;
    ctx->h_out_pinned = nullptr;
    ctx->pinned_cap_blocks = 0;
    GC_HIP(hipHostMalloc((void**)&ctx->h_blocks_pinned, sizeof(gc_block) * (size_t)nch, hipHostMallocMapped));
    GC_HIP(hipHostMalloc((void**)&ctx->h_out_pinned, sizeof(double) * (size_t)nch * 32 * GC_OUT_STRIDE, hipHostMallocMapped));
    ctx->pinned_cap_blocks = nch * 32;
  }
  // Results of the fast kernel arrive as host-mapped tagged 16-byte records (corr_common.h TaggedSlot): a
  // stream synchronise costs ~15-20 us of wake-up latency per epoch, polling the tags does not.
  const bool poll = GC_TUNE_ENV("GC_TRACK_NO_POLL") == nullptr;
  int poll_timeout_ms = 2000;  // per epoch; GC_TRACK_POLL_TIMEOUT_MS overrides
  if (const char* ev = std::getenv("GC_TRACK_POLL_TIMEOUT_MS")) poll_timeout_ms = std::max(1, std::atoi(ev));
  const int64_t need_slots = (int64_t)nch * 32 * GC_OUT_STRIDE;
  if (ctx->tagged_cap < need_slots) {
    GC_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_tagged_pinned) (void)hipHostFree(ctx->h_tagged_pinned);
    ctx->h_tagged_pinned = nullptr;
    ctx->tagged_cap = 0;
    GC_HIP(hipHostMalloc(&ctx->h_tagged_pinned, sizeof(gcorr::TaggedSlot) * (size_t)need_slots, hipHostMallocMapped));
    ctx->tagged_cap = need_slots;
  }
  GC_HIP(hipStreamSynchronize(ctx->stream));
  std::memset(ctx->h_tagged_pinned, 0, sizeof(gcorr::TaggedSlot) * (size_t)ctx->tagged_cap);
  volatile gcorr::TaggedSlot* tagged = (volatile gcorr::TaggedSlot*)ctx->h_tagged_pinned;
  gc_block* blocks = ctx->h_blocks_pinned;
  double* partial = ctx->h_out_pinned;

  const int n_epochs = p->n_epochs;
  std::fill(out, out + (size_t)nch * GC_TRK_NFIELDS * n_epochs, 0.0);

  // Persistent mode: the loop is still closed HERE (tracking.m:302-335 below), but nothing is launched per epoch - one
  // cooperative launch of the fast kernel's persistent instantiation polls the descriptors this loop writes into
  // host-mapped memory and answers with tagged records (devloop.h, host_loop).  Launch-per-epoch costs ~5 us of dispatch
  // latency and ~13 us of kernel wall time per epoch; the persistent kernel answers in a few microseconds.  Covered:
  // single-arm R = 1 channels on the transition-mask kernel with one-wave workgroups, any record format (GPS L1 C/A, BDS B1I,
  // GLONASS); everything else keeps launching.
  ctx->last_track_mode = 0;
  bool persist = poll && !any_mixed && max_arms == 1 && fast_nominal > 0 && gc_fast_table_mode(ctx) == 0 && p->pilot_combine == 0 &&
                 p->table_phase_count == 0 && n_epochs > 0 &&
                 !(GC_TUNE_ENV("GC_TRACK_PERSIST") && std::atoi(GC_TUNE_ENV("GC_TRACK_PERSIST")) == 0);
  const bool i8c_rec = ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL;  // int8 I/Q or Q/I (the derived-arm instantiation's only format)
  for (int c = 0; c < nch && persist; ++c) {
    const HostChannel& hcn = ctx->ch[init[c].channel];
    persist = hcn.arms == 1 && hcn.index_scale == 1.0 && hcn.mult[0] == 1.0 && hcn.window[0] == 0;
  }
  // ... and on the lane kernel's persistent instantiation (corr_lane.hip, host_loop) for one- and two-arm channels of any
  // rate and index scale (GPS L5, BDS B2a / B3I, Galileo E5a / E5b / E1 B+C, BDS B1C narrow-band): member 0's first wave
  // gathers the team's sums, hands them to the host and relays the host's next descriptor
  const bool derived_nominal = any_mixed && all_mixed_derived && !any_three_plain;  // three arms, the third derived (E1-C CBOC)
  bool persist_lane = !persist && poll && ((!any_mixed && max_arms <= 2) || (derived_nominal && i8c_rec)) && n_epochs > 0 &&
                      !(GC_TUNE_ENV("GC_TRACK_PERSIST") && std::atoi(GC_TUNE_ENV("GC_TRACK_PERSIST")) == 0);
  bool share_lane_nominal = true;
  for (int c = 0; c < nch && persist_lane; ++c) {
    gc_block probe;
    std::memset(&probe, 0, sizeof probe);
    probe.channel = init[c].channel;
    probe.el_spacing = p->el_spacing;
    share_lane_nominal = share_lane_nominal && gc_block_shares_el_lane(ctx, probe);
  }
  if (persist_lane) persist = true;
  gcorr::DevLoopArgs pa;
  std::memset(&pa, 0, sizeof pa);
  gcorr::DevLoopArgs* d_pargs = nullptr;
  gcorr::msg_t* h_desc = nullptr;  // host-mapped descriptor messages [nch][kDescWords]
  auto persist_free = [&]() {  // the buffers stay with the context (GcBuf: freeing would wait for every stream of the device)
    gc_persistent_done(ctx);  // the kernel has ended (or was never launched): its grid leaves the device's ledger
    pa.chan = nullptr;
    pa.desc_msg = nullptr;
    pa.part_msg = nullptr;
    d_pargs = nullptr;
    h_desc = nullptr;
  };
  // descriptor of team c for epoch e (tag e + 1): payload first, tag last - a device read that sees the tag sees the payload
  auto write_desc = [&](int c, int e, const gc_block* b, unsigned long long status_word) {
    unsigned long long q[gcorr::kDescWords] = {0};
    if (b) std::memcpy(q, b, sizeof(gc_block));
    q[gcorr::kDescWords - 1] = status_word;
    volatile unsigned int* w = reinterpret_cast<volatile unsigned int*>(h_desc + (size_t)c * gcorr::kDescWords);
    for (int i = 0; i < gcorr::kDescWords; ++i) {
      w[4 * i + 0] = (unsigned int)q[i];
      w[4 * i + 1] = (unsigned int)(q[i] >> 32);
      w[4 * i + 3] = 0u;
    }
    std::atomic_thread_fence(std::memory_order_release);
    for (int i = 0; i < gcorr::kDescWords; ++i) w[4 * i + 2] = (unsigned int)e + 1u;
    std::atomic_thread_fence(std::memory_order_release);
  };
  // every team stops whatever epoch it is waiting for (tag 0xffffffff is accepted for any epoch)
  auto persist_stop = [&]() {
    for (int c = 0; c < nch; ++c) {
      volatile unsigned int* w = reinterpret_cast<volatile unsigned int*>(h_desc + (size_t)c * gcorr::kDescWords);
      for (int i = 0; i < gcorr::kDescWords; ++i) {
        w[4 * i + 0] = 3u;
        w[4 * i + 1] = 0u;
      }
      std::atomic_thread_fence(std::memory_order_release);
      for (int i = 0; i < gcorr::kDescWords; ++i) w[4 * i + 2] = 0xffffffffu;
    }
    std::atomic_thread_fence(std::memory_order_release);
  };
  // members per team of the persistent kernel (its all-gather covers up to 32); the host sees ONE record group per channel
  int psplits_dev = std::max(1, std::min({32, (4 * ctx->compute_units + nch_dev - 1) / nch_dev, std::max(1, approx_chunks * 8 / (fast_nominal == 2 ? 16 : 8) / 48)}));
  if (persist_lane)  // members of a lane-kernel team: workgroups of 8 waves (as gc_track_device)
    psplits_dev = std::max(1, std::min({max_arms == 1 ? 8 : max_arms == 2 ? 6 : 4, approx_chunks * 8 / (64 * 8 * 2), std::max(1, 2 * ctx->compute_units / nch_dev)}));
  if (const char* ev = GC_TUNE_ENV("GC_TRACK_SPLITS")) psplits_dev = std::max(1, std::min(persist_lane ? (max_arms == 1 ? 8 : max_arms == 2 ? 6 : 4) : 32, std::atoi(ev)));
  // A grid the device cannot hold whole (the teams spin on each other's messages) is refused by the launch: the teams are halved
  // until it fits - 192 channels x 1 member is still one launch for the whole call, where falling back to a launch per epoch cost
  // 33-90 us per epoch from 48 GPS L1 C/A channels on (bench.py closed_loop_sweep, round 4)
  while (persist) {
    bool refused = false;
    pa.n_epochs = n_epochs;
    pa.splits = psplits_dev;
    pa.if_nsamples = ctx->if_nsamples;
    pa.host_loop = 1;
    pa.host_tagged = ctx->h_tagged_pinned;
    std::vector<gcorr::DevLoopChan> hc((size_t)nch);
    std::memset(hc.data(), 0, sizeof(gcorr::DevLoopChan) * (size_t)nch);
    for (int c = 0; c < nch; ++c) hc[c].blk.channel = init[c].channel;  // table staging needs the channel of each team
    auto take = [&](int slot, size_t bytes, bool host) -> hipError_t { return gc_buf_reserve(ctx->trk[slot], bytes, host); };
    hipError_t e = take(gc_context::TRK_CHAN, sizeof(gcorr::DevLoopChan) * (size_t)nch, false);
    if (e == hipSuccess) e = take(gc_context::TRK_DESC, sizeof(gcorr::msg_t) * (size_t)nch * gcorr::kDescWords, false);
    const size_t ppart = sizeof(gcorr::msg_t) * (size_t)nch * psplits_dev * (persist_lane ? 6 * max_arms : 2 * 2);  // the teams' partial-sum messages
    if (e == hipSuccess) e = take(gc_context::TRK_PART, ppart, false);
    if (e == hipSuccess) e = take(gc_context::TRK_ARGS, sizeof pa, false);
    if (e == hipSuccess) e = take(gc_context::TRK_HDESC, sizeof(gcorr::msg_t) * (size_t)nch * gcorr::kDescWords, true);
    if (e == hipSuccess) {
      pa.chan = (gcorr::DevLoopChan*)ctx->trk[gc_context::TRK_CHAN].p;
      pa.desc_msg = (gcorr::msg_t*)ctx->trk[gc_context::TRK_DESC].p;
      pa.part_msg = (gcorr::msg_t*)ctx->trk[gc_context::TRK_PART].p;
      d_pargs = (gcorr::DevLoopArgs*)ctx->trk[gc_context::TRK_ARGS].p;
      h_desc = (gcorr::msg_t*)ctx->trk[gc_context::TRK_HDESC].p;
      e = hipMemsetAsync(pa.part_msg, 0, ppart, ctx->stream);
    }
    if (e == hipSuccess) {
      std::memset(h_desc, 0, sizeof(gcorr::msg_t) * (size_t)nch * gcorr::kDescWords);
      pa.host_desc = h_desc;
      e = hipMemsetAsync(pa.desc_msg, 0, sizeof(gcorr::msg_t) * (size_t)nch * gcorr::kDescWords, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(pa.chan, hc.data(), sizeof(gcorr::DevLoopChan) * (size_t)nch, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pargs, &pa, sizeof pa, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
      gcorr::KArgs a;
      std::memset(&a, 0, sizeof a);
      a.if_base = ctx->d_if;
      a.chans = ctx->d_channels;
      a.fs = ctx->fs;
      a.inv_fs = 1.0 / ctx->fs;
      a.nblocks = nch;
      a.splits = psplits_dev;
      a.bpw = 1;
      a.stride = 1;
      a.share_el = (p->el_spacing == 0.5) ? 1 : 0;
      a.devloop = d_pargs;
      a.xcd_swizzle = 0;
      if (persist_lane) {
        a.share_el = 0;
        a.derived = derived_nominal ? 1 : 0;
        const int lrc = gc_launch_devloop_lane(ctx, a, (unsigned int)(nch * psplits_dev), max_arms, share_lane_nominal && !derived_nominal, 8);
        if (lrc != GC_OK) e = hipErrorUnknown;
        refused = lrc == GC_E_NOFIT;   // (only a grid that did not fit is retried with smaller teams)
      } else {
        const int lrc = gc_launch_devloop(ctx, a, (unsigned int)(nch * psplits_dev), fast_nominal == 2, a.share_el != 0);
        if (lrc != GC_OK) e = hipErrorUnknown;
        refused = lrc == GC_E_NOFIT;
      }
    }
    if (e != hipSuccess && refused && psplits_dev > 1) {
      (void)hipGetLastError();
      gc_persistent_done(ctx);
      psplits_dev = (psplits_dev + 1) / 2;
      continue;
    }
    ctx->last_track_mode = e == hipSuccess ? 1 : 0;
    if (e == hipSuccess && GC_TUNE_ENV("GC_TRACK_DEBUG"))
      std::fprintf(stderr, "gc_track: persistent kernel: %d channels x %d members (%s kernel)\n", nch, psplits_dev, persist_lane ? "lane" : "fast");
    if (e == hipSuccess) ctx->last_kernel = persist_lane ? 0 : 1;  // gc_debug_last_kernel: lane / fast kernel (persistent instantiation)
    if (e != hipSuccess) {  // could not set the persistent kernel up: launch per epoch
      if (GC_TUNE_ENV("GC_TRACK_DEBUG"))
        std::fprintf(stderr, "gc_track: persistent kernel refused (%s): %d channels x %d members, lane %d - launching per epoch\n", hipGetErrorString(e), nch,
                     psplits_dev, (int)persist_lane);
      (void)hipGetLastError();
      persist_free();
      persist = false;
      persist_lane = false;
    }
    break;
  }

  double tau1code, tau2code, tau1carr, tau2carr;
  calc_loop_coef(p->dll_noise_bw, p->dll_damping, 1.0, &tau1code, &tau2code);   // tracking.m:100-102
  calc_loop_coef(p->pll_noise_bw, p->pll_damping, 0.25, &tau1carr, &tau2carr);  // tracking.m:109-110
  const double pdi = p->int_time;

  std::vector<ChanState> st((size_t)nch);
  for (int c = 0; c < nch; ++c) {
    ChanState& s = st[c];
    s.active = true;
    // tracking.m:150-152; positions count from the IF buffer's first sample, which is record sample r->origin (gc_track_resume on a
    // window: a run that starts in a window other than the record's first one must not read from that window's start)
    s.pos = p->skip_samples + init[c].code_phase - 1 - (r ? r->origin : 0);
    s.table_phase = init[c].table_phase;
    s.code_freq = s.code_freq_basis = init[c].code_freq;  // :163 / GPS_L5C :165
    s.carr_freq = s.carr_basis = init[c].acquired_freq;   // :167-168
    if (r && r->resume) {  // continue where the previous window stopped
      const gc_channel_state& g = r->state[c];
      s.active = g.status == 0;
      s.aborted = g.status == 2;
      s.pos = g.next_sample - r->origin;
      s.code_freq = g.code_freq;
      s.rem_code = g.rem_code_phase;
      s.carr_freq = g.carr_freq;
      s.rem_carr = g.rem_carr_phase;
      s.old_code_nco = g.old_code_nco;
      s.old_code_err = g.old_code_error;
      s.old_carr_nco = g.old_carr_nco;
      s.old_carr_err = g.old_carr_error;
      s.d_carr_err = g.d_carr_error;
      s.d2_carr_err = g.d2_carr_error;
      s.table_phase = g.table_phase;
    }
  }
  const int64_t origin = r ? r->origin : 0;

  std::vector<int> slot((size_t)nch);
  std::atomic<bool> any_range{false}, any_diverged{false};  // set by whichever host thread serves the channel
  double t_launch = 0.0, t_wait = 0.0;  // GC_TRACK_TIMING: host time in the launch call / until the records arrived
  const auto t_loop0 = std::chrono::steady_clock::now();

  // tracking.m:219-245 for channel c at epoch e: block size and position -> descriptor.  false: the channel ends here (its NCO
  // left the finite numbers, or the record holds no whole block any more); a persistent team is told to stop.
  auto prepare = [&](int c, int e, gc_block& b) -> bool {
    ChanState& s = st[c];
    const double step = s.code_freq / p->sampling_freq;                      // :219
    if (!(step > 0.0) || !(step < 1e6) || !std::isfinite(s.carr_freq)) {
      // all-zero sums make atan(0/0) = NaN of the carrier and code NCOs; MATLAB then fails in fread(fid, NaN): stop the channel
      s.active = false;
      s.aborted = true;
      any_diverged = true;
      if (persist) write_desc(c, e, nullptr, 2ull);
      return false;
    }
    const int n = (int)std::ceil((p->code_length - s.rem_code) / step);     // :222
    if (s.pos < 0 || (uint64_t)(s.pos + n) > ctx->if_nsamples) {            // :241-245
      s.active = false;
      s.aborted = true;
      any_range = true;
      if (persist) write_desc(c, e, nullptr, 2ull);  // this channel's team stops
      return false;
    }
    std::memset(&b, 0, sizeof b);
    b.channel = init[c].channel;
    b.blksize = n;
    b.first_sample = s.pos;
    b.rem_code_phase = s.rem_code;
    if (p->table_phase_count > 0 && s.table_phase > 0)  // GPS_L2C tracking.m:261: index + codeLength*(CLCodePhase-1)
      b.table_offset[1] = (int32_t)p->code_length * (s.table_phase - 1);
    b.code_phase_step = step;
    b.el_spacing = p->el_spacing;
    b.carr_freq = s.carr_freq;
    b.rem_carr_phase = s.rem_carr;
    return true;
  };
  // tracking.m:249-348 for channel c at epoch e: the recorded state, the discriminators and loop filters, the next block's state
  auto close_epoch = [&](int c, int e, const gc_block& b, const double (&sums)[GC_OUT_STRIDE]) {
      ChanState& s = st[c];
      const double R = ctx->ch[b.channel].index_scale;
      const double i_e = sums[0], q_e = sums[1], i_p = sums[2], q_p = sums[3], i_l = sums[4], q_l = sums[5];
      double* o = out + (size_t)c * GC_TRK_NFIELDS * n_epochs;
      auto rec = [&](int f, double v) { o[(size_t)f * n_epochs + e] = v; };
      rec(GC_TRK_ABSOLUTE_SAMPLE, (double)(s.pos + origin));  // :212-216
      rec(GC_TRK_REM_CODE_PHASE, s.rem_code);      // :249
      rec(GC_TRK_REM_CARR_PHASE, s.rem_carr);      // :277
      const int n = b.blksize;
      const double step = b.code_phase_step;
      // remCodePhase update, :273 (R = 1) / GAL_E1C tracking.m:268 (R = 2).  tcode(blksize) is the
      // colon end point ((blksize-1)*codePhaseStep + remCodePhase) * R.
      const double t_last = ((n - 1) * step + s.rem_code) * R;
      const double rem_code_new = (R != 1.0) ? (t_last / R + step) - p->code_length : (t_last + step) - p->code_length;
      // remCarrPhase update, :280-283
      const double time_n = (double)n / p->sampling_freq;
      const double trig_n = ((s.carr_freq * 2.0 * kPi) * time_n) + s.rem_carr;
      const double rem_carr_new = std::fmod(trig_n, 2 * kPi);
      s.pos += n;
      s.rem_code = rem_code_new;
      s.rem_carr = rem_carr_new;

      // ---- PLL discriminator (:305) and pilot combining ---------------------------------
      double carr_err = std::atan(q_p / i_p) / (2.0 * kPi);
      double code_err = (std::sqrt(i_e * i_e + q_e * q_e) - std::sqrt(i_l * i_l + q_l * q_l)) /
                        (std::sqrt(i_e * i_e + q_e * q_e) + std::sqrt(i_l * i_l + q_l * q_l));  // :322-323
      double pilot6[6] = {sums[6], sums[7], sums[8], sums[9], sums[10], sums[11]};  // arm 1 as correlated
      if (p->pilot_combine != 0) {
        if (p->pilot_combine == 4) {
          // BDS B1C wide-band: arms {data, pilot BOC(1,1), pilot BOC(6,1)} -> one pilot (WB_tracking.m:364-369)
          const double a61 = -std::sqrt(4.0 / 33.0), a11 = std::sqrt(29.0 / 33.0);
          for (int x = 0; x < 3; ++x) {
            const double i11 = sums[6 + 2 * x], q11 = sums[7 + 2 * x], i61 = sums[12 + 2 * x], q61 = sums[13 + 2 * x];
            pilot6[2 * x] = a61 * i61 + a11 * q11;
            pilot6[2 * x + 1] = a61 * q61 - a11 * i11;
          }
        } else if (p->pilot_combine == 5) {
          // Galileo E1-C CBOC(6,1,1/11): pilot subcarrier sqrt(10/11) sc_BOC(1,1) - sqrt(1/11) sc_BOC(6,1), both in phase
          // (Galileo OS SIS ICD 2.3.3; BASELINE config 3 - the reference itself tracks E1 with BOC(1,1) only)
          const double a11 = std::sqrt(10.0 / 11.0), a61 = -std::sqrt(1.0 / 11.0);
          for (int x = 0; x < 3; ++x) {
            const double i11 = sums[6 + 2 * x], q11 = sums[7 + 2 * x], i61 = sums[12 + 2 * x], q61 = sums[13 + 2 * x];
            pilot6[2 * x] = a11 * i11 + a61 * i61;
            pilot6[2 * x + 1] = a11 * q11 + a61 * q61;
          }
        }
        const double pi_e = pilot6[0], pq_e = pilot6[1], pi_p = pilot6[2], pq_p = pilot6[3], pi_l = pilot6[4], pq_l = pilot6[5];
        double carr_err_q;
        if (p->pilot_combine == 1) {
          // QI = (I_PQ + 1i*Q_PQ) * exp(-1i*pi/2), GPS_L5C tracking.m:340
          const double cr = std::cos(kPi / 2), ci = -std::sin(kPi / 2);
          const double re = pi_p * cr - pq_p * ci;
          const double im = pi_p * ci + pq_p * cr;
          carr_err_q = std::atan(im / re) / (2.0 * kPi);
        } else if (p->pilot_combine == 3) {
          carr_err_q = std::atan(-pi_p / pq_p) / (2.0 * kPi);  // BDS/B1C NB_tracking.m:341
        } else {
          carr_err_q = std::atan(pq_p / pi_p) / (2.0 * kPi);  // GAL_E1C tracking.m:309; WB_tracking.m:381
        }
        double code_err_q = (std::sqrt(pi_e * pi_e + pq_e * pq_e) - std::sqrt(pi_l * pi_l + pq_l * pq_l)) /
                            (std::sqrt(pi_e * pi_e + pq_e * pq_e) + std::sqrt(pi_l * pi_l + pq_l * pq_l));
        const bool pll_w = p->pll_weight[0] != 0.0 || p->pll_weight[1] != 0.0;
        const bool dll_w = p->dll_weight[0] != 0.0 || p->dll_weight[1] != 0.0;
        if (p->dll_scale != 0.0) {  // NB_tracking.m:346-348
          code_err = code_err * p->dll_scale;
          code_err_q = code_err_q * p->dll_scale;
        }
        if (pll_w)  // (carrError*11 + p11_carrError*29)/40, NB_tracking.m:342; (carrError*1 + p_carrError*3)/4, WB :382
          carr_err = (carr_err * p->pll_weight[0] + carr_err_q * p->pll_weight[1]) / (p->pll_weight[0] + p->pll_weight[1]);
        else
          carr_err = (carr_err + carr_err_q) / 2;
        if (dll_w && p->pilot_combine == 4)  // codeError*factor + p_codeError*(1-factor), WB_tracking.m:403
          code_err = code_err * p->dll_weight[0] + code_err_q * p->dll_weight[1];
        else if (dll_w)  // (codeError*11 + p11_codeError*29)/40, NB_tracking.m:349
          code_err = (code_err * p->dll_weight[0] + code_err_q * p->dll_weight[1]) / (p->dll_weight[0] + p->dll_weight[1]);
        else
          code_err = (code_err + code_err_q) / 2;
      }
      double carr_nco;
      if (p->pll_kind == GC_PLL_2ND_ORDER) {
        carr_nco = s.old_carr_nco + (tau2carr / tau1carr) * (carr_err - s.old_carr_err) + carr_err * (pdi / tau1carr);  // :308-309
        s.old_carr_nco = carr_nco;
        s.old_carr_err = carr_err;
      } else {
        s.d2_carr_err = s.d2_carr_err + carr_err * p->pf3;  // GPS_L5C tracking.m:351-353
        s.d_carr_err = s.d2_carr_err + carr_err * p->pf2 + s.d_carr_err;
        carr_nco = s.d_carr_err + carr_err * p->pf1;
      }
      rec(GC_TRK_CARR_FREQ, s.carr_freq);  // :314
      s.carr_freq = s.carr_basis + carr_nco;  // :317
      // ---- DLL (:326-335) -----------------------------------------------------------------
      const double code_nco = s.old_code_nco + (tau2code / tau1code) * (code_err - s.old_code_err) + code_err * (pdi / tau1code);
      s.old_code_nco = code_nco;
      s.old_code_err = code_err;
      rec(GC_TRK_CODE_FREQ, s.code_freq);  // :332
      s.code_freq = s.code_freq_basis - code_nco;  // :335
      rec(GC_TRK_DLL_DISCR, code_err);
      rec(GC_TRK_DLL_DISCR_FILT, code_nco);
      rec(GC_TRK_PLL_DISCR, carr_err);
      rec(GC_TRK_PLL_DISCR_FILT, carr_nco);
      rec(GC_TRK_I_E, i_e);
      rec(GC_TRK_Q_E, q_e);
      rec(GC_TRK_I_P, i_p);
      rec(GC_TRK_Q_P, q_p);
      rec(GC_TRK_I_L, i_l);
      rec(GC_TRK_Q_L, q_l);
      if (ctx->ch[b.channel].arms >= 2) {
        rec(GC_TRK_PILOT_I_E, pilot6[0]);
        rec(GC_TRK_PILOT_Q_E, pilot6[1]);
        rec(GC_TRK_PILOT_I_P, pilot6[2]);
        rec(GC_TRK_PILOT_Q_P, pilot6[3]);
        rec(GC_TRK_PILOT_I_L, pilot6[4]);
        rec(GC_TRK_PILOT_Q_L, pilot6[5]);
      }
      if (p->table_phase_count > 0 && s.table_phase > 0 && p->pilot_combine != 0) {  // GPS_L2C tracking.m:357-360
        s.table_phase += 1;
        if (s.table_phase >= p->table_phase_count + 1) s.table_phase = 1;
      }
      s.epochs = e + 1;
  };

  // ---- persistent kernel, every channel at its own pace ---------------------------------------------------------------
  // The channels share nothing (tracking.m:133): each team of the persistent kernel waits for ITS descriptor and answers with
  // ITS record group.  So the host does not wait for the slowest channel of an epoch before it closes any: whichever record
  // group carries its channel's next tag is closed at once and that channel's next descriptor goes out, while the other teams
  // are still correlating or their records are on the way.  Lock step (the loop below) cost an epoch the sum of the PCIe round
  // trip, the team's work and the twelve closures; this way the round trip of one channel hides behind the others' work.
  // (Not under pause_at_end: a window of a streamed record must end all channels at the same epoch.)
  const bool async = persist && poll && !(r && r->pause_at_end) && GC_TUNE_ENV("GC_TRACK_LOCKSTEP") == nullptr;
  if (async) {
    const int arms6 = max_arms * 6;
    struct alignas(64) Slot {  // a channel's loop state on cache lines of its own: neighbouring channels may belong to other threads
      gc_block cur;
      int ep = 0;
      bool waiting = false;
      std::chrono::steady_clock::time_point sent;
    };
    std::vector<Slot> sl((size_t)nch);
    std::atomic<bool> lost{false};
    std::atomic<int> lost_epoch{0};
    // Is the one host thread that closes all loops what bounds the epoch?  Measured: no.  With the channels dealt out to 2, 3 and 4 host
    // threads (GC_TRACK_THREADS; a contiguous share each, this thread serves the first) twelve L1 C/A channels took 7.04 / 6.91 /
    // 6.93 us per epoch against 6.77 us on one thread, configs 3 and 4 lost 4-11 %: an epoch is one channel's own chain (descriptor
    // over PCIe, relay, correlate, all-gather, records back), and more pollers only add traffic on the lines the device writes.
    // So one thread is the default; the knob stays for hosts with slower cores.
    // ... at 12 channels.  With the closed_loop_sweep's 96 / 192 channels the one closer is what the epoch waits for: 14.2 / 20.4 us on
    // one thread, 12.5 / 15.8 us on four - so one more thread per 64 channels, four at most.
    int nthreads = std::min(4, 1 + nch / 64);
    if (const char* ev = GC_TUNE_ENV("GC_TRACK_THREADS")) nthreads = std::max(1, std::min({16, nch, std::atoi(ev)}));
    auto serve = [&](int t) {
      const int c0 = (int)((long long)t * nch / nthreads), c1 = (int)((long long)(t + 1) * nch / nthreads);
      int outstanding = 0;
      const auto now0 = std::chrono::steady_clock::now();
      for (int c = c0; c < c1; ++c) {
        if (!st[c].active) continue;
        if (!prepare(c, 0, sl[c].cur)) continue;
        write_desc(c, 0, &sl[c].cur, 0ull);
        sl[c].waiting = true;
        sl[c].sent = now0;
        ++outstanding;
      }
      unsigned int idle = 0, turns = 0;
      // a descriptor's time of sending, to the precision the timeout needs: refreshed every 256 turns of the poll loop WHETHER OR NOT
      // records arrived (a stamp refreshed only while idle stays at now0 for a whole busy run, and the first stall after
      // poll_timeout_ms of wall time would then read as a lost epoch), and again whenever the loop has been idle for 1024 turns
      auto stamp = now0;
      while (outstanding > 0 && !lost.load(std::memory_order_relaxed)) {
        bool progress = false;
        if ((++turns & 255u) == 0) stamp = std::chrono::steady_clock::now();
        for (int c = c0; c < c1; ++c) {
          if (!sl[c].waiting) continue;
          const unsigned int tag = (unsigned int)sl[c].ep + 1u;
          volatile gcorr::TaggedSlot* grp = tagged + (size_t)c * GC_OUT_STRIDE;
          bool ready = true;
          for (int v = arms6 - 1; v >= 0 && ready; --v) ready = grp[v].tag == tag;
          if (!ready) continue;
          std::atomic_thread_fence(std::memory_order_acquire);
          double sums[GC_OUT_STRIDE];
          for (int v = 0; v < GC_OUT_STRIDE; ++v) sums[v] = v < arms6 ? grp[v].value : 0.0;
          close_epoch(c, sl[c].ep, sl[c].cur, sums);
          progress = true;
          ++sl[c].ep;
          if (sl[c].ep < n_epochs && prepare(c, sl[c].ep, sl[c].cur)) {
            write_desc(c, sl[c].ep, &sl[c].cur, 0ull);
            sl[c].sent = stamp;
          } else {
            sl[c].waiting = false;  // all epochs done (the team leaves by itself) or the channel ended (prepare told its team)
            --outstanding;
          }
        }
        if (progress) {
          idle = 0;
        } else if ((++idle & 1023u) == 0) {
          // first epoch of the persistent kernel: code load + launch of the whole grid
          const auto now = stamp = std::chrono::steady_clock::now();
          for (int c = c0; c < c1; ++c)
            if (sl[c].waiting && now - sl[c].sent > std::chrono::milliseconds(sl[c].ep == 0 ? std::max(5000, poll_timeout_ms) : poll_timeout_ms)) {
              lost_epoch.store(sl[c].ep, std::memory_order_relaxed);
              lost.store(true, std::memory_order_relaxed);
              break;
            }
        }
      }
    };
    {
      std::vector<std::thread> helpers;
      for (int t = 1; t < nthreads; ++t) helpers.emplace_back(serve, t);
      serve(0);
      for (std::thread& h : helpers) h.join();
    }
    if (lost.load()) {
      persist_stop();
      GC_HIP(hipStreamSynchronize(ctx->stream));
      persist_free();
      gc_set_error("gc_track: result records of epoch %d did not arrive", lost_epoch.load());
      return GC_E_HIP;
    }
    t_wait = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop0).count();
  }

  for (int e = 0; e < n_epochs && !async; ++e) {
    if (r && r->pause_at_end) {
      bool fits = true;
      for (int c = 0; c < nch && fits; ++c) {
        const ChanState& s = st[c];
        if (!s.active) continue;
        const double step = s.code_freq / p->sampling_freq;
        if (!(step > 0.0) || !(step < 1e6)) continue;  // handled below (diverged NCO)
        const int n = (int)std::ceil((p->code_length - s.rem_code) / step);
        fits = s.pos >= 0 && (uint64_t)(s.pos + n) <= ctx->if_nsamples;
      }
      if (!fits) {
        r->paused = true;
        break;
      }
    }
    int nb = 0;
    for (int c = 0; c < nch; ++c) {
      if (!st[c].active) continue;
      if (!prepare(c, e, blocks[nb])) continue;
      slot[nb] = c;
      ++nb;
    }
    if (nb == 0) break;
    const bool derived = any_mixed && all_mixed_derived && !any_three_plain && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL;
    ctx->launch_derived = derived;
    int fast = derived ? 0 : any_mixed ? -1 : (gc_fast_lds_ok(ctx) && !ctx->force_generic) ? 2 : 0;
    for (int k = 0; k < nb && fast > 0; ++k) fast = std::min(fast, gc_block_lowrate_level(ctx, blocks[k]));
    bool share = true;
    for (int k = 0; k < nb && share; ++k) share = gc_block_shares_el(ctx, blocks[k]);
    ctx->scope_share_lane = true;
    for (int k = 0; k < nb && ctx->scope_share_lane; ++k) ctx->scope_share_lane = gc_block_shares_el_lane(ctx, blocks[k]);
    const unsigned int tag = (unsigned int)(e + 1);
    const bool polled = poll && fast >= 0;
    const auto tt0 = std::chrono::steady_clock::now();
    if (persist) {
      for (int k = 0; k < nb; ++k) write_desc(slot[k], e, &blocks[k], 0ull);
    } else {
      rc = gc_launch_correlator(ctx, blocks, nb, splits, splits == 1 ? partial : nullptr, partial, max_arms, fast, 0,
                                polled ? tag : 0u, share);
      if (rc) return rc;
    }
    const auto tt1 = std::chrono::steady_clock::now();
    bool signalled = false;
    if (polled) {
      // wait until every record of this launch carries the epoch tag (bounded: never hang here).  A busy device (other
      // contexts' kernels in front of ours) can delay a launch by much more than its own few microseconds: when the poll
      // budget runs out, the launch-per-epoch mode falls back to a stream synchronise and looks once more before giving up.
      const int arms6 = max_arms * 6;
      const int hs = persist ? 1 : splits;  // record groups per block as the host sees them (the persistent kernel's teams add up on the device)
      auto wait_tags = [&](std::chrono::milliseconds budget) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < nb * hs; ++k)
          for (int v = 0; v < arms6; ++v) {
            // record group of (block, split): blocks are numbered per launch, teams of the persistent kernel per channel
            const size_t grp = persist ? (size_t)slot[k] : (size_t)k;
            volatile gcorr::TaggedSlot* s = tagged + grp * GC_OUT_STRIDE + v;
            unsigned int spins = 0;
            while (s->tag != tag)
              if ((++spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > budget) return false;
          }
        return true;
      };
      // first epoch of the persistent kernel: code load + launch of the whole grid
      signalled = wait_tags(std::chrono::milliseconds((persist && e == 0) ? std::max(5000, poll_timeout_ms) : poll_timeout_ms));
      if (!signalled && !persist && hipStreamSynchronize(ctx->stream) == hipSuccess) signalled = wait_tags(std::chrono::milliseconds(1));
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    const auto tt2 = std::chrono::steady_clock::now();
    t_launch += std::chrono::duration<double, std::micro>(tt1 - tt0).count();
    t_wait += std::chrono::duration<double, std::micro>(tt2 - tt1).count();
    if (!signalled) {
      if (persist && GC_TUNE_ENV("GC_TRACK_TIMING")) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        std::fprintf(stderr, "gc_track persistent: epoch %d records missing; stream: %s; first tags:", e, hipGetErrorString(q));
        for (int k = 0; k < std::min(nb, 16); ++k) std::fprintf(stderr, " %u", tagged[(size_t)slot[k] * GC_OUT_STRIDE].tag);
        std::fprintf(stderr, "\n");
      }
      if (persist) persist_stop();
      GC_HIP(hipStreamSynchronize(ctx->stream));
      if (polled) {
        persist_free();
        gc_set_error("gc_track: result records of epoch %d did not arrive", e);
        return GC_E_HIP;
      }
    }

    for (int k = 0; k < nb; ++k) {
      const int c = slot[k];
      double sums[GC_OUT_STRIDE];
      for (int v = 0; v < GC_OUT_STRIDE; ++v) {
        double acc = 0.0;
        for (int sp = 0; sp < (persist ? 1 : splits); ++sp)
          acc += polled ? ((v < max_arms * 6) ? tagged[(persist ? (size_t)c : (size_t)k * splits + sp) * GC_OUT_STRIDE + v].value : 0.0)
                        : partial[((size_t)k * splits + sp) * GC_OUT_STRIDE + v];
        sums[v] = acc;
      }
      close_epoch(c, e, blocks[k], sums);
    }
  }

  // The reference processes channels one after the other and returns from the whole function at
  // the first short read (tracking.m:241-245): channels after the first aborted one are never run.
  int first_aborted = nch;
  for (int c = 0; c < nch; ++c)
    if (st[c].aborted) {
      first_aborted = c;
      break;
    }
  for (int c = 0; c < nch; ++c) {
    if (c > first_aborted) {
      double* o = out + (size_t)c * GC_TRK_NFIELDS * n_epochs;
      std::fill(o, o + (size_t)GC_TRK_NFIELDS * n_epochs, 0.0);
      epochs_done[c] = 0;
    } else {
      epochs_done[c] = st[c].epochs;
    }
  }
  if (r) {
    for (int c = 0; c < nch; ++c) {
      const ChanState& s = st[c];
      gc_channel_state& g = r->state[c];
      g.next_sample = s.pos + origin;
      g.code_freq = s.code_freq;
      g.rem_code_phase = s.rem_code;
      g.carr_freq = s.carr_freq;
      g.rem_carr_phase = s.rem_carr;
      g.old_code_nco = s.old_code_nco;
      g.old_code_error = s.old_code_err;
      g.old_carr_nco = s.old_carr_nco;
      g.old_carr_error = s.old_carr_err;
      g.d_carr_error = s.d_carr_err;
      g.d2_carr_error = s.d2_carr_err;
      g.table_phase = s.table_phase;
      g.status = s.aborted ? 2 : 0;
      g.reserved = 0;
    }
  }
  const bool persist_was = persist;
  if (persist) {  // teams that were not told to stop leave after n_epochs by themselves; the others were stopped above
    persist_stop();
    (void)hipStreamSynchronize(ctx->stream);
    persist_free();
  }
  if (GC_TUNE_ENV("GC_TRACK_TIMING") && n_epochs > 0)
    std::fprintf(stderr, "gc_track: per epoch %.2f us in the launch call / descriptor writes, %.2f us until the records arrived, %.2f us total (%s, %d workgroups per block)\n",
                 t_launch / n_epochs, t_wait / n_epochs,
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop0).count() / n_epochs,
                 persist_was ? (persist_lane ? "persistent lane kernel" : "persistent kernel") : "launch per epoch", persist_was ? psplits_dev : splits);
  if (any_diverged) {
    gc_set_error("gc_track: a channel's code / carrier NCO became non-finite (all-zero correlator sums?); its records end there");
    return GC_E_INVALID;
  }
  if (any_range) {
    gc_set_error("Not able to read the specified number of samples for tracking");
    return GC_E_RANGE;
  }
  return GC_OK;
}

// ---- device-side loop closure (devloop.h) -----------------------------------------------------------------------
// Same contract as gc_track for the signals the persistent kernel is instantiated for: single-arm channels whose
// blocks qualify for the transition-mask kernel with float2 tables (GPS L1 C/A, GLONASS L1OF, BDS B1I), int8 I/Q or Q/I
// records, no pilot.  Anything else returns GC_E_UNSUPPORTED and the caller uses gc_track.
extern "C" int gc_track_device(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init,
                               double* out, int32_t* epochs_done) {
  if (!ctx || !p || nch <= 0 || nch > GC_MAX_CHANNELS || !init || !out || !epochs_done || p->n_epochs <= 0) {
    gc_set_error("gc_track_device: bad arguments");
    return GC_E_INVALID;
  }
  if (!ctx->d_if) {
    gc_set_error("gc_track_device: no IF buffer loaded");
    return GC_E_STATE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  ctx->fs = p->sampling_freq;
  // channels that share the device during this call (gc_track_multi): teams are sized for all of them
  const int nch_dev = ctx->concurrent_jobs ? std::max(nch, ctx->concurrent_channels) : nch;
  int rc = gc_sync_channels(ctx);
  if (rc) return rc;
  gc_scope_reset(ctx);
  int max_arms = 1;
  bool single_r1 = true, all_derived = true;
  for (int c = 0; c < nch; ++c) {
    const int ci = init[c].channel;
    if (ci < 0 || ci >= GC_MAX_CHANNELS || !ctx->ch[ci].configured || !ctx->ch[ci].d_tab[0]) {
      gc_set_error("gc_track_device: channel %d not configured", ci);
      return GC_E_STATE;
    }
    const HostChannel& hcn = ctx->ch[ci];
    const bool hder = gc_channel_is_derived(hcn);  // three arms, the third derived from the second inside the lane kernel
    all_derived = all_derived && hder;
    for (int a = 0; a < hcn.arms; ++a)
      if (!hcn.d_tab[a] || (hcn.mult[a] != 1.0 && !(hder && a == 2))) {
        gc_set_error("gc_track_device: ramp multipliers other than a derived third arm are not covered (use gc_track)");
        return GC_E_UNSUPPORTED;
      }
    if ((rc = validate_track_channel(ctx, p, init[c], "gc_track_device"))) return rc;
    max_arms = std::max(max_arms, hcn.arms);
    single_r1 = single_r1 && hcn.arms == 1 && hcn.index_scale == 1.0;
    gc_scope_add(ctx, ci);
  }
  // three arms, the third derived from the second: Galileo E1-C CBOC (fold 5), BDS B1C wide-band (fold 4)
  const bool cboc = max_arms == 3 && all_derived && (p->pilot_combine == 5 || p->pilot_combine == 4);
  const bool i8c = ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL;  // int8 I/Q or Q/I record
  if ((max_arms > 2 && !cboc) || (p->pilot_combine > 3 && !cboc) || (p->pilot_combine != 0 && max_arms < 2) || (cboc && !i8c)) {
    gc_set_error("gc_track_device: configuration not covered by the persistent kernels (use gc_track)");
    return GC_E_UNSUPPORTED;
  }
  const int n_epochs = p->n_epochs;
  std::vector<gcorr::DevLoopChan> hc((size_t)nch);
  std::memset(hc.data(), 0, sizeof(gcorr::DevLoopChan) * (size_t)nch);
  int lowrate = 2;
  bool share = true;
  for (int c = 0; c < nch; ++c) {
    gcorr::DevLoopChan& s = hc[c];
    s.pos = p->skip_samples + init[c].code_phase - 1;
    s.code_freq = s.code_freq_basis = init[c].code_freq;
    s.carr_freq = s.carr_basis = init[c].acquired_freq;
    s.table_phase = init[c].table_phase;
    const double step = s.code_freq / p->sampling_freq;
    const int n = (int)std::ceil((p->code_length - s.rem_code) / step);
    gc_block& b = s.blk;
    b.channel = init[c].channel;
    if (p->table_phase_count > 0 && s.table_phase > 0)  // GPS_L2C tracking.m:261: index + codeLength*(CLCodePhase-1)
      b.table_offset[1] = (int32_t)p->code_length * (s.table_phase - 1);
    b.blksize = n;
    b.first_sample = s.pos;
    b.rem_code_phase = 0.0;
    b.code_phase_step = step;
    b.el_spacing = p->el_spacing;
    b.carr_freq = s.carr_freq;
    b.rem_carr_phase = 0.0;
    if (s.pos < 0 || (uint64_t)(s.pos + n) > ctx->if_nsamples) s.status = 2;  // not even one block: tracking.m:241-245
    gc_block probe = b;
    probe.code_phase_step = step * 1.001;  // head-room for the code NCO
    lowrate = std::min(lowrate, gc_block_lowrate_level(ctx, probe));
    share = share && gc_block_shares_el(ctx, b);
  }
  if (!i8c) lowrate = std::min(lowrate, 1);  // 16-sample chunks are an int8 I/Q format (corr_fast.hip)
  // transition-mask kernel (one-wave members) where it applies, else the lane kernel (16-wave member workgroups)
  bool any_window = false;
  for (int c = 0; c < nch; ++c)
    for (int a = 0; a < ctx->ch[init[c].channel].arms; ++a) any_window = any_window || ctx->ch[init[c].channel].window[a] != 0;
  const bool use_fast = single_r1 && lowrate > 0 && p->pilot_combine == 0 && gc_fast_table_mode(ctx) == 0 && !ctx->force_generic && !any_window;
  int splits, msgs_per_member, lane_waves = gcorr::kLaneWaves;
  bool share_lane = true;
  const int spl = lowrate == 2 ? 16 : 8;
  if (use_fast) {
    // team size: just under one lane-chunk per lane and member — the epoch is a latency chain
    const int chunks = (int)(p->code_length / (p->code_freq_basis / p->sampling_freq) / spl) + 1;
    splits = std::max(1, std::min({32, (4 * ctx->compute_units + nch_dev - 1) / nch_dev, std::max(1, chunks / 32)}));
    if (const char* e = GC_TUNE_ENV("GC_DEVLOOP_MEMBERS")) splits = std::max(1, std::min(32, std::atoi(e)));  // closer polls <= 62 messages
    if (ctx->persist_member_cap > 0) splits = std::min(splits, ctx->persist_member_cap);
    msgs_per_member = 2;
  } else {
    for (int c = 0; c < nch; ++c) share_lane = share_lane && gc_block_shares_el_lane(ctx, hc[c].blk);
    // member workgroups per channel: about four 64-sample steps per lane; the closer polls (members - 1) * 6 * arms <= 64 messages
    const int nsamp = hc[0].blk.blksize;
    lane_waves = 8;  // measured best for both the 1-ms and the 4-ms packages (scripts/devloop_lane_sweep.py)
    if (const char* e = GC_TUNE_ENV("GC_DEVLOOP_WAVES")) lane_waves = std::max(1, std::min(8, std::atoi(e)));  // the device-loop instantiations are bounded to 8 waves
    const int max_members = max_arms == 1 ? 8 : max_arms == 2 ? 6 : 4;  // (members - 1) * 6 * arms messages <= 64 lanes
    splits = std::max(1, std::min({max_members, nsamp / (64 * lane_waves * 2), std::max(1, 2 * ctx->compute_units / nch_dev)}));
    if (const char* e = GC_TUNE_ENV("GC_DEVLOOP_MEMBERS")) splits = std::max(1, std::min(max_members, std::atoi(e)));
    if (ctx->persist_member_cap > 0) splits = std::min(splits, ctx->persist_member_cap);
    msgs_per_member = 6 * max_arms;
  }
  // Teams on one XCD each (default).  Not next to other contexts' persistent kernels (gc_track_multi): two kernels that pin
  // their teams to the same XCDs were measured to run one after the other (E1 lane kernel 32 ms = its own 11 ms + the L1 C/A
  // kernel's 21 ms beside it), spread over the device they overlap completely and lose nothing alone (88.7 vs 89.7 ms).
  const bool xcd_local = GC_TUNE_ENV("GC_DEVLOOP_SPREAD") == nullptr && !ctx->concurrent_jobs;

  gcorr::DevLoopArgs ha;
  std::memset(&ha, 0, sizeof ha);
  ha.prm = *p;
  calc_loop_coef(p->dll_noise_bw, p->dll_damping, 1.0, &ha.tau1code, &ha.tau2code);
  calc_loop_coef(p->pll_noise_bw, p->pll_damping, 0.25, &ha.tau1carr, &ha.tau2carr);
  ha.k1code = ha.tau2code / ha.tau1code;
  ha.k2code = p->int_time / ha.tau1code;
  ha.k1carr = ha.tau2carr / ha.tau1carr;
  ha.k2carr = p->int_time / ha.tau1carr;
  ha.if_nsamples = ctx->if_nsamples;
  ha.n_epochs = n_epochs;
  ha.splits = splits;
  ha.code_index_scale_is_one = 1;
  if (const char* e = GC_TUNE_ENV("GC_DEVLOOP_SCOPE")) ha.reserved = std::atoi(e);  // message scope (devloop.h): 0 system (default), 2 agent
  ha.timing = GC_TUNE_ENV("GC_DEVLOOP_TIMING") ? std::atoi(GC_TUNE_ENV("GC_DEVLOOP_TIMING")) : 0;  // 1: host + in-kernel phase clocks, 2: host only
  ha.prefetch = GC_TUNE_ENV("GC_DEVLOOP_NO_PREFETCH") ? 0 : 1;  // the next epoch's first chunk fetched during the closure (corr_fast.hip)
  gcorr::DevLoopArgs* d_args = nullptr;
  const size_t rec_bytes = sizeof(double) * (size_t)nch * GC_TRK_NFIELDS * n_epochs;
  const size_t part_bytes = sizeof(gcorr::msg_t) * (size_t)nch * splits * msgs_per_member * (use_fast ? 2 : 1),  // fast kernel: two alternating halves
                desc_bytes = sizeof(gcorr::msg_t) * (size_t)nch * gcorr::kDescWords;
  hipError_t e = gc_buf_reserve(ctx->trk[gc_context::TRK_CHAN], sizeof(gcorr::DevLoopChan) * (size_t)nch, false);
  if (e == hipSuccess) e = gc_buf_reserve(ctx->trk[gc_context::TRK_PART], part_bytes, false);
  if (e == hipSuccess) e = gc_buf_reserve(ctx->trk[gc_context::TRK_DESC], desc_bytes, false);
  if (e == hipSuccess) e = gc_buf_reserve(ctx->trk[gc_context::TRK_RECORDS], rec_bytes, false);
  if (e == hipSuccess) e = gc_buf_reserve(ctx->trk[gc_context::TRK_ARGS], sizeof ha, false);
  const int cno_nk = (p->cno_interval > 1 && ctx->cno_out && p->cno_mode == GC_CNO_VSM) ? n_epochs / p->cno_interval : 0;
  const bool want_cno = cno_nk > 0 && (long long)nch * cno_nk <= ctx->cno_cap;
  const size_t cno_bytes = sizeof(double) * (size_t)nch * (size_t)std::max(cno_nk, 1);
  if (e == hipSuccess && want_cno) e = gc_buf_reserve(ctx->trk[gc_context::TRK_CNO], cno_bytes, false);
  if (e != hipSuccess) {
    gc_set_error("gc_track_device: %s", hipGetErrorString(e));
    return GC_E_NOMEM;
  }
  ha.one_writer = GC_TUNE_ENV("GC_DEVLOOP_ONE_WRITER") != nullptr;
  if (want_cno) {
    ha.cno = (double*)ctx->trk[gc_context::TRK_CNO].p;
    ha.cno_nk = cno_nk;
    if (hipMemsetAsync(ha.cno, 0, cno_bytes, ctx->stream) != hipSuccess) return GC_E_HIP;
  }
  ha.chan = (gcorr::DevLoopChan*)ctx->trk[gc_context::TRK_CHAN].p;
  ha.part_msg = (gcorr::msg_t*)ctx->trk[gc_context::TRK_PART].p;
  ha.desc_msg = (gcorr::msg_t*)ctx->trk[gc_context::TRK_DESC].p;
  ha.records = (double*)ctx->trk[gc_context::TRK_RECORDS].p;
  d_args = (gcorr::DevLoopArgs*)ctx->trk[gc_context::TRK_ARGS].p;
  if (e == hipSuccess) e = hipMemsetAsync(ha.records, 0, rec_bytes, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ha.part_msg, 0, part_bytes, ctx->stream);
  std::vector<gcorr::msg_t> hdesc((size_t)nch * gcorr::kDescWords);
  for (int c = 0; c < nch; ++c) {
    unsigned long long q[gcorr::kDescWords] = {0};
    std::memcpy(q, &hc[c].blk, sizeof(gc_block));
    q[gcorr::kDescWords - 1] = (unsigned long long)(hc[c].status == 2 ? 2 : 0);
    for (int i = 0; i < gcorr::kDescWords; ++i) hdesc[(size_t)c * gcorr::kDescWords + i] = gcorr::msg_t{(unsigned int)q[i], (unsigned int)(q[i] >> 32), 1u, 0u};
  }
  if (e == hipSuccess) e = hipMemcpyAsync(ha.desc_msg, hdesc.data(), desc_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(ha.chan, hc.data(), sizeof(gcorr::DevLoopChan) * (size_t)nch, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_args, &ha, sizeof ha, hipMemcpyHostToDevice, ctx->stream);
  auto cleanup = [&]() { gc_persistent_done(ctx); };  // the buffers stay with the context (GcBuf); the grid leaves the device's ledger
  if (e != hipSuccess) {
    cleanup();
    gc_set_error("gc_track_device: %s", hipGetErrorString(e));
    return GC_E_NOMEM;
  }
  gcorr::KArgs a;
  std::memset(&a, 0, sizeof a);
  a.if_base = ctx->d_if;
  a.blocks = nullptr;
  a.chans = ctx->d_channels;
  a.fs = ctx->fs;
  a.inv_fs = 1.0 / ctx->fs;
  a.nblocks = nch;
  a.splits = splits;
  a.bpw = 1;
  a.stride = 1;
  a.share_el = share ? 1 : 0;
  a.devloop = d_args;
  a.xcd_swizzle = xcd_local ? 1 : 0;
  const unsigned int grid = xcd_local ? (unsigned int)(((nch + 7) / 8) * 8 * splits) : (unsigned int)(nch * splits);
  a.derived = cboc ? 1 : 0;
  rc = use_fast ? gc_launch_devloop(ctx, a, grid, lowrate == 2, share) : gc_launch_devloop_lane(ctx, a, grid, max_arms, share_lane && !cboc, lane_waves);
  if (rc == GC_E_NOFIT && splits > 1) {
    // the grid does not fit the device whole: the same call again with teams half the size (see gc_track's persistent launch);
    // a structural refusal (no instantiation for these tables / arms) is not retried
    cleanup();
    const int cap0 = ctx->persist_member_cap;
    ctx->persist_member_cap = (splits + 1) / 2;
    rc = gc_track_device(ctx, p, nch, init, out, epochs_done);
    ctx->persist_member_cap = cap0;
    return rc;
  }
  if (rc == GC_E_NOFIT) rc = GC_E_UNSUPPORTED;  // one member per channel and still no room: the caller falls back to gc_track
  if (rc == GC_OK) {
    ctx->last_track_mode = 2;
    const auto t_l = std::chrono::steady_clock::now();
    e = hipStreamSynchronize(ctx->stream);
    const auto t_k = std::chrono::steady_clock::now();
    if (e == hipSuccess) e = hipMemcpy(out, ha.records, rec_bytes, hipMemcpyDeviceToHost);
    if (ha.timing)
      std::fprintf(stderr, "devloop: launch + kernel %.3f ms (%.2f us per epoch), records to host %.3f ms\n",
                   std::chrono::duration<double, std::milli>(t_k - t_l).count(),
                   std::chrono::duration<double, std::micro>(t_k - t_l).count() / n_epochs,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_k).count());
    if (e == hipSuccess) e = hipMemcpy(hc.data(), ha.chan, sizeof(gcorr::DevLoopChan) * (size_t)nch, hipMemcpyDeviceToHost);
    if (e == hipSuccess && want_cno) e = hipMemcpy(ctx->cno_out, ha.cno, cno_bytes, hipMemcpyDeviceToHost);  // computed by the closer (devloop.h)
    if (e != hipSuccess) {
      gc_set_error("gc_track_device: %s", hipGetErrorString(e));
      rc = GC_E_HIP;
    }
  }
  cleanup();
  if (rc) return rc;
  // same early-return semantics as gc_track: channels after the first exhausted one are never run
  int first_aborted = nch;
  bool timeout = false, diverged = false;
  for (int c = 0; c < nch; ++c) {
    timeout |= hc[c].status == 3;
    diverged |= hc[c].status == 4;
    if (ha.timing == 1 && hc[c].epochs_done > 0)
      std::fprintf(stderr, "devloop ch %d: correlate %.2f us, wait partials %.2f us, close+publish %.2f us per epoch (closer)\n", c,
                   hc[c].pad[0] / hc[c].epochs_done * 0.01, hc[c].pad[1] / hc[c].epochs_done * 0.01, hc[c].pad[2] / hc[c].epochs_done * 0.01);
    if ((hc[c].status == 2 || hc[c].status == 4) && first_aborted == nch) first_aborted = c;
  }
  if (timeout) {
    gc_set_error("gc_track_device: a team member timed out waiting for its epoch descriptor");
    return GC_E_HIP;
  }
  for (int c = 0; c < nch; ++c) {
    if (c > first_aborted) {
      double* o = out + (size_t)c * GC_TRK_NFIELDS * n_epochs;
      std::fill(o, o + (size_t)GC_TRK_NFIELDS * n_epochs, 0.0);
      epochs_done[c] = 0;
    } else {
      epochs_done[c] = hc[c].epochs_done;
    }
  }
  if (p->cno_mode != GC_CNO_VSM) gc_fill_cno_host(ctx, p, nch, out, epochs_done);  // Calc_CNo_PLD: from the records, as everywhere
  if (diverged) {
    gc_set_error("gc_track_device: a channel's code / carrier NCO became non-finite (all-zero correlator sums?); its records end there");
    return GC_E_INVALID;
  }
  if (first_aborted < nch) {
    gc_set_error("Not able to read the specified number of samples for tracking (channel slot %d)", first_aborted);
    return GC_E_RANGE;
  }
  return GC_OK;
}
