// corr_lane.hip — Early/Prompt/Late integrate-and-dump correlator for gfx950 (MI355X), any chipping rate.
//
// The kernel for signals whose replica crosses several table entries per 8-sample chunk (GPS L5, BDS
// B2a/B3I, Galileo E5 at 10.23 Mcps: 0.57 chip per sample at 18 Msps), where the transition-mask trick of
// corr_fast.hip does not apply and every sample needs its own table lookup.
//
// Replaces the vector expressions of GPS/GPS_L1CA/include/tracking.m:247-300 (and the R-scaled / multi-arm
// variants GAL_E1C/include/tracking.m:236-303, GPS_L5C/include/tracking.m:255-326):
//   T2  code-replica index ramps   tcode = a : step : b ; idx = ceil(tcode)+1      (:252-270)
//   T3  carrier replica            exp(-1i*((carrFreq*2*pi)*(n/fs) + remCarrPhase)) (:280-287)
//   T4  mix + six sums per arm                                                      (:291-300)
//
// Design (wave64; measured instruction costs in profiles/r01/ubench_valu_rates.txt — only f32 fma/mul/add
// and 32-bit integer add issue at full rate on gfx950, every other VALU instruction costs ~1.7x):
//   * lane = sample: a wavefront walks its block 64 consecutive samples at a time, so the 64 table lookups
//     of one ds_read fall into <= 40 consecutive LDS words — no bank conflicts (lane = 8-sample chunk put
//     the lanes 4.5 entries apart: 5.4 conflict cycles per read, measured);
//   * wave = work item: every wavefront owns whole (block, split) items, no barrier or LDS reduction after
//     the table staging; 16 wavefronts per workgroup share ONE staged copy of the channel's tables;
//   * tables of all arms interleaved per entry, as f32 when that fits half of LDS ({arm0, arm1} = one
//     ds_read_b64 per tap and sample, values feed v_fma_f32 directly), else as f16 (v_fma_mix_f32);
//   * ramp state per tap = ONE 64-bit integer Q (32.32 fixed point, biased so that its high word IS the
//     table index ceil(t)), advanced by one v_lshl_add_u64 per sample — no carry chains.  The 2^-33-chip
//     rounding of the per-step increment accumulates to < 1e-7 chip per block, which the near-tie window
//     absorbs: a wave-group containing a sample whose fraction lies inside the window (or any sample, in
//     blocks the host could not prove tie-free) is redone in float64 exactly as the reference computes it;
//   * with earlyLateSpc*R*M a multiple of 1/2 the late tap reads the early tap's entry + 2*spacing (SHARE);
//   * carrier: the sample times the BLOCK-UNIFORM step phasor exp(-i*2*pi*64*j*f/fs) of its step j inside a run of 128 steps (a
//     128-entry table of the wave in LDS, read as a broadcast), the sums turned once per run (Horner) and, when they are emptied,
//     by the lane's own phasor from the exact float64 phase - no per-lane recurrence;
//   * 6*ARMS float accumulators per lane, emptied every 256 samples of the lane into float64 totals of the wave (all 6*ARMS sums over the
//     64 lanes in one transposing reduction - wave_transpose_sum -, float64 add in LDS); results as doubles or (closed loop) as host-mapped tagged 16-byte records.
#include "corr_common.h"
#include "devloop.h"

using namespace gcorr;

namespace {

constexpr int kLW = kLaneWaves;  // wavefronts per workgroup
#ifndef GC_LANE_GRP
#define GC_LANE_GRP 4
#endif
constexpr int kGRP = GC_LANE_GRP;  // samples per lane and group: the loads of the next group fly under this one
// The per-lane float32 sums are emptied into float64 totals (one set per wave, in LDS) every GC_LANE_FLUSH_RUNS runs of
// kLaneReseedSteps steps and at the end of the block: with 2, no float32 partial sum covers more than 256 samples of a lane before
// it meets the other lanes in a 64-lane float32 tree and goes to float64 (the reference sums in float64).  In the
// closed loops a wave holds a fraction of a block (16 - 100 samples per lane) and the flush at the end of the block is the only one
// that ever happens; the batched launches' waves walk whole blocks (up to 2 812 samples per lane for a 10-ms BDS B1C block).
// Measured on one box (scripts/variants.sh, prof_shapes l5 / cboc, 10-s records): never flushing inside a block 2.414 / 1.758 ms,
// every 4 runs 2.454 / 1.789 (+1.7 %), every 2 runs 2.504 / 1.816 (+3.5 %); the batched replay of a loop's own records comes back
// 4x closer to the loop's sums (1.7e-8 -> 4e-9 of full scale).
#ifndef GC_LANE_FLUSH_RUNS
#define GC_LANE_FLUSH_RUNS 2
#endif
constexpr int kRhoPerWave = kLaneReseedSteps;

// TAB: 0 = f32 tables, 1 = f32 tables + ONE ramp for all three taps (earlyLateSpc*R*M == 1/2: HALF, below), 2 = f16 tables
// DEVLOOP = persistent launch with device-side loop closure (devloop.h): p.splits workgroups of 16 waves per channel, the
// block loop becomes the channel's epoch loop; sums are combined in LDS per workgroup, between workgroups by tagged
// messages, and wave 0 of the channel's first workgroup closes the loop.
template <int ARMS, int MODE, bool CL, int TAB, bool DEVLOOP = false, bool DER = false>
__global__ __launch_bounds__(DEVLOOP ? 8 * 64 : kLW * 64)  // device loop: at most 8 waves per member (256 VGPRs, no spills)
 void corr_epl_lane_kernel(const KArgs p, const InlineBlocks /*read via the segment pointer*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // DER: the last arm has no table: it is arm LA - 1 read at six times the ramp rate with a sign pattern (BOC(6,1) from
  // BOC(1,1): entry k6 of its padded table = entry p = (k6 + 5) / 6 of the neighbour's times (-1)^(p + k6), checked on the
  // host: gc_channel_is_derived).  Its sign comes out of the base ramp's fraction (lean_sample): no 49 104- or 122 760-entry table,
  // no ramp of its own.
  constexpr int LA = DER ? ARMS - 1 : ARMS;  // arms with a table in LDS
  constexpr bool kF16 = (TAB == 2);
  // PN: the derived arm's f32 image carries a third column, arm LA - 1 times (-1)^entry (gc_sync_channels): the sign of the
  // derived entry is then the sign of that column times one bit of the base ramp's fraction (below)
  constexpr bool PN = DER && !kF16 && GC_LANE_PN != 0;
  constexpr int AP = PN ? 4 : ArmPitch<LA>::v;  // values per staged entry
  // HALF: with earlyLateSpc*R*M == 1/2 (the reference's default 0.5-chip spacing on a 1x table: GPS L5, BDS B2a / B3I, Galileo
  // E5a / E5b, GPS L2C in doubled-code units) the late ramp is the early ramp + 1 exactly, and the prompt ramp t = u_E + 1/2 has
  // ceil(t) = ceil(u_E) + (frac(u_E) > 1/2): ONE ramp Q_E serves all three taps - table entries k_E and k_E + 1 come back from
  // one ds_read2, the prompt tap selects between them by the sign bit of Q_E's low word.  Per sample and two arms that is one
  // 64-bit add, one LDS instruction and two selects instead of two adds, three address computations and three LDS reads.
  constexpr bool kHalf = (TAB == 1);
  constexpr int kFlushRuns = GC_LANE_FLUSH_RUNS;
  constexpr int NT = kHalf ? 1 : 3;  // ramps carried per sample
  constexpr bool kReal = (MODE == I8_REAL || MODE == I16_REAL);
#ifdef GC_LANE_GRP_DER
  constexpr int GRP = DER ? GC_LANE_GRP_DER : kGRP;
#else
  // samples per lane and group (the derived arm's three extra ramps: 128 VGPRs hold two steps' indices, not four; the persistent
  // instantiations carry the closer's loop state in registers next to all this and are latency-, not throughput-bound: two as well)
  constexpr int GRP = (DER || DEVLOOP) ? 2 : kGRP;
#endif
  constexpr int bps = Fmt<MODE>::bps;
  typedef typename std::conditional<kF16, _Float16, float>::type tab_t;
  const tab_t* tab = reinterpret_cast<const tab_t*>(smem);  // [kGuard + maxn + kGuard][AP]

  long long wg = blockIdx.x;
  if (p.xcd_swizzle && !DEVLOOP) {
    // Workgroup b is dispatched to XCD b % 8.  Give every XCD one contiguous range of the descriptor list so
    // that neighbouring descriptors (the channels of one epoch, which read the same IF window) share an L2.
    const long long per = (long long)gridDim.x >> 3;  // the host rounds the grid up to a multiple of 8 when swizzling
    wg = (wg & 7) * per + (wg >> 3);
    if (wg >= p.total_wg) return;                     // (at most seven workgroups of the rounded grid)
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform (SGPR)
  // bpw == 1: the workgroup's 16 waves take 16 consecutive (block, split) items — the host keeps splits a
  // multiple of 16, so they share block and table.  bpw > 1 (periodic replay lists that interleave `stride`
  // channels epoch by epoch): the workgroup walks bpw consecutive epochs of ONE channel, wave w taking
  // epochs w, w + 16, ...
  // bpw == 1 and p.wide: ONE block per workgroup, split 16 ways over its waves, combined through LDS (big
  // lists of unrelated blocks: no partial buffer, no second kernel).
  const bool wave_items = p.bpw == 1;
  const bool wg_block = !DEVLOOP && wave_items && p.wide != 0;
  // DEVLOOP: p.splits member workgroups per channel; with xcd_swizzle a channel's members all run on one XCD
  // (workgroup b -> XCD b % 8): channel = (b % 8) + 8 * ((b / 8) / members), member = (b / 8) % members
  const int member = DEVLOOP ? (p.xcd_swizzle ? (int)((wg >> 3) % p.splits) : (int)(wg % p.splits)) : 0;
  const int nw = DEVLOOP ? (int)(blockDim.x >> 6) : kLW;  // waves of this workgroup (the device loop may launch fewer than 16)
  const int nsplit = DEVLOOP ? p.splits * nw : wg_block ? kLW : p.splits;
  const long long item = wg_block ? wg * kLW + wave : wave_items ? wg * kLW + wave : wg;
  const long long wq = DEVLOOP ? (p.xcd_swizzle ? (wg & 7) + 8 * ((wg >> 3) / p.splits) : wg / p.splits) : item / nsplit;
  const int split = DEVLOOP ? member * nw + wave : (int)(item - wq * nsplit);
  const long long grp = wq / p.stride;
  const int cslot = (int)(wq - grp * p.stride);

  // ---- stage the tables once per workgroup (all its blocks share channel and table offsets) ----------
  // DEVLOOP with a windowed table (GPS L2C's CL arm: 20 462 of 1 534 502 entries, the window moves on every epoch,
  // GPS_L2C/include/tracking.m:261): staged again per epoch from the descriptor's table offsets
  int maxn = 0;
  bool windowed = false;
  auto stage_tables = [&](const gc_block& blk0) __attribute__((always_inline)) {
    const DevChannel* __restrict__ chn0 = p.chans + blk0.channel;
    const int arms0 = chn0->arms;
    int nent[LA];
    const void* pre = kF16 ? (const void*)chn0->tabh : (const void*)chn0->tabf;
    // pre-interleaved copy usable as is: the same pitch AND the same layout - a derived channel's f32 image is {arm 0, arm 1,
    // arm 1 * (-1)^entry, 0}, which has the pitch of a genuine three-arm image {arm 0, arm 1, arm 2, 0}: the host routes such
    // channels to the DER instantiation (validate_blocks), this check makes a mix-up impossible instead of silent
    bool plain = pre != nullptr && (kF16 ? chn0->tabh_ap : chn0->tabf_ap) == AP && (kF16 || (chn0->derived != 0) == DER);
#pragma unroll
    for (int a = 0; a < LA; ++a) {
      nent[a] = 0;
      if (a < arms0) {
        const int off = blk0.table_offset[a];
        nent[a] = min(chn0->stage_len[a], chn0->nent[a] - off);
        maxn = max(maxn, nent[a]);
        plain &= off == 0 && chn0->stage_len[a] == chn0->nent[a];
        windowed |= chn0->stage_len[a] != chn0->nent[a];
      }
    }
    if (plain) {
      const uint4* __restrict__ src = reinterpret_cast<const uint4*>(pre);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      const int n16 = (kF16 ? chn0->tabh_bytes : chn0->tabf_bytes) >> 4;
#pragma unroll 4
      for (int i = threadIdx.x; i < n16; i += (int)blockDim.x) dst[i] = src[i];
    } else {
      tab_t* wtab = reinterpret_cast<tab_t*>(smem);
      const int total = maxn + 2 * kGuard;
      for (int i = threadIdx.x; i < total; i += (int)blockDim.x) {
        const int e = i - kGuard;
#pragma unroll
        for (int a = 0; a < AP; ++a) {
          float v = 0.0f;
          if (a < LA) {
            const int aa = a < LA ? a : 0;
            if (a < arms0 && e >= 0 && e < nent[aa]) v = (float)chn0->tab[aa][blk0.table_offset[aa] + e];
          } else if (PN && a == LA) {  // arm LA - 1 times (-1)^entry
            if (e >= 0 && e < nent[LA - 1]) v = (float)chn0->tab[LA - 1][blk0.table_offset[LA - 1] + e] * ((e & 1) ? -1.0f : 1.0f);
          }
          wtab[i * AP + a] = (tab_t)v;
        }
      }
    }
    __syncthreads();  // the only barrier
  };
  {
    const long long lb0 = min(DEVLOOP ? wq : wave_items ? (wg * kLW) / nsplit : grp * p.bpw * p.stride + cslot, (long long)p.nblocks - 1);
    stage_tables(DEVLOOP ? p.devloop->chan[lb0].blk : CL ? load_block(p, lb0) : p.blocks[lb0]);
  }
  if (wave_items && wq >= p.nblocks) return;

  // DEVLOOP scratch behind the tables: float red[16][GC_OUT_STRIDE] | gc_block + status | double dred[64]
  gc_block* sblk = reinterpret_cast<gc_block*>(smem + p.red_off + kLW * GC_OUT_STRIDE * sizeof(double));
  int* sstatus = reinterpret_cast<int*>(sblk + 1);
  double* dred = reinterpret_cast<double*>(smem + p.red_off + kLW * GC_OUT_STRIDE * sizeof(double) + 128);
  const int nloop = DEVLOOP ? p.devloop->n_epochs : p.bpw;
  DevLoopChan dl_st;  // DEVLOOP closer (member 0, wave 0): the channel's loop state, in registers across the epochs
  if constexpr (DEVLOOP) {
    if (member == 0 && wave == 0) dl_st = p.devloop->chan[min(wq, (long long)p.nblocks - 1)];
  }
  (void)dl_st;
  for (int bi = (DEVLOOP || wave_items) ? 0 : wave; bi < nloop; bi += (DEVLOOP || wave_items) ? 1 : kLW) {
  const long long lb = DEVLOOP ? wq : (grp * p.bpw + bi) * p.stride + cslot;
  if (lb >= p.nblocks) break;
  gc_block blk;
  if constexpr (DEVLOOP) {
    if (wave == 0 && !(member == 0 && bi > 0)) {
      // wave 0 polls the ten descriptor messages (one per lane) and hands the descriptor to the workgroup through LDS
      // host-fed run (DevLoopArgs::host_loop): the first descriptor comes from host memory to member 0, which relays it
      const bool from_host = p.devloop->host_loop && member == 0;
      const msg_t* dm = (from_host ? p.devloop->host_desc : p.devloop->desc_msg) + lb * kDescWords;
      msg_t m = {0u, 0u, 0u, 0u};
      unsigned int spins = 0;
      while (true) {
        if (lane < kDescWords) m = msg_load(dm + lane);
        const bool ok = lane >= kDescWords || m.z == (unsigned int)bi + 1u || m.z == 0xffffffffu;  // 0xffffffff: stop, any epoch
        if (__all(ok)) break;
        if (++spins > (1u << 22)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (from_host && lane < kDescWords && spins <= (1u << 22)) msg_store(p.devloop->desc_msg + lb * kDescWords + lane, m);
      union {
        gc_block b;
        unsigned long long q[sizeof(gc_block) / 8];
      } u;
#pragma unroll
      for (int i = 0; i < (int)(sizeof(gc_block) / 8); ++i)
        u.q[i] = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)m.y, i) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)m.x, i);
      int st = __builtin_amdgcn_readlane((int)m.x, kDescWords - 1);
      if (spins > (1u << 22)) st = 3;
      if (lane == 0) {
        *sblk = u.b;
        *sstatus = st;
        if (st == 3) p.devloop->chan[lb].status = 3;
      }
    }
    __syncthreads();
    if (*sstatus != 0) break;  // uniform over the workgroup: record exhausted / timed out
    blk = *sblk;
    // every wave is past the barrier above: nobody reads the old window any more (host-fed runs stage their first window here too:
    // the team's initial block carries no table offsets)
    if (windowed && (bi > 0 || p.devloop->host_loop)) stage_tables(blk);
  } else {
    blk = CL ? load_block(p, lb) : p.blocks[lb];
  }
  const DevChannel* __restrict__ chn = p.chans + blk.channel;
  const int arms_here = chn->arms;

  // ---- per-block uniform quantities ----------------------------------------------------------------
  const double R = chn->index_scale;
  const double M = chn->mult[0];
  const double rem = blk.rem_code_phase;
  const double step = blk.code_phase_step;
  const double d = blk.el_spacing;
  const int N = blk.blksize;
  const long long s0 = blk.first_sample;
  // colon() arguments exactly as the reference writes them (tracking.m:252-268; GAL_E1C tracking.m:236-262
  // for R = 2); x*1.0 is exact so R = 1 needs no special case.
  // block-uniform float64 values live in scalar registers (there is no scalar FP64 unit: computed on the VALU they would
  // each occupy a VGPR pair for the whole block, next to the 6*ARMS accumulators and the ramp state)
  auto uni = [](double v) __attribute__((always_inline)) -> double {
    const unsigned long long u = __double_as_longlong(v);
    return __longlong_as_double(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                (unsigned int)__builtin_amdgcn_readfirstlane((int)u));
  };
  const double aE = uni((rem - d) * R);
  const double aL = uni((rem + d) * R);
  const double aP = uni(rem * R);
  const double sp = uni(step * R);
  const double tau = uni(blk.carr_freq / p.fs);  // carrier turns per sample
  const double M6 = DER ? chn->mult[ARMS - 1] : 0.0;  // ramp multiplier of the derived arm
  const bool tie_free = (blk.reserved & 1) != 0;  // host-proved: no sample within the window of a table edge, on any ramp of the
                                                  // channel (gc_mark_tie_free searches the derived arm's six-times ramp too)

  // sample range of this split: a multiple of 64 samples per split
  const int per = (((N + nsplit - 1) / nsplit) + 63) / 64 * 64;
  const int ibeg = split * per;
  const int iend = min(N, ibeg + per);
#ifdef GC_LANE_FORCE_EXACT
  const unsigned int tie_e = 0x7fffffffu;
  (void)gc_tie_window_units(0.0, 0);
#else
  const unsigned int tie_e = gc_tie_window_units((fabs(aE) + fabs(aL) + (double)N * fabs(sp) + 1.0) * fmax(fabs(M), fabs(M6)), (per >> 6) + 1);
#endif
  // the derived arm reads its position out of the base ramp's fraction times six: so its window is that ramp's, times six
  const unsigned int tie_e6 = DER ? gc_tie_window_units6((fabs(aE) + fabs(aL) + (double)N * fabs(sp) + 1.0) * fmax(fabs(M), fabs(M6)), (per >> 6) + 1, M6) : 0u;
  (void)tie_e6;

  // Per-sample ramp step sp*M as a 64.64 fixed-point number (exact: a double has at most 64 fractional bits
  // here); one step of a lane = 64 samples = that number << 6, rounded to 32 fractional bits for Q.
  // corrQ: what kLaneReseedSteps steps of the rounded increment dQ miss of the exact advance, in 2^-32 units (block-uniform)
  auto drift_correction = [](unsigned long long sf, int si, unsigned long long dq) __attribute__((always_inline)) -> unsigned long long {
    const unsigned __int128 adv = (unsigned __int128)sf * (unsigned int)(64 * kLaneReseedSteps);  // fraction part, 64 fractional bits
    const unsigned long long fr = (unsigned long long)adv;
    const long long whole = (long long)si * (64 * kLaneReseedSteps) + (long long)(unsigned long long)(adv >> 64);
    const unsigned long long exact = ((unsigned long long)whole << 32) + (fr >> 32) + ((fr >> 31) & 1ull);
    return exact - dq * (unsigned long long)kLaneReseedSteps;  // mod 2^64: a small signed number
  };
  unsigned long long Sf, dQ;
  int Si;
  int el_off = 0;
  // Carrier.  exp(-i*theta) of sample ibeg + lane + 64*s factors into the lane's phasor at its first sample, the BLOCK-UNIFORM
  // rho_j = exp(-i*2*pi*64*j*tau) of step j inside a run of kLaneReseedSteps steps, and P^g (P = rho of a whole run) for run g.
  // rho_j lives in a table of this wave in LDS (filled here, read as a broadcast: no VALU work per sample beyond the product
  // x * rho_j itself); the sums run Horner-style - when run g >= 1 starts the accumulators turn by conj(P) - and are turned
  // once, at the end of the block, by the lane's exact float64-reduced phasor.  (Until round 3 every lane carried its own
  // phasor and rotated it by one step per sample: four more VALU instructions per sample and a recurrence to re-seed.)
  float2* const rho = reinterpret_cast<float2*>(smem + p.rho_off) + wave * kRhoPerWave;
  float turnC, turnS;  // conj(P) = turnC + i*turnS
  {
    const int nst = min((iend - ibeg + 63) >> 6, kLaneReseedSteps);  // steps this wave walks (uniform), at most one run's worth
    for (int j = lane; j < nst; j += 64) {
      const double x = (double)(64 * j) * tau;
      float s_, c_;
      sincospif(2.0f * (float)(x - floor(x)), &s_, &c_);  // range reduction in double, sincos in float
      rho[j] = make_float2(c_, s_);
    }
    const double xr = (double)(64 * kLaneReseedSteps) * tau;
    float s_, c_;
    sincospif(2.0f * (float)(xr - floor(xr)), &s_, &c_);
    turnC = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c_)));
    turnS = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_)));
    __builtin_amdgcn_wave_barrier();  // the table is this wave's own: its LDS operations execute in order, the compiler must not move the reads up
  }
  {
    const double y = sp * M;
    const double yi = floor(y);
    Sf = frac_to_u64(y - yi);
    Sf = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)(Sf >> 32)) << 32) |
         (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)Sf);
    Si = __builtin_amdgcn_readfirstlane((int)yi);
    const unsigned long long df = Sf << 6;  // fraction of the 64-sample step, 64 fractional bits
    const long long di = (long long)Si * 64 + (long long)(Sf >> 58);
    dQ = ((unsigned long long)di << 32) + (df >> 32) + ((df >> 31) & 1ull);
    (void)el_off;  // HALF: host-checked 2*d*R*M == 1 exactly (gc_block_shares_el_lane)
  }
  auto uni_u64 = [](unsigned long long u) __attribute__((always_inline)) -> unsigned long long {
    return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
           (unsigned int)__builtin_amdgcn_readfirstlane((int)u);
  };
  const unsigned long long corrQ = uni_u64(drift_correction(Sf, Si, dQ));

  float accr[ARMS][3], acci[ARMS][3];
#pragma unroll
  for (int a = 0; a < ARMS; ++a)
#pragma unroll
    for (int x = 0; x < 3; ++x) accr[a][x] = acci[a][x] = 0.0f;
  // this wave's float64 totals of the block: component v (I_E, Q_E, I_P, Q_P, I_L, Q_L per arm) in LDS, owned by lane v
  double* const totw = reinterpret_cast<double*>(smem + p.red_off) + wave * GC_OUT_STRIDE;
  if (lane < GC_OUT_STRIDE) totw[lane] = 0.0;
  __builtin_amdgcn_wave_barrier();

  int i = ibeg + lane;  // this lane's samples: i, i + 64, i + 128, ...
  {  // every lane walks the block (one past its last sample adds nothing): the flushes below reduce across all 64 lanes
    // Ramp state of this lane's first sample: the block-uniform value at sample ibeg from the reference's
    // doubles ((a + ibeg*d) * M), advanced by `lane` steps in exact fixed-point arithmetic, then narrowed to
    // Q = floor(t * 2^32) + 2^32 - 1, whose high word is ceil(t) unless t is within 2^-32 of an integer.
    unsigned long long Q[NT];
    {
      const double isp = __dmul_rn((double)ibeg, sp);
      const double base3[3] = {__dmul_rn(__dadd_rn(aE, isp), M), __dmul_rn(__dadd_rn(aP, isp), M),
                               __dmul_rn(__dadd_rn(aL, isp), M)};
      const unsigned __int128 prod = (unsigned __int128)Sf * (unsigned int)lane;
      const unsigned long long F = (unsigned long long)prod;
      const int I = lane * Si + (int)(unsigned long long)(prod >> 64);
#pragma unroll
      for (int x = 0; x < NT; ++x) {
        const Fx f0 = to_fx(base3[x]);
        const unsigned long long g = f0.G - F;  // t = k - g / 2^64
        const int k = f0.k0 + I + (f0.G < F ? 1 : 0);
        const unsigned long long gc = (g >> 32) + (((unsigned int)g != 0u) ? 1ull : 0ull);  // ceil(g / 2^32) <= 2^32
        const unsigned long long q = ((unsigned long long)(unsigned int)k << 32) + 0xffffffffull - gc;
        Q[x] = q;
      }
    }
    // The three taps' ramps differ by block-uniform amounts (the same lane offset was subtracted from each): ONE ramp per
    // set stays in vector registers, the prompt and late values are that plus a scalar pair
    auto uni64 = [](unsigned long long u) __attribute__((always_inline)) -> unsigned long long {
      return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
             (unsigned int)__builtin_amdgcn_readfirstlane((int)u);
    };
    unsigned long long Q0 = Q[0];
    const unsigned long long dTap[3] = {0ull, uni64(Q[NT > 1 ? 1 : 0] - Q[0]), uni64(Q[NT - 1] - Q[0])};
    const int i_first = i;
    int nturn = 0;  // runs completed so far (uniform): the accumulators have been turned by conj(P) that often
    const uint8_t* ptr = p.if_base + (long long)bps * (s0 + i);
    // one sample as it lies in the record; 16-bit samples stay 16-bit values all the way into the SDWA converts (which select
    // bytes 0 and 1 themselves): as 32-bit (or 16-bit integer) values the compiler zero-extended those that cross the loop's back edge, a v_and per sample
    typedef typename std::conditional<bps == 2, _Float16, typename std::conditional<bps == 4, unsigned int, unsigned char>::type>::type word_t;  // _Float16: 16 bits without an integer's extension rules
    auto load_sample = [](const uint8_t* q) __attribute__((always_inline)) -> word_t { return *reinterpret_cast<const word_t*>(q); };
    auto sample_of = [](word_t word, float& a, float& b) __attribute__((always_inline)) {
      float x0, x1 = 0.0f;
      if constexpr (MODE == I8_IQ || MODE == I8_QI) {
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(x0) : "v"(word));
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(x1) : "v"(word));
      } else if constexpr (MODE == I16_IQ || MODE == I16_QI) {
        x0 = cvt_half<0>(word);
        x1 = cvt_half<1>(word);
      } else if constexpr (MODE == I8_REAL) {
        x0 = cvt_byte<0>((unsigned int)word);
      } else {
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(x0) : "v"(word));
      }
      a = Fmt<MODE>::swap ? x1 : x0;
      b = Fmt<MODE>::swap ? x0 : x1;
    };
    // y = x * rho_j, rho_j = wc - i*ws
    auto mix = [&](word_t word, float2 r, float& yr, float& yi) __attribute__((always_inline)) {
      float a, b;
      sample_of(word, a, b);
      const float wc = r.x, ws = r.y;
      yr = kReal ? a * wc : fmaf(a, wc, b * ws);
      yi = kReal ? -a * ws : fmaf(b, wc, -a * ws);
    };
    // a run of kLaneReseedSteps steps is complete: the accumulators turn by conj(P), the ramps' rounding drift goes out
    // (corr_common.h: it keeps the near-tie window, and with it the share of blocks that need the test at all, independent of
    // the block length)
    auto turn = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < ARMS; ++a)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const float nr = fmaf(accr[a][x], turnC, -(acci[a][x] * turnS));
          const float ni = fmaf(accr[a][x], turnS, acci[a][x] * turnC);
          accr[a][x] = nr;
          acci[a][x] = ni;
        }
      ++nturn;
    };
    // empty the lane sums into the wave's float64 totals: turn them by the lane's own phasor - exp(-i*theta) at its first sample
    // of the current run, from the exact float64 phase -, add the 64 lanes (one transposing reduction for all components), the lane that ends up with component v adds it in float64
    auto flush = [&]() __attribute__((always_inline)) {
      const double ph = blk.rem_carr_phase * 0.15915494309189535 + (double)(i_first + 64 * kLaneReseedSteps * nturn) * tau;
      float wc, ws;
      sincospif(2.0f * (float)(ph - floor(ph)), &ws, &wc);
      float turned[ARMS * 6];
#pragma unroll
      for (int a = 0; a < ARMS; ++a)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          turned[a * 6 + 2 * x] = fmaf(accr[a][x], wc, acci[a][x] * ws);
          turned[a * 6 + 2 * x + 1] = fmaf(acci[a][x], wc, -(accr[a][x] * ws));
          accr[a][x] = 0.0f;
          acci[a][x] = 0.0f;
        }
      const float mine = wave_transpose_sum<ARMS * 6>(turned, lane);  // the wave's total of component `slot` in this lane
      const int slot = wave_transpose_slot(lane);
      // lanes L and L ^ 32 hold the same slot and the same total after the transposing sum: one of them owns the float64 add
      // (both doing it is the same read-modify-write of one address, right only while the wave runs in lock step)
      if (slot < ARMS * 6 && lane < 32) totw[slot] += (double)mine;
    };
    // returns the entry of the last LDS arm (the one a derived arm is built from)
    auto accumulate = [&](int x, int k, float yr, float yi) __attribute__((always_inline)) -> float {
      if constexpr (AP == 1) {
        const float cf = (float)tab[kGuard + k];
        accr[0][x] = fmaf(cf, yr, accr[0][x]);
        acci[0][x] = fmaf(cf, yi, acci[0][x]);
        return cf;
      } else {
        typedef tab_t vec_t __attribute__((ext_vector_type(AP)));
        const vec_t e = reinterpret_cast<const vec_t*>(tab)[kGuard + k];
#pragma unroll
        for (int ar = 0; ar < LA; ++ar) {
          const float cf = (float)e[ar];
          accr[ar][x] = fmaf(cf, yr, accr[ar][x]);
          acci[ar][x] = fmaf(cf, yi, acci[ar][x]);
        }
        return (float)e[PN ? LA : LA - 1];  // what a derived arm is built from: the last table arm, PN: times (-1)^entry
      }
    };
    // derived arm: padded-table entry k6 of the six-times-faster replica = entry p = (k6 + 5) / 6 of arm LA - 1 with the sign
    // (-1)^(p + k6); the quotient by a float reciprocal (exact for k6 < 2^21)
    auto accumulate_derived = [&](int x, int k6, float yr, float yi) __attribute__((always_inline)) {
      const int pidx = (int)(((float)(k6 + 5) + 0.5f) * 0.16666667f);
      const unsigned int sgn = ((unsigned int)(pidx + k6) & 1u) << 31;
      const float cf = __uint_as_float(__float_as_uint((float)tab[(kGuard + pidx) * AP + (LA - 1)]) ^ sgn);
      accr[ARMS - 1][x] = fmaf(cf, yr, accr[ARMS - 1][x]);
      acci[ARMS - 1][x] = fmaf(cf, yi, acci[ARMS - 1][x]);
    };
    // lean accumulate of one sample: y = x*exp(-i theta), one LDS read per tap serves every arm.  `lo` (HALF only): low word of
    // the early ramp's Q, whose sign bit says whether the prompt tap sits on the early (0) or the late (1) entry
    auto lean_sample = [&](word_t word, float2 r, const int (&k)[NT], const unsigned int (&lo)[NT]) {
      float yr, yi;
      mix(word, r, yr, yi);
      if constexpr (kHalf) {
        typedef tab_t vec_t __attribute__((ext_vector_type(AP)));
        const vec_t* tv = reinterpret_cast<const vec_t*>(tab) + (kGuard + k[0]);
        const vec_t e0 = tv[0], e1 = tv[1];        // early and late entries of every arm: one ds_read2
        const bool on_late = (int)lo[0] < 0;
#pragma unroll
        for (int ar = 0; ar < LA; ++ar) {
          float cE, cL;
          if constexpr (AP == 1) {
            cE = (float)e0[0];
            cL = (float)e1[0];
          } else {
            cE = (float)e0[ar];
            cL = (float)e1[ar];
          }
          const float cP = on_late ? cL : cE;
          accr[ar][0] = fmaf(cE, yr, accr[ar][0]);
          acci[ar][0] = fmaf(cE, yi, acci[ar][0]);
          accr[ar][1] = fmaf(cP, yr, accr[ar][1]);
          acci[ar][1] = fmaf(cP, yi, acci[ar][1]);
          accr[ar][2] = fmaf(cL, yr, accr[ar][2]);
          acci[ar][2] = fmaf(cL, yi, acci[ar][2]);
        }
        return;
      } else {
      const int kk[3] = {k[0], k[NT > 1 ? 1 : 0], k[NT - 1]};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const float base = accumulate(x, kk[x], yr, yi);
        if constexpr (DER) {
          // Away from ties (the caller redoes those exactly) ceil(ceil(6t) / 6) == ceil(t): the entry p the derived arm needs is
          // the one this tap just read, and only the sign (-1)^(p + k6) is left.  With t = n + f (0 < f < 1): p = n + 1,
          // k6 = 6n + floor(6f) + 1, so the sign is (-1)^(p + 1 + floor(6f)), and floor(6f) is odd exactly where frac(3f) >= 1/2 -
          // bit 31 of three times the ramp's low word (which holds f * 2^32 - 1).  No ramp at six times the rate, no second index:
          // one multiply-by-three, (PN: the table's third column carries (-1)^p already; else one add puts p's parity on that
          // bit) and one three-input bit operation per tap.
          const unsigned int lx = lo[NT == 3 ? x : 0];
          unsigned int t3 = lx + (lx << 1);
          if constexpr (!PN) t3 += (unsigned int)kk[x] << 31;
          const float cf = __uint_as_float(__float_as_uint(base) ^ (~t3 & 0x80000000u));
          accr[ARMS - 1][x] = fmaf(cf, yr, accr[ARMS - 1][x]);
          acci[ARMS - 1][x] = fmaf(cf, yi, acci[ARMS - 1][x]);
        }
      }
      }
    };
    // exact accumulate of one sample: MATLAB colon element i (tracking.m:252-270) in float64 — forwards from
    // a for the first half, backwards from the end point b for the second, mean of both in the exact middle
    auto exact_sample = [&](word_t word, float2 r, int is) __attribute__((always_inline)) {
      float yr, yi;
      mix(word, r, yr, yi);
      // colon() end points b = ((N-1)*step + rem -/+ d) * R, evaluated in the reference's order
      const double nm1s = __dmul_rn((double)(N - 1), step);
      const double bP = __dmul_rn(__dadd_rn(nm1s, rem), R);
      const double bE = __dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), -d), R);
      const double bL = __dmul_rn(__dadd_rn(__dadd_rn(nm1s, rem), d), R);
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const double ax = (x == 0) ? aE : (x == 1) ? aP : aL;
        const double bx = (x == 0) ? bE : (x == 1) ? bP : bL;
        double t;
        if (2 * is < N - 1)
          t = __dadd_rn(ax, __dmul_rn((double)is, sp));
        else if (2 * is > N - 1)
          t = __dadd_rn(bx, -__dmul_rn((double)(N - 1 - is), sp));
        else
          t = __dadd_rn(ax, bx) / 2.0;
        const int kx = (int)fmin(fmax(ceil(__dmul_rn(t, M)), (double)-kGuard), (double)(maxn + kGuard - 1));
        accumulate(x, kx, yr, yi);
        if constexpr (DER) {
          const int k6 = (int)fmin(fmax(ceil(__dmul_rn(t, M6)), 0.0), (double)(6 * (maxn - 2) + 1));
          accumulate_derived(x, k6, yr, yi);
        }
      }
    };
    // ramp step of one tap: table index of the CURRENT sample (high word), the low word into the running
    // min / max of the near-tie test, then Q += dQ
    // ramp step of all taps: table index of the CURRENT sample (high word), the low word into the running min / max of the
    // near-tie test, then Q0 += dQ
    auto ramp_step = [&](int (&k)[NT], bool test, unsigned int& dmin, unsigned int& dmax, unsigned int (&lo)[NT], unsigned int& dmin6,
                         unsigned int& dmax6) __attribute__((always_inline)) {
#pragma unroll
      for (int x = 0; x < NT; ++x) {
        const unsigned long long q = x == 0 ? Q0 : Q0 + dTap[NT == 3 ? x : 0];
        k[x] = (int)(unsigned int)(q >> 32);
        const unsigned int l = (unsigned int)q;
        lo[x] = l;
        if (test) {
          dmin = min(dmin, l);
          dmax = max(dmax, l);
          if constexpr (kHalf) {  // the prompt ramp crosses an entry where the early ramp's fraction passes 1/2
            dmin = min(dmin, l ^ 0x80000000u);
            dmax = max(dmax, l ^ 0x80000000u);
          }
          if constexpr (DER) {    // the derived arm crosses a sub-entry where six times the fraction passes an integer
            const unsigned int l6 = (l + (l << 1)) << 1;
            dmin6 = min(dmin6, l6);
            dmax6 = max(dmax6, l6);
          }
        }
      }
      Q0 += dQ;
      asm volatile("" : "+v"(Q0));  // keep the ramp a chain of adds: Q + j*dQ from precomputed multiples costs a register pair per j
    };

    // One group of GRP steps per lane.  TF (the block is tie-free, host-proved by gc_mark_tie_free's exact search): no
    // near-tie test and no exact path, so the accumulators never meet a control-flow join inside the loop.  Otherwise the
    // ramp stage tests the whole group: a sample lies within e chips of a table edge iff the low word of its Q is within
    // e*2^32 of 0 (mod 2^32).  Not measure-zero: with remCodePhase = 0 and the nominal code rate (every channel's first
    // block, tracking.m:163-165) 1.023e6/18e6 is rational and samples 3000k land exactly on edges.  Any suspect lane
    // sends the wave's group through the exact path.
    auto group = [&](const word_t (&cur)[GRP], const float2 (&r)[GRP], auto tf) {
      constexpr bool TF = decltype(tf)::value;
      int kg[GRP][NT];
      unsigned int lo[GRP][NT];
      unsigned int dmin = 0xffffffffu, dmax = 0u, dmin6 = 0xffffffffu, dmax6 = 0u;
#pragma unroll
      for (int j = 0; j < GRP; ++j) ramp_step(kg[j], !TF, dmin, dmax, lo[j], dmin6, dmax6);
      bool exact = false;
      if constexpr (!TF) {
        bool near = (dmin <= tie_e) | (dmax >= 0u - tie_e - 1u);
        if constexpr (DER) near |= (dmin6 <= tie_e6) | (dmax6 >= 0u - tie_e6 - 1u);
        exact = __any(near) != 0;
      }
      if (__builtin_expect(exact, 0)) {
        // one copy of the exact path, walked GRP times; the step's sample and rho are picked by compare-and-select (an array
        // indexed by the loop counter would live in scratch memory - and be stored there by every group of every block)
#pragma unroll 1
        for (int j = 0; j < GRP; ++j) {
          word_t w = cur[0];
          float2 rr = r[0];
#pragma unroll
          for (int q = 1; q < GRP; ++q) {
            w = (j == q) ? cur[q] : w;
            rr.x = (j == q) ? r[q].x : rr.x;
            rr.y = (j == q) ? r[q].y : rr.y;
          }
          exact_sample(w, rr, i + j * 64);
        }
      } else {
#pragma unroll
        for (int j = 0; j < GRP; ++j) lean_sample(cur[j], r[j], kg[j], lo[j]);
      }
      i += GRP * 64;
      ptr += (long long)GRP * bps * 64;
    };
    // rho of GRP consecutive steps from step j0 (a multiple of GRP) of the current run: 16-byte broadcast reads
    auto load_rho = [&](float2 (&r)[GRP], int j0) __attribute__((always_inline)) {
      static_assert(GRP % 2 == 0, "two table entries per read");
      const float4* r4 = reinterpret_cast<const float4*>(rho + j0);
#pragma unroll
      for (int j = 0; j < GRP / 2; ++j) {
        const float4 v = r4[j];
        r[2 * j] = make_float2(v.x, v.y);
        r[2 * j + 1] = make_float2(v.z, v.w);
      }
    };
    auto load_group = [&](word_t (&dst)[GRP], int ahead) {
#pragma unroll
      for (int j = 0; j < GRP; ++j) dst[j] = load_sample(ptr + (long long)(ahead * GRP + j) * bps * 64);
    };
    // Whole groups first.  Their number is the same for every lane of the wave (the split's range starts on a multiple of
    // 64 samples; only its last 64-sample step can be partial), so the loop control and the prefetch of the next group are
    // scalar: no exec masking around the loads, no copies of the sample registers between the two buffers.
    const int groups = __builtin_amdgcn_readfirstlane(((iend - ibeg) >> 6) / GRP);
    // two sample buffers, one loop exit (exits from the middle of the pair made the compiler copy all accumulators into
    // the registers the other exit expected, every group)
    static_assert(kLaneReseedSteps % (2 * GRP) == 0, "a run is a whole number of group pairs");
    constexpr int kPairs = kLaneReseedSteps / (2 * GRP);
    static_assert((kPairs & (kPairs - 1)) == 0, "pairs per run: a power of two");
    // the ramps' rounding drift goes out where a run ends; the accumulators turn there
    auto end_of_run = [&]() __attribute__((always_inline)) {
      if (kFlushRuns > 0 && (nturn + 1) % kFlushRuns == 0) {
        flush();   // (the emptied sums need no turn)
        ++nturn;
      } else {
        turn();
      }
      Q0 += corrQ;
    };
    // Two sample buffers; the loads of a group fly while the group before it is processed.  The prefetch inside the loop is
    // unconditional and the last pair runs behind the loop without one: a prefetch under a condition made the two paths meet
    // with different numbers of loads in flight, the compiler then waited for ALL of them (s_waitcnt vmcnt(0)) before the
    // first use - the prefetch had never been ahead of anything.
    auto main_loop = [&](auto tf) __attribute__((always_inline)) {
      word_t xa[GRP], xb[GRP];
      float2 ra[GRP], rb[GRP];
      load_group(xa, 0);
      const int pairs = groups >> 1;
      auto pair_body = [&](int pp, auto more) __attribute__((always_inline)) {
        const int j0 = (pp & (kPairs - 1)) * 2 * GRP;
        if (pp > 0 && j0 == 0) end_of_run();
        load_group(xb, 1);
        load_rho(ra, j0);
        load_rho(rb, j0 + GRP);
        group(xa, ra, tf);
        if constexpr (decltype(more)::value) load_group(xa, 1);
        group(xb, rb, tf);
      };
      const bool odd = (groups & 1) != 0;
      for (int pp = 0; pp < pairs - 1; ++pp) pair_body(pp, std::true_type{});
      if (pairs > 0) {
        if (odd) pair_body(pairs - 1, std::true_type{});
        else pair_body(pairs - 1, std::false_type{});
      }
      if (odd) {
        const int st = (groups - 1) * GRP;
        if (st > 0 && (st & (kLaneReseedSteps - 1)) == 0) end_of_run();
        load_rho(ra, st & (kLaneReseedSteps - 1));
        group(xa, ra, tf);
      }
    };
    if (groups > 0) {
      if (tie_free) main_loop(std::true_type{});
      else main_loop(std::false_type{});
    }
    // tail: fewer than GRP steps left for this wave (the step counter is uniform: a lane past its last sample adds nothing)
    for (int st = groups * GRP; ibeg + 64 * st < iend; ++st) {
      if (st > 0 && (st & (kLaneReseedSteps - 1)) == 0) end_of_run();
      const float2 r1 = rho[st & (kLaneReseedSteps - 1)];
      if (i < iend) {
        const word_t word = load_sample(ptr);
        int k1[NT];
        unsigned int lo1[NT];
        unsigned int dmin = 0xffffffffu, dmax = 0u, dmin6 = 0xffffffffu, dmax6 = 0u;
        ramp_step(k1, true, dmin, dmax, lo1, dmin6, dmax6);
        bool near = (dmin <= tie_e) | (dmax >= 0u - tie_e - 1u);
        if constexpr (DER) near |= (dmin6 <= tie_e6) | (dmax6 >= 0u - tie_e6 - 1u);
        if (!tie_free && near)
          exact_sample(word, r1, i);
        else
          lean_sample(word, r1, k1, lo1);
      }
      i += 64;
      ptr += (long long)bps * 64;
    }
    flush();  // what is left in the lane sums
  }

  // ---- the wave's float64 totals sit in LDS, component v owned by lane v: combine / store ------------------
  __builtin_amdgcn_wave_barrier();
  const double* tot_all = reinterpret_cast<const double*>(smem + p.red_off);  // [waves][GC_OUT_STRIDE]
  if constexpr (DEVLOOP) {
    const DevLoopArgs* dl = p.devloop;
    constexpr int NS = ARMS * 6;
    __syncthreads();
    if (wave == 0) {
      const unsigned int tag = (unsigned int)bi + 1u;
      double mine = 0.0;  // lanes 0 .. NS-1: this workgroup's sum of component `lane`
      if (lane < NS)
        for (int w = 0; w < nw; ++w) mine += tot_all[w * GC_OUT_STRIDE + lane];
      msg_t* pm = dl->part_msg + (lb * p.splits) * NS;
      if (member != 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
        if (lane < NS) msg_store(pm + member * NS + lane, msg_t{(unsigned int)bits, (unsigned int)(bits >> 32), tag, 0u});
      } else {
        // the closer: what does not need the sums first (the messages are still on their way), then the other workgroups'
        // sums, one message per lane
        const DevLoopPre dl_pre = devloop_pre(dl, dl_st, blk, R);
        const int nmsg = (p.splits - 1) * NS;
        msg_t m = {0u, 0u, 0u, 0u};
        unsigned int spins = 0;
        while (true) {
          if (lane < nmsg) m = msg_load(pm + NS + lane);
          const bool ok = lane >= nmsg || m.z == tag;
          if (__all(ok)) break;
          if (++spins > (1u << 22)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        int st;
        gc_block nxt = blk;
        double dl_rv[GC_TRK_NFIELDS];
        const int dl_arms = arms_here < ARMS ? arms_here : ARMS;
        if (spins > (1u << 22)) {
          st = 3;
        } else {
          dred[lane] = (lane < nmsg) ? __longlong_as_double((long long)(((unsigned long long)m.y << 32) | m.x)) : 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          if (lane < NS)
            for (int k = 0; k < p.splits - 1; ++k) mine += dred[k * NS + lane];
          double sums[NS];
#pragma unroll
          for (int v = 0; v < NS; ++v) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
            sums[v] = __longlong_as_double((long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(bits >> 32), v) << 32) |
                                                       (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)bits, v)));
          }
#pragma unroll
          for (int f = 0; f < GC_TRK_NFIELDS; ++f) dl_rv[f] = 0.0;
          if (dl->host_loop) {
            // host-fed: the team's sums go to the host as tagged records (lane v holds component v), a system fence pushes them
            // out of the L2, and the next descriptor comes back from host memory (tag bi + 2); the relay below is the usual one
            if (lane < NS) {
              TaggedSlot rec;
              rec.value = mine;
              rec.tag = tag;
              rec.zero = 0u;
              *reinterpret_cast<uint4*>(reinterpret_cast<TaggedSlot*>(dl->host_tagged) + lb * GC_OUT_STRIDE + lane) = *reinterpret_cast<const uint4*>(&rec);
            }
            __threadfence_system();
            st = 1;
            if (bi + 1 < nloop) {
              const msg_t* hd = dl->host_desc + lb * kDescWords;
              msg_t hm = {0u, 0u, 0u, 0u};
              unsigned int hs = 0;
              while (true) {
                if (lane < kDescWords) hm = msg_load(hd + lane);
                const bool ok = lane >= kDescWords || hm.z == tag + 1u || hm.z == 0xffffffffu;
                if (__all(ok)) break;
                if (++hs > (1u << 22)) break;
                __builtin_amdgcn_s_sleep(2);
              }
              union {
                gc_block b;
                unsigned long long q[sizeof(gc_block) / 8];
              } hu;
#pragma unroll
              for (int i = 0; i < (int)(sizeof(gc_block) / 8); ++i)
                hu.q[i] = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)hm.y, i) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)hm.x, i);
              nxt = hu.b;
              st = (hs > (1u << 22)) ? 3 : __builtin_amdgcn_readlane((int)hm.x, kDescWords - 1);  // status word of the host's descriptor
            }
          } else {
            st = devloop_post<ARMS>(dl, dl_st, nxt, bi, sums, dl_arms, R, dl_pre, [&](int f, double v) { dl_rv[f] = v; });
          }
        }
        if (lane == 0) {
          *sblk = nxt;
          *sstatus = (st == 1) ? 0 : st;  // 1 = all epochs done: the loop ends by itself
          if (st == 3) dl->chan[lb].status = 3;
        }
        union {
          gc_block b;
          unsigned long long q[sizeof(gc_block) / 8];
        } u;
        u.b = nxt;
        unsigned long long word = (unsigned long long)((st == 2 || st == 3) ? st : 0);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(gc_block) / 8); ++i) word = (lane == i) ? u.q[i] : word;
        if (lane < kDescWords && bi + 1 < nloop)
          msg_store(dl->desc_msg + lb * kDescWords + lane, msg_t{(unsigned int)word, (unsigned int)(word >> 32), tag + 1u, 0u});
        if (st != 3 && !dl->host_loop) devloop_commit(dl, dl->chan + lb, dl_st, lb, bi, dl_rv, dl_arms, lane);  // records and state, off the critical path
      }
    }
    __syncthreads();  // the closer's workgroup waits for the new descriptor; the scratch is free again
  } else if (wg_block) {
    // one block per workgroup: the 16 waves' float64 totals are added in a fixed order
    __syncthreads();
    if (threadIdx.x < GC_OUT_STRIDE) {
      double s = 0.0;
      if ((int)threadIdx.x < arms_here * 6 && (int)threadIdx.x < ARMS * 6)
        for (int w = 0; w < kLW; ++w) s += tot_all[w * GC_OUT_STRIDE + threadIdx.x];
      p.out[lb * GC_OUT_STRIDE + threadIdx.x] = s;
    }
  } else if (CL) {
    // lane v stores total v as its 16-byte tagged record
    TaggedSlot* ts = p.tagged + (lb * p.splits + split) * GC_OUT_STRIDE;
    if (lane < ARMS * 6) {
      TaggedSlot rec;
      rec.value = (lane < arms_here * 6) ? totw[lane] : 0.0;
      rec.tag = p.notify_tag;
      rec.zero = 0u;
      *reinterpret_cast<uint4*>(ts + lane) = *reinterpret_cast<const uint4*>(&rec);
    }
  } else if (lane < GC_OUT_STRIDE) {
    double* o = (p.splits == 1) ? p.out + lb * GC_OUT_STRIDE : p.partial + (lb * p.splits + split) * GC_OUT_STRIDE;
    o[lane] = (lane < ARMS * 6 && lane < arms_here * 6) ? totw[lane] : 0.0;
  }
  }  // bpw loop
}

#ifdef GC_LANE_PROBE
// ISA probe (scripts/lane_probe.sh): only the instantiations whose inner loops are being looked at
template __global__ void corr_epl_lane_kernel<2, I8_IQ, false, 1>(const KArgs, const InlineBlocks);
template __global__ void corr_epl_lane_kernel<3, I8_IQ, false, 0, false, true>(const KArgs, const InlineBlocks);
template __global__ void corr_epl_lane_kernel<2, I8_IQ, false, 1, true>(const KArgs, const InlineBlocks);
template __global__ void corr_epl_lane_kernel<3, I8_IQ, false, 0, true, true>(const KArgs, const InlineBlocks);
}  // namespace
#else
// The 148 instantiations (arms x record format x closed loop x table kind, + the persistent ones) took one compiler process
// 248 s.  The build compiles this file four times instead (cu_sdr_collection_amd/build.py): GC_LANE_PART = 1, 2, 3 hold the
// kernels of one-, two- and three-arm channels behind the gc_lane_part_* entry points below, GC_LANE_PART = 0 the dispatchers
// (no kernel).  Without the macro everything is in one unit (scripts/variants.sh, scripts/lane_probe.sh).
#ifndef GC_LANE_PART
#define GC_LANE_PART (-1)
#endif
#define GC_LANE_HAS(part) (GC_LANE_PART == -1 || GC_LANE_PART == (part))

template <typename K>
void launch_one(gc_context* ctx, K kernel, const KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem) {
  if (smem > 64 * 1024)  // above the default dynamic-LDS limit (gfx950 has 160 KiB per workgroup)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kernel, grid, dim3(kLW * 64), smem, ctx->stream, a, ib);
}

template <int ARMS, int MODE>
void launch_tab(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem, int tabkind) {
  const bool cl = a.tagged != nullptr;
  if (cl) {
    if (tabkind == 0) launch_one(ctx, corr_epl_lane_kernel<ARMS, MODE, true, 0>, a, ib, grid, smem);
    else if (tabkind == 1) launch_one(ctx, corr_epl_lane_kernel<ARMS, MODE, true, 1>, a, ib, grid, smem);
    else launch_one(ctx, corr_epl_lane_kernel<ARMS, MODE, true, 2>, a, ib, grid, smem);
  } else {
    if (tabkind == 0) launch_one(ctx, corr_epl_lane_kernel<ARMS, MODE, false, 0>, a, ib, grid, smem);
    else if (tabkind == 1) launch_one(ctx, corr_epl_lane_kernel<ARMS, MODE, false, 1>, a, ib, grid, smem);
    else launch_one(ctx, corr_epl_lane_kernel<ARMS, MODE, false, 2>, a, ib, grid, smem);
  }
}

int record_mode(const gc_context* ctx) {
  if (ctx->if_dtype == GC_I8) return ctx->if_layout == GC_IQ ? I8_IQ : ctx->if_layout == GC_QI ? I8_QI : I8_REAL;
  return ctx->if_layout == GC_IQ ? I16_IQ : ctx->if_layout == GC_QI ? I16_QI : I16_REAL;
}

template <int ARMS>
int launch_mode(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem, int tabkind) {
  switch (record_mode(ctx)) {
    case I8_IQ: launch_tab<ARMS, I8_IQ>(ctx, a, ib, grid, smem, tabkind); break;
    case I8_QI: launch_tab<ARMS, I8_QI>(ctx, a, ib, grid, smem, tabkind); break;
    case I16_IQ: launch_tab<ARMS, I16_IQ>(ctx, a, ib, grid, smem, tabkind); break;
    case I16_QI: launch_tab<ARMS, I16_QI>(ctx, a, ib, grid, smem, tabkind); break;
    case I8_REAL: launch_tab<ARMS, I8_REAL>(ctx, a, ib, grid, smem, tabkind); break;
    default: launch_tab<ARMS, I16_REAL>(ctx, a, ib, grid, smem, tabkind); break;
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int launch_persistent_fn(gc_context* ctx, const void* fn, KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem, int waves) {
  void* args[2] = {(void*)&a, (void*)&ib};
  if (smem > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  GC_PERSIST(gc_launch_persistent(ctx, fn, grid, dim3(waves * 64), args, (unsigned int)smem));
  return GC_OK;
}

template <int ARMS, int MODE>
int launch_lane_devloop(gc_context* ctx, KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem, bool share, int waves) {
  const void* fn = share ? (const void*)corr_epl_lane_kernel<ARMS, MODE, false, 1, true> : (const void*)corr_epl_lane_kernel<ARMS, MODE, false, 0, true>;
  return launch_persistent_fn(ctx, fn, a, ib, grid, smem, waves);
}

// the persistent instantiations of ARMS-arm channels: f32 tables for every record format, f16 tables (half) for int8 I/Q and Q/I
template <int ARMS>
int launch_devloop_arms(gc_context* ctx, KArgs& a, const InlineBlocks& ib, dim3 g, size_t smem, bool share_el, int waves, bool half) {
  const bool qi = ctx->if_layout == GC_QI;
  if (half)
    return launch_persistent_fn(ctx, qi ? (const void*)corr_epl_lane_kernel<ARMS, I8_QI, false, 2, true> : (const void*)corr_epl_lane_kernel<ARMS, I8_IQ, false, 2, true>,
                                a, ib, g, smem, waves);
  switch (record_mode(ctx)) {
    case I8_IQ: return launch_lane_devloop<ARMS, I8_IQ>(ctx, a, ib, g, smem, share_el, waves);
    case I8_QI: return launch_lane_devloop<ARMS, I8_QI>(ctx, a, ib, g, smem, share_el, waves);
    case I16_IQ: return launch_lane_devloop<ARMS, I16_IQ>(ctx, a, ib, g, smem, share_el, waves);
    case I16_QI: return launch_lane_devloop<ARMS, I16_QI>(ctx, a, ib, g, smem, share_el, waves);
    case I8_REAL: return launch_lane_devloop<ARMS, I8_REAL>(ctx, a, ib, g, smem, share_el, waves);
    default: return launch_lane_devloop<ARMS, I16_REAL>(ctx, a, ib, g, smem, share_el, waves);
  }
}

}  // namespace

// ---- entry points of the parts (one-, two-, three-arm kernels) ----------------------------------------------------------------
int gc_lane_part_mode1(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, int tabkind);
int gc_lane_part_mode2(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, int tabkind);
int gc_lane_part_mode3(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, int tabkind);
int gc_lane_part_devloop1(gc_context* ctx, KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, bool share_el, int waves, bool half);
int gc_lane_part_devloop2(gc_context* ctx, KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, bool share_el, int waves, bool half);
int gc_lane_part_derived(gc_context* ctx, KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, bool half, bool devloop, int waves);

#if GC_LANE_HAS(1)
int gc_lane_part_mode1(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, int tabkind) {
  return launch_mode<1>(ctx, a, ib, dim3(grid), smem, tabkind);
}
int gc_lane_part_devloop1(gc_context* ctx, KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, bool share_el, int waves, bool half) {
  return launch_devloop_arms<1>(ctx, a, ib, dim3(grid), smem, share_el, waves, half);
}
#endif
#if GC_LANE_HAS(2)
int gc_lane_part_mode2(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, int tabkind) {
  return launch_mode<2>(ctx, a, ib, dim3(grid), smem, tabkind);
}
int gc_lane_part_devloop2(gc_context* ctx, KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, bool share_el, int waves, bool half) {
  return launch_devloop_arms<2>(ctx, a, ib, dim3(grid), smem, share_el, waves, half);
}
#endif
#if GC_LANE_HAS(3)
int gc_lane_part_mode3(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, int tabkind) {
  return launch_mode<3>(ctx, a, ib, dim3(grid), smem, tabkind);
}
// three arms, the third derived from the second (int8 I/Q or Q/I records): f32 or f16 (half) tables; launch per call or persistent
int gc_lane_part_derived(gc_context* ctx, KArgs& a, const InlineBlocks& ib, unsigned int grid, size_t smem, bool half, bool devloop, int waves) {
  const bool qi = ctx->if_layout == GC_QI;
  if (devloop) {
    const void* fn = half ? (qi ? (const void*)corr_epl_lane_kernel<3, I8_QI, false, 2, true, true> : (const void*)corr_epl_lane_kernel<3, I8_IQ, false, 2, true, true>)
                          : (qi ? (const void*)corr_epl_lane_kernel<3, I8_QI, false, 0, true, true> : (const void*)corr_epl_lane_kernel<3, I8_IQ, false, 0, true, true>);
    return launch_persistent_fn(ctx, fn, a, ib, dim3(grid), smem, waves);
  }
  const bool cl = a.tagged != nullptr, h = half;
  if (qi) {
    if (cl && h) launch_one(ctx, corr_epl_lane_kernel<3, I8_QI, true, 2, false, true>, a, ib, dim3(grid), smem);
    else if (cl) launch_one(ctx, corr_epl_lane_kernel<3, I8_QI, true, 0, false, true>, a, ib, dim3(grid), smem);
    else if (h) launch_one(ctx, corr_epl_lane_kernel<3, I8_QI, false, 2, false, true>, a, ib, dim3(grid), smem);
    else launch_one(ctx, corr_epl_lane_kernel<3, I8_QI, false, 0, false, true>, a, ib, dim3(grid), smem);
  } else {
    if (cl && h) launch_one(ctx, corr_epl_lane_kernel<3, I8_IQ, true, 2, false, true>, a, ib, dim3(grid), smem);
    else if (cl) launch_one(ctx, corr_epl_lane_kernel<3, I8_IQ, true, 0, false, true>, a, ib, dim3(grid), smem);
    else if (h) launch_one(ctx, corr_epl_lane_kernel<3, I8_IQ, false, 2, false, true>, a, ib, dim3(grid), smem);
    else launch_one(ctx, corr_epl_lane_kernel<3, I8_IQ, false, 0, false, true>, a, ib, dim3(grid), smem);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}
#endif

#if GC_LANE_HAS(0)
// Persistent tracker with device-side loop closure on the lane kernel: grid = channel slots x a.splits member workgroups.
// f32 tables only (<= 96 KiB), int8 I/Q or Q/I records, one or two arms.
int gc_launch_devloop_lane(gc_context* ctx, const KArgs& a_in, unsigned int grid, int max_arms, bool share_el, int waves) {
  KArgs a = a_in;
  InlineBlocks ib;
  std::memset(&ib, 0, sizeof ib);
  const bool der = a.derived != 0 && max_arms == 3;  // third arm derived from the second: two tables in LDS (host-fed runs only)
  const int ap = gc_arm_pitch(der ? 2 : max_arms);
  // a derived arm's f32 image has four values per entry (the third: arm 1 times (-1)^entry, DevChannel::tabf_ap) and may take
  // most of the LDS - a persistent member is alone on its CU anyway; every other f32 table stays below 96 KiB
  const size_t f32_bytes = (((size_t)ctx->max_stage_len + 2 * kGuard) * ((der && GC_LANE_PN != 0) ? 4 : ap) * 4 + 15) / 16 * 16;
  const size_t f16_bytes = (((size_t)ctx->max_stage_len + 2 * kGuard) * ap * 2 + 15) / 16 * 16;
  const bool half_tables = f32_bytes > (der ? 136u : 96u) * 1024;  // BDS B1C: two 20 462-entry arms = 164 KB as f32, 82 KB as f16
  const size_t tab_bytes = half_tables ? f16_bytes : f32_bytes;
  const bool i8c = ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL;  // int8 I/Q or Q/I
  if ((half_tables && (f16_bytes + 4096 + kLW * kRhoPerWave * sizeof(float2) > 160 * 1024 || !i8c)) || (max_arms > 2 && !der) || (der && !i8c)) {
    gc_set_error("device loop on the lane kernel: tables above 156 KiB as f16, f16 tables or a derived arm on a record other than int8 I/Q, "
                 "or three independent arms are not instantiated");
    return GC_E_UNSUPPORTED;
  }
  a.red_off = (int)tab_bytes;
  a.rho_off = (int)(tab_bytes + kLW * GC_OUT_STRIDE * sizeof(double) + 128 + 64 * sizeof(double));  // a multiple of 16
  const size_t smem = (size_t)a.rho_off + (size_t)kLW * kRhoPerWave * sizeof(float2);
  if (waves < 1 || waves > kLW) return GC_E_INVALID;
  if (der) return gc_lane_part_derived(ctx, a, ib, grid, smem, half_tables, true, waves);
  if (max_arms == 1) return gc_lane_part_devloop1(ctx, a, ib, grid, smem, share_el, waves, half_tables);
  return gc_lane_part_devloop2(ctx, a, ib, grid, smem, share_el, waves, half_tables);
}

// share_el: every block of the launch has 2*el_spacing*R*M an exact positive integer
int gc_launch_correlator_lane(gc_context* ctx, const KArgs& a_in, const InlineBlocks& ib, unsigned int grid, int max_arms,
                              bool share_el) {
  KArgs a = a_in;
  const int ap = gc_arm_pitch(a.derived ? 2 : max_arms);  // a derived third arm has no table of its own ...
  const size_t entries = (size_t)ctx->max_stage_len + 2 * kGuard;
  // ... but its f32 image carries a third column, arm 1 times (-1)^entry: four values per entry (DevChannel::tabf_ap), up to 136 KiB
  const size_t f32_bytes = (entries * ((a.derived && GC_LANE_PN != 0) ? 4 : ap) * 4 + 15) / 16 * 16;
  const size_t f16_bytes = (entries * ap * 2 + 15) / 16 * 16;
  int tabkind;
  size_t smem;
  if (f32_bytes <= (a.derived ? 136u : 96u) * 1024) {  // f32 tables: one 16-wave workgroup per CU still fits next to a second one up to 80 KiB
    tabkind = (share_el && !a.derived) ? 1 : 0;
    smem = f32_bytes;
    a.red_off = (int)f32_bytes;
  } else if (f16_bytes + 2048 <= 160 * 1024) {
    tabkind = 2;
    smem = f16_bytes;
    a.red_off = (int)f16_bytes;
  } else {
    gc_set_error("code tables need %zu bytes of LDS (> 160 KiB); set a window with gc_set_code_window", f16_bytes);
    return GC_E_UNSUPPORTED;
  }
  smem += kLW * GC_OUT_STRIDE * sizeof(double);  // the waves' float64 totals (also the cross-wave scratch of the one-block-per-workgroup mode)
  smem = (smem + 15) / 16 * 16;
  a.rho_off = (int)smem;
  smem += (size_t)kLW * kRhoPerWave * sizeof(float2);  // the waves' carrier-step tables (16 KiB)
  if (smem > 160 * 1024) {
    gc_set_error("code tables need %zu bytes of LDS with the kernel's scratch (> 160 KiB); set a window with gc_set_code_window", smem);
    return GC_E_UNSUPPORTED;
  }
  if (a.derived) {
    if (max_arms != 3 || tabkind == 1 || ctx->if_dtype != GC_I8 || ctx->if_layout == GC_REAL) {
      gc_set_error("internal: derived-arm launch with %d arms / table kind %d", max_arms, tabkind);
      return GC_E_INVALID;
    }
    return gc_lane_part_derived(ctx, a, ib, grid, smem, tabkind == 2, false, 0);
  }
  switch (max_arms) {
    case 1: return gc_lane_part_mode1(ctx, a, ib, grid, smem, tabkind);
    case 2: return gc_lane_part_mode2(ctx, a, ib, grid, smem, tabkind);
    default: return gc_lane_part_mode3(ctx, a, ib, grid, smem, tabkind);
  }
}
#endif  // dispatchers

#endif  // GC_LANE_PROBE
