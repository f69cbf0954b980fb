// corr_fast.hip — fast path of the E/P/L correlator for "low-rate" replicas: the table index
// advances by less than one entry over a lane-chunk of SPL samples ((SPL-1)*step*R*M < 1).
//   SPL = 16: GPS L1 C/A (17.6 samples/chip at 18 Msps), GLONASS (23.5)
//   SPL =  8: B1I / E1 BOC(1,1) / B1C BOC(1,1) (8.8), L2C (7.8), and L1 C/A at lower rates
// Same arithmetic contract as corr_kernel.hip (tracking.m:247-300); DESIGN.md 4.1 has the measurements.
//
//   * a lane-chunk sees at most ONE table transition per tap, so the replica over the chunk is c1 before and c2
//     after sample u: the six sums are c1*P_x + c2*(T - P_x) with T = sum_j y_j shared by all taps and arms and
//     P_x = sum_{j <= u_x} y_j;
//   * the running sums P_j are formed by two fused multiply-adds per component and sample and parked in LDS with
//     ds_write_addtid_b32 (2 LDS cycles per wave-store; wider stores made the LDS store path the limit); P_x is
//     one ds_read2_b32 per ramp: no per-sample work per tap, no compare / select / gather per sample;
//   * int8/int16 samples are converted by SDWA sign-extending v_cvt_f32_i32 (one op per component);
//   * every lane of a wave walks the same number of chunks, so loop counter, load base and edge tests are scalar;
//     two word buffers alternate: the next chunk's 16-byte loads are in flight while this one is processed;
//   * LDS tables: float2 {c, dc} (one wave per workgroup), int8 pairs {c, dc} or plain floats c (four waves);
//   * the carrier base rotation is applied Horner-style to the accumulators (acc = acc*conj(rho) + U)
//     and once more at the end with the exact per-thread phase;
//   * the transition position u = g/(step*R*M) is a float quotient; chunks where any u is within
//     4e-6 of an integer (a sample within ~1e-7 chip of a chip edge, where float32 and the reference's
//     float64 rounding could disagree) take the exact double-precision path.
#include <cstdlib>

#include "corr_common.h"
#include "devloop.h"

using namespace gcorr;

namespace {

constexpr float kTieTol = 4e-6f;
constexpr int kFW = 64;  // one wavefront per workgroup
#ifndef GC_SCHED_GROUP
#define GC_SCHED_GROUP 4
#endif

// Running sums of sample j to LDS: ds_write_addtid_b32 (address = M0 + offset + 4*lane, no address VGPR) takes 2 LDS
// cycles per wave-store, half of any other 4-byte store (MI355X_MICROARCH.md, LDS): the stores, not the VALU,
// were this kernel's limit with ds_write2(st64)_b32/b64.
template <int OFF_RE, int OFF_IM>
__device__ __forceinline__ void store_prefix(float tr, float ti, unsigned int lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"
               :
               : "v"(tr), "v"(ti), "s"(lds_base), "n"(OFF_RE), "n"(OFF_IM)
               : "memory", "m0");
}

// Sum of a double over each 32-lane half of the wavefront with DPP row shifts / broadcast (the total of lanes 0..31
// lands in lane 31, of lanes 32..63 in lane 63); the order of the additions is fixed.
__device__ __forceinline__ double half_sum_f64(double v) {
  auto dpp64 = [](double x, auto ctrl, auto row_mask) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  };
  v += dpp64(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
  v += dpp64(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
  v += dpp64(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
  v += dpp64(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
  v += dpp64(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
  return v;
}
__device__ __forceinline__ double rl_f64(double x, int lane) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), lane) << 32) |
                                          (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, lane)));
}

// CL = closed-loop variant: descriptors from the kernel-argument segment, results as host-mapped tagged
// records.  The replay instantiation (CL = false) carries none of that code.
// SHARE = every block of the launch has earlyLateSpc*R*M == 1/2 (host-checked): early and late share one
// ramp and one pair of partial sums.  A separate instantiation per value — both variants inside one kernel measured 25 % slower.
// WIDE = four wavefronts per workgroup sharing ONE staged table kept as int8 pairs {c, dc} (2 bytes per
// entry instead of 8): for tables that would otherwise leave one wave per SIMD (Galileo E1 B+C: 2 x 8186
// entries = 131 KB as float2, 33 KB as int8 pairs).  Every wave owns its own blocks / splits; the only
// barrier is the one after staging.
// DEVLOOP = persistent launch with device-side loop closure (devloop.h): the block loop becomes the epoch loop of ONE
// channel, descriptors come from the channel's device state, sums go to the team's slot array.
template <int ARMS, int MODE, int SPL, bool CL, bool SHARE_EL, int WIDE, bool DEVLOOP = false>
__global__ __launch_bounds__(WIDE != 0 ? 256 : kFW) void corr_epl_fast_kernel(const KArgs p, const InlineBlocks /*read via the segment pointer*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = SPL * Fmt<MODE>::bps / 4;
  constexpr bool kReal = (MODE == I8_REAL || MODE == I16_REAL);
  constexpr int kShift = (SPL == 16) ? 4 : 3;

  long long wg = blockIdx.x;
  if (p.xcd_swizzle && !DEVLOOP) {
    const long long per = (long long)gridDim.x >> 3;  // the host rounds the grid up to a multiple of 8 when swizzling
    wg = (wg & 7) * per + (wg >> 3);
    if (wg >= p.total_wg) return;                     // (at most seven workgroups of the rounded grid)
  }
  // Workgroup -> blocks.  With bpw > 1 (replay lists that interleave `stride` channels epoch by
  // epoch) one workgroup walks bpw consecutive epochs of ONE channel, so the code table is staged
  // into LDS once per bpw blocks; neighbouring workgroups hold the other channels of the same
  // epochs and read the same IF window through the same L2.
  const int lane = threadIdx.x & 63;
  const int wave = WIDE != 0 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;  // wave-uniform (SGPR)
  // WIDE, single block per item (closed loop / small lists): the four waves take four consecutive
  // (block, split) items — the host keeps splits a multiple of 4 so they share block and table;
  // WIDE, periodic replay: the waves interleave over the workgroup's bpw blocks.
  const bool wave_items = WIDE != 0 && p.bpw == 1;
  const long long item = wave_items ? wg * 4 + wave : wg;
  // DEVLOOP with xcd_swizzle: a channel's whole team on ONE XCD (workgroup b runs on XCD b % 8), so that the team's
  // atomics and partial sums meet in that XCD's L2: channel = (b % 8) + 8 * ((b / 8) / splits), split = (b / 8) % splits
  const long long wq = (DEVLOOP && p.xcd_swizzle) ? (item & 7) + 8 * ((item >> 3) / p.splits) : item / p.splits;
  const int split = (DEVLOOP && p.xcd_swizzle) ? (int)((item >> 3) % p.splits) : (int)(item - wq * p.splits);
  const long long grp = wq / p.stride;
  const int cslot = (int)(wq - grp * p.stride);

  // ---- stage {c[k], c[k+1]-c[k]} for k = -1 .. nent (c[-1] := c[0], c[>=nent] := 0), once per
  //      workgroup: the host guarantees that all blocks of a workgroup share channel and offsets
  constexpr bool TABF = (WIDE == 2);  // WIDE with plain float code values c[k] (4 bytes per entry): small tables, no conversions
  float2* tab2[ARMS];
  unsigned short* tab16[ARMS];
  float* tabf[ARMS];
  {
    const long long lb0 = min(grp * p.bpw * p.stride + cslot, (long long)p.nblocks - 1);
    const gc_block blk0 = DEVLOOP ? p.devloop->chan[lb0].blk : CL ? load_block(p, lb0) : p.blocks[lb0];
    const DevChannel* __restrict__ chn0 = p.chans + blk0.channel;
#pragma unroll
    for (int a = 0; a < ARMS; ++a) {
      const int aa = (a < chn0->arms) ? a : 0;
      tab2[a] = reinterpret_cast<float2*>(smem + 8 * (size_t)chn0->lds_off[aa]);
      tab16[a] = reinterpret_cast<unsigned short*>(smem + 2 * (size_t)chn0->lds_off[aa]);
      tabf[a] = reinterpret_cast<float*>(smem + 4 * (size_t)chn0->lds_off[aa]);
      if (a < chn0->arms) {
        const int off = blk0.table_offset[a];
        const int n = min(chn0->stage_len[a], chn0->nent[a] - off);
        // window-relative entry i <-> absolute entry off + i of the pre-differenced table; all loads
        // are independent coalesced reads (one wait), which matters for the closed loop where a
        // launch is only a few microseconds long
        if constexpr (TABF) {
          const float2* __restrict__ src = chn0->tab2[a] + off;
#pragma unroll 4
          for (int i = threadIdx.x; i < n + 3; i += 256) tabf[a][i] = src[i].x;
          if (threadIdx.x == 0) tabf[a][n + 3] = 0.0f;
        } else if constexpr (WIDE != 0) {
          const unsigned short* __restrict__ src = chn0->tab2b[a] + off;
#pragma unroll 4
          for (int i = threadIdx.x; i < n + 3; i += 256) tab16[a][i] = src[i];
        } else {
          const float2* __restrict__ src = chn0->tab2[a] + off;
#pragma unroll 4
          for (int i = lane; i < n + 3; i += kFW) tab2[a][i] = src[i];
        }
      }
    }
    __syncthreads();
  }
  // table entry k (k = -1 .. n+1) as {c[k], c[k+1] - c[k]} (TABF: {c[k], c[k+1]})
  auto table_entry = [&](int ar, int k) -> float2 {
    if constexpr (TABF) {
      return make_float2(tabf[ar][k + 1], tabf[ar][k + 2]);
    } else if constexpr (WIDE != 0) {
      const unsigned int e = tab16[ar][k + 1];
      float c, dc;
      asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(c) : "v"(e));
      asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(dc) : "v"(e));
      return make_float2(c, dc);
    } else {
      return tab2[ar][k + 1];
    }
  };
  // running sums of a lane-chunk, per wave [SPL][re | im][64 lanes] floats
  float* pfx = reinterpret_cast<float*>(smem + p.red_off + (DEVLOOP ? 8 * 3 * 64 : 64)) + (WIDE != 0 ? wave * SPL * 2 * kFW : 0);
  unsigned int pfx_m0 = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)pfx);  // its LDS byte address (low half of the flat address)
  if (wave_items && wq >= p.nblocks) return;

  // DEVLOOP: the launch's constants (loop-filter coefficients, thresholds, buffer addresses: ~60 words) in registers for the whole
  // run.  Read through the pointer, every one of them was a vector load + s_waitcnt vmcnt(0) inside the closure - the compiler cannot
  // know that nothing writes the struct - : some twenty dependent cache round trips per epoch, and a wait that also drains the next
  // epoch's sample fetch (pf_w below).
  [[maybe_unused]] DevLoopArgs dl_loc;
  if constexpr (DEVLOOP) dl_loc = *p.devloop;
  const int nloop = DEVLOOP ? dl_loc.n_epochs : p.bpw;
  gc_block dl_next;  // DEVLOOP: the descriptor this member computed for the next epoch
  (void)dl_next;
  DevLoopChan dl_st;  // DEVLOOP: the channel's loop state, in registers across the epochs (every member keeps its own copy)
  if constexpr (DEVLOOP) dl_st = p.devloop->chan[min(wq, (long long)p.nblocks - 1)];
  (void)dl_st;
  // DEVLOOP: the next epoch's samples, fetched while this epoch's loop is being closed.  The next block starts exactly where this one
  // ends (tracking.m:219-222, 249: absoluteSample advances by blksize) - only its LENGTH waits for the closure -, so the 16-byte
  // chunk every lane will read first is known before the discriminators are: the load goes out right after the team's partial
  // sums have arrived and its ~1.5 us of memory latency pass under the closure instead of in front of the next correlation.
  // pf_off: byte offset the words were read from (per lane), -1: nothing fetched; the next epoch checks it against its own.
  constexpr int kPfWords = SPL * Fmt<MODE>::bps / 4;
  [[maybe_unused]] unsigned int pf_w[kPfWords];
  [[maybe_unused]] long long pf_off = -1;
  for (int bi = (WIDE != 0 && !wave_items) ? wave : 0; bi < nloop; bi += (WIDE != 0 && !wave_items) ? 4 : 1) {
  const long long lb = DEVLOOP ? wq : (grp * p.bpw + bi) * p.stride + cslot;
  if (lb >= p.nblocks) break;
  gc_block blk;
  if constexpr (DEVLOOP) {
    // every member closes the loop itself (all-gather of the partial sums, see below), so from the second epoch on the
    // descriptor is the one it computed; the first one comes from the host
    if (dl_loc.host_loop) {
      // host-fed: member 0 polls the ten descriptor messages in host memory (one PCIe read per lane and round) and relays
      // them to the team; the others poll the relay.  Bounded: a lost host must not hang the device.
      const DevLoopArgs* dl = &dl_loc;
      const msg_t* dm = (split == 0 ? dl->host_desc : dl->desc_msg) + lb * kDescWords;
      msg_t m = {0u, 0u, 0u, 0u};
      unsigned int spins = 0;
      while (true) {
        if (lane < kDescWords) m = msg_load(dm + lane);
        const bool ok = lane >= kDescWords || m.z == (unsigned int)bi + 1u || m.z == 0xffffffffu;  // 0xffffffff: stop, any epoch
        if (__all(ok)) break;
        if (++spins > (1u << 22)) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (spins > (1u << 22)) {
        if (lane == 0) dl->chan[lb].status = 3;
        if (split == 0 && lane < kDescWords) msg_store(dl->desc_msg + lb * kDescWords + lane, msg_t{3u, 0u, (unsigned int)bi + 1u, 0u});
        break;
      }
      if (split == 0 && lane < kDescWords) msg_store(dl->desc_msg + lb * kDescWords + lane, m);
      union {
        gc_block b;
        unsigned long long q[sizeof(gc_block) / 8];
      } u;
#pragma unroll
      for (int i = 0; i < (int)(sizeof(gc_block) / 8); ++i)
        u.q[i] = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)m.y, i) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)m.x, i);
      if (__builtin_amdgcn_readlane((int)m.x, kDescWords - 1) != 0) break;  // status word: channel finished / record exhausted
      blk = u.b;
    } else if (bi > 0) {
      blk = dl_next;
    } else {
      if (dl_st.status != 0) break;  // record exhausted before the first block (tracking.m:241-245)
      blk = dl_st.blk;
    }
  } else {
    blk = CL ? load_block(p, lb) : p.blocks[lb];
  }
  unsigned long long dl_t0 = 0;
  if constexpr (DEVLOOP) dl_t0 = __builtin_amdgcn_s_memrealtime();
  const DevChannel* __restrict__ chn = p.chans + blk.channel;
  const int arms_here = chn->arms;

  // ---- per-block uniform quantities (see corr_kernel.hip for the reference line citations) --------
  const double R = chn->index_scale;
  const double M = chn->mult[0];
  const double rem = blk.rem_code_phase;
  const double step = blk.code_phase_step;
  const double d = blk.el_spacing;
  const int N = blk.blksize;
  const long long s0 = blk.first_sample;
  const double aE = (rem - d) * R;
  const double aL = (rem + d) * R;
  const double aP = rem * R;
  const double sp = step * R;
  const double tau = blk.carr_freq * p.inv_fs;
  // 1 / (step*R*M) by v_rcp_f64 + one Newton step (relative error ~1e-15: far inside the float rounding of uk)
  const double spM = sp * M;
  double rspM = __builtin_amdgcn_rcp(spM);
  rspM = fma(rspM, fma(-spM, rspM, 1.0), rspM);
  rspM = fma(rspM, fma(-spM, rspM, 1.0), rspM);
  const float uk = (float)(rspM * 2.3283064365386963e-10);  // g_hi (2^-32 units) -> u
  const bool tie_free = __builtin_amdgcn_readfirstlane((int)(blk.reserved & 1)) != 0;  // scalar

  // lanes 0..SPL-1: delta^j = exp(-i*2*pi*j*tau); lane SPL: the chunk stride SPL*kFW samples
  float myC, myS;
  unsigned int myJlo, myJhi;
  int myJint;
  {
    const int j = (lane < SPL) ? lane : SPL * kFW;
    const double x = (double)j * tau;
    // range reduction in double, sincos in float
    sincospif(2.0f * (float)(x - floor(x)), &myS, &myC);
    const double y = (double)j * (sp * M);
    const double yi = floor(y);
    const unsigned long long jf = frac_to_u64(y - yi);
    myJint = (int)yi;
    myJlo = (unsigned int)jf;
    myJhi = (unsigned int)(jf >> 32);
  }
  float C[SPL], S[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    C[j] = rl_f(myC, j);
    S[j] = rl_f(myS, j);
  }
  const float rotC = rl_f(myC, SPL), rotS = rl_f(myS, SPL);
  const unsigned long long Df = ((unsigned long long)rl_u(myJhi, SPL) << 32) | rl_u(myJlo, SPL);
  const int Di = __builtin_amdgcn_readlane(myJint, SPL);

  const long long q0 = s0 >> kShift;
  const long long q1 = (s0 + N - 1) >> kShift;
  const int nchunks = (int)(q1 - q0 + 1);
  const int cps = (nchunks + p.splits - 1) / p.splits;
  const int cbeg = split * cps;
  const int cend = min(nchunks, cbeg + cps);

  float accr[ARMS][3], acci[ARMS][3];
#pragma unroll
  for (int a = 0; a < ARMS; ++a)
#pragma unroll
    for (int x = 0; x < 3; ++x) accr[a][x] = acci[a][x] = 0.0f;

  // Uniform trip count: every lane walks `iters` chunks c = c0 + 64*k.  Only the last iteration can have idle lanes
  // (their words are zeroed) and only the first / last chunk of a lane can be partial, so the loop counter, the load
  // base and every edge test are scalar; the lanes carry nothing but the two ramps and the accumulators.
  const int iters = (cend - cbeg + kFW - 1) / kFW;
  const int c0 = cbeg + lane;
  float wc = 1.0f, ws = 0.0f;
  if (iters > 0) {
    constexpr int CB = SPL * Fmt<MODE>::bps;  // bytes per lane-chunk
    const int i00 = (int)((q0 + c0) * SPL - s0);
    constexpr bool SHARE = SHARE_EL;
    constexpr int NS = SHARE ? 2 : 3;
    Fx fx[NS];
    const double isp = __dmul_rn((double)i00, sp);
    fx[0] = to_fx(__dmul_rn(__dadd_rn(aE, isp), M));
    fx[1] = to_fx(__dmul_rn(__dadd_rn(aP, isp), M));
    if constexpr (!SHARE) fx[2] = to_fx(__dmul_rn(__dadd_rn(aL, isp), M));
    const uint8_t* __restrict__ base = p.if_base;

    // With earlyLateSpc*R*M = 1/2 (the reference's default 0.5-chip spacing) the early and late ramps differ by
    // exactly one table entry: same fraction, same transition position in every chunk, so they share one ramp and
    // one pair of partial sums.  Decided once per block.
    const bool share_broken = SHARE_EL && __builtin_amdgcn_readfirstlane(2.0 * d * R * M == 1.0 ? 0 : 1) != 0;  // host-checked (gc_block_shares_el); else every chunk goes exact
    // ramp state per set (0 = early, and late when shared; 1 = prompt; 2 = late): t = kk - (ghi:glo) / 2^64
    unsigned int glo[NS], ghi[NS];
    int kk[NS];
#pragma unroll
    for (int sx = 0; sx < NS; ++sx) {
      glo[sx] = (unsigned int)fx[sx].G;
      ghi[sx] = (unsigned int)(fx[sx].G >> 32);
      kk[sx] = fx[sx].k0;
    }
    const unsigned int Dlo = (unsigned int)Df, Dhi = (unsigned int)(Df >> 32);
    const uint8_t* __restrict__ bs = base + (long long)CB * (q0 + cbeg);  // uniform
    const unsigned int voff = (unsigned int)lane * CB;
    const unsigned int voff_last = min(voff, (unsigned int)(cend - 1 - cbeg - (iters - 1) * kFW) * CB);

    auto load_k = [&](const int k, unsigned int (&w)[NW]) {
      const uint8_t* __restrict__ pk = bs + (size_t)k * (size_t)(kFW * CB);  // scalar
      const unsigned int off = (k == iters - 1) ? voff_last : voff;         // idle lanes of the last iteration re-read the last chunk
      load_words<MODE, SPL>(pk + off, 0, w);
    };

    auto process = [&](unsigned int (&w)[NW], const int k) {
      const bool last = (k == iters - 1);
      if ((k == 0) | last) {
        int kq = k;
        asm volatile("" : "+v"(kq));  // opaque: nothing of this rare branch is to be precomputed outside the loop
        const int i0 = i00 + kq * (SPL * kFW);
        if (last && c0 + kq * kFW >= cend) {  // idle lane: no samples, and a table index that exists (0 * garbage could be NaN)
#pragma unroll
          for (int q = 0; q < NW; ++q) w[q] = 0u;
#pragma unroll
          for (int sx = 0; sx < NS; ++sx) kk[sx] = 0;
        }
        if ((i0 < 0) | (i0 + SPL > N)) mask_words<MODE, SPL>(w, i0, N);
      }

      // transition positions and the near-tie filter
      float gh[NS];
#pragma unroll
      for (int sx = 0; sx < NS; ++sx) asm("v_cvt_f32_u32_e32 %0, %1" : "=v"(gh[sx]) : "v"(ghi[sx]));
      bool exact = share_broken;
      if (!tie_free) {  // blocks the host proved tie-free (gc_mark_tie_free) skip the filter
        bool suspect = false;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) {
          const float u = gh[sx] * uk;
          suspect |= fabsf(u - rintf(u)) < kTieTol;
        }
        exact |= __any(suspect) != 0;
      }

      float Ur[ARMS][3], Ui[ARMS][3];
      if (exact) {
        // ---- exact path (~1e-4 of wave-chunks): float64 index per sample, as the reference ------
#pragma unroll
        for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
          for (int x = 0; x < 3; ++x) Ur[ar][x] = Ui[ar][x] = 0.0f;
        // A rolled loop that re-reads the chunk's samples from memory (L1-resident) and carries delta^j by
        // recurrence: the rare path must not set the kernel's register budget (unrolled it cost 200 VGPRs).
        {
          int kq = k, Nq = N;
          asm volatile("" : "+v"(kq), "+v"(Nq));  // opaque: the rare path's constants are formed here, not carried in registers
          const int c = c0 + kq * kFW;
          const bool act = c < cend;
          const int i0 = i00 + kq * (SPL * kFW);
          const double bP = __dmul_rn(__dadd_rn(__dmul_rn((double)(Nq - 1), step), rem), R);
          const double bE = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(Nq - 1), step), rem), -d), R);
          const double bL = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)(Nq - 1), step), rem), d), R);
          const uint8_t* sp8 = base + (long long)CB * (q0 + min(c, cend - 1));
          float cr = 1.0f, ci = 0.0f;  // delta^j = cr - i*ci
#pragma unroll 1
          for (int j = 0; j < SPL; ++j) {
            const int i = i0 + j;
            float x0, x1 = 0.0f;
            if constexpr (Fmt<MODE>::bps == 2 && !kReal) {
              x0 = (float)(signed char)sp8[2 * j];
              x1 = (float)(signed char)sp8[2 * j + 1];
            } else if constexpr (Fmt<MODE>::bps == 4) {
              x0 = (float)reinterpret_cast<const short*>(sp8)[2 * j];
              x1 = (float)reinterpret_cast<const short*>(sp8)[2 * j + 1];
            } else if constexpr (Fmt<MODE>::bps == 1) {
              x0 = (float)(signed char)sp8[j];
            } else {
              x0 = (float)reinterpret_cast<const short*>(sp8)[j];
            }
            float a = Fmt<MODE>::swap ? x1 : x0, b = Fmt<MODE>::swap ? x0 : x1;
            if ((unsigned int)i >= (unsigned int)Nq || !act) a = b = 0.0f;  // edge chunk / idle lane
            const float yr = a * cr + b * ci;
            const float yi = b * cr - a * ci;
            const float ncr = cr * C[1] - ci * S[1], nci = cr * S[1] + ci * C[1];
            cr = ncr;
            ci = nci;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
              const double ax = (x == 0) ? aE : (x == 1) ? aP : aL;
              const double bx = (x == 0) ? bE : (x == 1) ? bP : bL;
              double t;
              if (2 * i < Nq - 1)
                t = __dadd_rn(ax, __dmul_rn((double)i, sp));
              else if (2 * i > Nq - 1)
                t = __dadd_rn(bx, -__dmul_rn((double)(Nq - 1 - i), sp));
              else
                t = __dadd_rn(ax, bx) / 2.0;
              int kx = (int)ceil(__dmul_rn(t, M));
              kx = max(-1, min(kx, 0x3fffffff));
#pragma unroll
              for (int ar = 0; ar < ARMS; ++ar) {
                const int kc = min(kx, chn->stage_len[(ar < arms_here) ? ar : 0] + 1);
                const float cf = table_entry(ar, kc).x;
                Ur[ar][x] = fmaf(cf, yr, Ur[ar][x]);
                Ui[ar][x] = fmaf(cf, yi, Ui[ar][x]);
              }
            }
          }
        }
      } else {
        // ---- fast path: the running sums P[j] = y_0 + .. + y_j go to LDS as they are formed ([j][lane]: conflict-
        //      free 8-byte stores), two fused multiply-adds per component and sample; the sum after a tap's
        //      transition is then T - P[floor(u)]: two lookups per chunk, no per-sample work per tap ---------------
        float Tr = 0.f, Ti = 0.f;
        static_for<0, SPL>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          // keep the scheduler from hoisting all the conversions to the top of the chunk (register pressure)
          if constexpr (j % GC_SCHED_GROUP == 0 && j != 0) __builtin_amdgcn_sched_barrier(0);
          float a, b;
          sample_ab<MODE, j, NW>(w, a, b);
          if constexpr (kReal) {
            Tr = fmaf(a, C[j], Tr);
            Ti = fmaf(-a, S[j], Ti);
          } else {
            Tr = fmaf(a, C[j], fmaf(b, S[j], Tr));
            Ti = fmaf(b, C[j], fmaf(-a, S[j], Ti));
          }
          store_prefix<(2 * j) * kFW * 4, (2 * j + 1) * kFW * 4>(Tr, Ti, pfx_m0);
        });
        float Sr[NS], Si[NS], Pr[NS], Pi[NS];  // sums after / before the transition
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) {
          const int m = min((int)(gh[sx] * uk), SPL - 1);  // samples 0 .. m lie before the transition (P[SPL-1] = T: none after)
          Pr[sx] = pfx[(2 * m) * kFW + lane];
          Pi[sx] = pfx[(2 * m + 1) * kFW + lane];
          Sr[sx] = Tr - Pr[sx];
          Si[sx] = Ti - Pi[sx];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) {
#pragma unroll
          for (int ar = 0; ar < ARMS; ++ar) {
            const int sx = (SHARE && x == 2) ? 0 : x;
            const float2 cd = table_entry(ar, kk[sx] + ((SHARE && x == 2) ? 1 : 0));
            // {c, dc}: c*T + dc*S;  {c1, c2}: c1*P + c2*S
            Ur[ar][x] = fmaf(cd.x, TABF ? Pr[sx] : Tr, cd.y * Sr[sx]);
            Ui[ar][x] = fmaf(cd.x, TABF ? Pi[sx] : Ti, cd.y * Si[sx]);
          }
        }
      }
      // Horner step: acc = acc * conj(rho) + U, rho = delta^(SPL*kFW) = rotC - i rotS
#pragma unroll
      for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const float nr = fmaf(accr[ar][x], rotC, fmaf(-acci[ar][x], rotS, Ur[ar][x]));
          const float ni = fmaf(accr[ar][x], rotS, fmaf(acci[ar][x], rotC, Ui[ar][x]));
          accr[ar][x] = nr;
          acci[ar][x] = ni;
        }
      // next chunk: t += 64*SPL*step*R*M, exactly: the 64-bit fraction's borrow carries into the integer part
#pragma unroll
      for (int sx = 0; sx < NS; ++sx)
        asm("v_sub_co_u32_e32 %0, vcc, %0, %3\n\tv_subb_co_u32_e32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32_e32 %2, vcc, %2, %5, vcc"
            : "+v"(glo[sx]), "+v"(ghi[sx]), "+v"(kk[sx])
            : "v"(Dlo), "v"(Dhi), "v"(Di)
            : "vcc");
    };

    // two word buffers, alternating: the next chunk's loads are in flight while this one is processed, no copies
    unsigned int wa[NW], wb[NW];
    bool fetched = false;
    if constexpr (DEVLOOP) {
      // the words fetched during the last closure are this epoch's first chunk if they came from the same place in every lane
      const long long want = (long long)CB * (q0 + cbeg) + (long long)((0 == iters - 1) ? voff_last : voff);
      fetched = __all(pf_off == want) != 0;
      if (fetched) {
#pragma unroll
        for (int q = 0; q < NW; ++q) wa[q] = pf_w[q];
      }
      pf_off = -1;
    }
    if (!fetched) load_k(0, wa);
    for (int k = 0;; k += 2) {
      if (k + 1 < iters) load_k(k + 1, wb);
      process(wa, k);
      if (k + 1 >= iters) break;
      if (k + 2 < iters) load_k(k + 2, wa);
      process(wb, k + 1);
      if (k + 2 >= iters) break;
    }
    // exact carrier phase at the first sample of this thread's LAST chunk
    const double ph = blk.rem_carr_phase * 0.15915494309189535 + (double)(i00 + (iters - 1) * (SPL * kFW)) * tau;
    sincospif(2.0f * (float)(ph - floor(ph)), &ws, &wc);
  }

  // ---- rotate into the absolute frame and reduce across the wavefront (DPP) ------------------------
  // The lanes' sums (<= 288 samples each at 18 Msps) are float32; what they add up to is the block's float64 total in the
  // reference (tracking.m:291-300).  A float32 tree over the 64 lanes rounds the prompt sum (tens of thousands, all lanes in phase)
  // at 3e-3 per level - the largest error of the whole kernel - so the closed-loop instantiations, whose sums steer the next
  // block's geometry, add the lanes in float64 (half_sum_f64: ~120 instructions per block, nothing against their latency);
  // the batched replay keeps the float32 tree unless GC_FAST_F64_TOTALS is compiled in (measured cost: DESIGN.md 4.1).
#ifndef GC_FAST_F64_TOTALS
#define GC_FAST_F64_TOTALS 0
#endif
  constexpr bool kF64Tot = CL || DEVLOOP || GC_FAST_F64_TOTALS != 0;
  double* o = (p.splits == 1) ? p.out + lb * GC_OUT_STRIDE : p.partial + (lb * p.splits + split) * GC_OUT_STRIDE;
  float tot[ARMS * 6];
  [[maybe_unused]] double totd[ARMS * 6];
#pragma unroll
  for (int ar = 0; ar < ARMS; ++ar)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const float vr = wc * accr[ar][x] + ws * acci[ar][x], vi = wc * acci[ar][x] - ws * accr[ar][x];
      if constexpr (kF64Tot) {
        const double hr = half_sum_f64((double)vr), hi = half_sum_f64((double)vi);   // lanes 31 / 63: the halves' sums
        totd[ar * 6 + 2 * x] = rl_f64(hr, 31) + rl_f64(hr, 63);
        totd[ar * 6 + 2 * x + 1] = rl_f64(hi, 31) + rl_f64(hi, 63);
        tot[ar * 6 + 2 * x] = (float)totd[ar * 6 + 2 * x];
        tot[ar * 6 + 2 * x + 1] = (float)totd[ar * 6 + 2 * x + 1];
      } else {
        tot[ar * 6 + 2 * x] = wave_sum_lane63(vr);
        tot[ar * 6 + 2 * x + 1] = wave_sum_lane63(vi);
      }
    }
  if constexpr (DEVLOOP) {
    // All-gather: every member posts its six partial sums as two messages {f, f, f, tag} into this epoch's half of the
    // channel's message array (two halves alternate: a member can be at most one epoch ahead of the slowest reader), polls
    // the whole team's messages — lane 32h + k takes message h of member k — adds them in double in a fixed order (DPP),
    // and closes the loop itself: identical inputs, identical instructions, identical next descriptor in every member, so
    // there is ONE message hop per epoch and no descriptor broadcast.  Member 0 alone writes records and host-visible state.
#if defined(GC_DEVLOOP_ARGS_BY_POINTER) && GC_DEVLOOP_ARGS_BY_POINTER
    const DevLoopArgs* dl = p.devloop;  // A/B (scripts/variants.sh): the closure reads the constants through the pointer, as before round 6
#else
    const DevLoopArgs* dl = &dl_loc;
#endif
    const unsigned int tag = (unsigned int)bi + 1u;
    msg_t* pm = dl->part_msg + ((lb * 2 + (bi & 1)) * dl->splits) * 2;
    if (lane == 63) {
      msg_store(pm + split * 2, msg_t{__float_as_uint(tot[0]), __float_as_uint(tot[1]), __float_as_uint(tot[2]), tag}, dl->reserved);
      msg_store(pm + split * 2 + 1, msg_t{__float_as_uint(tot[3]), __float_as_uint(tot[4]), __float_as_uint(tot[5]), tag}, dl->reserved);
    }
    const unsigned long long dl_t1 = __builtin_amdgcn_s_memrealtime();
    // what does not need the sums, while the messages are on their way
    const DevLoopPre dl_pre = devloop_pre(dl, dl_st, blk, 1.0);
    const int mk = lane & 31, mh = lane >> 5;
    const bool mine = mk < dl->splits;  // splits <= 32
    msg_t m = {0u, 0u, 0u, 0u};
    unsigned int spins = 0;
    while (true) {
      if (mine) m = msg_load(pm + 2 * mk + mh, dl->reserved);
      const bool ok = !mine || m.w == tag;
      if (__all(ok)) break;
      if (++spins > (1u << 22)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (spins > (1u << 22)) {  // a lost team member must not hang the device: everybody gives up after the same bound
      if (lane == 0) dl->chan[lb].status = 3;
      break;
    }
    const unsigned long long dl_t2 = __builtin_amdgcn_s_memrealtime();
    if (!dl->host_loop && dl->prefetch) {
      // next epoch's first chunk (see pf_w above): the block of blksize N' (N' = N or a sample either side) from s0 + N on, cut into
      // the members' shares as the loop's head does; a member whose share moves with N' finds the mismatch there and reloads
      constexpr int CBn = SPL * Fmt<MODE>::bps;
      const long long s0n = blk.first_sample + blk.blksize;
      const long long q0n = s0n >> kShift, q1n = (s0n + blk.blksize - 1) >> kShift;
      const int nchn = (int)(q1n - q0n + 1);
      const int cpsn = (nchn + p.splits - 1) / p.splits;
      const int cbegn = split * cpsn, cendn = min(nchn, cbegn + cpsn);
      const int itersn = (cendn - cbegn + kFW - 1) / kFW;
      if (itersn > 0 && (unsigned long long)(s0n + blk.blksize + 2 * SPL) <= dl->if_nsamples) {
        const unsigned int voffn = (unsigned int)lane * CBn;
        const unsigned int vlastn = min(voffn, (unsigned int)(cendn - 1 - cbegn - (itersn - 1) * kFW) * CBn);
        pf_off = (long long)CBn * (q0n + cbegn) + (long long)((0 == itersn - 1) ? vlastn : voffn);
        load_words<MODE, SPL>(p.if_base + pf_off, 0, pf_w);
      }
    }
    double part[3] = {mine ? (double)__uint_as_float(m.x) : 0.0, mine ? (double)__uint_as_float(m.y) : 0.0,
                      mine ? (double)__uint_as_float(m.z) : 0.0};
#pragma unroll
    for (int v = 0; v < 3; ++v) part[v] = half_sum_f64(part[v]);  // lanes 31 / 63: the sums of the two halves
    double sums[6];
#pragma unroll
    for (int v = 0; v < 6; ++v) sums[v] = rl_f64(part[v % 3], v < 3 ? 31 : 63);
    if (dl->host_loop) {
      // host-fed: member 0 hands the team's six sums to the host as tagged records (one 16-byte store per lane) and the
      // loop goes back to waiting for the host's next descriptor
      if (split == 0) {
        TaggedSlot* ts = reinterpret_cast<TaggedSlot*>(dl->host_tagged) + lb * GC_OUT_STRIDE;
        double mine = 0.0;
#pragma unroll
        for (int v = 0; v < 6; ++v) mine = (lane == v) ? sums[v] : mine;
        if (lane < 6) {
          TaggedSlot rec;
          rec.value = mine;
          rec.tag = tag;
          rec.zero = 0u;
          *reinterpret_cast<uint4*>(ts + lane) = *reinterpret_cast<const uint4*>(&rec);
        }
        __threadfence_system();  // out of the L2 now: this wave goes on polling, not to a kernel end
      }
      continue;
    }
    dl_next = blk;
    double dl_rv[GC_TRK_NFIELDS];
    // member 0 writes the records, member 1 (the only one when the team has one member: then member 0) keeps the C/N0 sums
    const bool cno_member = split == (p.splits > 1 ? 1 : 0);
    const int st = devloop_post<1>(dl, dl_st, dl_next, bi, sums, 1, 1.0, dl_pre, [&](int f, double v) { dl_rv[f] = v; }, cno_member,
                                   cno_member && lane == 0 ? (long long)lb : -1LL);
    {
      // teams of 17 members or more: member f + 2 stores field f of the record every member holds (member 0 keeps the state, member 1 the C/N0 sums)
      const bool spread = !dl->one_writer && p.splits >= GC_TRK_PILOT_I_E + 2;
      if (spread || split == 0) devloop_commit(dl, dl->chan + lb, dl_st, lb, bi, dl_rv, 1, lane, split, spread);
    }
    if (split == 0) {
      if (dl->timing == 1 && lane == 0) {  // phase clocks (100 MHz): correlate | wait for partials | close
        const unsigned long long dl_t3 = __builtin_amdgcn_s_memrealtime();
        DevLoopChan* cc = dl->chan + lb;
        cc->pad[0] += (double)(dl_t1 - dl_t0);
        cc->pad[1] += (double)(dl_t2 - dl_t1);
        cc->pad[2] += (double)(dl_t3 - dl_t2);
      }
    }
    if (st != 0) break;
  } else if (CL) {
    // lane v takes total v (broadcast from lane 63) and stores its 16-byte tagged record
    TaggedSlot* ts = p.tagged + (lb * p.splits + split) * GC_OUT_STRIDE;
    double mine = 0.0;  // (kF64Tot: every lane holds the float64 totals)
#pragma unroll
    for (int v = 0; v < ARMS * 6; ++v) mine = (lane == v) ? totd[v] : mine;
    if (lane < ARMS * 6) {
      TaggedSlot rec;
      rec.value = (lane < arms_here * 6) ? mine : 0.0;
      rec.tag = p.notify_tag;
      rec.zero = 0u;
      *reinterpret_cast<uint4*>(ts + lane) = *reinterpret_cast<const uint4*>(&rec);
    }
  } else if (lane == 63) {
#pragma unroll
    for (int v = 0; v < ARMS * 6; ++v) o[v] = (v < arms_here * 6) ? (kF64Tot ? totd[v] : (double)tot[v]) : 0.0;
    for (int v = ARMS * 6; v < GC_OUT_STRIDE; ++v) o[v] = 0.0;
  }
  }  // bpw loop

}
template <int ARMS, int MODE, int SPL>
void launch_variant(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem) {
  const bool cl = a.tagged != nullptr;
  // the shared-early/late instantiation exists for single-arm channels (GPS L1 C/A, B1I, GLONASS)
  const bool share = (ARMS == 1) && a.share_el != 0;
  if (a.wide == 2) {
    // four-wave workgroups + plain float tables (small single-arm tables, big replay lists)
    if constexpr (ARMS == 1 && (MODE == I8_IQ || MODE == I8_QI)) {
      if (share) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, true, 2>), grid, dim3(256), smem, ctx->stream, a, ib);
      else hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, false, 2>), grid, dim3(256), smem, ctx->stream, a, ib);
    }
    return;
  }
  if (a.wide) {
    // four-wave workgroups + int8-pair tables: instantiated for the 8-sample chunk, int8 I/Q records,
    // one or two arms (Galileo E1 B / B+C and similar 8000-20000-entry tables)
    if constexpr (ARMS <= 2 && (MODE == I8_IQ || MODE == I8_QI)) {
      if constexpr (SPL == 8) {
        if (cl) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, true, false, 1>), grid, dim3(256), smem, ctx->stream, a, ib);
        else if (share) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, ARMS == 1, 1>), grid, dim3(256), smem, ctx->stream, a, ib);
        else hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, false, 1>), grid, dim3(256), smem, ctx->stream, a, ib);
      } else {
        // 16-sample chunks: replay only (chosen by the launcher for the prefix-sum variant)
        if (share) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, ARMS == 1, 1>), grid, dim3(256), smem, ctx->stream, a, ib);
        else hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, false, 1>), grid, dim3(256), smem, ctx->stream, a, ib);
      }
    }
    return;
  }
  if constexpr (ARMS == 1) {
    if (cl && share) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, true, true, 0>), grid, dim3(kFW), smem, ctx->stream, a, ib);
    else if (cl) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, true, false, 0>), grid, dim3(kFW), smem, ctx->stream, a, ib);
    else if (share) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, true, 0>), grid, dim3(kFW), smem, ctx->stream, a, ib);
    else hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, false, 0>), grid, dim3(kFW), smem, ctx->stream, a, ib);
  } else {
    (void)share;
    if (cl) hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, true, false, 0>), grid, dim3(kFW), smem, ctx->stream, a, ib);
    else hipLaunchKernelGGL((corr_epl_fast_kernel<ARMS, MODE, SPL, false, false, 0>), grid, dim3(kFW), smem, ctx->stream, a, ib);
  }
}

template <int ARMS, int SPL>
int launch_fast_mode(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem) {
  int mode;
  if (ctx->if_dtype == GC_I8)
    mode = ctx->if_layout == GC_IQ ? I8_IQ : ctx->if_layout == GC_QI ? I8_QI : I8_REAL;
  else
    mode = ctx->if_layout == GC_IQ ? I16_IQ : ctx->if_layout == GC_QI ? I16_QI : I16_REAL;
  switch (mode) {
    case I8_IQ: launch_variant<ARMS, I8_IQ, SPL>(ctx, a, ib, grid, smem); break;
    case I8_QI: launch_variant<ARMS, I8_QI, SPL>(ctx, a, ib, grid, smem); break;
    case I16_IQ: launch_variant<ARMS, I16_IQ, 8>(ctx, a, ib, grid, smem); break;
    case I16_QI: launch_variant<ARMS, I16_QI, 8>(ctx, a, ib, grid, smem); break;
    case I8_REAL: launch_variant<ARMS, I8_REAL, 8>(ctx, a, ib, grid, smem); break;
    default: launch_variant<ARMS, I16_REAL, 8>(ctx, a, ib, grid, smem); break;
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

template <int MODE, int SPL>
int launch_devloop_one(gc_context* ctx, KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem, bool share) {
  void* args[2] = {(void*)&a, (void*)&ib};
  const void* fn = share ? (const void*)corr_epl_fast_kernel<1, MODE, SPL, false, true, 0, true>
                         : (const void*)corr_epl_fast_kernel<1, MODE, SPL, false, false, 0, true>;
  // cooperative: every team member must be resident while the others spin on the epoch flag
  GC_PERSIST(gc_launch_persistent(ctx, fn, grid, dim3(kFW), args, (unsigned int)smem));
  return GC_OK;
}

// 16-sample chunks exist for int8 I/Q and Q/I records only; every other record format takes 8-sample chunks
int launch_devloop_mode(gc_context* ctx, KArgs& a, const InlineBlocks& ib, dim3 grid, size_t smem, bool share, bool spl16) {
  const int lay = ctx->if_layout;
  if (ctx->if_dtype == GC_I8 && lay != GC_REAL) {
    if (spl16) return lay == GC_QI ? launch_devloop_one<I8_QI, 16>(ctx, a, ib, grid, smem, share) : launch_devloop_one<I8_IQ, 16>(ctx, a, ib, grid, smem, share);
    return lay == GC_QI ? launch_devloop_one<I8_QI, 8>(ctx, a, ib, grid, smem, share) : launch_devloop_one<I8_IQ, 8>(ctx, a, ib, grid, smem, share);
  }
  if (ctx->if_dtype == GC_I8) return launch_devloop_one<I8_REAL, 8>(ctx, a, ib, grid, smem, share);
  if (lay == GC_REAL) return launch_devloop_one<I16_REAL, 8>(ctx, a, ib, grid, smem, share);
  return lay == GC_QI ? launch_devloop_one<I16_QI, 8>(ctx, a, ib, grid, smem, share) : launch_devloop_one<I16_IQ, 8>(ctx, a, ib, grid, smem, share);
}

}  // namespace

// Persistent single-arm tracker (any record format) with device-side loop closure: grid = channels x splits one-wave workgroups.
int gc_launch_devloop(gc_context* ctx, const KArgs& a_in, unsigned int grid, bool spl16, bool share_el) {
  KArgs a = a_in;
  InlineBlocks ib;
  std::memset(&ib, 0, sizeof ib);
  a.red_off = 8 * ctx->max_lds_bytes;
  const size_t smem = (size_t)a.red_off + 8 * 3 * 64 + (size_t)16 * kFW * sizeof(float2);  // + the closer's reduction scratch + prefix sums
  return launch_devloop_mode(ctx, a, ib, dim3(grid), smem, share_el, spl16 && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL);
}

bool gc_fast_prefers_wide() { return true; }

// spl16: every block satisfies 15*step*R*M < 1 and the samples are int8 I/Q
int gc_launch_correlator_fast(gc_context* ctx, const KArgs& a, const InlineBlocks& ib, unsigned int grid, int max_arms,
                              bool spl16) {
  // float2 tables: 8 bytes per staged entry (lds_off counts entries here)
  const bool wide = spl16 && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL;  // 16-sample chunks
  // + per wave the running prefix sums of one lane-chunk ([SPL][64] float2) of the prefix-sum variant
  size_t smem = (size_t)a.red_off + 64 + (size_t)(a.wide ? 4 : 1) * (wide ? 16 : 8) * kFW * sizeof(float2);
  if (const char* e = GC_TUNE_ENV("GC_FAST_EXTRA_LDS")) smem += (size_t)std::atoi(e);  // tuning: occupancy experiments
  if (wide) {
    switch (max_arms) {
      case 1: return launch_fast_mode<1, 16>(ctx, a, ib, dim3(grid), smem);
      case 2: return launch_fast_mode<2, 16>(ctx, a, ib, dim3(grid), smem);
      default: return launch_fast_mode<3, 16>(ctx, a, ib, dim3(grid), smem);
    }
  }
  switch (max_arms) {
    case 1: return launch_fast_mode<1, 8>(ctx, a, ib, dim3(grid), smem);
    case 2: return launch_fast_mode<2, 8>(ctx, a, ib, dim3(grid), smem);
    default: return launch_fast_mode<3, 8>(ctx, a, ib, dim3(grid), smem);
  }
}
