// devloop.h — device-side loop closure: the tracking loop of tracking.m:184-348 without the
// per-epoch host round trip.  ONE persistent launch runs all epochs; the workgroups that share a channel (its
// "team": one per split) exchange two kinds of self-validating 16-byte messages per epoch — {payload, epoch tag}
// written by ONE store instruction and read by ONE load instruction, both at system scope (sc0 sc1: no cache on the
// way), so no flag, counter, atomic or fence is needed:
//   fast kernel (corr_fast.hip): all-gather - every member stores its six partial sums as two messages {f, f, f, tag}, polls the
//   whole team's messages (one per lane), adds them in double in a fixed order and closes the loop ITSELF (discriminators,
//   loop filters, next block geometry - the same float64 statements as gc_track's host loop, tracking.m:302-335); identical
//   inputs give identical descriptors, so nothing is broadcast: ONE one-way propagation delay per epoch.  Member 0 writes
//   the epoch's records.
//   lane kernel (corr_lane.hip): gather and broadcast - member 0 (the closer) polls the other members' messages, closes the
//   loop, writes the records and publishes the next descriptor as ten messages {word, tag} that the other members poll.
// Channels never wait for each other.  All team workgroups must be co-resident: the launch is cooperative.  Polls are bounded.
#pragma once
#include "gc_internal.h"

namespace gcorr {

constexpr int kDescWords = 10;  // gc_block (9 x 8 bytes) + status word

typedef unsigned int msg_t __attribute__((ext_vector_type(4)));  // one 16-byte message

struct DevLoopChan {
  gc_block blk;              // initial descriptor (host) / last descriptor (closer)
  int status;                // 0 running, 1 all epochs done, 2 record exhausted (tracking.m:241-245), 3 wait timed out,
                             // 4 code NCO diverged: codeFreq not finite / not positive (MATLAB's fread(fid, NaN) errors there)
  int epochs_done;
  int pad0[2];
  // loop state, touched by the closing member only
  long long pos;
  double code_freq, code_freq_basis, rem_code;
  double carr_freq, carr_basis, rem_carr;
  double old_code_nco, old_code_err, old_carr_nco, old_carr_err;  // 2nd-order PLL / DLL
  double d2_carr_err, d_carr_err;                                 // 3-state PLL
  double pad[3];
  int table_phase;  // GPS L2C CLCodePhase (1-based segment of the CL code; 0: none), GPS_L2C/include/tracking.m:261,357-360
  int cno_n;        // C/N0 intervals completed; cno_value is the latest one when cno_ready
  double cno_z0, cno_s1, cno_s2, cno_value;  // running sums of Z - Z0 and (Z - Z0)^2 over the current interval (Z = I_P^2 + Q_P^2)
  int cno_ready, pad_i;
};

struct DevLoopArgs {
  DevLoopChan* chan;   // [nch]
  msg_t* desc_msg;     // [nch][kDescWords]   {word lo, word hi, tag, 0}, tag = epoch + 1
  msg_t* part_msg;     // [nch][splits][2]    {f, f, f, tag}  (fast kernel) | [nch][members][12] {double, tag, 0} (lane kernel)
  double* records;     // [nch][GC_TRK_NFIELDS][n_epochs]
  gc_track_params prm;
  double tau1code, tau2code, tau1carr, tau2carr;
  // the loop filters' quotients (tau2/tau1, int_time/tau1: tracking.m:308-309,327-328), formed once on the host - four
  // dependent float64 divisions per epoch otherwise
  double k1code, k2code, k1carr, k2carr;
  unsigned long long if_nsamples;
  int n_epochs;
  int splits;
  int code_index_scale_is_one;  // R == 1 (the only case wired up)
  int reserved;                 // message scope (msg_load / msg_store): 0 = system
  int timing;                   // GC_DEVLOOP_TIMING: the closer accumulates its phase clocks in DevLoopChan::pad (costs ~1 us per epoch)
  int prefetch;                 // fast kernel: fetch the next epoch's first chunk during the closure (corr_fast.hip; GC_DEVLOOP_NO_PREFETCH=1 in the tuning build: 0)
  int pad_prefetch;
  // Host-fed variant (gc_track's persistent mode): the HOST closes the loop (tracking.m:302-335 stay where the reference has
  // them) but nothing is launched per epoch: member 0 of a team polls the channel's descriptor messages in host-mapped
  // memory (tag = epoch + 1), relays them to its team through desc_msg, and every member writes its six partial sums as
  // tagged 16-byte records straight into host-mapped memory, where the host polls them.
  // C/N0 by the variance-summing method inside the loop (gc_track_params::cno_interval): [nch][cno_nk], device memory
  double* cno;
  int cno_nk, one_writer;  // GC_DEVLOOP_ONE_WRITER (tuning): member 0 stores the whole record
  int host_loop;
  const msg_t* host_desc;       // [nch][kDescWords], host-mapped
  void* host_tagged;            // TaggedSlot [nch][splits][GC_OUT_STRIDE], host-mapped
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// Message scope (g_msg_scope, set per launch from DevLoopArgs::reserved): 0 = system (sc0 sc1: past every cache),
// 1 = workgroup (sc0: past the CU's L1 only — the team's L2 is the meeting point; valid when the whole team runs on ONE
// XCD, which the launch arranges and the kernel verifies through HW_REG_XCC_ID before switching to it), 2 = agent (sc1).
__device__ __forceinline__ msg_t msg_load(const msg_t* p, int scope = 0) {
  msg_t v;
  if (scope == 1)
    asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (scope == 2)
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void msg_store(msg_t* p, msg_t v, int scope = 0) {
  if (scope == 1)
    asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  else if (scope == 2)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// Closing member: lane-uniform float64 restatement of tracking.m:273-348 for one channel and epoch — the same statements
// as gc_track's host loop (track.hip), pilot handling modes 0-3 included — in two parts, so that the closer of the fast
// kernel can run the first one while the other members' partial sums are still on their way:
//   devloop_pre   what depends only on the epoch's own geometry and NCO state: the remainders the next block starts from
//                 (tracking.m:273-283: one fmod, one division);
//   devloop_post  what needs the sums: discriminators, loop filters, records, the next block (tracking.m:302-335, 219-222).
// `st` is the caller's copy of the loop state (the fast kernel keeps it in registers across epochs), `gch` the channel's slot
// in device memory, kept up to date for the host by lane 0.  `sums`: 6 per arm; R: the channel's index scale.  On return b
// holds the next epoch's block geometry; the return value is the channel status (0 = keep going).
struct DevLoopPre {
  double rem_code_new, rem_carr_new;
};

__device__ inline DevLoopPre devloop_pre(const DevLoopArgs* __restrict__ dl, const DevLoopChan& st, const gc_block& b, double R) {
  const gc_track_params& p = dl->prm;
  const double kPi = 3.141592653589793;
  const int n = b.blksize;
  const double step = b.code_phase_step;
  DevLoopPre r;
  const double t_last = ((n - 1) * step + st.rem_code) * R;                    // tcode(blksize), :273 / GAL_E1C :268
  r.rem_code_new = (R != 1.0) ? (t_last / R + step) - p.code_length : (t_last + step) - p.code_length;
  const double time_n = (double)n / p.sampling_freq;                            // :280-283
  const double trig_n = ((st.carr_freq * 2.0 * kPi) * time_n) + st.rem_carr;
  r.rem_carr_new = fmod(trig_n, 2 * kPi);
  return r;
}

template <int MAXARMS, class Rec>
__device__ inline int devloop_post(const DevLoopArgs* __restrict__ dl, DevLoopChan& st, gc_block& b, int e, const double (&sums)[6 * MAXARMS],
                                   int arms, double R, const DevLoopPre& pre, Rec&& rec, bool do_cno = true, long long cno_slot = -1) {
  // rec(field, value) takes the epoch's record: straight to device memory (lane kernel) or into registers, to be stored by
  // devloop_commit AFTER the next descriptor is on its way (fast kernel)
  const gc_track_params& p = dl->prm;
  const double kPi = 3.141592653589793;
  const int n = b.blksize;
  const double i_e = sums[0], q_e = sums[1], i_p = sums[2], q_p = sums[3], i_l = sums[4], q_l = sums[5];
  // do_cno: in an all-gather team every member runs this closure, ONE of them (not the one that writes the records) keeps the
  // C/N0 sums - they feed nothing back, so the other members need not carry them on the epoch's critical path; that member
  // stores the finished value itself (cno_slot >= 0), the single closer of the lane kernel leaves it to devloop_commit
  if (do_cno && p.cno_interval > 0 && dl->cno) {
    // CNoVSM(I_P(loopCnt-K+1:loopCnt), Q_P(...), T) when rem(loopCnt, K) == 0 (tracking.m:351-358; Common/CNoVSM.m:38-47):
    // Z = I^2 + Q^2, Zm = mean(Z), Zv = var(Z) (N-1), Pav = sqrt(Zm^2 - Zv), Nv = (Zm - Pav)/2, 10*log10(|Pav/(2*Nv)/T|).
    // Sums of Z - Z0 (Z0 = the interval's first Z) keep the one-pass variance free of cancellation.
    const int K = p.cno_interval, kk = e % K;
    const double z = i_p * i_p + q_p * q_p;
    if (kk == 0) {
      st.cno_z0 = z;
      st.cno_s1 = 0.0;
      st.cno_s2 = 0.0;
    }
    const double dz = z - st.cno_z0;
    st.cno_s1 += dz;
    st.cno_s2 += dz * dz;
    st.cno_ready = 0;
    if (kk == K - 1 && K > 1) {
      const double m = st.cno_s1 / K, zm = st.cno_z0 + m, zv = (st.cno_s2 - st.cno_s1 * m) / (K - 1);
      const double d = zm * zm - zv;
      double ratio;  // |Pav / (2*Nv)|
      if (d >= 0.0) {
        const double pav = sqrt(d);
        ratio = fabs(pav / (zm - pav));
      } else {  // Pav imaginary: |i*s / (Zm - i*s)|
        const double s2 = -d;
        ratio = sqrt(s2 / (zm * zm + s2));
      }
      st.cno_value = 10.0 * log10(ratio / p.cno_acc_time);
      st.cno_n = (e + 1) / K;
      st.cno_ready = 1;
      if (cno_slot >= 0 && st.cno_n >= 1 && st.cno_n <= dl->cno_nk) {
        dl->cno[cno_slot * dl->cno_nk + st.cno_n - 1] = st.cno_value;
        st.cno_ready = 0;
      }
    }
  }
  rec(GC_TRK_ABSOLUTE_SAMPLE, (double)st.pos);
  rec(GC_TRK_REM_CODE_PHASE, st.rem_code);
  rec(GC_TRK_REM_CARR_PHASE, st.rem_carr);
  const double rem_code_new = pre.rem_code_new;
  const double rem_carr_new = pre.rem_carr_new;
  double carr_err = atan(q_p / i_p) / (2.0 * kPi);                              // :305
  double code_err = (sqrt(i_e * i_e + q_e * q_e) - sqrt(i_l * i_l + q_l * q_l)) /
                    (sqrt(i_e * i_e + q_e * q_e) + sqrt(i_l * i_l + q_l * q_l));  // :322-323
  if constexpr (MAXARMS >= 2) {
  if (p.pilot_combine != 0 && arms >= 2) {
    double p6[6] = {sums[6], sums[7], sums[8], sums[9], sums[10], sums[11]};  // the pilot arm as correlated
    if constexpr (MAXARMS >= 3) {
      if (p.pilot_combine == 5 && arms >= 3) {
        // Galileo E1-C CBOC(6,1,1/11): arms {data, pilot BOC(1,1), pilot BOC(6,1)} folded in phase, then as mode 2 (track.hip)
        const double a11 = sqrt(10.0 / 11.0), a61 = -sqrt(1.0 / 11.0);
#pragma unroll
        for (int v = 0; v < 6; ++v) p6[v] = a11 * sums[6 + v] + a61 * sums[12 + v];
      } else if (p.pilot_combine == 4 && arms >= 3) {
        // BDS B1C wide-band: arms {data, pilot BOC(1,1), pilot BOC(6,1)} -> one pilot (WB_tracking.m:364-369; track.hip)
        const double a61 = -sqrt(4.0 / 33.0), a11 = sqrt(29.0 / 33.0);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const double i11 = sums[6 + 2 * x], q11 = sums[7 + 2 * x], i61 = sums[12 + 2 * x], q61 = sums[13 + 2 * x];
          p6[2 * x] = a61 * i61 + a11 * q11;
          p6[2 * x + 1] = a61 * q61 - a11 * i11;
        }
      }
    }
    const double pi_e = p6[0], pq_e = p6[1], pi_p = p6[2], pq_p = p6[3], pi_l = p6[4], pq_l = p6[5];
    double carr_err_q;
    if (p.pilot_combine == 1) {  // QI = (I_PQ + 1i*Q_PQ) * exp(-1i*pi/2), GPS_L5C tracking.m:340
      const double cr = cos(kPi / 2), ci = -sin(kPi / 2);
      const double re = pi_p * cr - pq_p * ci;
      const double im = pi_p * ci + pq_p * cr;
      carr_err_q = atan(im / re) / (2.0 * kPi);
    } else if (p.pilot_combine == 3) {
      carr_err_q = atan(-pi_p / pq_p) / (2.0 * kPi);  // BDS/B1C NB_tracking.m:341
    } else {
      carr_err_q = atan(pq_p / pi_p) / (2.0 * kPi);   // GAL_E1C tracking.m:309
    }
    double code_err_q = (sqrt(pi_e * pi_e + pq_e * pq_e) - sqrt(pi_l * pi_l + pq_l * pq_l)) /
                        (sqrt(pi_e * pi_e + pq_e * pq_e) + sqrt(pi_l * pi_l + pq_l * pq_l));
    const bool pll_w = p.pll_weight[0] != 0.0 || p.pll_weight[1] != 0.0;
    const bool dll_w = p.dll_weight[0] != 0.0 || p.dll_weight[1] != 0.0;
    if (p.dll_scale != 0.0) {
      code_err = code_err * p.dll_scale;
      code_err_q = code_err_q * p.dll_scale;
    }
    carr_err = pll_w ? (carr_err * p.pll_weight[0] + carr_err_q * p.pll_weight[1]) / (p.pll_weight[0] + p.pll_weight[1])
                     : (carr_err + carr_err_q) / 2;
    if (dll_w && p.pilot_combine == 4)  // codeError*factor + p_codeError*(1-factor), WB_tracking.m:403
      code_err = code_err * p.dll_weight[0] + code_err_q * p.dll_weight[1];
    else
      code_err = dll_w ? (code_err * p.dll_weight[0] + code_err_q * p.dll_weight[1]) / (p.dll_weight[0] + p.dll_weight[1])
                       : (code_err + code_err_q) / 2;
    rec(GC_TRK_PILOT_I_E, pi_e);
    rec(GC_TRK_PILOT_Q_E, pq_e);
    rec(GC_TRK_PILOT_I_P, pi_p);
    rec(GC_TRK_PILOT_Q_P, pq_p);
    rec(GC_TRK_PILOT_I_L, pi_l);
    rec(GC_TRK_PILOT_Q_L, pq_l);
  } else if (arms >= 2) {
#pragma unroll
    for (int v = 0; v < 6; ++v) rec(GC_TRK_PILOT_I_E + v, sums[6 + v]);
  }
  }
  double carr_nco;
  double old_carr_nco = st.old_carr_nco, old_carr_err = st.old_carr_err, d2 = st.d2_carr_err, d1 = st.d_carr_err;
  if (p.pll_kind == GC_PLL_2ND_ORDER) {
    carr_nco = old_carr_nco + dl->k1carr * (carr_err - old_carr_err) + carr_err * dl->k2carr;  // :308-309
    old_carr_nco = carr_nco;
    old_carr_err = carr_err;
  } else {
    d2 = d2 + carr_err * p.pf3;  // GPS_L5C tracking.m:351-353
    d1 = d2 + carr_err * p.pf2 + d1;
    carr_nco = d1 + carr_err * p.pf1;
  }
  rec(GC_TRK_CARR_FREQ, st.carr_freq);
  const double carr_freq_new = st.carr_basis + carr_nco;                        // :317
  const double code_nco = st.old_code_nco + dl->k1code * (code_err - st.old_code_err) + code_err * dl->k2code;
  rec(GC_TRK_CODE_FREQ, st.code_freq);
  const double code_freq_new = st.code_freq_basis - code_nco;                   // :335
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
  // next block geometry (:219-222) or the end
  const long long pos_new = st.pos + n;
  const double step_new = code_freq_new / p.sampling_freq;
  const int n_new = (int)ceil((p.code_length - rem_code_new) / step_new);
  int status = 0;
  if (e + 1 >= dl->n_epochs)
    status = 1;
  else if (!(step_new > 0.0) || !(step_new < 1e6) || !(carr_freq_new == carr_freq_new))
    status = 4;  // non-finite or non-positive code step (all-zero sums give atan(0/0) = NaN): no block can be cut from it
  else if (pos_new < 0 || (unsigned long long)(pos_new + n_new) > dl->if_nsamples)
    status = 2;
  st.pos = pos_new;
  st.rem_code = rem_code_new;
  st.rem_carr = rem_carr_new;
  st.carr_freq = carr_freq_new;
  st.code_freq = code_freq_new;
  st.old_code_nco = code_nco;
  st.old_code_err = code_err;
  st.old_carr_nco = old_carr_nco;
  st.old_carr_err = old_carr_err;
  st.d2_carr_err = d2;
  st.d_carr_err = d1;
  st.epochs_done = e + 1;
  st.status = status;
  if (p.table_phase_count > 0 && st.table_phase > 0) {  // GPS_L2C tracking.m:357-360, then :261 for the next block
    if (p.pilot_combine != 0) {
      st.table_phase += 1;
      if (st.table_phase >= p.table_phase_count + 1) st.table_phase = 1;
    }
    b.table_offset[1] = (int)p.code_length * (st.table_phase - 1);
  }
  b.blksize = n_new;
  b.first_sample = pos_new;
  b.rem_code_phase = rem_code_new;
  b.code_phase_step = step_new;
  b.carr_freq = carr_freq_new;
  b.rem_carr_phase = rem_carr_new;
  return status;
}

// The epoch's record and the loop state to device memory (lane 0), off the critical path.
// spread: every member of an all-gather team holds the same record (it ran the same closure on the same sums), so in teams of
// 17 members or more member f + 2 stores field f - one store instead of fifteen by the member whose partial sums the next epoch
// then waits for (5.77 us per epoch against 5.90; selecting the fields with a runtime modulo instead of a compare cost 0.5 us).
// Member 0 keeps the state.
__device__ inline void devloop_commit(const DevLoopArgs* __restrict__ dl, DevLoopChan* gch, const DevLoopChan& st, long long slot, int e,
                                      const double (&rv)[GC_TRK_NFIELDS], int arms, int lane, int member = 0, bool spread = false) {
  if (lane != 0) return;
  double* o = dl->records + (size_t)slot * GC_TRK_NFIELDS * dl->n_epochs;
  const int nf = (arms >= 2) ? GC_TRK_NFIELDS : GC_TRK_PILOT_I_E;  // the pilot fields follow the fifteen common ones
#pragma unroll
  for (int f = 0; f < GC_TRK_NFIELDS; ++f)
    if (f < nf && (!spread || f + 2 == member)) o[(size_t)f * dl->n_epochs + e] = rv[f];
  if (member != 0) return;
  // the state in device memory is what the host reads when the launch has ended (status, epochs done, the loop state a resumed
  // call starts from): the last epoch's is the one that counts; every 256th keeps a hung run diagnosable
  if (spread && st.status == 0 && (e & 255) != 255) return;
  gch->pos = st.pos;
  gch->rem_code = st.rem_code;
  gch->rem_carr = st.rem_carr;
  gch->carr_freq = st.carr_freq;
  gch->code_freq = st.code_freq;
  gch->old_code_nco = st.old_code_nco;
  gch->old_code_err = st.old_code_err;
  gch->old_carr_nco = st.old_carr_nco;
  gch->old_carr_err = st.old_carr_err;
  gch->d2_carr_err = st.d2_carr_err;
  gch->d_carr_err = st.d_carr_err;
  gch->epochs_done = st.epochs_done;
  gch->status = st.status;
  gch->table_phase = st.table_phase;
  if (dl->cno && st.cno_ready && st.cno_n >= 1 && st.cno_n <= dl->cno_nk) dl->cno[slot * dl->cno_nk + st.cno_n - 1] = st.cno_value;
}

// All parts back to back on the state held in device memory (lane kernel's closer): state read and written in place, records
// stored as they are formed.
template <int MAXARMS>
__device__ inline int devloop_close(const DevLoopArgs* __restrict__ dl, DevLoopChan* ch, gc_block& b, long long slot, int e,
                                    const double (&sums)[6 * MAXARMS], int arms, double R, int lane) {
  double* o = dl->records + (size_t)slot * GC_TRK_NFIELDS * dl->n_epochs;
  const DevLoopPre pre = devloop_pre(dl, *ch, b, R);
  return devloop_post<MAXARMS>(dl, *ch, b, e, sums, arms, R, pre, [&](int f, double v) {
    if (lane == 0) o[(size_t)f * dl->n_epochs + e] = v;
  });
}
#endif

}  // namespace gcorr
