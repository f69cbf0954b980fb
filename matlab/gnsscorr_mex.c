/*
 * gnsscorr_mex.c — MEX gateway: MATLAB <-> libgnsscorr.so (include/gnsscorr.h).
 *
 * This is the reference-side binding a maintainer of CU-SDR-Collection adds so that
 * include/tracking.m and include/acquisition.m can hand their inner loops to the GPU
 * (INTEGRATION.md).  It is a pure translation layer (mxArray <-> C structs); every computation
 * lives behind the C-ABI.  MATLAB/mex.h do not exist in the build container, so this file is not
 * compiled by __graft_entry__.build(); build it on a MATLAB host with
 *
 *     mex -R2018a -I<repo>/include gnsscorr_mex.c -L<repo>/cu-sdr-collection_amd/lib -lgnsscorr
 *
 * Usage:  out = gnsscorr_mex(cmd, args...)
 *   h    = gnsscorr_mex('create', device_id)
 *          gnsscorr_mex('destroy', h)
 *          gnsscorr_mex('open_if_file', h, fileName, skipBytes, nSamples, dataType, fileType, samplingFreq[, 'IQ'|'QI'])
 *          gnsscorr_mex('share_if', hDst, hSrc)        % hDst reads hSrc's record (same GPU, no copy): one record, several packages
 *          gnsscorr_mex('load_if', h, int8_or_int16_vector, fileType, samplingFreq)
 *          gnsscorr_mex('set_channel', h, channelIdx0, {paddedCode1, ...}, indexScale)
 *   sums = gnsscorr_mex('correlate', h, blocks)      % blocks: 8 x nblocks double, see below
 *   [trk, epochs, status] = gnsscorr_mex('track', h, params_struct, channels)   % channels: 5 x nch
 *   res  = gnsscorr_mex('acquire_coarse', h, acq_struct, sampledCodes)          % int8 spc x nprn
 *   f    = gnsscorr_mex('acquire_fine_l1ca', h, acq_struct, caCode, codePhase, coarseFreq)
 *   [trk, epochs, status] = gnsscorr_mex('track_device', h, params_struct, channels)  % loop closed on the GPU; status
 *          GC_E_UNSUPPORTED (-6) for configurations it does not cover: fall back to 'track'
 *          gnsscorr_mex('load_if_packed2', h, uint8(packed))                     % 2-bit packed complex record
 *   s    = gnsscorr_mex('fine_sums', h, fine_struct, int8(code))                 % (2*ncodes) x nbins per-code sums
 *   c    = gnsscorr_mex('preamble_xcorr', h, I_P, int8(preamble_ms))             % single, non-negative lags
 *          gnsscorr_mex('set_channel', h, ch, {codes...}, indexScale, armMult, windowEntries)   % optional per-arm vectors:
 *          armMult = [1 1 6] for a BOC(6,1) arm (WB_tracking.m:293), windowEntries for GPS L2C CL (gc_set_code_window)
 *   res  = gnsscorr_mex('acquire_coarse_multi', h, acq_struct, sampledCodes, narms)    % int8 spc x (nprn*narms), data+pilot
 *   nrows = gnsscorr_mex('acq_shift_prepare', h, shift_struct)                    % circular-shift family (BDS B1I, GPS L2C, BDS B1C)
 *   [rowMax, rowArg] = gnsscorr_mex('acq_shift_search', h, int8(codes), weights, nrows)  % codes n x narms; weights [] = ones
 *   row  = gnsscorr_mex('acq_shift_row', h, row0, n)                              % one results row (single), for the 2nd-peak rule
 *   picks = gnsscorr_mex('acq_shift_search_batch', h, int8(chips), int32(index0), weights, rule, exclude, period, narms)  % a package's whole search: 4 x nPRN
 *   x    = gnsscorr_mex('read_if', h, firstSample0, n, 'int8'|'int16', valuesPerSample)   % raw record samples back
 *   [name, cus] = gnsscorr_mex('device_info', h)
 *   [ties, maxDev, eps] = gnsscorr_mex('acq_guard_stats', h)                         % float64 guard of the last search (gc_acq_guard_stats)
 *   n    = gnsscorr_mex('device_count')                                              % HIP devices visible: one context per device
 */
#include <string.h>

#include "gnsscorr.h"
#include "mex.h"

static gc_context* g_ctx[16];

static void fail(const char* where) { mexErrMsgIdAndTxt("gnsscorr:error", "%s: %s", where, gc_last_error()); }

static void at_exit(void) {
  for (int i = 0; i < 16; ++i)
    if (g_ctx[i]) {
      gc_destroy(g_ctx[i]);
      g_ctx[i] = NULL;
    }
}

static gc_context* handle(const mxArray* a) {
  int i = (int)mxGetScalar(a);
  if (i < 0 || i >= 16 || !g_ctx[i]) mexErrMsgIdAndTxt("gnsscorr:handle", "invalid context handle");
  return g_ctx[i];
}

static double field(const mxArray* s, const char* name) {
  const mxArray* f = mxGetField(s, 0, name);
  if (!f) mexErrMsgIdAndTxt("gnsscorr:field", "missing field %s", name);
  return mxGetScalar(f);
}

static void fill_acq(const mxArray* s, gc_acq_params* p) {
  memset(p, 0, sizeof *p);
  p->sampling_freq = field(s, "samplingFreq");
  p->code_freq_basis = field(s, "codeFreqBasis");
  p->code_length = field(s, "codeLength");
  p->intermediate_freq = field(s, "IF");
  p->search_band = field(s, "acqSearchBand");
  p->search_step = field(s, "acqSearchStep");
  p->non_coh_time = (int32_t)field(s, "acqNonCohTime");
  p->first_sample = (int64_t)field(s, "firstSample");
  p->source = mxGetField(s, 0, "source") ? (int32_t)field(s, "source") : 0; /* 1: the signal 'acq_condition' left on the device */
  /* optional: a B1C-type search as a carrier-per-bin search (gc_acq_params.block_len ...) */
  if (mxGetField(s, 0, "blockLen")) p->block_len = (int32_t)field(s, "blockLen");
  if (mxGetField(s, 0, "codeSamples")) p->code_samples = (int32_t)field(s, "codeSamples");
  if (mxGetField(s, 0, "nBins")) p->n_bins = (int32_t)field(s, "nBins");
  if (mxGetField(s, 0, "armWeight")) {
    const mxArray* w = mxGetField(s, 0, "armWeight");
    for (mwSize i = 0; i < mxGetNumberOfElements(w) && i < 4; ++i) p->arm_weight[i] = mxGetDoubles(w)[i];
  }
}

/* gc_track_params from the struct gnsscorr_tracking.m builds (settings fields + the per-package extras) */
static void track_params_from(const mxArray* s, gc_track_params* out) {
  gc_track_params p;
  memset(&p, 0, sizeof p);
  p.sampling_freq = field(s, "samplingFreq");
  p.code_freq_basis = field(s, "codeFreqBasis");
  p.code_length = field(s, "codeLength");
  p.el_spacing = field(s, "dllCorrelatorSpacing");
  p.int_time = field(s, "intTime");
  p.dll_noise_bw = field(s, "dllNoiseBandwidth");
  p.dll_damping = field(s, "dllDampingRatio");
  p.pll_noise_bw = field(s, "pllNoiseBandwidth");
  p.pll_damping = field(s, "pllDampingRatio");
  p.pll_kind = mxGetField(s, 0, "pllKind") ? (int32_t)field(s, "pllKind") : GC_PLL_2ND_ORDER;
  /* skipSamples: where the record's first sample sits, in SAMPLES (tracking.m:145-153: skipNumberOfBytes for schar components,
     skipNumberOfBytes/2 for int16 components); a plain settings struct still works for schar records */
  p.skip_samples = (int64_t)(mxGetField(s, 0, "skipSamples") ? field(s, "skipSamples") : field(s, "skipNumberOfBytes"));
  p.n_epochs = (int32_t)(mxGetField(s, 0, "numEpochs") ? field(s, "numEpochs") : field(s, "msToProcess"));
  /* optional (API v2): pilot handling of the multi-component packages, see gc_track_params in gnsscorr.h */
  if (mxGetField(s, 0, "pilotCombine")) p.pilot_combine = (int32_t)field(s, "pilotCombine");
  if (mxGetField(s, 0, "pf1")) { p.pf1 = field(s, "pf1"); p.pf2 = field(s, "pf2"); p.pf3 = field(s, "pf3"); }
  if (mxGetField(s, 0, "pllWeight")) { const double* w = mxGetDoubles(mxGetField(s, 0, "pllWeight")); p.pll_weight[0] = w[0]; p.pll_weight[1] = w[1]; }
  if (mxGetField(s, 0, "dllWeight")) { const double* w = mxGetDoubles(mxGetField(s, 0, "dllWeight")); p.dll_weight[0] = w[0]; p.dll_weight[1] = w[1]; }
  if (mxGetField(s, 0, "dllScale")) p.dll_scale = field(s, "dllScale");
  if (mxGetField(s, 0, "tablePhaseCount")) p.table_phase_count = (int32_t)field(s, "tablePhaseCount");
  /* optional (API v3): C/N0 by CNoVSM inside the loop every cnoInterval epochs (settings.CNo.VSMinterval, .accTime) -> 4th output */
  if (mxGetField(s, 0, "cnoInterval")) { p.cno_interval = (int32_t)field(s, "cnoInterval"); p.cno_acc_time = field(s, "cnoAccTime"); }
  /* cnoMode (gc_cno_mode): 0 CNoVSM; 1..3 Calc_CNo_PLD of BDS B2a / B1C (no pilot / pilot pair swapped / straight), cnoAccTime = intTime */
  if (mxGetField(s, 0, "cnoMode")) p.cno_mode = (int32_t)field(s, "cnoMode");
  *out = p;
}

/* channels: 5 (or 6) x nch rows = channel index, PRN, acquiredFreq, codeFreq, codePhase[, CLCodePhase] (preRun.m:65-73) */
static gc_channel_init* channel_inits_from(const mxArray* a, int* nch_out) {
  const int crow = (int)mxGetM(a);
  const int nch = (int)mxGetN(a);
  const double* c = mxGetDoubles(a);
  gc_channel_init* init = (gc_channel_init*)mxCalloc((size_t)nch, sizeof *init);
  for (int i = 0; i < nch; ++i) {
    init[i].channel = (int32_t)c[crow * i];
    init[i].prn = (int32_t)c[crow * i + 1];
    init[i].acquired_freq = c[crow * i + 2];
    init[i].code_freq = c[crow * i + 3];
    init[i].code_phase = (int64_t)c[crow * i + 4];
    if (crow >= 6) init[i].table_phase = (int32_t)c[crow * i + 5];
  }
  *nch_out = nch;
  return init;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  char cmd[32];
  if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("gnsscorr:usage", "first argument: command string");
  mexAtExit(at_exit);

  if (!strcmp(cmd, "create")) {
    int slot = -1;
    for (int i = 0; i < 16 && slot < 0; ++i)
      if (!g_ctx[i]) slot = i;
    if (slot < 0) mexErrMsgIdAndTxt("gnsscorr:handle", "too many contexts");
    if (gc_create(&g_ctx[slot], nrhs > 1 ? (int)mxGetScalar(prhs[1]) : 0)) fail("gc_create");
    mexLock();
    plhs[0] = mxCreateDoubleScalar(slot);
  } else if (!strcmp(cmd, "destroy")) {
    int i = (int)mxGetScalar(prhs[1]);
    handle(prhs[1]);
    gc_destroy(g_ctx[i]);
    g_ctx[i] = NULL;
    mexUnlock();
  } else if (!strcmp(cmd, "open_if_file")) {
    /* settings.fileName, skipNumberOfBytes*dataAdaptCoeff, dataType 'schar'|'int16', fileType 1|2 */
    char path[4096], dtype[16];
    mxGetString(prhs[2], path, sizeof path);
    mxGetString(prhs[5], dtype, sizeof dtype);
    int layout = (int)mxGetScalar(prhs[6]) == 1 ? GC_REAL : GC_IQ;
    if (nrhs > 8 && layout == GC_IQ) { /* optional sample order 'IQ' | 'QI' (GLONASS front ends: GLO_GL1/include/tracking.m:227) */
      char order[8] = "";
      mxGetString(prhs[8], order, sizeof order);
      if (!strcmp(order, "QI")) layout = GC_QI;
    }
    if (gc_open_if_file(handle(prhs[1]), path, (uint64_t)mxGetScalar(prhs[3]), (uint64_t)mxGetScalar(prhs[4]),
                        !strcmp(dtype, "int16") ? GC_I16 : GC_I8, layout))
      fail("gc_open_if_file");
    if (gc_set_sampling_freq(handle(prhs[1]), mxGetScalar(prhs[7]))) fail("gc_set_sampling_freq");
  } else if (!strcmp(cmd, "share_if")) {
    if (gc_share_if(handle(prhs[1]), handle(prhs[2]))) fail("gc_share_if");
  } else if (!strcmp(cmd, "load_if")) {
    int dt = mxIsInt16(prhs[2]) ? GC_I16 : GC_I8;
    if (!mxIsInt8(prhs[2]) && !mxIsInt16(prhs[2])) mexErrMsgIdAndTxt("gnsscorr:type", "IF samples must be int8 or int16");
    int layout = (int)mxGetScalar(prhs[3]) == 1 ? GC_REAL : GC_IQ;
    uint64_t n = (uint64_t)mxGetNumberOfElements(prhs[2]) / (layout == GC_REAL ? 1 : 2);
    if (gc_load_if(handle(prhs[1]), mxGetData(prhs[2]), n, dt, layout)) fail("gc_load_if");
    if (gc_set_sampling_freq(handle(prhs[1]), mxGetScalar(prhs[4]))) fail("gc_set_sampling_freq");
  } else if (!strcmp(cmd, "set_channel")) {
    /* cell array of padded code tables [c(end) c c(1)] as built at tracking.m:158 (double or int8) */
    gc_context* c = handle(prhs[1]);
    int ch = (int)mxGetScalar(prhs[2]);
    int arms = (int)mxGetNumberOfElements(prhs[3]);
    if (gc_set_channel(c, ch, arms, nrhs > 4 ? mxGetScalar(prhs[4]) : 1.0)) fail("gc_set_channel");
    for (int a = 0; a < arms; ++a) {
      const mxArray* t = mxGetCell(prhs[3], a);
      int n = (int)mxGetNumberOfElements(t);
      int8_t* tmp = (int8_t*)mxMalloc((size_t)n);
      if (mxIsDouble(t)) {
        const double* d = mxGetDoubles(t);
        for (int i = 0; i < n; ++i) tmp[i] = (int8_t)d[i];
      } else {
        memcpy(tmp, mxGetData(t), (size_t)n);
      }
      /* optional: per-arm ramp multipliers (6 for a BOC(6,1) arm, WB_tracking.m:293) and table windows (GPS L2C CL) */
      double mult = nrhs > 5 && (int)mxGetNumberOfElements(prhs[5]) > a ? mxGetDoubles(prhs[5])[a] : 1.0;
      int rc = gc_set_code(c, ch, a, tmp, n, mult);
      mxFree(tmp);
      if (rc) fail("gc_set_code");
      if (nrhs > 6 && (int)mxGetNumberOfElements(prhs[6]) > a && mxGetDoubles(prhs[6])[a] > 0)
        if (gc_set_code_window(c, ch, a, (int)mxGetDoubles(prhs[6])[a])) fail("gc_set_code_window");
    }
  } else if (!strcmp(cmd, "correlate")) {
    /* blocks rows: channel, first_sample (0-based), blksize, remCodePhase, codePhaseStep,
     * earlyLateSpc, carrFreq, remCarrPhase — the quantities of tracking.m:212-222,249,277 */
    const double* b = mxGetDoubles(prhs[2]);
    int n = (int)mxGetN(prhs[2]);
    if (mxGetM(prhs[2]) != 8) mexErrMsgIdAndTxt("gnsscorr:usage", "blocks must be 8 x nblocks");
    gc_block* blk = (gc_block*)mxCalloc((size_t)n, sizeof(gc_block));
    for (int i = 0; i < n; ++i) {
      const double* r = b + 8 * i;
      blk[i].channel = (int32_t)r[0];
      blk[i].first_sample = (int64_t)r[1];
      blk[i].blksize = (int32_t)r[2];
      blk[i].rem_code_phase = r[3];
      blk[i].code_phase_step = r[4];
      blk[i].el_spacing = r[5];
      blk[i].carr_freq = r[6];
      blk[i].rem_carr_phase = r[7];
    }
    plhs[0] = mxCreateDoubleMatrix(GC_OUT_STRIDE, (mwSize)n, mxREAL); /* rows: I_E Q_E I_P Q_P I_L Q_L per arm */
    int rc = gc_correlate(handle(prhs[1]), n, blk, mxGetDoubles(plhs[0]));
    mxFree(blk);
    if (rc == GC_E_RANGE) mexErrMsgIdAndTxt("gnsscorr:range", "%s", gc_last_error()); /* tracking.m:241-245 */
    if (rc) fail("gc_correlate");
  } else if (!strcmp(cmd, "track") || !strcmp(cmd, "track_device") || !strcmp(cmd, "track_file")) {
    /* [trk, epochs, status] = gnsscorr_mex('track', h, p, chanTable)
       [trk, epochs, status] = gnsscorr_mex('track_file', h, p, chanTable, fileName, windowSamples, dataType, fileType[, 'QI']):
       the same on a file of any size, at most 2 * windowSamples samples resident (gc_track_file) */
    gc_track_params p;
    track_params_from(prhs[2], &p);
    int nch = 0;
    gc_channel_init* init = channel_inits_from(prhs[3], &nch);
    /* trk(epoch, (channel-1)*GC_TRK_NFIELDS + field): one column per (channel, field), fields in gc_track_field order */
    plhs[0] = mxCreateDoubleMatrix((mwSize)p.n_epochs, (mwSize)GC_TRK_NFIELDS * (mwSize)nch, mxREAL);
    int32_t* done = (int32_t*)mxCalloc((size_t)nch, sizeof *done);
    /* [trk, epochs, status, cno] = ...: cno(k, channel) = CNoVSM of the k-th interval (p.cnoInterval > 1), computed inside the loop;
       with p.cnoMode > 0: cno(j, k, channel), j = DataCNo, PilotCNo, combined CNo, DataPLD, PilotPLD (Calc_CNo_PLD.m + tracking.m:409-432) */
    const mwSize nk = p.cno_interval > 1 ? (mwSize)(p.n_epochs / p.cno_interval) : 0;
    const mwSize nv = p.cno_mode != GC_CNO_VSM ? GC_CNO_NPLD : 1;
    mxArray* cno = NULL;
    if (nlhs > 3 && nk > 0) {
      cno = mxCreateDoubleMatrix(nv * nk, (mwSize)nch, mxREAL);  /* column-major: element (j + nv*k, channel) */
      if (gc_set_cno_output(handle(prhs[1]), mxGetDoubles(cno), (int64_t)(nv * nk * (mwSize)nch))) fail("gc_set_cno_output");
    }
    int rc;
    if (!strcmp(cmd, "track_file")) {
      char path[4096], dtype[16];
      mxGetString(prhs[4], path, sizeof path);
      mxGetString(prhs[6], dtype, sizeof dtype);
      int layout = (int)mxGetScalar(prhs[7]) == 1 ? GC_REAL : GC_IQ;
      if (nrhs > 8 && layout == GC_IQ) {
        char order[8] = "";
        mxGetString(prhs[8], order, sizeof order);
        if (!strcmp(order, "QI")) layout = GC_QI;
      }
      if (gc_set_sampling_freq(handle(prhs[1]), p.sampling_freq)) fail("gc_set_sampling_freq");
      rc = gc_track_file(handle(prhs[1]), path, 0, !strcmp(dtype, "int16") ? GC_I16 : GC_I8, layout, (uint64_t)mxGetScalar(prhs[5]), &p,
                         nch, init, mxGetDoubles(plhs[0]), done);
    } else {
      /* track_device: the same loop closed on the GPU in one persistent launch (gc_track_device); GC_E_UNSUPPORTED for the
         configurations it does not cover - the caller then falls back to 'track' */
      rc = !strcmp(cmd, "track_device") ? gc_track_device(handle(prhs[1]), &p, nch, init, mxGetDoubles(plhs[0]), done)
                                        : gc_track(handle(prhs[1]), &p, nch, init, mxGetDoubles(plhs[0]), done);
    }
    if (nlhs > 1) {
      plhs[1] = mxCreateDoubleMatrix(1, (mwSize)nch, mxREAL);
      for (int i = 0; i < nch; ++i) mxGetDoubles(plhs[1])[i] = done[i];
    }
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(rc);
    if (cno) (void)gc_set_cno_output(handle(prhs[1]), NULL, 0);
    if (nlhs > 3) plhs[3] = cno ? cno : mxCreateDoubleMatrix(0, 0, mxREAL);
    mxFree(init);
    mxFree(done);
    if (rc && rc != GC_E_RANGE && !(rc == GC_E_UNSUPPORTED && nlhs > 2)) fail("gc_track"); /* GC_E_RANGE = the reference's short-read return */
  } else if (!strcmp(cmd, "acquire_coarse") || !strcmp(cmd, "acquire_coarse_multi")) {
    /* acquire_coarse_multi: narms sampled codes per PRN in consecutive columns (data, pilot: GPS_L5C/include/acquisition.m:175-216) */
    gc_acq_params p;
    fill_acq(prhs[2], &p);
    const int narms = !strcmp(cmd, "acquire_coarse_multi") ? (int)mxGetScalar(prhs[4]) : 1;
    if (narms < 1 || mxGetN(prhs[3]) % (mwSize)narms) mexErrMsgIdAndTxt("gnsscorr:usage", "sampledCodes: spc x (nprn*narms)");
    int nprn = (int)mxGetN(prhs[3]) / narms;
    gc_acq_result* r = (gc_acq_result*)mxCalloc((size_t)nprn, sizeof *r);
    if (gc_acquire_coarse_multi(handle(prhs[1]), &p, nprn, narms, (const int8_t*)mxGetData(prhs[3]), r)) fail("gc_acquire_coarse_multi");
    plhs[0] = mxCreateDoubleMatrix(5, (mwSize)nprn, mxREAL); /* rows: bin, codePhase, peak, peakMetric, coarseFreq */
    for (int i = 0; i < nprn; ++i) {
      double* o = mxGetDoubles(plhs[0]) + 5 * i;
      o[0] = r[i].coarse_bin;
      o[1] = r[i].code_phase;
      o[2] = r[i].peak;
      o[3] = r[i].peak_metric;
      o[4] = r[i].coarse_freq;
    }
    mxFree(r);
  } else if (!strcmp(cmd, "acquire_fine_l1ca")) {
    gc_acq_params p;
    fill_acq(prhs[2], &p);
    double f = 0;
    if (gc_acquire_fine_l1ca(handle(prhs[1]), &p, (const int8_t*)mxGetData(prhs[3]), (int)mxGetScalar(prhs[4]),
                             mxGetScalar(prhs[5]), &f))
      fail("gc_acquire_fine_l1ca");
    plhs[0] = mxCreateDoubleScalar(f);
  } else if (!strcmp(cmd, "acq_condition")) {
    /* [newFs, newIF, n] = gnsscorr_mex('acq_condition', h, struct(samplingFreq, IF, bandwidth, firstSample, nSamples[, firOrder])):
       acquisition.m:46-111 on the GPU - zero-phase FIR band-pass + band-pass-sampling decimation of longSignal; later searches
       with .source = 1 read the conditioned signal */
    const mxArray* s = prhs[2];
    gc_acq_front_params p;
    gc_acq_front_result r;
    memset(&p, 0, sizeof p);
    p.sampling_freq = field(s, "samplingFreq");
    p.intermediate_freq = field(s, "IF");
    p.bandwidth = field(s, "bandwidth");
    p.first_sample = (int64_t)field(s, "firstSample");
    p.n_samples = (int64_t)field(s, "nSamples");
    p.fir_order = mxGetField(s, 0, "firOrder") ? (int32_t)field(s, "firOrder") : 700;
    p.band_margin = mxGetField(s, 0, "bandMargin") ? field(s, "bandMargin") : 0.0;
    if (gc_acq_condition(handle(prhs[1]), &p, &r)) fail("gc_acq_condition");
    plhs[0] = mxCreateDoubleScalar(r.sampling_freq);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(r.intermediate_freq);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar((double)r.n_samples);
  } else if (!strcmp(cmd, "load_if_packed2")) {
    /* gnsscorr_mex('load_if_packed2', h, uint8(packed)): 2-bit packed complex samples (unpack_cplx.m:32-49) expanded on the GPU */
    if (gc_load_if_packed2(handle(prhs[1]), mxGetData(prhs[2]), (uint64_t)mxGetNumberOfElements(prhs[2]))) fail("gc_load_if_packed2");
  } else if (!strcmp(cmd, "fine_sums")) {
    /* s = gnsscorr_mex('fine_sums', h, fineParams, int8(code)): sumPerCode(bin, code) of the packages' fine-frequency stages.
       codeFreq = 0: `code` is a replica already sampled, one entry per sample.  Several replicas at once: int8 codeLength x nrep,
       s = 2*ncodes x (nbins*nrep), replica-major.  dcRe / dcIm: the mean removed from the samples first (GPS_L2C acquisition.m:144) */
    const mxArray* s = prhs[2];
    gc_fine_params p;
    memset(&p, 0, sizeof p);
    p.sampling_freq = field(s, "samplingFreq");
    p.code_freq = field(s, "codeFreq");
    p.f0 = field(s, "f0");
    p.fstep = field(s, "fstep");
    p.first_sample = (int64_t)field(s, "firstSample");
    p.spc = (int32_t)field(s, "samplesPerCode");
    p.ncodes = (int32_t)field(s, "ncodes");
    p.nbins = (int32_t)field(s, "nbins");
    p.code_len = (int32_t)field(s, "codeLength");
    p.index_offset = mxGetField(s, 0, "indexOffset") ? (int32_t)field(s, "indexOffset") : 0;
    p.source = mxGetField(s, 0, "source") ? (int32_t)field(s, "source") : 0;
    if (mxGetField(s, 0, "dcRe")) { p.dc_re = field(s, "dcRe"); p.dc_im = field(s, "dcIm"); }
    const mwSize nrep = mxGetNumberOfElements(prhs[3]) / (mwSize)p.code_len;
    if (nrep < 1 || nrep * (mwSize)p.code_len != mxGetNumberOfElements(prhs[3]))
      mexErrMsgIdAndTxt("gnsscorr:args", "fine_sums: the code holds %d entries, not a multiple of codeLength = %d",
                        (int)mxGetNumberOfElements(prhs[3]), (int)p.code_len);
    plhs[0] = mxCreateDoubleMatrix((mwSize)(2 * p.ncodes), (mwSize)p.nbins * nrep, mxREAL); /* (re, im) pairs per code, one column per bin */
    int64_t* first = (int64_t*)mxCalloc(nrep, sizeof *first);
    double* f0 = (double*)mxCalloc(nrep, sizeof *f0);
    for (mwSize i = 0; i < nrep; ++i) { first[i] = p.first_sample; f0[i] = p.f0; }
    const int rc = gc_acquire_fine_sums_batch(handle(prhs[1]), &p, (int)nrep, (const int8_t*)mxGetData(prhs[3]), first, f0, mxGetDoubles(plhs[0]));
    mxFree(first);
    mxFree(f0);
    if (rc) fail("gc_acquire_fine_sums");
  } else if (!strcmp(cmd, "acq_from_record")) {
    /* gnsscorr_mex('acq_from_record', h, firstSample, n): the record's samples (int16, Q/I, real: any format) as the complex float
       signal the searches read with .source = 1 */
    if (gc_acq_signal_from_record(handle(prhs[1]), (int64_t)mxGetScalar(prhs[2]), (int64_t)mxGetScalar(prhs[3]))) fail("gc_acq_signal_from_record");
  } else if (!strcmp(cmd, "acq_set_signal")) {
    /* gnsscorr_mex('acq_set_signal', h, single([re; im])): longSignal itself - any complex row - as that signal */
    if (!mxIsSingle(prhs[2])) mexErrMsgIdAndTxt("gnsscorr:args", "acq_set_signal: single([re; im]) expected");
    if (gc_acq_set_signal(handle(prhs[1]), (const float*)mxGetData(prhs[2]), (int64_t)(mxGetNumberOfElements(prhs[2]) / 2))) fail("gc_acq_set_signal");
  } else if (!strcmp(cmd, "signal_stats")) {
    /* s = gnsscorr_mex('signal_stats', h, firstSample, n[, source]): [real(mean(x)), imag(mean(x)), var(x)] of n samples on the device */
    double mr, mi, v;
    if (gc_acq_signal_stats(handle(prhs[1]), (int64_t)mxGetScalar(prhs[2]), (int64_t)mxGetScalar(prhs[3]),
                            nrhs > 4 ? (int32_t)mxGetScalar(prhs[4]) : 0, &mr, &mi, &v))
      fail("gc_acq_signal_stats");
    plhs[0] = mxCreateDoubleMatrix(1, 3, mxREAL);
    mxGetDoubles(plhs[0])[0] = mr;
    mxGetDoubles(plhs[0])[1] = mi;
    mxGetDoubles(plhs[0])[2] = v;
  } else if (!strcmp(cmd, "preamble_xcorr") || !strcmp(cmd, "sync_xcorr")) {
    /* c = gnsscorr_mex('sync_xcorr', h, I_P, int8(pattern)[, zeroIsPlus]): the non-negative lags of xcorr(hard-limited I_P, pattern)
       of every package's NAVdecoding.m (GPS_L1CA :69-85 ...); zeroIsPlus = 1: Galileo E1's bits = (I_P < 0) */
    if (nrhs < 4 || !mxIsDouble(prhs[2]) || !mxIsInt8(prhs[3]))
      mexErrMsgIdAndTxt("gnsscorr:args", "sync_xcorr: (h, double I_P, int8 pattern[, zeroIsPlus])");
    const mwSize n = mxGetNumberOfElements(prhs[2]);
    const int flags = (nrhs > 4 && mxGetScalar(prhs[4]) != 0) ? GC_SYNC_ZERO_IS_PLUS : 0;
    plhs[0] = mxCreateNumericMatrix(1, n, mxSINGLE_CLASS, mxREAL);
    if (gc_sync_xcorr(handle(prhs[1]), mxGetDoubles(prhs[2]), (int64_t)n, (const int8_t*)mxGetData(prhs[3]),
                      (int)mxGetNumberOfElements(prhs[3]), flags, (float*)mxGetData(plhs[0])))
      fail("gc_sync_xcorr");
  } else if (!strcmp(cmd, "acq_shift_prepare")) {
    /* the circular-shift search of BDS/B1I acquisition.m:100-167, GPS_L2C :119-190, BDS/B1C :120-200: wipe-off + forward
       transforms of the signal blocks, once per call */
    const mxArray* s = prhs[2];
    gc_acq_shift_params p;
    memset(&p, 0, sizeof p);
    p.sampling_freq = field(s, "samplingFreq");
    p.carrier_f0 = field(s, "carrierF0");
    p.carrier_step = field(s, "carrierStep");
    p.first_sample = (int64_t)field(s, "firstSample");
    p.n = (int32_t)field(s, "samplesPerBlock");
    p.n_signals = (int32_t)field(s, "nSignals");
    p.n_carriers = (int32_t)field(s, "nCarriers");
    p.n_bins = (int32_t)field(s, "nBins");
    p.n_arms_max = mxGetField(s, 0, "nArmsMax") ? (int32_t)field(s, "nArmsMax") : 1;
    p.source = mxGetField(s, 0, "source") ? (int32_t)field(s, "source") : 0;
    if (gc_acq_shift_prepare(handle(prhs[1]), &p)) fail("gc_acq_shift_prepare");
    plhs[0] = mxCreateDoubleScalar((double)p.n_carriers * p.n_signals * p.n_bins); /* number of result rows */
  } else if (!strcmp(cmd, "acq_shift_search")) {
    /* codes: int8 n x narms, zero-padded replicas; rows ordered ((carrier*nSignals + signal)*nBins + bin).  The outputs are sized
       from what the context prepared (gc_acq_shift_dims), never from the caller's numbers: the library writes `rows` values and
       reads n * narms code samples whatever the arguments say. */
    int32_t n = 0, nrows = 0, amax = 0;
    if (nrhs < 3) mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search: handle and codes are needed");
    if (gc_acq_shift_dims(handle(prhs[1]), &n, &nrows, &amax)) fail("gc_acq_shift_dims");
    const int narms = (int)mxGetN(prhs[2]);
    if (!mxIsInt8(prhs[2]) || (int32_t)mxGetM(prhs[2]) != n || narms < 1 || narms > amax)
      mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search: codes must be int8 of %d rows (samplesPerBlock) and 1..%d columns", (int)n, (int)amax);
    if (nrhs > 4 && (int32_t)mxGetScalar(prhs[4]) != nrows)
      mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search: the prepared search has %d rows, not %d", (int)nrows, (int)mxGetScalar(prhs[4]));
    const double* w = NULL;
    if (nrhs > 3 && !mxIsEmpty(prhs[3])) {
      if (!mxIsDouble(prhs[3]) || (int)mxGetNumberOfElements(prhs[3]) < narms)
        mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search: one double weight per code arm");
      w = mxGetDoubles(prhs[3]);
    }
    plhs[0] = mxCreateNumericMatrix(1, (mwSize)nrows, mxSINGLE_CLASS, mxREAL);
    mxArray* arg = mxCreateNumericMatrix(1, (mwSize)nrows, mxINT32_CLASS, mxREAL);
    if (gc_acq_shift_search(handle(prhs[1]), narms, (const int8_t*)mxGetData(prhs[2]), w, (float*)mxGetData(plhs[0]),
                            (int32_t*)mxGetData(arg)))
      fail("gc_acq_shift_search");
    if (nlhs > 1) plhs[1] = arg; /* 0-based first position of each row's maximum */
    else mxDestroyArray(arg);
  } else if (!strcmp(cmd, "acq_shift_search_batch")) {
    /* picks = gnsscorr_mex('acq_shift_search_batch', h, int8(codes), int32(index0), weights, rule, exclude, period, narms)
       codes: one column per (PRN, arm), the arms of a PRN next to each other.  index0 = [] : columns of samplesPerBlock sampled,
       zero-padded replicas; otherwise columns of chips and the ONE 0-based index vector that samples them all (zero padding on the
       device).  rule 0 | 1 | 2 = GC_SHIFT_PICK_GLOBAL | _SEQUENTIAL | _SEQUENTIAL_PAIRS.  picks: 4 x nPRN double
       [row0; codePhase0; peak; secondPeak], row0 = -1: nothing above 0.  [] when the library answers GC_E_UNSUPPORTED (block
       lengths without specialised transforms) or GC_E_NOMEM (the whole list's buffers do not fit): search PRN by PRN with 'acq_shift_search' / 'acq_shift_row' then. */
    int32_t n = 0, nrows = 0, amax = 0;
    if (nrhs < 9) mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search_batch: handle, codes, index, weights, rule, exclude, period, narms");
    if (gc_acq_shift_dims(handle(prhs[1]), &n, &nrows, &amax)) fail("gc_acq_shift_dims");
    const int narms = (int)mxGetScalar(prhs[8]);
    const int ncols = (int)mxGetN(prhs[2]), code_len = (int)mxGetM(prhs[2]);
    const int have_index = !mxIsEmpty(prhs[3]);
    if (!mxIsInt8(prhs[2]) || narms < 1 || narms > amax || ncols < narms || ncols % narms != 0 || (!have_index && code_len != n))
      mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search_batch: codes must be int8, one column per (PRN, arm), %d rows without an index vector", (int)n);
    if (have_index && (!mxIsInt32(prhs[3]) || (int32_t)mxGetNumberOfElements(prhs[3]) > n))
      mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search_batch: the index vector must be int32 with at most %d entries", (int)n);
    const double* w = NULL;
    if (!mxIsEmpty(prhs[4])) {
      if (!mxIsDouble(prhs[4]) || (int)mxGetNumberOfElements(prhs[4]) < narms)
        mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_search_batch: one double weight per code arm");
      w = mxGetDoubles(prhs[4]);
    }
    const int nprn = ncols / narms;
    gc_acq_shift_pick* picks = (gc_acq_shift_pick*)mxCalloc((mwSize)nprn, sizeof(gc_acq_shift_pick));
    const int rc = gc_acq_shift_search_batch(handle(prhs[1]), nprn, narms, (const int8_t*)mxGetData(prhs[2]), code_len,
                                             have_index ? (const int32_t*)mxGetData(prhs[3]) : NULL,
                                             have_index ? (int)mxGetNumberOfElements(prhs[3]) : 0, w, (int)mxGetScalar(prhs[5]),
                                             (int)mxGetScalar(prhs[6]), (int)mxGetScalar(prhs[7]), picks);
    if (rc == GC_E_UNSUPPORTED || rc == GC_E_NOMEM) { /* NOMEM: all PRNs' spectra at once do not fit; one PRN's still may */
      plhs[0] = mxCreateDoubleMatrix(0, 0, mxREAL);
    } else {
      if (rc) fail("gc_acq_shift_search_batch");
      plhs[0] = mxCreateDoubleMatrix(4, (mwSize)nprn, mxREAL);
      double* o = mxGetDoubles(plhs[0]);
      for (int k = 0; k < nprn; ++k) {
        o[4 * k] = picks[k].row;
        o[4 * k + 1] = picks[k].code_phase;
        o[4 * k + 2] = picks[k].peak;
        o[4 * k + 3] = picks[k].second_peak;
      }
    }
    mxFree(picks);
  } else if (!strcmp(cmd, "acq_shift_row")) {
    /* r = gnsscorr_mex('acq_shift_row', h, row0[, n]): one correlation row; its length is the prepared samplesPerBlock */
    int32_t n = 0;
    if (gc_acq_shift_dims(handle(prhs[1]), &n, NULL, NULL)) fail("gc_acq_shift_dims");
    if (nrhs > 3 && (int32_t)mxGetScalar(prhs[3]) != n)
      mexErrMsgIdAndTxt("gnsscorr:args", "acq_shift_row: a row of the prepared search has %d samples, not %d", (int)n, (int)mxGetScalar(prhs[3]));
    plhs[0] = mxCreateNumericMatrix(1, (mwSize)n, mxSINGLE_CLASS, mxREAL);
    if (gc_acq_shift_row(handle(prhs[1]), (int)mxGetScalar(prhs[2]), (float*)mxGetData(plhs[0]))) fail("gc_acq_shift_row");
  } else if (!strcmp(cmd, "read_if")) {
    /* x = gnsscorr_mex('read_if', h, firstSample0, n[, class[, valuesPerSample]]): raw record samples back in the record's own
       class and order ('int8' | 'int16', 2 values per complex sample, 1 per real one): taken from the context (gc_if_format) -
       gc_read_if writes n * bytes-per-sample of the RECORD; a class / count that disagrees with it is an error, not an overflow */
    int dt = 0, lay = 0;
    if (gc_if_format(handle(prhs[1]), &dt, &lay)) fail("gc_if_format");
    const mwSize n = (mwSize)mxGetScalar(prhs[3]);
    const mwSize per = lay == GC_REAL ? 1 : 2;
    if (nrhs > 4) {
      char cls[16] = "";
      mxGetString(prhs[4], cls, sizeof cls);
      if (strcmp(cls, dt == GC_I16 ? "int16" : "int8"))
        mexErrMsgIdAndTxt("gnsscorr:args", "read_if: the record holds %s samples, not %s", dt == GC_I16 ? "int16" : "int8", cls);
    }
    if (nrhs > 5 && (mwSize)mxGetScalar(prhs[5]) != per)
      mexErrMsgIdAndTxt("gnsscorr:args", "read_if: the record holds %d value(s) per sample", (int)per);
    plhs[0] = mxCreateNumericMatrix(1, n * per, dt == GC_I16 ? mxINT16_CLASS : mxINT8_CLASS, mxREAL);
    if (gc_read_if(handle(prhs[1]), (uint64_t)mxGetScalar(prhs[2]), (uint64_t)n, mxGetData(plhs[0]))) fail("gc_read_if");
  } else if (!strcmp(cmd, "if_info")) {
    /* info = gnsscorr_mex('if_info', h): [samples, class (0 int8 / 1 int16), order (0 real / 1 I,Q / 2 Q,I)] of the record the
       context holds; samples = 0: none (never loaded, or a windowed run detached its last window) */
    void* ptr = NULL;
    uint64_t n = 0;
    int dt = 0, lay = 0;
    plhs[0] = mxCreateDoubleMatrix(1, 3, mxREAL);
    if (gc_if_buffer(handle(prhs[1]), &ptr, &n) == GC_OK && ptr != NULL && n > 0 && gc_if_format(handle(prhs[1]), &dt, &lay) == GC_OK) {
      mxGetDoubles(plhs[0])[0] = (double)n;
      mxGetDoubles(plhs[0])[1] = (double)dt;
      mxGetDoubles(plhs[0])[2] = (double)lay;
    }
  } else if (!strcmp(cmd, "acq_guard_stats")) {
    /* [ties, maxDev, eps] = gnsscorr_mex('acq_guard_stats', h): the float64 guard of the context's last search (gc_acq_guard_stats) */
    int32_t ties = 0;
    double dev = 0.0, eps = 0.0;
    if (gc_acq_guard_stats(handle(prhs[1]), &ties, &dev, &eps)) fail("gc_acq_guard_stats");
    plhs[0] = mxCreateDoubleScalar(ties);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(dev);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(eps);
  } else if (!strcmp(cmd, "device_count")) {
    int n = 0;
    if (gc_device_count(&n)) fail("gc_device_count");
    plhs[0] = mxCreateDoubleScalar(n);
  } else if (!strcmp(cmd, "device_info")) {
    char name[128] = "";
    int cus = 0;
    if (gc_device_info(handle(prhs[1]), name, (int)sizeof name, &cus)) fail("gc_device_info");
    plhs[0] = mxCreateString(name);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(cus);
  } else {
    mexErrMsgIdAndTxt("gnsscorr:usage", "unknown command %s", cmd);
  }
}
