function [trackResults, channel] = gnsscorr_tracking(fid, channel, settings, name)
%GNSSCORR_TRACKING  [trackResults, channel] = tracking(fid, channel, settings) of package NAME with the correlator on an MI355X.
%   Same arguments and results as the package's include/tracking.m (NB_tracking.m / WB_tracking.m for BDS B1C), so that
%   postProcessing.m:124 is called unchanged: fid is the open IF file (fopen(fid) names it; it is uploaded to HBM once and kept
%   for later calls), channel comes from preRun.m, settings from initSettings.m.  The per-epoch loop of tracking.m:133-368 -
%   block sizing, replica ramps, carrier wipe-off, six (12, 18) sums, discriminators, loop filters - runs inside libgnsscorr
%   (gc_track: correlator kernels on the GPU, tracking.m:302-335 in C++ on the host, as BASELINE.json's north_star wants);
%   this file builds the package's code tables with the package's own generators, fills trackResults exactly as the
%   package creates it, and runs the package's own C/N0 estimator over the recorded prompt sums.
%   Written for this repository; not a copy of any reference file.

pkg = gnsscorr_package(name, settings);
n = pkg.numEpochs;
nCh = settings.numberOfChannels;

%--- result structure as the package's tracking.m creates it (tracking.m:47-86) ---------------------------------------
trackResults.status         = '-';
trackResults.absoluteSample = zeros(1, n);
trackResults.codeFreq       = inf(1, n);
trackResults.carrFreq       = inf(1, n);
trackResults.I_P = zeros(1, n);  trackResults.I_E = zeros(1, n);  trackResults.I_L = zeros(1, n);
trackResults.Q_E = zeros(1, n);  trackResults.Q_P = zeros(1, n);  trackResults.Q_L = zeros(1, n);
pilotNames = {};
if pkg.pilotOn && strcmp(pkg.pilotFields, 'prompt'), pilotNames = {'Pilot_I_P', 'Pilot_Q_P'}; end
if pkg.pilotOn && strcmp(pkg.pilotFields, 'all')
    pilotNames = {'Pilot_I_P', 'Pilot_I_E', 'Pilot_I_L', 'Pilot_Q_E', 'Pilot_Q_P', 'Pilot_Q_L'};
end
for k = 1:numel(pilotNames), trackResults.(pilotNames{k}) = zeros(1, n); end
trackResults.dllDiscr       = inf(1, n);
trackResults.dllDiscrFilt   = inf(1, n);
trackResults.pllDiscr       = inf(1, n);
trackResults.pllDiscrFilt   = inf(1, n);
trackResults.remCodePhase   = inf(1, n);
trackResults.remCarrPhase   = inf(1, n);
if strcmp(pkg.cno, 'VSM')
    trackResults.CNo.VSMValue = zeros(1, floor(n / settings.CNo.VSMinterval));
    trackResults.CNo.VSMIndex = zeros(1, floor(n / settings.CNo.VSMinterval));
else
    nRec = floor(n / settings.CNoInterval);
    trackResults.DataCNo = zeros(1, nRec);  trackResults.DataPLD = zeros(1, nRec);
    if pkg.pilotOn
        combined = 'B1C_CNo';
        if strcmp(name, 'BDS_B2a'), combined = 'B2a_CNo'; end
        trackResults.PilotCNo = zeros(1, nRec);  trackResults.PilotPLD = zeros(1, nRec);  trackResults.(combined) = zeros(1, nRec);
    end
end
trackResults = repmat(trackResults, 1, nCh);

%--- which channels run (tracking.m:136; GLO_GL1 tracking.m:138) ---------------------------------------------------------
active = [];
for c = 1:nCh
    if pkg.byStatus
        on = channel(c).status ~= '-';
    else
        on = channel(c).(pkg.idField) ~= 0;
    end
    if on, active(end+1) = c; end %#ok<AGROW>
end
if isempty(active), return; end

%--- the record: uploaded once per file, from byte 0, so that absoluteSample stays file-relative (ftell, tracking.m:212-216) ---
% A record larger than the device: settings.gnsscorrWindowSamples > 0 tracks it window by window instead (gc_track_file: two
% alternating device buffers of that many samples, the next window read and uploaded while the current one is tracked).
fileName = fopen(fid);
order = 'IQ';
if pkg.qiOrder, order = 'QI'; end
windowed = isfield(settings, 'gnsscorrWindowSamples') && settings.gnsscorrWindowSamples > 0;
% One context per RECORD, not per file name: the key carries what decides the bytes in HBM - sample class, file type, sample
% order and the file's size and date - so a different dataType / fileType or a rewritten file under the same name gets a fresh
% context instead of a stale record, and the resident record is (re)loaded whenever the context does not hold one (a windowed
% call leaves none behind: it attaches and detaches its own device windows).
info = dir(fileName);
key = sprintf('%s|%s|%d|%s|%d|%.6f', fileName, settings.dataType, settings.fileType, order, info(1).bytes, info(1).datenum);
h = gnsscorr_context(key);
if isempty(h)
    h = gnsscorr_context(key, 'new');
end
if ~windowed
    loaded = gnsscorr_mex('if_info', h);           % [samples, class (0 int8 / 1 int16), order (0 real / 1 I,Q / 2 Q,I)]
    if loaded(1) == 0
        gnsscorr_mex('open_if_file', h, fileName, 0, 0, settings.dataType, settings.fileType, settings.samplingFreq, order);
    end
end

%--- start of the first block in samples (tracking.m:145-153) --------------------------------------------------------------
skip = settings.(pkg.skipField);
if strcmp(settings.dataType, 'int16')
    if ~pkg.int16Branch
        error('gnsscorr:int16', '%s: the package''s tracking.m has no int16 branch (its fseek assumes one byte per component)', name);
    end
    if rem(skip, 2) ~= 0, error('gnsscorr:int16', 'int16 record: dataAdaptCoeff*skipNumberOfBytes bytes is not a whole number of samples'); end
    skip = skip / 2;                 % dataAdaptCoeff*(skip + (codePhase-1)*2) bytes of 2-byte components
end
if ~pkg.minusOne, skip = skip + 1; end

%--- loop parameters (tracking.m:94-110; GPS_L5C tracking.m:106-111) ----------------------------------------------------------
p.samplingFreq = settings.samplingFreq;
p.codeFreqBasis = settings.codeFreqBasis;       p.codeLength = settings.codeLength;
p.dllCorrelatorSpacing = settings.dllCorrelatorSpacing;
if pkg.doubledCode                                 % GPS_L2C tracking.m:107-109,171: everything in units of the RZ-doubled code
    p.codeFreqBasis = 2 * settings.codeFreqBasis;  p.codeLength = 2 * settings.codeLength;
    p.dllCorrelatorSpacing = 2 * settings.dllCorrelatorSpacing;
end
p.intTime = settings.intTime;
p.dllNoiseBandwidth = settings.dllNoiseBandwidth;  p.dllDampingRatio = settings.dllDampingRatio;
p.pllNoiseBandwidth = settings.pllNoiseBandwidth;  p.pllDampingRatio = settings.pllDampingRatio;
p.pllKind = pkg.pllKind;
if pkg.pllKind == 1
    [p.pf3, p.pf2, p.pf1] = calcLoopCoefCarr(settings);      % the package's own Common/calcLoopCoefCarr.m
end
p.pilotCombine = pkg.pilotMode;
if ~isempty(pkg.pllWeight), p.pllWeight = pkg.pllWeight; end
if ~isempty(pkg.dllWeight), p.dllWeight = pkg.dllWeight; end
if pkg.dllScale ~= 0, p.dllScale = pkg.dllScale; end
p.tablePhaseCount = pkg.phaseCount;
p.skipSamples = skip;
p.numEpochs = n;
% C/N0 and the lock detector come back with the records (4th output): CNoVSM every CNo.VSMinterval epochs inside the loops
% (tracking.m:351-358), Calc_CNo_PLD every CNoInterval epochs with the bookkeeping of BDS/B2a tracking.m:409-432
if strcmp(pkg.cno, 'VSM')
    p.cnoInterval = settings.CNo.VSMinterval;  p.cnoAccTime = settings.CNo.accTime;  p.cnoMode = 0;
else
    p.cnoInterval = settings.CNoInterval;  p.cnoAccTime = settings.intTime;
    p.cnoMode = 1;                                              % data arm only
    if pkg.pilotOn, p.cnoMode = 2; end                          % Calc_CNo_PLD.m:72-75: pilot prompt pair read as (Q, I)
    if pkg.pilotOn && pkg.pilotMode == 4, p.cnoMode = 3; end    % B1C wide-band loop (pilotTRKflag == 2): as (I, Q)
end

%--- code tables and the channel table (tracking.m:156-170) ----------------------------------------------------------------------
chanTable = zeros(6, numel(active));
for k = 1:numel(active)
    c = active(k);
    id = channel(c).(pkg.idField);
    trackResults(c).PRN = id;                                   % tracking.m:138
    tables = pkg.tables(id, settings);
    for a = 1:numel(tables), tables{a} = int8(tables{a}); end
    gnsscorr_mex('set_channel', h, c - 1, tables, pkg.indexScale, pkg.armMult, pkg.windows);
    codeFreq = p.codeFreqBasis;
    if pkg.codeFreqFromChannel, codeFreq = channel(c).codeFreq; end
    clPhase = 0;
    if pkg.phaseCount > 0, clPhase = channel(c).CLCodePhase; end
    chanTable(:, k) = [c - 1; id; channel(c).acquiredFreq; codeFreq; channel(c).codePhase; clPhase];
end

%--- the loops -----------------------------------------------------------------------------------------------------------------------
if windowed
    [trk, epochs, status, cno] = gnsscorr_mex('track_file', h, p, chanTable, fileName, settings.gnsscorrWindowSamples, settings.dataType, ...
                                              settings.fileType, order);
else
    status = -6;
    if isfield(settings, 'gnsscorrDeviceLoop') && settings.gnsscorrDeviceLoop
        % tracking.m:302-335 closed on the GPU as well, one persistent launch for all epochs (gc_track_device); -6 = a
        % configuration it does not cover (three-arm channels whose third arm is not derivable): the host-closed loop then
        [trk, epochs, status, cno] = gnsscorr_mex('track_device', h, p, chanTable);
    end
    if status == -6
        [trk, epochs, status, cno] = gnsscorr_mex('track', h, p, chanTable);  % trk(epoch, (k-1)*21 + field), fields as gc_track_field
    end
end

%--- records (tracking.m:212-216,249,277,314,332,338-348) ------------------------------------------------------------------------
names = {'absoluteSample','codeFreq','carrFreq','I_E','Q_E','I_P','Q_P','I_L','Q_L','dllDiscr','dllDiscrFilt','pllDiscr','pllDiscrFilt', ...
         'remCodePhase','remCarrPhase','Pilot_I_E','Pilot_Q_E','Pilot_I_P','Pilot_Q_P','Pilot_I_L','Pilot_Q_L'};
for k = 1:numel(active)
    c = active(k);
    m = epochs(k);
    for f = 1:numel(names)
        if f > 15 && ~any(strcmp(names{f}, pilotNames)), continue; end
        trackResults(c).(names{f})(1:m) = trk(1:m, (k - 1) * numel(names) + f).';
    end
    if pkg.doubledCode && m > 0
        % GPS_L2C tracking.m:223,250,376,382-383: records are in single-code units; absoluteSample is pushed back by the
        % code-phase remainder expressed in samples
        step = trackResults(c).codeFreq(1:m) / settings.samplingFreq;
        trackResults(c).absoluteSample(1:m) = trackResults(c).absoluteSample(1:m) + 1 - trackResults(c).remCodePhase(1:m) ./ step;
        trackResults(c).remCodePhase(1:m) = trackResults(c).remCodePhase(1:m) / 2;
        trackResults(c).codeFreq(1:m)     = trackResults(c).codeFreq(1:m) / 2;
        trackResults(c).dllDiscr(1:m)     = trackResults(c).dllDiscr(1:m) / 2;
        trackResults(c).dllDiscrFilt(1:m) = trackResults(c).dllDiscrFilt(1:m) / 2;
    end
    if strcmp(pkg.cno, 'VSM')                                    % tracking.m:351-358
        vsm = settings.CNo.VSMinterval;
        cnt = 0;
        for e = vsm:vsm:m
            cnt = cnt + 1;
            if isempty(cno)                                      % VSMinterval = 1: nothing to average inside the loop
                trackResults(c).CNo.VSMValue(cnt) = CNoVSM(trackResults(c).I_P(e-vsm+1:e), trackResults(c).Q_P(e-vsm+1:e), settings.CNo.accTime);
            else
                trackResults(c).CNo.VSMValue(cnt) = cno(cnt, k);
            end
            trackResults(c).CNo.VSMIndex(cnt) = e;
        end
    else                                                         % BDS/B2a tracking.m:409-432
        iv = settings.CNoInterval;
        for e = iv:iv:m
            cnt = e / iv;
            trackResults(c).DataCNo(cnt) = cno(5 * (cnt - 1) + 1, k);
            trackResults(c).DataPLD(cnt) = cno(5 * (cnt - 1) + 4, k);
            if pkg.pilotOn
                trackResults(c).PilotCNo(cnt) = cno(5 * (cnt - 1) + 2, k);
                trackResults(c).(combined)(cnt) = cno(5 * (cnt - 1) + 3, k);
                trackResults(c).PilotPLD(cnt) = cno(5 * (cnt - 1) + 5, k);
            end
        end
    end
    if m == n, trackResults(c).status = channel(c).status; end   % tracking.m:365
end
if status == -2                                                 % GC_E_RANGE: the reference's short read (tracking.m:241-245)
    disp('Not able to read the specified number of samples  for tracking, exiting!')
end
end
