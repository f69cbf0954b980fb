function acqResults = gnsscorr_acquisition_shift(longSignal, settings, name)
%GNSSCORR_ACQUISITION_SHIFT  acqResults = acquisition(longSignal, settings) of the packages whose Doppler bins are circular
%   shifts of ONE signal spectrum - BDS B1I, GPS L2C, BDS B1C (SURVEY.md section 8a row A5) - with the transforms on an MI355X.
%   Same arguments and result as the package's include/acquisition.m (resampling off), so postProcessing.m:100 is called
%   unchanged.  The GPU mixes the signal block(s) with the package's carriers and transforms them once ('acq_shift_prepare'),
%   then forms every bin of a PRN as a shifted product with the code spectrum inside the inverse transform and returns each
%   result row's maximum.  A package's whole PRN list is ONE call ('acq_shift_search_batch': chip tables and the one index vector
%   that samples them go up, the row selection rule, the first maximum of the winning row and the second peak come back as four
%   numbers per PRN); thresholds, the CL segment search and B1C's 25-Hz fine search stay here.  Block lengths the transforms have
%   no specialised passes for (and settings.gnsscorrPerPRN) search PRN by PRN: 'acq_shift_search' returns the row maxima,
%   'acq_shift_row' the one winning row, and the selection rules run here.
%   Written for this repository; not a copy of any reference file.

[h, src, hasRecord] = gnsscorr_upload(longSignal, settings);
settings.gnsscorrSource = src;              % 0: the searches read the int8 record; 1: its float copy / longSignal itself
nLong = numel(longSignal);
switch name
    case 'BDS_B1I', acqResults = b1i(h, settings);
    case 'GPS_L2C', acqResults = l2c(h, settings);
    case 'BDS_B1C'
        if settings.samplingFreq > settings.resamplingThreshold && settings.resamplingflag == 1
            if ~hasRecord, error('gnsscorr:acquisition', 'the conditioning block runs on integer sample values (int8 / int16 records)'); end
            acqResults = b1cConditioned(h, settings, nLong);
        else
            acqResults = b1c(h, settings, nLong);
        end
    otherwise, error('gnsscorr:acquisition', 'package %s is not served by this wrapper', name);
end
end

%---------------------------------------------------------------------------------------------------------------------------
function t = sampled(code, k, ts, tc, last, firstOne)
% code(ceil(ts*k/tc)) with the end points the packages' table makers force
idx = ceil(ts * k / tc);
if firstOne, idx(1) = 1; end
if last > 0, idx(end) = last; end
t = code(idx);
end

function idx = sampleIndex(k, ts, tc, last, firstOne)
% the index vector of sampled(): it depends on the rates only, one vector serves every PRN's code
idx = ceil(ts * k / tc);
if firstOne, idx(1) = 1; end
if last > 0, idx(end) = last; end
end

function picks = batched(h, chips, idx, w, rule, exclude, period, narms)
% the whole PRN list in one call: chips (one column per PRN and arm) and the one index vector go up, the row selection, the
% first maximum of the winning row and the second peak happen in the library; [] = search PRN by PRN (no specialised transform
% for this block length, or settings.gnsscorrPerPRN set)
picks = gnsscorr_mex('acq_shift_search_batch', h, int8(chips), int32(idx - 1), w, rule, exclude, period, narms);
end

function s = secondPeak(corr, codePhase, exclude, period)
% largest value of one code period outside +-exclude samples of the peak; the three range cases of
% BDS/B1I acquisition.m:141-156 and GPS_L2C acquisition.m:77-91
e1 = codePhase - exclude;  e2 = codePhase + exclude;
if e1 < 2
    rng = e2:(period + e1);
elseif e2 >= period
    rng = (e2 - period + 1):e1;
else
    rng = [1:e1, e2:period];
end
s = max(corr(rng));
end

function [best, freqShift, binIdx] = sequentialBest(rmax, nShifts, nSignals, nBins)
% The packages walk shift by shift and bin by bin and keep a running maximum (BDS/B1I acquisition.m:87-122, GPS_L2C :46-66);
% the last bin is skipped for every shift but the first.  rmax: the row maxima in the library's order
% ((shift * nSignals + signal) * nBins + bin); best = [shift signal bin] (1-based) or [].
prevMax = 0;  best = [];  freqShift = 0;  binIdx = 0;
for it = 1:nShifts
    for b = 1:nBins
        if b == nBins && it > 1, continue; end
        if nSignals == 2
            p1 = rmax((it - 1) * 2 * nBins + b);  p2 = rmax(((it - 1) * 2 + 1) * nBins + b);
            if p1 > prevMax || p2 > prevMax
                if p1 > p2
                    prevMax = p1;  best = [it 1 b];
                else
                    prevMax = p2;  best = [it 2 b];
                end
                freqShift = it;  binIdx = b;
            end
        elseif rmax((it - 1) * nBins + b) > prevMax
            prevMax = rmax((it - 1) * nBins + b);  best = [it 1 b];
            freqShift = it;  binIdx = b;
        end
    end
end
end

%---------------------------------------------------------------------------------------------------------------------------
function acqResults = b1i(h, settings)
% BDS/B1I/include/acquisition.m: two 4-ms signal blocks, Nshifts interleaved carriers, the 2-code replica zero-padded
Ncodes = 2;  Nblocks = 4;                                                                  % :34-35
fs = settings.samplingFreq;  ts = 1 / fs;
spb = round(fs / (settings.codeFreqBasis / (Nblocks * settings.codeLength)));               % :36-37
freqRes = fs / spb;                                                                         % :43
nBins = round(settings.acqSearchBand * 1e3 / freqRes) + 1;                                  % :44
if ~isfield(settings, 'stepSize') || isempty(settings.stepSize)                             % :46-59
    stepSize = 0.5 / (Nblocks * settings.codeLength / settings.codeFreqBasis);
elseif settings.stepSize == freqRes
    stepSize = settings.stepSize;
else
    cand = 1:0.25:freqRes / 2;
    cand = cand(rem(freqRes, cand) == 0);
    d = cand - settings.stepSize;
    [~, m] = min(abs(d));
    if d(m) > 0, stepSize = cand(m - 1); else, stepSize = cand(m); end
end
nShifts = freqRes / stepSize;                                                               % :61
spc2 = round(fs / (settings.codeFreqBasis / (Ncodes * settings.codeLength)));               % makeCaTableDMA.m
initFreq = settings.IF + (settings.acqSearchBand / 2) * 1000;                               % :66
q.samplingFreq = fs;  q.carrierF0 = initFreq;  q.carrierStep = freqRes / nShifts;  q.firstSample = 0;
q.samplesPerBlock = spb;  q.nSignals = 2;  q.nCarriers = nShifts;  q.nBins = nBins;  q.nArmsMax = 1;  q.source = settings.gnsscorrSource;
nRows = gnsscorr_mex('acq_shift_prepare', h, q);
acqResults.carrFreq = zeros(1, 58);  acqResults.codePhase = zeros(1, 58);  acqResults.peakMetric = zeros(1, 58);
chip = round(fs / settings.codeFreqBasis);                                                  % :139
prns = settings.acqSatelliteList;
picks = [];
if ~isfield(settings, 'gnsscorrPerPRN')
    chips = zeros(2 * 2046, numel(prns));
    for k = 1:numel(prns)
        ca = generateCAcode53(prns(k));
        chips(:, k) = [ca ca].';
    end
    picks = batched(h, chips, sampleIndex(1:spc2, ts, 1 / settings.codeFreqBasis, Ncodes * 2046, false), [], 2, chip, spb / Nblocks, 1);
end
for k = 1:numel(prns)
    PRN = prns(k);
    if ~isempty(picks)
        % :87-122 (which (carrier, block, bin) wins), :126 and :141-156 inside the call
        if picks(1, k) < 0, continue; end
        row = picks(1, k);  codePhase = picks(2, k) + 1;  maxPeak = picks(3, k);  second = picks(4, k);
        freqShift = floor(row / (2 * nBins)) + 1;  binIdx = rem(row, nBins) + 1;
    else
        ca = generateCAcode53(PRN);
        table = sampled([ca ca], 1:spc2, ts, 1 / settings.codeFreqBasis, Ncodes * 2046, false);
        local = [table, zeros(1, spb / Ncodes)];                                            % :86
        rmax = double(gnsscorr_mex('acq_shift_search', h, int8(local(:)), [], nRows));
        [best, freqShift, binIdx] = sequentialBest(rmax, nShifts, 2, nBins);
        if isempty(best), continue; end
        corr = double(gnsscorr_mex('acq_shift_row', h, ((best(1) - 1) * 2 + best(2) - 1) * nBins + best(3) - 1, spb));
        [maxPeak, codePhase] = max(corr);                                                   % :126
        second = secondPeak(corr, codePhase, chip, spb / Nblocks);
    end
    acqResults.peakMetric(PRN) = maxPeak / second;                                          % :160
    if maxPeak / second > settings.acqThreshold                                             % :163
        acqResults.codePhase(PRN) = codePhase;
        acqResults.carrFreq(PRN) = initFreq - freqRes * (binIdx - 1) + (freqRes / nShifts) * (freqShift - 1);   % :168
    end
end
end

%---------------------------------------------------------------------------------------------------------------------------
function acqResults = l2c(h, settings)
% GPS/GPS_L2C/include/acquisition.m: CM search over a 40-ms block, then (pilot on) which of the 75 CL segments it lies in
Nblocks = 2;                                                                                % :13
fs = settings.samplingFreq;  ts = 1 / fs;
spc = round(fs / (settings.codeFreqBasis / settings.codeLength));                           % :14-15
chip = round(fs / settings.codeFreqBasis);                                                  % :16
spb = spc * Nblocks;                                                                        % :17
freqRes = fs / spb;                                                                         % :22
nBins = round(settings.acqSearchBand * 1e3 / freqRes) + 1;                                  % :23
nShifts = freqRes / settings.acqStep;                                                       % :25
initFreq = settings.IF + (settings.acqSearchBand / 2) * 1000;                               % :33
q.samplingFreq = fs;  q.carrierF0 = initFreq;  q.carrierStep = -(freqRes / nShifts);  q.firstSample = 0;
q.samplesPerBlock = spb;  q.nSignals = 1;  q.nCarriers = nShifts;  q.nBins = nBins;  q.nArmsMax = 1;  q.source = settings.gnsscorrSource;
nRows = gnsscorr_mex('acq_shift_prepare', h, q);
acqResults.carrFreq = zeros(1, 32);  acqResults.codePhase = zeros(1, 32);  acqResults.peakMetric = zeros(1, 32);
tc = 1 / (settings.codeFreqBasis * 2);
prns = settings.acqSatelliteList;
picks = [];
if ~isfield(settings, 'gnsscorrPerPRN')
    chips = zeros(settings.codeLength * 2, numel(prns));
    for k = 1:numel(prns)
        cm = generateCMcode(prns(k), settings);
        chips(:, k) = cm(:);
    end
    picks = batched(h, chips, sampleIndex(0:spc - 1, ts, tc, settings.codeLength * 2, true), [], 1, chip, spb / Nblocks, 1);
end
for k = 1:numel(prns)
    PRN = prns(k);
    if ~isempty(picks)
        % :46-66 (which (carrier, bin) wins), :72 and :77-91 inside the call
        if picks(1, k) < 0, continue; end
        row = picks(1, k);  codePhase = picks(2, k) + 1;  maxPeak = picks(3, k);  second = picks(4, k);
        freqShift = floor(row / nBins) + 1;  binIdx = rem(row, nBins) + 1;
    else
        cm = generateCMcode(PRN, settings);
        table = sampled(cm, 0:spc - 1, ts, tc, settings.codeLength * 2, true);              % makeCMTable.m
        local = [table, zeros(1, spc)];                                                     % :44
        rmax = double(gnsscorr_mex('acq_shift_search', h, int8(local(:)), [], nRows));
        [best, freqShift, binIdx] = sequentialBest(rmax, nShifts, 1, nBins);
        if isempty(best), continue; end
        corr = double(gnsscorr_mex('acq_shift_row', h, (best(1) - 1) * nBins + best(3) - 1, spb));
        [maxPeak, codePhase] = max(corr);                                                   % :72
        second = secondPeak(corr, codePhase, chip, spb / Nblocks);
    end
    acqResults.peakMetric(PRN) = maxPeak / second;                                          % :94
    if maxPeak / second > settings.acqThreshold                                             % :97
        carr = initFreq - freqRes * (binIdx - 1) - (freqRes / nShifts) * (freqShift - 1);   % :101
        acqResults.carrFreq(PRN) = carr;
        acqResults.codePhase(PRN) = codePhase;
        if settings.pilotTRKflag == 1                                                       % :140-166: 75 one-period correlations
            % sig - mean(sig) wiped with the carrier and each of the 75 CL segments sampled like the CM table: the segments go to
            % the GPU as 75 replicas of one entry per sample (codeFreq = 0), the mean as the dc term
            st = gnsscorr_mex('signal_stats', h, codePhase - 1, spc, settings.gnsscorrSource);
            cl = generateCLcode(PRN, settings);
            idx = ceil(ts * (0:spc - 1) / tc);
            idx(1) = 1;
            if settings.acqCohT <= 10, idx(end) = settings.codeLength; else, idx(end) = settings.codeLength * 2; end
            windows = zeros(spc, 75);
            for ind = 1:75
                windows(:, ind) = cl(idx + settings.codeLength * 2 * (ind - 1)).';
            end
            q2 = struct('samplingFreq', fs, 'codeFreq', 0, 'f0', carr, 'fstep', 0, 'firstSample', codePhase - 1, 'samplesPerCode', spc, ...
                        'ncodes', 1, 'nbins', 1, 'codeLength', spc, 'indexOffset', 0, 'dcRe', st(1), 'dcIm', st(2), 'source', settings.gnsscorrSource);
            s = gnsscorr_mex('fine_sums', h, q2, int8(windows));                            % 2 x 75
            power = abs(s(1, :) + 1i * s(2, :));
            [~, seg] = max(power);
            acqResults.CLCodePhase(PRN) = seg;                                              % :165 (the field grows to the highest PRN found)
        end
    end
end
end

%---------------------------------------------------------------------------------------------------------------------------
function acqResults = b1cConditioned(h, settings, nLong)
% BDS/B1C/include/acquisition.m with settings.resamplingflag: the conditioning block (:50-122, BW = 9 MHz, band edges widened by
% 0.002) runs on the GPU and every length follows the new rate, so the (10 + acqCohT)-ms transform is no longer a size the
% transforms take.  The search runs carrier by carrier instead ('acquire_coarse_multi' with blockLen / codeSamples / nBins /
% armWeight): circshift(IQfreqDom, k) is the carrier moved by k*fs/N, and the N-point circular correlation with the replica's
% samplesXmsLen samples is the linear one of the block followed by a repeat of its first samplesXmsLen samples.
c.samplingFreq = settings.samplingFreq;  c.IF = settings.IF;  c.bandwidth = 9e6;  c.bandMargin = 0.002;     % :61-67
c.firstSample = 0;  c.nSamples = nLong;
oldFreq = settings.samplingFreq;  oldIF = settings.IF;
[settings.samplingFreq, settings.IF, nCond] = gnsscorr_mex('acq_condition', h, c);
fs = settings.samplingFreq;  ts = 1 / fs;
spc = round(fs / (settings.codeFreqBasis / settings.codeLength));
xLen = round(spc / 10 * settings.acqCohT);
n = round(spc / 10 * (10 + settings.acqCohT));
nBins = round(settings.acqSearchBand * 2 / settings.acqStep) + 1;
pilot = settings.pilotACQflag == 1;
fineStep = 25;  nFine = round(settings.acqStep / 25) * 2 + 1;
initFreq = settings.IF + settings.acqSearchBand;
a = struct('samplingFreq', fs, 'codeFreqBasis', settings.codeFreqBasis, 'codeLength', settings.codeLength, 'IF', settings.IF, ...
           'acqSearchBand', settings.acqSearchBand, 'acqSearchStep', fs / n, 'acqNonCohTime', 1, 'firstSample', 0, 'source', 1, ...
           'blockLen', n, 'codeSamples', xLen, 'nBins', nBins);
narms = 1;
if pilot, a.armWeight = [sqrt(11) / sqrt(40), sqrt(29) / sqrt(40)];  narms = 2; end            % :186-187
nMax = max(settings.acqSatelliteList);
acqResults.carrFreq = zeros(1, nMax);  acqResults.codePhase = zeros(1, nMax);  acqResults.peakMetric = zeros(1, nMax);
tc = 1 / settings.codeFreqBasis / 2;
for PRN = settings.acqSatelliteList
    dtab = sampled(generateDataBOC11(settings, PRN), 1:spc, ts, tc, settings.codeLength * 2, true);
    tabs = dtab(1:xLen).';
    if pilot
        ptab = sampled(generatePilotBOC11(settings, PRN), 1:spc, ts, tc, settings.codeLength * 2, true);
        tabs = [tabs, ptab(1:xLen).'];
    end
    r = gnsscorr_mex('acquire_coarse_multi', h, a, int8(tabs), narms);                      % rows: bin, codePhase, peak, metric, freq
    selFreq = initFreq - (r(1) - 1) * settings.acqStep;                                     % :194
    codePhase = r(2);
    acqResults.peakMetric(PRN) = r(4);                                                      % :199
    if codePhase + spc - 1 > nCond, codePhase = codePhase - spc; end                        % :232-234
    if acqResults.peakMetric(PRN) > settings.acqThreshold
        full = dtab(:);
        if pilot, full = [dtab(:), ptab(:)]; end
        q3 = struct('samplingFreq', fs, 'codeFreq', 0, 'f0', selFreq + settings.acqStep, 'fstep', fineStep, 'firstSample', codePhase - 1, ...
                    'samplesPerCode', spc, 'ncodes', 1, 'nbins', nFine, 'codeLength', spc, 'indexOffset', 0, 'source', 1);
        s = gnsscorr_mex('fine_sums', h, q3, int8(full));
        s = abs(s(1, :) + 1i * s(2, :));
        fine = s(1:nFine);
        if pilot, fine = (s(1:nFine) * 11 + s(nFine + 1:2 * nFine) * 29) / 40; end
        [~, m] = max(fine);
        carr = q3.f0 - fineStep * (m - 1);
        if carr == 0, carr = 1; end                                                         % :253-255
        acqResults.codePhase(PRN) = floor((codePhase - 1) / fs * oldFreq) + 1;              % :280-284
        if settings.IF >= fs / 2
            doppler = (fs - settings.IF) - carr;
        else
            doppler = carr - settings.IF;
        end
        acqResults.carrFreq(PRN) = doppler + oldIF;                                         % :296
    end
end
end

%---------------------------------------------------------------------------------------------------------------------------
function acqResults = b1c(h, settings, nLong)
% BDS/B1C/include/acquisition.m: one (10 + acqCohT)-ms spectrum, bins as shifts, data and pilot BOC(1,1) replicas combined
% sqrt(11):sqrt(29), metric peak/sigPower, 25-Hz fine search over one code period
fs = settings.samplingFreq;  ts = 1 / fs;
spc = round(fs / (settings.codeFreqBasis / settings.codeLength));                           % :108-109
xLen = round(spc / 10 * settings.acqCohT);                                                  % :111
n = round(spc / 10 * (10 + settings.acqCohT));                                              % :113
nBins = round(settings.acqSearchBand * 2 / settings.acqStep) + 1;                           % :120
pilot = settings.pilotACQflag == 1;
fineStep = 25;  nFine = round(settings.acqStep / 25) * 2 + 1;                                % :129-130
st = gnsscorr_mex('signal_stats', h, 0, xLen, settings.gnsscorrSource);
sigPower = sqrt(st(3) * xLen);                                                              % :138
initFreq = settings.IF + settings.acqSearchBand;                                            % :141
q.samplingFreq = fs;  q.carrierF0 = initFreq;  q.carrierStep = 0;  q.firstSample = 0;
q.samplesPerBlock = n;  q.nSignals = 1;  q.nCarriers = 1;  q.nBins = nBins;  q.nArmsMax = 2;  q.source = settings.gnsscorrSource;
nRows = gnsscorr_mex('acq_shift_prepare', h, q);
nMax = max(settings.acqSatelliteList);
acqResults.carrFreq = zeros(1, nMax);  acqResults.codePhase = zeros(1, nMax);  acqResults.peakMetric = zeros(1, nMax);
tc = 1 / settings.codeFreqBasis / 2;
prns = settings.acqSatelliteList;
w = [];
narms = 1;
if pilot
    w = [sqrt(11) / sqrt(40), sqrt(29) / sqrt(40)];                                         % :186-187
    narms = 2;
end
picks = [];
if ~isfield(settings, 'gnsscorrPerPRN')
    chips = zeros(settings.codeLength * 2, narms * numel(prns));
    for k = 1:numel(prns)
        d = generateDataBOC11(settings, prns(k));
        chips(:, (k - 1) * narms + 1) = d(:);
        if pilot
            pl = generatePilotBOC11(settings, prns(k));
            chips(:, (k - 1) * narms + 2) = pl(:);
        end
    end
    idx = sampleIndex(1:spc, ts, tc, settings.codeLength * 2, true);
    picks = batched(h, chips, idx(1:xLen), w, 0, 0, 1, narms);                              % [table(1:samplesXmsLen) zeros], :155-156
end
for k = 1:numel(prns)
    PRN = prns(k);
    dtab = [];  ptab = [];                   % the spc-long sampled tables: only the per-PRN search and the fine stage of a detection need them
    if ~isempty(picks)
        % :193 max(max(results, [], 2)) and [~, codePhase] = max(max(results)) inside the call
        binIdx = picks(1, k) + 1;  peak = picks(3, k);  codePhase = picks(2, k) + 1;
    else
        dtab = sampled(generateDataBOC11(settings, PRN), 1:spc, ts, tc, settings.codeLength * 2, true);   % makeDataTable.m
        if pilot, ptab = sampled(generatePilotBOC11(settings, PRN), 1:spc, ts, tc, settings.codeLength * 2, true); end
        arms = [dtab(1:xLen), zeros(1, n - xLen)].';                                        % :155-156
        if pilot, arms = [arms, [ptab(1:xLen), zeros(1, n - xLen)].']; end
        [rmax, rarg] = gnsscorr_mex('acq_shift_search', h, int8(arms), w, nRows);
        rmax = double(rmax);  rarg = double(rarg);
        [peak, binIdx] = max(rmax);                                                         % :193 max(max(results, [], 2))
        codePhase = min(rarg(rmax == peak)) + 1;                                            % [~, codePhase] = max(max(results))
    end
    selFreq = initFreq - (binIdx - 1) * settings.acqStep;                                   % :194
    acqResults.peakMetric(PRN) = peak / sigPower;                                           % :199
    if codePhase + spc - 1 > nLong, codePhase = codePhase - spc; end                        % :232-234
    if acqResults.peakMetric(PRN) > settings.acqThreshold
        % one code period against the sampled BOC tables at nFine carriers (:242-250): the tables go to the GPU as replicas of one
        % entry per sample (codeFreq = 0), data and pilot in one call
        if isempty(dtab)
            dtab = sampled(generateDataBOC11(settings, PRN), 1:spc, ts, tc, settings.codeLength * 2, true);   % makeDataTable.m
            if pilot, ptab = sampled(generatePilotBOC11(settings, PRN), 1:spc, ts, tc, settings.codeLength * 2, true); end
        end
        tabs = dtab(:);
        if pilot, tabs = [dtab(:), ptab(:)]; end
        q3 = struct('samplingFreq', fs, 'codeFreq', 0, 'f0', selFreq + settings.acqStep, 'fstep', fineStep, 'firstSample', codePhase - 1, ...
                    'samplesPerCode', spc, 'ncodes', 1, 'nbins', nFine, 'codeLength', spc, 'indexOffset', 0, 'source', settings.gnsscorrSource);
        s = gnsscorr_mex('fine_sums', h, q3, int8(tabs));                                   % 2 x (nFine * narms)
        s = abs(s(1, :) + 1i * s(2, :));
        fine = s(1:nFine);
        if pilot, fine = (s(1:nFine) * 11 + s(nFine + 1:2 * nFine) * 29) / 40; end
        [~, m] = max(fine);
        carr = q3.f0 - fineStep * (m - 1);
        if carr == 0, carr = 1; end                                                         % :253-255
        acqResults.carrFreq(PRN) = carr;
        acqResults.codePhase(PRN) = codePhase;
    end
end
end
