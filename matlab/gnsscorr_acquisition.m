function acqResults = gnsscorr_acquisition(longSignal, settings, name)
%GNSSCORR_ACQUISITION  acqResults = acquisition(longSignal, settings) of package NAME with the searches on an MI355X.
%   Same arguments and result as the package's include/acquisition.m (resampling off), so that postProcessing.m:100 is called
%   unchanged.  longSignal is the complex row postProcessing.m:88-96 builds from the file (data1 + 1i*data2): it is
%   re-quantised to the int8 / int16 samples it came from (exact: the values are integers) and uploaded as the record.
%   Covered here: the packages that follow GPS L1 C/A's scheme (SURVEY.md section 8a rows A1-A4) - GPS L1 C/A, GPS L5,
%   Galileo E1 / E5a / E5b, BDS B2a / B3I, GLONASS L1 / L2.  Coarse search (acquisition.m:158-200) and the per-code sums of the
%   fine stage (:206-248) run on the GPU; the package's hypothesis search over those 20-100 complex numbers per bin is below.
%   Written for this repository; not a copy of any reference file.

[h, src, hasRecord] = gnsscorr_upload(longSignal, settings);

%--- optional input conditioning (acquisition.m:46-111): zero-phase FIR(700) band-pass + band-pass-sampling decimation -----
flag = 0;
if isfield(settings, 'resamplingflag'), flag = settings.resamplingflag; end
if isfield(settings, 'resamplingFlag'), flag = settings.resamplingFlag; end      % BDS/B3I spells it so
resampled = settings.samplingFreq > settings.resamplingThreshold && flag == 1;
if resampled && ~hasRecord
    error('gnsscorr:acquisition', 'the conditioning block runs on integer sample values (int8 / int16 records)');
end
if resampled
    switch name                                                       % BW of the package's block (acquisition.m:58 and its twins)
        case {'GPS_L1CA', 'GPS_L5C', 'BDS_B2a', 'BDS_B3I'}, c.bandwidth = settings.codeFreqBasis * 2 + 0.5e6;
        case {'GAL_E5a', 'GAL_E5b'}, c.bandwidth = 20.46e6;
        case 'GAL_E1C', c.bandwidth = 20.552e6;
        case {'GLO_GL1', 'GLO_GL2'}, c.bandwidth = 9e6;                                                            % GLO_GL1 acquisition.m:57
        otherwise, error('gnsscorr:acquisition', 'the resampling front end of %s is not wired', name);
    end
    c.bandMargin = 0.002 * any(strcmp(name, {'GPS_L5C', 'BDS_B2a'}));            % wp = [w1*2/fs-0.002 w2*2/fs+0.002] (GPS_L5C acquisition.m:69)
    mirror = any(strcmp(name, {'GPS_L1CA', 'GPS_L5C', 'BDS_B2a'}));              % the other packages map back as carrFreq - IF only (GAL_E5a :292)
    c.samplingFreq = settings.samplingFreq;  c.IF = settings.IF;
    c.firstSample = 0;  c.nSamples = numel(longSignal);
    oldFreq = settings.samplingFreq;  oldIF = settings.IF;
    [settings.samplingFreq, settings.IF] = gnsscorr_mex('acq_condition', h, c);                                      % :81,95
end

fs = settings.samplingFreq;  ts = 1 / fs;
spc = round(fs / (settings.codeFreqBasis / settings.codeLength));      % acquisition.m:116
a.samplingFreq = fs;  a.codeFreqBasis = settings.codeFreqBasis;  a.codeLength = settings.codeLength;  a.IF = settings.IF;
a.acqSearchBand = settings.acqSearchBand;  a.acqSearchStep = settings.acqSearchStep;  a.acqNonCohTime = settings.acqNonCohTime;
a.firstSample = 0;  a.source = max(double(resampled), src);
NH20 = [1 1 1 1 1 -1 1 1 -1 -1 1 -1 1 -1 1 1 -1 -1 -1 1];              % GPS_L5C acquisition.m:131
CS25 = [1 1 -1 -1 -1 1 1 1 1 1 1 1 -1 1 -1 1 -1 -1 1 -1 -1 1 1 -1 1];  % GAL_E1C acquisition.m:138

f.nResults = 32;  f.boc = false;  f.ncodes = 20;  f.fineStep = 25;  f.indexOffset = 1;  f.search = 'circular';  f.glonass = false;
switch name
    case 'GPS_L1CA'
        f.coarse = @(p) {generateCAcode(p)};  f.search = 'l1ca';
    case 'GPS_L5C'
        f.coarse = @(p) {generateL5Icode(p, settings), generateL5Qcode(p, settings)};
        f.fine = @(p) {generateL5Qcode(p, settings)};  f.secondary = @(p) NH20;                       % :228-252
    case 'GAL_E5a'
        f.nResults = 50;  f.ncodes = 100;  f.fineStep = 5;
        f.coarse = @(p) {generateE5aIcode(p, 1), generateE5aQcode(p, 1)};
        f.fine = @(p) {generateE5aQcode(p, 1)};  f.secondary = @(p) generateE5aQ_secondary(p);
    case 'GAL_E5b'
        f.nResults = 50;  f.search = 'none';                                                            % GAL_E5b acquisition.m:227
        f.coarse = @(p) {generateE5bIcode(p, 1), generateE5bQcode(p, 1)};
    case 'BDS_B2a'
        f.nResults = max(settings.acqSatelliteList);  f.ncodes = max(10, settings.acqNonCohTime);  f.search = 'noncoh';   % BDS/B2a acquisition.m:139,156
        f.coarse = @(p) {generateB2aDataCode(p, settings), generateB2aPilotCode(p, settings)};
        f.fine = f.coarse;
    case 'BDS_B3I'
        f.nResults = 63;  f.indexOffset = 0;  f.search = 'b3i';
        f.coarse = @(p) {generateB3Icode(p)};  f.fine = f.coarse;
    case 'GAL_E1C'
        f.nResults = 50;  f.boc = true;  f.ncodes = 25;  f.fineStep = 10;  f.indexOffset = 0;  f.search = 'split';
        f.coarse = @(p) {generateE1Bcode(p), generateE1Ccode(p)};
        f.fine = @(p) {generateE1Ccode(p)};  f.secondary = @(p) CS25;
    case {'GLO_GL1', 'GLO_GL2'}
        f.nResults = 21;  f.glonass = true;
    otherwise
        error('gnsscorr:acquisition', 'package %s is not served by this wrapper', name);
end
acqResults.carrFreq   = zeros(1, f.nResults);
acqResults.codePhase  = zeros(1, f.nResults);
acqResults.peakMetric = zeros(1, f.nResults);
if f.glonass
    if resampled
        acqResults = glonass(acqResults, h, a, settings, spc, oldFreq, oldIF);
    else
        acqResults = glonass(acqResults, h, a, settings, spc, 0, 0);
    end
    return
end

%--- sampled replicas (makeCaTable.m:59-67; BOC tables makeE1BTable.m:43-55) ----------------------------------------------
tc = 1 / settings.codeFreqBasis;  L = settings.codeLength;
if f.boc, tc = tc / 2;  L = 2 * L; end
idx = ceil(ts * (1:spc) / tc);
idx(end) = L;
if f.boc, idx(1) = 1; end
prns = settings.acqSatelliteList;
first = f.coarse(prns(1));
narms = numel(first);
tables = zeros(spc, narms * numel(prns));
for k = 1:numel(prns)
    c = f.coarse(prns(k));
    for m = 1:narms, tables(:, (k - 1) * narms + m) = c{m}(idx).'; end
end
res = gnsscorr_mex('acquire_coarse_multi', h, a, int8(tables), narms);    % rows: bin, codePhase, peak, peakMetric, coarseFreq

nfine = 0;
if f.fineStep > 0, nfine = round(settings.acqSearchStep / f.fineStep) + 1; end
for k = 1:numel(prns)
    p = prns(k);
    acqResults.peakMetric(p) = res(4, k);                                   % acquisition.m:200
    if res(4, k) <= settings.acqThreshold, continue; end                   % :206
    acqResults.codePhase(p) = res(2, k);                                    % :256
    switch f.search
        case 'none'
            carr = res(5, k);
        case 'l1ca'
            carr = gnsscorr_mex('acquire_fine_l1ca', h, a, int8(generateCAcode(p)), res(2, k), res(5, k));
        otherwise
            q.samplingFreq = fs;  q.codeFreq = 1 / tc;  q.f0 = res(5, k) + settings.acqSearchStep / 2;  q.fstep = f.fineStep;
            q.firstSample = res(2, k) - 1;  q.samplesPerCode = spc;  q.ncodes = f.ncodes;  q.nbins = nfine;
            q.codeLength = L;  q.indexOffset = f.indexOffset;  q.source = a.source;
            codes = f.fine(p);
            sums = cell(1, numel(codes));
            for m = 1:numel(codes)
                s = gnsscorr_mex('fine_sums', h, q, int8(codes{m}));        % (re, im) pairs per code: 2*ncodes x nbins
                sums{m} = s(1:2:end, :) + 1i * s(2:2:end, :);
            end
            power = zeros(1, nfine);
            for b = 1:nfine
                v = sums{1}(:, b).';
                switch f.search
                    case 'circular'                                         % GPS_L5C acquisition.m:243-248
                        power(b) = circular(v, f.secondary(p));
                    case 'split'                                            % GAL_E1C acquisition.m:237-245
                        power(b) = splitsum(v, f.secondary(p));
                    case 'noncoh'                                           % BDS/B2a acquisition.m:259-274
                        power(b) = sum(abs(v)) + sum(abs(sums{2}(:, b)));
                    case 'b3i'                                              % BDS/B3I acquisition.m:252-271
                        if (p >= 1 && p <= 5) || (p >= 59 && p <= 63)
                            p1 = sum(abs(v(1:2:19) + v(2:2:20)));
                            p2 = abs(v(1)) + abs(v(20)) + sum(abs(v(2:2:18) + v(3:2:19)));
                            power(b) = max(p1, p2);
                        else
                            power(b) = splitsum(v, NH20);
                        end
                end
            end
            [~, best] = max(power);
            carr = q.f0 - f.fineStep * (best - 1);
    end
    if carr == 0, carr = 1; end                                             % :258-260
    if resampled                                                            % :264-276: back to the record's rate and IF
        acqResults.codePhase(p) = floor((res(2, k) - 1) / settings.samplingFreq * oldFreq) + 1;
        if mirror && settings.IF >= settings.samplingFreq / 2
            doppler = (settings.samplingFreq - settings.IF) - carr;
        else
            doppler = carr - settings.IF;
        end
        carr = doppler + oldIF;
    end
    acqResults.carrFreq(p) = carr;
end
end

function best = circular(v, sec)
% the secondary code tried at every circular shift, coherent sum
best = 0;
for k = 1:numel(sec)
    best = max(best, abs(sum(v .* sec)));
    sec = circshift(sec, 1);
end
end

function best = splitsum(v, sec)
% the secondary code aligned, then every shift with the sum split at the possible data-bit edge
best = abs(sum(v .* sec));
for k = 1:numel(sec) - 1
    t = v .* circshift(sec, k);
    best = max(best, abs(sum(t(1:k))) + abs(sum(t(k+1:end))));
end
end

function acqResults = glonass(acqResults, h, a, settings, spc, oldFreq, oldIF)
% GLO_GL1/include/acquisition.m:120-200: per frequency number K the L1CA scheme around IF - freqSpacing*K with the common
% 511-chip code; fine stage over 40 codes in 25-Hz bins against the 10-ms meander.  Results at index K + 8.
fs = settings.samplingFreq;  ts = 1 / fs;
table = generateCAcode(0, fs, spc);
code40 = generateCAcode(0, fs, 40 * spc);
nfine = round(settings.acqSearchStep / 25) + 1;
for K = settings.acqSatelliteList
    a.IF = settings.IF - settings.freqSpacing * K;
    r = gnsscorr_mex('acquire_coarse_multi', h, a, int8(table(:)), 1);
    acqResults.peakMetric(K + 8) = r(4);
    if r(4) <= settings.acqThreshold, continue; end
    % the 40-code replica goes to the GPU as it is, one entry per sample (codeFreq = 0); the 40 per-code sums of every bin come back
    q = struct('samplingFreq', fs, 'codeFreq', 0, 'f0', r(5) + settings.acqSearchStep / 2, 'fstep', 25, 'firstSample', r(2) - 1, ...
               'samplesPerCode', spc, 'ncodes', 40, 'nbins', nfine, 'codeLength', 40 * spc, 'indexOffset', 0, 'source', a.source);
    s = gnsscorr_mex('fine_sums', h, q, int8(code40(:)));
    s = s(1:2:end, :) + 1i * s(2:2:end, :);                                 % 40 x nfine
    power = zeros(1, nfine);  freqs = zeros(1, nfine);
    for b = 1:nfine
        freqs(b) = q.f0 - 25 * (b - 1);
        perCode = s(:, b).';
        best = 0;
        for c = 1:20
            best = max(best, abs(sum(perCode(c:c+9)) - sum(perCode(c+10:c+19))));
        end
        power(b) = best;
    end
    [~, best] = max(power);
    acqResults.carrFreq(K + 8) = freqs(best);
    acqResults.codePhase(K + 8) = r(2);
    if acqResults.carrFreq(K + 8) == 0, acqResults.carrFreq(K + 8) = 1; end      % :263-265
    if oldFreq > 0                                                                % :267-285, after the input conditioning
        acqResults.codePhase(K + 8) = floor((r(2) - 1) / fs * oldFreq) + 1;
        if settings.IF >= fs / 2
            doppler = (fs - settings.IF) - acqResults.carrFreq(K + 8);
        else
            doppler = acqResults.carrFreq(K + 8) - settings.IF;
        end
        % the reference assigns the mapped frequency to a field it spells carrFreqcarrFreq (:284) and leaves carrFreq in the
        % resampled band; a drop-in returns what the reference returns
        acqResults.carrFreqcarrFreq(K + 8) = doppler + oldIF;
    end
end
end
