function acqResults = acquisition_gpu(h, settings)
%ACQUISITION_GPU  Drop-in for include/acquisition.m (GPS L1 C/A, resampling off) on an MI355X.
%   acqResults = acquisition_gpu(h, settings)
%   The record plays the role of longSignal: it starts at settings.skipNumberOfBytes samples into
%   the IF buffer of context h (postProcessing.m:74-96).  Written for this repository.
acqResults.carrFreq   = zeros(1, 32);
acqResults.codePhase  = zeros(1, 32);
acqResults.peakMetric = zeros(1, 32);
a = settings;  a.firstSample = settings.skipNumberOfBytes;
prns = settings.acqSatelliteList;
spc = round(settings.samplingFreq / (settings.codeFreqBasis / settings.codeLength));
codes = zeros(spc, numel(prns), 'int8');
for k = 1:numel(prns), codes(:,k) = int8(makeCaTable(prns(k), settings)); end
res = gnsscorr_mex('acquire_coarse', h, a, codes);      % rows: bin, codePhase, peak, metric, freq
for k = 1:numel(prns)
    p = prns(k);
    acqResults.peakMetric(p) = res(4,k);
    if res(4,k) > settings.acqThreshold
        acqResults.carrFreq(p)  = gnsscorr_mex('acquire_fine_l1ca', h, a, int8(generateCAcode(p)), res(2,k), res(5,k));
        acqResults.codePhase(p) = res(2,k);
    end
end
end
