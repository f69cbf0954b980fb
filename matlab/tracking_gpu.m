function [trackResults, channel] = tracking_gpu(h, channel, settings)
%TRACKING_GPU  Drop-in for include/tracking.m (GPS L1 C/A) with the correlator on an MI355X.
%   [trackResults, channel] = tracking_gpu(h, channel, settings)
%   h is a gnsscorr context whose IF record was loaded with
%       gnsscorr_mex('open_if_file', h, settings.fileName, dataAdaptCoeff*settings.skipNumberOfBytes, ...
%                    0, settings.dataType, settings.fileType, settings.samplingFreq)
%   and takes the place of the fid argument of the reference function.  The six sums of
%   tracking.m:247-300 come from gc_track (one kernel launch per epoch for all channels); the
%   discriminators and loop filters run on the host inside libgnsscorr (C++), C/N0 here.
%   Written for this repository; it is not a copy of the reference's tracking.m.
fields = {'absoluteSample','codeFreq','carrFreq','I_E','Q_E','I_P','Q_P','I_L','Q_L', ...
          'dllDiscr','dllDiscrFilt','pllDiscr','pllDiscrFilt','remCodePhase','remCarrPhase'};
nCh = settings.numberOfChannels;
nEp = settings.msToProcess;
blank = struct('status','-','PRN',0);
for k = 1:numel(fields), blank.(fields{k}) = zeros(1, nEp); end
blank.CNo = struct('VSMValue', [], 'VSMIndex', []);
trackResults = repmat(blank, 1, nCh);
active = find([channel.PRN] ~= 0);
if isempty(active), return; end
chanTable = zeros(5, numel(active));
for k = 1:numel(active)
    c = active(k);
    code = generateCAcode(channel(c).PRN);
    gnsscorr_mex('set_channel', h, c-1, {int8([code(end) code code(1)])}, 1);
    chanTable(:,k) = [c-1; channel(c).PRN; channel(c).acquiredFreq; settings.codeFreqBasis; channel(c).codePhase];
    trackResults(c).PRN = channel(c).PRN;
end
[trk, epochs, status] = gnsscorr_mex('track', h, settings, chanTable);
for k = 1:numel(active)
    c = active(k);
    for f = 1:numel(fields)
        trackResults(c).(fields{f}) = trk(:, f, k).';
    end
    vsm = settings.CNo.VSMinterval;
    for e = vsm:vsm:epochs(k)
        trackResults(c).CNo.VSMValue(end+1) = CNoVSM(trackResults(c).I_P(e-vsm+1:e), ...
            trackResults(c).Q_P(e-vsm+1:e), settings.CNo.accTime);
        trackResults(c).CNo.VSMIndex(end+1) = e;
    end
    if epochs(k) == nEp, trackResults(c).status = channel(c).status; end
end
if status ~= 0
    disp('Not able to read the specified number of samples  for tracking, exiting!')
end
end
