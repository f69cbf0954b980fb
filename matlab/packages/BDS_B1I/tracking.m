function [trackResults, channel] = tracking(fid, channel, settings)
%TRACKING  Drop-in for this package's include/tracking.m: same signature, the loop on an MI355X (matlab/gnsscorr_tracking.m).
[trackResults, channel] = gnsscorr_tracking(fid, channel, settings, 'BDS_B1I');
end
