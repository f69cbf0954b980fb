function [trackResults, channel] = NB_tracking(fid, channel, settings)
%NB_TRACKING  Drop-in for BDS/B1C/include/NB_tracking.m: same signature, the loop on an MI355X (matlab/gnsscorr_tracking.m).
[trackResults, channel] = gnsscorr_tracking(fid, channel, settings, 'BDS_B1C_NB');
end
