function [trackResults, channel] = WB_tracking(fid, channel, settings)
%WB_TRACKING  Drop-in for BDS/B1C/include/WB_tracking.m: same signature, the loop on an MI355X (matlab/gnsscorr_tracking.m).
[trackResults, channel] = gnsscorr_tracking(fid, channel, settings, 'BDS_B1C_WB');
end
