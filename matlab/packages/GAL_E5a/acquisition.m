function acqResults = acquisition(longSignal, settings)
%ACQUISITION  Drop-in for this package's include/acquisition.m: same signature, the searches on an MI355X (matlab/gnsscorr_acquisition.m).
acqResults = gnsscorr_acquisition(longSignal, settings, 'GAL_E5a');
end
