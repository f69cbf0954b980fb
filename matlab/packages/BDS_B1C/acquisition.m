function acqResults = acquisition(longSignal, settings)
%ACQUISITION  Drop-in for this package's include/acquisition.m: same signature, the transforms on an MI355X (matlab/gnsscorr_acquisition_shift.m).
acqResults = gnsscorr_acquisition_shift(longSignal, settings, 'BDS_B1C');
end
