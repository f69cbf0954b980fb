function [h, src, hasRecord] = gnsscorr_upload(longSignal, settings)
%GNSSCORR_UPLOAD  longSignal of acquisition(longSignal, settings) onto the GPU (matlab/gnsscorr_acquisition*.m).
%   postProcessing.m:88-96 builds longSignal as data1 + 1i*data2 from the file, so its values are the file's integers: int8
%   values are uploaded as the record the searches read in place (src = 0); int16 values (settings.dataType = 'int16') as an int16
%   record whose float copy the searches read (src = 1, 'acq_from_record'); any other complex row goes up as it is in single
%   precision (src = 1, no record: hasRecord = false - the conditioning block needs one).
%   Written for this repository; not a copy of any reference file.
x = [real(longSignal); imag(longSignal)];
x = x(:).';
h = gnsscorr_context('longSignal', 'new');
integers = ~any(x ~= round(x));
hasRecord = integers && max(abs(x)) <= 32767;
if hasRecord && max(abs(x)) <= 127
    gnsscorr_mex('load_if', h, int8(x), 2, settings.samplingFreq);
    src = 0;
elseif hasRecord
    gnsscorr_mex('load_if', h, int16(x), 2, settings.samplingFreq);
    gnsscorr_mex('acq_from_record', h, 0, numel(longSignal));
    src = 1;
else
    gnsscorr_mex('acq_set_signal', h, single(x));
    src = 1;
end
end
