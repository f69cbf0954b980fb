function h = gnsscorr_context(key, varargin)
%GNSSCORR_CONTEXT  One GPU context per IF record, kept between calls (persistent), so that the reference's call sequence
%   acquisition(data, settings) ... tracking(fid, channel, settings) (postProcessing.m:100,124) needs no extra argument.
%   h = gnsscorr_context(key)            handle of the record KEY (a file name, or 'longSignal'), [] if none yet
%   h = gnsscorr_context(key, 'new')     a fresh context for KEY (an older one is destroyed)
%       gnsscorr_context('', 'clear')    destroys every context (also done by the MEX file's mexAtExit)
persistent keys handles
if isempty(keys), keys = {};  handles = []; end
cmd = '';
if nargin > 1, cmd = varargin{1}; end
if strcmp(cmd, 'clear')
    for k = 1:numel(handles), gnsscorr_mex('destroy', handles(k)); end
    keys = {};  handles = [];  h = [];
    return
end
idx = 0;
for k = 1:numel(keys)
    if strcmp(keys{k}, key), idx = k; end
end
if strcmp(cmd, 'new')
    if idx > 0, gnsscorr_mex('destroy', handles(idx)); else, idx = numel(keys) + 1; end
    device = 0;
    if nargin > 2, device = varargin{2}; end
    keys{idx} = key;
    handles(idx) = gnsscorr_mex('create', device);
end
if idx > 0, h = handles(idx); else, h = []; end
end
