function pkg = gnsscorr_package(name, settings)
%GNSSCORR_PACKAGE  Per-package table of the GPU drop-ins (gnsscorr_tracking.m / gnsscorr_acquisition.m).
%   pkg = gnsscorr_package(name, settings) describes how package NAME of CU-SDR-Collection builds its code replicas, closes its
%   loops and records its results, so that ONE generic tracking / acquisition wrapper serves all twelve packages.  Every entry
%   cites the lines of the package's own tracking.m it stands for.  The code generators and calcLoopCoef*.m called from here
%   are the PACKAGE'S OWN files (they stay MATLAB, SURVEY.md section 8b); CNoVSM / Calc_CNo_PLD are evaluated by the library
%   with the loops and come back with the records.
%   Written for this repository; not a copy of any reference file.

pkg.name        = name;
pkg.idField     = 'PRN';     % channel field naming the satellite
pkg.byStatus    = false;     % GLONASS: a channel is active when status ~= '-' (K = 0 is a valid frequency number)
pkg.indexScale  = 1;         % R: 2 for BOC(1,1) / RZ half-chip tables
pkg.armMult     = [];        % per-arm ramp multipliers ([1 1 6] with a BOC(6,1) arm)
pkg.windows     = [];        % per-arm LDS staging windows (GPS L2C CL)
pkg.pllKind     = 1;         % 0: second-order PLL of GPS L1 C/A (tracking.m:308-311), 1: three-state filter (calcLoopCoefCarr.m)
pkg.pilotMode   = 0;         % gc_track_params.pilot_combine when the pilot is tracked
pkg.pllWeight   = [];
pkg.dllWeight   = [];
pkg.dllScale    = 0;
pkg.phaseCount  = 0;         % GPS L2C: 75 CL segments
pkg.codeFreqFromChannel = false;   % initial codeFreq = channel.codeFreq (GPS_L5C tracking.m:165)
pkg.doubledCode = false;     % GPS L2C runs on the RZ-doubled code (tracking.m:107-109,171)
pkg.minusOne    = true;      % fseek to skip + codePhase - 1 (GPS_L2C tracking.m:153 omits the -1)
pkg.int16Branch = false;     % tracking.m has the int16 seek / ftell branch
pkg.qiOrder     = false;     % GLONASS front ends deliver Q first (GLO_GL1 tracking.m:227)
pkg.pilotFields = 'prompt';  % which Pilot_* fields trackResults holds: 'none' | 'prompt' | 'all'
pkg.cno         = 'VSM';     % 'VSM': CNoVSM.m every CNo.VSMinterval epochs; 'PLD': Calc_CNo_PLD.m every CNoInterval epochs
pkg.skipField   = 'skipNumberOfBytes';
pilotOn = isfield(settings, 'pilotTRKflag') && settings.pilotTRKflag == 1;
pad = @(c, n) [c(n) c c(1)];   % [c(end) c c(1)], tracking.m:158

switch name
    case 'GPS_L1CA'
        pkg.pllKind = 0;  pkg.int16Branch = true;
        pkg.tables = @(id, s) {pad(generateCAcode(id), 1023)};                                   % tracking.m:156-158
    case 'BDS_B1I'
        pkg.tables = @(id, s) {pad(generateCAcode53(id), s.codeLength)};                         % BDS/B1I tracking.m:144-146
    case 'BDS_B3I'
        pkg.codeFreqFromChannel = true;  pkg.int16Branch = true;
        pkg.tables = @(id, s) {pad(generateB3Icode(id), s.codeLength)};                          % BDS/B3I tracking.m:150-152
    case {'GLO_GL1', 'GLO_GL2'}
        pkg.idField = 'K';  pkg.byStatus = true;  pkg.qiOrder = true;  pkg.skipField = 'skipNumberOfSamples';
        pkg.tables = @(id, s) {pad(generateCAcode(0, s.codeFreqBasis, 511), 511)};               % GLO_GL1 tracking.m:88-90
    case 'GPS_L5C'
        pkg.codeFreqFromChannel = true;  pkg.pilotMode = 1;                                      % pilot rotated by -pi/2, :336-348
        if pilotOn
            pkg.tables = @(id, s) {pad(generateL5Icode(id, s), s.codeLength), pad(generateL5Qcode(id, s), s.codeLength)};   % :151-159
        else
            pkg.tables = @(id, s) {pad(generateL5Icode(id, s), s.codeLength)};
        end
    case 'BDS_B2a'
        pkg.codeFreqFromChannel = true;  pkg.pilotMode = 1;  pkg.cno = 'PLD';
        if pilotOn
            pkg.tables = @(id, s) {pad(generateB2aDataCode(id, s), s.codeLength), pad(generateB2aPilotCode(id, s), s.codeLength)};   % BDS/B2a tracking.m:156-165
        else
            pkg.tables = @(id, s) {pad(generateB2aDataCode(id, s), s.codeLength)};
        end
    case {'GAL_E5a', 'GAL_E5b'}
        pkg.codeFreqFromChannel = true;  pkg.pilotMode = 1;  pkg.int16Branch = true;
        if strcmp(name, 'GAL_E5a')
            gi = @(id) generateE5aIcode(id, 2);  gq = @(id) generateE5aQcode(id, 1);             % GAL_E5a tracking.m:148-156
        else
            gi = @(id) generateE5bIcode(id, 2);  gq = @(id) generateE5bQcode(id, 1);
        end
        if pilotOn
            pkg.tables = @(id, s) {gnsscorr_first(pad(gi(id), s.codeLength), s.codeLength + 2), pad(gq(id), s.codeLength)};
        else
            pkg.tables = @(id, s) {gnsscorr_first(pad(gi(id), s.codeLength), s.codeLength + 2)};
        end
    case 'GAL_E1C'
        pkg.indexScale = 2;  pkg.pilotMode = 2;  pkg.pilotFields = 'none';                        % GAL_E1C tracking.m:139-150,303-311
        if pilotOn
            pkg.tables = @(id, s) {pad(generateE1Bcode(id), 2*s.codeLength), pad(generateE1Ccode(id), 2*s.codeLength)};
        else
            pkg.tables = @(id, s) {pad(generateE1Bcode(id), 2*s.codeLength)};
        end
    case 'BDS_B1C_NB'
        pilotOn = true;                                                                            % NB_tracking.m always tracks data + pilot BOC(1,1)
        pkg.indexScale = 2;  pkg.codeFreqFromChannel = true;  pkg.pilotMode = 3;  pkg.cno = 'PLD';
        pkg.pllWeight = [11 29];  pkg.dllWeight = [11 29];  pkg.dllScale = 1 - settings.dllCorrelatorSpacing;   % NB_tracking.m:341-349
        pkg.tables = @(id, s) {pad(generateDataBOC11(s, id), 2*s.codeLength), pad(generatePilotBOC11(s, id), 2*s.codeLength)};
    case 'BDS_B1C_WB'
        pilotOn = true;
        pkg.indexScale = 2;  pkg.armMult = [1 1 6];  pkg.codeFreqFromChannel = true;  pkg.pilotMode = 4;  pkg.cno = 'PLD';
        pkg.pilotFields = 'all';
        factor = CalcWeighingFactor(settings);                                                     % WB_tracking.m:134
        pkg.pllWeight = [1 3];  pkg.dllWeight = [factor, 1 - factor];  pkg.dllScale = 1 - settings.dllCorrelatorSpacing;   % :382,396,403
        pkg.tables = @(id, s) {pad(generateDataBOC11(s, id), 2*s.codeLength), pad(generatePilotBOC11(s, id), 2*s.codeLength), ...
                               pad(generatePilotBOC61(s, id), 12*s.codeLength)};                    % :176-188
    case 'GPS_L2C'
        pkg.doubledCode = true;  pkg.minusOne = false;  pkg.pilotMode = 2;  pkg.pilotFields = 'all';
        pilotOn = isfield(settings, 'pilotTRKflag') && settings.pilotTRKflag ~= 0;
        if pilotOn
            pkg.phaseCount = 75;  pkg.windows = [0, 2*settings.codeLength + 4];
            pkg.tables = @(id, s) {pad(generateCMcode(id, s), 2*s.codeLength), pad(generateCLcode(id, s), 2*s.CLCodeLength)};   % GPS_L2C tracking.m:156-166
        else
            pkg.tables = @(id, s) {pad(generateCMcode(id, s), 2*s.codeLength)};
        end
    otherwise
        error('gnsscorr:package', 'unknown package %s', name);
end
pkg.pilotOn = pilotOn && pkg.pilotMode ~= 0;
if ~pkg.pilotOn
    pkg.pilotMode = 0;  pkg.pllWeight = [];  pkg.dllWeight = [];  pkg.dllScale = 0;
end
% epochs to process: msToProcess for the 1-ms packages, round(msToProcess/1000/intTime) otherwise (GAL_E1C tracking.m:51)
pkg.numEpochs = round(settings.msToProcess / 1000 / settings.intTime);
end

function v = gnsscorr_first(v, n)
% the first n entries of a table (GAL_E5a tracking.m:148-150 pads the TIERED code; only codeLength + 2 entries can be indexed)
v = v(1:n);
end
