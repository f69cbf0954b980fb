"""Receiver settings struct — mirror of GPS/GPS_L1CA/initSettings.m (field names unchanged so a
MATLAB `settings` struct maps 1:1 onto it)."""
from __future__ import annotations

from types import SimpleNamespace


def skip_samples(settings) -> int:
    """Samples in front of the acquisition buffer and of every tracking seek.  The reference seeks
    dataAdaptCoeff*settings.skipNumberOfBytes BYTES (postProcessing.m:74; tracking.m:145-153; GLONASS names the field
    skipNumberOfSamples, GLO_GL1/include/postProcessing.m:74): with schar components that is skipNumberOfBytes samples,
    with int16 components (2 bytes each) skipNumberOfBytes/2 - an odd value would start in the middle of a component."""
    skip = int(getattr(settings, "skipNumberOfSamples", getattr(settings, "skipNumberOfBytes", 0)))
    if str(getattr(settings, "dataType", "schar")) == "int16":
        if skip % 2:
            raise ValueError("int16 record: dataAdaptCoeff*skipNumberOfBytes bytes is not a whole number of samples")
        return skip // 2
    return skip


def initSettings() -> SimpleNamespace:
    """GPS L1 C/A defaults (GPS/GPS_L1CA/initSettings.m:44-136)."""
    s = SimpleNamespace()
    s.msToProcess = 60000            # :44
    s.numberOfChannels = 12          # :47
    s.skipNumberOfBytes = 0          # :53
    s.fileName = "../../../L1_IF20KHz_FS18MHz.bin"   # :58
    s.dataType = "schar"             # :60
    s.fileType = 2                   # :65  (1 = real, 2 = I/Q interleaved)
    s.IF = 20e3                      # :68
    s.samplingFreq = 18e6            # :69
    s.codeFreqBasis = 1.023e6        # :70
    s.codeLength = 1023.0            # :73
    s.skipAcquisition = 0            # :77
    s.acqSatelliteList = list(range(1, 33))  # :80
    s.acqSearchBand = 7000           # :83
    s.acqNonCohTime = 20             # :85
    s.acqThreshold = 3.5             # :87
    s.acqSearchStep = 500            # :89
    s.resamplingThreshold = 8e6      # :91
    s.resamplingflag = 0             # :93
    s.dllDampingRatio = 0.7          # :97
    s.dllNoiseBandwidth = 1.5        # :98
    s.dllCorrelatorSpacing = 0.5     # :99
    s.pllDampingRatio = 0.7          # :102
    s.pllNoiseBandwidth = 20         # :103
    s.intTime = 0.001                # :105
    s.navSolPeriod = 500
    s.elevationMask = 5
    s.useTropCorr = 1
    s.truePosition = SimpleNamespace(E=float("nan"), N=float("nan"), U=float("nan"))
    s.plotTracking = 1
    s.plotAcquisition = 1
    s.plotNavigation = 1
    s.c = 299792458
    s.startOffset = 68.802
    s.CNo = SimpleNamespace(accTime=0.001, VSMinterval=40)  # :133-136
    return s


def initSettings_GAL_E1C() -> SimpleNamespace:
    """Galileo E1 B/C defaults (GAL/GAL_E1C/initSettings.m): only the fields the hot path reads."""
    s = initSettings()
    s.codeLength = 4092              # :74  (8184 half-chips after BOC(1,1))
    s.codeFreqBasis = 1.023e6        # :71
    s.samplingFreq = 18e6            # :70
    s.IF = 20e3                      # :69
    s.acqSatelliteList = list(range(1, 37))   # the package searches Galileo SVIDs 1..36 (results are sized for 50)
    s.resamplingThreshold = 50e6
    s.navSolPeriod = 200
    s.elevationMask = 10
    s.acqSearchBand = 7000           # :83
    s.acqNonCohTime = 1              # :85
    s.acqSearchStep = 150            # :87
    s.acqThreshold = 10              # :89
    s.dllDampingRatio = 0.7          # :96
    s.dllNoiseBandwidth = 1.5        # :97
    s.dllCorrelatorSpacing = 0.3     # :98
    s.pllDampingRatio = 0.7          # :101
    s.pllNoiseBandwidth = 15         # :102
    s.intTime = 0.004                # :104
    s.pilotTRKflag = 1               # :106
    s.CNo = SimpleNamespace(accTime=0.004, VSMinterval=400)  # :138-140
    return s


def initSettings_GPS_L5C() -> SimpleNamespace:
    """GPS L5 defaults (GPS/GPS_L5C/initSettings.m): only the fields the hot path reads."""
    s = initSettings()
    s.IF = 20e3                      # :63
    s.samplingFreq = 18e6            # :64
    s.codeLength = 10230             # :67
    s.codeFreqBasis = 10.23e6        # :68
    s.acqSearchBand = 5000           # :77
    s.acqNonCohTime = 25             # :79
    s.acqThreshold = 4.5             # :81
    s.acqSearchStep = 500            # :83
    s.dllDampingRatio = 0.7          # :90
    s.dllNoiseBandwidth = 2          # :91
    s.dllCorrelatorSpacing = 0.5     # :92
    s.pllDampingRatio = 0.7          # :94
    s.pllNoiseBandwidth = 15         # :95
    s.intTime = 0.001                # :97
    s.pilotTRKflag = 0               # :99
    s.CNo = SimpleNamespace(accTime=0.001, VSMinterval=400)  # :128-130
    s.carrFreqBasis = 1176.45e6      # :132
    s.resamplingThreshold = 50e6
    return s


def initSettings_GLO_GL1() -> SimpleNamespace:
    """GLONASS L1OF defaults (GLO/GLO_GL1/initSettings.m): only the fields the hot path reads."""
    s = initSettings()
    s.freqSpacing = 562.5e3          # :73
    del s.skipNumberOfBytes
    s.skipNumberOfSamples = 0        # :55 (this package names the field ...Samples; same dataAdaptCoeff* arithmetic)
    s.acqSatelliteList = list(range(-7, 7))  # :89 frequency numbers K
    s.acqSearchBand = 5000           # :91
    s.acqNonCohTime = 20             # :93
    s.acqThreshold = 2.0             # :95
    s.acqSearchStep = 500            # :97
    s.IF = 0.0                       # :77  nominal IF of channel K = 0
    s.samplingFreq = 12e6            # :79
    s.codeFreqBasis = 0.511e6        # :80
    s.codeLength = 511               # :83
    s.dllNoiseBandwidth = 2          # :107
    s.dllCorrelatorSpacing = 0.5     # :108
    s.pllNoiseBandwidth = 25         # :112
    s.intTime = 0.001                # :114
    s.CNo = SimpleNamespace(accTime=0.001, VSMinterval=40)  # :144-146
    s.resamplingThreshold = 18e6
    s.startOffset = 65.0
    return s


def initSettings_GLO_GL2() -> SimpleNamespace:
    """GLONASS L2OF defaults: GLO/GLO_GL2/initSettings.m differs from GLO_GL1's in the file name (:61) and in
    freqSpacing = 437.5 kHz (:73) only."""
    s = initSettings_GLO_GL1()
    s.freqSpacing = 437.5e3          # :73
    return s


def initSettings_BDS_B1I() -> SimpleNamespace:
    """BDS B1I defaults (BDS/B1I/initSettings.m): only the fields the hot path reads."""
    s = initSettings()
    s.IF = 20e3                      # :71
    s.samplingFreq = 18e6            # :72
    s.codeFreqBasis = 2.046e6        # :73
    s.codeLength = 2046              # :76
    s.acqSatelliteList = list(range(6, 59))  # :83
    s.acqSearchBand = 10             # :85 (kHz in this package's acquisition.m)
    s.acqThreshold = 2               # :87
    s.stepSize = 125                 # :94
    s.dllDampingRatio = 0.7          # :99
    s.dllNoiseBandwidth = 4          # :100
    s.dllCorrelatorSpacing = 0.5     # :101
    s.pllDampingRatio = 0.7          # :104
    s.pllNoiseBandwidth = 35         # :105
    s.intTime = 0.001                # :107
    s.CNo = SimpleNamespace(accTime=0.001, VSMinterval=400)  # :141-143
    s.resamplingThreshold = 9e6
    s.elevationMask = 10
    s.startOffset = 120.0
    s.saveResults = 0
    s.skipNumberOfSamples = 0        # declared next to skipNumberOfBytes; tracking.m and postProcessing.m read ...Bytes
    del s.acqNonCohTime, s.acqSearchStep     # this package's circshift search has neither (acqSearchBand is in kHz, stepSize in Hz)
    return s


def _ten23(s, **kw):
    """Common part of the 10.23-Mcps, 10230-chip, 1-ms packages (B2a, B3I, E5a, E5b)."""
    s.IF = 20e3
    s.samplingFreq = 18e6
    s.codeLength = 10230
    s.codeFreqBasis = 10.23e6
    s.acqSearchBand = 5000
    s.dllDampingRatio = 0.7
    s.dllCorrelatorSpacing = 0.5
    s.pllDampingRatio = 0.7
    s.intTime = 0.001
    drop = kw.pop("_drop", ())
    for k, v in kw.items():
        setattr(s, k, v)
    for k in drop:
        delattr(s, k)
    return s


def initSettings_BDS_B2a() -> SimpleNamespace:
    """BDS B2a defaults (BDS/B2a/initSettings.m:43-127)."""
    return _ten23(initSettings(), numberOfChannels=12, acqSatelliteList=list(range(19, 31)) + list(range(32, 47)) + [59, 60],  # :45,73
                  acqNonCohTime=15, acqThreshold=5, acqSearchStep=500,          # :78-82
                  dllNoiseBandwidth=2, pllNoiseBandwidth=15, pilotTRKflag=0,    # :90,94,98
                  CNoInterval=200, carrFreqBasis=1176.45e6, resamplingThreshold=50e6, _drop=("CNo",))  # :125 (B2a estimates C/N0 with Calc_CNo_PLD), :127


def initSettings_BDS_B3I() -> SimpleNamespace:
    """BDS B3I defaults (BDS/B3I/initSettings.m:43-132)."""
    return _ten23(initSettings(), numberOfChannels=15, acqSatelliteList=list(range(1, 64)),      # :45,76
                  acqNonCohTime=10, acqThreshold=3, acqSearchStep=500,          # :80-84
                  dllNoiseBandwidth=2, pllNoiseBandwidth=15,                    # :92,96
                  CNo=SimpleNamespace(accTime=0.001, VSMinterval=40), carrFreqBasis=1268.520e6,  # :127-132
                  resamplingThreshold=45e6, startOffset=94.0, resamplingFlag=0, _drop=("resamplingflag",))   # this package spells it resamplingFlag


def initSettings_GAL_E5a() -> SimpleNamespace:
    """Galileo E5a defaults (GAL/GAL_E5a/initSettings.m:6-104)."""
    return _ten23(initSettings(), numberOfChannels=12, acqSatelliteList=list(range(1, 37)),      # :9,44
                  acqNonCohTime=15, acqThreshold=4.5, acqSearchStep=500,        # :49-53
                  dllNoiseBandwidth=1.5, pllNoiseBandwidth=15, pilotTRKflag=1,  # :61,66,70
                  CNo=SimpleNamespace(accTime=0.001, VSMinterval=100), carrFreqBasis=1176.45e6, resamplingThreshold=45e6)  # :100-104


def initSettings_GAL_E5b() -> SimpleNamespace:
    """Galileo E5b defaults (GAL/GAL_E5b/initSettings.m:44-144; the second dll block :103-105 wins)."""
    return _ten23(initSettings(), numberOfChannels=12, acqSatelliteList=list(range(1, 37)),      # :47,83
                  acqNonCohTime=15, acqThreshold=4.5, acqSearchStep=60,         # :88-92
                  dllNoiseBandwidth=1.5, pllNoiseBandwidth=25, pilotTRKflag=1,  # :104,108,112
                  CNo=SimpleNamespace(accTime=0.001, VSMinterval=100), carrFreqBasis=1207.14e6, resamplingThreshold=45e6)  # :140-144


def initSettings_BDS_B1C() -> SimpleNamespace:
    """BDS B1C defaults (BDS/B1C/initSettings.m:55-143): 10-ms integration, 0.06-chip correlator spacing."""
    s = initSettings()
    s.IF = 20e3                      # :55
    s.samplingFreq = 18e6            # :56
    s.FEBW = 27e6                    # :59 front-end bandwidth (CalcWeighingFactor.m)
    s.acqSatelliteList = list(range(1, 63))  # :67
    s.pilotTRKflag = 1               # :71
    s.numberOfChannels = 15          # :73
    s.codeLength = 10230             # :82
    s.codeFreqBasis = 1.023e6        # :84
    s.carrFreqBasis = 1575.42e6      # :86
    s.acqSearchBand = 5000           # :92
    s.acqCohT = 10                   # :95
    s.acqStep = 1000 / s.acqCohT / 2  # :97
    s.acqThreshold = 10              # :99
    s.pilotACQflag = 1               # :69
    s.dllDampingRatio = 0.7          # :107
    s.dllNoiseBandwidth = 1          # :108
    s.dllCorrelatorSpacing = 0.06    # :111
    s.pllDampingRatio = 0.7          # :113
    s.pllNoiseBandwidth = 18         # :114
    s.intTime = 0.01                 # :116
    s.CNoInterval = 50               # :143
    if hasattr(s, "CNo"):
        del s.CNo                    # B1C estimates C/N0 with Calc_CNo_PLD (not on the hot path)
    s.resamplingThreshold = 15e6
    s.navSolPeriod = 200
    del s.acqNonCohTime, s.acqSearchStep     # circshift search: acqCohT / acqStep instead
    return s


def initSettings_GPS_L2C() -> SimpleNamespace:
    """GPS L2C defaults (GPS/GPS_L2C/initSettings.m:44-143): 8 Msps, 20-ms blocks of the RZ-multiplexed CM / CL codes."""
    s = initSettings()
    s.numberOfChannels = 12          # :47
    s.IF = 20e3                      # :69
    s.samplingFreq = 8e6             # :70
    s.codeLength = 10230             # :74
    s.codeFreqBasis = 0.5115e6       # :75
    s.CLCodeLength = 10230 * 75      # :76
    s.acqSatelliteList = list(range(1, 33))  # :84
    s.acqSearchBand = 10             # :87 (kHz in this package's acquisition.m)
    s.acqThreshold = 1.5             # :89
    s.acqCohT = 20                   # :91
    s.acqStep = (1000 / 2) / 20 / 2  # :94
    s.dllDampingRatio = 0.7          # :102
    s.dllNoiseBandwidth = 4          # :103
    s.dllCorrelatorSpacing = 0.25    # :104
    s.pllDampingRatio = 0.7          # :106
    s.pllNoiseBandwidth = 10         # :107
    s.intTime = 0.02                 # :109
    s.pilotTRKflag = 0               # :111
    s.CNo = SimpleNamespace(accTime=0.02, VSMinterval=40)  # :141-143
    s.resamplingThreshold = 6e6
    del s.acqNonCohTime, s.acqSearchStep     # circshift search: acqCohT / acqStep instead
    return s
