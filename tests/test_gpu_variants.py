"""Kernel features behind the reference's signal-variant matrix that the L1 C/A, E1 and L5
tests do not reach: ternary RZ tables and table windows/offsets (GPS L2C CM/CL), very long blocks, three
arms, Q/I sample order in the closed loop (GLONASS), int16 records in the closed loop."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import gnss_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _noise_iq(n, seed, scale=20):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(scale * rng.standard_normal(2 * n)), -127, 127).astype(np.int8)


def _block(b, k, **kw):
    b[k].channel = kw["channel"]
    b[k].blksize = kw["n"]
    b[k].first_sample = kw["s0"]
    b[k].rem_code_phase = kw["rem"]
    b[k].code_phase_step = kw["step"]
    b[k].el_spacing = kw["d"]
    b[k].carr_freq = kw["f"]
    b[k].rem_carr_phase = kw["phi"]
    for a, off in enumerate(kw.get("off", ())):
        b[k].table_offset[a] = off


def test_l2c_style_ternary_tables_windows_and_long_blocks(engine):
    """GPS_L2C/include/tracking.m: RZ-interleaved CM code ({+-1, 0}, 20 460 entries, 20-ms blocks of
    160 000 samples at 8 Msps) and the CL arm read through a window of a much longer table
    (index + 2L*(CLCodePhase-1), :261)."""
    fs, L = 8e6, 20460.0
    n_if = 340000
    iq = _noise_iq(n_if, 1)
    rng = np.random.default_rng(2)
    cm = rng.choice([-1.0, 1.0], size=10230)
    cm_rz = np.zeros(20460)
    cm_rz[0::2] = cm                       # generateCMcode.m:110-111: chip, 0, chip, 0, ...
    cl_long = np.zeros(20460 * 6)
    cl_long[1::2] = rng.choice([-1.0, 1.0], size=10230 * 6)
    tab_cm = O.pad_code(cm_rz)
    tab_cl = O.pad_code(cl_long)           # long table; one block touches a 20 462-entry window
    engine.load_if(iq, fs=fs)
    engine.set_channel(0, [tab_cm.astype(np.int8), tab_cl.astype(np.int8)], windows=[0, 20462])
    step = 2 * 0.5115e6 / fs * (1 + 1e-6)
    b = engine.make_blocks(3)
    descs = []
    for k, clphase in enumerate((1, 3, 6)):
        rem = float(rng.uniform(0, step))
        n = O.blksize_for(L, rem, step)
        d = dict(channel=0, n=n, s0=int(rng.integers(0, n_if - n)), rem=rem, step=step, d=0.25,
                 f=float(rng.uniform(-4e3, 4e3)), phi=float(rng.uniform(-3, 3)), off=(0, int(L) * (clphase - 1)))
        descs.append(d)
        _block(b, k, **d)
    got = engine.correlate(b)
    for k, d in enumerate(descs):
        raw = O.raw_from_if(iq, d["s0"], d["n"])
        ref, _, _ = O.correlate_block(raw, [tab_cm, tab_cl], d["rem"], d["step"], d["d"], d["f"], d["phi"], fs, L,
                                      table_offset=list(d["off"]))
        sc = np.sum(np.abs(iq[2 * d["s0"]:2 * (d["s0"] + d["n"])].astype(np.float64)))
        assert d["n"] > 159000
        assert np.max(np.abs(got[k, :2] - ref)) < TOL * sc, k
        assert not got[k, 2].any()


def test_three_arms_same_ramp(engine):
    iq = _noise_iq(60000, 3)
    rng = np.random.default_rng(4)
    tabs = [O.pad_code(rng.choice([-1.0, 1.0], size=2046)) for _ in range(3)]
    engine.load_if(iq, fs=18e6)
    engine.set_channel(5, [t.astype(np.int8) for t in tabs], index_scale=2.0)
    step = 1.023e6 / 18e6
    b = engine.make_blocks(4)
    descs = []
    for k in range(4):
        rem = float(rng.uniform(0, step))
        d = dict(channel=5, n=O.blksize_for(1023.0, rem, step), s0=int(rng.integers(0, 40000)), rem=rem, step=step,
                 d=0.3, f=2.2e4, phi=0.1 * k)
        descs.append(d)
        _block(b, k, **d)
    for generic in (False, True):
        engine.force_generic_kernel(generic)
        try:
            got = engine.correlate(b)
        finally:
            engine.force_generic_kernel(False)
        for k, d in enumerate(descs):
            ref, _, _ = O.correlate_block(O.raw_from_if(iq, d["s0"], d["n"]), tabs, d["rem"], d["step"], d["d"], d["f"],
                                          d["phi"], 18e6, 1023.0, r=2.0)
            sc = np.sum(np.abs(iq[2 * d["s0"]:2 * (d["s0"] + d["n"])].astype(np.float64)))
            assert np.max(np.abs(got[k] - ref)) < TOL * sc, (generic, k)


def test_b1c_wideband_mixed_ramp_multipliers(engine):
    """BDS/B1C/include/WB_tracking.m:176-188,285-369: data BOC(1,1), pilot BOC(1,1) (index ceil(t), t the
    R = 2 ramp) and pilot BOC(6,1) (index ceil(6*t) into a 12*L+2 table), 10-ms blocks, d = 0.06 chip."""
    fs, L = 18e6, 10230.0
    n_if = 200000
    iq = _noise_iq(n_if, 5)
    rng = np.random.default_rng(6)
    chips_d, chips_p = rng.choice([-1.0, 1.0], size=10230), rng.choice([-1.0, 1.0], size=10230)
    boc11 = lambda c: (np.repeat(c, 2) * np.tile([-1.0, 1.0], c.size))          # generateDataBOC11.m: chip x [-1,+1]
    boc61 = lambda c: (np.repeat(c, 12) * np.tile([-1.0, 1.0], 6 * c.size))     # generatePilotBOC61.m:89-96
    tabs = [O.pad_code(boc11(chips_d)), O.pad_code(boc11(chips_p)), O.pad_code(boc61(chips_p))]
    assert tabs[2].shape == (122762,)
    engine.load_if(iq, fs=fs)
    engine.set_channel(7, [t.astype(np.int8) for t in tabs], index_scale=2.0, arm_mult=[1, 1, 6])
    step = 1.023e6 / fs * (1 - 2e-6)
    b = engine.make_blocks(2)
    descs = []
    for k in range(2):
        rem = 0.0 if k == 0 else float(rng.uniform(0, step))
        n = O.blksize_for(L, rem, step)
        d = dict(channel=7, n=n, s0=int(rng.integers(0, n_if - n)), rem=rem, step=step, d=0.06,
                 f=float(rng.uniform(1.5e4, 2.5e4)), phi=float(rng.uniform(-3, 3)))
        descs.append(d)
        _block(b, k, **d)
    got = engine.correlate(b)
    for k, d in enumerate(descs):
        ref, _, _ = O.correlate_block(O.raw_from_if(iq, d["s0"], d["n"]), tabs, d["rem"], d["step"], d["d"], d["f"],
                                      d["phi"], fs, L, r=2.0, arm_mult=[1.0, 1.0, 6.0])
        sc = np.sum(np.abs(iq[2 * d["s0"]:2 * (d["s0"] + d["n"])].astype(np.float64)))
        assert d["n"] > 179000 and np.max(np.abs(got[k] - ref)) < TOL * sc, k


def test_closed_loop_with_int16_and_qi_records(engine, l1ca_scene):
    """tracking.m:145-148,212-213 (int16) and GLO_GL1/include/tracking.m:227 (Q,I order): the closed loop on
    a 16-bit copy and on a component-swapped copy of the record reproduces the int8 I/Q run."""
    import cu_sdr_collection_amd as P
    from cu_sdr_collection_amd import _lib as L
    from cu_sdr_collection_amd.receiver import track_params
    S, sats, iq = l1ca_scene
    S.msToProcess = 60
    inits = []
    for i, s in enumerate(sats[:2]):
        inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 4.0,
                                       code_freq=S.codeFreqBasis, code_phase=int(np.ceil(s.code_phase_samples)) + 1))

    def run(load):
        load()
        for i, s in enumerate(sats[:2]):
            engine.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
        f, done, st = engine.track(track_params(S), inits)
        assert st == 0 and list(done) == [60, 60]
        return f

    base = run(lambda: engine.load_if(iq, fs=S.samplingFreq))
    i16 = run(lambda: engine.load_if(iq.astype(np.int16), fs=S.samplingFreq))
    swapped = np.empty_like(iq)
    swapped[0::2], swapped[1::2] = iq[1::2], iq[0::2]
    qi = run(lambda: engine.load_if(swapped, layout=L.GC_QI, fs=S.samplingFreq))
    scale = 2.0 * 18000 * 28.0
    for other in (i16, qi):
        assert np.array_equal(other["absoluteSample"], base["absoluteSample"])
        for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L"):
            assert np.max(np.abs(other[f] - base[f])) < 1e-6 * scale, f
        assert np.max(np.abs(other["carrFreq"] - base["carrFreq"])) < 1e-4


def test_lane_kernel_work_decompositions(engine):
    """corr_lane.hip's three work decompositions against the oracle at a 10.23-Mcps rate (0.57 chip per
    sample: several table entries per 8 samples, so the lane kernel is the one that runs): (A) periodic replay
    list, one block per wavefront; (B) few blocks, split 16-fold and more with partial sums; (C) a long list of
    unrelated blocks, one per workgroup, combined in LDS.  Two channels with different two-arm tables; one of
    them long enough (2 x 16 384 entries) that the tables are staged as f16 instead of f32."""
    fs = 18e6
    n_if = 400000
    iq = _noise_iq(n_if, 11)
    rng = np.random.default_rng(12)
    engine.load_if(iq, fs=fs)
    L = {6: 10230.0, 7: 16382.0}
    tabs = {}
    for ch in (6, 7):
        tabs[ch] = [O.pad_code(rng.choice([-1.0, 1.0], size=int(L[ch]))) for _ in range(2)]
        engine.set_channel(ch, [t.astype(np.int8) for t in tabs[ch]])
    step = 10.23e6 / fs

    def make(nb, chan_of, nmax=None):
        b = engine.make_blocks(nb)
        descs = []
        for k in range(nb):
            ch = chan_of(k)
            rem = float(rng.uniform(0, step)) if k % 5 else 0.0   # rem = 0: samples exactly on table edges
            st = step * (1 + float(rng.uniform(-2e-6, 2e-6))) if k % 5 else step
            n = O.blksize_for(L[ch], rem, st)
            if nmax:
                n = min(n, nmax)
            d = dict(channel=ch, n=n, s0=int(rng.integers(0, n_if - n)), rem=rem, step=st, d=0.5 if k % 3 else 0.3,
                     f=float(rng.uniform(-5e3, 5e3)) + 2.0e6 * (k % 2), phi=float(rng.uniform(-3, 3)))
            descs.append(d)
            _block(b, k, **d)
        return b, descs

    def check(got, descs, idx):
        for k in idx:
            d = descs[k]
            ref, _, _ = O.correlate_block(O.raw_from_if(iq, d["s0"], d["n"]), tabs[d["channel"]], d["rem"], d["step"],
                                          d["d"], d["f"], d["phi"], fs, L[d["channel"]])
            sc = np.sum(np.abs(iq[2 * d["s0"]:2 * (d["s0"] + d["n"])].astype(np.float64)))
            assert np.max(np.abs(got[k, :2] - ref)) < TOL * sc, (k, d)
            assert not got[k, 2].any()

    # (B) three blocks: split over many wavefronts
    b, descs = make(3, lambda k: 6 + (k & 1))
    check(engine.correlate(b), descs, range(3))
    # (C) 700 unrelated blocks (short ones, to keep the oracle quick): one workgroup each
    b, descs = make(700, lambda k: 6 + ((k * 7919) >> 3 & 1), nmax=3000)
    got = engine.correlate(b)
    check(got, descs, list(range(0, 700, 37)) + [699])
    # (A) periodic replay list: channels 6, 7, 6, 7, ... over 1100 epochs
    b, descs = make(2200, lambda k: 6 + (k & 1), nmax=2500)
    engine.replay_prepare(b)
    engine.replay_launch()
    got = engine.replay_fetch()
    check(got, descs, list(range(0, 2200, 97)) + [2198, 2199])
    assert np.max(np.abs(got - engine.correlate(b))) < 0.2   # decomposition (C) of the same list: float sums only reorder


@pytest.mark.tuning
@pytest.mark.parametrize("case", ["l5_50msps_share", "l5_50msps_three_ramps", "e1_share", "e1_three_ramps", "b1i_one_arm_qi", "l5_50msps_int16", "e1_int16_qi"])
def test_multi_transition_kernel_equals_the_other_kernels_and_the_oracle(engine, monkeypatch, case):
    """corr_multi.hip: periodic replay lists whose 16-sample chunks cross up to 2 (Galileo E1 / BDS B1I-type tables at 18 Msps) or
    4 (10.23-Mcps codes at 50 Msps) table entries.  The same list through the kernel that took it before (GC_NO_MULTI=1: the lane
    kernel or the 8-sample single-transition kernel) and, block by block, through the float64 oracle (tracking.m:247-300); blocks
    that start on exact chip edges with the nominal rational step (the tie-dense first block of every channel) included."""
    cfg = {"l5_50msps_share": dict(fs=50e6, L=10230, rate=10.23e6, R=1.0, arms=2, d=0.5, kt=4, layout="iq"),
           "l5_50msps_three_ramps": dict(fs=50e6, L=10230, rate=10.23e6, R=1.0, arms=2, d=0.3, kt=4, layout="iq"),
           "e1_share": dict(fs=18e6, L=4092, rate=1.023e6, R=2.0, arms=2, d=0.25, kt=2, layout="iq"),
           "e1_three_ramps": dict(fs=18e6, L=4092, rate=1.023e6, R=2.0, arms=2, d=0.1, kt=2, layout="iq"),
           "b1i_one_arm_qi": dict(fs=18e6, L=2046, rate=2.046e6, R=1.0, arms=1, d=0.5, kt=2, layout="qi"),
           "l5_50msps_int16": dict(fs=50e6, L=10230, rate=10.23e6, R=1.0, arms=2, d=0.5, kt=4, layout="iq", int16=True),
           "e1_int16_qi": dict(fs=18e6, L=4092, rate=1.023e6, R=2.0, arms=2, d=0.1, kt=2, layout="qi", int16=True)}[case]
    from cu_sdr_collection_amd import _lib as LIB
    fs, L, R, arms = cfg["fs"], float(cfg["L"]), cfg["R"], cfg["arms"]
    rng = np.random.default_rng(abs(hash(case)) % 1000 + 11)
    period, epochs = 3, 180                                   # 540 blocks >= GC_MULTI_MIN(1) x period x 256 CUs? no: see below
    monkeypatch.setenv("GC_MULTI_MIN", "1")
    cus = engine.device_info()[1]
    epochs = max(epochs, (8 * cus + period - 1) // period + 4)    # lists the replay launcher does not split: >= 8 blocks per CU
    step0 = cfg["rate"] / fs
    nmax = int(np.ceil(L / (step0 * (1 - 3e-5)))) + 2
    n_if = 6 * nmax
    iq = _noise_iq(n_if, 77)
    if cfg.get("int16"):
        iq = iq.astype(np.int16) * 37               # an int16 record (settings.dataType = 'int16'): values beyond the int8 range
    layout = LIB.GC_QI if cfg["layout"] == "qi" else LIB.GC_IQ
    engine.load_if(iq, layout=layout, fs=fs)
    tabs = {}
    for c in range(period):
        tabs[c] = [O.pad_code(rng.choice([-1.0, 1.0], size=int(L * R))) for _ in range(arms)]
        engine.set_channel(c, [t.astype(np.int8) for t in tabs[c]], index_scale=R)
    nb = period * epochs
    b = engine.make_blocks(nb)
    descs = []
    for k in range(nb):
        tie = k < period or k % 97 == 0                        # first block of every channel (and a few more): rem = 0, nominal step
        step = step0 if tie else step0 * (1 + float(rng.uniform(-3e-5, 3e-5)))
        rem = 0.0 if tie else float(rng.uniform(0, step))
        n = O.blksize_for(L, rem, step)
        dsc = dict(channel=k % period, n=n, s0=int(rng.integers(0, n_if - n)), rem=rem, step=step, d=cfg["d"],
                   f=float(rng.uniform(-2.5e4, 2.5e4)), phi=float(rng.uniform(-3, 3)))
        descs.append(dsc)
        _block(b, k, **dsc)
    engine.replay_prepare(b)
    engine.replay_launch()
    got = engine.replay_fetch().copy()
    assert engine.last_kernel() == 4, engine.last_kernel()
    monkeypatch.setenv("GC_NO_MULTI", "1")
    engine.replay_launch()
    other = engine.replay_fetch().copy()
    assert engine.last_kernel() != 4
    monkeypatch.delenv("GC_NO_MULTI")
    raw_all = iq.astype(np.float64)
    scale = np.array([np.sum(np.abs(raw_all[2 * d_["s0"]:2 * (d_["s0"] + d_["n"])])) for d_ in descs])
    dev = np.max(np.abs(got - other).reshape(nb, -1), axis=1) / scale
    assert dev.max() < 2 * TOL, (case, int(np.argmax(dev)), dev.max())
    assert not got[:, arms:].any()
    for k in list(range(period)) + [97, 194, nb - 1, nb // 2, nb // 3]:
        d_ = descs[k]
        raw = O.raw_from_if(iq, d_["s0"], d_["n"], swap_iq=cfg["layout"] == "qi")
        ref, _, _ = O.correlate_block(raw, tabs[d_["channel"]], d_["rem"], d_["step"], d_["d"], d_["f"], d_["phi"], fs, L, r=R)
        assert np.max(np.abs(got[k, :arms] - ref)) < TOL * scale[k], (case, k, np.max(np.abs(got[k, :arms] - ref)) / scale[k])


def _hybrid_list(engine, case):
    """A periodic replay list of three channels with a derived six-fold arm, prepared on `engine`: (blocks, descriptors, tables, record, cfg)."""
    cfg = {"e1_cboc": dict(fs=18e6, L=4092, rate=1.023e6, d=0.05, layout="iq"),
           "e1_cboc_qi": dict(fs=18e6, L=4092, rate=1.023e6, d=0.05, layout="qi"),
           "b1c_wb": dict(fs=18e6, L=10230, rate=1.023e6, d=0.06, layout="iq")}[case]
    from cu_sdr_collection_amd import _lib as LIB
    fs, L, R = cfg["fs"], float(cfg["L"]), 2.0
    rng = np.random.default_rng(abs(hash(case)) % 1000 + 23)
    period = 3
    cus = engine.device_info()[1]
    epochs = 16 * cus                                          # full rounds at 16 and 8 waves per CU, 0.8 at 12: the hybrid kernel's case (gc_cboc_takes)
    step0 = cfg["rate"] / fs
    nmax = int(np.ceil(L / (step0 * (1 - 3e-5)))) + 2
    n_if = 6 * nmax
    iq = _noise_iq(n_if, 79)
    layout = LIB.GC_QI if cfg["layout"] == "qi" else LIB.GC_IQ
    engine.load_if(iq, layout=layout, fs=fs)
    tabs = {}
    for c in range(period):
        data = rng.choice([-1.0, 1.0], size=int(L))
        pilot = rng.choice([-1.0, 1.0], size=int(L))
        boc11 = lambda x: (x[:, None] * np.array([1.0, -1.0])[None, :]).reshape(-1)
        boc61 = (pilot[:, None] * np.tile(np.array([1.0, -1.0]), 6)[None, :]).reshape(-1)
        tabs[c] = [O.pad_code(boc11(data)), O.pad_code(boc11(pilot)), O.pad_code(boc61)]
        engine.set_channel(c, [t.astype(np.int8) for t in tabs[c]], index_scale=R, arm_mult=[1, 1, 6])
    nb = period * epochs
    b = engine.make_blocks(nb)
    descs = []
    for k in range(nb):
        tie = k < period or k % 97 == 0
        step = step0 if tie else step0 * (1 + float(rng.uniform(-3e-5, 3e-5)))
        rem = 0.0 if tie else float(rng.uniform(0, step))
        n = O.blksize_for(L, rem, step)
        dsc = dict(channel=k % period, n=n, s0=int(rng.integers(0, n_if - n)), rem=rem, step=step, d=cfg["d"],
                   f=float(rng.uniform(-2.5e4, 2.5e4)), phi=float(rng.uniform(-3, 3)))
        descs.append(dsc)
        _block(b, k, **dsc)
    return b, descs, tabs, iq, cfg


@pytest.mark.parametrize("case", ["e1_cboc", "e1_cboc_qi", "b1c_wb"])
def test_hybrid_cboc_kernel_equals_the_lane_kernel_and_the_oracle(engine, case):
    """corr_cboc.hip: periodic replay lists of three-arm channels whose third arm is the six-fold replica of the second (Galileo E1-C
    CBOC as BASELINE config 3 words it; BDS/B1C/include/WB_tracking.m:285-317,338-369): the BOC(1,1) arms through the transition
    formulation, the BOC(6,1) arm as a per-sample sign on the carrier-wiped samples.  The kernel takes such lists from two rounds of
    sixteen epochs per CU, two thirds full, on (gc_cboc_takes).  The same list through the lane kernel's derived-arm instantiation (gc_force_generic_kernel) and, block by
    block, through the float64 oracle (every index from ceil(t) / ceil(6 t)); blocks that start on exact chip edges with the nominal
    rational step (tie-dense: sample 0 sits on an edge of all three tables of the prompt tap) included."""
    b, descs, tabs, iq, cfg = _hybrid_list(engine, case)
    fs, L, R, period, nb = cfg["fs"], float(cfg["L"]), 2.0, 3, len(descs)
    engine.replay_prepare(b)
    engine.replay_launch()
    got = engine.replay_fetch().copy()
    assert engine.last_kernel() == 5, engine.last_kernel()
    engine.force_generic_kernel(True)
    try:
        engine.replay_launch()
        other = engine.replay_fetch().copy()
        assert engine.last_kernel() == 0
    finally:
        engine.force_generic_kernel(False)
    raw_all = iq.astype(np.float64)
    scale = np.array([np.sum(np.abs(raw_all[2 * d_["s0"]:2 * (d_["s0"] + d_["n"])])) for d_ in descs])
    dev = np.max(np.abs(got - other).reshape(nb, -1), axis=1) / scale
    assert dev.max() < 2 * TOL, (case, int(np.argmax(dev)), dev.max())
    checked = list(range(period)) + [97, 194, nb - 1, nb // 2, nb // 3]
    for k in checked:
        d_ = descs[k]
        raw = O.raw_from_if(iq, d_["s0"], d_["n"], swap_iq=cfg["layout"] == "qi")
        ref, _, _ = O.correlate_block(raw, tabs[d_["channel"]], d_["rem"], d_["step"], d_["d"], d_["f"], d_["phi"], fs, L, r=R,
                                      arm_mult=[1.0, 1.0, 6.0])
        assert np.max(np.abs(got[k, :3] - ref)) < TOL * scale[k], (case, k, np.max(np.abs(got[k, :3] - ref)) / scale[k])
    # 15 epochs per CU: one round at 16 waves per CU (as long as its slowest wave), 1.25 rounds at 12 (under two thirds full) - lane kernel
    one = engine.make_blocks(period * 5 * (cus := engine.device_info()[1]))
    for k in range(period * 5 * cus):
        _block(one, k, **descs[k])
    engine.replay_prepare(one)
    engine.replay_launch()
    engine.replay_fetch()
    assert engine.last_kernel() == 0
    # a list too short for the kernel (a few blocks: its one round nearly empty) stays on the lane kernel
    few = engine.make_blocks(2 * period)
    for k in range(2 * period):
        _block(few, k, **descs[k])
    engine.replay_prepare(few)
    engine.replay_launch()
    short = engine.replay_fetch().copy()
    assert engine.last_kernel() == 0
    assert np.max(np.abs(short - got[:2 * period]).reshape(2 * period, -1), axis=1).max() < 2 * TOL * scale[:2 * period].max()


@pytest.mark.tuning
@pytest.mark.parametrize("case", ["e1_cboc", "b1c_wb"])
def test_hybrid_cboc_kernel_without_the_host_tie_marks_and_switched_off(engine, monkeypatch, case):
    """The whole list once more with no block marked tie-free (GC_NO_TIE_MARK: every chunk takes the in-kernel float position test
    and the integer test of the six-fold edges), and with GC_NO_CBOC=1: the lane kernel, prepared with its own narrow tie band."""
    b, descs, tabs, iq, cfg = _hybrid_list(engine, case)
    nb = len(descs)
    engine.replay_prepare(b)
    engine.replay_launch()
    got = engine.replay_fetch().copy()
    assert engine.last_kernel() == 5
    raw_all = iq.astype(np.float64)
    scale = np.array([np.sum(np.abs(raw_all[2 * d_["s0"]:2 * (d_["s0"] + d_["n"])])) for d_ in descs])
    monkeypatch.setenv("GC_NO_TIE_MARK", "1")
    engine.replay_prepare(b)
    engine.replay_launch()
    unmarked = engine.replay_fetch().copy()
    assert engine.last_kernel() == 5
    dev = np.max(np.abs(got - unmarked).reshape(nb, -1), axis=1) / scale
    assert dev.max() < 2 * TOL, (case, int(np.argmax(dev)), dev.max())
    monkeypatch.delenv("GC_NO_TIE_MARK")
    monkeypatch.setenv("GC_NO_CBOC", "1")
    engine.replay_prepare(b)
    engine.replay_launch()
    lane = engine.replay_fetch().copy()
    assert engine.last_kernel() == 0
    dev = np.max(np.abs(got - lane).reshape(nb, -1), axis=1) / scale
    assert dev.max() < 2 * TOL, (case, int(np.argmax(dev)), dev.max())
