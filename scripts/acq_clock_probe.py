"""Is the default GPS L1 C/A search clock- / power-limited?  Runs the search back to back for a few seconds while a thread samples
the device's shader clock and socket power from sysfs (hwmon freq1_input / power1_average, pp_dpm_sclk as a fall-back), and prints
the per-call times next to the samples taken while that call ran.  usage: python scripts/acq_clock_probe.py [seconds]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cu_sdr_collection_amd as P


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def sensors():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if _read(os.path.join(card, "vendor")) != "0x1002":
            continue
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("freq1_input", "power1_average", "power1_input", "power1_cap", "temp1_input", "temp2_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    out[os.path.basename(card.rstrip("/device")) + ":" + name] = p
        p = os.path.join(card, "pp_dpm_sclk")
        if os.path.exists(p):
            out["dpm_sclk"] = p
        break
    return out


def sample(paths):
    row = {}
    for k, p in paths.items():
        v = _read(p)
        if v is None:
            continue
        if k == "dpm_sclk":
            cur = [l for l in v.split("\n") if l.endswith("*")]
            row[k] = cur[0] if cur else v.replace("\n", " | ")
        else:
            try:
                row[k.split(":")[1]] = int(v)
            except ValueError:
                row[k.split(":")[1]] = v
    return row


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    paths = sensors()
    print("sensors:", paths)
    S = P.initSettings()
    sats = P.synth.scene(12, 5, S.samplingFreq)
    iq = P.synth.generate_if(sats, int(0.1 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
    eng = P.Engine(0)
    eng.load_if(iq, fs=S.samplingFreq)
    P.acquisition(eng, S)
    log, stop = [], False

    def watcher():
        while not stop:
            log.append((time.perf_counter(), sample(paths)))
            time.sleep(0.002)

    th = threading.Thread(target=watcher, daemon=True)
    time.sleep(0.5)          # idle clocks first
    th.start()
    time.sleep(0.1)
    calls = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter(); P.acquisition(eng, S); calls.append((t0, time.perf_counter()))
    stop = True
    th.join()
    ms = np.array([(b - a) * 1e3 for a, b in calls])
    print("calls: %d, first five %s ms, median %.3f, last five %s" % (len(ms), np.round(ms[:5], 3), np.median(ms), np.round(ms[-5:], 3)))
    keys = sorted({k for _, r in log for k in r if k != "dpm_sclk"})
    t0 = calls[0][0]
    for lo, hi in ((-0.1, 0.0), (0.0, 0.02), (0.02, 0.1), (0.1, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, seconds)):
        rows = [r for t, r in log if lo <= t - t0 < hi]
        cm = ms[[(lo <= a - t0 < hi) for a, _ in calls]] if hi > 0 else np.array([])
        line = "t in [%5.2f, %5.2f) s: %3d samples" % (lo, hi, len(rows))
        for k in keys:
            v = [r[k] for r in rows if isinstance(r.get(k), int)]
            if v:
                line += "  %s mean %.4g (min %.4g max %.4g)" % (k, np.mean(v), min(v), max(v))
        if len(cm):
            line += "  search %.3f ms" % np.median(cm)
        d = [r["dpm_sclk"] for r in rows if "dpm_sclk" in r]
        if d:
            line += "  dpm " + d[-1]
        print(line)


if __name__ == "__main__":
    main()
