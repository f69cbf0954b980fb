import os as _os
_os.environ.setdefault("GC_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cu-sdr-collection_amd", "lib", "libgnsscorr_tuning.so"))  # the GC_* switches used below exist in the tuning build only (docs/KNOBS.md)
import os, sys, subprocess, json
for waves in (2, 4, 8, 16):
    for members in (1, 2, 4, 6):
        env = dict(os.environ, GC_DEVLOOP_WAVES=str(waves), GC_DEVLOOP_MEMBERS=str(members))
        out = subprocess.run([sys.executable, "scripts/bench_variants.py"] + sys.argv[1:], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        for l in out:
            try:
                d = json.loads(l)
            except Exception:
                continue
            print("waves", waves, "members", members, d["shape"][:12], "device", d["device_loop_us_per_epoch"], "host", d["closed_loop_us_per_epoch"], flush=True)
