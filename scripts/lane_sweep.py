import os, sys, subprocess, json, glob
for lib in ["cu-sdr-collection_amd/lib/libgnsscorr.so"] + sorted(glob.glob("cu-sdr-collection_amd/lib/libgnsscorr_*.so")):
    env = dict(os.environ, GC_LIB_PATH=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "scripts/bench_variants.py", "--l5only"] + sys.argv[1:], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(os.path.basename(lib), d["replay_ms"], d["algorithmic_GBps"], d["closed_loop_us_per_epoch"], d["replay_vs_closed_loop_max_dev"], flush=True)
    except Exception as e:
        print(os.path.basename(lib), "FAILED", out.stderr[-300:])
