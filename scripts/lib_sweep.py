import os, sys, subprocess, json, glob
for lib in sorted(glob.glob("cu-sdr-collection_amd/lib/libgnsscorr_*.so")):
    env = dict(os.environ, GC_LIB_PATH=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--no-cpu"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(os.path.basename(lib), d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["closed_loop"]["us_per_epoch"], d["replay_vs_closed_loop_max_dev"])
