import os, sys, subprocess, json
for w in (3, 4, 5, 6):
    env = dict(os.environ, GC_LIB_PATH=os.path.abspath(f"cu-sdr-collection_amd/lib/libgnsscorr_w{w}.so"))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--no-cpu"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(w, d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["closed_loop"]["us_per_epoch"])
