import os, sys, time, subprocess, json
for s in (2, 4, 6, 8, 12, 17, 24):
    env = dict(os.environ, GC_TRACK_SPLITS=str(s))
    out = subprocess.run([sys.executable, "bench.py", "--seconds", "8", "--steps", "3", "--no-cpu"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(s, d["closed_loop"], d["roofline"]["kernel_ms"])
