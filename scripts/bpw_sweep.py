import os, sys, subprocess, json
for b in (1, 2, 4, 8, 16):
    env = dict(os.environ, GC_REPLAY_BPW=str(b))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--no-cpu"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(b, d["roofline"]["kernel_ms"], d["roofline"]["achieved"])
