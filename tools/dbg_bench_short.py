import sys, os, json, subprocess
for i in range(4):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-large-n", "--no-modes"], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1]); print("20 :", round(d["ms_per_step"]*1e3,2), round(d["step_ms_median"]*1e3,2))
r = subprocess.run([sys.executable, "bench.py", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-large-n", "--no-modes"], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1]); print("200:", round(d["ms_per_step"]*1e3,2), round(d["step_ms_median"]*1e3,2), round(d.get("step_ms_p95",0)*1e3,2))
