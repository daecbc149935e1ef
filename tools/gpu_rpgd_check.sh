#!/bin/bash
# RPGD-family tests + cfg4 bench (wide and single-launch forms) + kernel trace of the wide form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/g; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "rpgd or gradient or grad or golden or mlp" > $O/test.log 2>&1; echo "test rc=$?"; tail -4 $O/test.log
python bench.py --workload rpgd_cfg4 --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_wide.json 2> $O/bench_wide.err
CTK_RPGD_NARROW=1 python bench.py --workload rpgd_cfg4 --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_narrow.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/g/bench_wide.json", "gpurun_out/g/bench_narrow.json"):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 4), d["roofline"].get("kernel"), d["roofline"].get("kernel_us"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --workload rpgd_cfg4 --steps 30 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > $O/prof.json 2> $O/prof.err
cut -c1-60,200-300 $O/prof/p_kernel_stats.csv | head -6
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/g/prof/p_kernel_stats.csv")))
for r in rows[:5]:
    print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $O/prof -name "*.csv" -size +1M -delete
