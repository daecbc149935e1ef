#!/bin/bash
# MLP tests + MLP MPPI workloads with the pair form on and off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/h; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "mlp or MLP or golden or shard or serving" > $O/test.log 2>&1; echo "test rc=$?"; tail -4 $O/test.log
for wl in mppi_cfg5_shard mppi_mlp; do
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_$wl.json 2> $O/bench_$wl.err
  CTK_MPPI_NO_PAIR=1 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_${wl}_nopair.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/h/bench_*.json")):
    d = json.load(open(f)); r = d["roofline"]; print(f, round(d["ms_per_step"], 4), r.get("kernel"), round(r.get("kernel_us", 0), 2), r.get("frac"))
PY
