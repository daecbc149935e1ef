#!/bin/bash
# quick A/B of the MLP pair kernels: cfg5 shard + cfg4 + the MLP tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_rpgd.py -q -x > $O/test.log 2>&1; echo "test rc=$?"; tail -2 $O/test.log
for wl in mppi_cfg5_shard rpgd_cfg4 mppi_mlp; do
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl', round(d['ms_per_step']*1e3,1), 'us/step; kernel', round(r['kernel_us'],1), 'us; frac', round(r['frac'],4))"
done
