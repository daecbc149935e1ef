#!/bin/bash
# final pass of a round: full -m gpu suite, the profile pass, the microbenchmarks and side tables that DESIGN.md quotes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/final
F=gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -q > $F/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $F/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_run_profiles.sh "$1" > $F/profiles.log 2>&1; tail -3 $F/profiles.log
for d in diag_mlp_step diag_mlp_l3 diag_mlp_pair diag_pk_rate; do echo "## tools/$d" >> $F/microbench.txt; timeout -k 10 120 ./tools/$d >> $F/microbench.txt 2>&1; done
timeout -k 10 420 python tools/bench_env.py --steps 300 > $F/env_kernels.txt 2>&1; echo "bench_env rc=$?"
timeout -k 10 300 python tools/sweep_n.py > $F/sweep_n.txt 2>&1; echo "sweep rc=$?"
bash tools/rpgd_split.sh > /dev/null 2>&1; cp gpurun_out/split/split.txt $F/rpgd_wide_vs_narrow.txt 2>/dev/null
CTK_RPGD_NARROW=1 python bench.py --workload rpgd_cfg4 --steps 60 --warmup 10 --no-cpu-baseline --no-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('single-launch form (CTK_RPGD_NARROW=1): ms_per_step', round(d['ms_per_step'],4), 'kernel_us', round(d['roofline']['kernel_us'],1))" >> $F/rpgd_wide_vs_narrow.txt
CTK_MPPI_NO_PAIR=1 python bench.py --workload mppi_cfg5_shard --steps 100 --warmup 10 --no-cpu-baseline --no-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mppi_cfg5_shard, one wave per tile (CTK_MPPI_NO_PAIR=1): ms_per_step', round(d['ms_per_step'],4), 'kernel_us', round(d['roofline']['kernel_us'],1), 'frac', round(d['roofline']['frac'],4))" > $F/mppi_pair_vs_single.txt
echo done
