#!/bin/bash
# profile pass (ONE gpurun call; round 4: R=04, outputs under gpurun_out/p4): HBM traffic of the headline kernel, bench lines of every workload, two-rank rehearsals of the
# sharded bench, rocprofv3 kernel traces, MFMA counters, the co-execution microbenchmark, the environment table, the CEM phase stamps.
# Everything is CONDENSED ON THE BOX (tools/condense_profiles.py --out $O/condensed) before any size-based delete, and the condenser
# refuses inputs older than this run's start stamp: a summary under profiles/ can only come from the run that produced its CSVs
# (round 2 committed counters of an older kernel because a >1 MB CSV was deleted on the box and an older local copy was condensed).
# usage: gpurun -- bash tools/gpu_run_profiles.sh <commit>;  then copy gpurun_out/p4/condensed/* to profiles/
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=04
O=gpurun_out/p4
rm -rf $O; mkdir -p $O
date +%s > $O/run_started
echo "commit $1" > $O/summary.txt
say() { echo "$1 rc=$2" | tee -a $O/summary.txt; }
B="--no-cpu-baseline --no-large-n --no-modes"
# 0. parity margins of the reference-golden tests (tests/margins.py writes gpurun_out/parity_margins.txt)
timeout -k 10 600 python -m pytest tests -m gpu -q -k "golden or replays_reference" > $O/golden_tests.log 2>&1; say "golden tests" $?
# 1. HBM traffic of the headline kernel first: bench.py reports it from profiles/ (only while the kernel sources match its digest)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 bench.py --steps 50 --warmup 5 $B > /dev/null 2> $O/pmc_f.err; say "pmc fetch" $?
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 bench.py --steps 50 --warmup 5 $B > /dev/null 2> $O/pmc_w.err; say "pmc write" $?
python3 tools/pmc_traffic.py --fetch $O/pmc_f/p_counter_collection.csv --write $O/pmc_w/p_counter_collection.csv --workload mppi_cfg2 --samples buffer --commit "$1" --out $O/r${R}_traffic_mppi_cfg2_buffer.json; say "traffic" $?
cp $O/r${R}_traffic_mppi_cfg2_buffer.json profiles/r${R}_traffic_mppi_cfg2_buffer.json
# 2. bench lines
python bench.py --steps 200 --warmup 20 > $O/bench_200.json 2> $O/bench_200.err; say "bench200" $?
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large-n > $O/bench_20.json 2> $O/bench_20.err; say "bench20" $?
python bench.py --steps 200 --warmup 20 --samples device-rng $B > $O/bench_200_rng.json 2>/dev/null; say "bench200 rng" $?
for wl in mppi_cfg2_interp cem_cfg3 rpgd_cfg4 mppi_cfg5_shard mppi_cfg5 mppi_mlp mppi_gru mppi_default cem_default rpgd_default random_default; do
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_$wl.json 2> $O/bench_$wl.err; say "bench $wl" $?
done
CTK_NO_CEM_FUSED=1 python bench.py --workload cem_cfg3 --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_cem_cfg3_launch_per_phase.json 2>/dev/null; say "bench cem launch-per-phase" $?
# 3. two ranks sharing ONE GPU over gloo: plumbing of the sharded bench (exchange_us, per-rank roofline), not scaling
for wl in mppi_cfg5 cem_cfg3 rpgd_cfg4; do
  CTK_BENCH_SINGLE_DEVICE=1 CTK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 2 --workload $wl --steps 50 --warmup 5 > $O/bench_g2_$wl.json 2> $O/bench_g2_$wl.err; say "bench g2 $wl" $?
done
CTK_BENCH_FORCE_PG=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29872 bench.py --gpus 1 --workload mppi_cfg5 --steps 50 --warmup 5 > $O/bench_rccl1_cfg5.json 2> $O/bench_rccl1_cfg5.err; say "bench rccl1" $?
# one rank's share of configs[4] at G = 8 (N 8 192) with the RCCL collective actually issued: begin | exchange | end of the step the curve will be made of
CTK_BENCH_FORCE_PG=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29873 bench.py --gpus 1 --workload mppi_cfg5_shard --steps 100 --warmup 10 > $O/bench_rccl1_cfg5_shard.json 2> $O/bench_rccl1_cfg5_shard.err; say "bench rccl1 shard" $?
# `python bench.py --gpus 2` as a bare command (starts its own ranks): the shape of the driver's multi-GPU command, rehearsed on one GPU over gloo
CTK_BENCH_SINGLE_DEVICE=1 CTK_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_bare_gpus2.json 2> $O/bench_bare_gpus2.err; say "bench bare --gpus 2" $?
# 4. kernel traces
for wl in mppi_cfg2 rpgd_cfg4 mppi_cfg5_shard cem_cfg3 mppi_gru; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 100 --warmup 10 $B > $O/prof_$wl.json 2> $O/prof_$wl.err; say "prof $wl" $?
done
# 5. MFMA counters (own passes)
for wl in rpgd_cfg4 mppi_cfg5_shard mppi_cfg5; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU --output-format csv -d $O/pmc_mfma_$wl -o p -- python3 bench.py --workload $wl --steps 30 --warmup 5 $B > /dev/null 2> $O/pmc_mfma_$wl.err; say "pmc mfma $wl" $?
done
# 6. co-execution microbenchmark + its counters, environment table, CEM phase stamps, large-N sweep
./tools/diag_mfma_coexec 0 > $O/coexec.txt 2>&1; say "coexec" $?
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_coexec -o p -- ./tools/diag_mfma_coexec 0 > /dev/null 2> $O/pmc_coexec.err; say "pmc coexec" $?
python tools/bench_env.py --steps 300 > $O/env_kernels.txt 2> $O/env_kernels.err; say "bench_env" $?
./tools/diag_cem_fused 4096 30 409 > $O/cem_stamps_cfg3.txt 2>&1; say "cem stamps" $?
./tools/diag_cem_fused 200 40 40 > $O/cem_stamps_default.txt 2>&1
python tools/sweep_n.py > $O/sweep_n.txt 2> $O/sweep_n.err; say "sweep_n" $?
python tools/bench_resident.py > $O/resident.txt 2> $O/resident.err; say "bench_resident" $?
./tools/diag_mailbox_vram >> $O/resident.txt 2>&1; say "mailbox diag" $?
./tools/diag_wave_placement > $O/placement.txt 2>&1; say "wave placement" $?
# the two forms of the RPGD + MLP descent over sizes (the switch is read once per process), and where one iteration of the one-launch form goes
{ echo "== one launch per MPC step"; python tools/rpgd_forms.py 2>&1 | grep "us per"; echo "== phase launches (CTK_RPGD_NO_PERSISTENT=1)"; CTK_RPGD_NO_PERSISTENT=1 python tools/rpgd_forms.py 2>&1 | grep "us per"; } > $O/rpgd_forms.txt; say "rpgd forms" $?
bash tools/rpgd_stamps.sh > /dev/null 2>&1; cp gpurun_out/split/stamps.txt $O/rpgd_pers_stamps.txt 2>/dev/null; say "rpgd stamps" $?
python3 tools/soak_handoff.py 3000 > $O/soak_handoff.txt 2>&1; say "soak hand-off" $?
python tools/soak.py 1500 > $O/soak.txt 2>&1; say "soak" $?
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_largen -o p -- python3 tools/large_n_once.py > /dev/null 2> $O/pmc_largen.err; say "pmc large-N insts" $?
# 7. condense HERE, then drop the big CSVs
cp gpurun_out/parity_margins.txt $O/parity_margins.txt 2>/dev/null
python3 tools/condense_profiles.py --src $O --round 4 --out $O/condensed; say "condense" $?
find $O -name "*.csv" -size +1M -delete
