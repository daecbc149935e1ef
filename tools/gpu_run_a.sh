#!/bin/bash
# round-2 GPU pass A: tests, bench (200 / 20 steps), sharded rehearsals, rocprof of the MFMA kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/a
mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/gputests.log
python bench.py --steps 200 --warmup 20 > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large-n > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" | tee -a $O/summary.txt
for wl in cem_cfg3 rpgd_cfg4 mppi_cfg5_shard mppi_mlp mppi_gru; do
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?" | tee -a $O/summary.txt
done
# sharded rehearsal: 2 ranks on the one GPU over gloo (RCCL refuses two ranks per device), cfg5 workload, then CEM and RPGD shards
for wl in mppi_cfg5 cem_cfg3 rpgd_cfg4; do
  CTK_BENCH_SINGLE_DEVICE=1 CTK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 2 --workload $wl --steps 50 --warmup 5 > $O/bench_g2_$wl.json 2> $O/bench_g2_$wl.err; echo "bench g2 $wl rc=$?" | tee -a $O/summary.txt
done
# one-rank RCCL group with the collective issued (RCCL call path)
CTK_BENCH_FORCE_PG=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29872 bench.py --gpus 1 --workload mppi_cfg5 --steps 50 --warmup 5 > $O/bench_rccl1_cfg5.json 2> $O/bench_rccl1_cfg5.err; echo "bench rccl1 rc=$?" | tee -a $O/summary.txt
# kernel traces of the MFMA kernels + CEM trio
for wl in rpgd_cfg4 mppi_cfg5_shard cem_cfg3 mppi_cfg2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-large-n --no-modes > $O/prof_$wl.json 2> $O/prof_$wl.err; echo "prof $wl rc=$?" | tee -a $O/summary.txt
done
for wl in rpgd_cfg4 mppi_cfg5_shard; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$wl -o p -- python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > /dev/null 2> $O/pmc_mfma_$wl.err; echo "pmc mfma $wl rc=$?" | tee -a $O/summary.txt
done
# HBM traffic of the headline kernel (separate passes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > /dev/null 2> $O/pmc_f.err; echo "pmc fetch rc=$?" | tee -a $O/summary.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > /dev/null 2> $O/pmc_w.err; echo "pmc write rc=$?" | tee -a $O/summary.txt
python3 tools/pmc_traffic.py --fetch $(ls $O/pmc_f/*/*counter_collection.csv $O/pmc_f/*counter_collection.csv 2>/dev/null | head -1) --write $(ls $O/pmc_w/*/*counter_collection.csv $O/pmc_w/*counter_collection.csv 2>/dev/null | head -1) --workload mppi_cfg2 --samples buffer --commit "$1" --out $O/r02_traffic_mppi_cfg2_buffer.json; echo "traffic rc=$?" | tee -a $O/summary.txt
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
