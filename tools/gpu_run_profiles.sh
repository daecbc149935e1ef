#!/bin/bash
# round-2 profile pass: bench lines of every workload + rocprofv3 kernel traces + MFMA counters + HBM traffic of the headline kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/p
rm -rf $O; mkdir -p $O
# HBM traffic of the headline kernel first: bench.py reports it from profiles/ (only while the kernel sources match its digest)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > /dev/null 2> $O/pmc_f.err; echo "pmc fetch rc=$?" | tee -a $O/summary.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > /dev/null 2> $O/pmc_w.err; echo "pmc write rc=$?" | tee -a $O/summary.txt
python3 tools/pmc_traffic.py --fetch $O/pmc_f/p_counter_collection.csv --write $O/pmc_w/p_counter_collection.csv --workload mppi_cfg2 --samples buffer --commit "$1" --out $O/r02_traffic_mppi_cfg2_buffer.json; echo "traffic rc=$?" | tee -a $O/summary.txt
cp $O/r02_traffic_mppi_cfg2_buffer.json profiles/r02_traffic_mppi_cfg2_buffer.json
python bench.py --steps 200 --warmup 20 > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large-n > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" | tee -a $O/summary.txt
python bench.py --steps 200 --warmup 20 --samples device-rng --no-cpu-baseline --no-large-n --no-modes > $O/bench_200_rng.json 2>/dev/null; echo "bench200 rng rc=$?" | tee -a $O/summary.txt
for wl in mppi_cfg2_interp cem_cfg3 rpgd_cfg4 mppi_cfg5_shard mppi_cfg5 mppi_mlp mppi_gru mppi_default cem_default rpgd_default random_default; do
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-modes > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?" | tee -a $O/summary.txt
done
for wl in mppi_cfg5 cem_cfg3 rpgd_cfg4; do
  CTK_BENCH_SINGLE_DEVICE=1 CTK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 2 --workload $wl --steps 50 --warmup 5 > $O/bench_g2_$wl.json 2> $O/bench_g2_$wl.err; echo "bench g2 $wl rc=$?" | tee -a $O/summary.txt
done
CTK_BENCH_FORCE_PG=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29872 bench.py --gpus 1 --workload mppi_cfg5 --steps 50 --warmup 5 > $O/bench_rccl1_cfg5.json 2> $O/bench_rccl1_cfg5.err; echo "bench rccl1 rc=$?" | tee -a $O/summary.txt
for wl in mppi_cfg2 rpgd_cfg4 mppi_cfg5_shard cem_cfg3 mppi_gru; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-large-n --no-modes > $O/prof_$wl.json 2> $O/prof_$wl.err; echo "prof $wl rc=$?" | tee -a $O/summary.txt
done
for wl in rpgd_cfg4 mppi_cfg5_shard mppi_cfg5; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU --output-format csv -d $O/pmc_mfma_$wl -o p -- python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-large-n --no-modes > /dev/null 2> $O/pmc_mfma_$wl.err; echo "pmc mfma $wl rc=$?" | tee -a $O/summary.txt
done
find $O -name "*.csv" -size +1M -delete
