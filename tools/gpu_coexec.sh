#!/bin/bash
# round 3: MFMA / VALU co-execution microbenchmark (tools/diag_mfma_coexec.hip) with its counters, own rocprofv3 pass
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/coexec; rm -rf $O; mkdir -p $O
./tools/diag_mfma_coexec 0 > $O/coexec.txt 2>&1; echo "coexec rc=$?" | tee -a $O/summary.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc -o p -- ./tools/diag_mfma_coexec 0 > $O/pmc.txt 2> $O/pmc.err; echo "pmc rc=$?" | tee -a $O/summary.txt
python3 - <<'PY' > gpurun_out/coexec/pmc_summary.txt 2>&1
import csv, collections, glob
rows = collections.OrderedDict()
for f in glob.glob("gpurun_out/coexec/pmc/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r["Grid_Size"])
        rows.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | grid | dispatches | per-dispatch means of the counters")
for (k, g), c in rows.items():
    n = max(len(v) for v in c.values())
    print(k[:60], g, n, {name: round(sum(v) / len(v)) for name, v in sorted(c.items())})
PY
find $O -name "*.csv" -size +1M -delete
