#!/bin/bash
# in-kernel wall-clock stamps of the one-launch RPGD descent (ctk_rpgd_mlp_persistent): one producer's iteration 7 and the workers of its
# tile 0.  Variant library, never the product one: from control_toolkit_amd/csrc,
#   make BUILD=../../tools/_variants/b_pers LIB=../../tools/_variants/libctk_hip_PERS_STAMPS.so EXTRA=-DCTK_DIAG_PERS_STAMPS
# (tools/_variants/ is git-ignored and travels to the GPU box), loaded through CTK_HIP_LIBRARY.  The printf calls perturb the run they time
# (a producer line with a six-digit "flags seen" waited for a worker's printf): read the lines that agree.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/split
CTK_HIP_LIBRARY=$PWD/tools/_variants/libctk_hip_PERS_STAMPS.so python bench.py --workload rpgd_cfg4 --steps 4 --warmup 2 --no-cpu-baseline --no-modes --no-large-n 2>&1 | grep "pers " | tail -14 | cut -c1-220 | tee gpurun_out/split/stamps.txt
