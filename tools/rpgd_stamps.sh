#!/bin/bash
# in-kernel wall-clock stamps of ctk_rpgd_mlp_wide (variant library built with -DCTK_DIAG_WIDE_STAMPS into tools/_variants/),
# loaded through CTK_HIP_LIBRARY: the product library is never overwritten
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/split
CTK_HIP_LIBRARY=$PWD/tools/_variants/libctk_hip_WIDE_STAMPS.so python bench.py --workload rpgd_cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-large-n 2>&1 | grep "wide stamps" | tail -5 | tee gpurun_out/split/stamps.txt
