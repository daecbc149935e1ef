#!/bin/bash
# timing experiment: ctk_rpgd_descent<1> (cfg4) with one of its sweeps removed (variant libraries built with
# -DCTK_DIAG_RPGD_NO_FWD / -DCTK_DIAG_RPGD_NO_BWD into tools/_variants/; results of those runs are meaningless, only the time counts)
# variant libraries (not product; tools/_variants/ is git-ignored), e.g. for NO_FWD, from control_toolkit_amd/csrc:
#   hipcc $(CXXFLAGS of the Makefile) -DCTK_DIAG_RPGD_NO_FWD -c ctk_rpgd.hip -o /tmp/rpgd_v.o && hipcc -shared -fPIC --offload-arch=gfx950 \
#     -o ../../tools/_variants/libctk_hip_NO_FWD.so _build/ctk_api.o _build/ctk_mppi.o _build/ctk_sampled.o /tmp/rpgd_v.o _build/ctk_generic.o _build/ctk_generic_net.o
# macros: CTK_DIAG_RPGD_NO_FWD / CTK_DIAG_RPGD_NO_BWD (single-launch form), CTK_DIAG_WIDE_NO_CHAIN / CTK_DIAG_WIDE_NO_ADAM / CTK_DIAG_WIDE_STAMPS (wide form)
# The variants are loaded through CTK_HIP_LIBRARY (control_toolkit_amd/_capi.py: library_path); the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/split; mkdir -p $O
run() { python bench.py --workload rpgd_cfg4 --steps 60 --warmup 10 --no-cpu-baseline --no-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_us', d['roofline'].get('kernel_us'))"; }
run full | tee $O/split.txt
for v in "$@"; do CTK_HIP_LIBRARY=$PWD/tools/_variants/libctk_hip_$v.so run $v | tee -a $O/split.txt; done
