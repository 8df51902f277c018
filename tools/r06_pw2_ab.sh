#!/bin/bash
# Round 6, second pass over the window kernels: same-box A/B of the phase stage (tools/phase_stage_bench.py) for library variants under
# tools/_ab/libpw*.so (built beforehand with MM_EXTRA_HIPCC_FLAGS, loaded through MM_LIB_PATH) against the shipped library.
#   gpurun -- 'bash tools/r06_pw2_ab.sh > gpurun_out/r06_ab_phase_window_prefetch.txt 2>&1'
#   libpwnopre.so   -DMM_PW_PREFETCH=0: every round of the blur loop requests its magnitude planes itself (the round 3-5 form)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== rep $rep [shipped]";        python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  for v in tools/_ab/libpw*.so; do
    [ -f $v ] || continue
    echo "== rep $rep [$(basename $v .so)]"; MM_LIB_PATH=$PWD/$v python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  done
done
