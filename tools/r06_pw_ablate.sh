#!/bin/bash
# Round 6: where the window kernels' time goes -- measurement builds (results wrong by construction) under tools/_ab/libpwabl<N>.so, built with
# MM_EXTRA_HIPCC_FLAGS=-DMM_PW_ABLATE=<N> (1 no output stores, 2 no blur rounds, 4 every window reads window 0's frames: all loads L2 hits),
# each with the pair kernel (MM_PW_PAIR=1) and with one launch per level (MM_PW_PAIR=0, the default).
#   gpurun -- 'bash tools/r06_pw_ablate.sh > gpurun_out/r06_phase_window_ablation.txt 2>&1'
cd $GRAFT_REPO_ROOT
for pair in 1 0; do
  echo "== MM_PW_PAIR=$pair [shipped]"; MM_PW_PAIR=$pair python tools/phase_stage_bench.py 32 2>&1 | grep clips
  for v in tools/_ab/libpwabl*.so; do
    echo "== MM_PW_PAIR=$pair [$(basename $v .so)]"; MM_PW_PAIR=$pair MM_LIB_PATH=$PWD/$v python tools/phase_stage_bench.py 32 2>&1 | grep clips
  done
done
