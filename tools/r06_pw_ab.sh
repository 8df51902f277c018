#!/bin/bash
# Round 6: same-box A/B of the phase stage (tools/phase_stage_bench.py: wall time of the stage and the hipEvent times of its kernels).
#   gpurun -- 'bash tools/r06_pw_ab.sh > gpurun_out/r06_ab_phase_window_micro.txt 2>&1'
#   shipped          the one-workgroup window kernel with the DPP wave reduction and the halo-only LDS clear
#   pwnodpp          ... with the __shfl_down (ds_bpermute) reduction (-DMM_PW_DPP_REDUCE=0)
#   pwold            ... and the whole working region cleared (-DMM_PW_CLEAR_ALL=1): the round 3-5 kernel
#   MM_PW_SPLIT=2    phase_window2s_kernel: two workgroups per (window, band), split along time (verdict item 3b; first log: r06_ab_phase_window_split.txt)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== rep $rep [shipped]";        python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  for v in tools/_ab/libpw*.so; do
    [ -f $v ] || continue
    echo "== rep $rep [$(basename $v .so)]"; MM_LIB_PATH=$PWD/$v python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  done
  echo "== rep $rep [MM_PW_SPLIT=2]";  MM_PW_SPLIT=2 python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
done
