#!/bin/bash
# Round 6, verdict item 3(b): phase_window2_kernel split along time over two workgroups per (window, band) at 96 registers (two nine-wave workgroups
# per CU) -- same-box A/B of the phase stage (tools/phase_stage_bench.py: wall time of the stage and the hipEvent times of its kernels).
#   gpurun -- 'bash tools/r06_pw_ab.sh > gpurun_out/r06_ab_phase_window_split2.txt 2>&1'
#   shipped          the one-workgroup kernel (round 3-5 form, 162 registers)
#   MM_PW_SPLIT=2    phase_window2s_kernel: two workgroups per (window, band), plane loads in chunks of 4 frames
#   pwc2             ... in chunks of 2 frames (tools/_ab/libpwc2.so: -DMM_PW_CHUNK=2)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== rep $rep [shipped]";        python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  echo "== rep $rep [MM_PW_SPLIT=2]";  MM_PW_SPLIT=2 python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  for v in tools/_ab/libpw*.so; do
    [ -f $v ] || continue
    echo "== rep $rep [MM_PW_SPLIT=2 $(basename $v .so)]"; MM_PW_SPLIT=2 MM_LIB_PATH=$PWD/$v python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  done
done
