#!/bin/bash
# Round 6, third pass over the window kernels: LDS-only barriers and the pair kernel.  Same-box A/B of the phase stage (tools/phase_stage_bench.py,
# 32 and 256 clips, three repetitions): shipped library and tools/_ab/libpw*.so variants, each with the pair kernel (MM_PW_PAIR=1) and with one launch
# per level (MM_PW_PAIR=0, the default).   gpurun -- 'bash tools/r06_pw3_ab.sh > gpurun_out/r06_ab_phase_window_barrier.txt 2>&1'
#   libpwsync.so   __syncthreads() everywhere (rounds 3-5; in the log's run the shipped library had the LDS-only barriers, -DMM_PW_LDS_BARRIER=1)
#   libpwskew<N>.so  measurement: the first workgroup of every second CU starts N x 3.4 us late
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for pair in 1 0; do
    echo "== rep $rep MM_PW_PAIR=$pair [shipped]"; MM_PW_PAIR=$pair python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
    for v in tools/_ab/libpw*.so; do
      [ -f $v ] || continue
      echo "== rep $rep MM_PW_PAIR=$pair [$(basename $v .so)]"; MM_PW_PAIR=$pair MM_LIB_PATH=$PWD/$v python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
    done
  done
done
