#!/bin/bash
# Round 6, verdict item 1: the NEXT block's 256 -> 64 reduce conv inside the fused conv2_x kernel (wino_fused.hip NEXT), kernel-level same-box A/B.
#   gpurun -- 'bash tools/r06_next_ab.sh > gpurun_out/r06/next_ab.txt 2>&1'
# rows: the fused 3x3 + increase kernels of blocks 2-3 ("inc256", "+red64" = with the third GEMM), the separate 256 -> 64 launches, the step's totals
#   MM_FUSE_NEXT=0                 the round-5 schedule (two four-wave inc256 launches + two 256 -> 64 launches)
#   MM_FUSE_NEXT=0 MM_INC1_SHAPE=8 the eight-wave workgroup shape ALONE (one workgroup per CU, W2 + 16 KB exchange; no third GEMM)
#   MM_FUSE_NEXT=1                 shipped: block 2's kernel also runs block 3's reduce conv (eight waves, W2 + W3 in LDS: 145 KB)
#   measure lib, MM_INC1_SHAPE=2   cost proxy (results wrong): the third GEMM in the FOUR-wave kernel, its fragments read from W2's rows -- what the
#                                  lever would cost with two workgroups per CU kept (cannot be built: W2 + W3 = 128 KB per workgroup)
cd $GRAFT_REPO_ROOT
PAT="inc256|K=256 N=64 k1|totals"
for rep in 1 2 3; do
  for cfg in "MM_FUSE_NEXT=0" "MM_FUSE_NEXT=0 MM_INC1_SHAPE=8" "MM_FUSE_NEXT=1" "MM_FUSE_NEXT=1 MM_INC1_SHAPE=2 MM_LIB_PATH=$PWD/tools/_ab/libmeasure.so"; do
    echo "== rep $rep [$cfg]"
    env $cfg python tools/layer_table.py 32 1 2>&1 | grep -E "$PAT"
  done
done
