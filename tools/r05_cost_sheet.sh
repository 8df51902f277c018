#!/bin/bash
# Round 5, verdict item 1 (one kernel per conv2_x bottleneck): the COST side, measured with what exists.
#   gpurun -- 'bash tools/r05_cost_sheet.sh > gpurun_out/r05_conv2x_cost_sheet_measured.txt 2>&1'
# Needs tools/_ab/libmeasure.so (MM_EXTRA_HIPCC_FLAGS=-DMM_MEASURE build: MM_WF_ABLATE / MM_WF_LDS_PAD, results wrong by construction).
cd $GRAFT_REPO_ROOT
export MM_LIB_PATH=$PWD/tools/_ab/libmeasure.so   # (never copied over the shipped library)
PAT="inc256|incproj256|K=256 N=64 k1|wino_in6|maxpool"
for rep in 1 2; do
for cfg in "" "MM_WF_ABLATE=1" "MM_WF_ABLATE=3" "MM_WF_LDS_PAD=28000" "MM_WF_LDS_PAD=28000 MM_WF_ABLATE=3"; do
  echo "== rep $rep: fused 3x3 + increase kernels with [$cfg]  (ABLATE 1 = V rows from an L2-resident subset, 2 = residual rows too; LDS_PAD 28000 = one workgroup per CU)"
  env $cfg python tools/layer_table.py 32 1 2>&1 | grep -E "$PAT"
done; done
echo "== the 256 -> 64 reduce conv on 1.34x the rows (7 x 2 tile patches with their 1-pixel halo: 30 x 10 pixels per 28 x 8)"
for rep in 1 2; do
python tools/conv_bench.py 2048 56 56 256 64 1 1 0
python tools/conv_bench.py 2048 56 75 256 64 1 1 0
done
