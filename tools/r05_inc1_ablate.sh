#!/bin/bash
# Round 5: where the conv2_x fused 3x3 + increase kernel (wino_fused_kernel<3,2,1,2,1>, 56 % matrix-busy) loses its time -- parts switched off
# in a -DMM_MEASURE library (tools/_ab/libmeasure.so; results wrong by construction).  MM_WF_ABLATE bits: 4 no output-transform updates in
# the main loop, 8 no residual loads, 16 no output stores, 32 no exchange barriers, 64 no epilogue MFMAs, 128 no operand DMA in the main loop.
cd $GRAFT_REPO_ROOT
export MM_LIB_PATH=$PWD/tools/_ab/libmeasure.so   # (never copied over the shipped library)
for rep in 1 2; do
for a in 0 4 8 16 24 32 64 128 132 56 188 252; do
  echo -n "ABLATE=$a  "
  MM_WF_ABLATE=$a python tools/layer_table.py 32 1 2>&1 | grep -E "inc256 "
done; done
