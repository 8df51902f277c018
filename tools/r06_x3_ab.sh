#!/bin/bash
# Round 6, verdict item 7 (`extra` only): bf16x3 with the WEIGHTS pre-split into three bf16 planes (MM_X3_PRESPLIT=1, default) against the form that splits
# both operands in the loop (MM_X3_PRESPLIT=0) -- kernel level (rows tagged x3) and the whole step.   gpurun -- 'bash tools/r06_x3_ab.sh > gpurun_out/r06_ab_x3_presplit.txt 2>&1'
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do   # 0 (default): both operands split in the loop; 1: weights pre-split
  echo "== MM_X3_PRESPLIT=$v (rep $rep)"
  MM_X3_PRESPLIT=$v LT_PRECISION=bf16x3 python tools/layer_table.py 32 1 2>/dev/null | grep -E " x3|totals"
done; done
