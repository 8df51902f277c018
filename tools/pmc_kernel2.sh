#!/bin/bash
# Second PMC pass for one kernel: instruction mix and LDS behaviour (per launch averages).
# usage: gpurun -- 'bash tools/pmc_kernel2.sh wino_fused_kernel [clips] [winograd mode]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=${1:-conv_mfma_kernel}; CL=${2:-32}; MODE=${3:-1}
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pk2
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pk2 -o out -- python $R/tools/layer_table.py $CL $MODE > /tmp/pk2.log 2>&1
  f=$(find /tmp/pk2 -name "*counter_collection.csv" | head -1)
  python - "$f" "$PAT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    print(k[0], "grid", k[1], " ".join("%s=%.4g" % (c, v / n[(k, c)]) for c, v in d.items()))
PY
done
