#!/bin/bash
# PMC passes over the phase stage (pyramid + window kernels) of a bench run; prints per-kernel per-launch averages.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CL=${1:-8}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pp_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pp_$tag -o out -- python $R/bench.py --steps 1 --warmup 1 --clips $CL --no-cpu-baseline > /tmp/pp_$tag.log 2>&1
  f=$(find /tmp/pp_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "pyramid_kernel" not in k and "phase_window" not in k: continue
    k = k.split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, " ".join("%s=%.4g" % (c, v / cnt[(k, c)]) for c, v in d.items()))
PY
done
