#!/bin/bash
# Matrix-core utilisation of the conv engine over a bench run (rocprofv3 PMC, counters only + kernel trace).
# MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CL=${1:-32}
rm -rf /tmp/pm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES \
   --kernel-trace --output-format csv -d /tmp/pm -o out -- python $R/bench.py --steps 2 --warmup 1 --clips $CL --lanes 1 --no-cpu-baseline > /tmp/pm.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv_mfma_kernel" not in k: continue
    k = k.split("(")[0].replace("void mm::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
tot = collections.defaultdict(float)
print("%-46s %8s %10s %8s %8s %8s" % ("kernel", "launches", "MFMA busy", "issueStl", "waitcnt", "active"))
for k, d in sorted(agg.items()):
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)
    wc = d["SQ_WAVE_CYCLES"]
    print("%-46s %8d %9.1f%% %7.1f%% %7.1f%% %7.1f%%" % (k, n[(k, "GRBM_GUI_ACTIVE")], 100 * busy, 100 * d["SQ_WAIT_INST_ANY"] / wc,
          100 * d["SQ_WAIT_ANY"] / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / wc))
    for c, v in d.items(): tot[c] += v
print("ALL conv launches: MFMA busy %.1f%% of SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES %.4g, GRBM_GUI_ACTIVE %.4g summed over 8 XCDs)" %
      (100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * tot["GRBM_GUI_ACTIVE"] / 8.0), tot["SQ_VALU_MFMA_BUSY_CYCLES"], tot["GRBM_GUI_ACTIVE"]))
PY
