#!/bin/bash
# PMC counters of the phase-stage kernels (name substring $1, default pyramid_wave_kernel) over tools/phase_stage_bench.py at $2 clips:
# matrix-pipe busy, clock, stall split, instruction mix, LDS behaviour -- three separate counter passes (per-launch averages).
# usage: gpurun -- 'bash tools/pmc_phase.sh [pattern] [clips]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=${1:-pyramid_wave_kernel}; CL=${2:-32}
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pp
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pp -o out -- python $R/tools/phase_stage_bench.py $CL > /tmp/pp.log 2>&1
  f=$(find /tmp/pp -name "*counter_collection.csv" | head -1)
  t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
  python - "$f" "$t" "$PAT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
dur = collections.defaultdict(float); nd = collections.Counter()
pat = sys.argv[3]
for r in csv.DictReader(open(sys.argv[2])):
    if pat in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-48:]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; nd[k] += 1
for r in csv.DictReader(open(sys.argv[1])):
    if pat not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-48:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    print("%s: %.1f us/launch (profiled) |" % (k, dur[k] / max(nd[k], 1)), " ".join("%s=%.5g" % (c, v / n[(k, c)]) for c, v in sorted(d.items())))
PY
done
