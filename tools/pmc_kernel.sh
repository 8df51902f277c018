#!/bin/bash
# PMC counters of one kernel (name substring $1) over a single-stream step: matrix-pipe busy fraction, effective clock, wave
# stall breakdown.  usage: gpurun -- 'bash tools/pmc_kernel.sh wino_fused_kernel [clips] [winograd mode]'   (PK_DRIVER=<script>: another
# driver program than tools/layer_table.py, e.g. tools/probes/stem_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=${1:-conv_mfma_kernel}; CL=${2:-32}; MODE=${3:-1}
rm -rf /tmp/pk
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES \
   --kernel-trace --output-format csv -d /tmp/pk -o out -- python ${PK_DRIVER:-$R/tools/layer_table.py} $CL $MODE > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*counter_collection.csv" | head -1)
t=$(find /tmp/pk -name "*kernel_trace.csv" | head -1)
python - "$f" "$t" "$PAT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
pat = sys.argv[3]
dur = collections.defaultdict(float); nd = collections.Counter()
for r in csv.DictReader(open(sys.argv[2])):
    if pat in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; nd[k] += 1
for r in csv.DictReader(open(sys.argv[1])):
    if pat not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    L = n[(k, "GRBM_GUI_ACTIVE")]
    gui = d["GRBM_GUI_ACTIVE"] / L            # summed over 8 XCDs per launch
    us = dur[k] / max(nd[k], 1)
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)
    wc = d["SQ_WAVE_CYCLES"]
    print("%s grid %s x%d: %.1f us/launch, clock %.2f GHz, MFMA busy %.1f%%, MFMA insts/launch %.3g, issue-stall %.1f%% waitcnt/barrier %.1f%% issuing %.1f%%" %
          (k[0], k[1], L, us, gui / 8.0 / us / 1e3 if us else 0, 100 * busy, d["SQ_INSTS_MFMA"] / L, 100 * d["SQ_WAIT_INST_ANY"] / wc,
           100 * d["SQ_WAIT_ANY"] / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / wc))
PY
