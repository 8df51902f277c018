#!/bin/bash
# Which kernel does the vendor fp32 GEMM pick on the engine's shapes, and with what resources (tile, waves, LDS, VGPR/AGPR)?
# Cross-check only: nothing in the product calls a BLAS.  gpurun -- 'bash tools/vendor_kernel_probe.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/vk
rocprofv3 --kernel-trace --output-format csv -d /tmp/vk -o out -- python $R/tools/blas_probe.py > /tmp/vk.log 2>&1
cat /tmp/vk.log | tail -12
f=$(find /tmp/vk -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
seen = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    key = (k, r.get("Workgroup_Size_X"), r.get("Grid_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))
    s = seen.setdefault(key, [0, 0.0]); s[0] += 1; s[1] += d
for key, (n, ms) in seen.items():
    if ms / n < 0.05: continue
    print("%6d x %8.3f ms  wg=%s grid=%s lds=%s vgpr=%s agpr=%s sgpr=%s scratch=%s\n      %s" % ((n, ms / n) + key[1:] + (key[0][:300],)))
PY
