#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one source: python tools/kres.py mimamo-net_amd/csrc/wino_fused.hip [extra hipcc flags]
(hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel; honours the file's `mm-hipcc-flags:` line)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
extra = sys.argv[2:]
ff = []
for i, line in enumerate(open(src)):
    if i > 80: break
    if "mm-hipcc-flags:" in line: ff += line.split("mm-hipcc-flags:", 1)[1].split()
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + ROOT + "/include", "-I" + ROOT + "/mimamo-net_amd/csrc",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kres.o"] + ff + extra
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
cur = None
rows = {}
for line in r.stdout.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: [^:]*:\d+:\d+: Name: (\S+)", line) or re.search(r" Name: (\S+) \[", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void mm::", "")
        rows[cur] = {}
        continue
    m = re.search(r"\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    if "error" in line: print(line)
for k, v in rows.items():
    print("%-60s vgpr %3d agpr %3d spill %3d scratch %4d occ %d lds %6d sgpr %3d" % (k[:60], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1),
          v.get("ScratchSize", -1), v.get("Occupancy", -1), v.get("LDS Size", -1), v.get("SGPRs", -1)))
