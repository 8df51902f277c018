#!/bin/bash
# HBM/fabric traffic per bench step from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (TCC slot budget), --kernel-trace only, FETCH_SIZE doubled (gfx950 reports half the bytes of
# 16-byte-per-lane streaming reads; checked on a 1x1 conv whose operand read is known).  Writes
# gpurun_out/r06_conv_traffic_<clips>clips.json and gpurun_out/r06_phase_traffic_<clips>clips.json, each stamped with the hash of
# the kernel sources they were measured on (bench.py only quotes a summary whose hash matches).  usage: tools/pmc_traffic.sh [clips]
CLIPS=${1:-32}
STEPS=2; WARM=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bt_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/bt_$c -o out -- \
      python $R/bench.py --steps $STEPS --warmup $WARM --clips $CLIPS --no-cpu-baseline --no-extra > /tmp/bt_$c.log 2>&1
done
python - "$CLIPS" "$STEPS" "$WARM" "$R" <<'PY'
import csv, glob, json, sys
clips, steps, warm, root = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root)
import bench
nsteps = steps + warm + 2          # + the two single-stream steps of the roofline leg
for c in ("FETCH_SIZE", "WRITE_SIZE"):   # ... as the profiled run itself reports it (bench.py's line: hot_path_steps_executed)
    for line in open("/tmp/bt_%s.log" % c):
        if line.startswith('{"metric"'):
            got = json.loads(line).get("hot_path_steps_executed")
            assert got == nsteps, (c, got, nsteps)
groups = {"conv": ("conv_mfma_kernel", "wino_fused_kernel"), "winograd_transforms": ("wino_in", "wino_out"),
          "pyramid": ("pyramid_kernel", "pyramid_frame_kernel", "pyramid_wave_kernel"), "phase_frames_windows": ("phase_window2_kernel",)}
tot = {g: {} for g in groups}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/bt_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        for g, pats in groups.items():
            if any(p in r["Kernel_Name"] for p in pats):
                tot[g][c] = tot[g].get(c, 0.0) + float(r["Counter_Value"])
def bytes_per_step(g):
    return (tot[g].get("FETCH_SIZE", 0.0) * 1024 * 2 + tot[g].get("WRITE_SIZE", 0.0) * 1024) / nsteps
common = {"clips_per_gpu": clips, "steps_profiled": nsteps, "fetch_correction": 2.0, "kernel_source_hash": bench.kernel_source_hash(),
          "how": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over bench.py "
                 "--steps %d --warmup %d --no-extra; (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per step" % (steps, warm)}
conv = dict(common, bytes_per_step=bytes_per_step("conv"), kernels="conv_mfma_kernel + wino_fused_kernel (all launches of a step)",
            fetch_kb_raw_per_step=tot["conv"].get("FETCH_SIZE", 0) / nsteps, write_kb_per_step=tot["conv"].get("WRITE_SIZE", 0) / nsteps,
            winograd_transforms_bytes_per_step=bytes_per_step("winograd_transforms"))
phase = dict(common, bytes_per_step=bytes_per_step("pyramid") + bytes_per_step("phase_frames_windows"),
             pyramid_bytes_per_step=bytes_per_step("pyramid"), frames_windows_bytes_per_step=bytes_per_step("phase_frames_windows"))
json.dump(conv, open(root + "/gpurun_out/r06_conv_traffic_%dclips.json" % clips, "w"), indent=1)
json.dump(phase, open(root + "/gpurun_out/r06_phase_traffic_%dclips.json" % clips, "w"), indent=1)
print(json.dumps(conv)); print(json.dumps(phase))
PY
