#!/bin/bash
# HBM/fabric traffic of the conv engine over one bench step, from rocprofv3 PMC counters, collected exactly as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot budget), kernel-trace only,
# and FETCH_SIZE doubled (gfx950 reports half the bytes of 16-byte-per-lane streaming reads; verified on a 1x1
# conv whose operand read is known: raw FETCH = 0.5 x algorithmic).  Writes gpurun_out/conv_traffic.json.
# usage: tools/pmc_bench_traffic.sh [clips]
CLIPS=${1:-32}
STEPS=2; WARM=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bt_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/bt_$c -o out -- \
      python $R/bench.py --steps $STEPS --warmup $WARM --clips $CLIPS --no-cpu-baseline > /tmp/bt_$c.log 2>&1
done
python - "$CLIPS" "$STEPS" "$WARM" "$R" <<'PY'
import csv, glob, json, sys
clips, steps, warm, root = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
tot = {}
launches = 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/bt_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    s = 0.0; n = 0
    for r in csv.DictReader(open(f)):
        if "conv_mfma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
            s += float(r["Counter_Value"]); n += 1
    tot[c] = s; launches = n
nsteps = steps + warm + 1          # + the hipEvent-timed step of the roofline leg
fetch_b = tot["FETCH_SIZE"] * 1024 * 2 / nsteps
write_b = tot["WRITE_SIZE"] * 1024 / nsteps
out = {"clips_per_gpu": clips, "steps_profiled": nsteps, "conv_launches_per_step": launches // nsteps,
       "fetch_kb_raw_per_step": tot["FETCH_SIZE"] / nsteps, "write_kb_per_step": tot["WRITE_SIZE"] / nsteps,
       "fetch_correction": 2.0, "bytes_per_step": fetch_b + write_b,
       "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), conv_mfma_kernel dispatches of bench.py"}
json.dump(out, open(root + "/gpurun_out/conv_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
