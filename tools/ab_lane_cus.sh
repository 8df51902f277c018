#!/bin/bash
# VERDICT r3 item 1(a), the pipeline: same-box A/B of the default three free-running lanes against lanes confined to CU partitions
# (bench.py --lane-cus, stream.PartitionStream).  Writes one line per configuration: spec, ms_per_step, frames/s.
#   gpurun -- 'bash tools/ab_lane_cus.sh > gpurun_out/ab_lane_cus.txt'
cd "$(dirname "$0")/.."
run() {
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.2f ms/step %9.0f frames/s' % ('$*', d['ms_per_step'], d['value']))"
}
run --lanes 3
run --lanes 2
run --lanes 1
run --lane-cus 0-255/0-255/0-255
run --lane-cus 0-127/128-255
run --lane-cus 0-127/128-255/0-255
run --lane-cus 0-127/128-255/0-127/128-255
run --lane-cus 0-159/96-255
run --lane-cus 0-191/64-255
run --lane-cus 0-191/64-255/0-255
run --lane-cus 0-95/96-255/0-255
run --lanes 3
