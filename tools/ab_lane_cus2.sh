#!/bin/bash
# follow-up of ab_lane_cus.sh: why are two half-chip lanes 2x slower than the free-running lanes?  single masked lanes, and a
# kernel trace of the two-half-lane step (do the lanes overlap at all?)
cd "$(dirname "$0")/.."
run() {
  python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.2f ms/step %9.0f frames/s' % ('$*', d['ms_per_step'], d['value']))"
}
run --lanes 1
run --lane-cus 0-255
run --lane-cus 0-127
run --lane-cus 0-63
run --lane-cus 0-127/0-127
export TMPDIR=/tmp
for spec in 0-127/128-255 0-255/0-255; do
  d=/tmp/lt_$(echo $spec | tr '/' '_'); rm -rf $d
  rocprofv3 --kernel-trace -d $d -o t --output-format csv -- python bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 --lane-cus $spec > /dev/null 2>&1
  echo "trace $spec:"; python tools/trace_overlap.py "$d/**/*kernel_trace.csv" --tail-ms=300
done
d=/tmp/lt_plain; rm -rf $d
rocprofv3 --kernel-trace -d $d -o t --output-format csv -- python bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 --lanes 2 > /dev/null 2>&1
echo "trace plain --lanes 2:"; python tools/trace_overlap.py "$d/**/*kernel_trace.csv" --tail-ms=200
