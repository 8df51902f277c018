#!/bin/bash
# same-box sweep of lanes x hardware queues.  usage: tools/lanes_sweep.sh
run() {
  label="$1"; shift
  "$@" 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('%-40s %9.1f f/s %8.3f ms/step' % ('$label', d['value'], d['ms_per_step']))"
}
B="python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline"
for q in 4 8 16; do
  for l in 1 2 3 4 5 6 8; do
    run "queues=$q lanes=$l" env GPU_MAX_HW_QUEUES=$q $B --lanes $l
  done
done
