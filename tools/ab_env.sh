#!/bin/bash
# same-box A/B of one environment knob: tools/ab_env.sh VAR "v1 v2 ..." [extra bench args]
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
for v in $VALS; do
  env $VAR=$v python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('$VAR=$v $*: %9.1f f/s %8.3f ms/step | conv launches %7.3f ms (%d, single stream) frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline']['launches_per_step'], d['roofline']['frac']))"
done
done
