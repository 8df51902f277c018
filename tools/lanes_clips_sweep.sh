#!/bin/bash
# one box: lanes x clips per step (bench.py --no-cpu-baseline --no-extra), two repetitions
for rep in 1 2; do
for cfg in "--lanes 1" "--lanes 2" "--lanes 3" "--lanes 4" "--lanes 2 --clips 48" "--lanes 3 --clips 48" "--lanes 4 --clips 64" "--lanes 3 --clips 24"; do
  python bench.py --no-cpu-baseline --no-extra $cfg 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('$cfg: %9.1f f/s %8.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done
