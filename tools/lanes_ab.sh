for rep in 1 2; do for l in 1 2 3; do python bench.py --lanes $l --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('lanes $l: %9.1f f/s %8.3f ms/step' % (d['value'], d['ms_per_step']))"; done; done
