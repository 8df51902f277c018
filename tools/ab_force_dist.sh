#!/bin/bash
# same-box A/B: bench.py with and without the (one-rank) RCCL process group.  usage: tools/ab_force_dist.sh [reps]
R=${1:-1}
run() {
  label="$1"; shift
  "$@" 2>/dev/null | python -c "
import sys, json
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('%-60s %9.1f f/s %8.3f ms/step | conv launches %7.3f ms (single stream)' % ('$label', d['value'], d['ms_per_step'], d['roofline']['ms_per_step']))"
}
B="python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline"
for i in $(seq 1 $R); do
  run "plain" $B
  run "force-dist (rccl)" $B --force-dist
  run "force-dist lanes=1" $B --force-dist --lanes 1
  run "plain lanes=1" $B --lanes 1
  run "force-dist TORCH_NCCL_ASYNC_ERROR_HANDLING=0" env TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0 $B --force-dist
  run "force-dist GPU_MAX_HW_QUEUES=8" env GPU_MAX_HW_QUEUES=8 $B --force-dist
  run "plain GPU_MAX_HW_QUEUES=8" env GPU_MAX_HW_QUEUES=8 $B
  run "force-dist NCCL_MAX_NCHANNELS=1" env NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1 $B --force-dist
done
