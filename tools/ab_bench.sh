#!/bin/bash
# Same-box A/B of library variants: box-to-box spread (+-2 %) hides the 1 % effects this round is down to, so the
# variants are built HERE (cross-compiled), parked under tools/_ab/lib<TAG>.so (git-ignored, but they travel with the
# snapshot), and this script alternates them on ONE GPU box:   gpurun -- 'bash tools/ab_bench.sh [reps] [bench args]'
cd $GRAFT_REPO_ROOT
REPS=${1:-3}; shift
cp mimamo-net_amd/libmimamo_hip.so /tmp/_orig.so
for rep in $(seq $REPS); do for lib in tools/_ab/lib*.so; do
  tag=$(basename $lib .so); tag=${tag#lib}
  cp $lib mimamo-net_amd/libmimamo_hip.so
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', round(d['value']), 'frames/s', round(d['ms_per_step'],2), 'ms/step | conv', round(r['ms_per_step'],2), 'transforms', round(r['winograd_transforms']['ms_per_step'],2), 'phase', round(d['roofline_phase']['ms_per_step'],2))"
done; done
cp /tmp/_orig.so mimamo-net_amd/libmimamo_hip.so
