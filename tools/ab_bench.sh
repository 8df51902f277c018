#!/bin/bash
# Same-box A/B of library variants: box-to-box spread (+-2 %) hides the 1 % effects this round is down to, so the
# variants are built HERE (cross-compiled), parked under tools/_ab/lib<TAG>.so (git-ignored, but they travel with the
# snapshot), and this script alternates them (and the shipped library) on ONE GPU box:   gpurun -- 'bash tools/ab_bench.sh [reps] [bench args]'
# (variants are selected with MM_LIB_PATH -- nothing is ever copied over the shipped library)
cd $GRAFT_REPO_ROOT
REPS=${1:-3}; shift
for rep in $(seq $REPS); do for lib in shipped tools/_ab/lib*.so; do
  tag=$(basename $lib .so); tag=${tag#lib}
  if [ $lib = shipped ]; then unset MM_LIB_PATH; else export MM_LIB_PATH=$PWD/$lib; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$tag', round(d['value']), 'frames/s', round(d['ms_per_step'],2), 'ms/step | conv', round(r['ms_per_step'],2), 'transforms', round(r['winograd_transforms']['ms_per_step'],2), 'phase', round(d['roofline_phase']['ms_per_step'],2))"
done; done
