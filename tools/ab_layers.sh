#!/bin/bash
# Same-box A/B of library variants at the KERNEL level: for the shipped library and every tools/_ab/lib<TAG>.so (built here, travels with the
# snapshot) one single-stream step through tools/layer_table.py, rows matching PATTERN.   gpurun -- 'bash tools/ab_layers.sh "wino-fused|totals" [reps] [clips]'
# (variants are selected with MM_LIB_PATH -- nothing is ever copied over the shipped library)
cd $GRAFT_REPO_ROOT
PAT=${1:-totals}; REPS=${2:-2}; CLIPS=${3:-32}
for rep in $(seq $REPS); do for lib in shipped tools/_ab/lib*.so; do
  tag=$(basename $lib .so); tag=${tag#lib}
  echo "== $tag (rep $rep) ${AB_ENV}"
  if [ $lib = shipped ]; then env $AB_ENV python tools/layer_table.py $CLIPS 1 2>&1 | grep -E "$PAT"
  else env $AB_ENV MM_LIB_PATH=$PWD/$lib python tools/layer_table.py $CLIPS 1 2>&1 | grep -E "$PAT"; fi
done; done
