#!/bin/bash
# Same-box A/B of library variants at the KERNEL level: for every tools/_ab/lib<TAG>.so (built here, travels with the snapshot) one
# single-stream step through tools/layer_table.py, rows matching PATTERN.   gpurun -- 'bash tools/ab_layers.sh "wino-fused|totals" [reps] [clips]'
cd $GRAFT_REPO_ROOT
PAT=${1:-totals}; REPS=${2:-2}; CLIPS=${3:-32}
cp mimamo-net_amd/libmimamo_hip.so /tmp/_orig.so
for rep in $(seq $REPS); do for lib in /tmp/_orig.so tools/_ab/lib*.so; do
  tag=$(basename $lib .so); tag=${tag#lib}
  [ $lib != /tmp/_orig.so ] && cp $lib mimamo-net_amd/libmimamo_hip.so
  echo "== $tag (rep $rep) ${AB_ENV}"
  env $AB_ENV python tools/layer_table.py $CLIPS 1 2>&1 | grep -E "$PAT"
  cp /tmp/_orig.so mimamo-net_amd/libmimamo_hip.so
done; done
