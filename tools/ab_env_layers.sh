#!/bin/bash
# same-box A/B of one environment knob at the KERNEL level: tools/ab_env_layers.sh VAR "v1 v2 ..." PATTERN [reps] [clips]
# (rows of python tools/layer_table.py matching PATTERN, alternating the values)
VAR=$1; VALS=$2; PAT=$3; REPS=${4:-2}; CLIPS=${5:-32}
for rep in $(seq $REPS); do
for v in $VALS; do
  echo "== $VAR=$v (rep $rep)"
  env $VAR=$v python tools/layer_table.py $CLIPS 1 2>/dev/null | grep -E "$PAT"
done
done
