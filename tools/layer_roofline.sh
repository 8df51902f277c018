#!/bin/bash
# Per-launch limiting roofline (tools/layer_roofline.py): one plain run for the hipEvent times, two rocprofv3 PMC passes for the HBM
# bytes of every launch.  usage (GPU box): bash tools/layer_roofline.sh [clips]   ->  gpurun_out/r06_layer_table_<clips>clips.txt,
# gpurun_out/r06_layer_bytes_<clips>clips.json (copy both to profiles/)
CLIPS=${1:-32}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
MM_PROF_DUMP=/tmp/lr_dump.csv python $R/tools/layer_table.py $CLIPS 1 > /tmp/lr_plain.txt 2>&1 || { tail -5 /tmp/lr_plain.txt; exit 1; }
cp /tmp/lr_dump.csv /tmp/lr_dump_plain.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/lr_$c
  MM_PROF_DUMP=/tmp/lr_dump_pmc.csv rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/lr_$c -o out -- python $R/tools/layer_table.py $CLIPS 1 > /tmp/lr_$c.log 2>&1 \
      || { tail -5 /tmp/lr_$c.log; exit 1; }
done
python $R/tools/layer_roofline.py /tmp/lr_dump_plain.csv /tmp/lr_FETCH_SIZE /tmp/lr_WRITE_SIZE $CLIPS $R/gpurun_out/r06_layer_table_${CLIPS}clips.txt \
       $R/gpurun_out/r06_layer_bytes_${CLIPS}clips.json
