#!/bin/bash
# HBM traffic of one conv shape: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md: TCC slots).
# usage: pmc_traffic.sh <conv_bench args>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/tr_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$tag -o out -- python $R/tools/conv_bench.py "$@" > /tmp/tr_$tag.log 2>&1
  f=$(find /tmp/tr_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    if "conv_mfma" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c, v in agg.items():
    print("   %-14s %.5g per launch (raw counter units)" % (c, v / cnt[c]))
PY
done
grep TFLOP /tmp/tr_FETCH_SIZE.log
