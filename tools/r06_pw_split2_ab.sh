cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== rep $rep [default]"; python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
  echo "== rep $rep [MM_PW_SPLIT=2]"; MM_PW_SPLIT=2 python tools/phase_stage_bench.py 32 256 2>&1 | grep clips
done
