cd $GRAFT_REPO_ROOT
for t in 1 2 3 4; do python tools/conv_bench.py 1024 56 56 64 256 1 1 0 $t 20 1 2>&1 | tail -1; done
for t in 1 2 4; do python tools/conv_bench.py 1024 56 56 64 256 1 1 0 $t 20 0 2>&1 | tail -1; done
for t in 1 2 4; do python tools/conv_bench.py 1024 28 28 128 512 1 1 0 $t 20 1 2>&1 | tail -1; done
for t in 1 2 4; do python tools/conv_bench.py 1024 56 56 256 64 1 1 0 $t 20 0 2>&1 | tail -1; done
