for rep in 1 2; do
for t in 2 4 1; do python tools/conv_bench.py 2048 56 56 256 64 1 1 0 $t 20; done
for t in 1 4 2 5; do python tools/conv_bench.py 2048 28 28 512 128 1 1 0 $t 20; done
for t in 5 1; do python tools/conv_bench.py 2048 14 14 256 1024 1 1 0 $t 20 1; done
done
