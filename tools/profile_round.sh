#!/bin/bash
# Round evidence in one gpurun call: rocprofv3 kernel stats of the bench (single stream and default lanes), PMC traffic,
# matrix-pipe utilisation, the default bench line.  Everything lands in gpurun_out/ under r06_* names; copy to profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
for L in 1 3; do
  rm -rf /tmp/ks$L
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$L -o out -- python $R/bench.py --steps 5 --warmup 2 --lanes $L --no-cpu-baseline --no-extra > /tmp/ks$L.log 2>&1
  cp $(find /tmp/ks$L -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats_32clips_lanes$L.csv
done
bash $R/tools/pmc_traffic.sh 32 > $O/r06_pmc_traffic.log 2>&1
bash $R/tools/pmc_kernel.sh conv_mfma_kernel 32 1 > $O/r06_conv_mfma_util_32clips.txt 2>&1
bash $R/tools/pmc_kernel.sh wino_fused_kernel 32 1 >> $O/r06_conv_mfma_util_32clips.txt 2>&1
bash $R/tools/layer_roofline.sh 32 > $O/r06_layer_roofline.log 2>&1      # -> r06_layer_table_32clips.txt (per-launch floors) + r06_layer_bytes_32clips.json
# the bench line quotes the PMC traffic / per-launch byte list of the sources it runs from profiles/: put this call's files there first
cp $O/r06_conv_traffic_32clips.json $O/r06_phase_traffic_32clips.json $O/r06_layer_bytes_32clips.json $O/r06_layer_table_32clips.txt $R/profiles/
cd $R && python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err
# phase-stage counters (three passes) of the kernels the bundle's sources ship
bash $R/tools/pmc_phase.sh pyramid_wave_kernel 32 > $O/r06_pmc_phase_stage.txt 2>&1
bash $R/tools/pmc_phase.sh phase_window2_kernel 32 >> $O/r06_pmc_phase_stage.txt 2>&1
