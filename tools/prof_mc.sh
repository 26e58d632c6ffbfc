#!/bin/bash
# per-kernel durations of the marching-cubes extraction (rocprofv3 --kernel-trace --stats).  usage (on the GPU box): tools/prof_mc.sh [c2|c4]
cd /tmp && export TMPDIR=/tmp
CFG=${1:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/mc_trace_$CFG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_mcubes.py $CFG 5 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
grep -h "k_mc" $OUT/*/*kernel_stats.csv | cut -d, -f1-4
