#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the fusion kernel.
# usage: tools/profile_round.sh <round-tag> ; results under gpurun_out/<tag>/, summaries copied by hand into profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in c2 c4; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$CFG --output-format csv -- python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-scaling-reference --config $CFG > $OUT/bench_trace_$CFG.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$CFG --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-scaling-reference --config $CFG > $OUT/bench_fetch_$CFG.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$CFG --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-scaling-reference --config $CFG > $OUT/bench_write_$CFG.log 2>&1 || exit 1
done
python3 $ROOT/tools/pmc_summary.py $OUT
