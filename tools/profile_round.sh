#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the fusion kernel, for C2 / C4 / C5 in the early
# regime and C2 / C4 in the saturated one (--pre-frames 160: the SAT instantiation of k_integrate_pairs).
# usage: tools/profile_round.sh <round-tag> ; results under gpurun_out/<tag>/, summaries copied by hand into profiles/.
# (rocprofv3 is given the interpreter itself after `--`, never a wrapper; --pmc runs carry no trace flags.)
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extras --no-scaling-reference"
run() {   # name, bench args for the trace pass, bench args for the PMC passes
  local NAME=$1 TRACE_ARGS=$2 PMC_ARGS=$3
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$NAME --output-format csv -- python3 $ROOT/bench.py $TRACE_ARGS $COMMON > $OUT/bench_trace_$NAME.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$NAME --output-format csv -- python3 $ROOT/bench.py $PMC_ARGS $COMMON > $OUT/bench_fetch_$NAME.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$NAME --output-format csv -- python3 $ROOT/bench.py $PMC_ARGS $COMMON > $OUT/bench_write_$NAME.log 2>&1 || return 1
  echo "profiled $NAME"
}
run c2 "--config c2 --steps 100 --warmup 5" "--config c2 --steps 20 --warmup 3" || exit 1
run c4 "--config c4 --steps 100 --warmup 5" "--config c4 --steps 20 --warmup 3" || exit 1
run c2_sat "--config c2 --pre-frames 160 --steps 60 --warmup 5" "--config c2 --pre-frames 160 --steps 20 --warmup 3" || exit 1
run c4_sat "--config c4 --pre-frames 160 --steps 60 --warmup 5" "--config c4 --pre-frames 160 --steps 20 --warmup 3" || exit 1
run c5 "--config c5 --steps 20 --warmup 4" "--config c5 --steps 10 --warmup 2" || exit 1
python3 $ROOT/tools/pmc_summary.py $OUT
