#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE) for C2 / C4 / C5 -- every bench.py
# run launches the fusion kernel in its DEFER form (the stream's frames) and in its PLAIN read-modify-write form (the `roofline` leg behind the
# timed region) -- and, C2 / C4, past weight saturation (--pre-frames 160); plus marching-cubes extractions (tools/bench_mcubes.py).
# usage: tools/profile_round.sh <round-tag> [configs...] ; results under gpurun_out/<tag>/ (+ gpurun_out/<tag>_c3/: tools/profile_c3.sh), summaries copied by hand into profiles/.
# (rocprofv3 is given the interpreter itself after `--`, never a wrapper; --pmc runs carry no trace flags.)
set -o pipefail
TAG=${1:-r05}; shift
CFGS=${@:-c2 c4 c5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extras --no-scaling-reference"
run() {   # name, bench args for the trace pass, bench args for the PMC passes   (environment: as exported by the caller)
  local NAME=$1 TRACE_ARGS=$2 PMC_ARGS=$3
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$NAME --output-format csv -- python3 $ROOT/bench.py $TRACE_ARGS $COMMON > $OUT/bench_trace_$NAME.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$NAME --output-format csv -- python3 $ROOT/bench.py $PMC_ARGS $COMMON > $OUT/bench_fetch_$NAME.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$NAME --output-format csv -- python3 $ROOT/bench.py $PMC_ARGS $COMMON > $OUT/bench_write_$NAME.log 2>&1 || return 1
  echo "profiled $NAME"
}
for c in $CFGS; do
  ST=100; PST=20; [ $c = c5 ] && { ST=20; PST=10; }
  # one run holds both forms of the fusion kernel: <BR, true, false> (DEFER) on the stream, <BR, false, false> (plain) on bench.py's roofline leg
  run $c "--config $c --steps $ST --warmup 5" "--config $c --steps $PST --warmup 3" || exit 1
  [ $c = c5 ] || run ${c}_saturated "--config $c --pre-frames 160 --steps 60 --warmup 5" "--config $c --pre-frames 160 --steps 20 --warmup 3" || exit 1
done
for kind in fetch write; do
  CT=FETCH_SIZE; [ $kind = write ] && CT=WRITE_SIZE
  rocprofv3 --pmc $CT -d $OUT/pmc_${kind}_mc_c2 --output-format csv -- python3 $ROOT/tools/bench_mcubes.py c2 3 > $OUT/mc_$kind.log 2>&1 || exit 1
done
python3 $ROOT/tools/pmc_summary.py $OUT
# C3: the SDF tracker's launch (k_sdf_loop) -- kernel stats + SQ / FETCH / WRITE / L2 counters (the dominant kernel of a BASELINE config had no profile)
bash $ROOT/tools/profile_c3.sh ${TAG}_c3 k_sdf_loop > $OUT/c3.log 2>&1 || tail -5 $OUT/c3.log
