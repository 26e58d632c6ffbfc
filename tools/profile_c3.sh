#!/bin/bash
# Runs on the GPU box (via gpurun): C3 (512^3 @ 4 m, VGA, CameraPoseFinderSDF) -- rocprofv3 kernel-trace stats, then separate PMC passes for the tracking kernel
# (SQ instruction mix / busy / wait, FETCH_SIZE, WRITE_SIZE, L2 hit).  usage: tools/profile_c3.sh <tag> [kernel-substring]; results under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r05_c3}; KSUB=${2:-k_sdf}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--config c3 --no-cpu-baseline --no-extras --no-scaling-reference"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --steps 100 --warmup 5 $COMMON > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
pass() {   # name, counters
  rocprofv3 --pmc $2 -d $OUT/pmc_$1 --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 $COMMON > $OUT/bench_$1.log 2>&1 || { tail -5 $OUT/bench_$1.log; return 1; }
  python3 - "$OUT/pmc_$1" "$KSUB" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        per[r["Counter_Name"]][int(r.get("Dispatch_Id") or 0)] += float(r["Counter_Value"])
for k, d in sorted(per.items()):
    v = sorted(d.values())
    print("%-28s avg/launch = %.4g  median = %.4g  (n=%d)" % (k, sum(v) / len(v), v[len(v) // 2], len(v)))
PY
}
{
pass sq1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" || exit 1
pass sq2 "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" || exit 1
pass fetch "FETCH_SIZE" || exit 1
pass write "WRITE_SIZE" || exit 1
pass l2 "TCC_HIT_sum TCC_MISS_sum" || exit 1
pass tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" || true
} > $OUT/counters.txt 2>&1
grep -h '"value"' $OUT/bench_trace.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('C3 under the tracer:', j['value'], 'frames/s', j.get('stage_us'))" >> $OUT/counters.txt 2>&1 || true
head -25 $OUT/kernel_stats.csv; cat $OUT/counters.txt
