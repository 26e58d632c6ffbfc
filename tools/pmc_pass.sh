#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <config> "<counters>" [kernel-substring]   (GPU box) -> per-kernel average of each counter
TAG=$1; CFG=$2; CTRS=$3; KSUB=${4:-k_integrate_pairs}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS -d $OUT --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-scaling-reference --config $CFG > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-28s avg/launch = %.4g  (n=%d)" % (k, s / n, n))
PY
