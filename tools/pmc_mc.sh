cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_mc_pmc; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $OUT --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_mcubes.py c2 3 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if "k_mc_count" in r["Kernel_Name"] or "k_mc_emit" in r["Kernel_Name"]:
        a = acc[(r["Kernel_Name"][:12], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-14s %-22s avg/launch = %.4g (n=%d)" % (k[0], k[1], s / n, n))
PY
