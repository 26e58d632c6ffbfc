#!/bin/bash
# C2's fusion pass against its own floors, one box: product, no store, arithmetic + gathers only, pure brick read-modify-write, read only, write only,
# and the product kernel over half / a quarter of the queue (exp 8 / 9: what does the kernel cost when it has almost nothing to do?)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export KF_LIB=$ROOT/hybkinectfu_amd/libhybkf_exp.so
for m in 0 1 2 4 6 7 8 9; do
  printf "c2 exp %d: " $m
  KF_INTEGRATE_EXP=$m python3 $ROOT/tools/bench_integrate.py c2 100 | sed 's/^c2: //'
done
