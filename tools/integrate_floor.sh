#!/bin/bash
# The fusion pass (k_integrate_pairs) taken apart on ONE GPU box (boxes differ by up to ~15 %: only numbers of one call compare):
#   exp 0  the product kernel                      exp 1  no store            exp 2  no voxel load, no store = the arithmetic + depth gathers
#   exp 4  no arithmetic: every queued brick read and written back (all lanes)    exp 6 / 7  read only / write only
# Needs the experiments variant of the library: make -C hybkinectfu_amd/csrc experiments
# usage (GPU box): tools/integrate_floor.sh > gpurun_out/integrate_floor.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export KF_LIB=$ROOT/hybkinectfu_amd/libhybkf_exp.so
for cfg in c4 c2; do
  reps=30; [ $cfg = c2 ] && reps=100
  for m in 0 1 2 4 6 7; do
    printf "%s exp %d: " $cfg $m
    KF_INTEGRATE_EXP=$m python3 $ROOT/tools/bench_integrate.py $cfg $reps | sed 's/^c[24]: //'
  done
done
