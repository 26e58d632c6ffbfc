#!/bin/bash
# usage: tools/collect_profiles.sh <run-tag under gpurun_out/> <name-tag under profiles/> [configs...] : copies what tools/profile_round.sh left under gpurun_out/<run-tag>/
# (per config: the rocprofv3 kernel-stats CSV; the PMC summary; the traffic figures bench.py cites) to profiles/<name-tag>_*; gpurun_out/ itself is scratch.
set -e
cd "$(dirname "$0")/.."
RUN=$1; TAG=$2; shift 2
CFGS=${@:-c2 c2_saturated c4 c4_saturated c5}
for c in $CFGS; do
  f=$(find gpurun_out/$RUN/trace_$c -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${TAG}_${c}_kernel_stats.csv && echo "profiles/${TAG}_${c}_kernel_stats.csv"
done
# (gpurun_out/$RUN/summary.txt -- the PMC summary -- is merged into profiles/${TAG}_summary.txt by hand: one file, blocks from the runs that are current)
if [ -d gpurun_out/${RUN}_c3 ]; then
  cp gpurun_out/${RUN}_c3/kernel_stats.csv profiles/${TAG}_c3_kernel_stats.csv
  cp gpurun_out/${RUN}_c3/counters.txt profiles/${TAG}_c3_counters.txt
fi
