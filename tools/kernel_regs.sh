#!/bin/bash
# registers, scratch and occupancy of every kernel of one .hip file, as the product build compiles it:  tools/kernel_regs.sh integrate [filter] [extra flags]
f=${1:-integrate}; pat=${2:-.}; shift; shift
cd "$(dirname "$0")/../hybkinectfu_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-value -I../../include "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kregs_$$.o 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /TotalSGPRs:/ {s=$0; sub(/.*TotalSGPRs: /,"",s); sub(/ .*/,"",s)}
       / VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
       /ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o); printf "%-70s sgpr %3s vgpr %3s scratch %4s occ %s\n", name, s, v, sc, o}' | grep -E "$pat"
rm -f /tmp/kregs_$$.o
