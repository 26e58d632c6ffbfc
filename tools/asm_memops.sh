#!/bin/bash
# usage: tools/asm_memops.sh <file.hip> <mangled-kernel-substring> ["extra flags"]: device assembly of one kernel reduced to its memory
# operations, waits and readfirstlanes (are the loads of an iteration in flight together, or does each one get its own wait?)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Wno-unused-value -Iinclude $3 -S --cuda-device-only -o /tmp/asm_memops.s hybkinectfu_amd/csrc/$1 || exit 1
python3 - "$2" <<'PY'
import re, sys
txt = open('/tmp/asm_memops.s').read()
for f in re.split(r'\n(?=_Z[\w]+:)', txt):
    name = f.split(':', 1)[0]
    if sys.argv[1] not in name: continue
    lines = [l for l in f.splitlines() if (l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')) or l.startswith('.LBB')]
    print(name, len(lines), 'lines')
    for i, l in enumerate(lines):
        if 'global_' in l or 'vmcnt' in l or 'readfirstlane' in l or 'Loop Header' in l or 'buffer_' in l: print(i, l)
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', txt):
    if sys.argv[1] in m.group(1): print(m.group(1), 'sgpr', m.group(2), 'vgpr', m.group(3))
PY
