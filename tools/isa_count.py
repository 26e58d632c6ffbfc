#!/usr/bin/env python3
"""usage: tools/isa_count.py <file.s> <kernel-substring>: instruction mix and register counts of the matching kernels in a hipcc -S listing"""
import re, sys, collections
s = open(sys.argv[1]).read()
want = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    valu = [i for i in ins if i.startswith('v_')]
    print(name[:90])
    print('  instructions %d  valu %d (pk %d, mov %d, cndmask %d, f64 %d)  salu %d  vmem loads %d  stores %d  ds %d  waitcnt %d' % (
        len(ins), len(valu), sum(i.startswith('v_pk') for i in valu), sum(i.startswith('v_mov') or i.startswith('v_accvgpr') for i in valu),
        sum(i.startswith('v_cndmask') for i in valu), sum('f64' in i for i in valu), sum(i.startswith('s_') and not i.startswith('s_waitcnt') for i in ins),
        sum(i.startswith(('buffer_load', 'global_load', 'flat_load')) for i in ins), sum(i.startswith(('buffer_store', 'global_store', 'flat_store')) for i in ins),
        sum(i.startswith('ds_') for i in ins), sum(i.startswith('s_waitcnt') for i in ins)))
    top = collections.Counter(valu).most_common(14)
    print('  top valu:', ', '.join('%s %d' % t for t in top))
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    if want in m.group(1):
        d = dict(re.findall(r'\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)', m.group(2)))
        print('  %s: %s' % (m.group(1)[:60], d))
