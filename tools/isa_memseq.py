"""Diagnostic: the order of memory operations, waits and barriers of one kernel in a hipcc -S listing (run-length compressed), with its register counts.
usage: python tools/isa_memseq.py file.s <mangled-name-substring> [max_lines]"""
import re, sys
s = open(sys.argv[1]).read()
names = [m for m in re.findall(r'^(\S+):\s', s, re.M) if sys.argv[2] in m and not m.startswith('.')]
for name in names:
    i = s.index(name + ':')
    j = s.index('.end_amdhsa_kernel', i)
    body = s[i:j]
    print(name, re.findall(r'\.amdhsa_next_free_vgpr \d+|\.amdhsa_accum_offset \d+|\.amdhsa_private_segment_fixed_size \d+', body), len(body.split('\n')), 'lines')
    seq = []
    for l in body.split('\n'):
        x = l.strip()
        t = x.split(' ')[0]
        if t.startswith(('global_load', 'global_store', 's_barrier', 'global_atomic', 'scratch', 'buffer_', 'flat_')):
            seq.append(t)
        elif t == 's_waitcnt' and 'vmcnt' in x:
            seq.append(x.split(';')[0].strip())
    out, prev, c = [], None, 0
    for k in seq:
        if k == prev:
            c += 1
        else:
            if prev:
                out.append(f"{prev} x{c}")
            prev, c = k, 1
    out.append(f"{prev} x{c}")
    print('\n'.join(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 200]))
