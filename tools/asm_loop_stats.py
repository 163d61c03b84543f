"""Instruction mix and vmcnt waits of the loops of one kernel in a hipcc -S listing (device only).
usage: python tools/asm_loop_stats.py file.s <kernel-name-substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(s) if re.match(r'^_Z\S*:', l) and pat in l][0]
end = [i for i, l in enumerate(s[start:]) if 's_endpgm' in l][0] + start
body = s[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
print("kernel lines", len(body), "loops", loops)
keys = ('v_mfma', 'ds_read', 'ds_write', 'ds_bpermute', 'global_load_lds', 'buffer_load', 'global_load', 'global_store', 's_waitcnt', 's_barrier', 'scratch', 'v_', 's_')
for a, b in loops:
    c = Counter()
    for l in body[a:b]:
        t = l.strip().split(' ')[0] if l.strip() else ''
        for k in keys:
            if t.startswith(k):
                c[k if k in ('v_', 's_') else t] += 1
                break
    print(a, b, dict(c))
    print("  vmcnt waits:", [l.strip().split(';')[0].strip() for l in body[a:b] if 's_waitcnt' in l and 'vmcnt' in l])
