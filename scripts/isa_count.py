#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing: python scripts/isa_count.py file.s kernel_substring"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + sys.argv[2] + r'\S*:', l)][0]
fe = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
c = Counter()
for l in lines[start:fe]:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    op = l.split()[0]
    c[op.split('_')[0]] += 1
print(sum(c.values()), c.most_common(12))
