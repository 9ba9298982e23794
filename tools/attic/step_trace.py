#!/usr/bin/env python
"""Print the kernel sequence of one engine step from a rocprofv3 kernel trace (csv): python tools/step_trace.py trace.csv [step]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    m = re.search(r'(\w+)(<[^>]*>)?\(', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:40]


names = [short(r['Kernel_Name']) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('rbf_kernel<') and 'true>' in n]
nq = 1
while nq < len(idx) and idx[nq] == idx[nq - 1] + 1:
    nq += 1
starts = idx[::nq]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
s, e = starts[k], starts[k + 1]
t0 = int(rows[s]['Start_Timestamp'])
prev_end = t0
out = []
for r, n in zip(rows[s:e], names[s:e]):
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append((n, (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', ''))))
    prev_end = en
i = 0
while i < len(out):
    j = i
    while j + 1 < len(out) and out[j + 1][0] == out[i][0]:
        j += 1
    dur = sum(o[2] for o in out[i:j + 1])
    gap = sum(o[3] for o in out[i:j + 1])
    print("%9.1f  %-40s x%-3d dur %8.1f us  gap %7.1f us  grid %s" % (out[i][1], out[i][0], j - i + 1, dur, gap, out[i][4]))
    i = j + 1
print("step span %.1f us" % ((int(rows[e]['Start_Timestamp']) - t0) / 1e3))
