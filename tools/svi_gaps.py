#!/usr/bin/env python
"""Device idle gaps of the facade's SVI loop from a rocprofv3 kernel trace: every interval > 40 us in which NO kernel runs
(any stream), with the kernels on either side.  python tools/svi_gaps.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    m = re.search(r'(\w+)(<[^>]*>)?\(', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:40]


# skip the set-up phase: start at the 40th rbf_kernel<1, false> launch
t_end = 0
prev = None
out = []
for r in rows:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev is not None and st - t_end > 40000:
        out.append(((st - t_end) / 1e3, short(prev['Kernel_Name']), short(r['Kernel_Name']), st))
    if en > t_end:
        t_end, prev = en, r
tail = out[-40:]
for g, a, b, st in tail:
    print("%8.1f us idle   after %-28s before %s" % (g, a, b))
span = (int(rows[-1]['End_Timestamp']) - tail[0][3]) / 1e3
print("idle %.1f us of %.1f us (%.1f %%) over the last %d gaps" % (sum(g for g, _, _, _ in tail), span, 100 * sum(g for g, _, _, _ in tail) / span, len(tail)))
