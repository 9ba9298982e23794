#!/usr/bin/env python
"""Timeline of one engine step from a rocprofv3 kernel trace:
python tools/step_timeline.py <kernel_trace.csv> [step_index] [marker]
Steps are delimited by the batched K_uu covariance launch, rbf_kernel<P, true>, or by the first kernel whose short name
starts with `marker` (e.g. unpack_tril_kernel for E-steps with a cached K_uu chain, which have no covariance launch)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    m = re.search(r'(\w+)(<[^>]*>)?\(', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:40]


names = [short(r['Kernel_Name']) for r in rows]
marker = sys.argv[3] if len(sys.argv) > 3 else None
marks = [i for i, n in enumerate(names) if (n.startswith(marker) if marker else (n.startswith('rbf_kernel<') and 'true>' in n))]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 2
s, e = marks[k], marks[k + 1]
t0 = int(rows[s]['Start_Timestamp'])
run, last = 0, None
for r, n in zip(rows[s:e], names[s:e]):
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if n == last and n in ('potrf_step_kernel', '__amd_rocclr_copyBuffer'):
        run += 1
        continue
    if run:
        print('             ... + %d more %s' % (run, last))
        run = 0
    last = n
    print("%8.1f %8.1f  q%-3s %s" % ((st - t0) / 1e3, (en - st) / 1e3, r.get('Queue_Id', ''), n))
print("step span %.1f us" % ((int(rows[e]['Start_Timestamp']) - t0) / 1e3))
