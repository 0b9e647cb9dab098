#!/usr/bin/env python
"""Timeline of the last chained DBM launch (BM355_DEBUG=dch_stamps=file, csrc/bm_dbmchain.h):
   python tools/dch_timeline.py file [team=0]
per tile: start, end of the wait, end of the main loop, end of the epilogue, end of the publish (100 MHz clock -> us)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(256, -1, 8)
team = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for b in range(a.shape[0]):
    for t in range(a.shape[1]):
        r = a[b, t]
        if r[0] == 0 or r[7] != team:
            continue
        rows.append((r[0], b, r[5], r[6], r[1], r[2], r[3], r[4]))
rows.sort()
t0 = rows[0][0]
us = lambda x: (x - t0) / 100.0 if x else float('nan')
print('tiles of team %d: %d' % (team, len(rows)))
# per (family, pass, sweep): first start, last end, mean durations of the stages
from collections import defaultdict
g = defaultdict(list)
for st, b, code, ti, w, m, e, p in rows:
    g[int(code)].append((us(st), us(w), us(m), us(e), us(p)))
print('%8s %5s %9s %9s | %7s %7s %7s %7s' % ('code', 'n', 'first', 'last end', 'wait', 'loop', 'epi', 'publish'))
for code in sorted(g, key=lambda c: min(x[0] for x in g[c])):
    v = np.array(g[code])
    print('%8d %5d %9.2f %9.2f | %7.2f %7.2f %7.2f %7.2f' % (code, len(v), v[:, 0].min(), np.nanmax(v[:, 4]),
          np.nanmean(v[:, 1] - v[:, 0]), np.nanmean(v[:, 2] - v[:, 1]), np.nanmean(v[:, 3] - v[:, 2]), np.nanmean(v[:, 4] - v[:, 3])))
    if len(g) > 60 and min(x[0] for x in g[code]) > 400:
        break
