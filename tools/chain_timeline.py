#!/usr/bin/env python
"""Developer tool: print the timeline of the last chained launch (BM355_DEBUG=chain_stamps=file, csrc/bm_chain.h) for one team.
usage: chain_timeline.py file [team]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(256, 16, 8)
team = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for b in range(256):
    for t in range(16):
        s = a[b, t]
        if s[0] == 0 or s[7] != team:
            continue
        rows.append((int(s[5]), int(s[6]), b, s[0], s[1], s[2], s[3], s[4]))
if not rows:
    sys.exit('no stamps for team %d' % team)
t0 = min(r[3] for r in rows)
rows.sort()
print('team %d: %d tiles; times in us from the first claim (100 MHz clock)' % (team, len(rows)))
print('pass  tiles | claimed: first .. last | wait done: first .. last | loop done: median (after wait done) | epilogue: median | published: median, last')
for p in sorted(set(r[0] for r in rows)):
    R = [r for r in rows if r[0] == p]
    f = lambda k: np.array([r[k] - t0 for r in R]) * 0.01
    cl, wd, ld, ep, pb = f(3), f(4), f(5), f(6), f(7)
    print('%4d  %5d | %7.2f .. %7.2f | %7.2f .. %7.2f | %6.2f | %5.2f | %5.2f, %7.2f' % (
        p, len(R), cl.min(), cl.max(), wd.min(), wd.max(), np.median(ld - wd), np.median(ep - ld), np.median(pb - ep), pb.max()))
