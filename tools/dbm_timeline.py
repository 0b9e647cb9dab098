#!/usr/bin/env python
"""Timeline of the DBM update from a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv`): where one update
spends its time - the part in front of the mean-field loop, the loop, the part behind it - and the time with NO kernel running
on the device (host round trips, launch boundaries), per update and averaged.

    python tools/dbm_timeline.py <dir or *_kernel_trace.csv> [n_updates_to_skip]

An update is delimited by its LAST kernel (`maxnorm_scale_kernel` of the top layer); kernels of the second stream (particle
sweeps) overlap the mean-field loop and are listed separately."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'bm(?:64)?::(\w+)', name)
    s = m.group(1) if m else name[:40]
    g = re.search(r'Geo<([\d, ]+)>, (.*)>', name)
    if g:
        s += '<%s|%s>' % (g.group(1).replace(' ', ''), g.group(2).replace(' ', '').replace('false', 'f').replace('true', 't'))
    return s


def main():
    p = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if os.path.isdir(p):
        p = sorted(glob.glob(os.path.join(p, '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(p)) if 'bm::' in r['Kernel_Name']]
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
    main_q = max(set(r[qkey] for r in rows), key=lambda q: sum(1 for r in rows if r[qkey] == q))
    # split into updates at the last max-norm rescale of an update (two per update for a 2-layer stack: take every second)
    ends = [i for i, r in enumerate(rows) if 'maxnorm_scale_kernel' in r['Kernel_Name']]
    L = 2
    cuts = ends[L - 1::L]
    updates, a = [], 0
    for c in cuts:
        updates.append(rows[a:c + 1])
        a = c + 1
    updates = updates[skip:]
    print('%d updates after skipping %d; main queue %s' % (len(updates), skip, main_q))
    agg = defaultdict(list)
    for u in updates:
        mq = [r for r in u if r[qkey] == main_q]
        oq = [r for r in u if r[qkey] != main_q]
        t0, t1 = u[0]['s'], u[-1]['e']
        # device-idle time: union of all kernels' intervals
        iv = sorted((r['s'], r['e']) for r in u)
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        gaps = []
        for s, e in iv[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, cur_e))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        # segments on the main queue: prologue = up to the first mean-field pass, loop = the MF passes, tail = after the last
        is_mf = [bool(re.search(r'act_kernel<.*, (true|false), (true|false), true>', r['Kernel_Name'])) for r in mq]   # MF flavour
        mf_idx = [i for i, f in enumerate(is_mf) if f]
        if not mf_idx:
            continue
        a0, a1 = mf_idx[0], mf_idx[-1]
        agg['total'].append((t1 - t0) / 1e3)
        agg['busy'].append(busy / 1e3)
        agg['prologue'].append((mq[a0]['s'] - t0) / 1e3)
        agg['loop'].append((mq[a1]['e'] - mq[a0]['s']) / 1e3)
        agg['tail'].append((t1 - mq[a1]['e']) / 1e3)
        agg['n_mf_kernels'].append(len(mf_idx))
        agg['loop_kernel_time'].append(sum(mq[i]['e'] - mq[i]['s'] for i in mf_idx) / 1e3)
        agg['gaps>5us'].append(sum(g for g, _ in gaps if g > 5000) / 1e3)
        agg['n_gaps>5us'].append(sum(1 for g, _ in gaps if g > 5000))
        agg['other_queue_kernel_time'].append(sum(r['e'] - r['s'] for r in oq) / 1e3)
    for k, v in agg.items():
        print('%-26s mean %9.2f  min %9.2f  max %9.2f' % (k, sum(v) / len(v), min(v), max(v)))
    # between-update gap
    bet = [(updates[i + 1][0]['s'] - updates[i][-1]['e']) / 1e3 for i in range(len(updates) - 1)]
    if bet:
        print('%-26s mean %9.2f  min %9.2f  max %9.2f' % ('between updates', sum(bet) / len(bet), min(bet), max(bet)))
    # one update in detail
    u = updates[len(updates) // 2]
    t0 = u[0]['s']
    print('\none update, kernel by kernel (us from its first kernel; q = queue; gap = since the previous kernel END on any queue):')
    last_e = t0
    n_mf = 0
    for r in u:
        nm = short(r['Kernel_Name'])
        gap = (r['s'] - last_e) / 1e3
        mf = 'act_kernel' in nm and r[qkey] == main_q
        if mf:
            n_mf += 1
            if 6 < n_mf < 10 ** 9 and gap < 5 and n_mf % 20:
                last_e = max(last_e, r['e'])
                continue
        print('%9.2f  %7.2f us  gap %7.2f  q%s  %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap, r[qkey][-2:], nm))
        last_e = max(last_e, r['e'])


if __name__ == '__main__':
    main()
