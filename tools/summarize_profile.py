#!/usr/bin/env python
"""Condense gpurun_out/prof_rN (written by tools/profile_r1.sh) into profiles/:
kernel-trace stats, the PMC counters per kernel, and the HBM traffic figure that
bench.py reports as roofline.traffic (FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM
prescribes for wide coalesced reads on gfx950, plus WRITE_SIZE; both are in KiB)."""
import collections
import csv
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_r1'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r1'
os.makedirs('profiles', exist_ok=True)


totals = collections.defaultdict(float)      # counter -> sum over every dispatch of the engine's kernels
ndisp = collections.defaultdict(int)         # kernel -> dispatches seen in the FETCH pass


def agg(path):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if 'bm::' not in r['Kernel_Name']:
            continue
        a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        totals[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'FETCH_SIZE':
            ndisp[r['Kernel_Name']] += 1
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in a.items()}


shutil.copy(os.path.join(src, 'stats/s_kernel_stats.csv'), 'profiles/%s_kernel_stats.csv' % tag)
stats = {r['Name']: r for r in csv.DictReader(open(os.path.join(src, 'stats/s_kernel_stats.csv')))}
pmc = {}
for sub, f in (('fetch', 'f'), ('write', 'w'), ('sq', 'q')):
    for k, d in agg(os.path.join(src, sub, f + '_counter_collection.csv')).items():
        pmc.setdefault(k, {}).update(d)
# one grad_kernel dispatch per CD-1 update: per-update traffic = all engine dispatches / updates
n_updates = max(1, sum(n for k, n in ndisp.items() if 'grad_kernel' in k))
traffic = (2 * totals['FETCH_SIZE'] + totals['WRITE_SIZE']) * 1024 / n_updates
lines = ['# rocprofv3 summary %s — `python bench.py` (BernoulliRBM 784x1024, CD-1, batch 512, 1x MI355X)' % tag, '',
         '| kernel | calls | avg us (kernel-trace) | FETCH_SIZE KiB | x2 corrected MB | WRITE_SIZE KiB | MFMA busy % | LDS bank-conflict cycles |',
         '|---|---|---|---|---|---|---|---|']
for k, d in sorted(pmc.items()):
    st = stats.get(k, {})
    busy = 100.0 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024.0 / max(d.get('SQ_BUSY_CYCLES', 1) / 32.0, 1)
    fetch2 = 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e6
    lines.append('| `%s` | %s | %.2f | %.0f | %.1f | %.0f | %.1f | %.0f |' % (
        k.split('(')[0], st.get('Calls', '?'), float(st.get('AverageNs', 0)) / 1e3, d.get('FETCH_SIZE', 0), fetch2,
        d.get('WRITE_SIZE', 0), busy, d.get('SQ_LDS_BANK_CONFLICT', 0)))
lines += ['', 'MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs).',
          'HBM-side traffic per CD-1 update (all engine dispatches of the counter pass / %d updates, FETCH doubled + WRITE): %.1f MB' % (n_updates, traffic / 1e6),
          '(the working set is Infinity-Cache resident; these are L2-miss side counters, not DRAM bytes).', '']
open('profiles/%s_summary.md' % tag, 'w').write('\n'.join(lines))
json.dump({'traffic_bytes_per_update': traffic, 'pmc': pmc}, open('profiles/%s_pmc.json' % tag, 'w'), indent=1)
if os.path.exists(os.path.join(src, 'bench_default.json')):
    shutil.copy(os.path.join(src, 'bench_default.json'), 'profiles/%s_bench.json' % tag)
print('\n'.join(lines))
