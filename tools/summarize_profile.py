#!/usr/bin/env python
"""Condense gpurun_out/prof_rN/<config> (written by tools/profile.sh) into profiles/:
kernel-trace stats, the PMC counters per kernel, and the HBM traffic figure that bench.py reports as
roofline.traffic (FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM prescribes for wide coalesced
reads on gfx950, plus WRITE_SIZE; both are in KiB), per STEP of the bench command.

    python tools/summarize_profile.py gpurun_out/prof_r2 r2 [configs...]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_r2'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r2'
configs = sys.argv[3:] or ['rbm', 'gibbs', 'grbm', 'dbm', 'ais', 'aisfast', 'grbmfast']
os.makedirs('profiles', exist_ok=True)


def provenance():
    """source hash the profile was taken on (written by tools/profile.sh on the GPU box) and, when the tree this script runs
    in has the same library sources, its commit"""
    try:
        sha = open(os.path.join(src, 'source_sha16.txt')).read().strip()
    except OSError:
        return {}
    out = {'source_sha16': sha}
    try:
        import subprocess
        sys.path.insert(0, os.getcwd())
        import bench
        if bench.kernel_source_sha16() == sha:
            dirty = subprocess.run(['git', 'status', '--porcelain', '--', 'boltzmann_machines_amd/csrc', 'include'],
                                   capture_output=True, text=True).stdout.strip()
            head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
            out['head'] = head + (' + uncommitted changes of the library sources' if dirty else '')
    except Exception:
        pass
    return out


def find(d, pat):
    f = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return f[0] if f else None


def bench_steps(log):
    """steps + warmup + precondition launches are all profiled: per-step figures use the counted dispatches"""
    try:
        return json.loads(open(log).read().strip().splitlines()[-1])
    except Exception:
        return None


for cfg in configs:
    d = os.path.join(src, cfg)
    if not os.path.isdir(d):
        continue
    totals = collections.defaultdict(float)
    ndisp = collections.defaultdict(int)

    def agg(path):
        rows = list(csv.DictReader(open(path)))
        a = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            if 'bm::' not in r['Kernel_Name'] and 'bm64::' not in r['Kernel_Name']:
                continue
            a[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            totals[r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'FETCH_SIZE':
                ndisp[r['Kernel_Name']] += 1
        return {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in a.items()}

    sfile = find(os.path.join(d, 'stats'), '*kernel_stats.csv')
    stats = {}
    if sfile:
        shutil.copy(sfile, 'profiles/%s_%s_kernel_stats.csv' % (tag, cfg))
        stats = {r['Name']: r for r in csv.DictReader(open(sfile))}
    pmc = {}
    for sub in ('fetch', 'write', 'sq'):
        f = find(os.path.join(d, sub), '*counter_collection.csv')
        if f:
            for k, dd in agg(f).items():
                pmc.setdefault(k, {}).update(dd)
    # steps seen by the counter pass: one marker kernel per step
    marker = {'rbm': 'grad_kernel', 'gibbs': None, 'grbm': 'maxnorm_kernel', 'dbm': 'dbm_bias_multi_kernel', 'ais': 'ais_init_kernel',
              'aisfast': 'ais_init_kernel', 'grbmfast': 'maxnorm_kernel'}[cfg]
    if marker == 'dbm_bias_multi_kernel' and not any(marker in k for k in ndisp):
        marker = 'dbm_bias_kernel'                      # libraries before round 6: three bias launches per update
    if marker:
        per = {'dbm_bias_kernel': 3}.get(marker, 1)
        n_steps = max(1, sum(n for k, n in ndisp.items() if marker in k) // per)
    else:       # gibbs: one chained launch per step (10 sweeps; csrc/bm_chain.h), or 2 act launches per sweep
        n_steps = max(1, sum(n for k, n in ndisp.items() if 'act_chain_kernel' in k) +
                      sum(n for k, n in ndisp.items() if 'act_kernel' in k) // 20)
    traffic = (2 * totals['FETCH_SIZE'] + totals['WRITE_SIZE']) * 1024 / n_steps
    if cfg in ('ais', 'aisfast'):
        # the counter passes of this configuration run 20 betas of a 1000-beta step behind ~1500 launches of the launch tuner's
        # candidates (a new process tunes again): their total says nothing about a bench step.  The per-kernel table below is
        # per dispatch and stands; the per-step figure is withheld (bench.py prints `traffic: null` and why).
        traffic = None
    prov = provenance()
    lines = ['# rocprofv3 summary %s / %s - `python bench.py --config %s` on 1x MI355X' % (tag, cfg, cfg), '',
             'library sources %s, commit %s' % (prov.get('source_sha16', 'unrecorded'), prov.get('head', 'unrecorded')), '',
             '| kernel | calls | avg us (kernel-trace) | total % | FETCH_SIZE KiB | x2 corrected MB | WRITE_SIZE KiB | MFMA busy % | issue-stalled % (WAIT_INST_ANY) | parked % (WAIT_ANY) | MFMA instr f32 / bf16 | LDS bank-conflict cycles |',
             '|---|---|---|---|---|---|---|---|---|---|---|---|']
    names = sorted(set(pmc) | {k for k in stats if 'bm::' in k or 'bm64::' in k},
                   key=lambda k: -float(stats.get(k, {}).get('TotalDurationNs', 0) or 0))
    for k in names:
        dd, st = pmc.get(k, {}), stats.get(k, {})
        busy = 100.0 * dd.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024.0 / max(dd.get('SQ_BUSY_CYCLES', 1) / 32.0, 1)
        wc = max(dd.get('SQ_WAVE_CYCLES', 0), 1)
        lines.append('| `%s` | %s | %.2f | %s | %.0f | %.1f | %.0f | %s | %s | %s | %s | %s |' % (
            k.split('(')[0], st.get('Calls', '?'), float(st.get('AverageNs', 0) or 0) / 1e3, st.get('Percentage', '?'),
            dd.get('FETCH_SIZE', 0), 2 * dd.get('FETCH_SIZE', 0) * 1024 / 1e6, dd.get('WRITE_SIZE', 0),
            ('%.1f' % busy) if 'SQ_BUSY_CYCLES' in dd else '-',
            ('%.1f' % (100.0 * dd['SQ_WAIT_INST_ANY'] / wc)) if 'SQ_WAIT_INST_ANY' in dd else '-',
            ('%.1f' % (100.0 * dd['SQ_WAIT_ANY'] / wc)) if 'SQ_WAIT_ANY' in dd else '-',
            ('%.0f / %.0f' % (dd.get('SQ_INSTS_VALU_MFMA_F32', 0), dd.get('SQ_INSTS_VALU_MFMA_BF16', 0))) if 'SQ_INSTS_VALU_MFMA_F32' in dd else '-',
            ('%.0f' % dd['SQ_LDS_BANK_CONFLICT']) if 'SQ_LDS_BANK_CONFLICT' in dd else '-'))
    lines += ['', 'MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs); issue-stalled / parked % = the counter / SQ_WAVE_CYCLES',
              '(quad-cycle units, MI355X_MICROARCH.md); MFMA instr = wave instructions per launch.',
              ('HBM-side traffic per bench step (all engine dispatches of the counter pass / %d steps, FETCH doubled + WRITE): %.1f MB' % (n_steps, traffic / 1e6))
              if traffic is not None else 'HBM-side traffic per bench step: withheld - the counter passes (20 betas) are dominated by the launch tuner\'s candidate launches; the table is per dispatch',
              '(working sets up to 256 MB are Infinity-Cache resident; these are L2-miss side counters, not DRAM bytes).', '']
    b = os.path.join(src, cfg + '.bench.json')
    if os.path.exists(b) and os.path.getsize(b) > 10:
        shutil.copy(b, 'profiles/%s_%s_bench.json' % (tag, cfg))
        lines += ['bench line of the same tree: `%s`' % open(b).read().strip()[:700], '']
    open('profiles/%s_%s_summary.md' % (tag, cfg), 'w').write('\n'.join(lines))
    json.dump(dict(provenance(), **{'traffic_bytes_per_update': traffic, 'steps_in_counter_pass': n_steps, 'pmc': pmc}),
              open('profiles/%s_%s_pmc.json' % (tag, cfg), 'w'), indent=1)
    print('\n'.join(lines))
