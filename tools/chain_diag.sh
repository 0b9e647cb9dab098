#!/bin/bash
# developer tool: the chained launch (csrc/bm_chain.h) under its measurement knobs, and its rocprofv3 counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/chain_diag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])"; }
for c in ${CONFIGS:-gibbs rbm}; do
  for dbg in ${DBGS:-0 1 2 3}; do
    BM355_CHAIN=1 BM355_CHAIN_DBG=$dbg timeout 200 python $R/bench.py --config $c --no-cpu --no-others 2> $O/${c}_dbg$dbg.err | line "$c chain=1 dbg=$dbg"
  done
  BM355_CHAIN=0 timeout 200 python $R/bench.py --config $c --no-cpu --no-others 2> $O/${c}_off.err | line "$c chain=0"
  for m in 1 0; do
    B="python $R/bench.py --config $c --no-cpu --no-others --steps 20 --warmup 5 --precondition-s 0.1"
    BM355_CHAIN=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c.m$m/stats -o s -- $B > $O/$c.m$m.stats.log 2>&1
    BM355_CHAIN=$m rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$c.m$m/fetch -o f -- $B > $O/$c.m$m.fetch.log 2>&1
    BM355_CHAIN=$m rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/$c.m$m/sq -o q -- $B > $O/$c.m$m.sq.log 2>&1
    find $O/$c.m$m -name '*_kernel_trace.csv' -delete; find $O/$c.m$m -name '*.db' -delete
    python - <<PY
import csv, glob, collections
for kind in ('stats', 'fetch', 'sq'):
    for f in glob.glob('$O/$c.m$m/%s/**/*.csv' % kind, recursive=True):
        if 'kernel_stats' in f:
            for r in list(csv.DictReader(open(f)))[:4]:
                print('$c chain=$m', r['Name'][:70], r['Calls'], r['AverageNs'])
        if 'counter_collection' in f:
            a = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                if 'bm::' in r['Kernel_Name']:
                    k = (r['Kernel_Name'][:60], r['Counter_Name']); a[k][0] += float(r['Counter_Value']); a[k][1] += 1
            for k, v in sorted(a.items(), key=lambda kv: -kv[1][0])[:12]:
                print('$c chain=$m', k, 'avg %.0f over %d' % (v[0] / v[1], v[1]))
PY
  done
done
