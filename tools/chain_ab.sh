#!/bin/bash
# same-box A/B of the CD-1 update: chained passes (BM355_CHAIN=2) against per-pass launches (0), alternating runs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/chain_ab; mkdir -p $O
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_wall'))"; }
for rep in 1 2 3; do
  for m in 2 0; do
    BM355_CHAIN=$m timeout 200 python bench.py --config rbm --no-cpu --no-others 2> /dev/null | line "rbm chain=$m (2000 steps)"
    BM355_CHAIN=$m timeout 200 python bench.py --config rbm --no-cpu --no-others --steps 20 --warmup 5 2> /dev/null | line "rbm chain=$m (20 steps)"
  done
done
cd /tmp && export TMPDIR=/tmp
for m in 2 0; do
  BM355_CHAIN=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m -o s -- python $R/bench.py --config rbm --no-cpu --no-others --steps 300 --warmup 30 --precondition-s 0.1 > $O/m$m.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('$O/m$m/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print('chain=$m', r['Name'][:75], r['Calls'], r['AverageNs'])
PY
  find $O/m$m -name '*_kernel_trace.csv' -delete; find $O/m$m -name '*.db' -delete
done
