#!/bin/bash
# same-box A/B: prop-up from a maintained transpose W^T, x-major (default) against W k-major (BM355_UP_XM=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('kernels',{}); print('$1', d['ms_per_step'], d['roofline']['frac'], {n:v.get('avg_us') for n,v in k.items() if isinstance(v,dict)})"; }
for c in ${CONFIGS:-rbm grbm}; do
for rep in 1 2; do
  for m in 1 0; do
    BM355_UP_XM=$m timeout 200 python bench.py --config $c --no-cpu --no-others 2> /dev/null | line "$c up_xm=$m"
  done
done
done
