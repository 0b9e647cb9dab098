#!/bin/bash
# the DBM bench configuration with a loose mean-field tolerance (a SHORT loop, as in a trained model): A/B of switches
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in "" "$@"; do
    BM355_DEBUG="$v" python bench.py --config dbm --dbm-mf-tol ${TOL:-1e-3} --steps 200 --warmup 10 --no-cpu --no-others 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-28s %.4f ms  frac %.4f  sweeps %.2f' % ('[$v]', d['ms_per_step'], d['roofline']['frac'], d['config'].get('mean_field_sweeps_executed', -1)))"
  done
done
