#!/bin/bash
# the DBM bench configuration over 3000 updates (the mean-field loop shortens to ~13 sweeps as the model trains): A/B of switches
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for v in "" "$@"; do
    BM355_DEBUG="$v" python bench.py --config dbm --steps 3000 --warmup 10 --no-cpu --no-others 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-28s %.4f ms  frac %.4f  sweeps %.2f' % ('[$v]', d['ms_per_step'], d['roofline']['frac'], d['config'].get('mean_field_sweeps_executed', -1)))"
  done
done
