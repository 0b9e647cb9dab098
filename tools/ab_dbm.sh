#!/bin/bash
# A/B of BM355_DEBUG switches on the DBM bench configuration, alternating runs on one box:
#   bash tools/ab_dbm.sh "dbm_pcd_late=0" "dbm_overlap=0" ...     (the empty setting = default is always included)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for v in "" "$@"; do
    BM355_DEBUG="$v" python bench.py --config dbm --no-cpu --no-others --steps ${STEPS:-60} --warmup 8 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-28s %.4f ms  frac %.4f  sweeps %.2f' % ('[$v]', d['ms_per_step'], d['roofline']['frac'], d['config'].get('mean_field_sweeps_executed', -1)))"
  done
done
