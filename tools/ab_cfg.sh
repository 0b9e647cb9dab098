#!/bin/bash
# A/B of BM355_DEBUG switches on one bench configuration, alternating runs on one box:
#   bash tools/ab_cfg.sh grbm "dbm_overlap=0" ...     (the empty setting = default is always included)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CFG=$1; shift
for rep in 1 2 3; do
  for v in "" "$@"; do
    BM355_DEBUG="$v" timeout 300 python bench.py --config $CFG --no-cpu --no-others 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$CFG %-28s %.4f ms  frac %.4f' % ('[$v]', d['ms_per_step'], d['roofline']['frac']))"
  done
done
