#!/bin/bash
# A/B of BM355_DEBUG switches on the AIS bench configuration (shortened beta ladder), alternating runs on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for v in "" "$@"; do
    BM355_DEBUG="$v" python bench.py --config ais --no-cpu --no-others --steps 1 --warmup 1 --ais-betas ${BETAS:-200} 2>/tmp/ais_err.txt \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-28s %.3f ms per run  frac %.4f' % ('[$v]', d['ms_per_step'], d['roofline']['frac']))"
    grep "bm355 tune" /tmp/ais_err.txt | head -12
  done
done
