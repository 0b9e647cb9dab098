#!/bin/bash
# same-box A/B on the AIS configuration: current library (new) against tools/_old_libbm355.so (old), alternating
cd $GRAFT_REPO_ROOT
L=boltzmann_machines_amd/libbm355.so
cp $L /tmp/new.so
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp tools/_old_libbm355.so $L; else cp /tmp/new.so $L; fi
    touch $L
    python bench.py --config ais --no-cpu --no-others --steps 1 --warmup 1 --ais-betas ${BETAS:-300} 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v %.3f ms per run  frac %.4f  logZ %s' % (d['ms_per_step'], d['roofline']['frac'], d['config'].get('log_Z_mean', d['config'].get('log_Z'))))"
  done
done
cp /tmp/new.so $L
