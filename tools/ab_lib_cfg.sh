#!/bin/bash
# same-box A/B of one bench configuration: current library (new) against tools/_old_libbm355.so (old), alternating
#   bash tools/ab_lib_cfg.sh grbm [extra bench arguments]
cd $GRAFT_REPO_ROOT
CFG=${1:-grbm}; shift
L=boltzmann_machines_amd/libbm355.so
cp $L /tmp/new.so
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp tools/_old_libbm355.so $L; else cp /tmp/new.so $L; fi
    touch $L
    timeout 300 python bench.py --config $CFG --no-cpu --no-others "$@" 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$CFG $v %.4f ms  frac %.4f  sweeps %s' % (d['ms_per_step'], d['roofline']['frac'], d['config'].get('mean_field_sweeps_executed')))"
  done
done
cp /tmp/new.so $L; touch $L
