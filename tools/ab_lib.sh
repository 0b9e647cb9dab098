#!/bin/bash
# same-box A/B: current library (new) against tools/_old_libbm355.so (old), alternating
cd $GRAFT_REPO_ROOT
L=boltzmann_machines_amd/libbm355.so
cp $L /tmp/new.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$1', d['ms_per_step'], d['roofline']['frac'], k['act_up']['avg_us'], k['act_down']['avg_us'], k['grad']['avg_us'])"; }
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp tools/_old_libbm355.so $L; else cp /tmp/new.so $L; fi
    touch $L
    timeout 200 python bench.py --no-others --no-cpu 2>/dev/null | line "$v 2000"
    timeout 200 python bench.py --no-others --no-cpu --steps 20 --warmup 5 2>/dev/null | line "$v 20"
  done
done
cp /tmp/new.so $L
