#!/bin/bash
# A/B of two library builds on the SAME GPU box (box-to-box spread is ~1 %, larger than most single changes):
# copy the build to compare against to tools/_old_libbm355.so (git-ignored, travels with the gpurun snapshot), build
# the new one in place, then on the box:  bash tools/ab.sh
L=boltzmann_machines_amd/libbm355.so
cp $L /tmp/new.so
run() { python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['avg_us'] for k,v in d['roofline']['kernels'].items() if isinstance(v,dict)})"; }
for r in 1 2; do
  cp tools/_old_libbm355.so $L; touch $L; run old
  cp /tmp/new.so $L; touch $L; run new
done
