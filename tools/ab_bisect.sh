#!/bin/bash
# same-box A/B over the library builds under tools/ab/<sha>/ and HEAD (tools/ab_rbm.py), rounds in alternating order
# usage: bash tools/ab_bisect.sh [sha ...]      (default: every build under tools/ab)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
export ROC_ACTIVE_WAIT_TIMEOUT=1000000
LIST="${@:-$(ls tools/ab)}"
sleep 1; touch boltzmann_machines_amd/libbm355.so tools/ab/*/boltzmann_machines_amd/libbm355.so oracle/libbm_oracle.so
for r in 1 2 3; do
  for s in $LIST; do timeout 300 python tools/ab_rbm.py tools/ab/$s $s 2>&1 | tail -1 | tee -a $O/ab.log; done
  timeout 300 python tools/ab_rbm.py . HEAD 2>&1 | tail -1 | tee -a $O/ab.log
done
BM355_TUNE_LOG=1 timeout 300 python tools/ab_rbm.py . HEAD > $O/tune_head.log 2>&1
for s in $LIST; do BM355_TUNE_LOG=1 timeout 300 python tools/ab_rbm.py tools/ab/$s $s > $O/tune_$s.log 2>&1; done
