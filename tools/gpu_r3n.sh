#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O
run() { env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['frac_wall'])" | tee -a $O/ab.log; }
for r in 1 2 3; do
run "rbm20 default" A=1 "--steps 20 --warmup 5"
run "rbm20 activewait" ROC_ACTIVE_WAIT_TIMEOUT=1000000 "--steps 20 --warmup 5"
done
