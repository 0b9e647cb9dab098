#!/bin/bash
# round-3 GPU pass B: A/B of the block -> tile map (BM355_XCD_MAP=8|1: the 1-D slabs of round 2) on every config,
# then driver-length runs of the headline (wall vs event fraction)
O=gpurun_out/r3b; mkdir -p $O
run() { # label, env, args
  env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['frac_wall'], {k:v['avg_us'] for k,v in r.get('kernels',{}).items() if isinstance(v,dict)})" | tee -a $O/ab.log
}
for r in 1 2; do
  for c in rbm gibbs grbm dbm; do
    run "$c old1d" BM355_XCD_MAP=8 "--config $c"
    run "$c new2d" BM355_NOP=1 "--config $c"
  done
  run "ais old1d" BM355_XCD_MAP=1 "--config ais --ais-betas 100 --steps 1 --warmup 1"
  run "ais new2d" BM355_NOP=1 "--config ais --ais-betas 100 --steps 1 --warmup 1"
done
for r in 1 2 3; do run "rbm20" BM355_NOP=1 "--steps 20 --warmup 5"; done
for r in 1 2; do run "rbm20block" BM355_HOST_WAIT=block "--steps 20 --warmup 5"; done
BM355_TUNE_LOG=1 python bench.py --no-cpu --no-others --config grbm --steps 3 --warmup 1 2>&1 | grep "bm355 tune" | tee -a $O/ab.log
