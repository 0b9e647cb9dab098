#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/chain_diag; mkdir -p $O
cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'])"; }
for c in gibbs rbm; do
for dbg in ${DBGS:-0 4 2}; do
  BM355_CHAIN=1 BM355_CHAIN_DBG=$dbg timeout 200 python bench.py --config $c --no-cpu --no-others 2> $O/${c}_dbg$dbg.err | line "$c chain=1 dbg=$dbg"
done
BM355_CHAIN=0 timeout 200 python bench.py --config $c --no-cpu --no-others 2> $O/${c}_off.err | line "$c chain=0"
done
for dbg in 0; do
  BM355_CHAIN=1 BM355_CHAIN_DBG=$dbg BM355_CHAIN_STAMPS=$O/stamps_$dbg.bin timeout 200 python bench.py --config gibbs --no-cpu --no-others --steps 20 --warmup 5 --precondition-s 0.05 2> $O/g_st$dbg.err | line "gibbs stamps dbg=$dbg"
  python tools/chain_timeline.py $O/stamps_$dbg.bin 0
  BM355_CHAIN=1 BM355_CHAIN_DBG=$dbg BM355_CHAIN_STAMPS=$O/stamps_r$dbg.bin timeout 200 python bench.py --config rbm --no-cpu --no-others --steps 20 --warmup 5 --precondition-s 0.05 2> $O/r_st$dbg.err | line "rbm stamps dbg=$dbg"
  python tools/chain_timeline.py $O/stamps_r$dbg.bin 3
done
