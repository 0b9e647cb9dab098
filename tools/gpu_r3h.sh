#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_fast_binary_gpu.py -q -x --timeout 600 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['frac_wall'])" | tee -a $O/ab.log; }
for g in 8 2; do
  run "ais fast geo$g" BM355_BF3_GEO=$g "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
  run "gibbs fast geo$g" BM355_BF3_GEO=$g "--config gibbs --fast-binary"
done
run "ais fast auto" A=1 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
