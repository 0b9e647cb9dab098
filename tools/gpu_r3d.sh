#!/bin/bash
# round-3 GPU pass D: fast-binary mode - tests, then gibbs / ais with and without it
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_fast_binary_gpu.py -q -x -s --timeout 600 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
run() { env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['frac_wall'])" | tee -a $O/ab.log; }
run "gibbs f32" A=1 "--config gibbs"
run "gibbs fast" A=1 "--config gibbs --fast-binary"
run "ais f32" A=1 "--config ais --ais-betas 100 --steps 1 --warmup 1"
run "ais fast" A=1 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast geo4" BM355_BF3_GEO=4 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
tail -5 $O/err.log
