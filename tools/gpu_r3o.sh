#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3o; mkdir -p $O
run() { env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'])" | tee -a $O/ab.log; }
run "ais fast full" A=1 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast no-epilogue" BM355_BF3_ABL=1 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast no-kloop" BM355_BF3_ABL=2 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast neither" BM355_BF3_ABL=3 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
