#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_fast_binary_gpu.py tests/test_full_size_gpu.py tests/test_dbm_parity_gpu.py -q -x -s --timeout 600 > $O/pytest.log 2>&1; grep -i "fast-binary AIS\|passed\|failed\|error" $O/pytest.log | tail -12
run() { env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'])" | tee -a $O/ab.log; }
run "ais f32" A=1 "--config ais --ais-betas 100 --steps 1 --warmup 1"
run "ais fast narrow(2)" BM355_BF3_GEO=2 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast wide" A=1 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast wide no-epilogue" BM355_BF3_ABL=1 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
run "ais fast wide no-kloop" BM355_BF3_ABL=2 "--config ais --ais-betas 100 --steps 1 --warmup 1 --fast-binary"
