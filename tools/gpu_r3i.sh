#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O
timeout 600 python -m pytest tests/test_mf_persistent_gpu.py -q -x --timeout 300 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
run() { env $2 timeout 300 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['frac_wall'], d['config'].get('mean_field_sweeps_executed'))" | tee -a $O/ab.log; }
run "dbm per-layer" BM355_MF_PERSIST=0 "--config dbm"
run "dbm persistent" A=1 "--config dbm"
run "dbm persistent no-overlap" BM355_DBM_OVERLAP=0 "--config dbm"
grep -i "gave up" $O/err.log | head -3
