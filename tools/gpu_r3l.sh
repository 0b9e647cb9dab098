#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_rbm_parity_gpu.py tests/test_rbm_api_gpu.py -q -x --timeout 600 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { env $2 python bench.py --no-cpu --no-others $3 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['frac'], r['frac_wall'], {k:v['avg_us'] for k,v in r.get('kernels',{}).items() if isinstance(v,dict)})" | tee -a $O/ab.log; }
for r in 1 2; do
run "rbm eager" BM355_EPOCH_GRAPH=0 ""
run "rbm graph" A=1 ""
run "rbm20 eager" BM355_EPOCH_GRAPH=0 "--steps 20 --warmup 5"
run "rbm20 graph" A=1 "--steps 20 --warmup 5"
done
tail -3 $O/err.log
