cd $GRAFT_REPO_ROOT
./tools/probe_mf 2>&1 | tee gpurun_out/probe_mf_new.txt | cut -c 1-200 | head -40
python -m pytest tests/test_dbm_parity_gpu.py tests/test_full_size_gpu.py tests/test_rbm_parity_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python bench.py --config dbm --no-cpu --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbm', d['ms_per_step'], d['roofline']['frac'], d['config']['mean_field_sweeps_executed'])"; done
for i in 1 2; do python bench.py --steps 2000 --warmup 100 --no-others --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rbm', d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall'])"; done
