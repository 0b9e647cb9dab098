#!/bin/bash
# repeated 2-rank dry runs of the data-parallel bench paths on ONE device (developer tool): catches intermittent exchange
# time-outs; per-rank progress lines in gpurun_out/dryrun_<config>_<n>.err
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in ${RUNS:-1 2 3}; do
  for cfg in dbm rbm; do
    BM_BENCH_TRACE=1 BM_XCHG_TIMEOUT_S=6 timeout 120 python bench.py --config $cfg --gpus 2 --steps 20 --warmup 5 --no-cpu --no-others \
      2>gpurun_out/dryrun_${cfg}_$n.err | tail -1 > gpurun_out/dryrun_${cfg}_$n.json
    echo "$cfg run $n: rc=$? $(grep -c 'expired' gpurun_out/dryrun_${cfg}_$n.err) expired-lines; $(python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/dryrun_${cfg}_$n.json')); print(d['ms_per_step'], d['config'].get('data_parallel_check'))
except Exception as e: print('no json')")"
  done
done
