#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/tr2.json 2> $O/tr2.err; echo "rc=$?" >> $O/tr2.err
tail -c 1500 $O/tr2.json; tail -5 $O/tr2.err
timeout 600 python -m pytest tests/test_parallel_gpu.py tests/test_rbm_parity_gpu.py -q --timeout 600 2>&1 | tail -3
