#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_parallel_gpu.py -q --timeout 600 2>&1 | tail -3
timeout 600 python bench.py --gpus 2 --steps 30 --warmup 5 --no-others > $O/b2.json 2> $O/b2.err; echo rc=$?; tail -c 400 $O/b2.json
