#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O
timeout 600 python -m pytest tests/test_geometries_gpu.py -q -k "5 or 7" --timeout 600 2>&1 | tail -3
BM355_TUNE_LOG=1 python bench.py --no-cpu --no-others --config ais --ais-betas 100 --steps 1 --warmup 1 2>&1 | grep "bm355 tune: act\|metric" | grep -v XCD | cut -c1-430 | tee -a $O/tune.log
BM355_TUNE_LOG=1 python bench.py --no-cpu --no-others --config dbm 2>&1 | grep "bm355 tune: act\|metric" | grep -v XCD | cut -c1-430 | tee -a $O/tune.log
