#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
BM355_MF_DEBUG=1 python tools/bench_mf.py 2 2>&1 | grep -v amdgpu.ids | tail -9 | tee $O/mf.log
python tools/bench_mf.py 30 2>&1 | grep -v amdgpu.ids | tee -a $O/mf.log
timeout 600 python -m pytest tests/test_mf_persistent_gpu.py -q -x --timeout 300 2>&1 | tail -3
