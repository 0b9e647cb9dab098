#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
python tools/step_times.py 2>&1 | grep -v amdgpu.ids | tee $O/step_times.log
