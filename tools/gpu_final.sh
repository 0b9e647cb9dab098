#!/bin/bash
# the round-end sequence on one box: profiles of one configuration (optional), the whole -m gpu suite, smoke, the
# default bench line and a driver-length one.   usage: bash tools/gpu_final.sh [configs to re-profile...]
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
cd $GRAFT_REPO_ROOT
if [ $# -gt 0 ]; then bash tools/profile_r3.sh "$@" > $O/profile.log 2>&1; fi
python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench1.json 2> $O/bench1.err; echo "rc=$?" >> $O/bench1.err
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
