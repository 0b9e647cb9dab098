#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/profile_r3.sh ais aisfast > $O/profile.log 2>&1
python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py > $O/bench1.json 2> $O/bench1.err; echo "rc=$?" >> $O/bench1.err
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
