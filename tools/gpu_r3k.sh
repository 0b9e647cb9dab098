#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python bench.py > $O/bench1.json 2> $O/bench1.err; echo "rc=$?" >> $O/bench1.err
for r in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench20_$r.json 2> $O/bench20_$r.err; done
timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --others-budget-s 200 > $O/bench2.json 2> $O/bench2.err; echo "rc=$?" >> $O/bench2.err
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
