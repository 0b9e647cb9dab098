#!/bin/bash
# round-3 GPU pass A: the whole -m gpu suite, the default bench line (with other_configs), the N = 2 dry run
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py > $O/bench1.json 2> $O/bench1.err; echo "rc=$?" >> $O/bench1.err
timeout 600 python bench.py --gpus 2 --no-others --steps 200 --warmup 20 > $O/bench2.json 2> $O/bench2.err; echo "rc=$?" >> $O/bench2.err
timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 > $O/bench2o.json 2> $O/bench2o.err; echo "rc=$?" >> $O/bench2o.err
tail -3 $O/pytest.log; tail -c 600 $O/bench1.json; tail -c 400 $O/bench2.json
