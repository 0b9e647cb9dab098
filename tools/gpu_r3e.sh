#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-others --precondition-s 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/aisfast -o s -- $B --config ais --ais-betas 30 --steps 1 --warmup 1 --fast-binary > $O/aisfast.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gibbsfast -o s -- $B --config gibbs --steps 50 --warmup 10 --fast-binary > $O/gibbsfast.log 2>&1
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*.db' -delete
for d in aisfast gibbsfast; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -12 $f | cut -c1-220; done
rocprofv3 -L > $O/counters.txt 2>&1; grep -c . $O/counters.txt
