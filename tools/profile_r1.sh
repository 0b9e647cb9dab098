#!/bin/bash
# Reproduces profiles/r1_*: kernel-trace stats and PMC passes of the bench command.
# Run on the GPU box from the repo root:  bash tools/profile_r1.sh
set -u
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/prof_r1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 300 --warmup 30 --no-cpu > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python $R/bench.py --steps 60 --warmup 10 --no-cpu > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python $R/bench.py --steps 60 --warmup 10 --no-cpu > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F32 --output-format csv -d $OUT/sq -o q -- python $R/bench.py --steps 60 --warmup 10 --no-cpu > $OUT/sq.log 2>&1
cd $R && python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 2500 $OUT/bench_default.json
