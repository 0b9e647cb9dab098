#!/bin/bash
# Reproduces profiles/r<N>_*: rocprofv3 kernel-trace stats for every bench configuration (and the fast-binary AIS), PMC
# passes in their own runs (FETCH_SIZE / WRITE_SIZE separately, one SQ pass) of the bench command.
# Run on the GPU box from the repo root:  ROUND=r5 bash tools/profile.sh [configs...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_${ROUND:-r5}
mkdir -p $OUT
CONFIGS=${@:-rbm gibbs grbm dbm ais aisfast}
# which library the counters belong to (bench.py refuses a traffic figure whose sources differ from the running tree's)
(cd $R && python -c 'import bench; print(bench.kernel_source_sha16())') > $OUT/source_sha16.txt
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_BF16"
for c in $CONFIGS; do
  X=""; cc=$c
  case $c in
    rbm)   A="--steps 300 --warmup 30"; P="--steps 60 --warmup 10";;
    gibbs) A="--steps 60 --warmup 10";  P="--steps 20 --warmup 5";;
    grbm)  A="--steps 10 --warmup 3";   P="--steps 4 --warmup 2";;
    dbm)   A="--steps 10 --warmup 3";   P="--steps 4 --warmup 2";;
    ais)   A="--steps 1 --warmup 1 --ais-betas 60"; P="--steps 1 --warmup 0 --ais-betas 20";;
    grbmfast) cc=grbm; X="--fast-binary"; A="--steps 10 --warmup 3"; P="--steps 4 --warmup 2";;
    aisfast) cc=ais; X="--fast-binary"; A="--steps 1 --warmup 1 --ais-betas 60"; P="--steps 1 --warmup 0 --ais-betas 20";;
  esac
  B="python $R/bench.py --config $cc $X --no-cpu --no-others --precondition-s 0.1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/stats -o s -- $B $A > $OUT/$c.stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$c/fetch -o f -- $B $P > $OUT/$c.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$c/write -o w -- $B $P > $OUT/$c.write.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $OUT/$c/sq -o q -- $B $P > $OUT/$c.sq.log 2>&1
  # keep only what the summary needs (the merged-back directory is capped at 64 MiB)
  find $OUT/$c -name '*_kernel_trace.csv' -delete; find $OUT/$c -name '*.db' -delete
  (cd $R && python bench.py --config $cc $X --no-cpu --no-others > $OUT/$c.bench.json 2> $OUT/$c.bench.err)
  tail -c 300 $OUT/$c.bench.json; echo
  ls -la $OUT/$c/sq/* 2>/dev/null | head -3
done
