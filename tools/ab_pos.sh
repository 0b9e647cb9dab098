set -u
cd $GRAFT_REPO_ROOT
BM355_DEBUG=pos_overlap=1 python -m pytest tests/test_rbm_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
for m in 0 1; do
  echo "pos_overlap=$m"; BM355_DEBUG=pos_overlap=$m python bench.py --steps 2000 --warmup 100 --no-others --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall'], {k:(v.get('avg_us') if isinstance(v,dict) else v) for k,v in d['roofline'].get('kernels',{}).items()})"
done; done
mkdir -p gpurun_out/pos
cd /tmp && export TMPDIR=/tmp
BM355_DEBUG=pos_overlap=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pos/on -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-others --no-cpu --precondition-s 0.1 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/pos -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c "head -8 {} | cut -c 1-200"
find $GRAFT_REPO_ROOT/gpurun_out/pos -name '*_kernel_trace.csv' -size +40M -delete
