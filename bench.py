#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: Gibbs-steps/sec of the CD-k loop,
784x1024 Bernoulli RBM, batch 512 per GPU, fp32 MFMA.

    python bench.py --gpus N --steps K --warmup W [--k 1] [--no-cpu]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one CD-k update (base_rbm.py:566 `session.run(train_op)`) of one
512-row minibatch per GPU, already resident in HBM = k Gibbs steps.
N > 1 is data-parallel, weak scaling (512 rows per GPU): each rank runs the
chain + raw outer products on its rows, ONE RCCL all-reduce(sum) of the fused
[dW|dvb|dhb|q] buffer over xGMI, then every rank applies the identical update.
value = N * k * K / t  (512-row Gibbs steps per second, whole job).

The JSON line also carries
  roofline     : fp32-MFMA roofline of the CD-k update.  achieved = algorithmic
                 GEMM flops (2k+3)*2*B*V*H per update / HIP-event time of the
                 update's kernels on the engine stream; `kernels` holds the
                 per-kernel-class durations from a second, event-instrumented pass.
  cpu_baseline : the CPU oracle (restatement of the reference maths, NOT TF1
                 itself) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V, H, B = 784, 1024, 512
LR, MOM, L2 = 0.05, 0.9, 1e-5            # examples/rbm_mnist.py:160,166,55
PEAK_FP32_MFMA = 157.3                    # TFLOP/s, MI355X_MICROARCH.md
N_BATCHES = 20                            # synthetic batches resident in HBM, cycled


def synth(seed, rows):
    from boltzmann_machines_amd.utils import philox
    u = philox.uniform(87654321, 42 + seed, 0, rows * V).reshape(rows, V)
    X = (u < 0.1307).astype(np.float32)   # MNIST mean intensity; MNIST itself is not available offline
    W = philox.tf_random_normal((V, H), 0.01, 1337)     # reference W_init (base_rbm.py:277-279)
    return X, W


def _cpu_worker(k, threads, budget_s, q):
    os.environ['OMP_NUM_THREADS'] = str(threads)
    os.environ['OMP_WAIT_POLICY'] = 'passive'
    from oracle import oracle as orc
    X, W = synth(0, B)
    twin = orc.OracleRBM(V, H, l2=L2, sample_v_states=True)
    twin.p['W'][...] = W
    twin.set_seed(1337)
    twin.train_step(X, LR, MOM, k)        # warm-up (thread pool, page faults)
    n, t0 = 0, time.time()
    while time.time() - t0 < budget_s:
        twin.train_step(X, LR, MOM, k)
        n += 1
    q.put((n, time.time() - t0))


def cpu_baseline(k, budget_s=6.0):
    """The CPU oracle (restatement of the reference maths) timed on this box's host cores, in
    fresh processes so that the OpenMP team size can be chosen; the best of three team sizes is
    reported with its thread count."""
    import multiprocessing as mp
    ncpu = os.cpu_count() or 1
    best = None
    for threads in sorted({min(16, ncpu), min(64, ncpu), ncpu}):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        p = ctx.Process(target=_cpu_worker, args=(k, threads, budget_s, q))
        p.start()
        n, dt = q.get()
        p.join()
        rate = n * k / dt
        if best is None or rate > best[0]:
            best = (rate, threads, n, dt)
    rate, threads, n, dt = best
    return {'value': round(rate, 3), 'unit': 'Gibbs-steps/s (512-row)', 'cores': threads, 'kind': 'port',
            'sample': '%d CD-%d updates of the same 784x1024 batch-512 workload in %.1f s, OpenMP C oracle '
                      '(oracle/bm_oracle.c, restatement of base_rbm.py:415-479; TF1.3 cannot run here), best of '
                      'OMP team sizes 16/64/%d' % (n, k, dt, ncpu)}


def pmc_traffic():
    """HBM-side bytes per CD-1 update from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this same command, tools/profile_r1.sh + tools/summarize_profile.py); None
    when no profile has been collected for this tree.  Counters cannot be read live from inside the
    process, so this is the one roofline field that comes from profiles/."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))['traffic_bytes_per_update'])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--k', type=int, default=1, help='n_gibbs_steps of CD-k')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--precondition-s', type=float, default=0.4,
                    help='seconds of untimed updates BEFORE the W warm-up steps (launch tuning, instruction caches, '
                         'clock ramp of an idle GPU); parameters and RNG are reset afterwards, so the warm-up and the '
                         'timed steps start from the documented initial state.  0 disables')
    ap.add_argument('--native-comm', action='store_true',
                    help='data-parallel all-reduce through the library\'s own RCCL communicator (bm_comm_*) instead of '
                         'torch.distributed; torch (gloo) then only carries the 128-byte id and the timing barrier')
    ap.add_argument('--force-dp', action='store_true',
                    help='take the data-parallel code path (grad_step -> all-reduce -> apply_step) even at N=1')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)'
                         % (args.gpus, world))

    import torch
    from boltzmann_machines_amd import _ffi
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    lib = _ffi.load()
    if lib.bm_device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    torch.cuda.set_device(local_rank)
    _ffi.check(lib.bm_set_device(local_rank))
    dist = None
    use_dp = world > 1 or args.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if args.native_comm:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    k = args.k
    X, W = synth(rank, B * N_BATCHES)
    eng = RbmEngine(V, H, max_batch=B, l2=L2, sample_v_states=True, sample_h_states=True)
    eng.set('W', W)
    eng.seed(1337)
    eng.set_row_offset(rank * B)
    Xd = as_device(X)

    if use_dp:
        from boltzmann_machines_amd import parallel
        dev = torch.device('cuda', local_rank)
        if args.native_comm:
            comm = parallel.NativeComm.from_torch_rendezvous(rank, world)
            dp = parallel.DataParallelRBM(eng, rank, world, B, parallel.native_allreduce_on_engine_stream(eng, comm))
        else:
            dp = parallel.DataParallelRBM(eng, rank, world, B, parallel.torch_allreduce_on_engine_stream(eng, dev))

        def step(i):
            dp.train_step(Xd, LR, MOM, k, row=(i % N_BATCHES) * B)     # grad_step -> RCCL all-reduce -> apply_step
    else:
        def step(i):
            eng.train_step(Xd, B, LR, MOM, k, row=(i % N_BATCHES) * B)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Precondition the device: a GPU box that has been idle needs a few hundred ms of work before its clocks and
    # caches are in the state a training job runs in (a 25-step run from cold measured 74 us/update against 66.7
    # steady).  Untimed; the model is then put back to its initial state.
    if args.precondition_s > 0:
        # a FIXED number of updates (~ precondition_s at the steady 1-GPU rate): every rank of a data-parallel run
        # must issue the same number of collectives
        for i in range(int(args.precondition_s * 12500)):
            step(i)
        eng.sync()
        for name in ('vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means'):
            eng.set(name, 0.0)
        eng.set('W', W)
        eng.seed(1337)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for i in range(args.steps):
        step(i)
    ev_ms = eng.timer_stop()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cpu' if args.native_comm else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # second, instrumented pass: per-kernel-class durations (HIP events on the engine stream)
    kern = {}
    if rank == 0:
        n_prof = min(args.steps, 200)
        eng.profile(True)
        if not use_dp:
            for i in range(n_prof):
                step(i)
        else:   # kernels only (no collective) so that other ranks need not participate
            for i in range(n_prof):
                eng.grad_step(Xd, B, k, row=(i % N_BATCHES) * B)
                eng.apply_step(B * world, LR, MOM)
        kt = eng.kernel_times()
        eng.profile(False)
        # an event pair around a launch also times its two markers.  The unbracketed update was timed above
        # (ev_ms): the markers' cost per launch is (sum of the bracketed launches - that) / launches, taken off
        # so that the per-kernel figures are comparable with rocprofv3's kernel-trace durations
        n_launch = sum(n for _, (ms, n) in kt.items() if n)
        sum_us = 1e3 * sum(ms for _, (ms, n) in kt.items() if n) / n_prof
        bracket_us = 0.0
        if not use_dp and n_launch:
            bracket_us = max(0.0, (sum_us - 1e3 * ev_ms / args.steps) / (n_launch / n_prof))
        kern = {name: {'avg_us': round(1e3 * ms / n - bracket_us, 3), 'avg_us_with_markers': round(1e3 * ms / n, 3),
                       'launches_per_step': n // n_prof}
                for name, (ms, n) in kt.items() if n}
        kern['event_pair_overhead_us'] = round(bracket_us, 3)
    barrier()

    if rank == 0:
        F = 2.0 * B * V * H
        flops_update = (2 * k + 3) * F
        ms_step = 1e3 * dt / args.steps
        achieved = flops_update / (ev_ms / args.steps * 1e-3) / 1e12
        # dominant kernel = prop-up act_kernel (k+1 launches per update): its own roofline point
        up_us = kern.get('act_up', {}).get('avg_us')
        out = {
            'metric': 'Gibbs-steps/sec (CD-k, 784x1024 RBM, batch 512)',
            'value': round(world * k * args.steps / dt, 2),
            'unit': 'Gibbs-steps/s (512-row block sweeps h->v->h incl. CD-%d update)' % k,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 5), 'higher_is_better': True, 'scaling': 'weak',
            'precondition_s': args.precondition_s,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BernoulliRBM 784x1024 CD-%d batch=512 fp32 (BASELINE configs[1])' % k,
                       'n_visible': V, 'n_hidden': H, 'batch_per_gpu': B, 'global_batch': B * world,
                       'n_gibbs_steps': k, 'sample_v_states': True, 'sample_h_states': True,
                       'parallelism': 'dp%d' % world, 'dp_path': bool(use_dp),
                       'collective': ('bm_comm (in-library RCCL)' if args.native_comm else 'torch.distributed nccl') if use_dp else None},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': PEAK_FP32_MFMA,
                         'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_FP32_MFMA, 4),
                         'traffic': pmc_traffic() if k == 1 else None,
                         'scope': 'whole CD-%d update = (2k+3)*2*B*V*H = %.3f GFLOP per launch sequence, '
                                  'HIP events on the engine stream over the timed region' % (k, flops_update / 1e9),
                         'dominant_kernel': 'act_kernel (propagation GEMM + sigmoid + Philox Bernoulli, 3 of the 4 launches)',
                         'dominant_kernel_tflops': round(F / (up_us * 1e-6) / 1e12, 3) if up_us else None,
                         'kernels': kern},
        }
        if not args.no_cpu and world == 1:
            out['cpu_baseline'] = cpu_baseline(k)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
