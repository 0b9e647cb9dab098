#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: Gibbs-steps/sec of the CD-k loop,
784x1024 Bernoulli RBM, batch 512 per GPU, fp32 MFMA — plus the other BASELINE configurations.

    python bench.py --gpus N --steps K --warmup W [--config rbm|gibbs|grbm|dbm|ais] [--k 1] [--no-cpu]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

--config rbm (default) is BASELINE.json's metric on configs[1]: one "step" = one CD-k update
(base_rbm.py:566 `session.run(train_op)`) of one 512-row minibatch per GPU, already resident in HBM
= k Gibbs steps.  N > 1 is data-parallel, weak scaling (512 rows per GPU): each rank
runs the chain + raw outer products on its rows, ONE RCCL all-reduce(sum) of the fused
[dW|dvb|dhb|q] buffer over xGMI, then every rank applies the identical update.
value = N * k * K / t  (512-row Gibbs steps per second, whole job).

Other configurations (one JSON line each, same contract; they are the driver-reproducible form of the
numbers DESIGN.md quotes for BASELINE configs[2..4] and of SURVEY §8d(ii)):
  gibbs : pure sampling sweep (bm_rbm_gibbs, base_rbm.py:367-413) on the same 784x1024, batch 512 RBM;
          a step = 10 (or --k) block-Gibbs sweeps h->v->h in one call; reports MFMA fraction AND algorithmic GB/s.
  grbm  : configs[2] Gaussian-Bernoulli RBM 3072x5000, PCD-5, batch 256 (1-layer DBM path, README.md:96).
  dbm   : configs[3] DBM 784-512-1024, 512 rows + 512 particles per GPU, <= 50 mean-field sweeps, PCD-5;
          N > 1: data-parallel through the library's own RCCL communicator (bm_comm_*).
  ais   : configs[4] AIS log Z, 20 000 chains x 1000 betas; a step = one full run; N > 1 shards the chains
          (strong scaling) and ends with ONE all-gather of the log-weights.

The JSON line carries
  roofline     : fp32-MFMA roofline.  achieved = algorithmic GEMM flops per step (SURVEY §8d formulas) /
                 HIP-event time of the step's kernels on the engine stream; `kernels` holds the
                 per-kernel-class durations from a second, event-instrumented pass (rbm only).
  cpu_baseline : CPU restatements of the reference maths (NOT TF1 itself, which cannot run here) timed on
                 this box's host cores on a bounded sample: the canonical-order OpenMP C oracle and a
                 NumPy + BLAS (sgemm) restatement; CPU model, core count and threads are stated.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Host waits spin instead of sleeping (ROCclr polls for ROC_ACTIVE_WAIT_TIMEOUT microseconds before it blocks on an
# interrupt; read when the runtime initialises, hence set before anything imports it): the barrier that ends the timed
# region otherwise returns tens of microseconds after the last kernel - 1 % of a 20-update region.  Same policy as
# bm_set_device's hipDeviceScheduleSpin.
os.environ.setdefault('ROC_ACTIVE_WAIT_TIMEOUT', '1000000')

V, H, B = 784, 1024, 512
LR, MOM, L2 = 0.05, 0.9, 1e-5            # examples/rbm_mnist.py:160,166,55
PEAK_FP32_MFMA = 157.3                    # TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM = 8000.0                         # GB/s (spec), MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2500.0                   # TFLOP/s dense, MI355X_MICROARCH.md
PEAK_BF16X3 = PEAK_BF16_MFMA / 3.0        # an exact-product fp32 weight = three bf16 planes = three MFMA passes per flop
N_BATCHES = 20                            # synthetic batches resident in HBM, cycled


def synth(seed, rows, v=V, h=H):
    from boltzmann_machines_amd.utils import philox
    u = philox.uniform(87654321, 42 + seed, 0, rows * v).reshape(rows, v)
    X = (u < 0.1307).astype(np.float32)   # MNIST mean intensity; MNIST itself is not available offline
    W = philox.tf_random_normal((v, h), 0.01, 1337)     # reference W_init (base_rbm.py:277-279)
    return X, W


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_info():
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def _cpu_worker(kind, v, h, b, k, threads, budget_s, q):
    os.environ['OMP_NUM_THREADS'] = str(threads)
    os.environ['OMP_WAIT_POLICY'] = 'passive'
    # numpy (and its OpenBLAS pool) is already loaded when a spawned worker starts (it imports this module):
    # the pool size is set at run time
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=threads)
    except Exception:
        pass
    X, W = synth(0, b, v, h)
    if kind == 'oracle':
        from oracle import oracle as orc
        twin = orc.OracleRBM(v, h, l2=L2, sample_v_states=True)
        twin.p['W'][...] = W
        twin.set_seed(1337)
        step = lambda: twin.train_step(X, LR, MOM, k)
    else:
        from oracle import cpu_blas
        m = cpu_blas.BlasRBM(W, l2=L2, sample_v=True, sample_h=True, seed=1337)
        step = lambda: m.train_step(X, LR, MOM, k)
    step()                                # warm-up (thread pool, page faults)
    n, t0 = 0, time.time()
    while time.time() - t0 < budget_s:
        step()
        n += 1
    q.put((n, time.time() - t0))


def _cpu_rate(kind, v, h, b, k, threads, budget_s):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_worker, args=(kind, v, h, b, k, threads, budget_s, q))
    p.start()
    n, dt = q.get()
    p.join()
    return n * k / dt, n, dt


def cpu_baseline(k, budget_s=2.5):
    """CPU restatements of the reference maths timed on this box's host cores, each in a fresh process
    so that the thread-team size can be chosen.  Two restatements: (a) the canonical-order OpenMP C
    oracle (oracle/bm_oracle.c: the parity checker; sequential fma chains, so it cannot use a blocked
    sgemm) and (b) NumPy float32 + the BLAS numpy links (sgemm for the five GEMMs, oracle/cpu_blas.py).
    `value` is the faster of the two at the north-star shape; both, and the cfg1 shape (784x128, batch
    100: the shape BASELINE.json assigns to the reference's CPU path), are listed in `detail`."""
    model, ncpu = cpu_info()
    teams = sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)})
    detail = {}
    best = None
    for name, (v, h, b) in (('784x1024_b512', (V, H, B)), ('784x128_b100', (784, 128, 100))):
        for kind in ('oracle', 'blas'):
            top = None
            for threads in (teams if name == '784x1024_b512' else teams[:3]):
                rate, n, dt = _cpu_rate(kind, v, h, b, k, threads, budget_s)
                if top is None or rate > top[0]:
                    top = (rate, threads, n, dt)
            detail['%s_%s' % (name, kind)] = {'gibbs_steps_per_s': round(top[0], 2), 'threads': top[1],
                                              'updates_timed': top[2], 'seconds': round(top[3], 2)}
            if name == '784x1024_b512' and (best is None or top[0] > best[0]):
                best = top + (kind,)
    rate, threads, n, dt, kind = best
    try:
        blas = np.__config__.show(mode='dicts')['Build Dependencies']['blas']
        blas = '%s %s' % (blas.get('name'), blas.get('version'))
    except Exception:
        blas = 'unknown'
    return {'value': round(rate, 3), 'unit': 'Gibbs-steps/s (512-row)', 'cores': threads, 'kind': 'port',
            'cpu_model': model, 'nproc': ncpu, 'blas': blas, 'restatement': kind,
            'sample': '%d CD-%d updates of the same 784x1024 batch-512 workload in %.1f s with the %s restatement '
                      '(best of the OpenMP C oracle and NumPy+BLAS, thread teams %s); TF1.3 / python2 cannot run here'
                      % (n, k, dt, kind, '/'.join(str(t) for t in teams)),
            'detail': detail}


def kernel_source_sha16():
    """identifies the library a profile was taken on: sha256 over the sources of the shared library (csrc/, include/).  (The
    GPU box receives a snapshot without .git, so a commit id cannot be read there; tools/summarize_profile.py adds the
    commit whose tree has this hash when it files the profile.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'boltzmann_machines_amd', 'csrc', '*.h')) +
                   glob.glob(os.path.join(ROOT, 'boltzmann_machines_amd', 'csrc', '*.hip')) +
                   glob.glob(os.path.join(ROOT, 'include', '*.h')))
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0' + open(f, 'rb').read() + b'\0')
    return h.hexdigest()[:16]


def pmc_traffic(config):
    """HBM-side bytes per step from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    runs of this same command, tools/profile.sh + tools/summarize_profile.py); None when no profile has
    been collected for this configuration.  Counters cannot be read live from inside the process, so this
    is the one roofline field that comes from profiles/ - and ONLY from a profile taken on the library that is
    running: a file whose recorded source hash differs from this tree's yields `traffic: null` and says so."""
    import glob
    pats = ['r*_%s_pmc.json' % config] + (['r[0-9]_pmc.json'] if config == 'rbm' else [])
    files = []
    for p in pats:
        files += glob.glob(os.path.join(ROOT, 'profiles', p))
    files = sorted(files)
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        name = 'profiles/' + os.path.basename(files[-1])
        have, mine = d.get('source_sha16'), kernel_source_sha16()
        if have != mine:
            return (None, '%s is STALE: taken on library sources %s (commit %s), running %s - re-collect with tools/profile.sh'
                    % (name, have or 'unrecorded', d.get('head', 'unrecorded'), mine))
        if d.get('traffic_bytes_per_update') is None:
            return (None, '%s holds per-dispatch counters only: the counter passes of this configuration are dominated by the launch '
                          "tuner's candidate launches (tools/summarize_profile.py)" % name)
        return (float(d['traffic_bytes_per_update']),
                '%s (rocprofv3 --pmc passes of this command on these library sources, %s / commit %s; not measured in this run)'
                % (name, mine, d.get('head', '?')))
    except Exception:
        return None


# ------------------------------------------------------------------------------------ workloads
class Workload(object):
    """One benchmark configuration.  step(i) enqueues one step on the engine stream."""
    name = ''
    scaling = 'weak'

    def precondition_steps(self, seconds):
        return 0

    def reset(self):
        pass

    def run_steps(self, i0, n):
        """steps i0 .. i0+n-1; a workload may hand runs of steps to the engine in one call"""
        for i in range(i0, i0 + n):
            self.step(i)


class RbmCD(Workload):
    name = 'rbm'

    def __init__(self, args, rank, world, local_rank, dist):
        import torch
        from boltzmann_machines_amd.engine import RbmEngine, as_device
        self.k, self.world, self.rank = args.k, world, rank
        self.X, self.W = synth(rank, B * N_BATCHES)
        self.eng = eng = RbmEngine(V, H, max_batch=B, l2=L2, sample_v_states=True, sample_h_states=True)
        eng.set('W', self.W)
        eng.seed(1337)
        eng.set_row_offset(rank * B)
        self.Xd = as_device(self.X)
        # the initial state, resident: reset() restores it with device-to-device copies in stream order, so the GPU does
        # not idle (and clock down) between the untimed precondition steps and the warm-up steps
        self.init = {'W': as_device(self.W)}
        for name, n in (('vb', V), ('hb', H), ('dvb', V), ('dhb', H), ('q_means', H)):
            self.init[name] = as_device(np.zeros(n, dtype=np.float32))
        self.init['dW'] = as_device(np.zeros((V, H), dtype=np.float32))
        self.use_dp = world > 1 or args.force_dp
        self.collective, self.collective_note = None, None
        if self.use_dp:
            from boltzmann_machines_amd import parallel
            if args.delayed_grads:      # NON-parity mode: the reduction of step t runs under step t+1
                self.comm = get_comm(rank, world)
                self.dp = parallel.DelayedDataParallelRBM(eng, rank, world, B, comm=self.comm)
                self.collective = 'rccl (bm_comm), delayed'
                return
            mode, note = choose_collective(args, eng, rank, world, dist)
            self.collective, self.collective_note = mode, note
            self.reset()                 # (the start-up race ran the fused exchange on this engine's parameters)
            fused = None
            if mode == 'direct':
                ar = parallel.direct_allreduce_on_engine_stream(eng, args._xchg[id(eng)])
                if not args.no_fused_exchange:      # exchange + update in ONE kernel per rank (bm_rbm_exchange_apply_direct)
                    fused = args._xchg[id(eng)]
                    self.collective_note = (note or '') + '; reduce-scatter -> update of the owned slice -> all-gather of W fused in one launch'
            elif mode == 'rccl':
                self.comm = get_comm(rank, world)
                ar = parallel.native_allreduce_on_engine_stream(eng, self.comm)
            elif mode == 'torch':   # RCCL through torch.distributed (a nccl group next to the gloo default group)
                dev = torch.device('cuda', local_rank)
                ar = parallel.torch_allreduce_on_engine_stream(eng, dev, group=dist.new_group(backend='nccl'))
            else:                   # 'gloo': staged through the host (no device collective usable)
                ar = gloo_staged_allreduce(eng, dist)
            self.dp = parallel.DataParallelRBM(eng, rank, world, B, ar, fused=fused)

    def step(self, i):
        if self.use_dp:
            self.dp.train_step(self.Xd, LR, MOM, self.k, row=(i % N_BATCHES) * B)   # grad_step -> all-reduce -> apply_step
        else:
            self.eng.train_step(self.Xd, B, LR, MOM, self.k, row=(i % N_BATCHES) * B)

    def run_steps(self, i0, n):
        # consecutive minibatches go to the engine as ONE call, as BaseRBM.fit() hands them over (bm_rbm_train_epoch
        # loops over the batches in C: the same launches and RNG call counters as n train_step calls, no Python between)
        if self.use_dp:
            return Workload.run_steps(self, i0, n)
        i, end = i0, i0 + n
        while i < end:
            b0 = i % N_BATCHES
            m = min(end - i, N_BATCHES - b0)
            self.eng.train_epoch(self.Xd, m * B, B, LR, MOM, self.k, row=b0 * B)
            i += m

    def precondition_steps(self, seconds):
        return int(seconds * 12500)

    def reset(self):
        if self.use_dp and hasattr(getattr(self, 'dp', None), 'flush'):
            self.dp.flush()                      # delayed-gradient mode: nothing in flight across the reset
        for name, d in self.init.items():
            self.eng.set_from_device(name, d)
        self.eng.seed(1337)

    def kernel_pass(self, steps, ev_ms):
        """second, instrumented pass: per-kernel-class durations (HIP events on the engine stream)"""
        eng, k = self.eng, self.k
        n_prof = min(steps, 200)
        eng.profile(True)
        if not self.use_dp:
            for i in range(n_prof):
                self.step(i)
        else:   # kernels only (no collective) so that other ranks need not participate
            for i in range(n_prof):
                eng.grad_step(self.Xd, B, k, row=(i % N_BATCHES) * B)
                eng.apply_step(B * self.world, LR, MOM)
        kt = eng.kernel_times()
        eng.profile(False)
        # an event pair around a launch also times its two markers.  The unbracketed update was timed
        # (ev_ms): the markers' cost per launch is (sum of the bracketed launches - that) / launches, taken off
        # so that the per-kernel figures are comparable with rocprofv3's kernel-trace durations
        n_launch = sum(n for _, (ms, n) in kt.items() if n)
        sum_us = 1e3 * sum(ms for _, (ms, n) in kt.items() if n) / n_prof
        bracket_us = 0.0
        if not self.use_dp and n_launch:
            bracket_us = max(0.0, (sum_us - 1e3 * ev_ms / steps) / (n_launch / n_prof))
        kern = {name: {'avg_us': round(1e3 * ms / n - bracket_us, 3), 'avg_us_with_markers': round(1e3 * ms / n, 3),
                       'launches_per_step': n // n_prof}
                for name, (ms, n) in kt.items() if n}
        kern['event_pair_overhead_us'] = round(bracket_us, 3)
        return kern

    def report(self, args, world, dt, ev_ms):
        k = self.k
        F = 2.0 * B * V * H
        flops = (2 * k + 3) * F
        kern = self.kernel_pass(args.steps, ev_ms)
        up_us = kern.get('act_up', {}).get('avg_us')
        return {
            'metric': 'Gibbs-steps/sec (CD-k, 784x1024 RBM, batch 512)',
            'value': round(world * k * args.steps / dt, 2),
            'unit': 'Gibbs-steps/s (512-row block sweeps h->v->h incl. CD-%d update)' % k,
            'config': {'workload': 'BernoulliRBM 784x1024 CD-%d batch=512 fp32 (BASELINE configs[1])' % k,
                       'n_visible': V, 'n_hidden': H, 'batch_per_gpu': B, 'global_batch': B * world,
                       'n_gibbs_steps': k, 'sample_v_states': True, 'sample_h_states': True,
                       'parallelism': 'dp%d' % world, 'dp_path': bool(self.use_dp),
                       'collective': COLLECTIVE_NAMES.get(self.collective, self.collective) if self.use_dp else None,
                       'collective_note': self.collective_note,
                       'gradient_delay_steps': 1 if (self.use_dp and getattr(args, 'delayed_grads', False)) else 0},
            'flops_per_step': flops,
            'roofline_extra': {
                'traffic': pmc_traffic('rbm') if k == 1 else None,
                'scope': 'whole CD-%d update = (2k+3)*2*B*V*H = %.3f GFLOP per launch sequence, '
                         'HIP events on the engine stream over the timed region' % (k, flops / 1e9),
                'dominant_kernel': 'act_kernel (propagation GEMM + sigmoid + Philox Bernoulli, 3 of the 4 launches)',
                'dominant_kernel_tflops': round(F / (up_us * 1e-6) / 1e12, 3) if up_us else None,
                'kernels': kern},
        }


class RbmGibbs(Workload):
    """pure sampling sweep, SURVEY §8d(ii): bm_rbm_gibbs (base_rbm.py:367-413), no parameter update"""
    name = 'gibbs'

    def __init__(self, args, rank, world, local_rank, dist):
        from boltzmann_machines_amd._ffi import DeviceArray
        from boltzmann_machines_amd.engine import RbmEngine
        from boltzmann_machines_amd.utils import philox
        self.k = args.k if args.k > 1 else 10          # sweeps per call
        _, W = synth(rank, 4)
        self.eng = eng = RbmEngine(V, H, max_batch=B, sample_v_states=True, sample_h_states=True)
        eng.set('W', W)
        eng.seed(1337)
        eng.set_row_offset(rank * B)
        self.fast = bool(getattr(args, 'fast_binary', False))
        eng.set_fast_binary(self.fast, everywhere=True)     # (level 2: at this shape the default switch, level 1, leaves the fp32 path on - it is faster)
        h0 = (philox.uniform(87654321, 7 + rank, 0, B * H) < 0.5).astype(np.float32).reshape(B, H)
        self.Hd = DeviceArray.from_numpy(h0)
        self.Vd = DeviceArray((B, V))

    def step(self, i):
        self.eng.gibbs(self.Hd, self.Vd, B, self.k)

    def precondition_steps(self, seconds):
        return int(seconds * 30000 / self.k)

    def _launch_note(self):
        n, _, mode = self.eng.chain_stats()
        if n:
            return ('chained: the %d passes of a call are workgroups of ONE launch that hand their rows over inside an '
                    "XCD's L2 (csrc/bm_chain.h; BM355_DEBUG=chain=%d; bit-identical to the per-pass launches)" % (2 * self.k, mode))
        return 'one launch per pass'

    def report(self, args, world, dt, ev_ms):
        k = self.k
        F = 2.0 * B * V * H
        flops = k * 2 * F
        # algorithmic bytes of one block sweep: W read twice + states written and read once each way
        bytes_sweep = 2 * 4.0 * V * H + 2 * 4.0 * B * (V + H)
        gbps = k * bytes_sweep / (ev_ms / args.steps * 1e-3) / 1e9
        return {
            'metric': 'Gibbs-steps/sec (pure block-Gibbs sampling sweep, 784x1024 RBM, batch 512)',
            'value': round(world * k * args.steps / dt, 2),
            'unit': 'Gibbs-steps/s (512-row block sweeps h->v->h, sampling both ways, no update)',
            'config': {'workload': 'BernoulliRBM 784x1024 sampling sweep batch=512 fp32 (SURVEY 8d-ii, bm_rbm_gibbs)',
                       'n_visible': V, 'n_hidden': H, 'batch_per_gpu': B, 'sweeps_per_call': k,
                       'parallelism': 'replicas%d' % world, 'fast_binary': FAST_NOTE if self.fast else False,
                       'launches': self._launch_note()},
            'flops_per_step': flops,
            'roofline_extra': {
                'bf16x3_flop_fraction': 1.0 if self.fast else 0.0,
                'traffic_per_sweep_over_algorithmic': (round(pmc_traffic('gibbs')[0] / k / bytes_sweep, 2)
                                                       if pmc_traffic('gibbs') and pmc_traffic('gibbs')[0] else None),
                'scope': '%d sweeps per call, 2*2*B*V*H = %.3f GFLOP per sweep; the sweep is MFMA-bound (the 6.4 MB of W '
                         'and the 3.7 MB of states are L2 / Infinity-Cache resident), the HBM figure is the secondary '
                         'number north_star asks for' % (k, 2 * F / 1e9),
                'hbm': {'bound': 'hbm', 'achieved': round(gbps, 1), 'peak': PEAK_HBM, 'unit': 'GB/s',
                        'frac': round(gbps / PEAK_HBM, 4),
                        'algorithmic_bytes_per_sweep': bytes_sweep,
                        'traffic_per_sweep': (round(pmc_traffic('gibbs')[0] / k, 1)
                                              if pmc_traffic('gibbs') and pmc_traffic('gibbs')[0] else None)}},
        }


class _DbmBase(Workload):
    def _dp_setup(self, args, rank, world, dist):
        self.comm, self.collective, self.collective_note = None, None, None
        if world > 1 or args.force_dp:
            from boltzmann_machines_amd import parallel
            mode, note = choose_collective(args, self.eng, rank, world, dist)
            self.collective, self.collective_note = mode, note
            if mode == 'direct':
                x = args._xchg[id(self.eng)]
                self.comm = x
                fused = None
                if not args.no_fused_exchange and x.fused_ok():
                    # exchange + update fused with column-sliced ownership (bm_dbm_exchange_apply_direct): the max-norm
                    # rescale works on whole columns, so a rank owns columns of every W_i
                    fused = x
                    self.collective_note = (note or '') + ('; reduce-scatter of the owned COLUMNS -> update + max-norm on them -> '
                                                           'all-gather of W / W^T / norms (two launches around the max-norm kernels)')
                self.dp = parallel.DataParallelDBM(self.eng, rank, world,
                                                   parallel.direct_allreduce_on_engine_stream(self.eng, x), xchg=x, fused=fused)
            elif mode == 'gloo':
                self.comm = 'gloo'
                import torch

                def amax(v):
                    t = torch.tensor([v], dtype=torch.float32)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    return float(t.item())
                self.dp = parallel.DataParallelDBM(self.eng, rank, world, gloo_staged_allreduce(self.eng, dist),
                                                   allreduce_max=amax)
            else:
                self.comm = get_comm(rank, world)
                self.dp = parallel.DataParallelDBM(self.eng, rank, world,
                                                   parallel.native_allreduce_on_engine_stream(self.eng, self.comm), comm=self.comm)


class Grbm(_DbmBase):
    """BASELINE configs[2]: Gaussian-Bernoulli RBM 3072x5000, PCD-5, batch 256 (examples/dbm_cifar_naive.py:83-98;
    the reference RBM class has no PCD: realised as the 1-layer DBM path, README.md:96)"""
    name = 'grbm'
    GV, GH, GN, GK = 3072, 5000, 256, 5

    def __init__(self, args, rank, world, local_rank, dist):
        from boltzmann_machines_amd.engine import DbmEngine, as_device
        from boltzmann_machines_amd.utils import philox
        V_, H_, N_ = self.GV, self.GH, self.GN
        self.world = world
        self.eng = eng = DbmEngine(V_, [H_], v_unit=1, sample_v_states=True, n_particles=N_, batch_size=N_,
                                   max_mf_updates=1, l2=0.01)
        eng.set('W', philox.tf_random_normal((V_, H_), 0.0008, 1337))
        eng.set('v', philox.normal(1, 1 + rank, 0, N_ * V_).reshape(N_, V_))
        X = philox.normal(1, 100 + rank, 0, 4 * N_ * V_).reshape(4 * N_, V_).astype(np.float32)
        self.Xd = as_device(X)
        eng.seed(1)
        self.fast = bool(getattr(args, 'fast_binary', False))
        eng.set_fast_binary(self.fast)
        self.nmf = []
        self._dp_setup(args, rank, world, dist)

    def step(self, i):
        row = (i % 4) * self.GN
        if self.comm is not None:
            self.nmf.append(self.dp.train_step(self.Xd, 5e-4, 0.9, self.GK, row=row))
        else:
            self.nmf.append(self.eng.train_step(self.Xd, 5e-4, 0.9, self.GK, row=row)[0])

    def precondition_steps(self, seconds):
        return int(seconds * 500)

    def reset(self):
        self.nmf = []

    def report(self, args, world, dt, ev_ms):
        F = 2.0 * self.GN * self.GV * self.GH
        T = float(np.mean(self.nmf[-args.steps:])) if self.nmf else 0.0
        flops = (2 * self.GK + 3) * F          # prop-up of data 1F, 5 sweeps 10F, 2 outer products 2F
        return {
            'metric': 'Gibbs-steps/sec (PCD-5, 3072x5000 Gaussian-Bernoulli RBM, batch 256)',
            'value': round(world * self.GK * args.steps / dt, 2),
            'unit': 'Gibbs-steps/s (256-particle block sweeps incl. the PCD-5 update)',
            'config': {'workload': 'Gaussian-Bernoulli RBM 3072x5000 PCD-5 batch=256 fp32 (BASELINE configs[2])',
                       'n_visible': self.GV, 'n_hidden': self.GH, 'batch_per_gpu': self.GN, 'n_particles_per_gpu': self.GN,
                       'n_gibbs_steps': self.GK, 'mean_field_sweeps_executed': T, 'parallelism': 'dp%d' % world,
                       'collective': COLLECTIVE_NAMES.get(self.collective, self.collective),
                       'fast_binary': (FAST_NOTE + ' Here: the 5 top-down (h -> v) contractions of the PCD sweeps; the '
                                       'bottom-up ones read real-valued visibles and stay fp32, as do the data pass and '
                                       'the outer products.') if self.fast else False},
            'flops_per_step': flops,
            'roofline_extra': {'bf16x3_flop_fraction': (self.GK / (2.0 * self.GK + 3.0)) if self.fast else 0.0,
                               'scope': 'whole PCD-5 update = (2*5+3)*2*B*V*H = %.1f GFLOP (SURVEY 8d)' % (flops / 1e9),
                               'traffic': pmc_traffic('grbm')},
        }


class Dbm(_DbmBase):
    """BASELINE configs[3]: 2-layer DBM 784-512-1024, mean-field + PCD-5 (examples/dbm_mnist.py:250-284)"""
    name = 'dbm'
    DV, H1, H2, DN, DK = 784, 512, 1024, 512, 5

    def __init__(self, args, rank, world, local_rank, dist):
        from boltzmann_machines_amd.engine import DbmEngine, as_device
        from boltzmann_machines_amd.utils import philox
        V_, N_ = self.DV, self.DN
        self.world = world
        self.eng = eng = DbmEngine(V_, [self.H1, self.H2], n_particles=N_, batch_size=N_, max_mf_updates=50, mf_tol=float(getattr(args, 'dbm_mf_tol', 1e-7)),
                                   l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1], sparsity_cost=[1e-4, 5e-5])
        eng.set('W', philox.tf_random_normal((V_, self.H1), 0.01, 1337))
        eng.set('W_1', philox.tf_random_normal((self.H1, self.H2), 0.01, 1111))
        eng.set('v', (philox.uniform(1, 1 + rank, 0, N_ * V_) < 0.13).reshape(N_, V_))
        X = (philox.uniform(1, 200 + rank, 0, 4 * N_ * V_) < 0.13).astype(np.float32).reshape(4 * N_, V_)
        self.Xd = as_device(X)
        eng.seed(1)
        self.fast = bool(getattr(args, 'fast_binary', False))
        eng.set_fast_binary(self.fast, everywhere=True)     # (level 2: at this shape the default switch, level 1, leaves the fp32 path on - it is faster)
        self.nmf = []
        self._dp_setup(args, rank, world, dist)

    def step(self, i):
        row = (i % 4) * self.DN
        if self.comm is not None:
            self.nmf.append(self.dp.train_step(self.Xd, 2e-3, 0.9, self.DK, row=row))
        else:
            self.nmf.append(self.eng.train_step(self.Xd, 2e-3, 0.9, self.DK, row=row)[0])

    def precondition_steps(self, seconds):
        return int(seconds * 700)

    def reset(self):
        self.nmf = []

    def report(self, args, world, dt, ev_ms):
        V_, H1, H2, N_, k = self.DV, self.H1, self.H2, self.DN, self.DK
        T = float(np.mean(self.nmf[-args.steps:])) if self.nmf else 0.0
        # SURVEY §8d (X.W0 hoisted out of the sweeps): MF + PCD + gradients (msre is not fetched here)
        flops = 2.0 * N_ * V_ * H1 + T * (4.0 * N_ * H1 * H2) + k * N_ * (4.0 * V_ * H1 + 4.0 * H1 * H2) \
            + 2.0 * (N_ + N_) * (V_ * H1 + H1 * H2)
        return {
            'metric': 'DBM updates/sec (784-512-1024, mean-field + PCD-5, 512 rows + 512 particles per GPU)',
            'value': round(world * args.steps / dt, 2),
            'unit': 'updates/s x GPUs (each update = 512 rows + 512 particles per GPU; weak scaling)',
            'config': {'workload': '2-layer DBM 784-512-1024 mean-field (<=50 sweeps, tol %g) + PCD-5 fp32 (BASELINE configs[3]%s)'
                                   % (getattr(args, 'dbm_mf_tol', 1e-7), '' if getattr(args, 'dbm_mf_tol', 1e-7) == 1e-7 else '; NON-BASELINE tolerance'),
                       'layers': [V_, H1, H2], 'batch_per_gpu': N_, 'particles_per_gpu': N_, 'n_gibbs_steps': k,
                       'mean_field_sweeps_executed': T, 'parallelism': 'dp%d' % world,
                       'collective': ('%s: all-reduce(max) of the mean-field residual per sweep + one all-reduce(sum) of '
                                      'the fused gradient buffer' % COLLECTIVE_NAMES.get(self.collective, self.collective))
                       if self.comm is not None else None, 'collective_note': self.collective_note,
                       'fast_binary': (FAST_NOTE + ' Here: the PCD particle sweeps; mean-field (real-valued mu) and the '
                                       'outer products stay fp32.') if self.fast else False},
            'flops_per_step': flops,
            'roofline_extra': {'bf16x3_flop_fraction': (k * N_ * (4.0 * V_ * H1 + 4.0 * H1 * H2) / flops) if self.fast else 0.0,
                               'scope': 'whole update, SURVEY 8d formula with T = %.1f executed mean-field sweeps = %.2f GFLOP'
                                        % (T, flops / 1e9), 'traffic': pmc_traffic('dbm')},
        }


class Ais(_DbmBase):
    """BASELINE configs[4]: AIS log Z, 20 000 chains x 1000 betas on the 784-512-1024 DBM (dbm.py:650-736)"""
    name = 'ais'
    scaling = 'strong'
    DV, H1, H2 = 784, 512, 1024

    def __init__(self, args, rank, world, local_rank, dist):
        from boltzmann_machines_amd import parallel
        from boltzmann_machines_amd.engine import DbmEngine
        from boltzmann_machines_amd.utils import philox
        self.R, self.nb, self.k = args.ais_runs, args.ais_betas, max(args.k, 1)
        self.rank, self.world = rank, world
        self.eng = eng = DbmEngine(self.DV, [self.H1, self.H2], n_particles=8, batch_size=8)
        eng.set('W', philox.tf_random_normal((self.DV, self.H1), 0.01, 1337))
        eng.set('W_1', philox.tf_random_normal((self.H1, self.H2), 0.01, 1111))
        self.fast = bool(getattr(args, 'fast_binary', False))
        eng.set_fast_binary(self.fast)
        # the all-gather of the per-chain values: the same choice as the gradient exchange (direct peer-memory exchange
        # after its start-up self-check against gloo, raced against RCCL; --collective overrides)
        self.comm, self.xchg, self.collective, self.collective_note, self.dist = None, None, None, None, dist
        if world > 1 or args.force_dp:
            self.collective, self.collective_note = choose_collective(args, eng, rank, world, dist)
            if self.collective == 'direct':
                self.xchg = args._xchg[id(eng)]
            elif self.collective in ('rccl', 'torch'):
                self.comm = get_comm(rank, world)
        self.start, self.stop = parallel.shard(self.R, rank, world)
        self.last = None

    def step(self, i):
        if self.xchg is not None:
            self.last = self.eng.ais_sharded_direct(self.xchg, self.nb, self.R, self.k, 2222)   # shard + ONE exchange launch
        elif self.comm is not None:
            self.last = self.eng.ais_sharded(self.comm, self.nb, self.R, self.k, 2222)    # shard + ONE all-gather
        elif self.collective == 'gloo':       # ranks share a device and the direct path is off: through the host
            import torch
            from boltzmann_machines_amd import parallel

            def allgather(local, counts):
                bufs = [torch.zeros(max(counts), dtype=torch.float32) for _ in counts]
                mine = torch.zeros(max(counts), dtype=torch.float32)
                mine[:len(local)] = torch.from_numpy(local)
                self.dist.all_gather(bufs, mine)
                return np.concatenate([b.numpy()[:c] for b, c in zip(bufs, counts)])
            self.last = parallel.ais_sharded(lambda n, c0: self.eng.ais(self.nb, n, self.k, 2222, chain0=c0),
                                             self.R, self.rank, self.world, allgather)
        else:
            self.last = self.eng.ais(self.nb, self.R, self.k, 2222)

    def report(self, args, world, dt, ev_ms):
        from boltzmann_machines_amd.utils import log_mean_exp
        flops = self.k * 4.0 * self.R * self.H1 * (self.DV + self.H2) * self.nb / world     # per GPU and run (deduplicated)
        return {
            'metric': 'AIS beta-steps/sec (20000 chains, 784-512-1024 DBM)',
            'value': round(self.nb * args.steps / dt, 2),
            'unit': 'beta-steps/s over all %d chains (one step of this bench = one full %d-beta run)' % (self.R, self.nb),
            'config': {'workload': 'AIS log Z, %d chains x %d betas, k=%d, DBM 784-512-1024 fp32 (BASELINE configs[4])'
                                   % (self.R, self.nb, self.k),
                       'chains_per_gpu': self.stop - self.start, 'parallelism': 'chains/%d' % world,
                       'fast_binary': FAST_NOTE if self.fast else False,
                       'collective': (COLLECTIVE_NAMES.get(self.collective, self.collective) + ': ONE gather of the per-chain '
                                      'values per run') if self.collective else None,
                       'collective_note': self.collective_note,
                       'log_Z_estimate': float(log_mean_exp(self.last.astype(np.float64))) if self.last is not None else None},
            'flops_per_step': flops,
            'roofline_extra': {'bf16x3_flop_fraction': 1.0 if self.fast else 0.0,
                               'scope': 'one run = n_betas * 4*M*H1*(V+H2) = %.2f TFLOP per GPU (SURVEY 8d, shared pre-activations)'
                                        % (flops / 1e12), 'traffic': pmc_traffic('ais')},
        }


WORKLOADS = {w.name: w for w in (RbmCD, RbmGibbs, Grbm, Dbm, Ais)}
DEFAULTS = {'rbm': (2000, 100), 'gibbs': (300, 30), 'grbm': (30, 5), 'dbm': (40, 5), 'ais': (2, 1)}
# the short passes the default run adds behind the headline: (steps, warm-up, untimed precondition seconds)
OTHERS = (('gibbs', 100, 10, 0.2), ('grbm', 12, 3, 0.2), ('dbm', 40, 8, 0.3), ('ais', 1, 1, 0.0),
          # the opt-in bf16 x 3 mode only where it gains (AIS 1.9x, the 3072 x 5000 particle sweeps +4 %): at the 784 x 1024 shapes
          # it is slower than fp32 and bm_*_set_fast_binary(1) no longer takes effect there (profiles/r5_{gibbs,dbm}_summary.md)
          ('ais+fast_binary', 1, 1, 0.0), ('grbm+fast_binary', 12, 3, 0.2))
FAST_NOTE = ('NON-DEFAULT opt-in mode: exact-product bf16 x 3 on the bf16 matrix cores (csrc/bm_bf3.h); results agree with '
             'the f32 chain to fp32 round-off, not bit for bit; the roofline block prices the flops that run as bf16 x 3 '
             'against the bf16 peak / 3 and the rest against the fp32-MFMA peak (`bf16x3_flop_fraction`)')
COLLECTIVE_NAMES = {
    'direct': 'bm_xchg (in-library one-shot reduce-scatter + all-gather over peer-mapped memory / xGMI)',
    'rccl': 'bm_comm (in-library RCCL all-reduce)',
    'torch': 'torch.distributed nccl (RCCL) on the engine stream',
    'gloo': 'gloo, staged through the host (no device collective usable: ranks share a device and the direct path is off)',
}


_COMM = [None]


def get_comm(rank, world):
    """the process-wide RCCL communicator of the library (created once: communicator set-up takes seconds)"""
    if _COMM[0] is None:
        from boltzmann_machines_amd import parallel
        _COMM[0] = parallel.NativeComm.from_torch_rendezvous(rank, world)
    return _COMM[0]


def gloo_staged_allreduce(eng, dist):
    """all-reduce of the engine's grad buffer through the host over gloo (a dry-run path for boxes where neither
    device collective can run; it synchronises the stream)"""
    import ctypes as C
    import torch
    from boltzmann_machines_amd import _ffi
    grad = eng.device_view('grad')

    def allreduce_():
        eng.sync()
        host = grad.numpy()
        dist.all_reduce(torch.from_numpy(host))
        _ffi.check(_ffi.load().bm_h2d(grad.ptr, host.ctypes.data_as(C.c_void_p), host.nbytes))
    return allreduce_


def choose_collective(args, eng, rank, world, dist):
    """Which exchange the data-parallel step uses: --collective direct|rccl|torch|gloo, default `direct` (the one-shot
    peer-memory exchange) after a SELF-CHECK at start-up: the direct all-reduce of a rank-dependent test pattern in
    the engine's own grad buffer must equal the gloo (host) all-reduce of the same pattern to fp32 round-off and
    report no timed-out wait on any rank; otherwise the run falls back to the library's RCCL all-reduce (or to the
    host-staged gloo reducer when the ranks share a device, which RCCL refuses) and says so in `collective_note`."""
    import torch
    from boltzmann_machines_amd import parallel
    want = args.collective
    if args.torch_comm:
        want = 'torch'
    if not hasattr(args, '_xchg'):
        args._xchg = {}
    fallback = 'gloo' if args._shared_devices else 'rccl'
    if want != 'direct':
        if want in ('rccl', 'torch') and args._shared_devices:
            return 'gloo', 'RCCL refuses two ranks on one device (dry run on a box with fewer GPUs than ranks)'
        return want, None
    note = None
    try:
        # (a dry run with several ranks on one device: the exchange must not spin on every CU, see bm_xchg_set_max_workgroups)
        x = parallel.DirectExchange(eng, rank, world, max_workgroups=48 if args._shared_devices else None)
        # a lost rank or an unusable peer mapping must show up within a second at start-up (a wait that expires is
        # fatal for the exchange object: sticky status, NaN results); the timed run gets the library's default back
        x.set_timeout(1.0)
        grad = eng.device_view('grad')
        n = grad.shape[0]
        ok_local = True
        for trial in range(2):          # twice: the second pass runs on warm flags / staging
            rs = np.random.RandomState(1000 * trial + rank)
            pat = rs.standard_normal(n).astype(np.float32)
            _h2d(grad, pat)
            eng.sync()
            x.allreduce_grads()
            eng.sync()
            got = grad.numpy()
            ref = torch.from_numpy(pat.copy())
            if world > 1:
                dist.all_reduce(ref)
            ok_local = ok_local and x.status() == 0 and np.allclose(got, ref.numpy(), rtol=1e-5, atol=1e-5)
        flag = torch.tensor([0 if ok_local else 1], dtype=torch.int32)
        if world > 1:
            dist.all_reduce(flag)
        _h2d(grad, np.zeros(n, dtype=np.float32))
        if int(flag.item()) == 0:
            args._xchg[id(eng)] = x
            x.set_timeout(float(os.environ.get('BM_XCHG_TIMEOUT_S', '20')))
            note = 'start-up self-check against the gloo all-reduce passed on every rank'
            if world > 1 and not args._shared_devices and not args.no_collective_race:
                # ... and the faster of the two device collectives is used, by measurement on this very buffer.  Every
                # rank takes the same branch: whether RCCL came up is agreed on first (a communicator that fails on ONE
                # rank must not leave the others in a collective)
                comm, err = None, ''
                try:
                    comm = get_comm(rank, world)
                except Exception as e:       # noqa: BLE001
                    err = str(e)[:120]
                have = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
                dist.all_reduce(have, op=dist.ReduceOp.MIN)
                if int(have.item()) == 0:
                    note += '; rccl could not be set up on every rank (%s): not raced' % (err or 'another rank failed',)
                else:
                    def timed(fn, iters=30):
                        for _ in range(3):
                            fn()
                        eng.sync(); dist.barrier()
                        t0 = time.perf_counter()
                        for _ in range(iters):
                            fn()
                        eng.sync()
                        t = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        return float(t.item())
                    t_direct = timed(x.allreduce_grads)
                    t_rccl = timed(lambda: comm.allreduce_grads(eng))
                    note += '; all-reduce of the %.1f MB buffer: direct %.1f us, rccl %.1f us' % (4e-6 * n, 1e6 * t_direct, 1e6 * t_rccl)
                    if hasattr(x, 'exchange_apply') and hasattr(eng, 'H') and eng.H % 4 == 0 and not args.no_fused_exchange:
                        # the fused launch (reduce-scatter -> update of the owned slice -> all-gather of W) on zero
                        # gradients with lr = 0: the parameters do not move
                        t_fused = timed(lambda: x.exchange_apply(B * world, 0.0, 0.0))
                        x.gather_dw()
                        note += ', fused exchange + update %.1f us (against rccl + apply_step as two launches)' % (1e6 * t_fused,)
                    _h2d(grad, np.zeros(n, dtype=np.float32))
                    st = torch.tensor([int(x.status())], dtype=torch.int32)
                    dist.all_reduce(st, op=dist.ReduceOp.MAX)
                    note += '; exchange status after the race: %d on every rank' % int(st.item()) if int(st.item()) == 0 else \
                        '; a wait expired during the race (status %d)' % int(st.item())
                    if int(st.item()) != 0:
                        args._xchg.pop(id(eng), None)      # (nothing may ask a closed exchange for its status later)
                        x.close()
                        return 'rccl', note + ' -> rccl'
                    if t_rccl < t_direct:
                        return 'rccl', note + ' -> rccl'
            return 'direct', note
        note = 'direct exchange failed its start-up self-check on %d rank(s): fell back' % int(flag.item())
        x.close()
    except Exception as e:      # every rank takes the same branch only if the failure is symmetric: confirm it
        note = 'direct exchange could not be set up (%s): fell back' % (str(e)[:200],)
        flag = torch.tensor([1], dtype=torch.int32)
        if world > 1:
            dist.all_reduce(flag)
    return fallback, note


def check_data_parallel_run(wl, args, rank, world, dist, fatal=True):
    """After the timed region of a data-parallel run: (1) no in-kernel wait of the direct exchange ever expired - a
    wait that expires is fatal (sticky status word, NaN results): the run is then INVALID and the bench exits non-zero
    on every rank instead of printing a throughput built on partial sums; (2) the replicas hold identical parameters
    (CRC of W / vb / hb of every rank through gloo).  Returns a short record for the JSON line, None at world 1."""
    if dist is None or world <= 1 or not getattr(wl, 'use_dp', True) or not hasattr(wl, 'eng'):
        return None
    import zlib
    import torch
    status = 0
    for x in getattr(args, '_xchg', {}).values():
        # the fused exchange leaves the momentum buffer sharded over the ranks: complete the replicas (a collective: every
        # rank is here) before anything - the instrumented kernel pass, a checkpoint - reads or updates it
        if getattr(wl, 'dp', None) is not None and getattr(wl.dp, 'fused', None) is x:
            try:
                trace('exchange status before gather_dw: %d' % int(x.status()))
                x.gather_dw()
                wl.eng.sync()
                trace('gather_dw done')
            except Exception as e:       # noqa: BLE001 - shows up as a non-zero status below
                sys.stderr.write('bench: gather_dw failed: %s\n' % (str(e)[:200],))
        status = max(status, int(x.status()))
    crc, nonfinite = 0, 0
    try:
        names = ('W', 'vb', 'hb') if hasattr(wl.eng, 'H') else ('W', 'vb', 'hb', 'W_1', 'hb_1')
        for n in names:
            a = np.ascontiguousarray(wl.eng.get(n))
            crc = zlib.crc32(a.tobytes(), crc)
            nonfinite += int(a.size - np.count_nonzero(np.isfinite(a)))     # identical NaNs would pass the CRC
    except Exception:       # noqa: BLE001 - a workload without these variables
        crc = -1
    t = torch.tensor([status, crc & 0x7FFFFFFF, -(crc & 0x7FFFFFFF), nonfinite], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    worst, hi, neg_lo, bad = int(t[0]), int(t[1]), int(t[2]), int(t[3])
    if worst != 0 or bad != 0:
        why = ('a wait of the direct exchange expired (status %d)' % worst) if worst != 0 else \
            ('%d non-finite parameter values after the data-parallel run' % bad)
        sys.stderr.write('bench: %s: the run is invalid\n' % why)
        if fatal:
            sys.exit(3)
        return {'invalid': why}         # (every rank sees the same reduced values and takes the same branch)
    return {'exchange_status': 0, 'replicas_identical': bool(hi == -neg_lo), 'parameters_finite': True}


def start_guardian(line_out, record):
    """fork a child that prints `record` as the JSON line if this process dies before emit(); returns the write end of
    the pipe (None if fork is unavailable).  The child touches nothing but the pipe and the duplicated stdout."""
    try:
        r, w = os.pipe()
        line = json.dumps(record) + '\n'
        line_out.flush()
        pid = os.fork()
    except OSError:
        return None
    if pid == 0:
        try:
            os.close(w)
            got = os.read(r, 1)            # b'd': the parent is about to print; b'': the parent died
            if got != b'd':
                os.write(line_out.fileno(), line.encode())
        finally:
            os._exit(0)
    os.close(r)
    return w


def _h2d(darr, host):
    import ctypes as C
    from boltzmann_machines_amd import _ffi
    host = np.ascontiguousarray(host)
    _ffi.check(_ffi.load().bm_h2d(darr.ptr, host.ctypes.data_as(C.c_void_p), host.nbytes))


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks ourselves (one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, a free port on 127.0.0.1); rank 0's stdout is
    ours, so the ONE JSON line comes out unchanged."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p_ in procs:
        rc = p_.wait() or rc
    sys.exit(rc)


_T0 = time.perf_counter()


def trace(msg):
    """BM_BENCH_TRACE=1: time-stamped per-rank progress lines on stderr (where does a multi-rank run spend its time /
    which rank is late when an exchange wait expires)"""
    if os.environ.get('BM_BENCH_TRACE', '0') == '1':
        sys.stderr.write('[bench rank %s +%.2fs] %s\n' % (os.environ.get('RANK', '0'), time.perf_counter() - _T0, msg))
        sys.stderr.flush()


def measure(wl, steps, warmup, precondition_s, barrier, dist):
    """W untimed warm-up steps, then EXACTLY `steps` steps between two barriers (+ device synchronisation); returns
    (wall seconds, max over ranks; HIP-event milliseconds on the engine stream of this rank)."""
    import torch
    eng = wl.eng
    # Precondition the device: a GPU box that has been idle needs a few hundred ms of work before its clocks and
    # caches are in the state a training job runs in (a 25-step run from cold measured 74 us/update against 66.7
    # steady).  A FIXED number of steps (every rank of a data-parallel run must issue the same number of
    # collectives); untimed; the model is then put back to its initial state.
    if precondition_s > 0:
        trace('precondition: %d steps' % wl.precondition_steps(precondition_s))
        wl.run_steps(0, wl.precondition_steps(precondition_s))
        wl.reset()              # (in stream order where the workload can: no idle gap before the warm-up steps)
    trace('warm-up: %d steps' % warmup)
    wl.run_steps(0, warmup)
    barrier()
    trace('timed region: %d steps' % steps)
    t0 = time.perf_counter()
    eng.timer_start()
    wl.run_steps(0, steps)
    eng.timer_mark()                     # HIP event behind the last step (read after the clock has been stopped)
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = eng.timer_elapsed()
    trace('timed region done: %.3f s' % dt)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, ev_ms


def make_record(wl, rep, world, steps, warmup, precondition_s, dt, ev_ms):
    flops = rep.pop('flops_per_step')
    extra = rep.pop('roofline_extra')
    achieved = flops / (ev_ms / steps * 1e-3) / 1e12
    achieved_wall = flops / (dt / steps) / 1e12
    traffic = extra.pop('traffic', None)
    src = None
    if isinstance(traffic, tuple):
        traffic, src = traffic
    # the roof that binds: fp32 MFMA for the default path; for the opt-in fast-binary mode the fraction f of the flops
    # that runs as three bf16 passes is priced at the bf16 peak / 3, the rest at the fp32 peak (harmonic blend) -
    # never a frac above 1
    f16 = float(extra.get('bf16x3_flop_fraction', 0.0) or 0.0)
    peak = 1.0 / (f16 / PEAK_BF16X3 + (1.0 - f16) / PEAK_FP32_MFMA)
    bound = 'mfma' if f16 == 0.0 else ('mfma-bf16x3' if f16 == 1.0 else 'mfma (bf16x3 / fp32 blend)')
    roof = dict({'bound': bound, 'achieved': round(achieved, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                 'frac': round(achieved / peak, 4),
                 # the same flops over the WALL clock of the timed region (ms_per_step): what the driver's own clock sees
                 'frac_wall': round(achieved_wall / peak, 4),
                 'timebase': 'frac: HIP events on the engine stream around the timed steps; frac_wall: ms_per_step',
                 'traffic': traffic,
                 # HBM-side bytes come from a committed rocprofv3 PMC pass of this same command, not from this run
                 'traffic_source': src}, **extra)
    return {'metric': rep['metric'], 'value': rep['value'], 'unit': rep['unit'],
            'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(1e3 * dt / steps, 5), 'higher_is_better': True, 'scaling': wl.scaling,
            'precondition_s': precondition_s,
            'vs_baseline': None, 'dtype': 'bf16x3 products, f32 accumulate' if rep['config'].get('fast_binary') else 'f32',
            'data': 'synthetic', 'config': rep['config'], 'roofline': roof}


def with_summary(out):
    """the line with a compact `summary` of EVERY configuration right behind metric / value / unit - {config: [ms_per_step,
    roofline.frac (HIP events), roofline.frac_wall]} - so that a reader who keeps only the head or the tail of the line (the
    driver's record truncates the middle of `other_configs`) still sees the default-mode gibbs / grbm / dbm / ais numbers"""
    if not isinstance(out, dict) or 'roofline' not in out:
        return out
    summ = {out['config'].get('name', 'headline') if isinstance(out.get('config'), dict) else 'headline':
            [out['ms_per_step'], out['roofline'].get('frac'), out['roofline'].get('frac_wall')]}
    for name, rec in (out.get('other_configs') or {}).items():
        if isinstance(rec, dict) and 'roofline' in rec:
            summ[name] = [rec['ms_per_step'], rec['roofline'].get('frac'), rec['roofline'].get('frac_wall')]
        elif isinstance(rec, dict) and 'error' in rec:
            summ[name] = 'error'
    front = ('metric', 'value', 'unit')
    res = {k: out[k] for k in front if k in out}
    res['summary'] = summ
    res['summary_fields'] = '{config: [ms_per_step, roofline.frac, roofline.frac_wall]}'
    res.update({k: v for k, v in out.items() if k not in front})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', choices=sorted(WORKLOADS), default='rbm')
    ap.add_argument('--k', type=int, default=1, help='n_gibbs_steps of CD-k (rbm), sweeps per call (gibbs), AIS transitions per beta')
    ap.add_argument('--ais-runs', type=int, default=20000)
    ap.add_argument('--ais-betas', type=int, default=1000)
    ap.add_argument('--dbm-mf-tol', type=float, default=1e-7,
                    help='dbm: mean-field tolerance (BASELINE configs[3]: 1e-7 = the loop runs to its 50-sweep cap; a looser one '
                         'measures the update with the SHORT loop of a trained model - a developer setting, named in config.workload)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--fast-binary', action='store_true',
                    help='gibbs / ais / grbm / dbm: the opt-in exact-product bf16 x 3 mode (bm_*_set_fast_binary); a NON-default, '
                         'tolerance-parity mode, reported separately from the f32 figures')
    ap.add_argument('--no-others', action='store_true',
                    help='rbm: do not add the short passes of the other BASELINE configurations (`other_configs`)')
    ap.add_argument('--others-budget-s', type=float, default=150.0,
                    help='wall-clock budget of the `other_configs` passes; a pass still running when it expires is '
                         'abandoned (the headline line is printed with what finished)')
    ap.add_argument('--precondition-s', type=float, default=0.4,
                    help='seconds (approximate: a fixed step count) of untimed steps BEFORE the W warm-up steps (launch '
                         'tuning, instruction caches, clock ramp of an idle GPU); parameters and RNG are reset afterwards, '
                         'so the warm-up and the timed steps start from the documented initial state.  0 disables')
    ap.add_argument('--collective', choices=('direct', 'rccl', 'torch', 'gloo'), default='direct',
                    help='exchange step of the data-parallel configurations at N > 1: the library\'s one-shot peer-memory '
                         'exchange (bm_xchg_*, default, behind a start-up self-check with a fallback to rccl), the '
                         'library\'s RCCL all-reduce (bm_comm_*), torch.distributed nccl, or gloo staged through the host')
    ap.add_argument('--no-fused-exchange', action='store_true',
                    help='--collective direct, rbm: all-reduce and bm_rbm_apply_step as two launches instead of the fused '
                         'reduce-scatter -> update -> all-gather kernel (bm_rbm_exchange_apply_direct; same bits)')
    ap.add_argument('--no-collective-race', action='store_true',
                    help='--collective direct: do not time the direct exchange against RCCL at start-up (use direct)')
    ap.add_argument('--native-comm', action='store_true', help='(kept for old command lines) same as --collective rccl')
    ap.add_argument('--torch-comm', action='store_true', help='same as --collective torch')
    ap.add_argument('--delayed-grads', action='store_true',
                    help='rbm, NON-parity: delayed-gradient data parallelism (the update of step t is the reduced '
                         'gradient of step t-1; the all-reduce runs under the next step) over bm_comm')
    ap.add_argument('--force-dp', action='store_true',
                    help='take the data-parallel code path (grad_step -> all-reduce -> apply_step) even at N=1')
    args = ap.parse_args()
    if args.native_comm:
        args.collective = 'rccl'
    explicit_steps = args.steps is not None
    if args.steps is None:
        args.steps = DEFAULTS[args.config][0]
    if args.warmup is None:
        args.warmup = DEFAULTS[args.config][1]

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)                # does not return
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # The contract is ONE JSON line on stdout.  Libraries write there too (gloo prints "[Gloo] Rank 0 is connected to
    # 1 peer ranks" from C++ when a process group forms, RCCL its banner under NCCL_DEBUG): from here on file
    # descriptor 1 IS stderr, and the line goes to a private copy of the original stdout.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    import torch
    from boltzmann_machines_amd import _ffi
    lib = _ffi.load()
    ndev = lib.bm_device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    # fewer devices than ranks (a dry run of the N > 1 path on a 1-GPU box): ranks share devices.  RCCL refuses that;
    # the direct exchange and the gloo-staged reducer do not care
    args._shared_devices = ndev < world
    device = local_rank % ndev
    _ffi.check(lib.bm_set_device(device))      # first: the library's host-wait policy (spin) applies to a fresh context
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # rendezvous + timing barrier only.  A bounded time-out: a rank that died leaves the others in a gloo collective;
        # they must raise and exit non-zero, not wait for gloo's default half hour
        import datetime
        dist.init_process_group('gloo', rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get('BM_BENCH_GLOO_TIMEOUT_S', '240'))))

    def barrier():
        torch.cuda.synchronize()         # device-wide: covers the engine's streams
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    trace('process group up; building the workload')
    wl = WORKLOADS[args.config](args, rank, world, device, dist)
    trace('workload ready (collective: %s)' % (getattr(wl, 'collective', None),))
    dt, ev_ms = measure(wl, args.steps, args.warmup, args.precondition_s, barrier, dist)
    dp_check = check_data_parallel_run(wl, args, rank, world, dist)      # (after the clock stopped)
    rep = wl.report(args, world, dt, ev_ms) if rank == 0 else None
    barrier()
    out = make_record(wl, rep, world, args.steps, args.warmup, args.precondition_s, dt, ev_ms) if rank == 0 else None
    if rank == 0 and args.config == 'rbm':
        out['unit'] += '; steady state: %.1f s of untimed updates precede the warm-up steps' % args.precondition_s \
            if args.precondition_s > 0 else ''
    if rank == 0 and dp_check is not None:
        out['config']['data_parallel_check'] = dp_check
    if rank == 0 and args.config == 'rbm' and args._shared_devices:
        out['config']['devices_shared'] = 'DRY RUN: %d ranks on %d device(s); not a scaling figure' % (world, ndev)

    guardian = [None]

    def emit():
        if rank != 0:
            return
        if guardian[0] is not None:
            try:
                os.write(guardian[0], b'd')    # the parent prints its own line: the guardian leaves silently
                os.close(guardian[0])
            except OSError:
                pass
            guardian[0] = None
        # the ONE line of the contract is the last thing on stdout: flush what C libraries (the RCCL banner under
        # NCCL_DEBUG=VERSION) still hold in their stdio buffers first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        line_out.write(json.dumps(with_summary(out)) + '\n')
        line_out.flush()

    # ---- the other BASELINE configurations, short passes, embedded in the same line (rbm default run only).
    # Every rank runs the same sequence.  A watchdog bounds the whole block: if a pass hangs (a collective that
    # never completes on some rank), rank 0 prints the headline line with what finished and all ranks leave.
    if args.config == 'rbm' and not args.no_others and not args.delayed_grads:
        import threading
        others = {}
        if rank == 0:
            # The headline is measured; what follows (eight more workloads, collectives included at N > 1) must not be
            # able to lose it.  The watchdog below covers a hang; a GUARDIAN child covers a hard crash (a fault inside a
            # kernel or a collective kills the process without running any Python): it holds the finished headline
            # line and prints it if the parent's end of the pipe closes without the 'done' byte.
            guardian[0] = start_guardian(line_out, dict(out, other_configs={'_aborted': 'the process died during the other_configs passes'}))
            out['other_configs'] = others
        done = threading.Event()

        def watchdog():
            if not done.wait(args.others_budget_s):
                if rank == 0:
                    others['_aborted'] = 'other_configs exceeded --others-budget-s=%.0f s; abandoned' % args.others_budget_s
                    emit()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        wl.eng.close()
        del wl
        for name, st, wu, pre in OTHERS:
            t_begin = time.time()
            try:
                a2 = argparse.Namespace(**vars(args))
                a2.k = 1
                a2._xchg = {}
                a2.fast_binary = name.endswith('+fast_binary')
                name_wl = name.split('+')[0]
                w2 = WORKLOADS[name_wl](a2, rank, world, device, dist)
                dt2, ev2 = measure(w2, st, wu, pre, barrier, dist)
                # the same post-run check as the headline's (a collective: every rank is here): a pass whose exchange lost a
                # rank or whose parameters are not finite is recorded as an error, never as a throughput
                chk = check_data_parallel_run(w2, a2, rank, world, dist, fatal=False)
                if chk is not None and 'invalid' in chk:
                    raise RuntimeError(chk['invalid'])
                if rank == 0:
                    rec = make_record(w2, w2.report(argparse.Namespace(**dict(vars(a2), steps=st)), world, dt2, ev2),
                                      world, st, wu, pre, dt2, ev2)
                    for k_ in ('higher_is_better', 'vs_baseline', 'data', 'n_gpus'):
                        rec.pop(k_, None)
                    rec['pass_seconds'] = round(time.time() - t_begin, 1)
                    if chk is not None:
                        rec['config']['data_parallel_check'] = chk
                    others[name] = rec
                barrier()
                w2.eng.close()
                del w2
            except Exception as e:       # noqa: BLE001 - recorded in the line; the headline stands
                if rank == 0:
                    others[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        done.set()

    if rank == 0 and not args.no_cpu and world == 1 and args.config == 'rbm':
        out['cpu_baseline'] = cpu_baseline(args.k)
    emit()
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == '__main__':
    main()
