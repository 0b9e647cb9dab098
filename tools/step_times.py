#!/usr/bin/env python
"""Per-step HIP-event times of the CD-1 update after an idle gap (why a 20-step timed region reads slower per step
than a 2000-step one): precondition, synchronise (the bench barrier), then N steps with an event after each."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from boltzmann_machines_amd import _ffi  # noqa: E402

lib = _ffi.load()
_ffi.check(lib.bm_set_device(0))
torch.cuda.set_device(0)


class A(object):
    k, force_dp, delayed_grads = 1, False, False


wl = bench.RbmCD(A(), 0, 1, 0, None)
eng = wl.eng
st = torch.cuda.ExternalStream(eng.stream(), device=torch.device('cuda', 0))
for gap_ms in (0.0, 1.0, 20.0):
    for i in range(3000):
        wl.step(i)
    torch.cuda.synchronize()
    time.sleep(gap_ms * 1e-3)
    n = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    with torch.cuda.stream(st):
        ev[0].record()
        for i in range(n):
            wl.step(i)
            ev[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
    print('idle gap %.0f ms: us per step (events), steps 0-7: %s | mean 8-19 %.1f | mean 20-39 %.1f'
          % (gap_ms, ' '.join('%.1f' % x for x in t[:8]), t[8:20].mean(), t[20:].mean()))
