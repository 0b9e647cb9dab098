"""The drop-in boundary is a C-ABI: examples/rbm_c_abi.c is a plain C99 program (no Python, no torch, no C++) that
includes include/bm355.h, links libbm355.so and trains an RBM.  CPU: it compiles, links, and refuses to run
without a GPU (no CPU fallback).  GPU: its result is the ctypes binding's result, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, 'boltzmann_machines_amd')


def _build(tmp_path):
    from boltzmann_machines_amd import _ffi
    _ffi.load()                                            # builds libbm355.so if needed
    exe = str(tmp_path / 'rbm_c_abi')
    cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'rbm_c_abi.c'), '-L', LIBDIR, '-lbm355',
           '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath-link,/opt/rocm/lib', '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_program_builds_and_has_no_cpu_fallback(tmp_path):
    exe = _build(tmp_path)
    from boltzmann_machines_amd import _ffi
    if _ffi.load().bm_device_count() > 0:
        pytest.skip('a GPU is present: covered by the gpu test')
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and 'no CPU fallback' in r.stderr


@pytest.mark.gpu
def test_c_program_matches_ctypes_binding(gpu_lib, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = r.stdout.split()[-1]
    # the same run through the Python binding
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    V, H, B, steps = 96, 72, 48, 3
    s = np.uint32(12345)
    vals = np.empty(V * H + B * V, dtype=np.uint32)
    with np.errstate(over='ignore'):
        for i in range(len(vals)):
            s = np.uint32(s * np.uint32(1664525) + np.uint32(1013904223))
            vals[i] = s >> np.uint32(8)
    W = ((vals[:V * H].astype(np.float32) / np.float32(16777216.0) - np.float32(0.5)) * np.float32(0.2)).reshape(V, H)
    X = np.where(vals[V * H:] % 10 < 3, 1.0, 0.0).astype(np.float32).reshape(B, V)
    eng = RbmEngine(V, H, max_batch=B, l2=1e-4, sample_v_states=True, sample_h_states=True,
                    sparsity_damping=0.9, sparsity_target=0.1)
    eng.set('W', W)
    eng.seed(2024)
    Xd = as_device(X)
    for _ in range(steps):
        eng.train_step(Xd, B, 0.05, 0.5, 1)
    h = 1469598103934665603
    for b in eng.get('W').view(np.uint32).ravel().tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert got == '%016x' % h
    eng.close()
