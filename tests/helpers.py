"""Shared builders for the parity tests: a GPU engine and its CPU-oracle twin
with identical parameters, seeds and inputs."""
import numpy as np

from oracle import oracle as orc


def synth_data(B, V, seed, gaussian=False):
    if gaussian:
        return orc.normal(87654321, 43 + seed, 0, B * V).reshape(B, V)
    u = orc.uniform(87654321, 42 + seed, 0, B * V).reshape(B, V)
    return (u < 0.1307).astype(np.float32)


def make_pair(V, H, max_batch, w_seed=1337, w_std=0.01, **kw):
    from boltzmann_machines_amd.engine import RbmEngine
    eng = RbmEngine(V, H, max_batch=max_batch, **kw)
    twin = orc.OracleRBM(V, H, **kw)
    W = (orc.normal(87654321, w_seed, 0, V * H) * np.float32(w_std)).reshape(V, H).astype(np.float32)
    vb = (orc.uniform(87654321, w_seed + 1, 0, V) - np.float32(0.5)).astype(np.float32) * np.float32(0.2)
    hb = (orc.uniform(87654321, w_seed + 2, 0, H) - np.float32(0.5)).astype(np.float32) * np.float32(0.2)
    for name, val in (('W', W), ('vb', vb), ('hb', hb)):
        eng.set(name, val)
        twin.p[name][...] = val
    return eng, twin


def assert_state_equal(eng, twin, names=('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means')):
    for n in names:
        g = eng.get(n)
        c = twin.p[n]
        bad = int(np.sum(g.view(np.uint32) != c.view(np.uint32)))
        assert bad == 0, '%s: %d / %d elements differ bitwise (max abs diff %.3e)' % (
            n, bad, g.size, float(np.max(np.abs(g - c))))
