"""Comparison of a scenario run (tests/golden/scenarios.py) with the fixture the reference produced for it
(tests/golden/ref_<scenario>.npz, written by tests/golden/make_golden_from_reference.py).  Test infrastructure."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class OursPackage(object):
    """the names a scenario uses, bound to boltzmann_machines_amd"""

    def __init__(self):
        import boltzmann_machines_amd as bm
        from boltzmann_machines_amd.utils import RNG
        self.BernoulliRBM, self.GaussianRBM, self.MultinomialRBM, self.DBM, self.RNG = \
            bm.BernoulliRBM, bm.GaussianRBM, bm.MultinomialRBM, bm.DBM, RNG


def load(name):
    with np.load(os.path.join(GOLDEN, 'ref_%s.npz' % name)) as z:
        return {k: z[k] for k in z.files}


def compare(name, got, ref, rtol=1e-5, metrics_rtol=1e-5, atol=1e-7, n_mf_atol=0.0):
    """every array the reference returned, by name: same keys (the get_tf_params surface included), same shapes,
    integers / counters exact, reals within `rtol` of the array's scale (north_star: 1e-5 fp32 relative) plus `atol`
    = one float32 ulp of the unit-scale intermediates (probabilities, states) - momentum buffers of the biases are
    means of DIFFERENCES of such quantities and inherit their absolute, not their relative, round-off.
    Returns the report lines; raises AssertionError listing every mismatch."""
    ref = {k: v for k, v in ref.items() if k != 'min_bernoulli_margin'}
    errors, report = [], []
    missing, extra = sorted(set(ref) - set(got)), sorted(set(got) - set(ref))
    if missing:
        errors.append('missing outputs: %s' % missing)
    if extra:
        errors.append('outputs the reference does not return: %s' % extra)
    for k in sorted(set(ref) & set(got)):
        r, g = np.asarray(ref[k]), np.asarray(got[k])
        if r.shape != g.shape:
            errors.append('%s: shape %s != %s' % (k, g.shape, r.shape))
            continue
        if k.endswith('epoch_iter') or k == 'n_samples_generated' or r.dtype.kind in 'iub':
            if not np.array_equal(r, g):
                errors.append('%s: %s != %s' % (k, g.tolist(), r.tolist()))
            continue
        r64, g64 = r.astype(np.float64), g.astype(np.float64)
        if np.isnan(r64).any() or np.isnan(g64).any():
            if not np.array_equal(np.isnan(r64), np.isnan(g64)):
                errors.append('%s: NaN pattern differs' % k)
                continue
            r64, g64 = np.nan_to_num(r64), np.nan_to_num(g64)
        scale = max(float(np.max(np.abs(r64))) if r64.size else 0.0, 1e-30)
        diff = float(np.max(np.abs(r64 - g64))) if r64.size else 0.0
        tol = metrics_rtol if k == 'metrics' else rtol
        if k == 'metrics_n_mf_updates':              # epoch means of integer trip counts
            if not diff <= n_mf_atol + 1e-9:
                errors.append('%s: executed mean-field sweeps differ by %.2f (allowed %.2f)' % (k, diff, n_mf_atol))
            continue
        report.append('%-60s %.2e (abs %.2e)' % (k, diff / scale, diff))
        if not diff <= tol * scale + atol:
            errors.append('%s: max |diff| = %.3e (%.3e of max |ref|) > %.1e * scale + %.1e' % (k, diff, diff / scale, tol, atol))
    if errors:
        raise AssertionError('%s does not reproduce the reference fixture:\n  ' % name + '\n  '.join(errors))
    return report
