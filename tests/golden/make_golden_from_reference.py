"""Golden fixtures produced BY THE REFERENCE: imports the unmodified /root/reference/boltzmann_machines package on
top of the NumPy TF-1 stand-in (tests/tf1_shim), runs every scenario of tests/golden/scenarios.py through its public
API and writes tests/golden/ref_<scenario>.npz.

    python tests/golden/make_golden_from_reference.py [scenario ...]

Runs in the build container only (the reference checkout does not travel to the GPU box; the .npz files do).
Random streams of the reference's unseeded ops follow tests/golden/reference_rng_policy.py.

Near-ties (hard part 2).  A Bernoulli draw with |u - p| below the float32 round-off of p can come out the
other way under another summation order of the same matrix product; from there the two trajectories differ.  Such
draws are not avoidable at the BASELINE sizes (10^6 .. 10^8 draws per scenario), so they are RECORDED, not rejected:
every draw with |u - p| < TIE_EPS goes into the fixture as a row of `near_ties` = (label, op execution, row, column,
u - p), `label` being the public call the scenario announced with `pkg.mark()`; `near_tie_labels` / `near_tie_scopes`
name the labels and the graph scope of each draw, `n_bernoulli_draws` counts all draws.  tests/reference_fixtures.py
requires equality everywhere except on trajectories a recorded near-tie can have forked, and reports the counts."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TIE_EPS = 2e-6


class ReferencePackage(object):
    """the five names a scenario uses, bound to the reference's classes"""

    def __init__(self):
        from tests import reference_shim
        self.tf, bm = reference_shim.activate()
        from tests.golden import reference_rng_policy
        self.tf.set_rng_policy(reference_rng_policy.policy)
        from boltzmann_machines.rbm import BernoulliRBM, GaussianRBM, MultinomialRBM
        from boltzmann_machines.dbm import DBM
        from boltzmann_machines.utils import RNG
        self.BernoulliRBM, self.GaussianRBM, self.MultinomialRBM, self.DBM, self.RNG = \
            BernoulliRBM, GaussianRBM, MultinomialRBM, DBM, RNG
        self.reset()
        from tensorflow.contrib import distributions
        distributions.set_margin_trace(self._trace, TIE_EPS)

    def reset(self):
        self.margin, self.n_draw_ops, self.labels, self.label = np.inf, 0, ['start'], 0
        self.ties, self.tie_scopes, self.n_draws = [], [], 0

    def mark(self, label):
        """scenario hook: the draws that follow belong to the public call `label`"""
        if label not in self.labels:
            self.labels.append(label)
        self.label = self.labels.index(label)

    def _trace(self, scope, margin, ties, n):
        self.margin = min(self.margin, margin)
        self.n_draws += n
        for row, col, d in ties:
            self.ties.append((self.label, self.n_draw_ops, row, col, d))
            self.tie_scopes.append(scope)
        self.n_draw_ops += 1


def generate(name, pkg=None):
    from tests.golden import scenarios
    pkg = pkg or ReferencePackage()
    pkg.reset()
    d = tempfile.mkdtemp(prefix='bm_ref_%s_' % name)
    cwd = os.getcwd()
    try:
        os.chdir(d)
        out = scenarios.SCENARIOS[name](pkg, d)
    finally:
        os.chdir(cwd)
        shutil.rmtree(d, ignore_errors=True)
    out['min_bernoulli_margin'] = np.array([pkg.margin])
    out['near_ties'] = np.asarray(pkg.ties, dtype=np.float64).reshape(-1, 5)
    out['near_tie_labels'] = np.array(pkg.labels)
    out['near_tie_scopes'] = np.array(pkg.tie_scopes, dtype=str) if pkg.tie_scopes else np.zeros(0, dtype='<U1')
    out['n_bernoulli_draws'] = np.array([pkg.n_draws], dtype=np.int64)
    return out


def main(argv):
    from tests.golden import scenarios
    names = argv or sorted(scenarios.SCENARIOS)
    pkg = ReferencePackage()
    for name in names:
        out = generate(name, pkg)
        path = os.path.join(HERE, 'ref_%s.npz' % name)
        np.savez_compressed(path, **out)
        print('%-40s %3d arrays  %7.1f KiB  %.3g Bernoulli draws, %d within %.0e of a tie, min |u - p| = %.2e'
              % (name, len(out), os.path.getsize(path) / 1024.0, float(out['n_bernoulli_draws'][0]),
                 len(out['near_ties']), TIE_EPS, float(out['min_bernoulli_margin'][0])))


if __name__ == '__main__':
    main(sys.argv[1:])
