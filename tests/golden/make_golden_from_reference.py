"""Golden fixtures produced BY THE REFERENCE: imports the unmodified /root/reference/boltzmann_machines package on
top of the NumPy TF-1 stand-in (tests/tf1_shim), runs every scenario of tests/golden/scenarios.py through its public
API and writes tests/golden/ref_<scenario>.npz.

    python tests/golden/make_golden_from_reference.py [scenario ...]

Runs in the build container only (the reference checkout does not travel to the GPU box; the .npz files do).
Random streams of the reference's unseeded ops follow tests/golden/reference_rng_policy.py; a scenario whose Bernoulli
draws come closer than 2e-6 to a tie (|u - p|) is rejected, so that float32 round-off of a different summation order
cannot flip a sample of the recorded trajectory."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MIN_MARGIN = 2e-6


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
        self.margins = []
        from tensorflow.contrib import distributions
        distributions.set_margin_trace(lambda scope, m: self.margins.append((m, scope)))


def generate(name, pkg=None):
    from tests.golden import scenarios
    pkg = pkg or ReferencePackage()
    pkg.margins[:] = []
    d = tempfile.mkdtemp(prefix='bm_ref_%s_' % name)
    cwd = os.getcwd()
    try:
        os.chdir(d)
        out = scenarios.SCENARIOS[name](pkg, d)
    finally:
        os.chdir(cwd)
        shutil.rmtree(d, ignore_errors=True)
    margin = min(pkg.margins)[0] if pkg.margins else np.inf
    out['min_bernoulli_margin'] = np.array([margin])
    return out


def main(argv):
    from tests.golden import scenarios
    names = argv or sorted(scenarios.SCENARIOS)
    pkg = ReferencePackage()
    bad = []
    for name in names:
        out = generate(name, pkg)
        margin = float(out['min_bernoulli_margin'][0])
        path = os.path.join(HERE, 'ref_%s.npz' % name)
        if margin < MIN_MARGIN:
            print('%-40s REJECTED: a Bernoulli draw lies %.2e from a tie: pick other seeds for this scenario' % (name, margin))
            bad.append(name)
            continue
        np.savez_compressed(path, **out)
        print('%-40s %3d arrays  %7.1f KiB  min |u - p| = %.2e' % (name, len(out), os.path.getsize(path) / 1024.0, margin))
    if bad:
        raise SystemExit('rejected: %s' % ', '.join(bad))


if __name__ == '__main__':
    main(sys.argv[1:])
