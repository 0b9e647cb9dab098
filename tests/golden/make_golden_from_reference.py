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
requires equality everywhere except on trajectories a recorded near-tie can have forked, and reports the counts.

Mean-field trip counts (dbm.py:449-452).  The loop runs while `max |mu - mu_new| > mf_tol`; at the reference's default
mf_tol = 1e-7 the residual that ends it sits at the float32 noise floor (one ulp of a mean in [0.5, 1) is 5.96e-8), so the
deciding comparison - like a Bernoulli near-tie - can go the other way under another summation order of the same matrix
products.  Every execution of the loop whose count enters the progress line (`n_mf_upds`, `val.n_mf_upds`) is RECORDED as a
row of `mf_loops` = (progress line, column: 0 train / 1 validation, executed sweeps, may-end-one-sweep-earlier,
may-run-one-sweep-longer, final residual, last residual above the tolerance): "earlier" is set when the last residual above
the tolerance lies within MF_NOISE = 2^-23 of it, "longer" when the final residual is not zero (some mean still moved by an
ulp in the reference's own last sweep) and lies within MF_NOISE below it.  tests/reference_fixtures.py compares `metrics_n_mf_updates` EXACTLY except for the one
sweep such a recorded loop allows."""
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
MF_NOISE = 2.0 ** -23        # two ulps of a mean in [0.5, 1): what a one-ulp difference of a pre-activation moves the residual by


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
        self._hook_mean_field(DBM)

    def _hook_mean_field(self, DBM):
        """record every execution of the mean-field while_loop whose trip count the reference reports (module docstring)"""
        pkg, tf = self, self.tf
        tf.set_compare_trace(lambda scope, a, b: pkg._mf_cur.append((a, b)) if 'mean_field' in scope else None)
        run0 = tf.Session.run

        def run(sess, fetches, feed_dict=None):
            pkg._mf_cur = []
            out = run0(sess, fetches, feed_dict)
            m, tag = pkg._mf_model, pkg._mf_tag
            if pkg._mf_cur and tag is not None and isinstance(fetches, (list, tuple)) and \
                    any(f is m._n_mf_updates for f in fetches):
                res = [a for a, _ in pkg._mf_cur]
                tol, sweeps = pkg._mf_cur[-1][1], len(res) - 1
                ended_by_tol = not res[-1] > tol
                longer = ended_by_tol and sweeps < m.max_mf_updates and res[-1] > 0.0 and res[-1] + MF_NOISE > tol
                shorter = sweeps >= 1 and res[-2] - MF_NOISE <= tol
                pkg.mf_loops.append((tag[0], tag[1], sweeps, float(shorter), float(longer), res[-1],
                                     res[-2] if sweeps >= 1 else np.nan))
            pkg._mf_cur = []
            return out
        tf.Session.run = run
        train0, val0 = DBM._train_epoch, DBM._run_val_metrics

        def train(model, X):
            pkg._mf_line += 1
            pkg._mf_model, pkg._mf_tag = model, (pkg._mf_line, 0)
            try:
                return train0(model, X)
            finally:
                pkg._mf_tag = None

        def val(model, X_val):
            pkg._mf_model, pkg._mf_tag = model, (pkg._mf_line, 1)
            try:
                return val0(model, X_val)
            finally:
                pkg._mf_tag = None
        DBM._train_epoch, DBM._run_val_metrics = train, val

    def reset(self):
        self.margin, self.n_draw_ops, self.labels, self.label = np.inf, 0, ['start'], 0
        self.ties, self.tie_scopes, self.n_draws = [], [], 0
        self.mf_loops, self._mf_cur, self._mf_line, self._mf_model, self._mf_tag = [], [], -1, None, None

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
    if 'metrics_n_mf_updates' in out:
        out['mf_loops'] = np.asarray(pkg.mf_loops, dtype=np.float64).reshape(-1, 7)
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
