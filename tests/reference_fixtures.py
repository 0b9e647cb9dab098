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


FORK_EPS = 3e-7            # |u - p| below which the float32 round-off of another summation order can flip the draw
MAX_FORKED_FRACTION = 0.15  # of the rows of a row-local output (64 AIS chains x 2.3e6 draws each: 3-5 expected)


def tolerances(name):
    """north_star's 1e-5 float32 relative for every scenario; float64 at its own round-off"""
    from tests.golden import scenarios
    if name in ('rbm_float64', 'dbm_float64'):
        tol = dict(rtol=1e-11, metrics_rtol=1e-7, atol=1e-15)      # progress lines carry 9 significant digits
    else:
        tol = dict(rtol=1e-5, metrics_rtol=1e-5)
    return tol


def check(name, got):
    """compare a scenario run with the committed fixture of the reference; returns the report lines"""
    from tests.golden import scenarios
    return compare(name, got, load(name), row_local=scenarios.ROW_LOCAL.get(name),
                   row_aggregate=scenarios.ROW_AGGREGATE.get(name), **tolerances(name))


BOOKKEEPING = ('min_bernoulli_margin', 'near_ties', 'near_tie_labels', 'near_tie_scopes', 'n_bernoulli_draws', 'mf_loops')


def mf_trip_bounds(ref):
    """[lines, 2, 3] = (lowest, highest, flagged loops) mean trip count the progress-line entry (line, train / validation)
    may show: the reference's own counts, widened by ONE sweep for every loop the generator recorded as ending at the float32
    noise floor of its tolerance (tests/golden/make_golden_from_reference.py, `mf_loops`).  NaN where no loop was recorded
    (the entry repeats the last validation run of the same fit, or is absent)."""
    loops = np.asarray(ref.get('mf_loops', np.zeros((0, 7)))).reshape(-1, 7)
    n_lines = int(np.asarray(ref['metrics_n_mf_updates']).shape[0])
    out = np.full((n_lines, 2, 3), np.nan)
    for line in range(n_lines):
        for col in (0, 1):
            sel = loops[(loops[:, 0] == line) & (loops[:, 1] == col)]
            if len(sel):
                out[line, col] = (np.mean(sel[:, 2] - sel[:, 3]), np.mean(sel[:, 2] + sel[:, 4]), np.sum((sel[:, 3] + sel[:, 4]) > 0))
    return out


def near_ties(ref, label=None):
    """rows (label index, op execution, row, column, u - p) of the draws the generator recorded within TIE_EPS of a
    tie, optionally only those of the public call `label`"""
    t = np.asarray(ref.get('near_ties', np.zeros((0, 5)))).reshape(-1, 5)
    if label is None:
        return t
    labels = [str(x) for x in ref.get('near_tie_labels', [])]
    if label not in labels:
        return t[:0]
    return t[t[:, 0] == labels.index(label)]


def tie_summary(ref):
    t = near_ties(ref)
    labels = [str(x) for x in ref.get('near_tie_labels', [])]
    per = ', '.join('%s: %d' % (labels[int(i)], int((t[:, 0] == i).sum())) for i in sorted(set(t[:, 0].tolist())))
    n = int(np.asarray(ref.get('n_bernoulli_draws', [0]))[0])
    return 'near-ties recorded by the generator: %d of %.3g Bernoulli draws (%s), min |u - p| = %.2e' % (
        len(t), n, per or 'none', float(np.asarray(ref.get('min_bernoulli_margin', [np.inf]))[0]))


def compare(name, got, ref, rtol=1e-5, metrics_rtol=1e-5, atol=1e-7, n_mf_atol=0.0, row_local=None, row_aggregate=None):
    """every array the reference returned, by name: same keys (the get_tf_params surface included), same shapes,
    integers / counters exact, reals within `rtol` of the array's scale (north_star: 1e-5 fp32 relative) plus `atol`
    = one float32 ulp of the unit-scale intermediates (probabilities, states) - momentum buffers of the biases are
    means of DIFFERENCES of such quantities and inherit their absolute, not their relative, round-off.

    Near-ties (hard part 2).  `row_local` = {output: label}: the rows of that output are independent
    trajectories (minibatch rows of `transform`, AIS chains) produced by the public call `label`.  A row may differ
    from the fixture ONLY if the generator recorded a draw within TIE_EPS of a tie on that row under that label (the
    float32 round-off of another summation order can flip exactly such a draw, and the row's trajectory forks there);
    every other row must agree within the tolerance.  `row_aggregate` = {output: row-local output it is a function
    of}: compared only when no row of its source forked.  Every other output depends on ALL trajectories of the run
    (the parameters after `fit`): there a fork anywhere fails the comparison - the generator's seeds are chosen so
    that none happens - and the error names the recorded near-ties.  The report states the counts either way.
    Returns the report lines; raises AssertionError listing every mismatch."""
    row_local, row_aggregate = dict(row_local or {}), dict(row_aggregate or {})
    ties_line = tie_summary(ref)
    full = ref
    ref = {k: v for k, v in ref.items() if k not in BOOKKEEPING}
    errors, report = [], [ties_line]
    forked = {}
    missing, extra = sorted(set(ref) - set(got)), sorted(set(got) - set(ref))
    if missing:
        errors.append('missing outputs: %s' % missing)
    if extra:
        errors.append('outputs the reference does not return: %s' % extra)
    for k in sorted(set(ref) & set(got), key=lambda k: (k not in row_local, k)):   # row-local outputs first
        r, g = np.asarray(ref[k]), np.asarray(got[k])
        if r.shape != g.shape:
            errors.append('%s: shape %s != %s' % (k, g.shape, r.shape))
            continue
        if k in row_aggregate and forked.get(row_aggregate[k]):
            report.append('%-60s not compared: %d row(s) of %s forked at a recorded near-tie'
                          % (k, forked[row_aggregate[k]], row_aggregate[k]))
            continue
        if k in row_local:
            t = near_ties(full, row_local[k])
            t = t[np.abs(t[:, 4]) < FORK_EPS]
            eligible = np.zeros(r.shape[0], dtype=bool)
            eligible[np.unique(t[:, 2].astype(np.int64))] = True
            r2, g2 = r.astype(np.float64).reshape(r.shape[0], -1), g.astype(np.float64).reshape(r.shape[0], -1)
            scale = max(float(np.max(np.abs(r2))), 1e-30)
            bad = np.max(np.abs(r2 - g2), axis=1) > rtol * scale + atol
            forked[k] = int((bad & eligible).sum())
            report.append('%-60s %d rows, %d with a draw within %.0e of a tie, %d of those forked; all other rows %.2e'
                          % (k, r.shape[0], int(eligible.sum()), FORK_EPS, forked[k],
                             float(np.max(np.abs(r2 - g2)[~bad])) / scale if (~bad).any() else 0.0))
            if forked[k] > MAX_FORKED_FRACTION * r.shape[0]:
                errors.append('%s: %d of %d rows differ - more than near-ties explain' % (k, forked[k], r.shape[0]))
            if (bad & ~eligible).any():
                rows = np.flatnonzero(bad & ~eligible)
                errors.append('%s: %d row(s) without a recorded near-tie differ (first: row %d, max |diff| %.3e of scale %.3e)'
                              % (k, rows.size, rows[0], float(np.max(np.abs(r2 - g2)[rows[0]])), scale))
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
        if k == 'metrics_n_mf_updates':              # epoch means of integer trip counts, printed with one decimal
            bounds = mf_trip_bounds(full)
            r2, g2 = np.asarray(ref[k], dtype=np.float64), np.asarray(got[k], dtype=np.float64)
            lo, hi = r2 - n_mf_atol, r2 + n_mf_atol
            have = ~np.isnan(bounds[:, :r2.shape[1], 0])
            lo[have] = np.minimum(lo[have], bounds[:, :r2.shape[1], 0][have] - 0.05 - n_mf_atol)   # 0.05: the `.1f` of the progress line
            hi[have] = np.maximum(hi[have], bounds[:, :r2.shape[1], 1][have] + 0.05 + n_mf_atol)
            nflag = int(np.nansum(bounds[:, :, 2]))
            exact = bool(np.array_equal(np.isnan(r2), np.isnan(g2)) and np.array_equal(np.nan_to_num(r2), np.nan_to_num(g2)))
            report.append('%-60s reference %s, here %s: %s' % (
                k, np.round(r2.reshape(-1), 2).tolist(), np.round(g2.reshape(-1), 2).tolist(),
                'EXACT' if exact else 'within the one sweep of the %d loop(s) recorded at the noise floor of mf_tol' % nflag))
            ok = np.array_equal(np.isnan(r2), np.isnan(g2)) and bool(np.all((np.nan_to_num(g2) >= np.nan_to_num(lo) - 1e-9) &
                                                                          (np.nan_to_num(g2) <= np.nan_to_num(hi) + 1e-9)))
            if not ok:
                errors.append('%s: executed mean-field sweeps %s outside [%s, %s] (reference %s; %d loop(s) recorded at the noise floor)'
                              % (k, g2.tolist(), lo.tolist(), hi.tolist(), r2.tolist(), nflag))
            continue
        report.append('%-60s %.2e (abs %.2e)' % (k, diff / scale, diff))
        if not diff <= tol * scale + atol:
            errors.append('%s: max |diff| = %.3e (%.3e of max |ref|) > %.1e * scale + %.1e' % (k, diff, diff / scale, tol, atol))
    if errors:
        raise AssertionError('%s does not reproduce the reference fixture:\n  ' % name + '\n  '.join(errors) +
                             '\n  ' + ties_line + '\n  (a mismatch of an output that aggregates over all trajectories '
                             'can be a fork at one of these; tests/golden/make_golden_from_reference.py explains)')
    return report
