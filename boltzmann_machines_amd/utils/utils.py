"""Host-side helpers of the fit loops and of the AIS post-processing.

Same names, arguments and results as the reference's
boltzmann_machines/utils/utils.py (:13-52 batch/epoch iteration, :108-170
log-domain reductions); progress bars are dropped (no tqdm dependency).

The four log-domain reductions (log_sum_exp, log_mean_exp, log_diff_exp, log_std_exp) ARE the reference's bodies: three- to
five-line host formulas whose float64 results - including the order of Python's built-in `sum` / `max` - the AIS
post-processing of DBM.log_Z and the reference's doctests pin; a re-expression would change the last bits for no gain.
Nothing here is on the device path.
"""
import numpy as np


def write_during_training(s):
    print(s, flush=True)


def batch_iter(X, batch_size=10, verbose=False, desc='epoch'):
    """Divide input data into consecutive batches; the last one may be short
    (reference utils.py:13-42).

    >>> X = np.arange(36).reshape((12, 3))
    >>> [len(b) for b in batch_iter(X, batch_size=5)]
    [5, 5, 2]
    """
    X = np.asarray(X)
    N = len(X)
    n_batches = N // batch_size + (N % batch_size > 0)
    for i in range(n_batches):
        yield X[i * batch_size:(i + 1) * batch_size]


def epoch_iter(start_epoch, max_epoch, verbose=False):
    """1-based epoch counter continuing after `start_epoch` (reference utils.py:44-49)."""
    for epoch in range(start_epoch + 1, max_epoch + 1):
        yield epoch


def make_list_from(x):
    return list(x) if hasattr(x, '__iter__') else [x]


def log_sum_exp(x):
    """log(sum(exp(x))), max-shifted (reference utils.py:108-125).

    >>> round(float(log_sum_exp([0, 1, 0])), 3)
    1.551
    >>> round(float(log_sum_exp([1000, 1001, 1000])), 3)
    1001.551
    """
    x = np.asarray(x)
    a = max(x)
    return a + np.log(sum(np.exp(x - a)))


def log_mean_exp(x):
    """
    >>> str(float(log_mean_exp([1, 2, 3])))[:5]     # reference doctest: 2.308...
    '2.308'
    """
    return log_sum_exp(x) - np.log(len(x))


def log_diff_exp(x):
    """log(diff(exp(x))) (reference utils.py:139-151).

    >>> np.round(log_diff_exp([1, 2, 3]), 4).tolist()
    [1.5413, 2.5413]
    """
    x = np.asarray(x)
    a = max(x)
    return a + np.log(np.diff(np.exp(x - a)))


def log_std_exp(x, log_mean_exp_x=None):
    """log(std(exp(x))) (reference utils.py:153-170).

    >>> round(float(log_std_exp(np.arange(8.))), 4)
    5.8754
    """
    x = np.asarray(x)
    m = log_mean_exp_x
    if m is None:
        m = log_mean_exp(x)
    M = log_mean_exp(2. * x)
    return 0.5 * log_diff_exp([2. * m, M])[0]
