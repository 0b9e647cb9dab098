"""stand-in for `seaborn` (imported by the reference's utils/plot_utils.py at module level; plotting is never called)"""


def __getattr__(name):
    raise AttributeError('seaborn stand-in of the TF-1 shim: plotting is out of scope (%s)' % name)
