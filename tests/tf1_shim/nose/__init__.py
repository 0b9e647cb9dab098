"""stand-in for `nose` (the reference's utils/testing.py imports it at module level; not installable offline)"""


class tools(object):
    @staticmethod
    def nottest(f):
        return f


def run(*a, **kw):
    raise RuntimeError('nose is not available: the reference tests are driven by tests/test_reference_shim.py')
