from . import distributions  # noqa: F401
