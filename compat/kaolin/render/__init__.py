from . import spc  # noqa: F401
