from . import run_mbd  # noqa: F401
