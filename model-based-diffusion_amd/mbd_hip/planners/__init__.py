from . import mbd_planner  # noqa: F401
