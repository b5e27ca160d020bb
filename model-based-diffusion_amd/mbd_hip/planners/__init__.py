from . import mbd_planner, path_integral  # noqa: F401
