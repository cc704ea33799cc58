from . import boxes, misc  # noqa: F401
