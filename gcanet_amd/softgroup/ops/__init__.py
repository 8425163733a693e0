from .functions import *  # noqa: F401,F403  (same star-export as softgroup/ops/__init__.py)
