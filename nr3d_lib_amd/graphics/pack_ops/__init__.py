from .pack_ops import *  # noqa: F401,F403
