from .utils import *  # noqa: F401,F403
