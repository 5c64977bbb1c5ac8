from .forest import *  # noqa: F401,F403
