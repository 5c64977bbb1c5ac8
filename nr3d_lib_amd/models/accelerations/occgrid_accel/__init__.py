from .single import *  # noqa: F401,F403
