from .aabb import *  # noqa: F401,F403
from .forest import *  # noqa: F401,F403
from .batched import *  # noqa: F401,F403
