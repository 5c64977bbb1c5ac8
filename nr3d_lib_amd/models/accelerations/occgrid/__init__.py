from .utils import *  # noqa: F401,F403
from .ema_single import *  # noqa: F401,F403
from .ema_batched import *  # noqa: F401,F403
from .getter import *  # noqa: F401,F403
