from .lotd import *      # noqa: F401,F403
from .lotd_cfg import *  # noqa: F401,F403
from .lotd_helpers import *  # noqa: F401,F403
from .lotd_encoding import *  # noqa: F401,F403
from .lotd_batched import *  # noqa: F401,F403
from .lotd_forest import *  # noqa: F401,F403
