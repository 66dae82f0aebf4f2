from .communication_op import *  # noqa: F401,F403
from .parallel_state import *  # noqa: F401,F403
