from .infer import *  # noqa: F401,F403
from .train import *  # noqa: F401,F403
