from .infer import *  # noqa: F401,F403
from .metric import *  # noqa: F401,F403
from .train import *  # noqa: F401,F403
from .train_SSL import *  # noqa: F401,F403
