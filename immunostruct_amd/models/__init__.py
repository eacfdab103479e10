from .ablation_models import *  # noqa: F401,F403
from .comparative_models import *  # noqa: F401,F403
from .hybrid_models import *  # noqa: F401,F403
from .mapping import model_map  # noqa: F401
