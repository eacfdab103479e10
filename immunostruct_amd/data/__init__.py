from .synthetic_dataset import SyntheticImmunoDataset, SyntheticPairedDataset  # noqa: F401
from .utils import *  # noqa: F401,F403
