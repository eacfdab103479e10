from .augment import *  # noqa: F401,F403
from .device_dataset import DeviceResidentDataset  # noqa: F401
from .packed import *  # noqa: F401,F403
from .synthetic_dataset import SyntheticImmunoDataset, SyntheticPairedDataset  # noqa: F401
from .utils import *  # noqa: F401,F403
from .reference_inputs import *  # noqa: F401,F403
from . import tables  # noqa: F401
