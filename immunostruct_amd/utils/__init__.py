from .contrastive import PairedContrastiveLoss  # noqa: F401
from .loss import Losses  # noqa: F401
from .scheduler import LinearWarmupCosineAnnealingLR  # noqa: F401
from .seed import seed_everything  # noqa: F401
from .update_paths import update_paths  # noqa: F401
