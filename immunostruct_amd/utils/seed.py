"""``seed_everything`` (reference ``utils/seed.py:7-20``): seeds python / numpy / torch and turns on
deterministic algorithms -- which this package honours: every HIP reduction has a fixed order."""
import os
import random

import numpy as np
import torch


def seed_everything(seed) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.use_deterministic_algorithms(True, warn_only=True)
