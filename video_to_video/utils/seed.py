"""setup_seed of the reference (video_to_video/utils/seed.py:9): seed python, numpy and torch RNGs."""
import random

import numpy as np
import torch


def setup_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.backends.cudnn.deterministic = True
