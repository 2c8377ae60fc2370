from .schedules_sdedit import noise_schedule  # noqa: F401
from .diffusion_sdedit import GaussianDiffusion  # noqa: F401
