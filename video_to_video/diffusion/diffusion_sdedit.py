from star_amd.diffusion.diffusion_sdedit import GaussianDiffusion  # noqa: F401
