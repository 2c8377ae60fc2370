from star_amd.diffusion.schedules_sdedit import noise_schedule  # noqa: F401
