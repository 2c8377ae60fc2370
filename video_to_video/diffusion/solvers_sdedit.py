from star_amd.diffusion.solvers_sdedit import sample_dpmpp_2m_sde  # noqa: F401
