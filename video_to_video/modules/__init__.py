from star_amd.modules.unet_v2v import ControlledV2VUNet  # noqa: F401
