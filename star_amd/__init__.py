"""star_amd -- MI355X-native (gfx950) implementation of STAR's per-chunk denoising
hot path (spatial-temporal UNet + VideoControlNet, SVD temporal VAE) behind the
reference's own entry points.  Hand-written HIP kernels live in star_amd/csrc and
are reached through the C ABI in include/star_hip.h (ctypes binding: star_amd.lib).
"""
__version__ = "0.1.0"
