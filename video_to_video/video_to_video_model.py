"""Same import path as the reference's video_to_video/video_to_video_model.py (B1): the HIP-backed pipeline object."""
from star_amd.video_to_video_model import (VideoToVideo_sr, make_chunks, pad_to_fit,  # noqa: F401
                                           sliding_windows_1d)
