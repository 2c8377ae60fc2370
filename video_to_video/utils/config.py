"""The three live keys of the reference's global config (video_to_video/utils/config.py:160-169)."""
from types import SimpleNamespace

from star_amd.video_to_video_model import NEGATIVE_PROMPT, POSITIVE_PROMPT

cfg = SimpleNamespace(model_path=None, positive_prompt=POSITIVE_PROMPT, negative_prompt=NEGATIVE_PROMPT)
