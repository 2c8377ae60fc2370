"""The reference's CLI flow (video_super_resolution/scripts/inference_sr.py:30-84: load -> preprocess -> test ->
tensor2vid -> adain_color_fix -> save) through this repo's drop-in script, on a reduced model."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.mark.gpu
def test_inference_script_end_to_end(tmp_path):
    import frames_oracle as fo
    from star_amd.topology import SMALL_TEST_CONFIG, random_state_dict
    from star_amd.vae_topology import VaeConfig, random_vae_state_dict
    from video_super_resolution.scripts.inference_sr import STAR
    from video_to_video.utils.seed import setup_seed
    ucfg = SMALL_TEST_CONFIG
    vcfg = VaeConfig(block_out_channels=(64, 64, 128, 128))
    torch.save({"state_dict": random_state_dict(ucfg, seed=0)}, tmp_path / "unet.pt")      # the reference wraps it (:38-40)
    torch.save(random_vae_state_dict(vcfg, seed=0), tmp_path / "vae.pt")
    g = torch.Generator().manual_seed(5)
    torch.save(torch.randn(1, 77, ucfg.context_dim, generator=g), tmp_path / "neg.pt")
    prompt = torch.randn(1, 77, ucfg.context_dim, generator=g)
    clip = (torch.rand(3, 24, 40, 3, generator=g) * 255).to(torch.uint8).numpy()            # 3 frames of 24 x 40 RGB
    np.save(tmp_path / "clip.npy", clip)
    star = STAR(result_dir=str(tmp_path / "out"), file_name="clip.mp4", model_path=str(tmp_path / "unet.pt"),
                vae_path=str(tmp_path / "vae.pt"), solver_mode="normal", steps=2, upscale=4, dtype="f16",
                negative_embedding=str(tmp_path / "neg.pt"), unet_config=ucfg, vae_config=vcfg)
    saved = star.enhance_a_video(str(tmp_path / "clip.npy"), prompt)
    assert os.path.isfile(saved)
    if saved.endswith(".npy"):                       # no ffmpeg in this image: frames are written as an array
        frames = np.load(saved)
        assert frames.shape == (3, 96, 160, 3) and frames.dtype == np.uint8
        # the same run by hand: test() then the reference's two post-processing calls on the CPU oracle
        from inference_utils import preprocess
        lr = preprocess([f[:, :, ::-1] for f in clip])      # preprocess takes BGR frames (the reference's cv2 contract)
        setup_seed(666)
        with torch.no_grad():
            out = star.model.test({"video_data": lr.cuda(), "y": prompt, "target_res": (96, 160)}, 900, steps=2,
                                  solver_mode="normal", guide_scale=7.5, max_chunk_len=32)
        want = fo.postprocess(out, lr).numpy().astype(np.uint8)
        diff = np.abs(frames.astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.01        # uint8 truncation of values within 5e-3 of each other


@pytest.mark.gpu
def test_prompt_string_path_with_a_text_encoder(tmp_path):
    """the default CLI flow hands test() a prompt STRING (+ the fixed positive prompt) and encodes the negative prompt at
    construction (video_to_video_model.py:65-70, inference_sr.py:55): both go through opt.text_encoder."""
    from star_amd.topology import SMALL_TEST_CONFIG, random_state_dict
    from star_amd.vae_topology import VaeConfig, random_vae_state_dict
    from video_super_resolution.scripts.inference_sr import STAR
    ucfg = SMALL_TEST_CONFIG
    vcfg = VaeConfig(block_out_channels=(64, 64, 128, 128))
    torch.save({"state_dict": random_state_dict(ucfg, seed=0)}, tmp_path / "unet.pt")
    torch.save(random_vae_state_dict(vcfg, seed=0), tmp_path / "vae.pt")
    seen = []

    def encoder(text):
        seen.append(text)
        g = torch.Generator().manual_seed(len(text))
        return torch.randn(1, 77, ucfg.context_dim, generator=g)

    g = torch.Generator().manual_seed(5)
    np.save(tmp_path / "clip.npy", (torch.rand(2, 24, 40, 3, generator=g) * 255).to(torch.uint8).numpy())
    star = STAR(result_dir=str(tmp_path / "out"), file_name="clip.mp4", model_path=str(tmp_path / "unet.pt"),
                vae_path=str(tmp_path / "vae.pt"), solver_mode="normal", steps=2, upscale=4, dtype="f16",
                unet_config=ucfg, vae_config=vcfg, text_encoder=encoder)
    assert seen == [star.model.negative_prompt]
    saved = star.enhance_a_video(str(tmp_path / "clip.npy"), "a good video")
    assert os.path.isfile(saved) and seen[-1] == "a good video" + star.model.positive_prompt
