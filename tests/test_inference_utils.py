"""Contract of the frame helpers the CLI uses (reference inference_utils.py:16-106): BGR frames out of load_video, the flip
inside preprocess, truncating uint8 conversion and clean-up in save_video."""
import os

import numpy as np
import torch

import inference_utils as iu


def test_load_preprocess_roundtrip_is_rgb(tmp_path):
    rgb = (np.random.RandomState(0).rand(2, 6, 8, 3) * 255).astype(np.uint8)
    np.save(tmp_path / "clip.npy", rgb)
    frames, fps = iu.load_video(str(tmp_path / "clip.npy"))
    assert len(frames) == 2 and np.array_equal(frames[0], rgb[0][:, :, ::-1])          # BGR, as cv2 hands frames out
    x = iu.preprocess(frames)
    assert x.shape == (2, 3, 6, 8)
    want = torch.from_numpy(rgb).permute(0, 3, 1, 2).float() / 255.0 * 2 - 1           # channel 0 is R again
    assert torch.allclose(x, want, atol=1e-6)


def test_adjust_resolution_rule():
    assert iu.adjust_resolution(240, 426, 4) == (960, 1704)
    assert iu.adjust_resolution(100, 160, 4) == (720, 1152)                             # at least 720 rows
    assert iu.adjust_resolution(1080, 1920, 4) == (1214, 2158)                          # at most 1280 x 2048 pixels


def test_save_video_returns_the_written_path(tmp_path):
    vid = torch.rand(3, 8, 8, 3) * 255.9
    path = iu.save_video(vid, str(tmp_path / "out"), "a.mp4", fps=8)
    assert os.path.isfile(path) and os.listdir(tmp_path / "out") == [os.path.basename(path)]     # nothing else left behind
    if path.endswith(".npy"):
        assert np.array_equal(np.load(path), vid.numpy().astype(np.uint8))


def test_save_video_streams_raw_frames_into_ffmpeg(tmp_path, monkeypatch):
    """with an encoder on PATH the frames go down a pipe as rgb24 (no PNG round trip): a stand-in `ffmpeg` records its stdin."""
    fake = tmp_path / "ffmpeg"
    fake.write_text("#!/bin/sh\nfor a in \"$@\"; do out=\"$a\"; done\ncat > \"$out\"\n")
    fake.chmod(0o755)
    monkeypatch.setattr(iu, "_ffmpeg", lambda: str(fake))
    vid = (torch.rand(4, 6, 10, 3) * 255).to(torch.uint8)
    path = iu.save_video(vid, str(tmp_path / "out"), "b.mp4", fps=8)
    assert path.endswith("b.mp4")
    assert open(path, "rb").read() == vid.numpy().tobytes()


def _fake_decoder(tmp_path, w, h, rate, payload, rc=0):
    """a stand-in ffprobe / ffmpeg pair in one directory: the probe prints `w,h,rate`, the decoder writes `payload` to stdout"""
    d = tmp_path / "bin"
    d.mkdir()
    raw = tmp_path / "raw.bin"
    raw.write_bytes(payload)
    (d / "ffprobe").write_text(f"#!/bin/sh\necho '{w},{h},{rate}'\n")
    (d / "ffmpeg").write_text(f"#!/bin/sh\ncat '{raw}'\nexit {rc}\n")
    for f in ("ffprobe", "ffmpeg"):
        (d / f).chmod(0o755)
    return str(d / "ffmpeg")


def test_ffmpeg_decode_path_streams_frames_and_finds_ffprobe_next_to_ffmpeg(tmp_path, monkeypatch):
    """round-2 advisor: the probe path was derived with str.replace over the whole path, no return code was checked and the
    clip was captured whole.  Stand-in binaries in a directory whose NAME contains 'ffmpeg' (the str.replace trap)."""
    base = tmp_path / "opt_ffmpeg_6"
    base.mkdir()
    bgr = (np.random.RandomState(1).rand(3, 4, 6, 3) * 255).astype(np.uint8)
    exe = _fake_decoder(base, 6, 4, "30000/1001", bgr.tobytes())
    monkeypatch.setattr(iu, "_ffmpeg", lambda: exe)
    assert iu._ffprobe() == os.path.join(os.path.dirname(exe), "ffprobe")
    frames, fps = iu._load_video_ffmpeg("clip.mp4")
    assert len(frames) == 3 and all(np.array_equal(f, b) for f, b in zip(frames, bgr)) and abs(fps - 30000 / 1001) < 1e-9


def test_ffmpeg_decode_errors_are_reported(tmp_path, monkeypatch):
    import pytest
    a = tmp_path / "a"
    a.mkdir()
    exe = _fake_decoder(a, 6, 4, "25/1", b"x" * (6 * 4 * 3 + 5))            # a torn last frame
    monkeypatch.setattr(iu, "_ffmpeg", lambda: exe)
    with pytest.raises(RuntimeError, match="stray bytes"):
        iu._load_video_ffmpeg("clip.mp4")
    b = tmp_path / "b"
    b.mkdir()
    exe2 = _fake_decoder(b, 6, 4, "0/0", b"y" * (6 * 4 * 3))                 # an unusable r_frame_rate
    monkeypatch.setattr(iu, "_ffmpeg", lambda: exe2)
    with pytest.raises(RuntimeError, match="frame rate"):
        iu._load_video_ffmpeg("clip.mp4")
