"""Video I/O helpers with the reference's names (inference_utils.py:16-147).  cv2 / ffmpeg / torchvision are not in this
image, so frames can also be exchanged as .npy / .pt tensors; OpenCV and ffmpeg are used when present."""
import logging
import os
import shutil
import subprocess
import tempfile  # noqa: F401  (kept: callers of the reference module import it from here)
import threading

import numpy as np
import torch


def tensor2vid(video, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """[1, 3, F, H, W] in ~[-1, 1] -> [F, H, W, 3] in 0..255 (inference_utils.py:16-23): a layout change and an affine
    map, left to torch on whatever device holds the tensor.  The inference script does not call it: it uses the fused
    star_amd.frames.tensor2vid_color_fix (one HIP pass for tensor2vid + adain_color_fix) instead."""
    m = torch.tensor(mean, device=video.device).reshape(1, -1, 1, 1, 1)
    s = torch.tensor(std, device=video.device).reshape(1, -1, 1, 1, 1)
    video = (video * s + m).clamp_(0, 1) * 255.0
    return video[0].permute(1, 2, 3, 0)


def preprocess(input_frames):
    """list of HxWx3 uint8 frames in OpenCV's BGR order, as `load_video` returns them -> [F, 3, H, W] RGB in [-1, 1].
    Same contract as the reference (inference_utils.py:26-40: `frame[:, :, ::-1]`, to_tensor, clamp, (x - 0.5) / 0.5)."""
    out = [torch.from_numpy(np.ascontiguousarray(np.asarray(f)[:, :, ::-1])).permute(2, 0, 1).float() / 255.0 for f in input_frames]
    return (torch.stack(out).clamp_(0, 1) - 0.5) / 0.5


def adjust_resolution(h, w, up_scale):
    """target size rule of the reference (inference_utils.py:43-55): at least 720 rows, at most 1280x2048 pixels, even sides."""
    if h * up_scale < 720:
        s = 720 / h
    elif h * w * up_scale * up_scale > 1280 * 2048:
        s = float(np.sqrt(1280 * 2048 / (h * w)))
    else:
        s = up_scale
    return int(s * h // 2 * 2), int(s * w // 2 * 2)


def load_video(vid_path):
    """-> (list of HxWx3 uint8 frames in BGR order, fps), like the reference's cv2 reader (inference_utils.py:67-86).
    cv2 is not installed in this image: `.npy` / `.pt` files holding [F, H, W, 3] uint8 RGB frames are accepted as well
    (and handed out in BGR, so that `preprocess` applies to every source alike)."""
    if vid_path.endswith(".npy") or vid_path.endswith(".pt"):
        arr = np.load(vid_path) if vid_path.endswith(".npy") else torch.load(vid_path).numpy()
        return [np.ascontiguousarray(f[:, :, ::-1]) for f in arr], 8.0
    try:
        import cv2
    except ImportError as e:
        if _ffmpeg() is None:
            raise RuntimeError("neither OpenCV nor ffmpeg is installed here: pass frames as .npy / .pt ([F, H, W, 3] uint8 RGB)") from e
        return _load_video_ffmpeg(vid_path)
    cap = cv2.VideoCapture(vid_path)
    fps = cap.get(cv2.CAP_PROP_FPS)
    frames = []
    while True:
        ok, frame = cap.read()
        if not ok or frame is None:
            break
        frames.append(frame)
    cap.release()
    return frames, fps


def _ffmpeg():
    return shutil.which("ffmpeg")


def _ffprobe():
    """ffprobe next to the ffmpeg that was found (never by rewriting the path string), else whatever is on PATH."""
    exe = _ffmpeg()
    if exe:
        cand = os.path.join(os.path.dirname(exe), "ffprobe" + os.path.splitext(exe)[1])
        if os.path.isfile(cand):
            return cand
    return shutil.which("ffprobe")


def _load_video_ffmpeg(vid_path):
    """decode through an ffmpeg pipe (bgr24 raw frames) when cv2 is missing: frame geometry and rate from ffprobe, then the raw
    frames are READ FRAME BY FRAME from ffmpeg's stdout by a reader thread (no whole-clip capture buffer)."""
    probe_exe, exe = _ffprobe(), _ffmpeg()
    if not probe_exe or not exe:
        raise RuntimeError("load_video: neither cv2 nor an ffmpeg / ffprobe pair is available")
    probe = subprocess.run([probe_exe, "-v", "error", "-select_streams", "v:0", "-show_entries", "stream=width,height,r_frame_rate",
                            "-of", "csv=p=0", vid_path], capture_output=True, text=True)
    fields = probe.stdout.strip().split(",")
    if probe.returncode != 0 or len(fields) < 3:
        raise RuntimeError(f"load_video: ffprobe failed on {vid_path!r} (rc {probe.returncode}): {probe.stderr.strip()[-500:]}")
    w, h = int(fields[0]), int(fields[1])
    num, _, den = fields[2].partition("/")
    try:
        fps = float(num) / float(den) if den.strip() else float(num)
    except (ValueError, ZeroDivisionError):
        fps = 0.0
    if not fps > 0:
        raise RuntimeError(f"load_video: cannot parse the frame rate {fields[2]!r} of {vid_path!r}")
    nbytes = h * w * 3
    proc = subprocess.Popen([exe, "-v", "error", "-i", vid_path, "-f", "rawvideo", "-pix_fmt", "bgr24", "-"], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE)
    frames, tail = [], [b""]

    def reader():
        while True:
            buf = proc.stdout.read(nbytes)
            if len(buf) < nbytes:
                tail[0] = buf
                return
            frames.append(np.frombuffer(buf, dtype=np.uint8).reshape(h, w, 3))

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    err = proc.stderr.read()
    th.join()
    rc = proc.wait()
    if rc != 0 or not frames or tail[0]:
        raise RuntimeError(f"load_video: ffmpeg decode of {vid_path!r} failed (rc {rc}, {len(frames)} whole frames, {len(tail[0])} stray bytes): "
                           f"{err.decode(errors='replace').strip()[-500:]}")
    return frames, fps


def save_video(video, save_dir, file_name, fps=16.0):
    """[F, H, W, 3] RGB in 0..255 (float or uint8; truncated like the reference's astype('uint8')) -> lossless H.264
    (`-crf 0`, the reference's encoder settings, inference_utils.py:89-106).  The reference writes one PNG per frame into a
    temp directory and lets ffmpeg read them back; here the raw RGB frames are streamed into ffmpeg's stdin from a writer
    thread (no PNG encode + decode per frame, no temp files), which is what dominates once the model is fast (SURVEY.md
    section 8f rank 2).  Returns the path actually written: without an ffmpeg binary (this image) the frames go to
    `<file_name minus extension>.npy` in the same directory, and the reason is logged."""
    os.makedirs(save_dir, exist_ok=True)
    out_path = os.path.join(save_dir, file_name)
    arr = video.cpu().numpy() if torch.is_tensor(video) else np.asarray(video)
    arr = np.ascontiguousarray(arr.astype(np.uint8))
    log = logging.getLogger("star_amd")
    exe = _ffmpeg()
    if exe is not None and arr.ndim == 4 and arr.shape[-1] == 3:
        n, h, w, _ = arr.shape
        cmd = [exe, "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{w}x{h}", "-framerate", str(fps), "-i", "-",
               "-vcodec", "libx264", "-preset", "ultrafast", "-crf", "0", "-pix_fmt", "yuv420p", out_path]
        proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)

        def feed():
            try:
                for f in arr:
                    proc.stdin.write(f.tobytes())
            except (BrokenPipeError, OSError):
                pass
            finally:
                proc.stdin.close()

        t = threading.Thread(target=feed, daemon=True)
        t.start()
        err = proc.stderr.read()
        t.join()
        if proc.wait() == 0:
            return out_path
        log.error("save_video: ffmpeg failed (%d): %s", proc.returncode, err.decode(errors="replace")[-400:])
    else:
        log.warning("save_video: no ffmpeg binary; writing frames as .npy instead of %s", out_path)
    npy = os.path.splitext(out_path)[0] + ".npy"
    np.save(npy, arr)
    return npy


def collate_fn(data, device):
    """move the tensors of a sample dict to `device` (inference_utils.py:109-147)."""
    if isinstance(data, dict):
        return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    if torch.is_tensor(data):
        return data.to(device)
    return data
