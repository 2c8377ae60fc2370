"""Video I/O helpers with the reference's names (inference_utils.py:16-147).  cv2 / ffmpeg / torchvision are not in this
image, so frames can also be exchanged as .npy / .pt tensors; OpenCV and ffmpeg are used when present."""
import os
import subprocess
import tempfile

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
    """list of HxWx3 uint8 RGB frames -> [F, 3, H, W] in [-1, 1] (inference_utils.py:26-35)."""
    out = [torch.from_numpy(np.asarray(f)).permute(2, 0, 1).float() / 255.0 for f in input_frames]
    return (torch.stack(out) - 0.5) / 0.5


def load_video(vid_path):
    """-> (list of RGB uint8 frames, fps)."""
    if vid_path.endswith(".npy"):
        arr = np.load(vid_path)
        return [f for f in arr], 8.0
    if vid_path.endswith(".pt"):
        arr = torch.load(vid_path)
        return [f.numpy() for f in arr], 8.0
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("OpenCV is not installed here: pass frames as .npy / .pt ([F, H, W, 3] uint8)") from e
    cap = cv2.VideoCapture(vid_path)
    fps = cap.get(cv2.CAP_PROP_FPS)
    frames = []
    while True:
        ok, frame = cap.read()
        if not ok:
            break
        frames.append(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB))
    cap.release()
    return frames, fps


def save_video(video, save_dir, file_name, fps=16.0):
    """[F, H, W, 3] in 0..255 (float or uint8; truncated like the reference's astype('uint8')) -> mp4 through ffmpeg (libx264, crf 0) when available, else a .npy next to it."""
    os.makedirs(save_dir, exist_ok=True)
    out_path = os.path.join(save_dir, file_name)
    arr = video.cpu().numpy() if torch.is_tensor(video) else np.asarray(video)
    arr = arr.astype(np.uint8)
    try:
        from PIL import Image
        tmp = tempfile.mkdtemp()
        for i, f in enumerate(arr):
            Image.fromarray(f).save(os.path.join(tmp, "%06d.png" % (i + 1)))
        cmd = f'ffmpeg -y -f image2 -framerate {fps} -i {tmp}/%06d.png -vcodec libx264 -crf 0 -pix_fmt yuv420p "{out_path}"'
        if subprocess.call(cmd, shell=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0:
            return out_path
    except Exception:
        pass
    np.save(os.path.splitext(out_path)[0] + ".npy", arr)
    return os.path.splitext(out_path)[0] + ".npy"


def collate_fn(data, device):
    """move the tensors of a sample dict to `device` (inference_utils.py:109-147)."""
    if isinstance(data, dict):
        return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    if torch.is_tensor(data):
        return data.to(device)
    return data
