"""CLI drop-in for the reference's video_super_resolution/scripts/inference_sr.py (B5): same flags, same flow
(load video -> preprocess -> VideoToVideo_sr.test -> colour fix -> save), denoiser + VAE on the MI355X HIP path.

Extra flags (needed offline): --vae_path (diffusers AutoencoderKLTemporalDecoder state dict), --dtype f16|bf16,
--prompt_embedding / --negative_embedding (.pt [1,77,1024]) when open_clip is unavailable.
"""
import os
import sys
from argparse import ArgumentParser

import torch

base_path = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, base_path)
from inference_utils import collate_fn, load_video, preprocess, save_video, tensor2vid  # noqa: E402,F401
from video_super_resolution.color_fix import adain_color_fix  # noqa: E402,F401  (re-exported like the reference script)
from star_amd.frames import tensor2vid_color_fix  # noqa: E402
from video_to_video.utils.seed import setup_seed  # noqa: E402
from video_to_video.video_to_video_model import VideoToVideo_sr  # noqa: E402


class STAR:
    def __init__(self, result_dir="./results/", file_name="000_video.mp4", model_path="", solver_mode="fast", steps=15,
                 guide_scale=7.5, upscale=4, max_chunk_len=32, vae_path="", dtype="f16", negative_embedding="", **model_opts):
        self.model_path, self.result_dir, self.file_name = model_path, result_dir, file_name
        os.makedirs(self.result_dir, exist_ok=True)
        opt = dict(model_path=model_path, vae_path=vae_path, dtype={"f16": torch.float16, "bf16": torch.bfloat16}[dtype])
        if negative_embedding:
            opt["negative_y"] = torch.load(negative_embedding)
        opt.update(model_opts)      # e.g. unet_config / vae_config of a reduced model, text_encoder, rng
        self.model = VideoToVideo_sr(opt)
        steps = 15 if solver_mode == "fast" else steps       # reference: `fast` forces 15 (inference_sr.py:43)
        self.solver_mode, self.steps, self.guide_scale = solver_mode, steps, guide_scale
        self.upscale, self.max_chunk_len = upscale, max_chunk_len

    def enhance_a_video(self, video_path, prompt):
        text = prompt if torch.is_tensor(prompt) else prompt + self.model.positive_prompt
        input_frames, input_fps = load_video(video_path)
        video_data = preprocess(input_frames)
        _, _, h, w = video_data.shape
        target_h, target_w = h * self.upscale, w * self.upscale
        pre_data = {"video_data": video_data, "y": text, "target_res": (target_h, target_w)}
        total_noise_levels = 900
        setup_seed(666)
        with torch.no_grad():
            data_tensor = collate_fn(pre_data, "cuda:0")
            output = self.model.test(data_tensor, total_noise_levels, steps=self.steps, solver_mode=self.solver_mode,
                                     guide_scale=self.guide_scale, max_chunk_len=self.max_chunk_len, return_device=True)
            # tensor2vid + adain_color_fix (inference_sr.py:47-48) + save_video's uint8 truncation in one pass over the frames while
            # they are still in HBM: [F, H, W, 3] bytes cross PCIe (157 MB at cfg2 instead of 628 MB of fp32)
            output = tensor2vid_color_fix(output, data_tensor["video_data"], ctx=self.model.generator.ctx, as_uint8=True)
        return save_video(output.cpu(), self.result_dir, self.file_name, fps=input_fps)


def parse_args():
    p = ArgumentParser()
    p.add_argument("--input_path", required=True, type=str)
    p.add_argument("--save_dir", type=str, default="results")
    p.add_argument("--file_name", type=str)
    p.add_argument("--model_path", type=str, default="./pretrained_weight/model.pt")
    p.add_argument("--prompt", type=str, default="a good video")
    p.add_argument("--upscale", type=int, default=4)
    p.add_argument("--max_chunk_len", type=int, default=32)
    p.add_argument("--cfg", type=float, default=7.5)
    p.add_argument("--solver_mode", type=str, default="fast")
    p.add_argument("--steps", type=int, default=15)
    p.add_argument("--vae_path", type=str, default="./pretrained_weight/svd_vae.safetensors")
    p.add_argument("--dtype", type=str, default="f16", choices=["f16", "bf16"])
    p.add_argument("--prompt_embedding", type=str, default="")
    p.add_argument("--negative_embedding", type=str, default="")
    p.add_argument("--text_encoder_path", type=str, default="",
                   help="open_clip ViT-H-14 state dict for the text tower (tokenisation still needs the open_clip package)")
    return p.parse_args()


def main():
    a = parse_args()
    assert a.solver_mode in ("fast", "normal")
    star = STAR(result_dir=a.save_dir, file_name=a.file_name or os.path.basename(a.input_path), model_path=a.model_path,
                solver_mode=a.solver_mode, steps=a.steps, guide_scale=a.cfg, upscale=a.upscale, max_chunk_len=a.max_chunk_len,
                vae_path=a.vae_path, dtype=a.dtype, negative_embedding=a.negative_embedding,
                **({"text_state_dict": a.text_encoder_path} if a.text_encoder_path else {}))
    prompt = torch.load(a.prompt_embedding) if a.prompt_embedding else a.prompt
    print("saved", star.enhance_a_video(a.input_path, prompt))


if __name__ == "__main__":
    main()
