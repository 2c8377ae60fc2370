"""`VideoToVideo_sr`: the pipeline object = the drop-in boundary B1.

Mirrors video_to_video/video_to_video_model.py:20-161 -- same constructor / `test()` signature, same order of
operations and RNG consumption (VAE posterior sample per frame -> randn_like in diffuse -> randint for the solver
noise seed) -- with the denoiser and the VAE running as hand-written HIP kernels behind include/star_hip.h.

Differences forced by the offline image (all explicit, none silent):
  * open_clip is not installed: `input['y']` / `negative_y` may be given as precomputed [1, 77, 1024] embeddings;
    a prompt *string* needs `opt.text_encoder` (any callable str -> [1, 77, 1024]), an importable open_clip, or
    `opt.text_state_dict` + `opt.tokenizer` (star_amd/modules/embedder.py restates the OpenCLIP text tower).
  * diffusers / HF hub are not available: the VAE weights come from `opt.vae_path` (a state dict with diffusers'
    AutoencoderKLTemporalDecoder keys) or `opt.vae_state_dict`.
  * `opt.dtype` selects fp16 (reference: generator.half() + autocast, :42,98) or bf16 storage for the HIP path.
"""
import os
from typing import Any, Dict

import torch
import torch.nn.functional as F

from . import frames
from . import lib as L
from .diffusion import GaussianDiffusion, noise_schedule
from .geometry import make_chunks, pad_to_fit, sliding_windows_1d  # noqa: F401  (re-exported like the reference module)
from .modules.unet_v2v import ControlledV2VUNet
from .topology import UNetConfig
from .vae import AutoencoderKLTemporalDecoder
from .vae_topology import VaeConfig

POSITIVE_PROMPT = ("Cinematic, High Contrast, highly detailed, taken using a Canon EOS R camera, hyper detailed photo - "
                   "realistic maximum detail, 32k, Color Grading, ultra HD, extreme meticulous detailing, skin pore detailing, "
                   "hyper sharpness, perfect without deformations.")
NEGATIVE_PROMPT = ("painting, oil painting, illustration, drawing, art, sketch, oil painting, cartoon, CG Style, 3D render, "
                   "unreal engine, blurring, dirty, messy, worst quality, low quality, frames, watermark, signature, jpeg "
                   "artifacts, deformed, lowres, over-smooth")


def _opt(opt, name, default=None):
    if isinstance(opt, dict):
        return opt.get(name, default)
    return getattr(opt, name, default)


def _cpu_noise_sampler(gen):
    """solver noise source drawing N(0,1) tensors from a CPU generator (identical on every rank / device)."""
    class _Sampler:
        def __init__(self, x, sigma_min, sigma_max, seed=None):
            self.shape, self.device = x.shape, x.device

        def __call__(self, sigma, sigma_next):
            return torch.randn(self.shape, generator=gen, dtype=torch.float32).to(self.device)
    return _Sampler


class VideoToVideo_sr:
    def __init__(self, opt, device=torch.device("cuda:0")):
        self.opt = opt
        self.device = torch.device(device)
        dtype = _opt(opt, "dtype", torch.float16)
        # 'cuda' without an index = this rank's current device (never silently GPU 0)
        dev_index = L.device_index(self.device)
        library = _opt(opt, "library")             # tests may pass the emulator build explicitly

        # text encoder (video_to_video_model.py:26-29).  opt.text_encoder: any callable str -> [1, 77, 1024]; otherwise the
        # OpenCLIP wrapper (star_amd/modules/embedder.py) from open_clip itself or from opt.text_state_dict (+ opt.tokenizer).
        # Built lazily: a pipeline that only ever sees precomputed embeddings never needs it.
        self.text_encoder = _opt(opt, "text_encoder")
        self._text_encoder_error = None
        if self.text_encoder is None:
            try:
                from .modules.embedder import FrozenOpenCLIPEmbedder
                self.text_encoder = FrozenOpenCLIPEmbedder(device=self.device, pretrained="laion2b_s32b_b79k",
                                                           text_state_dict=_opt(opt, "text_state_dict"), tokenizer=_opt(opt, "tokenizer"),
                                                           dtype=dtype, library=library, runtime=_opt(opt, "text_runtime"))
            except ImportError as e:     # open_clip missing and no weights / tokenizer given: only prompt STRINGS are affected
                self._text_encoder_error = str(e)

        # U-Net with ControlNet (:32-43)
        unet_cfg = _opt(opt, "unet_config", UNetConfig())
        generator = ControlledV2VUNet(unet_cfg, dtype=dtype, device=dev_index, library=library)
        sd = _opt(opt, "state_dict")
        if sd is None:
            model_path = _opt(opt, "model_path")
            if not model_path:
                raise ValueError("opt.model_path (a reference .pt checkpoint) or opt.state_dict is required")
            sd = torch.load(model_path, map_location="cpu")
            if "state_dict" in sd:
                sd = sd["state_dict"]
        generator.load_state_dict(sd, strict=False)
        generator.release_host_weights()
        self.generator = generator
        frames.register_context(generator.ctx)   # the module-level frame helpers (colour fix) share this context

        # noise schedule (:46-53)
        sigmas = noise_schedule(schedule="logsnr_cosine_interp", n=1000, zero_terminal_snr=True, scale_min=2.0, scale_max=4.0)
        self.diffusion = GaussianDiffusion(sigmas=sigmas)

        # temporal VAE (:57-63)
        vae_cfg = _opt(opt, "vae_config", VaeConfig())
        vsd = _opt(opt, "vae_state_dict")
        if vsd is None:
            vae_path = _opt(opt, "vae_path")
            if not vae_path:
                raise ValueError("opt.vae_path (diffusers AutoencoderKLTemporalDecoder state dict) or opt.vae_state_dict is required: "
                                 "the HF hub is not reachable from here")
            if vae_path.endswith(".safetensors"):
                from safetensors.torch import load_file
                vsd = load_file(vae_path)
            else:
                vsd = torch.load(vae_path, map_location="cpu")
        self.vae = AutoencoderKLTemporalDecoder(vae_cfg, dtype=dtype, device=dev_index, library=library).load_state_dict(vsd)

        self.negative_prompt = NEGATIVE_PROMPT     # utils/config.py:160-169
        self.positive_prompt = POSITIVE_PROMPT
        neg = _opt(opt, "negative_y")
        if neg is None:
            neg = self._encode_text(self.negative_prompt)
        self.negative_y = neg.to(self._tensor_device)
        # optional CPU torch.Generator: when given, EVERY random draw of test() (VAE posterior noise, diffuse noise,
        # solver noise) comes from it in the reference's consumption order and is copied to the device.  This makes the
        # noise identical across devices/ranks (needed when one video's chunks are sharded over GPUs, and for parity tests).
        self.rng = _opt(opt, "rng")
        self.chunk_executor = _opt(opt, "chunk_executor")   # star_amd.parallel.ChunkSharder for multi-GPU chunk sharding
        self.frame_sharder = _opt(opt, "frame_sharder")

    @property
    def _tensor_device(self):
        return self.generator.ctx.torch_device

    def _encode_text(self, y):
        if torch.is_tensor(y):
            return y.detach()
        if self.text_encoder is None:
            raise RuntimeError("a prompt string needs a text encoder: " + (self._text_encoder_error or
                               "pass precomputed [1, 77, 1024] embeddings or opt.text_encoder"))
        return self.text_encoder(y).detach()

    def test(self, input: Dict[str, Any], total_noise_levels=1000, steps=50, solver_mode="fast", guide_scale=7.5, max_chunk_len=32,
             return_device=False):
        video_data = input["video_data"]
        y = input["y"]
        (target_h, target_w) = input["target_res"]
        dev = self._tensor_device

        # F.interpolate(..., mode='bilinear') + F.pad(..., 'constant', 1) of the reference (:81-87) as one HIP pass
        frames_num, h, w = video_data.shape[0], int(target_h), int(target_w)
        padding = pad_to_fit(h, w)
        video_data = self.generator.ctx.resize_pad(video_data.to(dev, torch.float32), (h, w), padding, 1.0)
        video_data = video_data.unsqueeze(0)
        bs = 1

        video_data_feature = self.vae_encode(video_data)
        self.vae.ctx.trim()                      # reference: torch.cuda.empty_cache() (:94) -- the encoder's arena is not needed while sampling
        y = self._encode_text(y).to(dev)

        t = torch.LongTensor([total_noise_levels - 1]).to(dev)
        noise = None
        if self.rng is not None:
            noise = torch.randn(video_data_feature.shape, generator=self.rng, dtype=torch.float32).to(dev)
        noised_lr = self.diffusion.diffuse(video_data_feature, t, noise=noise)
        model_kwargs = [{"y": y}, {"y": self.negative_y}, {"hint": video_data_feature}]
        chunk_inds = make_chunks(frames_num, interp_f_num=0, max_chunk_len=max_chunk_len) if frames_num > max_chunk_len else None
        gen_vid = self.diffusion.sample_sr(
            noise=noised_lr, model=self.generator, model_kwargs=model_kwargs, guide_scale=guide_scale, guide_rescale=0.2,
            solver="dpmpp_2m_sde", solver_mode=solver_mode, return_intermediate=None, steps=steps,
            t_max=total_noise_levels - 1, t_min=0, discretization="trailing", chunk_inds=chunk_inds,
            chunk_executor=self.chunk_executor if chunk_inds is not None else None,
            **({"noise_sampler_cls": _cpu_noise_sampler(self.rng)} if self.rng is not None else {}))

        self.generator.ctx.trim()                # reference: torch.cuda.empty_cache() (:124) -- the denoiser's arena is not needed while decoding
        vid_tensor_gen = self.vae_decode_chunk(gen_vid, chunk_size=3)
        w1, w2, h1, h2 = padding
        vid_tensor_gen = vid_tensor_gen[:, :, h1:h + h1, w1:w + w1]
        gen_video = vid_tensor_gen.reshape(bs, frames_num, *vid_tensor_gen.shape[1:]).permute(0, 2, 1, 3, 4)
        gen_video = gen_video.type(torch.float32)
        return gen_video.contiguous() if return_device else gen_video.cpu()   # reference: .cpu() (:139)

    def temporal_vae_decode(self, z, num_f):
        return self.vae.decode(z / self.vae.config.scaling_factor, num_frames=num_f).sample

    def vae_decode_chunk(self, z, chunk_size=3):
        """groups of <= chunk_size frames, boundaries at multiples of chunk_size from frame 0 (:144-151)."""
        z = z.permute(0, 2, 1, 3, 4).reshape(-1, z.shape[1], z.shape[3], z.shape[4])
        groups = [(i, min(i + chunk_size, z.shape[0])) for i in range(0, z.shape[0], chunk_size)]
        if self.frame_sharder is not None:
            return self.frame_sharder.map_groups(groups, lambda a, b: self.temporal_vae_decode(z[a:b], b - a))
        return torch.cat([self.temporal_vae_decode(z[a:b], b - a) for a, b in groups])

    def vae_encode(self, t, chunk_size=1):
        """one frame per encoder pass; the posterior noise is drawn for ALL frames in frame order (:153-161).  With a
        frame_sharder (and the shared CPU generator that makes every rank draw identical noise) each rank encodes only its
        own frames and the latents are all-gathered."""
        num_f = t.shape[1]
        t = t.reshape(-1, *t.shape[2:])
        f = self.vae.cfg.downsample
        lat_shape = (chunk_size, self.vae.cfg.latent_channels, t.shape[-2] // f, t.shape[-1] // f)
        groups = [(i, min(i + chunk_size, t.shape[0])) for i in range(0, t.shape[0], chunk_size)]
        eps = None
        if self.rng is not None:
            eps = [torch.randn((b - a,) + lat_shape[1:], generator=self.rng, dtype=torch.float32) for a, b in groups]

        def enc(a, b):
            dist_ = self.vae.encode(t[a:b]).latent_dist
            if eps is None:
                return dist_.sample()
            return dist_.mean + dist_.std * eps[a // chunk_size].to(dist_.mean.device)

        if self.frame_sharder is not None and eps is not None:
            z = self.frame_sharder.map_groups(groups, enc)
        else:
            z = torch.cat([enc(a, b) for a, b in groups], dim=0)
        z = z.reshape(1, num_f, *z.shape[1:]).permute(0, 2, 1, 3, 4)
        return z * self.vae.config.scaling_factor
