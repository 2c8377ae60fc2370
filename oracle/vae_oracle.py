"""CPU oracle of the SVD temporal VAE (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED.  The arithmetic lives in diffusers==0.30.0 (reference requirements.txt:14; class
AutoencoderKLTemporalDecoder; call sites video_to_video_model.py:16,57-63,142,158), which is neither vendored in
/root/reference nor installed here, and no reference test pins it.  This file restates the published architecture
(models/autoencoders/autoencoder_kl_temporal_decoder.py, models/autoencoders/vae.py Encoder +
DiagonalGaussianDistribution, models/unets/unet_3d_blocks.py MidBlockTemporalDecoder / UpBlockTemporalDecoder,
models/resnet.py ResnetBlock2D / TemporalResnetBlock / SpatioTemporalResBlock / AlphaBlender / Downsample2D /
Upsample2D, models/attention_processor.py Attention) from memory as plain fp32 PyTorch over a diffusers-keyed state
dict.  The HIP path is checked against THIS restatement on synthetic weights; it must be re-pinned against real
diffusers outputs when that package is available.
"""
import torch
import torch.nn.functional as F


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet2d(sd, p, x):
    """ResnetBlock2D(temb_channels=None, eps=1e-6, silu, output_scale_factor=1)."""
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def attention(sd, p, x):
    """Attention(heads=1, dim_head=C, norm_num_groups=32, eps=1e-6, residual_connection=True, bias=True)."""
    n, c, h, w = x.shape
    t = _gn(sd, p + ".group_norm", x.reshape(n, c, h * w), 1e-6).transpose(1, 2)
    q = F.linear(t, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(t, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(t, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)).squeeze(1)
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(n, c, h, w) + x


def temporal_resnet(sd, p, x5):
    """TemporalResnetBlock(eps=1e-5): Conv3d (3,1,1) pad (1,0,0), GroupNorm over (C/32, F, H, W)."""
    h = F.conv3d(F.silu(_gn(sd, p + ".norm1", x5, 1e-5)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=(1, 0, 0))
    h = F.conv3d(F.silu(_gn(sd, p + ".norm2", h, 1e-5)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=(1, 0, 0))
    return x5 + h


def st_resblock(sd, p, x, num_frames):
    """SpatioTemporalResBlock with AlphaBlender(merge_strategy='learned', switch_spatial_to_temporal_mix=True) and
    image_only_indicator = zeros: alpha = 1 - sigmoid(mix_factor); out = alpha * x_spatial + (1 - alpha) * x_temporal."""
    xs = resnet2d(sd, p + ".spatial_res_block", x)
    n, c, h, w = xs.shape
    b = n // num_frames
    x5 = xs.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    xt = temporal_resnet(sd, p + ".temporal_res_block", x5)
    alpha = 1.0 - torch.sigmoid(sd[p + ".time_mixer.mix_factor"])
    out = alpha * x5 + (1.0 - alpha) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, h, w)


def encode_moments(sd, cfg, x):
    """Encoder + quant_conv -> moments [n, 2L, h, w] (mean | logvar)."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet2d(sd, f"encoder.down_blocks.{i}.resnets.{j}", h)
        if i != nb - 1:   # Downsample2D(padding=0): F.pad (0,1,0,1) then conv stride 2
            h = F.pad(h, (0, 1, 0, 1))
            h = F.conv2d(h, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = resnet2d(sd, "encoder.mid_block.resnets.0", h)
    h = attention(sd, "encoder.mid_block.attentions.0", h)
    h = resnet2d(sd, "encoder.mid_block.resnets.1", h)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def sample_posterior(moments, noise):
    """DiagonalGaussianDistribution.sample: mean + exp(0.5 * clamp(logvar, -30, 20)) * noise."""
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def decode(sd, cfg, z, num_frames):
    """TemporalDecoder.forward(z, image_only_indicator=zeros(b, num_frames), num_frames)."""
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = st_resblock(sd, "decoder.mid_block.resnets.0", h, num_frames)
    h = attention(sd, "decoder.mid_block.attentions.0", h)
    h = st_resblock(sd, "decoder.mid_block.resnets.1", h, num_frames)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = st_resblock(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, num_frames)
        if i != nb - 1:   # Upsample2D: nearest x2 + conv
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, 1e-6))
    h = F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
    n, c, hh, ww = h.shape
    b = n // num_frames
    h5 = h.reshape(b, num_frames, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = F.conv3d(h5, sd["decoder.time_conv_out.weight"], sd["decoder.time_conv_out.bias"], padding=(1, 0, 0))
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
