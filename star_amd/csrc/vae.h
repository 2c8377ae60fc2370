// vae.h -- SVD temporal VAE (diffusers AutoencoderKLTemporalDecoder) graph executor.
#pragma once
#include "graph.h"

namespace star {

struct VaeCfg {
  int in_ch = 3, out_ch = 3, latent = 4;
  int n_blocks = 4;
  int block_out[8] = {128, 256, 512, 512, 0, 0, 0, 0};
  int layers_per_block = 2;
};

struct Res2DW { int cin = 0, cout = 0; NormW n1, n2; LinW c1, c2, sc; bool has_sc = false; };
struct AttnVW { int C = 0; NormW gn; LinW q, k, v_as_a, out; DevW bv; };   // single head, head dim = C
struct STResW { Res2DW sp; NormW tn1, tn2; LinW tc1, tc2; };                 // tc2 pre-scaled by sigmoid(mix_factor)

struct VaeModel {
  VaeCfg cfg;
  // encoder
  LinW e_conv_in;                       // im2col-64
  std::vector<std::vector<Res2DW>> e_res;
  std::vector<LinW> e_down;             // size n_blocks-1
  Res2DW e_mid0, e_mid1; AttnVW e_attn;
  NormW e_norm_out; LinW e_conv_out;    // conv_out fused with quant_conv (512 -> 2*latent)
  // decoder
  LinW d_conv_in;                       // im2col-64
  STResW d_mid0, d_mid1; AttnVW d_attn;
  std::vector<std::vector<STResW>> d_res;
  std::vector<LinW> d_up;               // size n_blocks-1
  NormW d_norm_out; LinW d_conv_out;
  DevW d_time_w, d_time_b;              // fp32 [out][out][3], [out]
  std::vector<void*> owned;
  ~VaeModel();
};

int vae_build(Ctx* ctx, const VaeCfg& cfg);
// x: fp32 device [n, 3, H, W] -> moments rows fp32 [n*(H/f)*(W/f), 2*latent] (mean | logvar), f = 2^(n_blocks-1)
int vae_encode(Ctx* ctx, const float* x, float* moments, int n, int H, int W);
// z: fp32 device [n, latent, h, w] (already divided by the scaling factor), one temporal group of n frames
// -> out fp32 [n, 3, h*f, w*f]
int vae_decode(Ctx* ctx, const float* z, float* out, int n, int h, int w);

}  // namespace star
