// unet.h -- host-side model of ControlledV2VUNet + VideoControlNet for the C++ graph executor.
#pragma once
#include <array>
#include <map>
#include "graph.h"
#include <string>
#include <vector>

namespace star {

struct UNetCfg {
  int in_dim = 4, dim = 320, context_dim = 1024, out_dim = 4;
  int n_levels = 4;
  int dim_mult[8] = {1, 2, 4, 4, 0, 0, 0, 0};
  int num_heads = 8, head_dim = 64, num_res_blocks = 2;
  int attn_levels = 3;   // levels 0..attn_levels-1 carry transformers (attn_scales 1, 1/2, 1/4)
  int embed_dim() const { return dim * 4; }
};

struct ResW {
  int cin = 0, cout = 0;
  NormW gn1, gn2;
  LinW conv1, conv2, emb, skip;
  bool has_skip = false;
  NormW tgn[4];
  LinW tconv[4];
};
struct TBlockW {
  NormW n1, n2, n3;
  LinW qkv1, out1;       // self attention (fused q|k|v rows)
  LinW q2, kv2, qkv2, out2;  // spatial: q2 + kv2 (context); temporal: qkv2 (self)
  LinW qkv1_tq, qkv2_tq;  // temporal blocks of width 320: the same projections with their 64-row tiles ordered (q_h, k_h, v_h) per head, for
                          // the fused projection + attention kernel (gemm_tq.h); empty elsewhere
  LinW ff1, ff2;         // ff1 GEGLU-interleaved
  LinW ffpo;             // [W_po | W_po W_ff2] over the operand [h2 | g]: ff2 (+ h2) and the transformer's proj_out in one GEMM (graph.h: compose_ff2_proj_out)
  DevW local1, local2;   // fp32 LIEM gate weights
};
struct STW { int C = 0, heads = 0; NormW norm; LinW proj_in, proj_out; TBlockW tb; };
struct TTW { int C = 0, inner = 0, heads = 0; NormW norm; LinW proj_in, proj_out; TBlockW tb; };
struct ConvW { int C = 0; LinW conv; };

enum ModKind : int { M_RES = 0, M_ST = 1, M_TT = 2, M_DOWN = 3, M_UP = 4 };
struct Mod { int kind; int idx; };

struct Net {
  LinW time0, time2;
  LinW stem;             // im2col K = 64
  std::vector<std::vector<Mod>> input_blocks;   // block 0 = [TT] (stem conv handled separately)
  std::vector<Mod> middle;
  std::vector<std::vector<Mod>> output_blocks;
  std::vector<ResW> res;
  std::vector<STW> st;
  std::vector<TTW> tt;
  std::vector<ConvW> convs;
  // control net only
  std::vector<LinW> zero_convs;
  LinW middle_out, hint;
  // main net only
  NormW out_norm;
  LinW out_conv;
};

// one captured forward (hipGraph) per (guidance branches, frames, latent size): the launch sequence of a forward depends on nothing
// else.  Inputs and outputs go through staging buffers the graph was captured with; the sinusoidal timestep row is refreshed in
// `tsin` before every launch.  Pool blocks the graph addresses stay cached between launches; `pool_gen` detects a trimmed pool.
struct UNetGraph {
  rt::GraphExec exec;
  bool captured = false;
  uint64_t pool_gen = 0;
  float* xt = nullptr; float* hint = nullptr; float* y[2] = {nullptr, nullptr}; float* out[2] = {nullptr, nullptr}; float* tsin = nullptr;
  size_t n_x = 0, n_hint = 0, n_y = 0, n_out = 0;
};
struct UNetModel {
  UNetCfg cfg;
  Net main, control;
  std::vector<void*> owned;   // device allocations
  std::map<std::array<int, 4>, UNetGraph> graphs;   // key: {nb, F, H, W}
  hipStream_t gstream = nullptr;                     // capture / replay stream
  ~UNetModel();
};

int unet_build(Ctx* ctx, const UNetCfg& cfg);
int unet_forward(Ctx* ctx, const float* xt, long long t, const float* y, const float* hint, float* out, int F, int H, int W);
// nb = 1 or 2 guidance branches (different text contexts) sharing all context-independent work
int unet_forward_n(Ctx* ctx, const float* xt, long long t, const float* const* ys, const float* hint, float* const* outs, int nb,
                   int F, int H, int W, void* const* control_tap = nullptr, int n_tap = 0);
// VideoControlNet.forward alone (unet_v2v.py:2134-2206): the zero-conv'd residuals of the encoder half + the middle block,
// as channels-last rows [F*H_l*W_l, C_l] in the storage dtype
int controlnet_forward(Ctx* ctx, const float* xt, long long t, const float* y, const float* hint, void* const* residuals, int n,
                       int F, int H, int W);
// run one module built on the fly from staged tensors `prefix.*` (unit parity against reference blocks)
int module_run(Ctx* ctx, int kind, const char* prefix, int cin, int cout, int heads, int embed_dim, int context_dim,
               const void* x, const float* emb, const float* context, void* out, int F, int H, int W);

}  // namespace star
