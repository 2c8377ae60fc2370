// optypes.h -- operator mode enums shared by kernels, launchers and the C ABI
#pragma once
namespace star {

enum GemmAMode : int {
  A_PLAIN = 0,     // A[m][k] = A + m*lda + k
  A_CONV3X3 = 1,   // 3x3 conv over NHWC, stride s, pad (pad_t, pad_l); K = 9*Cin
  A_CONV3X3_UP = 2,// nearest 2x upsample + drop first/last row (unet_v2v.py:563-564) fused in front of a 3x3 pad-1 conv
  A_TCONV3 = 3,    // Conv3d (3,1,1) pad (1,0,0) over frames: K = 3*Cin, source row m + (tap-1)*HW
};

enum GemmEpi : int {
  EPI_BIAS = 1,    // + bias[n] (fp32)
  EPI_RES = 2,     // + res[m][n] (T, row stride ldr)
  EPI_GEGLU = 4,   // weight rows interleaved in 32-row (value, gate) blocks: out[m][n/2] = (v+bv) * gelu_erf(g+bg)
  EPI_OUT_F32 = 8, // store fp32 instead of T
  EPI_ROWAFF = 32, // LayerNorm folded into the GEMM: out = a_m * acc + b_m * colsum[n] + bias[n], (a_m, b_m) = rowab[m] (gemm.h)
  EPI_GELU_TANH = 16, // out = gelu_tanh(acc + bias): the DiT MLP's activation (sat's gelu_impl, cogvideox-based/transformer.py:202-313)
};


enum LnMode : int {
  LN_PLAIN = 0,
  LN_GATE_LINEAR = 1,  // x' = sigmoid(w0*max_c(x) + w1*mean_c(x)) * x   (temporal LIEM), then LN(x')
  LN_GATE_MAP = 2,     // x' = sigmoid(conv7x7([max_c, mean_c]))(token) * x (spatial LIEM), then LN(x'); needs maps
  LN_STATS_ONLY = 3,   // write maps[token] = (max_c, mean_c); no LN output
};

}  // namespace star
