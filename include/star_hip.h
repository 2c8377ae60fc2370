/* star_hip.h -- C ABI of libstar_hip.so: the MI355X (gfx950) implementation of
 * STAR's per-chunk denoising hot path.
 *
 * The reference (NJU-PCALab/STAR) has no FFI: its seam is the Python object
 * VideoToVideo_sr and the callables it wires together.
 * Each entry point below names the reference interface it replaces; the
 * reference-side binding is the ctypes stub shown in INTEGRATION.md
 * (star_amd/lib.py is that stub, shipped).
 *
 * Conventions: plain pointers and sizes only; all device pointers are
 * caller-owned (e.g. PyTorch-ROCm tensors) unless stated; every call enqueues
 * on the context's stream and returns 0 on success, non-zero on error with a
 * message available from star_last_error().  One host thread per context.
 * Activation layout is channels-last: [frames, H, W, C] == [tokens, C].
 */
#ifndef STAR_HIP_H_
#define STAR_HIP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct star_ctx star_ctx;

enum { STAR_F16 = 0, STAR_BF16 = 1, STAR_F32 = 2 };

/* A-operand gather modes of star_gemm (see star_amd/csrc/gemm.h).  Weight matrices W are [N][K], K contiguous:
 *   STAR_A_PLAIN      K = the Linear's input width;
 *   STAR_A_CONV3X3 / STAR_A_CONV3X3_UP   K = 9 * Cin (Cin % 64 == 0) with K index ((c / 64) * 9 + tap) * 64 + c % 64, tap = ky * 3 + kx: the nine
 *                     taps of a 64-channel block are adjacent (the kernel walks them back to back so that their shifted re-reads of the same
 *                     input lines hit the L2); star_amd.lib.pack_conv3x3_weight converts nn.Conv2d's [Cout, Cin, 3, 3];
 *   STAR_A_TCONV3     K = 3 * Cin with K index tap * Cin + c (nn.Conv3d (3,1,1)'s [Cout, Cin, 3, 1, 1], tap-major). */
enum { STAR_A_PLAIN = 0, STAR_A_CONV3X3 = 1, STAR_A_CONV3X3_UP = 2, STAR_A_TCONV3 = 3 };
/* epilogue flags of star_gemm */
enum { STAR_EPI_BIAS = 1, STAR_EPI_RES = 2, STAR_EPI_GEGLU = 4, STAR_EPI_OUT_F32 = 8,
       STAR_EPI_GELU_TANH = 16,  /* out = gelu_tanh(acc + bias): the DiT MLP activation (plain-A layers, no residual) */
       STAR_EPI_ROWAFF = 32      /* a LayerNorm folded into this projection: out = a_m * acc + b_m * colsum[n] + bias[n] with the
                                    star_gemm_desc.rowab / .colsum operands (plain-A layers, 16-bit output, no residual; needs
                                    STAR_EPI_BIAS).  star_layer_norm_rowab produces rowab. */ };

/* ---- context ------------------------------------------------------------ */
/* replaces: VideoToVideo_sr.__init__ device selection (video_to_video_model.py:21-34,42) */
/* returns 0, or: 1 null out, 2 bad dtype, 3 no such device, 4 hipSetDevice failed, 5 out of memory, 6 the device is not a gfx950
 * with the 160 KB LDS opt-in (the library carries gfx950 code objects only; there is no fallback path).
 * Every compute entry point below returns non-zero and sets star_last_error() when one of its kernels could not be launched. */
int star_ctx_create(int device_id, int dtype, star_ctx** out);
void star_ctx_destroy(star_ctx* ctx);
const char* star_last_error(star_ctx* ctx);
int star_set_stream(star_ctx* ctx, void* hip_stream);   /* use the caller's hipStream_t */
int star_sync(star_ctx* ctx);
int star_is_hostemu(void);                               /* 1 only in the test-tooling emulator build */
int star_has_bench_variants(void);                       /* 1 only in builds with -DSTAR_BENCH_VARIANTS (tools/bench, emulator): losing A/B
                                                            kernels and timing ablations; the product library answers 0 and rejects their ids */
/* replaces: torch.cuda.empty_cache() at the phase boundaries of test() (video_to_video_model.py:94,106,127): synchronises the
 * context's stream and returns its cached free blocks to the driver (the UNet and the VAE context each keep their own arena) */
int star_pool_trim(star_ctx* ctx);
size_t star_pool_bytes(star_ctx* ctx);
size_t star_pool_peak_bytes(star_ctx* ctx);
/* diagnostic: how many star_gemm / internal GEMM launches of this context were split into full rounds of big tiles + a remainder of
 * 128 x 128 tiles (see star_gemm_desc.force_tile) */
int64_t star_gemm_split_count(star_ctx* ctx);
/* diagnostic: how many GroupNorms of this context were finalized from their producer's partial statistics (star_gemm_gn / the
 * forward's conv, temporal-conv and proj_out epilogues) instead of running a statistics pass over their input */
int64_t star_gn_fused_count(star_ctx* ctx);
/* ... and how many LayerNorm row-coefficient / LIEM-map passes were served from their producer's row statistics (star_gemm_rowstats) */
int64_t star_ln_fused_count(star_ctx* ctx);

/* ---- kernel-level entry points (unit parity; each is one HIP kernel family) */
typedef struct star_gemm_desc {
  const void* A; const void* W; void* C; const float* bias; const void* res;
  int32_t M, N, K, lda, ldc, ldr;
  int32_t mode;                      /* STAR_A_* */
  int32_t H, Wd, Cin, Ho, Wo, stride, pad_t, pad_l;   /* conv geometry, NHWC */
  int32_t HW, F;                     /* temporal-conv geometry */
  int32_t up_crop;                   /* STAR_A_CONV3X3_UP: rows cropped top+bottom after the 2x upsample (1 UNet, 0 VAE) */
  int32_t epi;                       /* STAR_EPI_* */
  int32_t force_tile;                /* 0 = auto; 1 256x256, 2 256x320, 3 128x128, 4 256x128, 9 2 x (128x256), 30 the A-stationary K = 320 kernel
                                        (gemm_as.h: plain A, STAR_EPI_ROWAFF layers); 1000 + n = automatic choice with the tile rounds balanced
                                        for n compute units instead of the device's (the launcher gives a poorly filled last round of big
                                        tiles to a second launch of 128 x 128 tiles); 18 the persistent one-wave-per-SIMD tile (gemm_p.h:
                                        plain A, 16-bit output, bias / residual / STAR_EPI_ROWAFF), 2000 + n the same on n resident workgroups;
                                        17 / 19 the scheduled one-wave-per-SIMD tiles 256x256 / 256x320 (gemm.h SCHED: plain, 3x3 and temporal
                                        convs, bias / residual 16-bit epilogues); anything else: bench build only */
  const float* rowab;                /* STAR_EPI_ROWAFF (a LayerNorm folded into this projection, unet_v2v.py:448-450 + the Linear behind it): */
  const float* colsum;               /*   out = a_m * acc + b_m * colsum[n] + bias[n], (a_m, b_m) = rowab[m] fp32 pairs, colsum fp32 [N]; else null */
} star_gemm_desc;
/* replaces: nn.Linear / nn.Conv2d / nn.Conv3d(3,1,1) / nn.Conv1d(k=1) call sites
 * (unet_v2v.py:151-155,274,294,500,526,553,612,639,648,717,1005,1025,1209-1220) */
int star_gemm(star_ctx* ctx, const star_gemm_desc* d);
/* star_gemm whose epilogue ALSO writes the GroupNorm partial statistics of its output -- what the ResBlock / transformer layers in front
 * of an nn.GroupNorm do in the forward (unet_v2v.py:609-640,1209-1220: conv / temporal conv / proj_out -> GroupNorm), so that the norm needs
 * no statistics pass of its own.  gn_partial: fp32 [ceil(M / 32)][N / 2][2] = (sum, sum of squares) of the STORED 16-bit outputs over
 * the 32 rows of a slot, per pair of adjacent channels.  The flavour exists for the 256 x 320, 128 x 128 and scheduled 256 x 256 tiles on
 * plain / 3x3 / temporal-conv layers with the bias (+ residual) 16-bit epilogue; *wrote = 1 if the partials were written, 0 if the launcher's
 * tile has no such flavour (the output is computed either way). */
int star_gemm_gn(star_ctx* ctx, const star_gemm_desc* d, float* gn_partial, int32_t* wrote);
/* star_gemm whose epilogue ALSO writes per-ROW statistics of its output -- what the projections in front of an nn.LayerNorm do in the
 * forward (unet_v2v.py:448-450,466-490: proj_in / to_out + residual -> norm1 / norm2 / norm3 and the LIEM gates), so that the norm's row
 * coefficients need no pass over the rows.  ln_partial: fp32 [M][parts][4] = (sum, sum of squares, max, 0) of the STORED 16-bit outputs
 * of row m over the columns of one part; the buffer holds parts_cap parts per row, *parts receives the number written (2 per 320-column
 * tile of the 256 x 320 tile, 2 per 128-column tile of the 128 x 128 tile).  Plain-A layers with the bias (+ residual) 16-bit epilogue;
 * *wrote = 0 (and nothing is written) when the launcher's tile has no such flavour or parts_cap is too small. */
int star_gemm_rowstats(star_ctx* ctx, const star_gemm_desc* d, float* ln_partial, int32_t parts_cap, int32_t* parts, int32_t* wrote);


/* replaces: xformers.ops.memory_efficient_attention(q,k,v) for spatial self- and text cross-attention
 * (unet_v2v.py:158-195).  Q/K/V/O are token matrices; head h uses columns [64h, 64h+64) of each row. */
typedef struct star_attn_desc {
  const void* Q; const void* K; const void* V; void* O;
  int32_t ldq, ldk, ldv, ldo;
  int64_t bsq, bsk, bsv, bso;        /* batch (frame) strides in elements; 0 = shared by all batches */
  int32_t Nq, Nk, heads, batch;
  float scale;
  int32_t variant;                   /* 9 = the product kernel (0 is accepted as "default" = 9); any other id is rejected unless
                                        the library was built with -DSTAR_BENCH_VARIANTS (some ablation ids compute wrong results) */
  int32_t causal;                    /* 1: key j is visible to query i only for j <= i (Nq == Nk): open_clip's text attn_mask,
                                        embedder.py:59 */
  int32_t reserved;                  /* 0 */
} star_attn_desc;
int star_attn_fwd(star_ctx* ctx, const star_attn_desc* d);

/* replaces: the same call on '(b h w) f c' tensors = attention over the frame axis (unet_v2v.py:479-489);
 * rows are tokens f*HW + pixel, no transposes are materialised. */
typedef struct star_tattn_desc {
  const void* Q; const void* K; const void* V; void* O;
  int32_t ldq, ldk, ldv, ldo;
  int32_t F, HW, heads;
  float scale;
} star_tattn_desc;
int star_temporal_attn_fwd(star_ctx* ctx, const star_tattn_desc* d);

/* replaces: to_q / to_k / to_v (Linear, no bias) behind norm1 / norm2 AND the attention over the frame axis of the temporal branch in one
 * kernel (unet_v2v.py:479-489, :151-195; level-0 width: C = 320, 5 heads, F <= 32 frames): x rows [F*HW][lda] -> O rows [F*HW][ldo].
 * W: [960][320] = the LayerNorm-folded q | k | v weights (gamma o W) with their 64-row tiles ordered (q_h, k_h, v_h) per head;
 * bias / colsum fp32 [960] in the same order; rowab from star_layer_norm_rowab.  Bit-identical to star_gemm (STAR_EPI_ROWAFF) followed
 * by star_temporal_attn_fwd; q | k | v never reach HBM. */
typedef struct star_tq_desc {
  const void* A; const void* W; void* O; const float* bias; const float* colsum; const float* rowab;
  int32_t lda, ldo, HW, F, C, heads;
  float scale;
} star_tq_desc;
int star_temporal_qkv_attn(star_ctx* ctx, const star_tq_desc* d);

/* replaces: nn.GroupNorm(32, C) [+ nn.SiLU] on 4-D (per-frame stats) and 5-D (whole-chunk stats) tensors
 * (unet_v2v.py:268,610,635,1002,1210-1219); rows_per_stat = H*W or F*H*W */
int star_group_norm(star_ctx* ctx, const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma,
                    const float* beta, int32_t rows, int32_t C, int32_t rows_per_stat, float eps, int32_t silu);
/* the same GroupNorm on a tensor whose producer wrote its partial statistics (star_gemm_gn): finalize from the partials (the rows of the
 * 32-row slots that a statistics boundary cuts are re-read from x), then the apply pass -- two launches, x is read once.  C % 64 == 0. */
int star_group_norm_from_partials(star_ctx* ctx, const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma,
                                  const float* beta, int32_t rows, int32_t C, int32_t rows_per_stat, float eps, int32_t silu,
                                  const float* gn_partial);
/* replaces: nn.LayerNorm (unet_v2v.py:448-450) with the LIEM gates SpatialAttention (:380-394) /
 * TemporalLocalAttention (:396-411) fused in front.  mode: 0 plain, 1 linear gate, 2 7x7-map gate, 3 maps only */
int star_layer_norm(star_ctx* ctx, const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma,
                    const float* beta, int32_t rows, int32_t C, float eps, int32_t mode, const float* gate_w,
                    float* maps, int32_t H, int32_t W);
/* the statistics half of a LayerNorm that is folded into the projection behind it (STAR_EPI_ROWAFF): no normalised tensor is
 * written; rowab[row] = (a, b) fp32 pairs with LN(gate * x) = (a * x + b) * gamma + beta (gate = the LIEM gate of `mode`, 1 for
 * mode 0), which star_gemm applies as out = a_m * (x W'^T) + b_m * colsum[n] + bias'[n] with W' = gamma o W,
 * colsum[n] = sum_k W'[n][k], bias'[n] = sum_k beta[k] W[n][k] + bias[n] (prepared once per layer by the caller). */
int star_layer_norm_rowab(star_ctx* ctx, const void* x, int32_t ldx, float* rowab, int32_t rows, int32_t C, float eps, int32_t mode,
                          const float* gate_w, float* maps, int32_t H, int32_t W);
/* star_layer_norm_rowab (or, with mode 3, the LIEM maps of star_layer_norm) from the row statistics the producer of x wrote
 * (star_gemm_rowstats): 16 x parts bytes are read per row instead of the row.  mode 3: maps[row] = (max, mean), rowab unused. */
int star_layer_norm_rowab_from_partials(star_ctx* ctx, const float* ln_partial, int32_t parts, float* rowab, int32_t rows, int32_t C,
                                        float eps, int32_t mode, const float* gate_w, float* maps, int32_t H, int32_t W);
/* replaces: torch.cat([x, skip + control], dim=1) (unet_v2v.py:1792) and residual adds */
int star_concat_add(star_ctx* ctx, const void* a, const void* b, const void* c, void* out, int32_t rows, int32_t C1, int32_t C2);
int star_add(star_ctx* ctx, const void* a, const void* b, void* out, int64_t n);
/* replaces: the (b c f h w) <-> (b f) c h w rearranges at the UNet boundary (unet_v2v.py:1772,1808) */
int star_stem_im2col(star_ctx* ctx, const float* latent, void* out, int32_t Cl, int32_t F, int32_t H, int32_t W);
int star_rows_to_latent(star_ctx* ctx, const float* rows, float* out, int32_t Cl, int32_t ld, int64_t ntok);
/* replaces: time_embed / emb_layers Linear on the [1, C] embedding (unet_v2v.py:1340-1342, 626-633) */
int star_gemv(star_ctx* ctx, const float* x, const void* W, const float* b, float* y, int32_t N, int32_t K, int32_t silu_in, int32_t silu_out);
int star_cast(star_ctx* ctx, const float* x, void* y, int64_t n);

/* ---- weights + model (B4: torch.load(...)/load_state_dict, video_to_video_model.py:36-43) ------------- */
typedef struct star_unet_config {
  int32_t in_dim, dim, context_dim, out_dim;
  int32_t n_levels; int32_t dim_mult[8];
  int32_t num_heads, head_dim, num_res_blocks, attn_levels;
} star_unet_config;
/* stage one tensor of the reference state dict (host pointer, reference key name, fp32/fp16/bf16) */
int star_load_tensor(star_ctx* ctx, const char* name, const void* host_ptr, const int64_t* shape, int32_t ndim, int32_t dtype);
/* repack the staged tensors (NHWC conv weights, fused QKV, GEGLU interleave) and upload; frees the staging copies */
int star_unet_build(star_ctx* ctx, const star_unet_config* cfg);
/* replaces: ControlledV2VUNet.forward(x, t, y, hint=...) -> v-prediction (B2; unet_v2v.py:1717-1809,
 * called from GaussianDiffusion.denoise diffusion_sdedit.py:81,88).  xt, hint, out: fp32 device [1, 4, f, h, w];
 * y: fp32 device [77, context_dim]; t: the integer timestep. */
int star_unet_forward(star_ctx* ctx, const float* xt, int64_t t, const float* y, const float* hint, float* out,
                      int32_t f, int32_t h, int32_t w);
/* replaces: the TWO sequential denoiser calls of classifier-free guidance in GaussianDiffusion.denoise
 * (diffusion_sdedit.py:81 cond, :88 uncond): same xt / t / hint, two text contexts.  Bit-identical to two
 * star_unet_forward calls, but everything that does not depend on the text context (up to and including the
 * self-attention of each net's first spatial transformer) is computed once. */
int star_unet_forward_cfg(star_ctx* ctx, const float* xt, int64_t t, const float* y_cond, const float* y_uncond,
                          const float* hint, float* out_cond, float* out_uncond, int32_t f, int32_t h, int32_t w);
/* Replay the UNet forward from a captured hipGraph (off by default).  A forward is a fixed sequence of a few thousand kernel launches (about 3400 per CFG pair at cfg2 size) that
 * depends on (guidance branches, f, h, w) only; with enable != 0 the first forward of a shape runs as usual (it sizes the activation
 * pool), the second is captured on a stream the context owns, later ones copy xt / y / hint into the graph's staging buffers, refresh
 * the timestep row and issue one hipGraphLaunch, ordered against the context's stream with events.  Results are bit-identical to the
 * eager path.  A trimmed pool (star_pool_trim, or an allocation that had to drop the cache) invalidates the graph; it is re-captured. */
int star_unet_graph(star_ctx* ctx, int32_t enable);
/* replaces: VideoControlNet.forward (unet_v2v.py:2134-2206) on its own -- the 13 residuals ControlledV2VUNet.forward adds to the skip
 * connections and the middle block (:1746-1748, 1790-1800): residuals[i] receives channels-last rows [f * H_i * W_i, C_i] in the
 * context's storage dtype, i = 0..11 the zero-conv'd encoder outputs (level sizes (H, W) -> (H/2 + 1, W/2) per Downsample), 12 the
 * middle block's.  star_unet_forward computes the same tensors internally; this entry exists so that parity tests can look at them. */
int star_controlnet_forward(star_ctx* ctx, const float* xt, int64_t t, const float* y, const float* hint, void* const* residuals,
                            int32_t n_residuals, int32_t f, int32_t h, int32_t w);
/* one reference module (ResBlock / SpatialTransformer / TemporalTransformer / Downsample / Upsample) built from
 * staged tensors `prefix.*`; kind: 0 res, 1 spatial, 2 temporal, 3 down, 4 up.  x/out: channels-last rows (ctx dtype) */
int star_module_run(star_ctx* ctx, int32_t kind, const char* prefix, int32_t cin, int32_t cout, int32_t heads,
                    int32_t embed_dim, int32_t context_dim, const void* x, const float* emb, const float* context,
                    void* out, int32_t f, int32_t h, int32_t w);
int star_clear_staged(star_ctx* ctx);

/* ---- SVD temporal VAE (B3: vae.encode / vae.decode, video_to_video_model.py:141-161; diffusers un-vendored) ---- */
typedef struct star_vae_config {
  int32_t in_ch, out_ch, latent, n_blocks; int32_t block_out[8]; int32_t layers_per_block;
} star_vae_config;
/* build from staged tensors named as diffusers' AutoencoderKLTemporalDecoder state dict */
int star_vae_build(star_ctx* ctx, const star_vae_config* cfg);
/* replaces: vae.encode(x).latent_dist -> moments rows fp32 [n*h*w, 2*latent] = (mean | logvar), one frame per pass */
int star_vae_encode(star_ctx* ctx, const float* x, float* moments, int32_t n, int32_t H, int32_t W);
/* replaces: vae.decode(z, num_frames=n).sample for ONE temporal group of n frames; z fp32 [n, latent, h, w] */
int star_vae_decode(star_ctx* ctx, const float* z, float* out, int32_t n, int32_t h, int32_t w);
int star_softmax_rows(star_ctx* ctx, const float* s, int32_t lds, void* p, int32_t ldp, int32_t rows, int32_t n, float scale);

/* ---- CogVideoX-5B DiT block (STAR's CogVideoX variant, cogvideox-based/sat) ------------------ */
typedef struct star_dit_config {
  int32_t hidden, heads, time_embed_dim, n_layers;   /* 3072, 48, 512, 42 at full size; head dim is 64 */
  float ln_eps;                                       /* sat layernorm_epsilon (1e-5) */
} star_dit_config;
/* build from staged tensors named as the SAT checkpoint under `model.diffusion_model.`:
 * transformer.layers.{i}.{input_layernorm, post_attention_layernorm, attention.query_key_value, attention.dense,
 * mlp.dense_h_to_4h, mlp.dense_4h_to_h}.{weight,bias}, transformer.layers.{i}.{spa_local,temp_local}.conv1.weight (the LIEM gates
 * of cogvideox-based/transformer.py:316-348,485-486) and mixins.adaln_layer.{adaLN_modulations.{i}.1, query_layernorm_list.{i},
 * key_layernorm_list.{i}}.{weight,bias} */
int star_dit_build(star_ctx* ctx, const star_dit_config* cfg);
/* replaces: AdaLNMixin.layer_forward (cogvideox-based/sat/dit_video_concat.py:482-563) for layer `layer`, with the 3-D rotary
 * embedding (:254-346) and the QK LayerNorm (:571-598) inside its attention.  hidden_in / hidden_out: device rows
 * [text_len + T*H*W][hidden] in the context's storage dtype (text tokens first, video tokens in (t h w) order, batch 1);
 * emb: fp32 device [time_embed_dim], the timestep embedding kwargs["emb"]. */
int star_dit_block_forward(star_ctx* ctx, int32_t layer, const void* hidden_in, const float* emb, void* hidden_out,
                           int32_t text_len, int32_t T, int32_t H, int32_t W);

/* ---- full-resolution frame kernels either side of the diffusion path ---------- */
/* replaces: F.interpolate(video_data, [target_h, target_w], mode='bilinear') + F.pad(video_data, padding, 'constant', 1)
 * in VideoToVideo_sr.test (video_to_video_model.py:81-87).  src: fp32 device planes x [h][w] (planes = F*3);
 * dst: fp32 device planes x [th + pad_t + pad_b][tw + pad_l + pad_r]. */
int star_resize_pad(star_ctx* ctx, const float* src, float* dst, int32_t planes, int32_t h, int32_t w, int32_t th, int32_t tw,
                    int32_t pad_l, int32_t pad_r, int32_t pad_t, int32_t pad_b, float pad_value);
/* replaces: calc_mean_std (color_fix.py:62-74) on v = x*scale + shift (clamped to [0,1] when clamp01): per contiguous
 * plane of n fp32 values -> stats[plane] = (mean, sqrt(unbiased var + eps)) fp32 pairs.  Reproducible (no atomics). */
int star_plane_stats(star_ctx* ctx, const float* x, float* stats, int32_t planes, int64_t n, float scale, float shift,
                     int32_t clamp01, float eps);
/* replaces: tensor2vid (inference_utils.py:16-23) followed by adain_color_fix (color_fix.py:15-29,76-89) as called from
 * inference_sr.py:47-48.  x: fp32 device [1, C, F, H, W] in ~[-1, 1] (the pipeline output); src: fp32 device
 * [F, C, h, w] low-resolution clip in [-1, 1]; out: fp32 device [F, H, W, C] in [0, 255]. */
int star_color_fix(star_ctx* ctx, const float* x, const float* src, float* out, int32_t F, int32_t C, int32_t H, int32_t W,
                   int32_t h, int32_t w);
/* the same with what save_video does to its input folded in (`.astype('uint8')`, inference_utils.py:92): out is uint8 device
 * [F, H, W, C], truncated -- the frames leave the GPU as bytes (a quarter of the PCIe traffic, no conversion pass on the host). */
int star_color_fix_u8(star_ctx* ctx, const float* x, const float* src, uint8_t* out, int32_t F, int32_t C, int32_t H, int32_t W,
                      int32_t h, int32_t w);
/* replaces: adain_color_fix(target, source) on its own (color_fix.py:15-29): target fp32 device [F, H, W, C] in [0, 255]
 * (a tensor2vid result), src as above -> out fp32 device [F, H, W, C] in [0, 255]. */
int star_adain_color_fix(star_ctx* ctx, const float* target, const float* src, float* out, int32_t F, int32_t C, int32_t H,
                         int32_t W, int32_t h, int32_t w);

/* ---- OpenCLIP ViT-H/14 text tower --------------------------------------------------------------
 * replaces: FrozenOpenCLIPEmbedder.text_transformer_forward + ln_final (video_to_video/modules/embedder.py:53-72) on the tensors
 * staged with star_load_tensor under open_clip's names (transformer.resblocks.{i}.{ln_1,ln_2}.{weight,bias},
 * .attn.in_proj_{weight,bias}, .attn.out_proj.*, .mlp.c_fc.*, .mlp.c_proj.*, ln_final.*).
 * star_text_forward: x = token_embedding(tokens) + positional_embedding as [batch * tokens, width] rows in the context's 16-bit
 * type; the first run_layers blocks run (layers - 1 for layer = 'penultimate'), causal self-attention, then ln_final; out like x. */
int star_text_build(star_ctx* ctx, int32_t width, int32_t heads, int32_t layers);
int star_text_forward(star_ctx* ctx, const void* x, int32_t batch, int32_t tokens, int32_t run_layers, void* out);

/* ---- live per-kernel-family timing (HIP events on the launch stream; used by bench.py's roofline leg) --------- */
enum { STAR_PK_ATTN_SELF = 0, STAR_PK_ATTN_CROSS, STAR_PK_TATTN, STAR_PK_GEMM, STAR_PK_CONV, STAR_PK_TCONV, STAR_PK_GN,
       STAR_PK_LN, STAR_PK_MISC, STAR_PK_COUNT };
typedef struct star_prof_entry { double ms; double flops; double bytes; double max_flops; double max_flops_ms; int64_t launches; } star_prof_entry;
int star_profile_begin(star_ctx* ctx);
/* the same, bracketing only the kernel families whose bit (1 << STAR_PK_*) is set in kind_mask: two event records per launch cost stream
 * time (measured 1.4 s of a 61 s clip with every launch bracketed), so bench.py times only the dominant kernel inside its timed region */
int star_profile_begin_kinds(star_ctx* ctx, uint32_t kind_mask);
/* synchronises, fills out[STAR_PK_COUNT] (max_flops_ms = mean duration of the family's largest launches), stops profiling */
int star_profile_end(star_ctx* ctx, star_prof_entry* out);

#ifdef __cplusplus
}
#endif
#endif /* STAR_HIP_H_ */
