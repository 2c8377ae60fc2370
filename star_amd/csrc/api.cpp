// api.cpp -- the C ABI (include/star_hip.h) over the C++ launchers.
#include "../../include/star_hip.h"
#include "ops.h"
#include "unet.h"
#include "vae.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>

using namespace star;
namespace star {
int dit_build(Ctx* ctx, int D, int heads, int E, int n_layers, float ln_eps);
int dit_block_forward(Ctx* ctx, int layer, const void* x_in, const float* emb, void* x_out, int text_len, int T, int H, int W);
void dit_release(Ctx* ctx);
void text_release(Ctx* ctx);
int text_build(Ctx* ctx, int W, int heads, int n_layers);
int text_forward(Ctx* ctx, const void* x_in, int batch, int tokens, int run_layers, void* out);
}

struct star_ctx { Ctx c; };

// ---- launch-status plumbing: STAR_LAUNCH (prim.h) records the first refused launch / attribute here; every compute entry point
// of the ABI ends in finish(), which turns it into a non-zero return code + star_last_error() text.  Without this a forward whose
// kernels never ran (no gfx950 code object, LDS opt-in refused) would return 0 with uninitialised output buffers.
static thread_local std::string g_launch_err;
namespace star {
void rt_note_launch_error(const char* what) {
  if (!g_launch_err.empty()) return;
  g_launch_err = what;
  g_launch_err += ": ";
  g_launch_err += rt::last_error_string();   // also clears HIP's sticky "last error"
}
}
// every entry point that takes a context: a null context is error 1 (there is nowhere to put a message); otherwise follow the
// CONTEXT's device, not the caller's current one
#define STAR_ENTER(h) do { if (!(h)) return 1; rt::set_device((h)->c.device); } while (0)

static int finish(star_ctx* h, int rc) {
  if (rc) { g_launch_err.clear(); return rc; }
  if (!g_launch_err.empty()) {
    const std::string e = "kernel launch failed: " + g_launch_err;
    g_launch_err.clear();
    return h->c.fail(e);
  }
  return 0;
}

extern "C" {

int star_is_hostemu(void) {
#ifdef STAR_HOSTEMU
  return 1;
#else
  return 0;
#endif
}

int star_has_bench_variants(void) {
#ifdef STAR_BENCH_VARIANTS
  return 1;
#else
  return 0;
#endif
}

int star_ctx_create(int device_id, int dtype, star_ctx** out) {
  if (!out) return 1;
  *out = nullptr;
  if (dtype != DT_F16 && dtype != DT_BF16) return 2;
  if (rt::device_count() <= device_id) return 3;  // no CPU fallback: a gfx950 device is required
  if (rt::set_device(device_id)) return 4;
  if (!rt::device_is_gfx950(device_id)) return 6;   // the code objects are gfx950-only and the GEMM tiles need the 160 KB LDS opt-in
  star_ctx* h = new star_ctx();
  h->c.device = device_id;
  h->c.dtype = dtype;
  void* z = nullptr;
  if (rt::dev_malloc(&z, 256)) { delete h; return 5; }
  rt::memset_async(z, 0, 256, nullptr);
  rt::stream_sync(nullptr);
  h->c.zero_page = z;
  h->c.num_cus = rt::device_cu_count(device_id);
  *out = h;
  return 0;
}

void star_ctx_destroy(star_ctx* h) {
  if (!h) return;
  rt::stream_sync(h->c.stream);
  h->c.unet.reset();
  h->c.vae.reset();
  dit_release(&h->c);
  text_release(&h->c);
  h->c.pool.release();
  if (h->c.zero_page) rt::dev_free(h->c.zero_page);
  delete h;
}

const char* star_last_error(star_ctx* h) { return h ? h->c.err.c_str() : "null ctx"; }
int star_set_stream(star_ctx* h, void* s) {
  // the arena hands freed blocks out again immediately on the assumption that all work is ordered on ONE stream: when the
  // caller moves to another stream, everything enqueued on the old one is drained first (rare: once per torch stream context)
  if (h->c.stream != (hipStream_t)s) {
    rt::set_device(h->c.device);
    if (rt::stream_sync(h->c.stream)) return h->c.fail(std::string("set_stream: sync of the previous stream failed: ") + rt::last_error_string());
  }
  h->c.stream = (hipStream_t)s;
  return 0;
}
int star_sync(star_ctx* h) {
  STAR_ENTER(h);
  if (rt::stream_sync(h->c.stream)) return h->c.fail(std::string("sync failed: ") + rt::last_error_string());
  return 0;
}
int star_pool_trim(star_ctx* h) {   // return the cached (free) blocks to the driver: phase boundaries of test(), where the reference calls empty_cache()
  if (!h) return 1;
  rt::set_device(h->c.device);
  if (rt::stream_sync(h->c.stream)) return h->c.fail(std::string("pool_trim: sync failed: ") + rt::last_error_string());
  h->c.pool.trim();
  return 0;
}
size_t star_pool_bytes(star_ctx* h) { return h->c.pool.total(); }
size_t star_pool_peak_bytes(star_ctx* h) { return h->c.pool.peak(); }
int64_t star_gemm_split_count(star_ctx* h) { return h ? (int64_t)h->c.gemm_splits : 0; }
int64_t star_gn_fused_count(star_ctx* h) { return h ? (int64_t)h->c.gn_fused : 0; }
int64_t star_ln_fused_count(star_ctx* h) { return h ? (int64_t)h->c.ln_fused : 0; }

static int gemm_from_desc(star_ctx* h, const star_gemm_desc* d, float* gn_partial, bool* gn_done, float* ln_partial = nullptr, int ln_cap = 0,
                          int* ln_parts = nullptr, bool* ln_done = nullptr) {
  STAR_ENTER(h);
  GemmArgs a;
  a.gn_partial = gn_partial; a.gn_done = gn_done;
  a.ln_partial = ln_partial; a.ln_parts_cap = ln_cap; a.ln_parts = ln_parts; a.ln_done = ln_done;
  a.A = d->A; a.W = d->W; a.C = d->C; a.bias = d->bias; a.res = d->res;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr;
  a.mode = d->mode; a.H = d->H; a.Wd = d->Wd; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.HW = d->HW; a.F = d->F; a.up_crop = d->up_crop;
  a.epi = d->epi; a.force_tile = d->force_tile; a.rowab = d->rowab; a.colsum = d->colsum;
  if (a.force_tile >= 2000) { a.assume_cus = a.force_tile - 2000; a.force_tile = 18; }   // the persistent tile on that many resident workgroups (tests)
  else if (a.force_tile >= 1000) { a.assume_cus = a.force_tile - 1000; a.force_tile = 0; }   // automatic choice, rounds balanced for that many CUs (tests)
  return finish(h, op_gemm(&h->c, a));
}
int star_gemm(star_ctx* h, const star_gemm_desc* d) { return gemm_from_desc(h, d, nullptr, nullptr); }
int star_gemm_gn(star_ctx* h, const star_gemm_desc* d, float* gn_partial, int32_t* wrote) {
  bool done = false;
  if (wrote) *wrote = 0;
  if (!h) return 1;
  if (!gn_partial) return finish(h, h->c.fail("gemm_gn: null partial buffer"));
  const int rc = gemm_from_desc(h, d, gn_partial, &done);
  if (wrote) *wrote = done ? 1 : 0;
  return rc;
}


int star_attn_fwd(star_ctx* h, const star_attn_desc* d) {
  STAR_ENTER(h);
  AttnArgs a;
  a.Q = d->Q; a.K = d->K; a.V = d->V; a.O = d->O;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.bsq = d->bsq; a.bsk = d->bsk; a.bsv = d->bsv; a.bso = d->bso;
  a.Nq = d->Nq; a.Nk = d->Nk; a.heads = d->heads; a.batch = d->batch; a.scale = d->scale; a.variant = d->variant ? d->variant : 9; a.causal = d->causal;
  return finish(h, op_flash_attn(&h->c, a));
}
int star_temporal_attn_fwd(star_ctx* h, const star_tattn_desc* d) {
  STAR_ENTER(h);
  TAttnArgs a;
  a.Q = d->Q; a.K = d->K; a.V = d->V; a.O = d->O;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.F = d->F; a.HW = d->HW; a.heads = d->heads; a.scale = d->scale;
  return finish(h, op_temporal_attn(&h->c, a));
}
int star_temporal_qkv_attn(star_ctx* h, const star_tq_desc* d) {
  STAR_ENTER(h);
  if (!d) return 1;
  TqArgs a;
  a.A = d->A; a.W = d->W; a.O = d->O; a.bias = d->bias; a.colsum = d->colsum; a.rowab = d->rowab;
  a.lda = d->lda; a.ldo = d->ldo; a.HW = d->HW; a.F = d->F; a.C = d->C; a.heads = d->heads; a.scale = d->scale;
  return finish(h, op_temporal_qkv_attn(&h->c, a));
}
int star_gemm_rowstats(star_ctx* h, const star_gemm_desc* d, float* ln_partial, int32_t parts_cap, int32_t* parts, int32_t* wrote) {
  bool done = false;
  int np = 0;
  if (wrote) *wrote = 0;
  if (parts) *parts = 0;
  if (!h) return 1;
  if (!ln_partial || parts_cap <= 0) return finish(h, h->c.fail("gemm_rowstats: null partial buffer"));
  const int rc = gemm_from_desc(h, d, nullptr, nullptr, ln_partial, parts_cap, &np, &done);
  if (wrote) *wrote = done ? 1 : 0;
  if (parts) *parts = done ? np : 0;
  return rc;
}
int star_layer_norm_rowab_from_partials(star_ctx* h, const float* ln_partial, int32_t parts, float* rowab, int32_t rows, int32_t C, float eps,
                                        int32_t mode, const float* gate_w, float* maps, int32_t H, int32_t W) {
  STAR_ENTER(h);
  return finish(h, op_layer_norm_from_partials(&h->c, ln_partial, parts, rows, C, eps, mode, gate_w, maps, H, W, rowab));
}
int star_group_norm_from_partials(star_ctx* h, const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma, const float* beta,
                                  int32_t rows, int32_t C, int32_t rows_per_stat, float eps, int32_t silu, const float* gn_partial) {
  STAR_ENTER(h);
  if (!y) return finish(h, h->c.fail("group_norm_from_partials: null output"));
  if (!gn_partial) return finish(h, h->c.fail("group_norm_from_partials: null partial buffer (star_gemm_gn writes it)"));
  return finish(h, op_group_norm_fused(&h->c, x, ldx, y, ldy, gamma, beta, rows, C, rows_per_stat, eps, silu != 0, gn_partial));
}
int star_group_norm(star_ctx* h, const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma,
                    const float* beta, int32_t rows, int32_t C, int32_t rows_per_stat, float eps, int32_t silu) {
  STAR_ENTER(h);
  return finish(h, op_group_norm(&h->c, x, ldx, y, ldy, gamma, beta, rows, C, rows_per_stat, eps, silu != 0));
}
int star_layer_norm(star_ctx* h, const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma,
                    const float* beta, int32_t rows, int32_t C, float eps, int32_t mode, const float* gate_w,
                    float* maps, int32_t H, int32_t W) {
  STAR_ENTER(h);
  return finish(h, op_layer_norm(&h->c, x, ldx, y, ldy, gamma, beta, rows, C, eps, mode, gate_w, maps, H, W));
}
int star_layer_norm_rowab(star_ctx* h, const void* x, int32_t ldx, float* rowab, int32_t rows, int32_t C, float eps, int32_t mode,
                          const float* gate_w, float* maps, int32_t H, int32_t W) {
  STAR_ENTER(h);
  if (!rowab) return finish(h, h->c.fail("layer_norm_rowab: null output"));
  if (mode == LN_STATS_ONLY) return finish(h, h->c.fail("layer_norm_rowab: mode 3 (maps only) has no row statistics; use star_layer_norm"));
  return finish(h, op_layer_norm(&h->c, x, ldx, nullptr, 8, nullptr, nullptr, rows, C, eps, mode, gate_w, maps, H, W, rowab));
}
int star_concat_add(star_ctx* h, const void* a, const void* b, const void* c, void* out, int32_t rows, int32_t C1, int32_t C2) {
  STAR_ENTER(h);
  return finish(h, op_concat_add(&h->c, a, b, c, out, rows, C1, C2));
}
int star_add(star_ctx* h, const void* a, const void* b, void* out, int64_t n) {
  STAR_ENTER(h);
  return finish(h, op_add(&h->c, a, b, out, n));
}
int star_stem_im2col(star_ctx* h, const float* latent, void* out, int32_t Cl, int32_t F, int32_t H, int32_t W) {
  STAR_ENTER(h);
  return finish(h, op_stem_im2col(&h->c, latent, out, Cl, F, H, W));
}
int star_rows_to_latent(star_ctx* h, const float* rows, float* out, int32_t Cl, int32_t ld, int64_t ntok) {
  STAR_ENTER(h);
  return finish(h, op_rows_to_latent(&h->c, rows, out, Cl, ld, ntok));
}
int star_gemv(star_ctx* h, const float* x, const void* W, const float* b, float* y, int32_t N, int32_t K, int32_t silu_in, int32_t silu_out) {
  STAR_ENTER(h);
  return finish(h, op_gemv(&h->c, x, W, b, y, N, K, silu_in != 0, silu_out != 0));
}
int star_cast(star_ctx* h, const float* x, void* y, int64_t n) {
  STAR_ENTER(h);
  return finish(h, op_cast(&h->c, x, y, n));
}


int star_load_tensor(star_ctx* h, const char* name, const void* host, const int64_t* shape, int32_t ndim, int32_t dtype) {
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.resize(n);
  if (dtype == DT_F32) memcpy(t.data.data(), host, n * 4);
  else if (dtype == DT_F16) { const f16* s = (const f16*)host; for (size_t i = 0; i < n; ++i) t.data[i] = (float)s[i]; }
  else if (dtype == DT_BF16) { const uint16_t* s = (const uint16_t*)host; for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)s[i] << 16; memcpy(&t.data[i], &u, 4); } }
  else return h->c.fail("load_tensor: bad dtype");
  h->c.host_tensors[name] = std::move(t);
  return 0;
}
int star_clear_staged(star_ctx* h) { h->c.host_tensors.clear(); return 0; }
int star_unet_build(star_ctx* h, const star_unet_config* c) {
  STAR_ENTER(h);
  UNetCfg cfg;
  cfg.in_dim = c->in_dim; cfg.dim = c->dim; cfg.context_dim = c->context_dim; cfg.out_dim = c->out_dim;
  cfg.n_levels = c->n_levels;
  for (int i = 0; i < 8; ++i) cfg.dim_mult[i] = c->dim_mult[i];
  cfg.num_heads = c->num_heads; cfg.head_dim = c->head_dim; cfg.num_res_blocks = c->num_res_blocks; cfg.attn_levels = c->attn_levels;
  if (cfg.head_dim != 64) return h->c.fail("unet_build: head_dim must be 64");
  if (cfg.dim % 64) return h->c.fail("unet_build: dim must be a multiple of 64");
  if (cfg.n_levels < 1 || cfg.n_levels > 8) return h->c.fail("unet_build: bad n_levels");
  return finish(h, unet_build(&h->c, cfg));
}
int star_unet_forward(star_ctx* h, const float* xt, int64_t t, const float* y, const float* hint, float* out, int32_t f, int32_t hh, int32_t w) {
  STAR_ENTER(h);
  return finish(h, unet_forward(&h->c, xt, (long long)t, y, hint, out, f, hh, w));
}
int star_unet_forward_cfg(star_ctx* h, const float* xt, int64_t t, const float* y_cond, const float* y_uncond, const float* hint,
                          float* out_cond, float* out_uncond, int32_t f, int32_t hh, int32_t w) {
  STAR_ENTER(h);
  const float* ys[2] = {y_cond, y_uncond};
  float* outs[2] = {out_cond, out_uncond};
  return finish(h, unet_forward_n(&h->c, xt, (long long)t, ys, hint, outs, 2, f, hh, w));
}
int star_unet_graph(star_ctx* h, int32_t enable) {
  if (!h) return 1;
  h->c.unet_graph = enable != 0;
  return 0;
}
int star_controlnet_forward(star_ctx* h, const float* xt, int64_t t, const float* y, const float* hint, void* const* residuals,
                            int32_t n_residuals, int32_t f, int32_t hh, int32_t w) {
  STAR_ENTER(h);
  return finish(h, controlnet_forward(&h->c, xt, (long long)t, y, hint, residuals, n_residuals, f, hh, w));
}
int star_module_run(star_ctx* h, int32_t kind, const char* prefix, int32_t cin, int32_t cout, int32_t heads, int32_t embed_dim,
                    int32_t context_dim, const void* x, const float* emb, const float* context, void* out, int32_t f, int32_t hh, int32_t w) {
  STAR_ENTER(h);
  return finish(h, module_run(&h->c, kind, prefix, cin, cout, heads, embed_dim, context_dim, x, emb, context, out, f, hh, w));
}


int star_vae_build(star_ctx* h, const star_vae_config* c) {
  STAR_ENTER(h);
  VaeCfg cfg;
  cfg.in_ch = c->in_ch; cfg.out_ch = c->out_ch; cfg.latent = c->latent; cfg.n_blocks = c->n_blocks;
  for (int i = 0; i < 8; ++i) cfg.block_out[i] = c->block_out[i];
  cfg.layers_per_block = c->layers_per_block;
  if (cfg.n_blocks < 1 || cfg.n_blocks > 8) return h->c.fail("vae_build: bad n_blocks");
  for (int i = 0; i < cfg.n_blocks; ++i) if (cfg.block_out[i] % 64) return h->c.fail("vae_build: block_out channels must be multiples of 64");
  return finish(h, vae_build(&h->c, cfg));
}
int star_vae_encode(star_ctx* h, const float* x, float* moments, int32_t n, int32_t H, int32_t W) {
  STAR_ENTER(h);
  return finish(h, vae_encode(&h->c, x, moments, n, H, W));
}
int star_vae_decode(star_ctx* h, const float* z, float* out, int32_t n, int32_t hh, int32_t w) {
  STAR_ENTER(h);
  return finish(h, vae_decode(&h->c, z, out, n, hh, w));
}
int star_dit_build(star_ctx* h, const star_dit_config* c) {
  if (!h || !c) return -1;
  rt::set_device(h->c.device);
  return finish(h, dit_build(&h->c, c->hidden, c->heads, c->time_embed_dim, c->n_layers, c->ln_eps));
}
int star_dit_block_forward(star_ctx* h, int32_t layer, const void* hidden_in, const float* emb, void* hidden_out, int32_t text_len,
                           int32_t T, int32_t H, int32_t W) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  return finish(h, dit_block_forward(&h->c, layer, hidden_in, emb, hidden_out, text_len, T, H, W));
}
int star_softmax_rows(star_ctx* h, const float* s, int32_t lds, void* p, int32_t ldp, int32_t rows, int32_t n, float scale) {
  STAR_ENTER(h);
  return finish(h, op_softmax_rows(&h->c, s, lds, p, ldp, rows, n, scale));
}


int star_resize_pad(star_ctx* h, const float* src, float* dst, int32_t planes, int32_t hh, int32_t w, int32_t th, int32_t tw,
                    int32_t pad_l, int32_t pad_r, int32_t pad_t, int32_t pad_b, float pad_value) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  return finish(h, op_resize_pad(&h->c, src, dst, planes, hh, w, th, tw, pad_l, pad_r, pad_t, pad_b, pad_value));
}
int star_plane_stats(star_ctx* h, const float* x, float* stats, int32_t planes, int64_t n, float scale, float shift,
                     int32_t clamp01, float eps) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  return finish(h, op_plane_stats(&h->c, x, stats, planes, (long long)n, scale, shift, clamp01 != 0, eps));
}
int star_color_fix(star_ctx* h, const float* x, const float* src, float* out, int32_t F, int32_t C, int32_t H, int32_t W,
                   int32_t hh, int32_t w) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  return finish(h, op_color_fix(&h->c, x, true, src, out, F, C, H, W, hh, w));
}
int star_color_fix_u8(star_ctx* h, const float* x, const float* src, uint8_t* out, int32_t F, int32_t C, int32_t H, int32_t W,
                      int32_t hh, int32_t w) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  if (!x || !src || !out) return finish(h, h->c.fail("color_fix_u8: null pointer"));
  return finish(h, op_color_fix(&h->c, x, true, src, nullptr, F, C, H, W, hh, w, out));
}
int star_adain_color_fix(star_ctx* h, const float* target, const float* src, float* out, int32_t F, int32_t C, int32_t H, int32_t W,
                         int32_t hh, int32_t w) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  return finish(h, op_color_fix(&h->c, target, false, src, out, F, C, H, W, hh, w));
}

int star_profile_begin(star_ctx* h) {
  for (auto& r : h->c.prof) { rt::event_destroy(r.e0); rt::event_destroy(r.e1); }
  h->c.prof.clear();
  h->c.prof_mask = ~0u;
  h->c.profiling = true;
  return 0;
}
int star_profile_begin_kinds(star_ctx* h, uint32_t kind_mask) {
  const int rc = star_profile_begin(h);
  if (!rc) h->c.prof_mask = kind_mask;
  return rc;
}
int star_profile_end(star_ctx* h, star_prof_entry* out) {
  STAR_ENTER(h);
  rt::stream_sync(h->c.stream);
  if (const char* path = getenv("STAR_PROF_DETAIL")) {   // per-launch records: kind, dims, ms
    if (FILE* f = fopen(path, "a")) {
      for (auto& r : h->c.prof) fprintf(f, "%d,%d,%d,%d,%d,%.6f,%.6g\n", r.kind, r.d0, r.d1, r.d2, r.d3, rt::event_elapsed_ms(r.e0, r.e1), r.flops);
      fclose(f);
    }
  }
  for (int k = 0; k < PK_COUNT; ++k) out[k] = star_prof_entry{0, 0, 0, 0, 0, 0};
  double big_ms[PK_COUNT] = {0};   // the family's largest launches (equal, maximal algorithmic work): sum and count -> a true mean
  long long big_n[PK_COUNT] = {0};
  for (auto& r : h->c.prof) {
    const double ms = rt::event_elapsed_ms(r.e0, r.e1);
    star_prof_entry& e = out[r.kind];
    e.ms += ms; e.flops += r.flops; e.bytes += r.bytes; e.launches += 1;
    if (r.flops > e.max_flops) { e.max_flops = r.flops; big_ms[r.kind] = ms; big_n[r.kind] = 1; }
    else if (r.flops == e.max_flops && r.flops > 0) { big_ms[r.kind] += ms; big_n[r.kind] += 1; }
    rt::event_destroy(r.e0); rt::event_destroy(r.e1);
  }
  for (int k = 0; k < PK_COUNT; ++k) if (big_n[k]) out[k].max_flops_ms = big_ms[k] / (double)big_n[k];
  h->c.prof.clear();
  h->c.profiling = false;
  return 0;
}

int star_text_build(star_ctx* h, int32_t width, int32_t heads, int32_t layers) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  return finish(h, text_build(&h->c, width, heads, layers));
}
int star_text_forward(star_ctx* h, const void* x, int32_t batch, int32_t tokens, int32_t run_layers, void* out) {
  if (!h) return -1;
  rt::set_device(h->c.device);
  if (!x || !out) return finish(h, h->c.fail("text_forward: null pointer"));
  return finish(h, text_forward(&h->c, x, batch, tokens, run_layers, out));
}

}  // extern "C"
