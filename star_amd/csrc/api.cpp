// api.cpp -- the C ABI (include/star_hip.h) over the C++ launchers.
#include "../../include/star_hip.h"
#include "ops.h"

using namespace star;

struct star_ctx { Ctx c; };

extern "C" {

int star_is_hostemu(void) {
#ifdef STAR_HOSTEMU
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
  star_ctx* h = new star_ctx();
  h->c.device = device_id;
  h->c.dtype = dtype;
  void* z = nullptr;
  if (rt::dev_malloc(&z, 256)) { delete h; return 5; }
  rt::memset_async(z, 0, 256, nullptr);
  rt::stream_sync(nullptr);
  h->c.zero_page = z;
  *out = h;
  return 0;
}

void star_ctx_destroy(star_ctx* h) {
  if (!h) return;
  rt::stream_sync(h->c.stream);
  h->c.unet.reset();
  h->c.vae.reset();
  h->c.pool.release();
  if (h->c.zero_page) rt::dev_free(h->c.zero_page);
  delete h;
}

const char* star_last_error(star_ctx* h) { return h ? h->c.err.c_str() : "null ctx"; }
int star_set_stream(star_ctx* h, void* s) { h->c.stream = (hipStream_t)s; return 0; }
int star_sync(star_ctx* h) {
  if (rt::stream_sync(h->c.stream)) return h->c.fail(std::string("sync failed: ") + rt::last_error_string());
  return 0;
}
size_t star_pool_bytes(star_ctx* h) { return h->c.pool.total(); }
size_t star_pool_peak_bytes(star_ctx* h) { return h->c.pool.peak(); }

int star_gemm(star_ctx* h, const star_gemm_desc* d) {
  GemmArgs a;
  a.A = d->A; a.W = d->W; a.C = d->C; a.bias = d->bias; a.res = d->res;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr;
  a.mode = d->mode; a.H = d->H; a.Wd = d->Wd; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.HW = d->HW; a.F = d->F;
  a.epi = d->epi; a.force_tile = d->force_tile;
  return op_gemm(&h->c, a);
}

}  // extern "C"
