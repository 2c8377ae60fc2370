// gemm.cpp -- argument checks and storage-type dispatch of the GEMM / implicit-conv operator (kernels: gemm.h; tile selection and
// launch: gemm_impl.h, compiled per type in gemm_f16.cpp / gemm_bf16.cpp).
#include <cstdlib>
#include "ops.h"

namespace star {

int launch_gemm_f16(Ctx* ctx, const GemmArgs& a);
int launch_gemm_bf16(Ctx* ctx, const GemmArgs& a);
bool gemm_astat_applies(const GemmArgs& a);          // gemm_as.cpp: A-stationary kernel for the wide short-K layers of level 0
int launch_gemm_astat(Ctx* ctx, const GemmArgs& a);

// A/B switches, read ONCE per process (the launch path runs ~3400 times per CFG forward: no environ scans there, and a concurrent
// setenv from another Python thread cannot race a launch)
struct GemmEnv {
  bool no_astat; int group_m; bool has_group_m;
  GemmEnv() {
    no_astat = std::getenv("STAR_NO_ASTAT") != nullptr;
    const char* e = std::getenv("STAR_GEMM_GROUP_M");
    has_group_m = e != nullptr;
    group_m = e ? std::atoi(e) : -1;
  }
};
#ifdef STAR_BENCH_VARIANTS   // the A/B tools (tools/ab_gemm_*.py) flip the switches between launches of one process
static GemmEnv gemm_env() { return GemmEnv(); }
#else
static const GemmEnv& gemm_env() { static const GemmEnv e; return e; }
#endif

int op_gemm(Ctx* ctx, const GemmArgs& a) {
  if (a.K % 64 != 0) return ctx->fail("gemm: K must be a multiple of 64");
  if (a.lda % 8 != 0) return ctx->fail("gemm: lda must be a multiple of 8");
  if (a.mode != A_PLAIN && a.Cin % 64 != 0) return ctx->fail("gemm: conv Cin must be a multiple of 64");
  if ((a.epi & EPI_GEGLU) && (a.N % 64 != 0)) return ctx->fail("gemm: GEGLU needs N % 64 == 0");
  if ((a.epi & (EPI_GEGLU | EPI_GELU_TANH)) && (a.epi & EPI_OUT_F32)) return ctx->fail("gemm: GEGLU / GELU with fp32 output is not supported");
  if (a.epi & EPI_OUT_F32) {
    if (a.N % 4 || a.ldc % 4 || ((a.epi & EPI_RES) && a.ldr % 4)) return ctx->fail("gemm: fp32 output needs N, ldc (and ldr) multiples of 4");
  } else {
    if (a.N % 8 || a.ldc % 8 || ((a.epi & EPI_RES) && a.ldr % 8)) return ctx->fail("gemm: N, ldc (and ldr) must be multiples of 8");
  }
  if (a.M <= 0 || a.N <= 0) return 0;
  // the gathered modes address their input with 32-bit offsets from a per-tile base (gemm.h): keep them below 2^31
  if (a.mode == A_CONV3X3) {   // input rows one 256-row output tile can reach from its first source row
    const long long hw = (long long)a.Ho * a.Wo;
    const long long rows = hw > 256 ? 2 * ((256 / a.Wo + 3LL) * a.stride + 3) : (256 / hw + 2) * (a.H + 3LL);
    if (rows * a.Wd * a.lda * 2 >= (1LL << 31)) return ctx->fail("gemm: conv image rows too large for 32-bit gather offsets");
  }
  if (a.mode == A_TCONV3 && (256LL * a.lda + a.Cin) * 2 >= (1LL << 31))
    return ctx->fail("gemm: temporal-conv rows too large for 32-bit gather offsets");
  ProfScope ps(ctx, a.mode == A_PLAIN ? PK_GEMM : (a.mode == A_TCONV3 ? PK_TCONV : PK_CONV), 2.0 * a.M * (double)a.N * a.K,
               ((double)a.M * (a.mode == A_PLAIN ? a.K : a.Cin) + (double)a.M * ((a.epi & EPI_GEGLU) ? a.N / 2 : a.N)) * 2.0,
               a.M, a.N, a.K, a.epi);
  if ((a.force_tile >= 29 && a.force_tile <= 50) || (a.force_tile == 0 && gemm_astat_applies(a) && !gemm_env().no_astat)) return launch_gemm_astat(ctx, a);
  if (a.group_m < 0 && gemm_env().has_group_m) {   // A/B aid: STAR_GEMM_GROUP_M overrides the automatic choice of the tile walk (gemm.h)
    GemmArgs b = a;
    b.group_m = gemm_env().group_m;
    if (ctx->dtype == DT_F16) return launch_gemm_f16(ctx, b);
    if (ctx->dtype == DT_BF16) return launch_gemm_bf16(ctx, b);
  }
  if (ctx->dtype == DT_F16) return launch_gemm_f16(ctx, a);
  if (ctx->dtype == DT_BF16) return launch_gemm_bf16(ctx, a);
  return ctx->fail("gemm: unsupported dtype");
}

}  // namespace star
