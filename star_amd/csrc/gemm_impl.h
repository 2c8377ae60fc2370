// gemm_impl.h -- tile selection + launch for gemm_kernel (see gemm.h), instantiated once per storage type in gemm_f16.cpp /
// gemm_bf16.cpp (two translation units: gemm_kernel has ~100 instantiations per type and is the longest compile of the build).
#pragma once
#include <cstdlib>
#include "ops.h"
#include "gemm.h"

namespace star {

// ALLEPI: also instantiate the tanh-GELU and folded-LayerNorm epilogue flavours (the auto-selected tiles 1-4 only: every
// flavour is one more kernel per tile, mode and dtype, and these are the longest compiles of the build)
// GNS: also instantiate the GroupNorm-statistics flavours (EPIF 16 / 17) of the plain / 3x3 / temporal-conv modes (tiles 2, 3 and 17)
template <class T, int BM, int BN, int WM, int WN, int MINW, bool F32OUT, bool STAGGER, int PIPE = 0, bool ALLEPI = false, int SCHED = 0, int ABLV = 0, bool GNS = false>
static int launch_gemm_f(Ctx* ctx, const GemmArgs& a) {
  GemmParams p{};
  p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.res = a.res; p.zero_page = ctx->zero_page;
  p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldc = a.ldc; p.ldr = a.ldr;
  p.H = a.H; p.Wd = a.Wd; p.Cin = a.Cin; p.Ho = a.Ho; p.Wo = a.Wo; p.stride = a.stride; p.pad_t = a.pad_t; p.pad_l = a.pad_l;
  p.HW = a.HW; p.F = a.F; p.up_crop = a.up_crop; p.epi = a.epi;
  p.rowab = a.rowab; p.colsum = a.colsum;
  p.gn_partial = a.gn_partial;
  // temporal conv: frame-interleaved tile walk (gemm.h: t_walk) when the launch covers whole frames and a frame is at least one tile row:
  // level 0 -5.6 %, level 1 -2.0 %, bit-identical (profiles/r05_cbench_tconv_walk.txt); STAR_NO_TCONV_WALK=1 = row-major (A/B, read once)
  { static const bool tw = std::getenv("STAR_NO_TCONV_WALK") == nullptr;
    p.t_walk = (tw && a.mode == A_TCONV3 && a.m_off == 0 && a.m_end == 0 && a.F > 1 && (long long)a.F * a.HW == a.M && a.HW >= BM) ? 1 : 0; }
  p.ln_partial = a.ln_partial;
  p.m_off = a.m_off;
  p.tiles_m = ((a.m_end > 0 ? a.m_end : a.M) - a.m_off + BM - 1) / BM;
  p.tiles_n = (a.N + BN - 1) / BN;
  p.group_m = a.group_m >= 0 ? a.group_m : (p.tiles_n >= 12 ? 8 : 1);   // auto: where a row of tiles is wide (measured +3-11 %, 8192^3 1140 -> 1233 TF/s; narrower rows lose 0-4 %: profiles/r03_gemm_group_ab.txt)
  // 2 stages x 64 k, or PIPE ring slots x 32 k; never less than the epilogue's per-wave staging blocks
  constexpr size_t smem_loop = PIPE ? (size_t)PIPE * (BM + BN) * 64 : 2 * (size_t)(BM + BN) * 128;
  constexpr size_t smem_epi = (size_t)WM * WN * 32 * (BN / WN * 2 + 8);
  static_assert(smem_epi <= smem_loop, "the epilogue's staging blocks must fit under the bias slice");
  constexpr size_t smem = smem_loop + 2 * (size_t)BN * sizeof(float);   // + this tile's bias slice (+ column sums of a folded LayerNorm)
  unsigned nwg = (unsigned)(p.tiles_m * p.tiles_n);
  if (SCHED == 0 && a.persist > 0 && nwg > (unsigned)a.persist) nwg = (unsigned)a.persist;   // persistent tile walk (gemm.h)
  dim3 grid(nwg), block(WM * WN * 64);
  // 16-bit epilogue flavour (compile-time in the kernel): plain / + residual / GEGLU (plain-A layers only, no residual)
  const bool res = !F32OUT && (a.epi & EPI_RES), geglu = !F32OUT && (a.epi & EPI_GEGLU), gelut = !F32OUT && (a.epi & EPI_GELU_TANH);
  if ((geglu || gelut) && (res || a.mode != A_PLAIN || (geglu && gelut))) return ctx->fail("gemm: GEGLU / tanh-GELU are for plain-A layers without a residual");
  if (F32OUT && (a.epi & EPI_ROWAFF)) return ctx->fail("gemm: the folded-LayerNorm epilogue has no fp32-output form");
  const bool rowaff = !F32OUT && (a.epi & EPI_ROWAFF);
  if (rowaff && (res || gelut || a.mode != A_PLAIN || !a.rowab || !a.colsum || !(a.epi & EPI_BIAS)))
    return ctx->fail("gemm: the folded-LayerNorm epilogue is for plain-A layers without a residual and needs rowab, colsum and bias");
#define STAR_GEMM_GO(MODE, EF) STAR_LAUNCH((gemm_kernel<T, BM, BN, WM, WN, MODE, MINW, F32OUT, STAGGER, ABLV, PIPE, EF, SCHED>), grid, block, smem, ctx->stream, p)
  if (a.ln_partial) {   // per-row statistics of the output (plain-A layers; checked by launch_gemm)
    if constexpr (GNS && !F32OUT && SCHED == 0) {
      p.ln_parts = p.tiles_n * WN;
      if (a.mode != A_PLAIN || p.ln_parts > a.ln_parts_cap) return ctx->fail("gemm: row statistics: plain-A layers, parts within the caller's buffer");
      if (res) STAR_GEMM_GO(A_PLAIN, 33); else STAR_GEMM_GO(A_PLAIN, 32);
      if (a.ln_parts) *a.ln_parts = p.ln_parts;
      if (a.ln_done) *a.ln_done = true;
      return 0;
    } else return ctx->fail("gemm: this tile has no row-statistics flavour");
  }
  if (a.gn_partial) {   // the caller (launch_gemm) has checked mode and epilogue; the flavour exists for GNS tiles only
    if constexpr (GNS && !F32OUT) {
      // (the scheduled one-wave-per-SIMD tile has the flavour without a residual only: with the residual's registers beside its 256
      // accumulators the bf16 instantiation spilled one register; launch_gemm drops the request there)
      // (round 6: the 256 x 320 scheduled tile, BN == 320, is tried WITH the residual flavour: its fifth block column is in architectural registers anyway)
      constexpr bool RES_GN = SCHED == 0 || BN == 320;
      if constexpr (!RES_GN) { if (res) return ctx->fail("gemm: the scheduled tile has no GroupNorm-statistics flavour with a residual"); }
#define STAR_GEMM_GO_GN(MODE) do { if constexpr (RES_GN) { if (res) STAR_GEMM_GO(MODE, 17); else STAR_GEMM_GO(MODE, 16); } else STAR_GEMM_GO(MODE, 16); } while (0)
      switch (a.mode) {
        case A_PLAIN: STAR_GEMM_GO_GN(A_PLAIN); break;
        case A_CONV3X3: STAR_GEMM_GO_GN(A_CONV3X3); break;
        case A_TCONV3: STAR_GEMM_GO_GN(A_TCONV3); break;
        default: return ctx->fail("gemm: no GroupNorm-statistics flavour of this mode");
      }
#undef STAR_GEMM_GO_GN
      if (a.gn_done) *a.gn_done = true;
      return 0;
    } else return ctx->fail("gemm: this tile has no GroupNorm-statistics flavour");
  }
  switch (a.mode) {
    case A_PLAIN:
      if constexpr (F32OUT) STAR_GEMM_GO(A_PLAIN, 0);
      else {
        if (rowaff || gelut) {
          if constexpr (ALLEPI) {
            if (rowaff) { if (geglu) STAR_GEMM_GO(A_PLAIN, 10); else STAR_GEMM_GO(A_PLAIN, 8); }
            else STAR_GEMM_GO(A_PLAIN, 4);
          } else return ctx->fail("gemm: this tile has no tanh-GELU / folded-LayerNorm epilogue (tiles 1-4 do)");
        }
        else if (geglu) { if constexpr (SCHED != 0) return ctx->fail("gemm: no GEGLU flavour of the scheduled tile"); else STAR_GEMM_GO(A_PLAIN, 2); }
        else if (res) STAR_GEMM_GO(A_PLAIN, 1); else STAR_GEMM_GO(A_PLAIN, 0);
      }
      break;
    case A_CONV3X3:
      if constexpr (F32OUT) STAR_GEMM_GO(A_CONV3X3, 0); else { if (res) STAR_GEMM_GO(A_CONV3X3, 1); else STAR_GEMM_GO(A_CONV3X3, 0); }
      break;
    case A_CONV3X3_UP:
      if constexpr (SCHED != 0) return ctx->fail("gemm: the scheduled tile has no nearest-x2 conv mode");
      else if constexpr (F32OUT) STAR_GEMM_GO(A_CONV3X3_UP, 0); else { if (res) STAR_GEMM_GO(A_CONV3X3_UP, 1); else STAR_GEMM_GO(A_CONV3X3_UP, 0); }
      break;
    case A_TCONV3:
      if constexpr (F32OUT) STAR_GEMM_GO(A_TCONV3, 0); else { if (res) STAR_GEMM_GO(A_TCONV3, 1); else STAR_GEMM_GO(A_TCONV3, 0); }
      break;
    default: return ctx->fail("gemm: bad A mode");
  }
#undef STAR_GEMM_GO
  return 0;
}

template <class T, int BM, int BN, int WM, int WN, int MINW, bool STAGGER = false, int PIPE = 0, bool ALLEPI = false, bool GNS = false>
static int launch_gemm_t(Ctx* ctx, const GemmArgs& a) {
  if (a.epi & EPI_OUT_F32) return launch_gemm_f<T, BM, BN, WM, WN, MINW, true, false, 0>(ctx, a);
  return launch_gemm_f<T, BM, BN, WM, WN, MINW, false, STAGGER, PIPE, ALLEPI, 0, 0, GNS>(ctx, a);
}

int launch_gemm_persist(Ctx* ctx, const GemmArgs& a);   // gemm_p.cpp
bool gemm_persist_covers(const GemmArgs& a);

#ifdef STAR_BENCH_VARIANTS
static bool no_sched320_env() { return std::getenv("STAR_NO_SCHED320") != nullptr; }
static bool no_sched_env() { return std::getenv("STAR_NO_SCHED") != nullptr || std::getenv("STAR_SCHED17") == nullptr; }
static bool no_persist_env() { return std::getenv("STAR_NO_PERSIST") != nullptr; }
#else
static bool no_persist_env() { static const bool v = std::getenv("STAR_NO_PERSIST") != nullptr; return v; }   // read once (A/B switch)
static bool no_sched320_env() { static const bool v = std::getenv("STAR_NO_SCHED320") != nullptr; return v; }   // read once (A/B switch)
// tile 17 is OFF unless STAR_SCHED17=1 since the end of round 6: it wins 4-7 % per layer in kernel loops and single forwards and LOSES as whole
// clips -- three alternations on one box 55.76 against 55.57 s without it (-0.34 %), -0.24 % and -0.2 % on two other boxes
// (profiles/r06_same_box_tile17_clips.txt, r06_same_box_switches*.txt); the layers go to tile 19 (same arithmetic, bit-identical outputs)
static bool no_sched_env() { static const bool v = std::getenv("STAR_NO_SCHED") != nullptr || std::getenv("STAR_SCHED17") == nullptr; return v; }   // read once (A/B switch)
#endif

template <class T>
static int launch_gemm(Ctx* ctx, const GemmArgs& a) {
  int tile = a.force_tile;
  if (tile >= 100 && tile < 1000) {   // A/B: 1xx = tile xx walked by 256 persistent workgroups, 2xx by 512
    GemmArgs b = a;
    b.force_tile = tile % 100;
    b.persist = 256 * (tile / 100);
    return launch_gemm<T>(ctx, b);
  }
  if (!tile) {
    const bool geglu = (a.epi & EPI_GEGLU) != 0;
    // long-K gathered layers whose width is a multiple of 256 and that fill the chip at least twice (3x3 convs and temporal convs of
    // the 1280-wide levels): the one-wave-per-SIMD tile, 4-7 % faster there and bit-identical (profiles/r03_gemm_sched_ab.txt);
    // with fewer tiles the 256 x 320 tile's smaller tail wins, and plain-A layers of these shapes tie
    const bool sched_ok = (a.mode == A_CONV3X3 || a.mode == A_TCONV3) && a.N % 256 == 0 && a.K >= 2560 &&
                          !(a.epi & (EPI_OUT_F32 | EPI_ROWAFF | EPI_GELU_TANH | EPI_GEGLU)) &&
                          (int64_t)((a.M + 255) / 256) * (a.N / 256) >= 512 && !no_sched_env();
    // plain-A layers with K >= 512 whose width is a whole number of 256-column tiles and that fill the chip at least twice: the
    // persistent one-wave-per-SIMD tile (gemm_p.h), 5-9 % ahead of the 8-wave tiles there (q | k | v, out-projections and FF-out of
    // level 2, the GEGLU projections of levels 1-2, the DiT dense layers: profiles/r04_gemm_ab.txt); narrower or ragged widths lose
    // to the 256 x 320 tile (N = 640 would waste a sixth of three 256-column tiles)
    // ... and whose tiles fill the resident workgroups' rounds to >= 88 % (the persistent walk is static: 1080 tiles on 256 CUs are 5
    // rounds for some workgroups; there the 256 x 320 tile + tail split measured ahead, profiles/r04_gemm_ab_v3_auto.txt)
    // (round 6, measured and dropped: accepting a half-empty last column tile -- the level-1 q | k | v, N = 1920 -- is +3.0 % in cbench
    // (profiles/r06_cbench_ffpo_tiles.txt) and -8 % IN SITU (0.599 -> 0.647 ms, same box: profiles/r06_forward_detail_f16_ragged18.txt /
    // _noragged18.txt): the in-situ figure decides.  Asking for the output's GroupNorm partials keeps a layer off this tile; the only such
    // layer it could run, the level-2 composed FF-out / proj_out GEMM, is 3.6 % faster on the 256 x 320 tile + tail split anyway: ADVICE r05.)
    bool persist_ok = gemm_persist_covers(a) && a.K >= 512 && a.N % 256 == 0 && a.N >= 1024 && !no_persist_env() && !a.gn_partial && !a.ln_partial;
    if (persist_ok) {
      const int cus = a.assume_cus > 0 ? a.assume_cus : (ctx->num_cus > 0 ? ctx->num_cus : 256);
      const int64_t nt = (int64_t)((a.M + 255) / 256) * (a.N / 256), rounds = (nt + cus - 1) / cus;
      persist_ok = nt >= 2 * (int64_t)cus && (double)nt >= 0.88 * (double)(rounds * cus);
    }
    // round 6: the same scheduled loop on the 256 x 320 tile (4 waves x (128 x 160), 320 accumulators per wave: tile 19) for the long-K
    // layers the 8-wave 256 x 320 tile ran: 9 fragment reads per 20 MFMAs instead of 7 per 10, bit-identical.  Decided IN SITU (whole
    // forwards with / without on one box, profiles/r06_forward_detail_f16_tile19_*.txt / _notile19_*.txt): 3 x 3 convs -7 ... -9.5 % (K = 2880
    // ... 23040; the level-0 conv2 with residual + statistics -1.4 %), temporal convs K = 1920 -3.8 %, K = 3840 at level 3 -7.4 % (K = 960:
    // 0 in cbench, not taken), plain K = 6400 -2.4 % but K = 2560 +3.6 % (the 8-wave tile hides a short loop's prologue better: plain layers
    // from K = 3200 only): -4.9 ms per forward.  STAR_NO_SCHED320=1: tile 2 (A/B switch, read once).
    const bool sched320_ok = !geglu && a.N % 320 == 0 && !(a.M <= 4096 && a.N <= 1024) && !a.ln_partial &&
                             !(a.epi & (EPI_OUT_F32 | EPI_ROWAFF | EPI_GELU_TANH | EPI_GEGLU)) &&
                             ((a.mode == A_CONV3X3 && a.K >= 2880) || (a.mode == A_TCONV3 && a.K >= 1920) || (a.mode == A_PLAIN && a.K >= 3200)) &&
                             !no_sched320_env();
    if (persist_ok) tile = 18;
    else if (sched_ok) tile = 17;
    else if (sched320_ok) tile = 19;
    else if (a.M <= 4096 && a.N <= 1024) tile = 3;                     // small problems: more, smaller tiles
    else if (!geglu && a.N % 320 == 0) tile = 2;                        // 320 / 640 / 960 / 1280 / 1920-wide layers
    // (rounds 1-2 sent the short-K GEGLU layers to tile 9, two 4-wave workgroups per CU hiding each other's GELU epilogue;
    // with the erfc-form GELU and the compile-time epilogue the 8-wave 256x256 tile is 7 % faster there: 1.89 vs 2.04 ms at
    // 843264 x 2560 x 320, profiles/r02_gemm_tiles_after_valu.txt)
    else if (a.N <= 128) tile = 4;
    else tile = 1;
  }
  // LayerNorm row statistics in the epilogue: tiles 2 and 3, plain-A layers with the bias (+ residual) 16-bit epilogue, never together
  // with the GroupNorm partials; elsewhere the request is dropped (ln_done stays false)
  if (a.ln_partial) {
    const int bn = tile == 2 ? 320 : 128;
    const bool ok_ = (tile == 2 || tile == 3) && a.mode == A_PLAIN && !a.gn_partial &&
                     !(a.epi & (EPI_OUT_F32 | EPI_ROWAFF | EPI_GELU_TANH | EPI_GEGLU)) && !(a.N & 7) && ((a.N + bn - 1) / bn) * 2 <= a.ln_parts_cap;
    if (!ok_) {
      GemmArgs b = a;
      b.ln_partial = nullptr;
      return launch_gemm<T>(ctx, b);
    }
  }
  // GroupNorm statistics in the epilogue: the flavour exists for tiles 1-4 and 17 on plain / 3x3 / temporal-conv layers with the
  // bias (+ residual) 16-bit epilogue (round 6: + the power-of-two tiles 1 and 4, which run the VAE's 128 / 256 / 512-wide layers);
  // elsewhere the request is dropped (gn_done stays false: the consumer runs its own pass)
  if (a.gn_partial && (!(tile == 1 || tile == 2 || tile == 3 || tile == 4 || tile == 17 || tile == 19) || (tile == 17 && (a.epi & EPI_RES)) || !(a.mode == A_PLAIN || a.mode == A_CONV3X3 || a.mode == A_TCONV3) ||
                       (a.epi & (EPI_OUT_F32 | EPI_ROWAFF | EPI_GELU_TANH | EPI_GEGLU)) || (a.N & 7))) {
    GemmArgs b = a;
    b.gn_partial = nullptr;
    return launch_gemm<T>(ctx, b);
  }
  // ---- tail split.  The big tiles run ONE workgroup per CU, so a launch is ceil(tiles / CUs) rounds and the last round can be mostly
  // empty: the 1280-wide layers of level 2 (M = 55296: 864 tiles of 256 x 320 = 3.4 rounds, 1080 of 256 x 256 = 4.2) spent 16 % of
  // their time with three quarters of the chip idle.  When the last round is poorly filled, the launch covers only the tile rows of
  // the FULL rounds and the remaining rows go to a second launch of 128 x 128 tiles (tile 3: two workgroups per CU, every mode and
  // epilogue flavour, same k order per output -- bit-identical).  Decided by a cost model in units of one big tile's time.
  if (!a.force_tile && a.m_off == 0 && a.m_end == 0 && !a.ln_partial && (tile == 1 || tile == 2 || tile == 17 || tile == 19)) {   // (row statistics: the remainder tile would cut the rows into a different number of parts)   // (the persistent tile 18 is only chosen where its rounds are full)
    const int cus = a.assume_cus > 0 ? a.assume_cus : (ctx->num_cus > 0 ? ctx->num_cus : 256);
    const int bm = 256, bn = (tile == 2 || tile == 19) ? 320 : 256;
    const long long tm = (a.M + bm - 1) / bm, tn = (a.N + bn - 1) / bn, nt = tm * tn;
    const long long full = nt / cus;                      // full rounds
    if (full >= 1 && nt % cus != 0) {
      const long long main_rows_t = full * cus / tn;      // tile rows of the main launch (its tiles fit `full` rounds)
      const long long rem_rows = a.M - main_rows_t * bm;
      // the remainder's tile is 128 x 128 (tile 3: every mode and flavour).  The 128 x 320 tile 10 was tried as the remainder of the
      // 320-multiple widths and made the split layers 6-9 % SLOWER than unsplit (its 32-deep ring loop; profiles/r04_same_box_r03_vs_r04.txt)
      const int rbn = 128, rtile = 3;
      const long long ns = ((rem_rows + 127) / 128) * ((a.N + rbn - 1) / rbn);
      const double small_work = 128.0 * rbn / ((double)bm * bn);            // one remainder tile in units of a big one
      double rem_t = (double)ns * small_work / cus;                         // remainder work per CU ...
      if (rem_t < small_work) rem_t = small_work;                           // ... never less than one remainder tile
      rem_t /= 0.6;                                                         // the 128 x 128 tiles run at ~0.6 of the big tiles' rate
      const double t_split = (double)((main_rows_t * tn + cus - 1) / cus) + rem_t, t_plain = (double)((nt + cus - 1) / cus);
      if (main_rows_t >= 1 && rem_rows > 0 && t_split < 0.95 * t_plain) {
        GemmArgs m = a, r = a;
        m.force_tile = tile; m.m_end = (int)(main_rows_t * bm);
        r.force_tile = rtile; r.m_off = (int)(main_rows_t * bm);
        ++ctx->gemm_splits;
        if (int rc = launch_gemm<T>(ctx, m)) return rc;
        return launch_gemm<T>(ctx, r);
      }
    }
  }
  switch (tile) {
    // 8 waves per workgroup (2 per SIMD, <= 256 VGPRs each; 192 / 236 used, no spills): measured 1.3-3.6x faster than
    // 4-wave variants of the same tiles on the K = 320 layers (profiles/r01_gemm_tile_sweep.txt)
    case 1: return launch_gemm_t<T, 256, 256, 4, 2, 2, false, 0, true, true>(ctx, a);
    case 2: return launch_gemm_t<T, 256, 320, 4, 2, 2, false, 0, true, true>(ctx, a);
    // 4 waves x (128 x 128), one wave per SIMD, hand-placed 2-stage loop (gemm.h SCHED): plain / 3x3 conv / temporal conv, 16-bit output
    case 17:
      if (a.epi & (EPI_OUT_F32 | EPI_ROWAFF | EPI_GELU_TANH | EPI_GEGLU)) return ctx->fail("gemm: tile 17 has the plain and residual 16-bit epilogues only");
      return launch_gemm_f<T, 256, 256, 2, 2, 1, false, false, 0, false, 1, 0, true>(ctx, a);
    // round 6: the scheduled loop on the 256 x 320 tile: 4 waves x (128 x 160), 320 accumulators per wave, 9 fragment reads per 20 MFMAs
    case 19:
      if (a.epi & (EPI_OUT_F32 | EPI_ROWAFF | EPI_GELU_TANH | EPI_GEGLU)) return ctx->fail("gemm: tile 19 has the plain and residual 16-bit epilogues only");
      return launch_gemm_f<T, 256, 320, 2, 2, 1, false, false, 0, false, 1, 0, true>(ctx, a);
    case 18: return launch_gemm_persist(ctx, a);   // persistent one-wave-per-SIMD tile with a wave-private epilogue (gemm_p.h)
#ifdef STAR_BENCH_VARIANTS   // round-6 A/Bs of tile 18's walk and store policy (bit-identical): 60 = PLAIN stores, the round-5 kernel (the product stores non-temporally since round 6; + 66: plain stores on column strips of 8);
    // 61 / 62 / 63 = row groups of 1 / 4 / 16 (product: 8 where a tile row is >= 12 tiles wide); 64 / 65 / 67 = column strips of 8 / 4 / 16
    case 60: return launch_gemm_persist(ctx, a);
    // 69 = tile 9 (two independent 4-wave workgroups per CU on 128 x 256 tiles: one group's GEGLU epilogue runs beside the other's K loop)
    // WITH the folded-LayerNorm flavour, so that it can run the model's GEGLU layers
    case 69: return launch_gemm_t<T, 128, 256, 2, 2, 2, false, 3, true>(ctx, a);
    case 61: case 62: case 63: case 64: case 65: case 66: case 67: {
      GemmArgs b = a;
      static const int gm[7] = {1, 4, 16, -8, -4, -8, -16};
      b.group_m = gm[tile - 61];
      b.force_tile = tile == 66 ? 60 : 18;
      return launch_gemm_persist(ctx, b);
    }
#endif
    case 3: return launch_gemm_t<T, 128, 128, 2, 2, 2, false, 0, true, true>(ctx, a);
    case 4: return launch_gemm_t<T, 256, 128, 4, 1, 1, false, 0, true, true>(ctx, a);
#ifdef STAR_BENCH_VARIANTS   // A/B experiments of round 1 / 2 that lost to the tiles above (profiles/r01_gemm_ablation.txt, r02_gemm8_ablation.txt)
    case 5: return launch_gemm_t<T, 256, 256, 4, 2, 2, true>(ctx, a);   // staggered wave groups (A/B)
    // (the 256x320 tile has no room for the carried fragments: 730+ VGPR spills when staggered)
    case 7: return launch_gemm_t<T, 256, 256, 4, 2, 2, false, 4>(ctx, a);   // pipelined main loop (ring of 4 x 32-k slots)
    case 8: return launch_gemm_t<T, 256, 320, 4, 2, 2, false, 4>(ctx, a);
    case 14: return launch_gemm_t<T, 256, 256, 2, 2, 1>(ctx, a);
    // timing ablations of tile 17 (garbage results): no DMA in the loop / every copy re-reads one 1 KB piece (L1 hits) / DMA never waited for / A pieces only
    case 20: case 21: case 22: case 23:
      if (a.mode != A_PLAIN || (a.epi & ~EPI_BIAS)) return ctx->fail("gemm: ablation tile");
      if (tile == 20) return launch_gemm_f<T, 256, 256, 2, 2, 1, false, false, 0, false, 1, 8>(ctx, a);
      if (tile == 21) return launch_gemm_f<T, 256, 256, 2, 2, 1, false, false, 0, false, 1, 9>(ctx, a);
      if (tile == 22) return launch_gemm_f<T, 256, 256, 2, 2, 1, false, false, 0, false, 1, 10>(ctx, a);
      return launch_gemm_f<T, 256, 256, 2, 2, 1, false, false, 0, false, 1, 11>(ctx, a);   // 4 waves x (128 x 128), one wave per SIMD, accumulators in AGPRs: 1/3 fewer LDS fragment reads
#endif
    // two independent 4-wave workgroups per CU on a 128 x 320 tile (two 32-deep LDS slots, 56 KB each): one group's prologue /
    // epilogue / store drain runs beside the other's MFMAs -- for the short-K layers, whose tiles spend most of their time
    // outside the K loop
    case 10: return launch_gemm_t<T, 128, 320, 2, 2, 2, false, 2>(ctx, a);
    // two independent 4-wave workgroups per CU (72 KB of LDS each): one group's epilogue and DMA latency hide behind the
    // other group's MFMA burst (auto-selected for the short-K GEGLU layers)
    case 9: return launch_gemm_t<T, 128, 256, 2, 2, 2, false, 3>(ctx, a);
  }
#ifdef STAR_BENCH_VARIANTS
  if (tile >= 11 && tile <= 16 && tile != 14 && a.mode == A_PLAIN && !(a.epi & EPI_OUT_F32)) {   // ablation probes of the 256x256 main loop
    GemmParams p{};
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.res = a.res; p.zero_page = ctx->zero_page;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldc = a.ldc; p.ldr = a.ldr; p.epi = a.epi;
    p.tiles_m = (a.M + 255) / 256; p.tiles_n = (a.N + 255) / 256;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(512);
    const size_t smem = 2 * (size_t)512 * 128 + 256 * sizeof(float);
    if (tile == 11) STAR_LAUNCH((gemm_kernel<T, 256, 256, 4, 2, A_PLAIN, 2, false, false, 1>), grid, block, smem, ctx->stream, p);
    if (tile == 12) STAR_LAUNCH((gemm_kernel<T, 256, 256, 4, 2, A_PLAIN, 2, false, false, 2>), grid, block, smem, ctx->stream, p);
    if (tile == 13) STAR_LAUNCH((gemm_kernel<T, 256, 256, 4, 2, A_PLAIN, 2, false, false, 3>), grid, block, smem, ctx->stream, p);
    if (tile == 15) STAR_LAUNCH((gemm_kernel<T, 256, 256, 4, 2, A_PLAIN, 2, false, false, 4>), grid, block, smem, ctx->stream, p);   // no K loop at all
    if (tile == 16) STAR_LAUNCH((gemm_kernel<T, 256, 256, 4, 2, A_PLAIN, 2, false, false, 5>), grid, block, smem, ctx->stream, p);   // no global stores
    return 0;
  }
#endif
  return ctx->fail("gemm: bad tile id (experimental tiles exist only in the bench build)");
}

}  // namespace star
