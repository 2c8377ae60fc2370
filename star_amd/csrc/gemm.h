// gemm.h -- the MFMA GEMM core behind every Linear / Conv2d / Conv3d(3,1,1) of the
// hot path (reference call sites unet_v2v.py:151-155,274,294,
// 500,526,612,639,648,717,1005,1025,1209-1220).
//
//   C[M,N] = epilogue( A'[M,K] * W[N,K]^T )
//
// * A' is never materialised for convolutions: the A-tile loader gathers the
//   NHWC (channels-last, tokens x C) activation rows for each (tap, channel
//   block) straight into LDS with 16-byte LDS-DMA (global_load_lds), pointing
//   out-of-image taps at a zero page.  K is ordered (tap, cin); weights are
//   repacked to [N][taps][Cin] at load time.
// * Tile BM x BN x 64, 32x32x16 MFMA, two LDS stages, one barrier per K tile.
//   LDS rows are 128 B (64 k-elements); the 16-B chunk c of row r lives at
//   chunk position c ^ ((r>>1)&7), applied on the *source* address because
//   LDS-DMA writes lane-linearly; fragment ds_read_b128 are then conflict-free.
// * The MFMA is issued "swapped" (rows = n, cols = m) so each lane ends up with
//   4 consecutive n for one m: the epilogue packs them, stages a 32-row block
//   through LDS and writes full 16-B row segments (with optional bias, residual
//   add, GEGLU pairing).
#pragma once
#include "prim.h"
#include "optypes.h"
#include <type_traits>

namespace star {

struct GemmParams {
  const void* A;
  const void* W;      // [N][K] (K contiguous)
  void* C;
  const float* bias;  // [N] or null
  const void* res;    // [M][ldr] or null
  const void* zero_page;
  int M, N, K;
  int lda, ldc, ldr;
  // conv geometry (A_CONV*): input [NB][H][Wd][lda>=Cin], output rows m = ((nb*Ho)+yo)*Wo+xo
  int H, Wd, Cin, Ho, Wo, stride, pad_t, pad_l;
  // temporal geometry (A_TCONV3): m = f*HW + p
  int HW, F;
  int up_crop;   // A_CONV3X3_UP: rows dropped at top and bottom of the 2x-upsampled image (1: UNet Upsample, 0: VAE Upsample2D)
  int epi;
  int tiles_m, tiles_n;
  int group_m;   // > 1: the tile walk runs column-major inside groups of group_m tile rows (see the kernel)
  int m_off;     // first output row of this launch (a multiple of the tile height): the launcher may give the last, partly filled round
                 // of big tiles to a second launch of small tiles (gemm_impl.h: tail split); tiles_m counts the rows from m_off on
  // EPI_ROWAFF (LayerNorm folded into this GEMM): out = a_m * acc + b_m * s_n + c_n with (a_m, b_m) = rowab[m], s_n = colsum[n],
  // c_n = bias[n]
  const float* rowab;   // [M][2]
  const float* colsum;  // [N]
  // EPIF bit 4 (GroupNorm statistics in the producer's epilogue, norm.h: gn_finalize_fused_kernel): gn_partial[ceil(M / 32)][N / 2][2] =
  // (sum, sum of squares) of the STORED 16-bit outputs over the 32 rows of a slot, per PAIR of adjacent channels (a GroupNorm group
  // is an even number of channels, so a pair never straddles two groups).  Every (slot, pair) is written exactly once, in a fixed order.
  float* gn_partial;
  // EPIF bit 5 (LayerNorm row statistics in the producer's epilogue, norm.h: ln_from_partials_kernel): ln_partial[M][ln_parts] x
  // (sum, sum of squares, max, 0) fp32 of the STORED 16-bit outputs of row m over the columns of part = tile_n * WN + wave column
  // (ln_parts = tiles_n * WN; every (row, part) is written exactly once)
  float* ln_partial;
  int ln_parts;
  // A_TCONV3: t_walk > 0 (= tile rows per pixel block enumeration is frame-interleaved): consecutive tile rows of the walk are the SAME pixel
  // block of consecutive frames, so the 32 workgroups an XCD runs at a time read each input frame tile three times within one
  // round -- from the L2 -- instead of once per round of ~100 tiles from beyond it.  Set by the launcher when the launch covers
  // whole frames (no tail split); 0 = row-major.
  int t_walk;
};

// exact (erf) GELU, F.gelu default (unet_v2v.py:504): gelu(x) = max(x, 0) - |x| q(|x|), q(t) = 0.5 erfc(t / sqrt 2).
// log2 q is smooth and nearly quadratic, so q = exp2(P7(z)), z = min(|x| / sqrt 2, 4.5), with a degree-7 polynomial fitted on
// Chebyshev nodes of [0, 4.5]: |relative error of q| <= 4e-6 everywhere, i.e. the NEGATIVE TAIL of gelu keeps its relative
// accuracy (6e-6; beyond z = 4.5 |gelu| < 7e-10, below the 16-bit denormals) and the absolute error is <= 6e-7 (fp32
// evaluation, checked against float64 on 4e5 points of [-8, 8]) -- far below the 16-bit output rounding.  One transcendental
// and 11 plain VALU per element instead of the Abramowitz-Stegun 7.1.26 form's rcp + exp2 + ~18 (whose 1 - erf cancellation
// also loses the tail): the GEGLU epilogue of the short-K feed-forward layers is VALU-bound.
STAR_DEV float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float z = fminf(ax * 0.70710678118654752440f, 4.5f);
  float p = -2.045475840e-05f;
  p = p * z + 4.882977128e-04f;
  p = p * z + -5.237886925e-03f;
  p = p * z + 3.395745580e-02f;
  p = p * z + -1.525140382e-01f;
  p = p * z + -9.170034400e-01f;
  p = p * z + -1.628095626e+00f;
  p = p * z + -9.999960965e-01f;
  return fmaxf(x, 0.f) - ax * fast_exp2(p);
}

// the same on a PAIR of values: the degree-7 polynomial as seven v_pk_fma_f32 (packed fp32 is full rate on CDNA3/4: half the
// instructions of two scalar chains; each element sees the same fma sequence as in gelu_erf -- bit-identical).  The GEGLU epilogues
// are VALU-bound (the A-stationary kernel's exceeds what 80 MFMAs shadow; the persistent tile's runs with the matrix pipe idle).
using f32x2 = vec<float, 2>;
STAR_DEV f32x2 gelu_erf2(f32x2 x) {
  f32x2 ax, z;
  ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
  z[0] = fminf(ax[0] * 0.70710678118654752440f, 4.5f); z[1] = fminf(ax[1] * 0.70710678118654752440f, 4.5f);
  f32x2 p = {-2.045475840e-05f, -2.045475840e-05f};
  const float c[7] = {4.882977128e-04f, -5.237886925e-03f, 3.395745580e-02f, -1.525140382e-01f, -9.170034400e-01f, -1.628095626e+00f, -9.999960965e-01f};
#pragma unroll
  for (int i = 0; i < 7; ++i) { const f32x2 cc = {c[i], c[i]}; p = __builtin_elementwise_fma(p, z, cc); }
  f32x2 r;
  r[0] = fmaxf(x[0], 0.f) - ax[0] * fast_exp2(p[0]);
  r[1] = fmaxf(x[1], 0.f) - ax[1] * fast_exp2(p[1]);
  return r;
}
STAR_DEV f32x4 gelu_erf4(f32x4 x) {
  f32x2 a = {x[0], x[1]}, b = {x[2], x[3]};
  a = gelu_erf2(a); b = gelu_erf2(b);
  f32x4 r = {a[0], a[1], b[0], b[1]};
  return r;
}

// tanh-form GELU (sat.mpu.utils.gelu_impl: 0.5 x (1 + tanh(0.79788456 x (1 + 0.044715 x^2)))) = x sigmoid(2 u)
STAR_DEV float gelu_tanh(float x) {
  const float u = 0.7978845608028654f * x * (1.0f + 0.044715f * x * x);
  return x * fast_rcp(1.0f + fast_exp2(-2.8853900817779268f * u));
}

template <class T, int BM, int BN, int WM, int WN, int AMODE, int MINW, bool F32OUT, bool STAGGER, int ABL = 0, int PIPE = 0, int EPIF = 0, int SCHED = 0>  // SCHED: hand-placed 2-stage loop for one wave per SIMD (4 waves x 128 x 128); ABL: ablation probes (bench only); PIPE: ring slots of the pipelined loop (0 = 2-stage loop); EPIF: 16-bit epilogue flavour (bit 0 residual add, bit 1 GEGLU, bit 2 tanh-GELU, bit 3 row-affine = folded LayerNorm, bit 4 GroupNorm partial statistics of the output, bit 5 per-row (LayerNorm) partial statistics of the output)
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(WM * WN * 64, MINW)
gemm_kernel(const GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BK = 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int NA = BM * 8 / NT, NW = BN * 8 / NT;  // 16-B chunks per thread per K tile
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");
  static_assert(NT % 8 == 0, "");
  constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128;
  constexpr int STAGE = A_STAGE + W_STAGE;
  constexpr int SMEM_LOOP = PIPE ? PIPE * (BM + BN) * 64 : 2 * STAGE;   // bytes of the main loop's buffers; bias[BN] (+ colsum[BN]) fp32 follow

  char* smem = dyn_smem();

  // ---- XCD-aware tile order: blocks that run on one XCD (bid % 8) walk consecutive
  // logical tile ids, and consecutive ids share the A row panel (different n tile).
  // Persistent tile walk: the launcher may start fewer workgroups than output tiles (a multiple of 8, so a workgroup's
  // tiles stay on its XCD); workgroup b then computes tiles b, b + gridDim.x, ...  The global stores of one tile drain while
  // the next tile's K loop runs, and there is no workgroup turnover between them.  (The body below is not re-indented.)
  const int nblk = p.tiles_m * p.tiles_n;
  for (int pb = blockIdx.x; pb < nblk; pb += gridDim.x) {
  // the thread id is re-read opaquely per tile: otherwise hipcc hoists every lane-derived address term out of the tile loop
  // and spills 50-120 registers to keep them alive across the whole body
  const int tid = opaque_int((int)threadIdx.x);
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = pb;
  {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  // The 32 workgroups an XCD runs at a time are 32 CONSECUTIVE ids.  Row-major ids make them one row of tiles when tiles_n >= 32:
  // one A panel and 32 W panels per K step (33 x 32 KB unique of the 64 requested: every second byte comes over the fabric, and
  // the loop is paced by that, profiles/r03_lds_dma_bw_probe.txt).  Walking column-major inside groups of group_m tile rows makes
  // them a group_m x (32 / group_m) block: 8 + 4 panels unique, the rest are L2 hits.
  int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  if (p.group_m > 1) {
    const int per_group = p.group_m * p.tiles_n;
    const int g = bid / per_group, in = bid - g * per_group;
    const int rows = p.tiles_m - g * p.group_m < p.group_m ? p.tiles_m - g * p.group_m : p.group_m;
    tile_m = g * p.group_m + in % rows;
    tile_n = in / rows;
  }
  if constexpr (AMODE == A_TCONV3) {
    if (p.t_walk > 0) {
      // tile rows [s_f, s_{f+1}) with s_f = (f * HW) / BM start inside frame f; k = j * F + f enumerates (pixel block j, frame f) for the
      // j every frame has; the frames with one tile row more come last
      const int Fn = p.F, k = tile_m;
      const int nmin = (int)(((long long)p.HW) / BM);          // every frame has nmin or nmin + 1 tile rows
      if (k < nmin * Fn) {
        const int j = k / Fn, f = k - j * Fn;
        tile_m = (int)(((long long)f * p.HW) / BM) + j;
      } else {
        int r = k - nmin * Fn, last = p.tiles_m - 1;
        for (int f = 0; f < Fn && r >= 0; ++f) {
          const int s0 = (int)(((long long)f * p.HW) / BM), s1 = f + 1 < Fn ? (int)(((long long)(f + 1) * p.HW) / BM) : p.tiles_m;
          for (int e = nmin; e < s1 - s0 && r >= 0; ++e) { if (r == 0) last = s0 + e; --r; }
        }
        tile_m = last;
      }
    }
  }
  tile_m += p.m_off / BM;   // (a multiple of BM by construction; added to the tile index so that m0 stays a provable multiple of BM --
                            // as "p.m_off + tile_m * BM" the 256 x 320 and 128 x 320 residual flavours spilled 1-5 registers)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  if constexpr (!F32OUT) {   // this tile's bias slice, read from LDS in the epilogue (visible after the main loop's barriers)
    float* bl = reinterpret_cast<float*>(smem + SMEM_LOOP);
    for (int n = tid; n < BN; n += NT) bl[n] = ((p.epi & EPI_BIAS) && n0 + n < p.N) ? p.bias[n0 + n] : 0.f;
    if constexpr ((EPIF & 8) != 0) {
      for (int n = tid; n < BN; n += NT) bl[BN + n] = (n0 + n < p.N) ? p.colsum[n0 + n] : 0.f;
    }
  }

  const T* __restrict__ Ag = (const T*)p.A;
  const T* __restrict__ Wg = (const T*)p.W;

  // ---- per-thread loader state
  // Plain operands (W always, A in A_PLAIN mode) are staged through the saddr form of the LDS-DMA (prim.h: glds16_su): a
  // wave-uniform tile pointer that advances by 128 B per K tile on the scalar unit + a loop-invariant 32-bit lane offset;
  // the gathered modes keep per-lane 64-bit addresses.  LDS destinations come from a uniform wave id (no v_readfirstlane).
  const int wvu = wave_uniform(wave);
  const int pos = tid & 7;
  uint32_t a_off[NA], w_off[NW];    // byte offsets from this tile's first A row / W row
  const char* a_tile = (const char*)(Ag + (size_t)m0 * p.lda);
  const char* w_tile = (const char*)(Wg + (size_t)n0 * p.K);
  // A rows handled by this thread: r_j = (j*NT + tid) >> 3
  // Gathered A (3x3 conv, temporal conv): LDS-DMA through a buffer descriptor whose base is wave-uniform -- the first input
  // frame this tile touches (conv) or one frame before the tile's first row (temporal conv; the three taps are then
  // non-negative multiples of a frame) -- a loop-invariant signed lane offset g_off to the lane's tap-0 source, a 9- / 3-bit
  // validity mask g_mask (bit = tap) and a uniform per-K-tile tap offset: two VALU adds/selects and one bit test per copy,
  // a padding tap is an out-of-range offset that the hardware turns into zeros (prim.h: glds16_buf).  The launcher checks
  // that the offsets (a few image rows / one row tile) stay below 2^31.  The nearest-x2 conv (A_CONV3X3_UP: 3 layers per net) keeps per-lane 64-bit addresses.
  const T* a_ptr[NA];             // A_CONV3X3_UP: frame base pointer (+ chunk offset)
  int a_y[NA], a_x[NA];           // A_CONV3X3_UP: upsampled-image coordinates of tap (0, 0)
  int g_off[NA];
  uint32_t g_mask[NA];
  // 3x3 conv: the base is the first input row the tile's first pixel reads (frame g_nb0, row g_y0): every valid source of the
  // tile lies at or behind it, within a few image rows (2160p VAE frames are > 2 GB: a frame-relative offset would not fit).
  // Temporal conv: the base moves with the tap (one frame = up to 4 GB at 2160p), the lane offset is tile-relative.
  const char* g_base = (const char*)Ag;
  if constexpr (AMODE == A_TCONV3) g_base = (const char*)(Ag + ((ptrdiff_t)m0 - p.HW) * p.lda);
  int g_nb0 = 0, g_y0 = 0;
  if constexpr (AMODE == A_CONV3X3) {
    const int hw = p.Ho * p.Wo;
    g_nb0 = m0 / hw;
    g_y0 = ((m0 - g_nb0 * hw) / p.Wo) * p.stride - p.pad_t;
    if (g_y0 < 0) g_y0 = 0;
    g_base = (const char*)(Ag + ((size_t)g_nb0 * p.H + g_y0) * p.Wd * p.lda);
  }
  BufRsrc g_rsrc = make_rsrc(g_base, GLDS_BUF_RANGE);
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int r = (j * NT + tid) >> 3;
    const int c = pos ^ ((r >> 1) & 7);
    int m = m0 + r;
    if (m > p.M - 1) m = p.M - 1;
    a_off[j] = (uint32_t)((m - m0) * p.lda + c * 8) * 2u;
    a_ptr[j] = nullptr; a_y[j] = a_x[j] = 0; g_off[j] = 0; g_mask[j] = 0;
    if constexpr (AMODE == A_TCONV3) {
      const int f = m / p.HW;
      g_off[j] = (int)a_off[j];
      g_mask[j] = (f >= 1 ? 1u : 0u) | 2u | (f + 1 < p.F ? 4u : 0u);
    } else if constexpr (AMODE == A_CONV3X3 || AMODE == A_CONV3X3_UP) {
      const int hw = p.Ho * p.Wo;
      const int nb = m / hw, rem = m - nb * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      a_ptr[j] = Ag + (size_t)nb * p.H * p.Wd * p.lda + c * 8;
      a_y[j] = yo * p.stride - p.pad_t;
      a_x[j] = xo * p.stride - p.pad_l;
      if constexpr (AMODE == A_CONV3X3) {
        g_off[j] = ((((nb - g_nb0) * p.H + a_y[j] - g_y0) * p.Wd + a_x[j]) * p.lda + c * 8) * 2;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
          const int yi = a_y[j] + t9 / 3, xi = a_x[j] + t9 % 3;
          if (yi >= 0 && yi < p.H && xi >= 0 && xi < p.Wd) g_mask[j] |= 1u << t9;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int r = (j * NT + tid) >> 3;
    const int c = pos ^ ((r >> 1) & 7);
    int n = n0 + r;
    if (n > p.N - 1) n = p.N - 1;
    w_off[j] = (uint32_t)((n - n0) * p.K + c * 8) * 2u;
  }

  const int nk = p.K / BK;
  int tap = 0, c0 = 0;  // (tap, channel offset) of the K tile being staged

  // One K tile's copies as separately placeable pieces (the scheduled loop puts them between MFMAs): stage_pre() fixes the
  // tile's uniform terms, stage_piece<J>() issues copy J (0..NA-1: A, NA..NA+NW-1: W), stage_post() advances (tap, c0)
  int st_ky = 0, st_kx = 0, st_tap_off = 0;
  uint32_t st_tap_bit = 1;
  auto stage_pre = [&]() STAR_ALWAYS_INLINE {
    if constexpr (AMODE == A_CONV3X3 || AMODE == A_CONV3X3_UP) { st_ky = tap / 3; st_kx = tap - st_ky * 3; }
    // uniform byte offset of this K tile's (tap, channel block) from a lane's tap-0 source
    if constexpr (AMODE == A_TCONV3) { st_tap_off = c0 * 2; g_rsrc = make_rsrc(g_base + (size_t)tap * p.HW * p.lda * 2, GLDS_BUF_RANGE); }
    if constexpr (AMODE == A_CONV3X3) st_tap_off = ((st_ky * p.Wd + st_kx) * p.lda + c0) * 2;
    st_tap_bit = 1u << tap;
  };
  auto stage_piece = [&](int kt, int buf, auto jc) STAR_ALWAYS_INLINE {
    constexpr int J = decltype(jc)::value;
    if constexpr (J < NA) {
      constexpr int j = J;
      // wave-uniform LDS base: chunk q = j*NT + tid -> byte q*16 ; wave base = (j*NT + wave*64)*16
      char* dst = smem + buf * A_STAGE + (size_t)(j * NT + wvu * 64) * 16;
      if constexpr (AMODE == A_PLAIN) {
        glds16_su(a_tile + (size_t)kt * (BK * 2), a_off[j], dst);
      } else if constexpr (AMODE == A_TCONV3 || AMODE == A_CONV3X3) {
        glds16_buf(g_rsrc, (g_mask[j] & st_tap_bit) ? (uint32_t)(g_off[j] + st_tap_off) : GLDS_BUF_OOB, dst);
      } else {  // A_CONV3X3_UP: conv input U[y][x] = X[(y+crop)>>1][x>>1], U is (2H-2*crop) x (2Wd)
        const int yu = a_y[j] + st_ky, xu = a_x[j] + st_kx;
        const void* src = (yu >= 0 && yu < 2 * p.H - 2 * p.up_crop && xu >= 0 && xu < 2 * p.Wd)
                  ? (const void*)(a_ptr[j] + ((size_t)((yu + p.up_crop) >> 1) * p.Wd + (xu >> 1)) * p.lda + c0) : p.zero_page;
        glds16(src, dst);
      }
    } else {
      constexpr int j = J - NA;
      glds16_su(w_tile + (size_t)kt * (BK * 2), w_off[j], smem + 2 * A_STAGE + buf * W_STAGE + (size_t)(j * NT + wvu * 64) * 16);
    }
  };
  auto stage_post = [&]() STAR_ALWAYS_INLINE {
    if constexpr (AMODE != A_PLAIN) {
      // 3x3 convs: the K loop walks the nine TAPS inside a 64-channel block (K index = (c / 64, tap, c % 64), the weights are packed
      // to match): consecutive K tiles re-read the same 128-byte input lines shifted by one pixel / one image row, which the XCD's L2 still
      // holds.  With the channel blocks inside a tap (the order until round 5) a line came back five K tiles x 32 workgroups later: every
      // tap was fetched from beyond the L2, 5.0 GB per level-0 launch against 0.54 GB of input (profiles/r05_pmc_kernels.txt);
      // -7.8 % / -3.8 % / -1.4 % at the 320 / 640 / 1280-wide levels (profiles/r05_cbench_conv_korder.txt).  The temporal conv's taps are
      // whole frames apart: it keeps tap-major order.  Channel-major measured 1-3 % slower with the row-major walk (round 5); with the
      // frame-interleaved walk (t_walk) it cuts the level-0 launch's FETCH_SIZE by 38 % and buys +2.2 % there, 0 at level 1, -2 % at levels
      // 2-3 (profiles/r06_cbench_tconv_korder.txt): 0.06 % of a forward, not worth a second weight layout -- measured and dropped in round 6.
      if constexpr (AMODE == A_TCONV3) {
        c0 += BK;
        if (c0 >= p.Cin) { c0 = 0; ++tap; }
      } else {
        ++tap;
        if (tap >= 9) { tap = 0; c0 += BK; }
      }
    }
  };
  // LDS of the 2-stage loop: [A stage 0 | A stage 1 | W stage 0 | W stage 1] -- with the stage a compile-time constant of the
  // unrolled loop, every fragment read is (loop-invariant register) + (16-bit immediate)
  auto stage = [&](int kt, int buf) {
    stage_pre();
    static_for<NA + NW>([&](auto jc) STAR_ALWAYS_INLINE { stage_piece(kt, buf, jc); });
    stage_post();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the scheduled 4 x 5 tile keeps its fifth block column in architectural registers and multiplies into it with inline-asm MFMAs
  // (prim.h: mfma32_vform), which the compiler's hazard recognizer cannot see: left alone it re-materialises these zeros with v_mov
  // right in front of the first MFMA that reads them -- a VALU write -> MFMA srcC read with no wait states, and the first register
  // arrived late on hardware (the emulator cannot show it).  Pinned here, far from the loop, the zeros are ordinary live values.
  if constexpr (SCHED != 0 && TN == 5) {
#pragma unroll
    for (int i = 0; i < TM; ++i) STAR_VGPR_PIN(acc[i][TN - 1]);
  }

  // fragment read offsets (bytes within a stage); row R, chunk c -> R*128 + ((c ^ ((R>>1)&7))<<4)
  const int frow = lane & 31, fhalf = lane >> 5;

  if constexpr (PIPE != 0) {
    // ---- pipelined main loop: 32-deep K steps through a ring of PIPE LDS slots, PIPE-1 steps of LDS-DMA in flight across
    // raw barriers (counted vmcnt, never 0 in steady state for PIPE > 2).  Ablation of the 2-stage loop at 8192^3 (profiles/
    // r01_gemm_ablation.txt): LDS-DMA + barrier alone 0.90 ms, MFMA + LDS reads alone 0.81 ms, together 1.10 ms -- with only
    // one 64-deep tile of DMA in flight its latency is exposed every tile.
    constexpr int SLOT = (BM + BN) * 64;            // bytes: [BM rows | BN rows] x 32 k
    constexpr int NIA = BM / 16, NIW = BN / 16;      // wave-instructions (64 chunks of 16 B) per step for A / W
    static_assert(NIA % (NT / 64) == 0, "A part must split evenly over the waves");
    constexpr int NWV = NT / 64;
    constexpr int LA = NIA / NWV;                    // A loads per wave per step
    constexpr int LW0 = NIW / NWV, WX = NIW % NWV;   // W loads per wave per step: LW0 (+1 for waves < WX)
    const int wv = wave_uniform(wave);
    const bool wextra = wv < WX;
    // per-wave loader rows: instruction j covers chunks [64 j, 64 j + 64): row = (64 j + lane) >> 2, pos = lane & 3
    const int lpos = lane & 3;
    const T* pa_ptr[LA]; int pa_y[LA], pa_x[LA], pa_f[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = (64 * (wv + NWV * i) + lane) >> 2;
      const int c = lpos ^ ((r >> 2) & 3);
      int m = m0 + r;
      if (m > p.M - 1) m = p.M - 1;
      if constexpr (AMODE == A_PLAIN) { pa_ptr[i] = Ag + (size_t)m * p.lda + c * 8; pa_y[i] = pa_x[i] = pa_f[i] = 0; }
      else if constexpr (AMODE == A_TCONV3) { pa_ptr[i] = Ag + (size_t)m * p.lda + c * 8; pa_f[i] = m / p.HW; pa_y[i] = pa_x[i] = 0; }
      else {
        const int hw = p.Ho * p.Wo;
        const int nb = m / hw, rem = m - nb * hw;
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        pa_ptr[i] = Ag + (size_t)nb * p.H * p.Wd * p.lda + c * 8;
        pa_y[i] = yo * p.stride - p.pad_t; pa_x[i] = xo * p.stride - p.pad_l; pa_f[i] = 0;
      }
    }
    // plain operands through the saddr LDS-DMA (uniform tile pointer + loop-invariant 32-bit lane offset), as in the 2-stage loop
    uint32_t pa_off[LA], pw_off[LW0 + 1];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = (64 * (wv + NWV * i) + lane) >> 2;
      const int c = lpos ^ ((r >> 2) & 3);
      int m = m0 + r;
      if (m > p.M - 1) m = p.M - 1;
      pa_off[i] = (uint32_t)((m - m0) * p.lda + c * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < LW0 + 1; ++i) {
      const int r = (64 * (wv + NWV * i) + lane) >> 2;
      const int c = lpos ^ ((r >> 2) & 3);
      int n = n0 + r;
      if (n > p.N - 1) n = p.N - 1;
      pw_off[i] = (uint32_t)((n - n0) * p.K + c * 8) * 2u;
    }
    const int nsteps = p.K / 32;
    int ptap = 0, pc0 = 0;
    constexpr int INF = PIPE - 2;                    // steps that may stay in flight while step st+1 is awaited
    auto issue = [&](int s_, int slot) {   // LDS-DMA of K step s_ into ring slot `slot` (= s_ % PIPE)
      char* abuf = smem + slot * SLOT;
      char* wbuf = abuf + BM * 64;
      int ky = 0, kx = 0;
      if constexpr (AMODE == A_CONV3X3 || AMODE == A_CONV3X3_UP) { ky = ptap / 3; kx = ptap - ky * 3; }
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        const void* src;
        if constexpr (AMODE == A_PLAIN) { glds16_su(a_tile + (size_t)s_ * 64, pa_off[i], abuf + (size_t)(wv + NWV * i) * 1024); continue; }
        else if constexpr (AMODE == A_TCONV3) {
          const int f = pa_f[i] + ptap - 1;
          src = (f >= 0 && f < p.F) ? (const void*)(pa_ptr[i] + (ptrdiff_t)(ptap - 1) * p.HW * p.lda + pc0) : p.zero_page;
        } else if constexpr (AMODE == A_CONV3X3) {
          const int yi = pa_y[i] + ky, xi = pa_x[i] + kx;
          src = (yi >= 0 && yi < p.H && xi >= 0 && xi < p.Wd) ? (const void*)(pa_ptr[i] + ((size_t)yi * p.Wd + xi) * p.lda + pc0) : p.zero_page;
        } else {
          const int yu = pa_y[i] + ky, xu = pa_x[i] + kx;
          src = (yu >= 0 && yu < 2 * p.H - 2 * p.up_crop && xu >= 0 && xu < 2 * p.Wd)
                    ? (const void*)(pa_ptr[i] + ((size_t)((yu + p.up_crop) >> 1) * p.Wd + (xu >> 1)) * p.lda + pc0) : p.zero_page;
        }
        glds16(src, abuf + (size_t)(wv + NWV * i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < LW0; ++i) glds16_su(w_tile + (size_t)s_ * 64, pw_off[i], wbuf + (size_t)(wv + NWV * i) * 1024);
      if (WX > 0 && wextra) glds16_su(w_tile + (size_t)s_ * 64, pw_off[LW0], wbuf + (size_t)(wv + NWV * LW0) * 1024);
      if constexpr (AMODE == A_TCONV3) { pc0 += 32; if (pc0 >= p.Cin) { pc0 = 0; ++ptap; } }
      else if constexpr (AMODE != A_PLAIN) {   // 3x3 convs: (c / 64, tap, c % 64) -- two 32-deep steps per (block, tap)
        pc0 += 32;
        if ((pc0 & 63) == 0) { pc0 -= 64; ++ptap; if (ptap >= 9) { ptap = 0; pc0 += 64; } }
      }
    };
    // wait until at most `steps_in_flight` K steps of this wave's DMA are outstanding
    auto wait_steps = [&](int steps_in_flight) {
      if constexpr (INF >= 2) {
        if (steps_in_flight >= 2) {
          if (WX > 0 && wextra) { STAR_WAIT_VMCNT_N(2 * (LA + LW0 + 1)); } else { STAR_WAIT_VMCNT_N(2 * (LA + LW0)); }
          return;
        }
      }
      if constexpr (INF >= 1) {
        if (steps_in_flight >= 1) {
          if (WX > 0 && wextra) { STAR_WAIT_VMCNT_N(LA + LW0 + 1); } else { STAR_WAIT_VMCNT_N(LA + LW0); }
          return;
        }
      }
      STAR_WAIT_VMCNT_N(0);
    };
    // prologue: PIPE-1 steps in flight, step 0 landed
    issue(0, 0);
#pragma unroll
    for (int q = 1; q < PIPE - 1; ++q)
      if (nsteps > q) issue(q, q);
    {
      const int ahead = (nsteps - 1 < PIPE - 2) ? nsteps - 1 : PIPE - 2;   // issued steps beyond step 0
      wait_steps(ahead);
    }
    barrier_keep_dma();
    // loop-invariant fragment addresses (slot 0): rows are 64 B, the swizzle ((R >> 2) & 3) depends on frow alone; ring slot and
    // 32-row block go into the ds_read's immediate offset where they fit its 16 bits (the step loop is unrolled PIPE-fold so the slot is a constant)
    const char* pafb[2];
    const char* pwfb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sw = ((u * 2 + fhalf) ^ ((frow >> 2) & 3)) << 4;
      pafb[u] = opaque(smem + (wm * WTM + frow) * 64 + sw);
      pwfb[u] = opaque(smem + BM * 64 + (wn * WTN + frow) * 64 + sw);
    }
    auto pstep = [&](int st, auto rs) STAR_ALWAYS_INLINE {
      constexpr int RS = decltype(rs)::value;             // ring slot read in this step
      constexpr int IS = (RS + PIPE - 1) % PIPE;          // slot to be filled: read in step st-1, every wave is past its barrier
      if (st + PIPE - 1 < nsteps) issue(st + PIPE - 1, IS);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        vec<T, 8> af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const vec<T, 8>*>(pafb[u] + RS * SLOT + i * 2048);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const vec<T, 8>*>(pwfb[u] + RS * SLOT + j * 2048);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<T>(wf[j], af[i], acc[i][j]);
      }
      // step st+1 must have landed before anyone reads it: after this wait at most the DMA of steps st+2.. is pending
      const int remaining = nsteps - 1 - st;         // steps after this one
      wait_steps(remaining - 1 < INF ? remaining - 1 : INF);
      barrier_keep_dma();
    };
    {
      int st = 0;
      for (; st + PIPE <= nsteps; st += PIPE) {          // PIPE steps per trip: the ring slot is a compile-time constant in each
        pstep(st, std::integral_constant<int, 0>{});
        if constexpr (PIPE > 1) pstep(st + 1, std::integral_constant<int, 1 % PIPE>{});
        if constexpr (PIPE > 2) pstep(st + 2, std::integral_constant<int, 2 % PIPE>{});
        if constexpr (PIPE > 3) pstep(st + 3, std::integral_constant<int, 3 % PIPE>{});
      }
      static_assert(PIPE <= 4, "the step loop is unrolled for rings of up to four slots");
      if (st < nsteps) { pstep(st, std::integral_constant<int, 0>{}); ++st; }
      if constexpr (PIPE > 2) { if (st < nsteps) { pstep(st, std::integral_constant<int, 1 % PIPE>{}); ++st; } }
      if constexpr (PIPE > 3) { if (st < nsteps) { pstep(st, std::integral_constant<int, 2 % PIPE>{}); ++st; } }
    }
  } else if constexpr (SCHED != 0) {
  // ---- one wave per SIMD, 128 x 128 per wave (TM = TN = 4, 256 accumulator registers): a third fewer LDS fragment bytes per MFMA
  // than the 8-wave tiles.  Nothing hides a stall of the one wave, so every LDS read and every LDS-DMA has a fixed place between
  // the MFMAs of the phase BEFORE the one that needs it (fences keep hipcc from moving them).  K tile kt (64 k, stage kt & 1) =
  // four phases of 16 MFMAs (one 16-k step each, fragments double-buffered); the 8 fragment reads of the next phase sit in the
  // first 8 MFMA gaps.  One barrier per tile, behind phase 2: every wave has read stage kt & 1 out (phase 3's fragments are in
  // registers) and tile kt+1 has landed.  Tile kt+2's copies go into the stage just read out: the A pieces in the last 8 gaps of
  // phase 3, the W pieces in the last 8 gaps of the next tile's phase 0 -- 1.5 to 2 tile times before they are waited for.
  // The pieces are whole 128-byte lines (8 rows per wave-instruction), as in the 8-wave loop: the 32-deep ring slots of the
  // pipelined loop above ask the L2 for every line twice (TCP_TCC_READ_REQ 134 M against 67 M at 8192^3, profiles/r03_gemm_sched_pmc.txt).
  // Round 6: the same placement for any TM x TN with TM + TN reads <= the even gaps and NA / NW copies <= the odd gaps of a phase: 4 x 5
  // blocks = 4 waves x (128 x 160), the 256 x 320 tile with one wave per SIMD (320 accumulators, 9 fragment reads per 20 MFMAs).
  constexpr int NM = TM * TN;                       // MFMAs per phase
  static_assert(PIPE == 0 && NT == 256 && !STAGGER && TM == 4 && (TN == 4 || TN == 5), "the scheduled loop is built for 4 waves, 128 rows per wave");
  static_assert(2 * (TM + TN) <= NM + 1 && 2 * NA <= NM && 2 * NW <= NM, "a phase's reads go into its even gaps, its copies into the odd ones");
  const char* afb[4];
  const char* wfb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((ks * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4;
    afb[ks] = opaque(smem + (wm * WTM + frow) * 128 + sw);
    wfb[ks] = opaque(smem + 2 * A_STAGE + (wn * WTN + frow) * 128 + sw);
  }
  vec<T, 8> fa[2][TM], fw[2][TN];
  auto frag_read = [&](auto cks, int sa, int swo, auto ci, auto cb) STAR_ALWAYS_INLINE {   // read #ci (0..TM-1: A blocks, TM..: W blocks) of 16-k step cks into buffer cb
    constexpr int KS = decltype(cks)::value, I = decltype(ci)::value, Bf = decltype(cb)::value;
    if constexpr (I < TM) fa[Bf][I] = *reinterpret_cast<const vec<T, 8>*>(afb[KS] + sa + I * 4096);
    else fw[Bf][I - TM] = *reinterpret_cast<const vec<T, 8>*>(wfb[KS] + swo + (I - TM) * 4096);
  };
  auto phase = [&](auto cb, auto body) STAR_ALWAYS_INLINE {
    constexpr int Bf = decltype(cb)::value;
    static_for<NM>([&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      constexpr int i = Q / TN, j = (i & 1) ? TN - 1 - (Q % TN) : (Q % TN);   // serpentine: consecutive MFMAs share one operand
      // 4 x 5 blocks are 320 accumulators: the first four block columns live in the 256 accumulation registers, the fifth in 64
      // architectural ones -- said explicitly, or hipcc rotates accumulators through v_accvgpr_read / _write (672 per K tile) and scratch
      if constexpr (TN == 5 && j == 4) mfma32_vform<T>(fw[Bf][j], fa[Bf][i], acc[i][j]);
      else {
        acc[i][j] = mfma32<T>(fw[Bf][j], fa[Bf][i], acc[i][j]);
        if constexpr (TN == 5) STAR_AGPR_PIN(acc[i][j]);
      }
      body(q);
      STAR_SCHED_FENCE();
    });
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  stage(0, 0);
  if (nk > 1) { stage(1, 1); STAR_WAIT_VMCNT_N(NA + NW); } else { STAR_WAIT_VMCNT(0); }
  barrier_keep_dma();
  static_for<TM + TN>([&](auto c) STAR_ALWAYS_INLINE { frag_read(B0{}, 0, 0, c, B0{}); });
  STAR_SCHED_FENCE();
  int sa = 0, swo = 0;                    // byte offsets of the stage tile kt is read from
  // STEADY: both copy groups are due; otherwise (first tile, last two) they are runtime-conditional
  // memory operations of a phase: the fragment reads in the even MFMA gaps, the copies (phases 3 and 0) in the odd ones.  Read
  // order W0 A0 W1 .. W(TN-1) A1 .. A(TM-1) = the order the next phase's MFMAs first need them.
  auto rd = [&](auto cks, auto q, auto cb) STAR_ALWAYS_INLINE {
    constexpr int Q = decltype(q)::value;
    if constexpr ((Q & 1) == 0 && (Q >> 1) < TM + TN) {
      constexpr int R = Q >> 1;
      constexpr int I = R == 0 ? TM : R == 1 ? 0 : R <= TN ? TM + R - 1 : R - TN;
      frag_read(cks, sa, swo, std::integral_constant<int, I>{}, cb);
    }
  };
  auto tile = [&](int kt, auto steady) STAR_ALWAYS_INLINE {
    constexpr bool STEADY = decltype(steady)::value;
    const bool iw = STEADY || (kt >= 1 && kt + 1 < nk), ia = STEADY || kt + 2 < nk;
    const int buf = kt & 1;
    phase(B0{}, [&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      rd(B1{}, q, B1{});
      if constexpr ((Q & 1) == 1 && (Q >> 1) < NW && ABL != 8 && ABL != 11) { if (iw) stage_piece(ABL == 9 ? 0 : kt + 1, buf ^ 1, std::integral_constant<int, (ABL == 9 ? NA : NA + (Q >> 1))>{}); }   // W pieces of tile kt+1
      if constexpr (Q == NM - 1 && ABL != 8) { if (iw) stage_post(); }
    });
    phase(B1{}, [&](auto q) STAR_ALWAYS_INLINE { rd(std::integral_constant<int, 2>{}, q, B0{}); });
    phase(B0{}, [&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      rd(std::integral_constant<int, 3>{}, q, B1{});
      if constexpr (Q == NM - 2 && ABL != 10) { STAR_WAIT_VMCNT(0); }
      if constexpr (Q == NM - 1) barrier_keep_dma();
    });
    sa ^= A_STAGE; swo ^= W_STAGE;
    phase(B1{}, [&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      if constexpr (Q == 0 && ABL != 8) { if (ia) stage_pre(); }
      rd(B0{}, q, B0{});
      if constexpr ((Q & 1) == 1 && (Q >> 1) < NA && ABL != 8) { if (ia) stage_piece(ABL == 9 ? 0 : kt + 2, buf, std::integral_constant<int, (ABL == 9 ? 0 : (Q >> 1))>{}); }   // A pieces of tile kt+2
    });
  };
  {
    int kt = 0;
    if (nk > 0) { tile(0, std::false_type{}); kt = 1; }
    for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{});
    for (; kt < nk; ++kt) tile(kt, std::false_type{});
  }
  } else {
  // Staggered wave groups.  The two waves that share a SIMD (w and w+4) belong to this same workgroup and meet at the
  // same barrier every K tile; left alone they run in lockstep -- both issue the next tile's LDS-DMA + address VALU, both
  // wait on LDS, both then fight for the matrix pipe -- and nothing overlaps.  Waves 4-7 ("late") therefore run one
  // k-step behind: they carry the fragments of their last k-step across the barrier and issue that MFMA burst first,
  // beside the early group's load/address/LDS-wait phase; from then on the two groups alternate.
  const bool late = STAGGER && (wave_uniform(wave) >= NT / 128);
  vec<T, 8> af_hold[TM], wf_hold[TN];
  auto mfma_step = [&](const vec<T, 8> (&a)[TM], const vec<T, 8> (&w)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<T>(w[j], a[i], acc[i][j]);
  };
  // fragment addresses: row R = (wave tile row) + 32 i + frow, 16-B chunk (2 ks + fhalf) ^ ((R >> 1) & 7).  The wave-tile
  // origins and the 32-row blocks are multiples of 16 rows, so the swizzle term depends on frow alone: one loop-invariant
  // address per k-step and operand; stage and block go into the ds_read's immediate offset
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tiles must keep the row swizzle period");
  const char* afb[4];
  const char* wfb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((ks * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4;
    afb[ks] = opaque(smem + (wm * WTM + frow) * 128 + sw);
    wfb[ks] = opaque(smem + 2 * A_STAGE + (wn * WTN + frow) * 128 + sw);
  }
  if constexpr (ABL != 4) stage(0, 0);
  auto ktile = [&](int kt, auto stg) STAR_ALWAYS_INLINE {
    constexpr int S = decltype(stg)::value;
    glds_wait();
    block_sync();
    if (kt + 1 < nk && (ABL != 3 || kt == 0)) stage(kt + 1, S ^ 1);
    if (late && kt > 0) mfma_step(af_hold, wf_hold);   // k-step 3 of the previous tile
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      vec<T, 8> af[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (ABL == 1) { for (int e = 0; e < 8; ++e) af[i][e] = from_f32<T>(0.001f * (float)(lane + i + ks)); }
        else af[i] = *reinterpret_cast<const vec<T, 8>*>(afb[ks] + S * A_STAGE + i * 4096);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (ABL == 1) { for (int e = 0; e < 8; ++e) wf[j][e] = from_f32<T>(0.002f * (float)(lane + j + kt)); }
        else wf[j] = *reinterpret_cast<const vec<T, 8>*>(wfb[ks] + S * W_STAGE + j * 4096);
      }
      if constexpr (ABL == 2) {   // keep the reads alive without MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][0][0] += to_f32<T>(af[i][0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[0][j][1] += to_f32<T>(wf[j][0]);
        continue;
      }
      if (ks == 3 && late) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af_hold[i] = af[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) wf_hold[j] = wf[j];
      } else {
        mfma_step(af, wf);
      }
    }
  };
  {
    const int nkk = (ABL == 4) ? 0 : nk;
    int kt = 0;
    for (; kt + 1 < nkk; kt += 2) {   // two K tiles per trip: the LDS stage is a compile-time constant in each half
      ktile(kt, std::integral_constant<int, 0>{});
      ktile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < nkk) ktile(kt, std::integral_constant<int, 0>{});
  }
  if (late) mfma_step(af_hold, wf_hold);

  }

  // ------------------------------------------------------------------ epilogue
  // lane holds, for m = i*32 + (lane&31): n = j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3).  N % 8 == 0 (T out) / N % 4 == 0
  // (fp32 out) is checked by the launcher, so every 4-/8-column group is either fully inside or fully outside.
  if constexpr (F32OUT) {
    // exact fp32 results straight from the accumulators (attention logits of the VAE, final latent prediction):
    // one 16-B store of 4 consecutive n per lane (fp32 tiles do not fit the LDS staging budget)
    const float* __restrict__ biasf = p.bias;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WTM + i * 32 + frow;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * WTN + j * 32 + 8 * g + 4 * fhalf;
          if (m < p.M && n < p.N) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][g * 4 + e];
            if (p.epi & EPI_BIAS) { const f32x4 b = *reinterpret_cast<const f32x4*>(biasf + n); o += b; }
            if (p.epi & EPI_RES) {
              const vec<T, 4> r = *reinterpret_cast<const vec<T, 4>*>((const T*)p.res + (size_t)m * p.ldr + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] += to_f32<T>(r[e]);
            }
            *reinterpret_cast<f32x4*>((float*)p.C + (size_t)m * p.ldc + n) = o;
          }
        }
    }
  } else {
    // The output of the K = 320 / 640 layers is what bounds them (C is 2-8x the bytes of A), so the epilogue is built for
    // the write stream: no global load is ever issued behind a store (vmcnt retires in order, a load behind stores waits
    // for their HBM write latency) -- the bias comes from LDS, the residual chunks of a 32-row block are all read before
    // the first store of the previous block is issued -- and the barriers between the staging steps do not drain vmcnt.
    // Residual / GEGLU are compile-time (EPIF: bit 0 residual, bit 1 GEGLU): every index below folds to constants, the
    // residual registers exist only where a residual is added, and the residual loads are unconditional (rows and columns
    // clamped into the matrix instead of exec-masked), so nothing of the epilogue stays live across the K loop.
    constexpr bool RESF = (EPIF & 1) != 0, GEGLUF = (EPIF & 2) != 0, GELUTF = (EPIF & 4) != 0, ROWAFF = (EPIF & 8) != 0;
    constexpr int out_wtn = GEGLUF ? WTN / 2 : WTN;   // output columns per wave
    const int out_n0 = GEGLUF ? (n0 + wn * WTN) / 2 : (n0 + wn * WTN);
    const int N_out = GEGLUF ? p.N / 2 : p.N;
    constexpr int pitch = WTN * 2 + 8;                // bytes; (pitch/4) % 4 == 2 -> conflict-free b64 writes
    constexpr int cpr = out_wtn / 8;                  // 16-B chunks per row
    constexpr int nchunks = 32 * cpr;
    static_assert(nchunks % 64 == 0, "a 32-row block must split evenly over the lanes");
    constexpr int RUN = nchunks / 64;                 // 16-B chunks per lane and 32-row block
    // STATS: every lane keeps ONE chunk column cc for the whole block, so that its eight (sum, sum of squares) accumulators are per
    // channel pair: a pass covers RPP = 64 / cpr whole rows (cpr = 20: 3 rows on 60 lanes, 11 passes instead of 10; for the
    // power-of-two widths this IS the plain mapping)
    constexpr bool STATS = (EPIF & 16) != 0;
    static_assert(!STATS || !(GEGLUF || GELUTF || ROWAFF), "GroupNorm statistics: plain and residual epilogues only");
    constexpr int RPP = 64 / cpr;
    constexpr int NPASS = STATS ? (32 + RPP - 1) / RPP : RUN;
    // ROWST: per-row (sum, sum of squares, max) of the stored outputs over this wave's columns, for the LayerNorm that reads the tensor next
    constexpr bool ROWST = (EPIF & 32) != 0;
    static_assert(!ROWST || !(GEGLUF || GELUTF || ROWAFF || STATS), "row statistics: plain and residual epilogues only");
    static_assert(!ROWST || (cpr % 2 == 0), "row statistics: two lanes share a row");
    // chunk u of this lane in a 32-row block: (row, chunk column); false = the lane has no chunk in this pass (STATS mapping only)
    auto chunk_of = [&](int u, int& row, int& cc) STAR_ALWAYS_INLINE -> bool {
      if constexpr (STATS) {
        const int r0 = lane / cpr;
        cc = lane - r0 * cpr;
        row = u * RPP + r0;
        const bool act = r0 < RPP && row < 32;
        if (row > 31) row = 31;
        return act;
      } else {
        const int q = lane + 64 * u;
        row = q / cpr; cc = q - row * cpr;
        return true;
      }
    };
    const float* bias_lds = reinterpret_cast<const float*>(smem + SMEM_LOOP) + wn * WTN;
    block_sync();                                     // all MFMA reads of the stages are done
    char* my = smem + wave * (32 * pitch);

    vec<T, 8> rv[RESF ? TM : 1][RESF ? NPASS : 1];
    auto load_res = [&](int i) STAR_ALWAYS_INLINE {
      if constexpr (RESF) {
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
          int row, cc;
          (void)chunk_of(u, row, cc);
          int m = m0 + wm * WTM + i * 32 + row, n = out_n0 + cc * 8;
          if (m > p.M - 1) m = p.M - 1;
          if (n > N_out - 8) n = N_out - 8;
          rv[i][u] = *reinterpret_cast<const vec<T, 8>*>((const T*)p.res + (size_t)m * p.ldr + n);
        }
      }
    };
    auto store_block = [&](int i) STAR_ALWAYS_INLINE {
      float ps[4], pq[4];   // STATS: this lane's chunk column, 4 channel pairs: sum / sum of squares over its rows of the block
#pragma unroll
      for (int e = 0; e < 4; ++e) { ps[e] = 0.f; pq[e] = 0.f; }
#pragma unroll
      for (int u = 0; u < NPASS; ++u) {
        int row, cc;
        const bool act = chunk_of(u, row, cc);
        const int m = m0 + wm * WTM + i * 32 + row, n = out_n0 + cc * 8;
        const vec<T, 4> lo = *reinterpret_cast<const vec<T, 4>*>(my + row * pitch + cc * 16);
        const vec<T, 4> hi = *reinterpret_cast<const vec<T, 4>*>(my + row * pitch + cc * 16 + 8);
        vec<T, 8> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ov[e] = lo[e]; ov[4 + e] = hi[e]; }
        if constexpr (RESF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(to_f32<T>(ov[e]) + to_f32<T>(rv[i][u][e]));
        }
        if constexpr (ROWST && RESF) {   // the block in LDS becomes the STORED values (the row phase below reads it back row by row)
          *reinterpret_cast<vec<T, 4>*>(my + row * pitch + cc * 16) = vec<T, 4>{ov[0], ov[1], ov[2], ov[3]};
          *reinterpret_cast<vec<T, 4>*>(my + row * pitch + cc * 16 + 8) = vec<T, 4>{ov[4], ov[5], ov[6], ov[7]};
        }
        if (act && m < p.M && n < N_out && (ABL != 5 || p.M < 0)) {
          // (non-temporal stores here are neutral, +-0.5 % on every level-0 shape: profiles/r05_cbench_gemm_nt.txt -- unlike the
          // A-stationary kernel, whose W panel lives in the L2 for the whole launch)
          *reinterpret_cast<vec<T, 8>*>((T*)p.C + (size_t)m * p.ldc + n) = ov;
          if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              vec<T, 2> pr;
              pr[0] = ov[2 * e]; pr[1] = ov[2 * e + 1];
              ps[e] = dot2_one<T>(pr, ps[e]);
              pq[e] = dot2_acc<T>(pr, pr, pq[e]);
            }
          }
        }
      }
      if constexpr (ROWST) {
        // lane (r, hf) sums half of row r's chunks (fixed order), the two halves meet through a lane swap; one 16-byte record per row
        wave_lds_fence();
        const int r = lane & 31, hf = lane >> 5;
        float rs = 0.f, rq = 0.f, rm = -3.0e38f;
#pragma unroll
        for (int c = 0; c < cpr / 2; ++c) {
          const int cc = hf * (cpr / 2) + c;
          if (out_n0 + cc * 8 < N_out) {   // wave-uniform per (hf, c) up to hf: both branches are cheap
            const vec<T, 4> lo = *reinterpret_cast<const vec<T, 4>*>(my + r * pitch + cc * 16);
            const vec<T, 4> hi = *reinterpret_cast<const vec<T, 4>*>(my + r * pitch + cc * 16 + 8);
            vec<T, 8> v8;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = lo[e]; v8[4 + e] = hi[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              vec<T, 2> pr;
              pr[0] = v8[2 * e]; pr[1] = v8[2 * e + 1];
              rs = dot2_one<T>(pr, rs);
              rq = dot2_acc<T>(pr, pr, rq);
            }
            rm = fmaxf(rm, max8<T>(v8));
          }
        }
        const float os = shfl_xor(rs, 32), oq = shfl_xor(rq, 32), om = shfl_xor(rm, 32);
        const int mrow = m0 + wm * WTM + i * 32 + r;
        if (hf == 0 && mrow < p.M) {
          f32x4 rec;
          rec[0] = rs + os; rec[1] = rq + oq; rec[2] = fmaxf(rm, om); rec[3] = 0.f;
          *reinterpret_cast<f32x4*>(p.ln_partial + ((size_t)mrow * p.ln_parts + (size_t)(tile_n * WN + wn)) * 4) = rec;
        }
      }
      if constexpr (STATS) {
        // the RPP lanes of a chunk column (lane, lane + cpr, ...) -> the first of them, in a fixed order; one slot row per 32-row block
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr ((cpr & (cpr - 1)) == 0) {
#pragma unroll
            for (int mk = cpr; mk < 64; mk <<= 1) { ps[e] += shfl_xor(ps[e], mk); pq[e] += shfl_xor(pq[e], mk); }
          } else {
            float s0 = ps[e], q0 = pq[e];
#pragma unroll
            for (int k = 1; k < RPP; ++k) { s0 += shfl(ps[e], (lane + k * cpr) & 63); q0 += shfl(pq[e], (lane + k * cpr) & 63); }
            ps[e] = s0; pq[e] = q0;
          }
        }
        const int row0 = m0 + wm * WTM + i * 32, n = out_n0 + lane * 8;
        if (lane < cpr && row0 < p.M && n < N_out) {
          float* dst = p.gn_partial + (size_t)(row0 >> 5) * p.N + n;
          f32x4 a, b;
          a[0] = ps[0]; a[1] = pq[0]; a[2] = ps[1]; a[3] = pq[1];
          b[0] = ps[2]; b[1] = pq[2]; b[2] = ps[3]; b[3] = pq[3];
          *reinterpret_cast<f32x4*>(dst) = a;
          *reinterpret_cast<f32x4*>(dst + 4) = b;
        }
      }
    };

    // folded LayerNorm: this lane's rows (the swapped MFMA gives a lane one row per 32-row block) and their (a, b)
    float ra[TM], rb[TM];
    if constexpr (ROWAFF) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int m = m0 + wm * WTM + i * 32 + frow;
        if (m > p.M - 1) m = p.M - 1;
        const vec<float, 2> ab = *reinterpret_cast<const vec<float, 2>*>(p.rowab + 2 * (size_t)m);
        ra[i] = ab[0]; rb[i] = ab[1];
      }
    }
    if constexpr (SCHED == 0) load_res(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // ---- registers -> LDS (T, row-major [32][out_wtn])
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (GEGLUF && (j & 1)) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * fhalf;  // local n within the wave tile (pre-GEGLU)
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e];
          if constexpr (ROWAFF) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(bias_lds + BN + nl), cb = *reinterpret_cast<const f32x4*>(bias_lds + nl);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ra[i] * v[e] + (rb[i] * cs[e] + cb[e]);
          } else
          v += *reinterpret_cast<const f32x4*>(bias_lds + nl);   // zeros when the layer has no bias
          int ncol = nl;
          if constexpr (GEGLUF) {
            f32x4 gt;
#pragma unroll
            for (int e = 0; e < 4; ++e) gt[e] = acc[i][(j + 1 < TN) ? j + 1 : j][g * 4 + e];
            if constexpr (ROWAFF) {
              const f32x4 cs = *reinterpret_cast<const f32x4*>(bias_lds + BN + nl + 32), cb = *reinterpret_cast<const f32x4*>(bias_lds + nl + 32);
#pragma unroll
              for (int e = 0; e < 4; ++e) gt[e] = ra[i] * gt[e] + (rb[i] * cs[e] + cb[e]);
            } else
            gt += *reinterpret_cast<const f32x4*>(bias_lds + nl + 32);
            v = v * gelu_erf4(gt);
            ncol = (j >> 1) * 32 + 8 * g + 4 * fhalf;
          }
          if constexpr (GELUTF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
          }
          vec<T, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
          *reinterpret_cast<vec<T, 4>*>(my + frow * pitch + ncol * 2) = o;
        }
      }
      barrier_keep_dma();
      // the next block's residual, ahead of this block's stores (the one-wave-per-SIMD tile has no registers for two blocks of it:
      // it loads each block's residual just before use)
      if constexpr (SCHED == 0) { if (i + 1 < TM) load_res(i + 1); } else load_res(i);
      // ---- LDS -> global: whole 16-B chunks along rows (+ residual, already in registers)
      store_block(i);
      barrier_keep_dma();
    }
  }
  if constexpr (F32OUT) block_sync();   // the next tile's LDS-DMA must not overtake this tile's last fragment reads
  if constexpr (SCHED != 0) break;      // the one-wave-per-SIMD tile is launched one workgroup per output tile: a visibly single trip keeps the workitem id out of scratch
  }   // persistent tile walk
}

}  // namespace star
