// gemm_p.h -- the PERSISTENT one-wave-per-SIMD GEMM ("tile 18") for the plain-A layers with K >= 512: fused q | k | v and
// out-projections of levels 1-2, FF-out, proj_in / proj_out (reference call sites unet_v2v.py:151-155,274,294,526) and the DiT
// dense layers.  Same arithmetic as gemm_kernel (gemm.h): 32x32x16 MFMA, k-steps of an output accumulated in the same order,
// the same epilogue expressions -- bit-identical results.
//
// What it changes against tile 17 (gemm.h SCHED: 4 waves x (128 x 128), hand-placed 2-stage loop, 1.56 us per 256 x 256 x 64
// K tile against 1.72 of the 8-wave tiles) is everything AROUND the K loop, which cost 8 us per output tile there (5 us in the
// 8-wave tiles: 13-23 % of every K = 640-1280 layer, profiles/r03_gemm_sched_ab.txt):
//   * a workgroup is resident for the whole launch and walks its output tiles; the K tiles of ALL its output tiles form ONE
//     stream: the LDS-DMA of the next output tile's first two K tiles is issued from inside the current tile's last two
//     (same places between the MFMAs as in steady state), so there is no cold prologue and no workgroup turnover per tile;
//   * operands are staged through hand-built buffer descriptors (prim.h: glds16_desc, invisible to hipcc's vmcnt bookkeeping):
//     the lane offsets are invariant for the whole kernel, a tile switch is a handful of scalar instructions, rows / columns past
//     the matrix edge read as zeros (no clamps);
//   * the first K tile of an output tile starts its accumulators with zero-C MFMAs (no 256 v_mov);
//   * the epilogue is WAVE-PRIVATE: a wave converts its own 128 x 128 block in 32 x 64 units through a private, XOR-swizzled
//     4 KB LDS block and stores whole 128-byte lines -- no workgroup barrier, no shared staging area under the K loop's stages;
//     the waves meet again at the next tile's first barrier, and the next tile's operands land meanwhile.  (The four waves of a
//     workgroup reach their epilogues together, so the matrix pipe idles for it: 2.8 us per output tile plain, 4.9 with the folded
//     LayerNorm, ~6 with GEGLU -- the epilogue is VALU-bound, 961 to 2600 instructions)
//     Bias / column-sum slices arrive by 4-byte LDS-DMA into a slot per tile parity.
// LDS: [A stage 0 | A stage 1 | W stage 0 | W stage 1] 128 KB, 4 staging blocks 16 KB, 2 x (bias | colsum) 4 KB = 148 KB.
#pragma once
#include "gemm.h"

namespace star {

// STPOL: cache policy of the output stores (prim.h: buf_store16_pol; 2 = non-temporal, the product since round 6; 0 = plain, the round-5 form, bench build only)
template <class T, int EPIF, int STPOL = 0>   // EPIF: bit 0 residual add, bit 1 GEGLU (weight rows in 32-row (value, gate) blocks), bit 3 row-affine (folded LayerNorm); 16-bit output
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 1)
gemm_persist_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, NT = 256, NP = 8;      // NP: 16-byte pieces per thread and operand and K tile
  constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128;
  constexpr int STG_OFF = 2 * A_STAGE + 2 * W_STAGE, STG = 32 * 128;
  constexpr int BIAS_OFF = STG_OFF + 4 * STG;              // 2 parities x [bias 256 | colsum 256] fp32
  constexpr bool RESF = (EPIF & 1) != 0, GEGLUF = (EPIF & 2) != 0, ROWAFF = (EPIF & 8) != 0;
  static_assert(!(RESF && GEGLUF), "GEGLU layers have no residual");
  char* smem = dyn_smem();
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wv = wave_uniform(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int nk = p.K / 64;
  const uint32_t k_bytes = (uint32_t)p.K * 2u;
  const int nblk = p.tiles_m * p.tiles_n;
  const int G = (int)gridDim.x;
  const int ntm = (nblk - (int)blockIdx.x + G - 1) / G;    // output tiles of this workgroup: blockIdx.x + i * G
  if (ntm <= 0) return;

  // ---- tile walk: the XCD-aware order and the grouped walk of gemm_kernel (gemm.h), by sequence index i of this workgroup
  auto tile_origin = [&](int i, int& m0, int& n0) STAR_ALWAYS_INLINE {
    int bid = (int)blockIdx.x + i * G;
    {
      const int q = nblk >> 3, r = nblk & 7;
      const int xcd = bid & 7, slot = bid >> 3;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    if (p.group_m < -1) {
      // round-6 A/B: ROW-major inside strips of -group_m tile COLUMNS: the 32 tiles an XCD runs together are (32 / strip) rows x strip
      // columns, and the following tiles keep the strip's W panels (hot in the L2) while the A panels stream past once per strip
      const int gn = -p.group_m, per_group = gn * p.tiles_m;
      const int g = bid / per_group, in = bid - g * per_group;
      const int cols = p.tiles_n - g * gn < gn ? p.tiles_n - g * gn : gn;
      tile_n = g * gn + in % cols;
      tile_m = in / cols;
    } else
    if (p.group_m > 1) {
      const int per_group = p.group_m * p.tiles_n;
      const int g = bid / per_group, in = bid - g * per_group;
      const int rows = p.tiles_m - g * p.group_m < p.group_m ? p.tiles_m - g * p.group_m : p.group_m;
      tile_m = g * p.group_m + in % rows;
      tile_n = in / rows;
    }
    m0 = p.m_off + tile_m * BM;
    n0 = tile_n * BN;
  };
  auto a_desc = [&](int m0) STAR_ALWAYS_INLINE {
    const int rows = p.M - m0 < BM ? p.M - m0 : BM;
    return make_desc((const char*)p.A + (size_t)m0 * p.lda * 2, (uint32_t)rows * (uint32_t)p.lda * 2u);
  };
  auto w_desc = [&](int n0) STAR_ALWAYS_INLINE {
    const int rows = p.N - n0 < BN ? p.N - n0 : BN;
    return make_desc((const char*)p.W + (size_t)n0 * p.K * 2, (uint32_t)rows * k_bytes);
  };

  // ---- loader: thread (j, tid) copies 16-byte chunk (tid & 7) ^ swizzle of tile row j * 32 + (tid >> 3); the lane offsets are
  // invariant for the whole kernel
  uint32_t a_off[NP], w_off[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int r = j * 32 + (tid >> 3);
    const int c = (tid & 7) ^ ((r >> 1) & 7);
    a_off[j] = (uint32_t)(r * p.lda + c * 8) * 2u;
    w_off[j] = (uint32_t)(r * p.K + c * 8) * 2u;
  }
  // the two DMA streams: A runs two K tiles ahead of the MFMAs, W one (each keeps its own output tile and K offset)
  int sa_i = 0, sw_i = 0;                  // sequence index of the output tile the stream is in
  uint32_t sa_k = 0, sw_k = 0;             // byte offset of the stream's K tile within a row
  BufDesc dA, dW;
  {
    int m0, n0;
    tile_origin(0, m0, n0);
    dA = a_desc(m0); dW = w_desc(n0);
  }
  bool sa_live = true, sw_live = true;     // false once the stream has run past this workgroup's last K tile
  auto a_piece = [&](int buf, auto jc) STAR_ALWAYS_INLINE {
    constexpr int j = decltype(jc)::value;
    glds16_desc(dA, a_off[j], sa_k, smem + buf * A_STAGE + (size_t)(j * NT + wv * 64) * 16);
  };
  auto w_piece = [&](int buf, auto jc) STAR_ALWAYS_INLINE {
    constexpr int j = decltype(jc)::value;
    glds16_desc(dW, w_off[j], sw_k, smem + 2 * A_STAGE + buf * W_STAGE + (size_t)(j * NT + wv * 64) * 16);
  };
  auto a_advance = [&]() STAR_ALWAYS_INLINE {
    sa_k += 128;
    if (sa_k >= k_bytes) {
      sa_k = 0; ++sa_i;
      if (sa_i < ntm) { int m0, n0; tile_origin(sa_i, m0, n0); dA = a_desc(m0); } else sa_live = false;
    }
  };
  auto w_advance = [&]() STAR_ALWAYS_INLINE {
    sw_k += 128;
    if (sw_k >= k_bytes) {
      sw_k = 0; ++sw_i;
      if (sw_i < ntm) { int m0, n0; tile_origin(sw_i, m0, n0); dW = w_desc(n0); } else sw_live = false;
    }
  };
  // this tile's bias (and column-sum) slice: 4 bytes per lane, wave wv brings columns [64 wv, 64 wv + 64)
  auto bias_pieces = [&](int n0, int par) STAR_ALWAYS_INLINE {
    const int cols = p.N - n0 < BN ? p.N - n0 : BN;
    char* slot = smem + BIAS_OFF + par * 2048 + wv * 256;
    const BufDesc db = make_desc((p.epi & EPI_BIAS) ? (const void*)(p.bias + n0) : (const void*)p.W, (p.epi & EPI_BIAS) ? (uint32_t)cols * 4u : 0u);
    glds4_desc(db, (uint32_t)(wv * 64 + lane) * 4u, 0u, slot);
    if constexpr (ROWAFF) {
      const BufDesc dc = make_desc(p.colsum + n0, (uint32_t)cols * 4u);
      glds4_desc(dc, (uint32_t)(wv * 64 + lane) * 4u, 0u, slot + 1024);
    }
  };

  // ---- fragments (gemm.h SCHED): row R = wave tile row + 32 i + frow, 16-byte chunk (2 ks + fhalf) ^ ((R >> 1) & 7)
  const char* afb[4];
  const char* wfb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((ks * 2 + fhalf) ^ ((frow >> 1) & 7)) << 4;
    afb[ks] = opaque(smem + (wm * 128 + frow) * 128 + sw);
    wfb[ks] = opaque(smem + 2 * A_STAGE + (wn * 128 + frow) * 128 + sw);
  }
  f32x16 acc[4][4];
  vec<T, 8> fa[2][4], fw[2][4];
  int sa = 0, swo = 0;   // byte offsets of the stage the current K tile is read from
  auto frag_read = [&](auto cks, auto ci, auto cb) STAR_ALWAYS_INLINE {   // read #ci (0-3: A blocks, 4-7: W blocks) of 16-k step cks into buffer cb
    constexpr int KS = decltype(cks)::value, I = decltype(ci)::value, Bf = decltype(cb)::value;
    if constexpr (I < 4) fa[Bf][I] = *reinterpret_cast<const vec<T, 8>*>(afb[KS] + sa + I * 4096);
    else fw[Bf][I - 4] = *reinterpret_cast<const vec<T, 8>*>(wfb[KS] + swo + (I - 4) * 4096);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  // 16 MFMAs of one 16-k step; FRESH: the accumulators start from zero (first K tile of an output tile)
  auto phase = [&](auto cb, auto fresh, auto body) STAR_ALWAYS_INLINE {
    constexpr int Bf = decltype(cb)::value;
    constexpr bool FRESH = decltype(fresh)::value;
    static_for<16>([&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      constexpr int i = Q >> 2, j = (i & 1) ? 3 - (Q & 3) : (Q & 3);   // serpentine: consecutive MFMAs share one operand
      if constexpr (FRESH) {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        acc[i][j] = mfma32<T>(fw[Bf][j], fa[Bf][i], zero);
      } else {
        acc[i][j] = mfma32<T>(fw[Bf][j], fa[Bf][i], acc[i][j]);
      }
      body(q);
      STAR_SCHED_FENCE();
    });
  };
  // fragment reads of the next phase in the even MFMA gaps, in the order its MFMAs first need them: W0 A0 W1 W2 W3 A1 A2 A3
  auto rd = [&](auto cks, auto q, auto cb) STAR_ALWAYS_INLINE {
    constexpr int Q = decltype(q)::value;
    if constexpr ((Q & 1) == 0) {
      constexpr int R = Q >> 1;
      constexpr int I = R == 0 ? 4 : R == 1 ? 0 : R == 2 ? 5 : R == 3 ? 6 : R == 4 ? 7 : R - 4;
      frag_read(cks, std::integral_constant<int, I>{}, cb);
    }
  };
  // One K tile (stage buf): phase 0 carries the W pieces of the NEXT K tile of the stream (into the other stage), phase 3 the A
  // pieces of the next-but-one (into this stage, read out behind the barrier of phase 2) -- across output tiles alike.
  // STEADY: both streams are known to be live (every K tile but the last two of the workgroup's whole stream): no branch between
  // the MFMAs.  (The first build tested the streams' flags around each of the 16 copies: 16 s_cbranch + v_cndmask / v_cmp per K
  // tile, 8192^3 4 % behind tile 17 instead of ahead.)
  auto ktile = [&](int buf, auto fresh, auto steady, int n0_bias, int par_bias) STAR_ALWAYS_INLINE {
    constexpr bool FRESH = decltype(fresh)::value, STEADY = decltype(steady)::value;
    const bool wl = STEADY || sw_live, al = STEADY || sa_live;
    phase(B0{}, fresh, [&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      rd(B1{}, q, B1{});
      if constexpr ((Q & 1) == 1) { if (wl) w_piece(buf ^ 1, std::integral_constant<int, (Q >> 1)>{}); }
      if constexpr (FRESH && Q == 14) bias_pieces(n0_bias, par_bias);
    });
    if (wl) w_advance();
    phase(B1{}, std::false_type{}, [&](auto q) STAR_ALWAYS_INLINE { rd(std::integral_constant<int, 2>{}, q, B0{}); });
    phase(B0{}, std::false_type{}, [&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      rd(std::integral_constant<int, 3>{}, q, B1{});
      if constexpr (Q == 14) { STAR_WAIT_VMCNT(0); }
      if constexpr (Q == 15) barrier_keep_dma();
    });
    sa ^= A_STAGE; swo ^= W_STAGE;
    phase(B1{}, std::false_type{}, [&](auto q) STAR_ALWAYS_INLINE {
      constexpr int Q = decltype(q)::value;
      rd(B0{}, q, B0{});
      if constexpr ((Q & 1) == 1) { if (al) a_piece(buf, std::integral_constant<int, (Q >> 1)>{}); }
    });
    if (al) a_advance();
  };

  // ---- epilogue (wave-private; every lane-derived address is re-derived per tile from an opaquely re-read thread id: hipcc
  // otherwise hoists ~50 loop-invariant address registers out of the tile loop and spills them to scratch)
  auto epilogue = [&](int m0, int n0, int par) STAR_ALWAYS_INLINE {
    const int tide = opaque_int((int)threadIdx.x);
    const int le = tide & 63, fr = le & 31, fh = le >> 5;
    char* stg = smem + STG_OFF + wv * STG;
    const char* stw = stg + fr * 128 + fh * 8;                                  // + ((c ^ (fr & 7)) << 4): this lane's 8-byte slots
    const int sw7 = fr & 7;
    const char* strd = stg + (le >> 3) * 128 + (((le & 7) ^ ((le >> 3) & 7)) << 4);   // LDS -> global: row 8 u + (le >> 3), 16-byte chunk le & 7
    const int rows_here = p.M - m0 < BM ? p.M - m0 : BM;
    const BufRsrc crs = make_rsrc((const char*)p.C + (size_t)m0 * p.ldc * 2, (uint32_t)((size_t)rows_here * p.ldc * 2));
    const float* bl = reinterpret_cast<const float*>(smem + BIAS_OFF + par * 2048) + wn * 128 + 4 * fh;
    float ra[4], rb[4];
    if constexpr (ROWAFF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int m = m0 + wm * 128 + i * 32 + fr;
        if (m > p.M - 1) m = p.M - 1;
        const vec<float, 2> ab = *reinterpret_cast<const vec<float, 2>*>(p.rowab + 2 * (size_t)m);
        ra[i] = ab[0]; rb[i] = ab[1];
      }
    }
    // byte offset of this lane's 16-byte chunk in row (wm * 128 + (le >> 3)) of the tile, for the two 64-column halves; a chunk past
    // column N gets an offset outside every descriptor range (dropped); rows past M fall outside the descriptor by themselves
    // (GEGLU: a wave's 128 columns give 64 outputs = ONE 128-byte line per row, columns (n0 + 128 wn) / 2 ...)
    const int n_out = GEGLUF ? p.N / 2 : p.N;
    const int colb = (GEGLUF ? (n0 + wn * 128) / 2 : n0 + wn * 128) + (le & 7) * 8;
    uint32_t voff = (uint32_t)((wm * 128 + (le >> 3)) * p.ldc + colb) * 2u;
    const uint32_t dead0 = colb < n_out ? 0u : GLDS_BUF_OOB, dead1 = colb + 64 < n_out ? 0u : GLDS_BUF_OOB;
    const uint32_t step = (uint32_t)p.ldc * 16u;                               // 8 rows down
    // the residual through a descriptor of its own (rows past M / columns past N read as zeros; those results are never stored)
    BufRsrc rrs = crs;
    uint32_t roff = 0;
    const uint32_t rstep = (uint32_t)p.ldr * 16u;
    if constexpr (RESF) {
      rrs = make_rsrc((const char*)p.res + (size_t)m0 * p.ldr * 2, (uint32_t)((size_t)rows_here * p.ldr * 2));
      roff = (uint32_t)((wm * 128 + (le >> 3)) * p.ldr + colb) * 2u;
    }
    static_for<8>([&](auto ut) STAR_ALWAYS_INLINE {
      constexpr int U = decltype(ut)::value, i = U >> 1, jh = U & 1;
      vec<T, 8> rv[RESF ? 4 : 1];
      if constexpr (RESF) {
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = __builtin_bit_cast(vec<T, 8>, buf_load16(rrs, (roff + (uint32_t)u * rstep + (uint32_t)(jh * 128)) | (jh ? dead1 : dead0)));
        if constexpr (jh == 1) roff += 4 * rstep;
      }
      // registers -> the wave's staging block: row fr, 16-byte chunk c = (column within the 64) / 8 at c ^ (fr & 7), half fh
      if constexpr (GEGLUF) {
        // value block 2 jh, gate block 2 jh + 1 -> 32 outputs per row: chunks 4 jh .. 4 jh + 3; the line is complete after jh = 1
        STAR_AGPR_PIN(acc[i][2 * jh]);
        STAR_AGPR_PIN(acc[i][2 * jh + 1]);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int nl = 2 * jh * 32 + 8 * g4;
          f32x4 v, gt;
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * jh][g4 * 4 + e]; gt[e] = acc[i][2 * jh + 1][g4 * 4 + e]; }
          const f32x4 cb = *reinterpret_cast<const f32x4*>(bl + nl), gb = *reinterpret_cast<const f32x4*>(bl + nl + 32);
          if constexpr (ROWAFF) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(bl + 256 + nl), gs = *reinterpret_cast<const f32x4*>(bl + 256 + nl + 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = ra[i] * v[e] + (rb[i] * cs[e] + cb[e]); gt[e] = ra[i] * gt[e] + (rb[i] * gs[e] + gb[e]); }
          } else {
            v += cb; gt += gb;
          }
          v = v * gelu_erf4(gt);
          vec<T, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
          *reinterpret_cast<vec<T, 4>*>(const_cast<char*>(stw) + (((jh * 4 + g4) ^ sw7) << 4)) = o;
        }
      } else {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        // the block's accumulators are in the accumulator half of the register file UNTIL HERE: without the pin the register allocator
        // moves all 256 of them to architectural registers at the top of the epilogue and spills ~20 loop-invariant values for it
        STAR_AGPR_PIN(acc[i][2 * jh + jj]);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int nl = (2 * jh + jj) * 32 + 8 * g4;   // column within the wave's 128 (+ 4 fh, in bl)
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][2 * jh + jj][g4 * 4 + e];
          const f32x4 cb = *reinterpret_cast<const f32x4*>(bl + nl);   // zeros when the layer has no bias
          if constexpr (ROWAFF) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(bl + 256 + nl);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ra[i] * v[e] + (rb[i] * cs[e] + cb[e]);
          } else {
            v += cb;
          }
          vec<T, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
          *reinterpret_cast<vec<T, 4>*>(const_cast<char*>(stw) + (((jj * 4 + g4) ^ sw7) << 4)) = o;
        }
      }
      }
      if constexpr (!GEGLUF || jh == 1) {
      wave_lds_order();
      // LDS -> global: 4 instructions of 8 whole 128-byte lines each (+ residual, already in registers)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vec<T, 8> ov = *reinterpret_cast<const vec<T, 8>*>(strd + u * 1024);
        if constexpr (RESF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(to_f32<T>(ov[e]) + to_f32<T>(rv[u][e]));
        }
        buf_store16_pol<STPOL>(crs, (voff + (uint32_t)(GEGLUF ? 0 : jh * 128)) | ((!GEGLUF && jh) ? dead1 : dead0), __builtin_bit_cast(u32x4, ov));
        voff += step;                                   // next 8 rows
      }
      if constexpr (!GEGLUF && jh == 0) voff -= 4 * step;   // back to the unit's first rows for the second half
      }
      wave_lds_order();   // the block is read out before the next unit writes it (DS operations of a wave execute in order)
      STAR_SCHED_FENCE();  // units stay apart: hoisting the next units' accumulator reads costs ~20 registers of scratch
    });
  };

  // ---- prologue: K tile 0 of the stream complete, the A pieces of K tile 1 behind it
  static_for<NP>([&](auto jc) STAR_ALWAYS_INLINE { a_piece(0, jc); });
  a_advance();
  static_for<NP>([&](auto jc) STAR_ALWAYS_INLINE { w_piece(0, jc); });
  w_advance();
  if (sa_live) {
    static_for<NP>([&](auto jc) STAR_ALWAYS_INLINE { a_piece(1, jc); });
    a_advance();
    STAR_WAIT_VMCNT_N(NP);
  } else {
    STAR_WAIT_VMCNT(0);
  }
  barrier_keep_dma();
  static_for<8>([&](auto c) STAR_ALWAYS_INLINE { frag_read(B0{}, c, B0{}); });
  STAR_SCHED_FENCE();

  int g = 0;   // index of the K tile in this workgroup's stream (stage = g & 1)
  for (int i = 0; i < ntm; ++i) {
    int m0, n0;
    tile_origin(i, m0, n0);
    // the last two K tiles of the workgroup's WHOLE stream have nothing (or only W) left to stage: guarded form; all others (stream
    // index <= ntm * nk - 3): steady form.  (With nk = 1 the last two K tiles belong to two output tiles.)
    int nks = ntm * nk - 2 - g;           // steady K tiles from this tile's first one on
    nks = nks < 0 ? 0 : (nks > nk ? nk : nks);
    if (nks > 0) ktile(g & 1, std::true_type{}, std::true_type{}, n0, i & 1);
    else ktile(g & 1, std::true_type{}, std::false_type{}, n0, i & 1);
    ++g;
    int kt = 1;
    for (; kt < nks; ++kt, ++g) ktile(g & 1, std::false_type{}, std::true_type{}, 0, 0);
    for (; kt < nk; ++kt, ++g) ktile(g & 1, std::false_type{}, std::false_type{}, 0, 0);
    epilogue(m0, n0, i & 1);
  }
}

}  // namespace star
