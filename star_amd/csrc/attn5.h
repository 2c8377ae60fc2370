// attn5.h -- flash_attn_v5_kernel: the lazy-maxima d = 64 flash attention of attn.h (flash_attn_v3_kernel<T, 2, 1, 0, 1>, the
// round-1/2 product kernel; replaces xformers.ops.memory_efficient_attention at unet_v2v.py:184-185) with the VALU that is
// NOT softmax taken out of the key-tile loop.  The ISA of the v3 loop (hipcc 7.2) carries, per 64-key tile and wave, 36 MFMAs
// against ~205 VALU issues of which only 160 are the softmax itself (64 v_exp, 64 row-sum adds, 32 v_cvt_pk):
//   * LDS-DMA staging recomputed four 64-bit global addresses from the key index every tile (v_min, v_mad_i64_i32,
//     v_lshl_add_u64) and made each LDS base uniform with a v_readfirstlane  ->  here a wave-uniform tile pointer (SALU add per
//     tile) plus a loop-invariant 32-bit lane offset (the saddr form of global_load_lds), LDS bases from a uniform wave id;
//   * every K / V^T fragment address was rebuilt from (buffer, row, swizzle) terms with two or three VALU adds per ds_read
//     ->  the tile loop is unrolled by two so the LDS buffer is a compile-time constant and folds, with the key block, into
//     the immediate offset of the ds_read; 8 loop-invariant address registers remain (the swizzle term of a fragment does not
//     depend on the 16-key step: ((key >> 1) & 7) is periodic in 16 keys);
//   * PKSUM (f16 only, long key ranges only): the row sum of a tile's 32 probabilities per lane is a tree of 15 v_pk_add_f16
//     on the ROUNDED probabilities (exactly what the PV MFMA multiplies) + 2 converts + 1 add instead of 32 fp32 adds.  The sum
//     still doubles as the overflow probe of the lazy maxima (an fp16 overflow gives inf, which fails the probe).  The rounding
//     error of a 5-level fp16 tree is zero-mean, <= 2.4e-3 of the tile's partial sum and averages out over the hundreds of
//     tiles of a spatial self-attention row; launches with Nk < 1024 (the 77-token cross-attention) keep the fp32 adds.
// Everything else -- operand-swapped S^T = K Q^T, scale and running max riding in an augmented k-step, lazy maxima with the
// row sum as overflow probe, P^T directly the B operand of O^T = V^T P^T, ds_read_b64_tr_b16 for V^T, XCD-aware block order,
// two 4-wave workgroups per CU -- is flash_attn_v3_kernel<T, 2, 1, 0, 1>; the results are bit-identical to it with PKSUM = 0.
#pragma once
#include "attn.h"

namespace star {

// CAUSAL = 1 (text tower, embedder.py:59: open_clip's attn_mask): key j is visible to query i only for j <= i; Nq == Nk, the mask
// is applied to the scores of every tile (and again on the recompute path), so masked probabilities are exact zeros.
// NW: waves per workgroup.  4 = the shipped form (two workgroups per CU, each staging its own K / V tiles); 8 = ONE 512-thread workgroup
// per CU whose eight waves share one K / V stage (round 6, VERDICT r05 item 5: half the L2 -> LDS bytes per flop; the waves run free
// between the per-tile barriers).  Same arithmetic per query row: bit-identical outputs.
// RTZ (f16 with PKSUM only): the probabilities are packed with v_cvt_pkrtz_f16_f32 (round toward zero) instead of v_cvt_pk_f16_f32 (nearest
// even).  The row sum is taken from the SAME rounded probabilities the PV MFMA multiplies, so the common downward shift of a
// truncation cancels in O = sum(P V) / sum(P); what remains is a per-element error of the same variance as nearest-even's.
template <class T, int PKSUM, int AUGK8 = 0, int CAUSAL = 0, int NW = 4, int RTZ = 0>   // AUGK8: the augmented k-step as a half-depth 32x32x8 MFMA (one useful k of 8 instead of 16: +1-4 %, bit-identical; shipped)
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(NW * 64, 8 / NW)
flash_attn_v5_kernel(const AttnParams p) {
  constexpr int NQ = 2, QW = 64, QB = NW * 64, KT = 64, TILE = KT * 128, BUFB = 2 * TILE;   // one buffer = K tile | V tile (16 KB)
  constexpr int NT = NW * 64, NJ = 512 / NT;     // threads; 16-byte chunks per thread and operand tile (64 rows x 8 chunks)
  static_assert(NW == 4 || NW == 8, "");
  constexpr float LAZY_BIG = 1024.0f;
  static_assert(PKSUM == 0 || sizeof(T) == 2, "");
  static_assert(RTZ == 0 || (PKSUM != 0 && __is_same(T, f16)), "");
  char* smem = dyn_smem();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = wave_uniform(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int bh = xcd + 8 * (slot / p.nqb);
  const int qb = slot % p.nqb;
  if (bh >= p.batch * p.heads) return;
  const int b = bh / p.heads, hd = bh % p.heads;
  const T* __restrict__ Qg = (const T*)p.Q + (size_t)b * p.bsq + hd * 64;
  const T* __restrict__ Kg = (const T*)p.K + (size_t)b * p.bsk + hd * 64;
  const T* __restrict__ Vg = (const T*)p.V + (size_t)b * p.bsv + hd * 64;
  T* __restrict__ Og = (T*)p.O + (size_t)b * p.bso + hd * 64;

  vec<T, 8> qf[NQ][4];
  const int q_base = qb * QB + wv * QW;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    int q = q_base + qi * 32 + lq;
    if (q > p.Nq - 1) q = p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const vec<T, 8> raw = *reinterpret_cast<const vec<T, 8>*>(Qg + (size_t)q * p.ldq + ks * 16 + h2 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[qi][ks][e] = from_f32<T>(to_f32<T>(raw[e]) * p.scale_log2e);   // softmax scale * log2(e) folded into Q
    }
  }
  // augmented k-step: K_aug[kv][0] = 1, Q_aug[q][0] = -m_run(q) -> the QK^T accumulators come out as (scaled score - running max)
  vec<T, 8> kaug, qaug[NQ];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    kaug[e] = from_f32<T>(0.f);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) qaug[qi][e] = from_f32<T>(0.f);
  }
  if (h2 == 0) kaug[0] = from_f32<T>(1.0f);

  // ---- K/V staging: thread (j, tid) copies 16-B chunk (tid & 7) ^ swizzle of tile row r_j = (j*256 + tid) >> 3
  const int pos = tid & 7;
  uint32_t koff[NJ], voff[NJ];     // loop-invariant byte offsets of this lane's chunks from the tile's first row
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int r = (j * NT + tid) >> 3;
    const int c = pos ^ ((r >> 1) & 7);
    koff[j] = (uint32_t)(r * p.ldk + c * 8) * 2u;
    voff[j] = (uint32_t)(r * p.ldv + c * 8) * 2u;
  }
  const size_t kstep = (size_t)KT * p.ldk * 2, vstep = (size_t)KT * p.ldv * 2;   // bytes per key tile
  const int nt = (p.Nk + KT - 1) / KT;
  const bool has_tail = (p.Nk & (KT - 1)) != 0;
  const int nfull = has_tail ? nt - 1 : nt;
  auto stage = [&](int t, int buf) STAR_ALWAYS_INLINE {
    char* kdst = smem + buf * BUFB;
    char* vdst = kdst + TILE;
    if (t < nfull) {
      const char* kt = (const char*)Kg + (size_t)t * kstep;   // wave-uniform
      const char* vt = (const char*)Vg + (size_t)t * vstep;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        glds16_su(kt, koff[j], kdst + (size_t)(j * NT + wv * 64) * 16);
        glds16_su(vt, voff[j], vdst + (size_t)(j * NT + wv * 64) * 16);
      }
    } else {                       // the ragged last tile: rows past Nk re-read the last key (masked in the scores)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = (j * NT + tid) >> 3;
        const int c = pos ^ ((r >> 1) & 7);
        int key = t * KT + r;
        if (key > p.Nk - 1) key = p.Nk - 1;
        glds16(Kg + (size_t)key * p.ldk + c * 8, kdst + (size_t)(j * NT + wv * 64) * 16);
        glds16(Vg + (size_t)key * p.ldv + c * 8, vdst + (size_t)(j * NT + wv * 64) * 16);
      }
    }
  };

  // ---- loop-invariant fragment addresses (bytes within a K or V tile)
  // K (A operand of S^T = K Q^T): row kb*32 + lq, 16-B chunk ks*2 + h2; kb only adds 4096 (the swizzle has period 16 rows)
  const char* kfo[4];              // absolute LDS addresses (buffer 0); opaque() keeps hipcc from re-deriving them per tile
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kfo[ks] = opaque(smem + swz_off(lq, ks * 2 + h2));
  // V^T (A operand of O^T = V^T P^T), see load_vt_frag: key = 16*tt + 8*half + 4*h + (i >> 2), d = 32*db + 16*(g & 1) + 4*(i & 3);
  // the 16-key step tt only adds 2048
  const char* vfo[2][2];
  {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int key = 8 * half + 4 * h2 + (i >> 2);
        const int d = 32 * db + 16 * (g & 1) + 4 * (i & 3);
        vfo[half][db] = opaque(smem + TILE + swz_off(key, d >> 3) + (d & 7) * 2);
      }
  }

  f32x16 oacc[NQ][2];
#pragma unroll
  for (int a = 0; a < NQ; ++a)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[a][c2][r] = 0.f;
  float m_run[NQ], l_run[NQ];      // m_run is always exactly representable in T (it is fed to the MFMA through Q_aug)
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) { m_run[qi] = 0.f; l_run[qi] = 0.f; }

  // one 64-key tile.  buf: std::integral_constant<int, 0 | 1> in the unrolled steady state (addresses fold into immediates),
  // a run-time int for the ragged last tile
  auto tile = [&](int t, auto mask_tag, auto first_tag, auto buf) STAR_ALWAYS_INLINE {
    constexpr bool MASK = decltype(mask_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;   // may be tile 0 (running max not set yet)
    const int bofs = (int)buf * BUFB;
    f32x16 s[NQ][2];
    auto scores = [&]() STAR_ALWAYS_INLINE {
#pragma unroll
      for (int a = 0; a < NQ; ++a)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[a][kb][r] = 0.f;
          if constexpr (AUGK8) {
            vec<T, 4> k4, q4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { k4[e] = kaug[e]; q4[e] = qaug[a][e]; }
            s[a][kb] = mfma32_k8<T>(k4, q4, s[a][kb]);
          } else
          s[a][kb] = mfma32<T>(kaug, qaug[a], s[a][kb]);      // -m_run broadcast over the 32 keys
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kfo[ks] + bofs + kb * 4096);
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) s[qi][kb] = mfma32<T>(kf, qf[qi][ks], s[qi][kb]);
        }
      if constexpr (MASK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            if (key >= p.Nk) {
#pragma unroll
              for (int qi = 0; qi < NQ; ++qi) s[qi][kb][r] = -1e30f;
            }
          }
      }
      if constexpr (CAUSAL != 0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
              if (key > q_base + qi * 32 + lq) s[qi][kb][r] = -1e30f;
            }
          }
      }
    };
    float m_tile[NQ];
    auto maxima = [&]() STAR_ALWAYS_INLINE {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float mx[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kb = g >> 1, o = (g & 1) * 8;
          const float a0 = fmaxf(fmaxf(s[qi][kb][o], s[qi][kb][o + 1]), s[qi][kb][o + 2]);
          const float a1 = fmaxf(fmaxf(s[qi][kb][o + 3], s[qi][kb][o + 4]), s[qi][kb][o + 5]);
          mx[g] = fmaxf(fmaxf(a0, a1), fmaxf(s[qi][kb][o + 6], s[qi][kb][o + 7]));
        }
        m_tile[qi] = pair_max(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])));
      }
    };
    // rare: move the running max, rescale O / l once, re-base this tile's scores
    auto rebase = [&]() STAR_ALWAYS_INLINE {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const float inc = (t == 0) ? m_tile[qi] : fmaxf(m_tile[qi], 0.f);
        const float m_new = to_f32<T>(from_f32<T>(m_run[qi] + inc));
        const float delta = m_new - m_run[qi];
        const float alpha = fast_exp2(-delta);
        m_run[qi] = m_new;
        l_run[qi] *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qi][db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qi][kb][r] -= delta;
        if (h2 == 0) qaug[qi][0] = from_f32<T>(-m_new);
      }
    };
    vec<T, 8> pf[NQ][4];
    float lsum[NQ];
    auto expo = [&](int qi) STAR_ALWAYS_INLINE {
      if constexpr (PKSUM != 0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            vec<T, 8> pk;
            if constexpr (RTZ != 0) {
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const vec<T, 2> h2v = cvt_pkrtz<T>(fast_exp2(s[qi][kb][8 * u + e]), fast_exp2(s[qi][kb][8 * u + e + 1]));
                pk[e] = h2v[0]; pk[e + 1] = h2v[1];
              }
            } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(fast_exp2(s[qi][kb][8 * u + e]));
            }
            pf[qi][kb * 2 + u] = pk;
          }
        vec<T, 2> h[16];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[w * 4 + e][0] = pf[qi][w][2 * e]; h[w * 4 + e][1] = pf[qi][w][2 * e + 1]; }
#pragma unroll
        for (int n = 8; n >= 1; n >>= 1)
#pragma unroll
          for (int i = 0; i < n; ++i) h[i] = pk_add<T>(h[i], h[i + n]);
        lsum[qi] = to_f32<T>(h[0][0]) + to_f32<T>(h[0][1]);
      } else {
        float ls0 = 0.f, ls1 = 0.f, ls2 = 0.f, ls3 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            vec<T, 8> pk;
            float e8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) e8[e] = fast_exp2(s[qi][kb][8 * u + e]);
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
              ls0 += e8[e]; ls1 += e8[e + 1];
              ls2 += e8[e + 2]; ls3 += e8[e + 3];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(e8[e]);
            pf[qi][kb * 2 + u] = pk;
          }
        lsum[qi] = (ls0 + ls1) + (ls2 + ls3);
      }
    };

    scores();
    if constexpr (FIRST) { if (t == 0) { maxima(); rebase(); } }
    bool bad = false;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { expo(qi); bad = bad || !(lsum[qi] <= LAZY_BIG); }
    if (wave_any(bad)) {          // some P is large (or overflowed): exact maxima from recomputed scores, then redo
      scores();
      maxima();
      rebase();
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) expo(qi);
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) l_run[qi] += lsum[qi];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        vec<T, 8> vf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const vec<T, 4> r = lds_read_tr<T>(vfo[half][db] + bofs + tt * 2048);
#pragma unroll
          for (int e = 0; e < 4; ++e) vf[half * 4 + e] = r[e];
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) oacc[qi][db] = mfma32<T>(vf, pf[qi][tt], oacc[qi][db]);
      }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  glds_wait();                      // the Q loads are done before the first hand-written LDS-DMA (prim.h: glds16_su)
  stage(0, 0);
  int t = 0;
  for (; t + 1 < nfull; t += 2) {   // two full tiles per trip: buffer 0 then buffer 1
    glds_wait(); block_sync();
    stage(t + 1, 1);
    tile(t, std::false_type{}, std::true_type{}, B0{});
    glds_wait(); block_sync();
    if (t + 2 < nt) stage(t + 2, 0);
    tile(t + 1, std::false_type{}, std::false_type{}, B1{});
  }
  if (t < nfull) {                  // odd number of full tiles: t is even here
    glds_wait(); block_sync();
    if (t + 1 < nt) stage(t + 1, 1);
    tile(t, std::false_type{}, std::true_type{}, B0{});
    ++t;
  }
  if (has_tail) {
    glds_wait(); block_sync();
    tile(nt - 1, std::true_type{}, std::true_type{}, (nt - 1) & 1);
  }

#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float l = pair_sum(l_run[qi]);
    const float inv = 1.0f / l;
    const int q = q_base + qi * 32 + lq;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        uint32_t w0[2], w1[2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          vec<T, 4> o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = from_f32<T>(oacc[qi][db][(2 * a + gg) * 4 + e] * inv);
          u32x2 pk = __builtin_bit_cast(u32x2, o4);
          if (gg == 0) { w0[0] = pk[0]; w0[1] = pk[1]; } else { w1[0] = pk[0]; w1[1] = pk[1]; }
        }
        const u32x2 x0 = permlane32_swap(w0[0], w1[0]);
        const u32x2 x1 = permlane32_swap(w0[1], w1[1]);
        u32x4 out;
        out[0] = x0[0]; out[1] = x1[0]; out[2] = x0[1]; out[3] = x1[1];
        if (q < p.Nq) *reinterpret_cast<u32x4*>(Og + (size_t)q * p.ldo + 32 * db + 16 * a + 8 * h2) = out;
      }
  }
}

}  // namespace star
