// attn.h -- attention kernels of the hot path (head dim 64, no mask, no dropout).
//
//  flash_attn_kernel : softmax(Q K^T / sqrt(64)) V with online softmax, for the
//      spatial self-attention (N = H*W up to 1.3e5 tokens; SURVEY.md K1, reference
//      unet_v2v.py:158-195 via xformers.ops.memory_efficient_attention) and the
//      cross-attention to the 77 text tokens (K2; K/V shared by all frames).
//  temporal_attn_kernel : the per-pixel attention over the frame axis (K3;
//      unet_v2v.py:479-489), one wavefront per (pixel, head), F <= 64.
//
// The product ships ONE spatial kernel, flash_attn_v5_kernel (attn5.h, AttnArgs::variant 9): the algorithm of
// flash_attn_v3_kernel<T, 2, LAZY=1, 0, ROWSUM=1> below (scale + running max folded into the MFMA, lazy row maxima with the
// row sum of P as overflow probe) with the non-softmax VALU taken out of the key-tile loop (+9-16 %).  The v3 kernel itself
// (variant 32, the product of rounds 1-2) and its other template modes are the measured A/B
// variants and ablation probes of profiles/r01_attn_ab.txt; they are instantiated only in the bench build
// (-DSTAR_BENCH_VARIANTS) and the test emulator:
//   2 / 3   v3 without lazy maxima (NQ = 2 / 1)      6 / 7   lazy maxima, one probe per tile / per block
//   8       key-half pipeline      10 / 15  chunk pipeline      21 / 22  K fragments read ahead / K-V ring of three
//   11-14, 16, 17  ablation probes (wrong results by construction)      40 / 41  attn7.h (round 3)
// (variants 0 / 1 / 4 / 5 / 20 -- baseline, v2, the tile-level software pipelines and the antiphase kernel -- were removed in round 3)
//
// Both compute S^T = K Q^T with the MFMA operands swapped, so a lane owns one
// query row: its 32 scores per 64-key tile sit in its own registers, the row max
// and row sum are lane-local (+ one exchange with lane^32), and the packed
// probabilities are directly the B operand of O^T = V^T P^T -- no shuffles, no
// LDS round trip for P.  V^T fragments come from ds_read_b64_tr_b16.
#pragma once
#include "prim.h"
#include <type_traits>

namespace star {

struct AttnParams {
  int variant;                       // 0 = baseline kernel, 1 = v2 (staggered streams + deferred rescale)
  const void* Q; const void* K; const void* V; void* O;
  int ldq, ldk, ldv, ldo;            // row strides (elements)
  long long bsq, bsk, bsv, bso;      // batch strides (elements); bsk = bsv = 0 for a shared context
  int Nq, Nk, heads, batch;
  float scale_log2e;                 // softmax scale * log2(e)
  int nqb;                           // q blocks per (batch, head)
};

// chunk swizzle shared with gemm.h: 16-B chunk c of 128-B row r lives at chunk c ^ ((r>>1)&7)
STAR_DEV int swz_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// V^T fragment for MFMA A operand: rows d = 32*db + (lane&31), k-slot j <-> key kb0 + 4h + (j&3) + 8*(j>>2), h = lane>>5
template <class T>
STAR_DEV vec<T, 8> load_vt_frag(const char* vbuf, int kb0, int db, int lane) {
  const int i = lane & 15, g = lane >> 4, h = lane >> 5;
  const int d = 32 * db + 16 * (g & 1) + 4 * (i & 3);
  vec<T, 8> out;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int key = kb0 + 8 * half + 4 * h + (i >> 2);
    const char* addr = vbuf + swz_off(key, d >> 3) + (d & 7) * 2;
    vec<T, 4> r = lds_read_tr<T>(addr);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[half * 4 + e] = r[e];
  }
  return out;
}

// flash_attn_v3_kernel: v2 plus two VALU-saving moves for the VALU-bound d = 64 regime (PMC: matrix pipe 38 % busy,
// VALU ~73 %): (1) the softmax scale is folded into Q and the running max into the MFMA through one augmented k-step
// (K_aug[kv][0] = 1, Q_aug[q][0] = -m_q), so the accumulators already hold (score - max) and P = exp2(acc): 4 extra
// MFMAs per tile replace 64 v_fma; m_q is kept exactly representable in T, a common row factor cancels in O = PV / l;
// (2) row sums are taken from the packed (rounded) probabilities with v_dot2c (32 ops instead of 64 adds).
// LAZY (variants 6 / 7): the row maxima are not computed at all on the common path.  P = exp2(score - running max) is
// formed directly; its row sum (needed anyway) doubles as the overflow probe: a lane whose partial row sum stays <= 2^10
// holds no P above 2^10, which is as good as a rescale threshold of 10.  Only when a sum exceeds that (or is inf/NaN)
// does the wave recompute the tile's scores from the K tile still in LDS, take the exact maxima, move the running max
// and redo the exponentials -- 38 v_max per tile traded for 9-18 extra MFMAs on the rare tiles where the max moves.
// LAZY = 1: one probe per tile; LAZY = 2: one probe per 32-row query block, PV of block 0 beside the exponentials of block 1.
// ROWSUM = 1: row sums as four chains of plain fp32 adds on the unrounded exponentials instead of v_dot2c on the packed P
// (v_dot2c costs ~10 cycles beyond its issue slot beside MFMAs; attn.o is built with -fno-slp-vectorize so the adds stay
// single-issue instead of being fused into v_pk_add_f32, which stalls beside MFMAs as well).
// KPRE = 1 (variant 21): all eight K fragments of a tile are read before its first MFMA (the probabilities of the previous
// tile are dead by then, so their 32 registers are free) instead of two at a time just ahead of their use.
// RING3 = 1 (variant 22): K/V tiles in a ring of three, the LDS-DMA of tile t+2 is issued at tile t and waited with a counted
// vmcnt behind a raw s_barrier, so a tile never waits for a DMA issued only one tile earlier.
template <class T, int NQ, int LAZY = 0, int ABL = 0, int ROWSUM = 0, int KPRE = 0, int RING3 = 0, int SPRIO = 0>   // NQ = 32-row query blocks per wave: 2 -> 2 waves/SIMD (256 VGPRs), 1 -> 4 waves/SIMD (128 VGPRs); ABL: ablation probes (bench only)
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, (NQ == 2 ? 2 : 4))
flash_attn_v3_kernel(const AttnParams p) {
  constexpr int QW = 32 * NQ, QB = 4 * QW, KT = 64, TILE = KT * 128;
  constexpr float RESCALE_THR = 8.0f;   // log2 units: P <= 2^8 before a rescale is forced
  char* smem = dyn_smem();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, lq = lane & 31;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int bh = xcd + 8 * (slot / p.nqb);
  const int qb = slot % p.nqb;
  if (bh >= p.batch * p.heads) return;
  // static priority experiments (bench): one of the two workgroups that share a CU always wins the issue arbitration
  if constexpr (SPRIO == 2) { if (bid & 1) STAR_SETPRIO(1); }
  if constexpr (SPRIO == 3) { if ((bid >> 8) & 1) STAR_SETPRIO(1); }
  if constexpr (SPRIO == 4) { if ((bid >> 3) & 1) STAR_SETPRIO(1); }
  const int b = bh / p.heads, hd = bh % p.heads;
  const T* __restrict__ Qg = (const T*)p.Q + (size_t)b * p.bsq + hd * 64;
  const T* __restrict__ Kg = (const T*)p.K + (size_t)b * p.bsk + hd * 64;
  const T* __restrict__ Vg = (const T*)p.V + (size_t)b * p.bsv + hd * 64;
  T* __restrict__ Og = (T*)p.O + (size_t)b * p.bso + hd * 64;

  vec<T, 8> qf[NQ][4];
  const int q_base = qb * QB + wave * QW;
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
  const int pos = tid & 7;
  auto stage = [&](int t, int buf) {
    char* kbuf = smem + buf * 2 * TILE;
    char* vbuf = kbuf + TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (j * 256 + tid) >> 3;
      const int c = pos ^ ((r >> 1) & 7);
      int key = t * KT + r;
      if (key > p.Nk - 1) key = p.Nk - 1;
      glds16(Kg + (size_t)key * p.ldk + c * 8, kbuf + (size_t)(j * 256 + wave * 64) * 16);
      glds16(Vg + (size_t)key * p.ldv + c * 8, vbuf + (size_t)(j * 256 + wave * 64) * 16);
    }
  };

  f32x16 oacc[NQ][2];
#pragma unroll
  for (int a = 0; a < NQ; ++a)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[a][c2][r] = 0.f;
  float m_run[NQ];
  float l_run[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) { m_run[qi] = 0.f; l_run[qi] = 0.f; }
  //          // always exactly representable in T (it is fed to the MFMA through Q_aug)
  const int nt = (p.Nk + KT - 1) / KT;

  auto tile = [&](int t, auto mask_tag, auto first_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;   // LAZY == 3: the first key tile is peeled out of the loop
    constexpr float LAZY_BIG = 1024.0f;
    const char* kbuf = smem + (RING3 == 2 ? (((t >> 1) & 1) * 2 + (t & 1)) : RING3 ? t % 3 : (t & 1)) * 2 * TILE;
    const char* vbuf = kbuf + TILE;
    f32x16 s[NQ][2];
    // ---- S^T = K Q^T for query blocks [q_lo, q_hi) (8 + 1 MFMAs per block, independent accumulators)
    auto scores = [&](int q_lo, int q_hi) {
      if constexpr (SPRIO == 1) STAR_SETPRIO(1);
#pragma unroll
      for (int a = 0; a < NQ; ++a) {
        if (a < q_lo || a >= q_hi) continue;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[a][kb][r] = 0.f;
          s[a][kb] = mfma32<T>(kaug, qaug[a], s[a][kb]);      // -m_run broadcast over the 32 keys
        }
      }
      if constexpr (KPRE == 1) {
        vec<T, 8> kfa[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) kfa[ks][kb] = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
              if (qi < q_lo || qi >= q_hi) continue;
              s[qi][kb] = mfma32<T>(kfa[ks][kb], qf[qi][ks], s[qi][kb]);
            }
        STAR_SCHED_GROUP(0x100, 8, 0);     // the eight ds_read_b128 first ...
        STAR_SCHED_GROUP(0x008, 20, 0);    // ... then the MFMAs back to back
      } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) {
            if (qi < q_lo || qi >= q_hi) continue;
            s[qi][kb] = mfma32<T>(kf, qf[qi][ks], s[qi][kb]);
          }
        }
      }
      if constexpr (SPRIO == 1) STAR_SETPRIO(0);
      if constexpr (MASK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            if (key >= p.Nk) {
#pragma unroll
              for (int qi = 0; qi < NQ; ++qi) {
                if (qi < q_lo || qi >= q_hi) continue;
                s[qi][kb][r] = -1e30f;
              }
            }
          }
      }
    };
    // ---- row maxima (already relative to the running max) of blocks [q_lo, q_hi)
    float m_tile[NQ];
    auto maxima = [&](int q_lo, int q_hi) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        if (qi < q_lo || qi >= q_hi) continue;
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
    // ---- rare: move the running max of blocks [q_lo, q_hi), rescale O / l once, re-base this tile's scores
    auto rebase = [&](int q_lo, int q_hi) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        if (qi < q_lo || qi >= q_hi) continue;
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
    auto expo = [&](int qi) {
      float ls0 = 0.f, ls1 = 0.f, ls2 = 0.f, ls3 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          vec<T, 8> pk;
          if constexpr (ABL == 5) {      // no softmax VALU at all: constant probabilities (keeps the MFMA / LDS / barrier skeleton)
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(0.001f);
            ls0 += s[qi][kb][8 * u];     // one VALU per 8 scores keeps the QK MFMAs alive
          } else if constexpr (ROWSUM == 1) {
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
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(ABL == 1 ? s[qi][kb][8 * u + e] * 0.001f : fast_exp2(s[qi][kb][8 * u + e]));
#pragma unroll
            for (int e = 0; e < 8; e += 4) {   // row sum of the ROUNDED probabilities (what PV multiplies), 2 per v_dot2c
              vec<T, 2> a, b2;
              a[0] = pk[e]; a[1] = pk[e + 1]; b2[0] = pk[e + 2]; b2[1] = pk[e + 3];
              ls0 = dot2_ones<T>(a, ls0);
              ls1 = dot2_ones<T>(b2, ls1);
            }
          }
          pf[qi][kb * 2 + u] = pk;
        }
      lsum[qi] = (ls0 + ls1) + (ls2 + ls3);
    };
    auto pv = [&](int q_lo, int q_hi) {
      if constexpr (RING3 == 1) {
        // all V^T fragments first, then the LDS-DMA of tile t+2, then the MFMAs: hipcc puts a vmcnt(0) in front of the first
        // transpose read that follows an LDS-DMA issue (it cannot tell the two apart), so the DMA is issued right BEHIND
        // this tile's reads and gets a whole tile before the next such wait
        vec<T, 8> vfa[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int db = 0; db < 2; ++db) vfa[tt][db] = load_vt_frag<T>(vbuf, tt * 16, db, lane);
        if (t + 2 < nt) stage(t + 2, (t + 2) % 3);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
              if (qi < q_lo || qi >= q_hi) continue;
              oacc[qi][db] = mfma32<T>(vfa[tt][db], pf[qi][tt], oacc[qi][db]);
            }
        return;
      }
      if constexpr (SPRIO == 1) STAR_SETPRIO(1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const vec<T, 8> vf = load_vt_frag<T>(vbuf, tt * 16, db, lane);
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) {
            if (qi < q_lo || qi >= q_hi) continue;
            if constexpr (ABL == 2) { oacc[qi][db][tt] += to_f32<T>(pf[qi][tt][db]) + to_f32<T>(vf[0]); continue; }
            oacc[qi][db] = mfma32<T>(vf, pf[qi][tt], oacc[qi][db]);
          }
        }
      if constexpr (SPRIO == 1) STAR_SETPRIO(0);
    };

    if constexpr (LAZY == 0) {
      scores(0, NQ);
      maxima(0, NQ);
      bool grow = m_tile[0] > RESCALE_THR;
      if constexpr (NQ == 2) grow = grow || m_tile[NQ - 1] > RESCALE_THR;
      if (t == 0 || wave_any(grow)) rebase(0, NQ);   // one wave-uniform rescale decision
      // ---- straight-line: P0 ; PV0 beside P1 ; PV1
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) { expo(qi); l_run[qi] += lsum[qi]; }
      pv(0, NQ);
      // interleave: the first 8 PV MFMAs (they only need P0) beside the VALU of expo(1)
      if constexpr (NQ == 2) for (int i = 0; i < 8; ++i) {
        STAR_SCHED_GROUP(0x008, 1, 0);   // 1 MFMA
        STAR_SCHED_GROUP(0x100, 2, 0);   // 2 DS reads
        STAR_SCHED_GROUP(0x002, 12, 0);  // 12 VALU
      }
    } else if constexpr (LAZY == 1) {
      scores(0, NQ);
      if (t == 0) { maxima(0, NQ); rebase(0, NQ); }
      bool bad = false;
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) { expo(qi); bad = bad || !(lsum[qi] <= LAZY_BIG); }
      if (wave_any(bad)) {          // some P is large (or overflowed): exact maxima from recomputed scores, then redo
        scores(0, NQ);
        maxima(0, NQ);
        rebase(0, NQ);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) expo(qi);
      }
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) l_run[qi] += lsum[qi];
      pv(0, NQ);
    } else if constexpr (LAZY == 4 || LAZY == 5) {
      // chunk pipeline (variants 10 / 11): the tile is four chunks (query block qi, key half kb) of 5 MFMAs and ~40 VALU
      // each; the exponentials of one chunk are issued in the MFMA gaps of the next.  Measured on gfx950
      // (tools/probe/overlap.hip): a VALU instruction costs ~8 cycles in a VALU-only stretch of a wave but ~1.3 when up to
      // six of them follow each 32-cycle MFMA.  K fragments of a key half serve both query blocks back to back, V
      // fragments likewise, so LDS traffic is unchanged.  LAZY == 5 adds sched_group_barrier patterns.
      static_assert(NQ == 2, "chunk pipeline is written for two query blocks per wave");
      constexpr bool HINTS = (LAZY == 5);
      vec<T, 8> kfr[4];
      auto load_k = [&](int kb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kfr[ks] = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
      };
      auto qk = [&](int qi, int kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qi][kb][r] = 0.f;
        s[qi][kb] = mfma32<T>(kaug, qaug[qi], s[qi][kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[qi][kb] = mfma32<T>(kfr[ks], qf[qi][ks], s[qi][kb]);
        if constexpr (MASK) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            if (key >= p.Nk) s[qi][kb][r] = -1e30f;
          }
        }
      };
      float lk[NQ];
      auto expo_c = [&](int qi, int kb) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float e8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) e8[e] = fast_exp2(s[qi][kb][8 * u + e]);
#pragma unroll
          for (int e = 0; e < 8; e += 4) { a0 += e8[e]; a1 += e8[e + 1]; a2 += e8[e + 2]; a3 += e8[e + 3]; }
          vec<T, 8> pk;
#pragma unroll
          for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(e8[e]);
          pf[qi][kb * 2 + u] = pk;
        }
        lk[qi] += (a0 + a1) + (a2 + a3);
      };
      vec<T, 8> vfr[2][2];
      auto load_v = [&](int kb) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int db = 0; db < 2; ++db) vfr[u][db] = load_vt_frag<T>(vbuf, (kb * 2 + u) * 16, db, lane);
      };
      auto pv_c = [&](int qi, int kb) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int db = 0; db < 2; ++db) oacc[qi][db] = mfma32<T>(vfr[u][db], pf[qi][kb * 2 + u], oacc[qi][db]);
      };
      auto fix = [&](int qi) {   // rare: exact maxima from recomputed scores, move the running max, redo this block's P
        load_k(0); qk(qi, 0);
        load_k(1); qk(qi, 1);
        maxima(qi, qi + 1);
        rebase(qi, qi + 1);
        lk[qi] = 0.f;
        expo_c(qi, 0);
        expo_c(qi, 1);
      };
      auto hints = [&](auto n_ds, auto n_mfma, auto n_valu) {   // integral_constant arguments: the builtin wants literals
        if constexpr (HINTS) {
          if constexpr (decltype(n_ds)::value > 0) STAR_SCHED_GROUP(0x100, decltype(n_ds)::value, 0);
#pragma unroll
          for (int i = 0; i < decltype(n_mfma)::value; ++i) {
            STAR_SCHED_GROUP(0x008, 1, 0);
            STAR_SCHED_GROUP(0x002, decltype(n_valu)::value, 0);
          }
        }
      };
      if constexpr (FIRST) {              // the first tile sets the running max from exact maxima
        scores(0, NQ);
        maxima(0, NQ);
        rebase(0, NQ);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) { expo(qi); l_run[qi] += lsum[qi]; }
        pv(0, NQ);
      } else {
        lk[0] = 0.f; lk[1] = 0.f;
        load_k(0);
        qk(0, 0);                                   // R0
        STAR_SCHED_FENCE();
        qk(1, 0); expo_c(0, 0);                     // R1: QK(q1,k0) beside exp(q0,k0)
        hints(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 8>{});
        STAR_SCHED_FENCE();
        load_k(1);
        qk(0, 1); expo_c(1, 0);                     // R2: QK(q0,k1) beside exp(q1,k0)
        hints(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 8>{});
        STAR_SCHED_FENCE();
        qk(1, 1); expo_c(0, 1);                     // R3: QK(q1,k1) beside exp(q0,k1)
        hints(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 8>{});
        if (wave_any(!(lk[0] <= LAZY_BIG))) fix(0);
        l_run[0] += lk[0];
        load_v(0);
        pv_c(0, 0); expo_c(1, 1);                   // R4: PV(q0,k0) beside exp(q1,k1)
        hints(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 10>{});
        if (wave_any(!(lk[1] <= LAZY_BIG))) fix(1);
        l_run[1] += lk[1];
        pv_c(1, 0);                                 // R5: the rest of PV
        load_v(1);
        pv_c(0, 1);
        pv_c(1, 1);
      }
    } else if constexpr (LAZY == 3) {
      // key-half pipeline: the tile is processed as two 32-key halves; the exponentials of one half run beside the
      // MFMAs of the other (QK of half 1 beside exp of half 0, PV of half 0 beside exp of half 1).  K / V fragments are
      // still shared by both query blocks, so LDS traffic is unchanged.
      auto scores_kb = [&](int kb) {
#pragma unroll
        for (int a = 0; a < NQ; ++a) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[a][kb][r] = 0.f;
          s[a][kb] = mfma32<T>(kaug, qaug[a], s[a][kb]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) s[qi][kb] = mfma32<T>(kf, qf[qi][ks], s[qi][kb]);
        }
        if constexpr (MASK) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            if (key >= p.Nk) {
#pragma unroll
              for (int qi = 0; qi < NQ; ++qi) s[qi][kb][r] = -1e30f;
            }
          }
        }
      };
      float lk[NQ];
      auto expo_kb = [&](int kb) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            vec<T, 8> pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(fast_exp2(s[qi][kb][8 * u + e]));
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
              vec<T, 2> a, b2;
              a[0] = pk[e]; a[1] = pk[e + 1]; b2[0] = pk[e + 2]; b2[1] = pk[e + 3];
              ls0 = dot2_ones<T>(a, ls0);
              ls1 = dot2_ones<T>(b2, ls1);
            }
            pf[qi][kb * 2 + u] = pk;
          }
          lk[qi] = ls0 + ls1;
        }
      };
      auto pv_kb = [&](int kb) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const vec<T, 8> vf = load_vt_frag<T>(vbuf, (kb * 2 + u) * 16, db, lane);
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) oacc[qi][db] = mfma32<T>(vf, pf[qi][kb * 2 + u], oacc[qi][db]);
          }
      };
      auto maxima_kb = [&](int kb) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          float mx[2];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int o = g * 8;
            const float a0 = fmaxf(fmaxf(s[qi][kb][o], s[qi][kb][o + 1]), s[qi][kb][o + 2]);
            const float a1 = fmaxf(fmaxf(s[qi][kb][o + 3], s[qi][kb][o + 4]), s[qi][kb][o + 5]);
            mx[g] = fmaxf(fmaxf(a0, a1), fmaxf(s[qi][kb][o + 6], s[qi][kb][o + 7]));
          }
          m_tile[qi] = pair_max(fmaxf(mx[0], mx[1]));
        }
      };
      auto probe = [&]() {
        bool bad = false;
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) bad = bad || !(lk[qi] <= LAZY_BIG);
        return wave_any(bad);
      };
      if constexpr (FIRST) {              // the first tile sets the running max from exact maxima
        scores(0, NQ);
        maxima(0, NQ);
        rebase(0, NQ);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) { expo(qi); l_run[qi] += lsum[qi]; }
        pv(0, NQ);
      } else {
        scores_kb(0);
        scores_kb(1);
        expo_kb(0);
#ifdef STAR_ATTN_V8_HINTS
        for (int i = 0; i < 10; ++i) {      // QK of half 1 beside the exponentials of half 0
          STAR_SCHED_GROUP(0x008, 1, 0);    // 1 MFMA
          STAR_SCHED_GROUP(0x002, 8, 0);    // 8 VALU
        }
#endif
        if (probe()) { scores_kb(0); maxima_kb(0); rebase(0, NQ); expo_kb(0); }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) l_run[qi] += lk[qi];
        pv_kb(0);
        expo_kb(1);
#ifdef STAR_ATTN_V8_HINTS
        for (int i = 0; i < 8; ++i) {       // PV of half 0 beside the exponentials of half 1
          STAR_SCHED_GROUP(0x008, 1, 0);
          STAR_SCHED_GROUP(0x002, 10, 0);
        }
#endif
        if (probe()) { scores_kb(1); maxima_kb(1); rebase(0, NQ); expo_kb(1); }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) l_run[qi] += lk[qi];
        pv_kb(1);
      }
    } else {
      scores(0, NQ);
      if (t == 0) { maxima(0, NQ); rebase(0, NQ); }
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        if (qi == 0) expo(0);
        if (wave_any(!(lsum[qi] <= LAZY_BIG))) {
          scores(qi, qi + 1);
          maxima(qi, qi + 1);
          rebase(qi, qi + 1);
          expo(qi);
        }
        l_run[qi] += lsum[qi];
        pv(qi, qi + 1);                    // PV of this block ...
        if (qi + 1 < NQ) expo(qi + 1);     // ... beside the exponentials of the next one
        if (qi + 1 < NQ) for (int i = 0; i < 8; ++i) {
          STAR_SCHED_GROUP(0x008, 1, 0);   // 1 MFMA
          STAR_SCHED_GROUP(0x100, 2, 0);   // 2 DS reads
          STAR_SCHED_GROUP(0x002, 12, 0);  // 12 VALU
        }
      }
    }
  };

  stage(0, 0);
  const bool has_tail = (p.Nk & (KT - 1)) != 0;
  const int nfull = has_tail ? nt - 1 : nt;
  if constexpr (LAZY >= 3) {
    if (nfull > 0) {
      glds_wait(); block_sync();
      if (1 < nt) stage(1, 1);
      tile(0, std::false_type{}, std::true_type{});
    }
    for (int t = 1; t < nfull; ++t) {
      glds_wait(); block_sync();
      if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
      tile(t, std::false_type{}, std::false_type{});
    }
    if (has_tail) {
      glds_wait(); block_sync();
      if (nt == 1) tile(0, std::true_type{}, std::true_type{});
      else tile(nt - 1, std::true_type{}, std::false_type{});
    }
  } else if constexpr (RING3 == 2) {
    // (bench) key tiles in PAIRS: one barrier + one DMA wait per 128 keys; the next pair lands while this pair is computed.
    // Four 16 KB slots (64 KB per workgroup, two workgroups per CU still fit)
    auto slot = [](int t) { return ((t >> 1) & 1) * 2 + (t & 1); };
    if (nt > 1) stage(1, slot(1));
    for (int t = 0; t < nfull; ++t) {
      if ((t & 1) == 0) {
        glds_wait(); block_sync();
        if (t + 2 < nt) stage(t + 2, slot(t + 2));
        if (t + 3 < nt) stage(t + 3, slot(t + 3));
      }
      tile(t, std::false_type{}, std::false_type{});
    }
    if (has_tail) {
      if (((nt - 1) & 1) == 0) { glds_wait(); block_sync(); }
      tile(nt - 1, std::true_type{}, std::false_type{});
    }
  } else if constexpr (RING3 == 1) {
    if (nt > 1) stage(1, 1);
    for (int t = 0; t < nfull; ++t) {
      // this wave's loads of tile t (4) were issued two tiles ago, those of tile t+1 (4) may stay in flight
      if (t + 1 < nt) { STAR_WAIT_VMCNT(4); } else { STAR_WAIT_VMCNT(0); }
      barrier_keep_dma();
      tile(t, std::false_type{}, std::false_type{});   // issues the LDS-DMA of tile t+2 behind its V reads (slot read in tile t-1)
    }
    if (has_tail) {
      STAR_WAIT_VMCNT(0);
      barrier_keep_dma();
      tile(nt - 1, std::true_type{}, std::false_type{});
    }
  } else {
    for (int t = 0; t < nfull; ++t) {
      if (ABL < 4 || t < 2) { glds_wait(); block_sync(); }
      if (t + 1 < nt && (ABL < 3 || t == 0)) stage(t + 1, (t + 1) & 1);
      tile(t, std::false_type{}, std::false_type{});
    }
    if (has_tail) {
      glds_wait();
      block_sync();
      tile(nt - 1, std::true_type{}, std::false_type{});
    }
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

// ------------------------------------------------------------------------------------------
struct TAttnParams {
  const void* Q; const void* K; const void* V; void* O;
  int ldq, ldk, ldv, ldo;   // row strides (elements) of the [F*HW, *] token matrices
  int F, HW, heads;
  float scale_log2e;
};

// one wavefront per (pixel, head); NB = number of 32-frame blocks (F <= 32 NB; NB = 3 / 4, one wave per SIMD with the whole
// 512-entry register file, cover the 65-128-frame chunks the reference's make_chunks can produce with --max_chunk_len 64)
template <class T, int NB>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256)
temporal_attn_kernel(const TAttnParams p) {
  char* smem = dyn_smem();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, lq = lane & 31;
  const long long item = (long long)blockIdx.x * 4 + wave;
  const long long nitems = (long long)p.HW * p.heads;
  const bool active = item < nitems;
  const long long it = active ? item : nitems - 1;
  const int pix = (int)(it / p.heads), hd = (int)(it % p.heads);
  char* vbuf = smem + wave * (NB * 32 * 128);

  const T* __restrict__ Qg = (const T*)p.Q + hd * 64;
  const T* __restrict__ Kg = (const T*)p.K + hd * 64;
  const T* __restrict__ Vg = (const T*)p.V + hd * 64;
  T* __restrict__ Og = (T*)p.O + hd * 64;

  // ---- stage V (frames x 64) into this wave's LDS slab (swizzled rows), 8 rows per LDS-DMA
#pragma unroll
  for (int j = 0; j < NB * 4; ++j) {
    const int r = j * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int f = r;
    if (f > p.F - 1) f = p.F - 1;
    glds16(Vg + ((size_t)f * p.HW + pix) * p.ldv + c * 8, vbuf + j * 1024);
  }
  // ---- Q / K fragments straight from global
  vec<T, 8> qf[NB][4], kf[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    int f = nb * 32 + lq;
    if (f > p.F - 1) f = p.F - 1;
    const size_t row = (size_t)f * p.HW + pix;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[nb][ks] = *reinterpret_cast<const vec<T, 8>*>(Qg + row * p.ldq + ks * 16 + h2 * 8);
      kf[nb][ks] = *reinterpret_cast<const vec<T, 8>*>(Kg + row * p.ldk + ks * 16 + h2 * 8);
    }
  }
  // ---- S^T[key][q]
  f32x16 sacc[NB][NB];  // [qi][kb]
#pragma unroll
  for (int qi = 0; qi < NB; ++qi)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[qi][kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sacc[qi][kb] = mfma32<T>(kf[kb][ks], qf[qi][ks], sacc[qi][kb]);
    }
  const float c = p.scale_log2e;
  vec<T, 8> pf[NB][NB * 2];
  float linv[NB];
#pragma unroll
  for (int qi = 0; qi < NB; ++qi) {
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
        if (key >= p.F) sacc[qi][kb][r] = -1e30f;
        mx = fmaxf(mx, sacc[qi][kb][r]);
      }
    mx = fmaxf(mx, shfl_xor(mx, 32)) * c;
    float ls = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        vec<T, 8> pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = fast_exp2(sacc[qi][kb][8 * u + e] * c - mx);
          ls += pv;
          pk[e] = from_f32<T>(pv);
        }
        pf[qi][kb * 2 + u] = pk;
      }
    ls += shfl_xor(ls, 32);
    linv[qi] = 1.0f / ls;
  }
  glds_wait();
  block_sync();
  // ---- O^T = V^T P^T
  f32x16 oacc[NB][2];
#pragma unroll
  for (int qi = 0; qi < NB; ++qi)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qi][db][r] = 0.f;
#pragma unroll
  for (int tt = 0; tt < NB * 2; ++tt)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const vec<T, 8> vf = load_vt_frag<T>(vbuf, tt * 16, db, lane);
#pragma unroll
      for (int qi = 0; qi < NB; ++qi) oacc[qi][db] = mfma32<T>(vf, pf[qi][tt], oacc[qi][db]);
    }
#pragma unroll
  for (int qi = 0; qi < NB; ++qi) {
    const int f = qi * 32 + lq;
    const size_t row = (size_t)(f < p.F ? f : p.F - 1) * p.HW + pix;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        uint32_t w0[2], w1[2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          vec<T, 4> o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = from_f32<T>(oacc[qi][db][(2 * a + gg) * 4 + e] * linv[qi]);
          u32x2 pk = __builtin_bit_cast(u32x2, o4);
          if (gg == 0) { w0[0] = pk[0]; w0[1] = pk[1]; } else { w1[0] = pk[0]; w1[1] = pk[1]; }
        }
        const u32x2 s0 = permlane32_swap(w0[0], w1[0]);
        const u32x2 s1 = permlane32_swap(w0[1], w1[1]);
        u32x4 out;
        out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
        if (active && f < p.F) *reinterpret_cast<u32x4*>(Og + row * p.ldo + 32 * db + 16 * a + 8 * h2) = out;
      }
  }
}

}  // namespace star
