// attn7.h -- flash_attn_v7_kernel: the d = 64 flash attention of attn5.h (same arithmetic: operand-swapped S^T = K Q^T, scale and
// running max riding in an augmented half-depth k-step, lazy row maxima with the row sum of P as overflow probe, P^T directly the
// B operand of O^T = V^T P^T, packed 16-bit row sums on long key ranges) restructured for MFMA || VALU overlap INSIDE one wave.
// Replaces xformers.ops.memory_efficient_attention at unet_v2v.py:184-185.
//
// Why: with two 4-wave workgroups per CU the v5 kernel's tile time measures exactly MFMA + VALU (profiles/r02_isa_census.txt,
// DESIGN.md 3.5): the two waves of a SIMD do not overlap one's matrix work with the other's vector work.  What does overlap is
// independent VALU issued by the SAME wave behind an MFMA (<= 5 issues per 32-cycle gap, MI355X_MICROARCH.md "one wave per
// SIMD").  So here ONE wave per SIMD owns NQ x 32 query rows and runs a software pipeline over (key tile, query block) steps:
//     step (t, j):   MFMA stream   QK^T of the NEXT step's block  (2 half-depth + 8 MFMAs)  and  PV of the PREVIOUS step's block (8)
//                    VALU stream   exp2 / pack / row sum / overflow probe of THIS step's block (~70 issues)
// The three are independent by construction (two score blocks and two probability blocks rotate), so the compiler's list
// scheduler -- steered with sched_group_barrier -- places ~4 vector issues behind every MFMA.  K and V^T fragments of a tile
// are read from LDS once per tile and reused by all NQ query blocks (24 LDS reads per 18 NQ MFMAs).
// K/V tiles: ring of four 16 KB slots filled by LDS-DMA two tiles ahead, one workgroup barrier per key tile.
// Built with -mllvm -amdgpu-mfma-vgpr-form (attn7.cpp): with the 512-register budget of one wave per SIMD hipcc otherwise
// selects the AGPR form for every MFMA result and pays a v_accvgpr_read per score (the dropped round-2 kernel).
// The ragged key tail is masked inside the augmented k-step (second slot: K side 1 on keys >= Nk, Q side -30000), so no tile is
// peeled.  Tile 0 takes the exact-maxima path unconditionally (running max not set yet), like v5.
#pragma once
#include "attn.h"
#include "attn5.h"
#include <utility>

namespace star {

#include "attn7_sched.inc"   // V7_SCHED_PK / V7_SCHED_F32: which vector instructions ride in the shadow of which MFMA


// ABL (timing ablations, wrong results by construction): 1 exp -> plain multiply, 2 no fragment refills, 4 no barrier / LDS-DMA, 8 no softmax VALU
template <class T, int NQ, int PKSUM, int ABL = 0>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 1)
flash_attn_v7_kernel(const AttnParams p) {
  constexpr int QW = 32 * NQ, QB = 4 * QW, KT = 64, TILE = KT * 128, SLOT = 2 * TILE, RING = 4;
  constexpr float LAZY_BIG = 1024.0f;
  static_assert(NQ >= 2 && NQ <= 4, "");
  static_assert(PKSUM == 0 || sizeof(T) == 2, "");
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
      STAR_AGPR_PIN(qf[qi][ks]);   // MFMA B operand only: lives in the accumulator half of the register file, read in place
    }
  }
  // augmented half-depth k-step (lanes h2 == 0 hold its slots 0..3):  slot 0: K side 1, Q side -m_run  -> accumulators come out
  // as (scaled score - running max);  slot 1: K side 1 on keys >= Nk (last tile only), Q side -30000  -> masked keys underflow
  vec<T, 4> kaug[2], qaug[NQ];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    kaug[0][e] = kaug[1][e] = from_f32<T>(0.f);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) qaug[qi][e] = from_f32<T>(0.f);
  }
  if (h2 == 0) {
    kaug[0][0] = kaug[1][0] = from_f32<T>(1.0f);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      qaug[qi][0] = from_f32<T>(30000.0f);    // until a block's first exact-maxima pass its probabilities overflow, so the probe fires at tile 0
      qaug[qi][1] = from_f32<T>(-30000.0f);
    }
  }

  // ---- K/V staging: thread (j, tid) copies 16-B chunk (tid & 7) ^ swizzle of tile row r_j = (j*256 + tid) >> 3
  const int pos = tid & 7;
  uint32_t koff[2], voff[2];       // loop-invariant byte offsets of this lane's two chunks from the tile's first row
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (j * 256 + tid) >> 3;
    const int c = pos ^ ((r >> 1) & 7);
    koff[j] = (uint32_t)(r * p.ldk + c * 8) * 2u;
    voff[j] = (uint32_t)(r * p.ldv + c * 8) * 2u;
  }
  const size_t kstep = (size_t)KT * p.ldk * 2, vstep = (size_t)KT * p.ldv * 2;   // bytes per key tile
  const int nt = (p.Nk + KT - 1) / KT;
  const bool has_tail = (p.Nk & (KT - 1)) != 0;
  const int nfull = has_tail ? nt - 1 : nt;
  auto stage = [&](int t) STAR_ALWAYS_INLINE {
    char* kdst = smem + (t & (RING - 1)) * SLOT;
    char* vdst = kdst + TILE;
    if (t < nfull) {
      const char* kt = (const char*)Kg + (size_t)t * kstep;   // wave-uniform
      const char* vt = (const char*)Vg + (size_t)t * vstep;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        glds16_su(kt, koff[j], kdst + (size_t)(j * 256 + wv * 64) * 16);
        glds16_su(vt, voff[j], vdst + (size_t)(j * 256 + wv * 64) * 16);
      }
    } else {                       // the ragged last tile: rows past Nk re-read the last key (masked in the augmented k-step)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = (j * 256 + tid) >> 3;
        const int c = pos ^ ((r >> 1) & 7);
        int key = t * KT + r;
        if (key > p.Nk - 1) key = p.Nk - 1;
        glds16_v(Kg + (size_t)key * p.ldk + c * 8, kdst + (size_t)(j * 256 + wv * 64) * 16);   // asm form: see prim.h
        glds16_v(Vg + (size_t)key * p.ldv + c * 8, vdst + (size_t)(j * 256 + wv * 64) * 16);
      }
    }
  };
  auto set_tail_mask = [&]() STAR_ALWAYS_INLINE {   // K side of the mask slot for the last (ragged) tile
    if (h2 == 0) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kaug[kb][1] = from_f32<T>((nt - 1) * KT + kb * 32 + lq >= p.Nk ? 1.0f : 0.0f);
    }
  };

  // ---- loop-invariant fragment addresses (bytes within ring slot 0); see attn5.h
  const char* kfo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kfo[ks] = opaque(smem + swz_off(lq, ks * 2 + h2));
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
  vec<T, 8> kf[4][2], vf[4][2];    // the K / V^T fragments of the tile the MFMA stream is working on (64 registers)
  auto load_k = [&](int t) STAR_ALWAYS_INLINE {
    const int so = (t & (RING - 1)) * SLOT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kf[ks][kb] = *reinterpret_cast<const vec<T, 8>*>(kfo[ks] + so + kb * 4096);
  };
  auto load_v = [&](int t) STAR_ALWAYS_INLINE {
    const int so = (t & (RING - 1)) * SLOT;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const vec<T, 4> r = lds_read_tr<T>(vfo[half][db] + so + tt * 2048);
#pragma unroll
          for (int e = 0; e < 4; ++e) vf[tt][db][half * 4 + e] = r[e];
        }
  };

  f32x16 oacc[NQ][2];
  float m_run[NQ], l_run[NQ];      // m_run is always exactly representable in T (it is fed to the MFMA through Q_aug)
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    m_run[qi] = 0.f; l_run[qi] = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qi][c2][r] = 0.f;
  }
  f32x16 s[2][2];                  // two score blocks (32 queries x 64 keys each) in rotation
  vec<T, 8> pf[2][4];              // two probability blocks in rotation
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { pf[1][tt][e] = from_f32<T>(0.f); vf[tt][0][e] = vf[tt][1][e] = from_f32<T>(0.f); }
  }

  auto qk = [&](f32x16 (&dst)[2], int qi_c, const vec<T, 8> (&kfr)[4][2], const vec<T, 4> (&ka)[2]) STAR_ALWAYS_INLINE {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[kb][r] = 0.f;
      dst[kb] = mfma32_k8<T>(ka[kb], qaug[qi_c], dst[kb]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) dst[kb] = mfma32<T>(kfr[ks][kb], qf[qi_c][ks], dst[kb]);
  };
  // exponentials, packing and row sum of one score block, un-pipelined (the rare exact-maxima path)
  auto expo = [&](const f32x16 (&sc)[2], vec<T, 8> (&pd)[4]) STAR_ALWAYS_INLINE -> float {
    float ex[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) ex[n] = fast_exp2(sc[n >> 4][n & 15]);
#pragma unroll
    for (int n = 0; n < 32; ++n) pd[n >> 3][n & 7] = from_f32<T>(ex[n]);
    if constexpr (PKSUM != 0) {
      vec<T, 2> h[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { h[i][0] = pd[i >> 2][2 * (i & 3)]; h[i][1] = pd[i >> 2][2 * (i & 3) + 1]; }
#pragma unroll
      for (int n = 8; n >= 1; n >>= 1)
#pragma unroll
        for (int i = 0; i < n; ++i) h[i] = pk_add<T>(h[2 * i], h[2 * i + 1]);
      return to_f32<T>(h[0][0]) + to_f32<T>(h[0][1]);
    } else {
      float c[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int n = 0; n < 32; ++n) c[n & 3] += ex[n];
      return (c[0] + c[1]) + (c[2] + c[3]);
    }
  };

  // one pipeline step: block (t, J) is exponentiated while the MFMAs work on the blocks either side of it.  TP = parity of t
  // (compile time: the score / probability slot of a step is (t * NQ + J) & 1).  The step is written as 18 chunks, one per MFMA,
  // each closed by a scheduling fence: chunk k = MFMA k + the vector instructions the generated table puts in its shadow + the
  // LDS fragment read that refills the operand registers MFMA k - 1 has just consumed.
  auto step = [&](int t, auto tptag, auto jtag) STAR_ALWAYS_INLINE {
    constexpr int TP = decltype(tptag)::value, J = decltype(jtag)::value;
    constexpr int CUR = (TP * NQ + J) & 1, NXT = CUR ^ 1;
    constexpr int JN = (J + 1) % NQ, JP = (J + NQ - 1) % NQ;
    const int so_k = ((t + 1 < nt ? t + 1 : nt - 1) & (RING - 1)) * SLOT;   // K of tile t + 1: QK of (t + 1, 0) is issued in step (t, NQ - 1)
    const int so_v = (t & (RING - 1)) * SLOT;                                // V of tile t: PV of (t, 0) is issued in step (t, 1)
    STAR_WAIT_LGKM0();   // fragment reads of the previous step landed long ago; tell the compiler so before this step issues its own
    STAR_SCHED_FENCE();
    float ex[32];
    vec<T, 2> nd[15], pc[16];
    float chn[4] = {0.f, 0.f, 0.f, 0.f};
    float f0 = 0.f, f1 = 0.f, lsum = 0.f;
    uint64_t probe = 0;
    auto refill = [&](auto mtag) STAR_ALWAYS_INLINE {   // MFMA M has issued: its A-operand fragment registers take the next tile's data
      constexpr int M = decltype(mtag)::value;
      if constexpr ((ABL & 2) != 0) return;
      if constexpr (J == NQ - 2 && M >= 2 && M < 10) {
        constexpr int ks = (M - 2) >> 1, kb = (M - 2) & 1;
        kf[ks][kb] = *reinterpret_cast<const vec<T, 8>*>(kfo[ks] + so_k + kb * 4096);
      }
      if constexpr (J == 0 && M >= 10) {
        constexpr int tt = (M - 10) >> 1, db = (M - 10) & 1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const vec<T, 4> r = lds_read_tr<T>(vfo[half][db] + so_v + tt * 2048);
#pragma unroll
          for (int e = 0; e < 4; ++e) vf[tt][db][half * 4 + e] = r[e];
        }
      }
    };
    static_for<18>([&](auto ktag) STAR_ALWAYS_INLINE {
      constexpr int K = decltype(ktag)::value;
      // ---- the MFMA of this chunk
      if constexpr (K < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[NXT][K][r] = 0.f;
        s[NXT][K] = pin_here(mfma32_k8<T>(kaug[K], qaug[JN], s[NXT][K]));
      } else if constexpr (K < 10) {
        constexpr int ks = (K - 2) >> 1, kb = (K - 2) & 1;
        s[NXT][kb] = pin_here(mfma32<T>(kf[ks][kb], qf[JN][ks], s[NXT][kb]));
      } else {
        constexpr int tt = (K - 10) >> 1, db = (K - 10) & 1;
        oacc[JP][db] = pin_here(mfma32<T>(vf[tt][db], pf[NXT][tt], oacc[JP][db]));
      }
      // ---- vector instructions in its shadow (softmax of the CURRENT block)
      constexpr int W = PKSUM ? V7_SCHED_PK_W : V7_SCHED_F32_W;
      static_for<W>([&](auto wtag) STAR_ALWAYS_INLINE {
        constexpr int op = PKSUM ? V7_SCHED_PK[K][decltype(wtag)::value < V7_SCHED_PK_W ? decltype(wtag)::value : 0]
                                 : V7_SCHED_F32[K][decltype(wtag)::value < V7_SCHED_F32_W ? decltype(wtag)::value : 0];
        constexpr int kind = op >> 8, ix = op & 255;
        if constexpr ((ABL & 8) != 0) { }
        else if constexpr (kind == 1) ex[ix] = pin_here((ABL & 1) ? s[CUR][ix >> 4][ix & 15] * 1.0009765f : fast_exp2(s[CUR][ix >> 4][ix & 15]));
        else if constexpr (kind == 2) {
          vec<T, 2> v;
          v[0] = from_f32<T>(ex[2 * ix]);
          v[1] = from_f32<T>(ex[2 * ix + 1]);
          pc[ix] = pin_here(v);   // the pack is issued HERE (otherwise instruction selection gathers the four packs of a fragment at its last element)
        } else if constexpr (kind == 3) {
          if constexpr (ix < 4) nd[ix] = pin_here(pk_add<T>(pc[ix], pc[ix + 4]));             // four running chains ...
          else if constexpr (ix < 8) nd[ix] = pin_here(pk_add<T>(nd[ix - 4], pc[ix + 4]));
          else if constexpr (ix < 12) nd[ix] = pin_here(pk_add<T>(nd[ix - 4], pc[ix + 4]));
          else if constexpr (ix == 12) nd[12] = pin_here(pk_add<T>(nd[8], nd[9]));            // ... combined at the end
          else if constexpr (ix == 13) nd[13] = pin_here(pk_add<T>(nd[10], nd[11]));
          else nd[14] = pin_here(pk_add<T>(nd[12], nd[13]));
        } else if constexpr (kind == 4) {
          if constexpr (ix == 0) f0 = pin_here(to_f32<T>(nd[14][0]));
          else if constexpr (ix == 1) f1 = pin_here(to_f32<T>(nd[14][1]));
          else lsum = pin_here(f0 + f1);
        } else if constexpr (kind == 5) chn[ix & 3] = pin_here(chn[ix & 3] + ex[ix]);
        else if constexpr (kind == 6) {
          if constexpr (ix == 0) f0 = pin_here(chn[0] + chn[1]);
          else if constexpr (ix == 1) f1 = pin_here(chn[2] + chn[3]);
          else lsum = pin_here(f0 + f1);
        } else if constexpr (kind == 7) probe = wave_ballot(!(lsum <= LAZY_BIG));   // compare issued here, branch behind the last MFMA
      });
      // ---- operand refill behind the previous MFMA
      if constexpr (K >= 1) refill(std::integral_constant<int, K - 1>{});
      STAR_SCHED_FENCE();
    });
    refill(std::integral_constant<int, 17>{});
#pragma unroll
    for (int i = 0; i < 16; ++i) { if constexpr ((ABL & 8) == 0) { pf[CUR][i >> 2][2 * (i & 3)] = pc[i][0]; pf[CUR][i >> 2][2 * (i & 3) + 1] = pc[i][1]; } }
    if (probe != 0) {                // some P is large (or overflowed; always at tile 0, see the prologue): exact maxima from
      vec<T, 8> kt[4][2];            // recomputed scores of THIS block (K tile t is still in its ring slot), then redo
      const int zz = opaque_int(0);  // everything below hangs on this: hipcc otherwise hoists the rare path's address and mask arithmetic above the branch, into every step
      const int so = (t & (RING - 1)) * SLOT + zz;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) kt[ks][kb] = *reinterpret_cast<const vec<T, 8>*>(kfo[ks] + so + kb * 4096);
      vec<T, 4> ka[2];               // K side of the augmented step for tile t (kaug may already carry tile t + 1's tail mask)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ka[kb][e] = from_f32<T>(0.f);
        if (h2 == 0) {
          ka[kb][0] = from_f32<T>(1.0f);
          ka[kb][1] = from_f32<T>(t * KT + kb * 32 + lq + zz >= p.Nk ? 1.0f : 0.0f);
        }
      }
      if (t == 0 && h2 + zz == 0) qaug[J][0] = from_f32<T>(0.f);   // tile 0: scores relative to the initial running max 0
      qk(s[CUR], J, kt, ka);
      float mx[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kb = g >> 1, o = (g & 1) * 8;
        const float a0 = fmaxf(fmaxf(s[CUR][kb][o], s[CUR][kb][o + 1]), s[CUR][kb][o + 2]);
        const float a1 = fmaxf(fmaxf(s[CUR][kb][o + 3], s[CUR][kb][o + 4]), s[CUR][kb][o + 5]);
        mx[g] = fmaxf(fmaxf(a0, a1), fmaxf(s[CUR][kb][o + 6], s[CUR][kb][o + 7]));
      }
      const float m_tile = pair_max(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])));
      const float inc = (t == 0) ? m_tile : fmaxf(m_tile, 0.f);
      const float m_new = to_f32<T>(from_f32<T>(m_run[J] + inc));
      const float delta = m_new - m_run[J];
      const float alpha = fast_exp2(-delta);
      m_run[J] = m_new;
      l_run[J] *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[J][db][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[CUR][kb][r] -= delta;
      if (h2 + zz == 0) qaug[J][0] = from_f32<T>(-m_new);
      lsum = expo(s[CUR], pf[CUR]);
    }
    l_run[J] += lsum;
  };
  // all NQ steps of key tile t
  auto tile = [&](int t, auto tptag) STAR_ALWAYS_INLINE {
    using std::integral_constant;
    // tile t + 1 has landed for everybody, and slot (t + 2) % 4 = slot of tile t - 2 is no longer read by anybody
    if constexpr ((ABL & 4) == 0) {
      glds_wait(); block_sync();
      if (t + 2 < nt) stage(t + 2);
    }
    step(t, tptag, integral_constant<int, 0>{});
    if constexpr (NQ >= 3) step(t, tptag, integral_constant<int, 1>{});
    if constexpr (NQ >= 4) step(t, tptag, integral_constant<int, 2>{});
    if (has_tail && t + 1 == nt - 1) set_tail_mask();   // the QK of a tile's last step is tile t + 1's
    step(t, tptag, integral_constant<int, NQ - 1>{});
  };

  // ---- prologue: tiles 0 and 1 in flight, scores of block (0, 0)
  glds_wait();                      // the Q loads are done before the first hand-written LDS-DMA (prim.h: glds16_su)
  stage(0);
  if (nt > 1) stage(1);
  if (has_tail && nt == 1) set_tail_mask();
  glds_wait(); block_sync();
  load_k(0);
  qk(s[0], 0, kf, kaug);
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    tile(t + 1, std::integral_constant<int, 1>{});
  }
  if (t < nt) tile(t, std::integral_constant<int, 0>{});
  // drain: PV of the last block (its probabilities sit in slot ((nt - 1) * NQ + NQ - 1) & 1)
  if (((nt - 1) * NQ + NQ - 1) & 1) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db) oacc[NQ - 1][db] = mfma32<T>(vf[tt][db], pf[1][tt], oacc[NQ - 1][db]);
  } else {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db) oacc[NQ - 1][db] = mfma32<T>(vf[tt][db], pf[0][tt], oacc[NQ - 1][db]);
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
