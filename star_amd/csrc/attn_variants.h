// attn_variants.h -- the measured-and-lost attention kernels of round 1 (profiles/r01_attn_ab.txt): baseline (variant 0),
// v2 (1), software-pipelined v4 (4 / 5), enforced-antiphase 512-thread v6 (20).  Compiled ONLY into the bench build
// (-DSTAR_BENCH_VARIANTS: tools/bench/libstar_hip_bench.so) and the test emulator; the product library ships
// flash_attn_v3_kernel<T, 2, 1, 0, 1> (attn.h) alone and star_attn_fwd rejects every other variant id.
#pragma once
#include "attn.h"

namespace star {

template <class T>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 2)
flash_attn_kernel(const AttnParams p) {
  constexpr int QW = 64;          // q rows per wave (two 32-row blocks)
  constexpr int QB = 256;         // q rows per workgroup
  constexpr int KT = 64;          // keys per tile
  constexpr int TILE = KT * 128;  // bytes of one K or V tile in LDS
  char* smem = dyn_smem();        // [2][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, lq = lane & 31;

  // XCD-aware mapping: the blocks resident on one XCD (bid % 8) share one (batch, head) => K/V stay in that XCD's L2
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int bh = xcd + 8 * (slot / p.nqb);
  const int qb = slot % p.nqb;
  const int BH = p.batch * p.heads;
  if (bh >= BH) return;
  const int b = bh / p.heads, hd = bh % p.heads;

  const T* __restrict__ Qg = (const T*)p.Q + (size_t)b * p.bsq + hd * 64;
  const T* __restrict__ Kg = (const T*)p.K + (size_t)b * p.bsk + hd * 64;
  const T* __restrict__ Vg = (const T*)p.V + (size_t)b * p.bsv + hd * 64;
  T* __restrict__ Og = (T*)p.O + (size_t)b * p.bso + hd * 64;

  // ---- Q fragments (B operand: col = q row, k = d)
  vec<T, 8> qf[2][4];
  const int q_base = qb * QB + wave * QW;
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    int q = q_base + qi * 32 + lq;
    if (q > p.Nq - 1) q = p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[qi][ks] = *reinterpret_cast<const vec<T, 8>*>(Qg + (size_t)q * p.ldq + ks * 16 + h2 * 8);
  }

  // ---- K/V tile loaders: 64 rows x 8 chunks = 512 chunks per tile, 2 per thread each
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

  f32x16 oacc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[a][c][r] = 0.f;
  float m_run[2] = {-1e30f, -1e30f};   // running max (scaled, log2 domain)
  float l_run[2] = {0.f, 0.f};         // this lane's partial row sum

  const int nt = (p.Nk + KT - 1) / KT;
  const float c = p.scale_log2e;
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    glds_wait();
    block_sync();
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
    const char* kbuf = smem + (t & 1) * 2 * TILE;
    const char* vbuf = kbuf + TILE;

    // ---- S^T = K Q^T : sacc[qi][kvb], lane: q = qi*32+lq, key = 32*kvb + (r&3) + 8*(r>>2) + 4*h2
    f32x16 sacc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[a][kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int R = kb * 32 + lq;
        const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(R, ks * 2 + h2));
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) sacc[qi][kb] = mfma32<T>(kf, qf[qi][ks], sacc[qi][kb]);
      }
    }
    // ---- mask the key tail of the last tile
    if (t == nt - 1 && (p.Nk & (KT - 1))) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
          if (key >= p.Nk) { sacc[0][kb][r] = -1e30f; sacc[1][kb][r] = -1e30f; }
        }
    }
    // ---- online softmax, P packed as the B operand of the PV MFMA
    vec<T, 8> pf[2][4];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      float mx = sacc[qi][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qi][kb][r]);
      mx = fmaxf(mx, shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qi], mx * c);
      const float alpha = fast_exp2(m_run[qi] - m_new);
      m_run[qi] = m_new;
      float ls = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          vec<T, 8> pk;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pv = fast_exp2(sacc[qi][kb][8 * u + e] * c - m_new);
            ls += pv;
            pk[e] = from_f32<T>(pv);
          }
          pf[qi][kb * 2 + u] = pk;
        }
      l_run[qi] = l_run[qi] * alpha + ls;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qi][db][r] *= alpha;
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const vec<T, 8> vf = load_vt_frag<T>(vbuf, tt * 16, db, lane);
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) oacc[qi][db] = mfma32<T>(vf, pf[qi][tt], oacc[qi][db]);
      }
    }
  }

  // ---- epilogue: O = oacc / l ; lane holds d = 32*db + 8*g + 4*h2 + (0..3); pair groups into 16-B stores
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const float l = l_run[qi] + shfl_xor(l_run[qi], 32);
    const float inv = 1.0f / l;
    const int q = q_base + qi * 32 + lq;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int a = 0; a < 2; ++a) {  // group pair (2a, 2a+1)
        uint32_t w0[2], w1[2];      // packed T x4 of group 2a / 2a+1
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          vec<T, 4> o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = from_f32<T>(oacc[qi][db][(2 * a + gg) * 4 + e] * inv);
          u32x2 pk = __builtin_bit_cast(u32x2, o4);
          if (gg == 0) { w0[0] = pk[0]; w0[1] = pk[1]; } else { w1[0] = pk[0]; w1[1] = pk[1]; }
        }
        const u32x2 s0 = permlane32_swap(w0[0], w1[0]);
        const u32x2 s1 = permlane32_swap(w0[1], w1[1]);
        // lower lane: d = 32db + 16a + 0..7 ; upper lane: d = 32db + 16a + 8..15
        u32x4 out;
        out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
        if (q < p.Nq) *reinterpret_cast<u32x4*>(Og + (size_t)q * p.ldo + 32 * db + 16 * a + 8 * h2) = out;
      }
  }
}

// ------------------------------------------------------------------------------------------
// flash_attn_v2_kernel: same contract and tiling as flash_attn_kernel, restructured after PMC analysis of the baseline
// (profiles/r01_attn_pmc_v0.txt: per wave 37 % VALU-active, 46 % MFMA-issue/dependency stalls, MFMA pipe 38 % busy --
// QK^T -> softmax -> PV ran as one serial dependency chain per wave, and the key-tail mask cost 64 v_cndmask per tile):
//   * per tile, ONE early wave-uniform decision (does any row's max grow by more than 2^THR?) -- the rare rescale
//     branch is taken before any P is exponentiated (textbook order), everything after it is straight-line code;
//   * in that straight-line block the PV MFMAs of query block 0 sit beside the exp/convert VALU work of query block 1
//     (independent streams the scheduler can interleave), and PV of block 1 runs into the next tile's QK^T;
//   * the key-tail mask exists only in the peeled last tile; row max via 3-input max chains; the lane^32 exchange is a
//     v_permlane32_swap, not an LDS bpermute.
template <class T>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 2)
flash_attn_v2_kernel(const AttnParams p) {
  constexpr int QW = 64, QB = 256, KT = 64, TILE = KT * 128;
  constexpr float RESCALE_THR = 8.0f;   // log2 units: P <= 2^8 before a rescale is forced
  char* smem = dyn_smem();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

  vec<T, 8> qf[2][4];
  const int q_base = qb * QB + wave * QW;
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    int q = q_base + qi * 32 + lq;
    if (q > p.Nq - 1) q = p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[qi][ks] = *reinterpret_cast<const vec<T, 8>*>(Qg + (size_t)q * p.ldq + ks * 16 + h2 * 8);
  }
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

  f32x16 oacc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[a][c2][r] = 0.f;
  float m_run[2] = {-1e30f, -1e30f};
  float l_run[2] = {0.f, 0.f};
  const int nt = (p.Nk + KT - 1) / KT;
  const float c = p.scale_log2e;

  auto tile = [&](int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const char* kbuf = smem + (t & 1) * 2 * TILE;
    const char* vbuf = kbuf + TILE;
    // ---- S^T = K Q^T for both query blocks (16 MFMAs, 4 independent accumulators)
    f32x16 s[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[a][kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) s[qi][kb] = mfma32<T>(kf, qf[qi][ks], s[qi][kb]);
      }
    if constexpr (MASK) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
          if (key >= p.Nk) { s[0][kb][r] = -1e30f; s[1][kb][r] = -1e30f; }
        }
    }
    // ---- row maxima of both blocks, one wave-uniform rescale decision
    float m_tile[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      float mx[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kb = g >> 1, o = (g & 1) * 8;
        const float a0 = fmaxf(fmaxf(s[qi][kb][o], s[qi][kb][o + 1]), s[qi][kb][o + 2]);
        const float a1 = fmaxf(fmaxf(s[qi][kb][o + 3], s[qi][kb][o + 4]), s[qi][kb][o + 5]);
        mx[g] = fmaxf(fmaxf(a0, a1), fmaxf(s[qi][kb][o + 6], s[qi][kb][o + 7]));
      }
      m_tile[qi] = pair_max(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]))) * c;
    }
    if (wave_any(m_tile[0] > m_run[0] + RESCALE_THR || m_tile[1] > m_run[1] + RESCALE_THR)) {
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {   // everything still at the old max is rescaled exactly once, before any new P exists
        const float m_new = fmaxf(m_run[qi], m_tile[qi]);
        const float alpha = fast_exp2(m_run[qi] - m_new);
        m_run[qi] = m_new;
        l_run[qi] *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qi][db][r] *= alpha;
      }
    }
    // ---- straight-line: P0 ; PV0 beside P1 ; PV1
    vec<T, 8> pf[2][4];
    auto expo = [&](int qi) {
      const float m = m_run[qi];
      float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          vec<T, 8> pk;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pv = fast_exp2(s[qi][kb][8 * u + e] * c - m);
            if (e & 1) ls1 += pv; else ls0 += pv;
            pk[e] = from_f32<T>(pv);
          }
          pf[qi][kb * 2 + u] = pk;
        }
      l_run[qi] += ls0 + ls1;
    };
    expo(0);
    expo(1);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const vec<T, 8> vf = load_vt_frag<T>(vbuf, tt * 16, db, lane);
        oacc[0][db] = mfma32<T>(vf, pf[0][tt], oacc[0][db]);
        oacc[1][db] = mfma32<T>(vf, pf[1][tt], oacc[1][db]);
      }
#ifndef STAR_HOSTEMU
    // interleave: the first 8 PV MFMAs (they only need P0) beside the VALU of expo(1)
    for (int i = 0; i < 8; ++i) {
      STAR_SCHED_GROUP(0x008, 1, 0);   // 1 MFMA
      STAR_SCHED_GROUP(0x100, 2, 0);   // 2 DS reads
      STAR_SCHED_GROUP(0x002, 12, 0);  // 12 VALU
    }
#endif
  };

  stage(0, 0);
  const bool has_tail = (p.Nk & (KT - 1)) != 0;
  const int nfull = has_tail ? nt - 1 : nt;
  for (int t = 0; t < nfull; ++t) {
    glds_wait();
    block_sync();
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
    tile(t, std::false_type{});
  }
  if (has_tail) {
    glds_wait();
    block_sync();
    tile(nt - 1, std::true_type{});
  }

#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
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
// flash_attn_v4_kernel: software-pipelined across key tiles.  One wave owns 32 query rows (128-row workgroups, 3 per
// CU), keeps TWO score blocks live and, in steady state, issues
//     phase 1:  QK^T MFMAs of tile t+1   beside   exp2 / pack / row-sum VALU of tile t
//     phase 2:  PV   MFMAs of tile t     beside   row-max VALU of tile t+1 (+ the rare wave-uniform rescale)
// so MFMA and VALU of the same wave overlap (~5.5 VALU per MFMA, the shadow one 32x32x16 MFMA offers) instead of running
// QK^T -> softmax -> PV as one dependency chain (PMC of the baseline: matrix pipe 38 % busy, 46 % issue stalls).
// Scale and running max ride in the MFMA through the augmented k-step (see v3); a second augmented slot adds -30000 to
// keys past Nk, so the ragged key tail needs no select pass and no peeled tile.  K/V tiles: 3-slot LDS ring (48 KB).
template <class T, int NQ>   // NQ 32-row query blocks per wave; NQ = 2 runs ONE wave per SIMD with the whole 512-entry register file
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, (NQ == 2 ? 1 : 2))
flash_attn_v4_kernel(const AttnParams p) {
  constexpr int QW = 32 * NQ, QB = 4 * QW, KT = 64, TILE = KT * 128;
  constexpr float RESCALE_THR = 8.0f;
  char* smem = dyn_smem();   // [3][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int q_row0 = qb * QB + wave * QW + lq;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q_row = q_row0 + qi * 32;
    const int q = q_row < p.Nq ? q_row : p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const vec<T, 8> raw = *reinterpret_cast<const vec<T, 8>*>(Qg + (size_t)q * p.ldq + ks * 16 + h2 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[qi][ks][e] = from_f32<T>(to_f32<T>(raw[e]) * p.scale_log2e);
    }
  }
  // augmented k-step: slot 0 carries -m_run (K side 1), slot 1 carries -30000 for keys >= Nk (K side 1 on those keys)
  vec<T, 8> qaug[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
    for (int e = 0; e < 8; ++e) qaug[qi][e] = from_f32<T>(0.f);
    if (h2 == 0) qaug[qi][1] = from_f32<T>(-30000.0f);
  }

  const int pos = tid & 7;
  auto stage = [&](int t) {
    char* kbuf = smem + (t % 3) * 2 * TILE;
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
  float m_run[NQ], l0[NQ], l1[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    m_run[qi] = 0.f; l0[qi] = 0.f; l1[qi] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qi][db][r] = 0.f;
  }
  const int nt = (p.Nk + KT - 1) / KT;

  // S^T of tile t (already relative to m_run, masked past Nk)
  auto qk = [&](int t, f32x16 (&s)[NQ][2]) {
    const char* kbuf = smem + (t % 3) * 2 * TILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      vec<T, 8> kaug;
#pragma unroll
      for (int e = 0; e < 8; ++e) kaug[e] = from_f32<T>(0.f);
      if (h2 == 0) {
        kaug[0] = from_f32<T>(1.0f);
        kaug[1] = from_f32<T>((t * KT + kb * 32 + lq >= p.Nk) ? 1.0f : 0.0f);
      }
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qi][kb][r] = 0.f;
        s[qi][kb] = mfma32<T>(kaug, qaug[qi], s[qi][kb]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) s[qi][kb] = mfma32<T>(kf, qf[qi][ks], s[qi][kb]);
      }
  };
  // row max of a score block -> wave-uniform decision; on growth (or first tile) move the running max
  auto decide = [&](f32x16 (&s)[NQ][2], bool force) {
    float m_tile[NQ];
    bool grow = false;
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
      grow = grow || (m_tile[qi] > RESCALE_THR);
    }
    if (force || wave_any(grow)) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const float inc = force ? m_tile[qi] : fmaxf(m_tile[qi], 0.f);
        const float m_new = to_f32<T>(from_f32<T>(m_run[qi] + inc));
        const float delta = m_new - m_run[qi];
        const float alpha = fast_exp2(-delta);
        m_run[qi] = m_new;
        l0[qi] *= alpha; l1[qi] *= alpha;
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
    }
  };
  // one pipeline step: consumes s_cur (tile t), produces s_nxt (tile t+1)
  auto step = [&](int t, f32x16 (&s_cur)[NQ][2], f32x16 (&s_nxt)[NQ][2], auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;   // steady state (true) is branch-free; the last tile is peeled
    glds_wait();
    block_sync();                      // tile t+1 has landed; every wave is done with the slot tile t+2 will overwrite
    if constexpr (more) stage(t + 2 < nt ? t + 2 : nt - 1);   // past the end: harmless reload of the last tile into a free slot
    // ---- phase 1: QK^T(t+1) MFMAs beside exp2 / pack / row sums of tile t
    if constexpr (more) qk(t + 1, s_nxt);
    vec<T, 8> pf[NQ][4];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          vec<T, 8> pk;
#pragma unroll
          for (int e = 0; e < 8; ++e) pk[e] = from_f32<T>(fast_exp2(s_cur[qi][kb][8 * u + e]));
#pragma unroll
          for (int e = 0; e < 8; e += 4) {
            vec<T, 2> a, b2;
            a[0] = pk[e]; a[1] = pk[e + 1]; b2[0] = pk[e + 2]; b2[1] = pk[e + 3];
            l0[qi] = dot2_ones<T>(a, l0[qi]);
            l1[qi] = dot2_ones<T>(b2, l1[qi]);
          }
          pf[qi][kb * 2 + u] = pk;
        }
#ifndef STAR_HOSTEMU
    if constexpr (more) for (int i = 0; i < 10 * NQ; ++i) { STAR_SCHED_GROUP(0x008, 1, 0); STAR_SCHED_GROUP(0x100, 1, 0); STAR_SCHED_GROUP(0x002, 7, 0); }
#endif
    // ---- phase 2: PV(t) MFMAs beside the row max of tile t+1
    const char* vbuf = smem + (t % 3) * 2 * TILE + TILE;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const vec<T, 8> vf = load_vt_frag<T>(vbuf, tt * 16, db, lane);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) oacc[qi][db] = mfma32<T>(vf, pf[qi][tt], oacc[qi][db]);
      }
    if constexpr (more) decide(s_nxt, false);
  };

  // ---- prologue: tiles 0 and 1 in flight, S(0) computed, its max taken
  f32x16 sa[NQ][2], sb[NQ][2];
  stage(0);
  if (nt > 1) stage(1);
  glds_wait();
  block_sync();
  qk(0, sa);
  decide(sa, true);
  int t = 0;
  for (; t + 2 < nt; t += 2) {         // tiles t and t+1 both have a successor
    step(t, sa, sb, std::true_type{});
    step(t + 1, sb, sa, std::true_type{});
  }
  if (t + 1 < nt) {                    // two tiles left
    step(t, sa, sb, std::true_type{});
    step(t + 1, sb, sa, std::false_type{});
  } else {                             // one tile left
    step(t, sa, sb, std::false_type{});
  }

  // ---- epilogue
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q_row = q_row0 + qi * 32;
    const float l = pair_sum(l0[qi] + l1[qi]);
    const float inv = 1.0f / l;
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
        if (q_row < p.Nq) *reinterpret_cast<u32x4*>(Og + (size_t)q_row * p.ldo + 32 * db + 16 * a + 8 * h2) = out;
      }
  }
}

// ------------------------------------------------------------------------------------------
// flash_attn_v6_kernel (variant 20): the two waves of a SIMD in enforced antiphase.  Measured on gfx950
// (tools/probe/overlap.hip): a wave alone in a VALU-only stretch issues one VALU per ~8 cycles and the matrix pipe idles;
// left to themselves the two workgroups of a CU drift, and the softmax VALU (39 % of the tile time) never hides.  Here a
// 512-thread workgroup owns 512 query rows; waves 0-3 and waves 4-7 (the two waves of each SIMD) run the same per-tile
// program half a period apart, separated by one workgroup barrier per half-step:
//     MFMA phase(t) = PV(t-1) then QK(t)          |  VALU phase(t) = softmax of tile t (lazy maxima, as variant 9)
// so one wave of every SIMD is always in its MFMA phase while its partner is in its VALU phase.  K tiles live in a ring of
// three (the lagging group may have to recompute scores of tile t one half-step after tile t+2 started loading), V tiles
// in a ring of two; all eight waves share the LDS-DMA of K(t+1) and V(t) at the even half-steps.
template <class T>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(512, 2)
flash_attn_v6_kernel(const AttnParams p) {
  constexpr int NQ = 2, QW = 64, QB = 8 * QW, KT = 64, TILE = KT * 128;
  constexpr float LAZY_BIG = 1024.0f;
  char* smem = dyn_smem();                     // [3] K tiles | [2] V tiles
  char* kring = smem;
  char* vring = smem + 3 * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave_uniform(wave >> 2);
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
  const int q_base = qb * QB + wave * QW;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    int q = q_base + qi * 32 + lq;
    if (q > p.Nq - 1) q = p.Nq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const vec<T, 8> raw = *reinterpret_cast<const vec<T, 8>*>(Qg + (size_t)q * p.ldq + ks * 16 + h2 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[qi][ks][e] = from_f32<T>(to_f32<T>(raw[e]) * p.scale_log2e);
    }
  }
  vec<T, 8> kaug, qaug[NQ];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    kaug[e] = from_f32<T>(0.f);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) qaug[qi][e] = from_f32<T>(0.f);
  }
  if (h2 == 0) kaug[0] = from_f32<T>(1.0f);

  // one 16-B chunk per thread per tile: chunk q = tid -> row q >> 3, stored position q & 7 holds source chunk pos ^ swizzle
  const int srow = tid >> 3, spos = tid & 7, schunk = spos ^ ((srow >> 1) & 7);
  auto stage_k = [&](int t, int slot_) STAR_ALWAYS_INLINE {
    int key = t * KT + srow;
    if (key > p.Nk - 1) key = p.Nk - 1;
    glds16(Kg + (size_t)key * p.ldk + schunk * 8, kring + slot_ * TILE + (size_t)(wave * 64) * 16);
  };
  auto stage_v = [&](int t, int slot_) STAR_ALWAYS_INLINE {
    int key = t * KT + srow;
    if (key > p.Nk - 1) key = p.Nk - 1;
    glds16(Vg + (size_t)key * p.ldv + schunk * 8, vring + slot_ * TILE + (size_t)(wave * 64) * 16);
  };

  f32x16 oacc[NQ][2];
#pragma unroll
  for (int a = 0; a < NQ; ++a)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[a][c2][r] = 0.f;
  float m_run[NQ] = {0.f, 0.f}, l_run[NQ] = {0.f, 0.f};
  const int T_ = (p.Nk + KT - 1) / KT;
  const bool has_tail = (p.Nk & (KT - 1)) != 0;

  f32x16 s[NQ][2];
  vec<T, 8> pf[NQ][4];
  float lsum[NQ], m_tile[NQ];

  auto scores = [&](int t, const char* kbuf) STAR_ALWAYS_INLINE {
#pragma unroll
    for (int a = 0; a < NQ; ++a)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[a][kb][r] = 0.f;
        s[a][kb] = mfma32<T>(kaug, qaug[a], s[a][kb]);
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8>*>(kbuf + swz_off(kb * 32 + lq, ks * 2 + h2));
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) s[qi][kb] = mfma32<T>(kf, qf[qi][ks], s[qi][kb]);
      }
    if (has_tail && t == T_ - 1) {             // key tail of the last tile
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
  };
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
  auto rebase = [&](bool first) STAR_ALWAYS_INLINE {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float inc = first ? m_tile[qi] : fmaxf(m_tile[qi], 0.f);
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
  auto expo = [&]() STAR_ALWAYS_INLINE {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
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
      lsum[qi] = (a0 + a1) + (a2 + a3);
    }
  };
  auto pv = [&](const char* vbuf) STAR_ALWAYS_INLINE {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const vec<T, 8> vf = load_vt_frag<T>(vbuf, tt * 16, db, lane);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) oacc[qi][db] = mfma32<T>(vf, pf[qi][tt], oacc[qi][db]);
      }
  };

  // ---- prologue: K(0)
  stage_k(0, 0);
  glds_wait();
  block_sync();
  // half-steps 2t (even: the DMA of K(t+1), V(t) starts) and 2t+1 (odd: it must have landed before the next half-step);
  // group 0 runs MFMA(t) | VALU(t), group 1 runs VALU(t-1) | MFMA(t): two straight-line loops, one per group, so that the
  // scores / probabilities are never live together across the loop edge
  auto start_dma = [&](int t) STAR_ALWAYS_INLINE {
    if (t + 1 < T_) stage_k(t + 1, (t + 1) % 3);
    if (t < T_) stage_v(t, t & 1);
  };
  auto softmax_first = [&]() STAR_ALWAYS_INLINE {            // tile 0 sets the running max from exact maxima
    maxima();
    rebase(true);
    expo();
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) l_run[qi] += lsum[qi];
  };
  auto softmax_lazy = [&](int t) STAR_ALWAYS_INLINE {         // tile t >= 1: no maxima unless the row-sum probe fails
    expo();
    bool bad = false;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) bad = bad || !(lsum[qi] <= LAZY_BIG);
    if (wave_any(bad)) {                         // rare: exact maxima from recomputed scores (K(t) is still in its ring slot)
      scores(t, kring + (t % 3) * TILE);
      maxima();
      rebase(false);
      expo();
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) l_run[qi] += lsum[qi];
  };
  // first and last tiles are peeled so that the steady-state loops carry no tile-index branches
  if (grp == 0) {                                // MFMA(t) | VALU(t)
    start_dma(0);
    scores(0, kring);
    barrier_keep_dma();
    softmax_first();
    glds_wait();
    barrier_keep_dma();
    for (int t = 1; t < T_; ++t) {
      start_dma(t);
      pv(vring + ((t - 1) & 1) * TILE);
      scores(t, kring + (t % 3) * TILE);
      barrier_keep_dma();
      softmax_lazy(t);
      glds_wait();
      barrier_keep_dma();
    }
    pv(vring + ((T_ - 1) & 1) * TILE);
    barrier_keep_dma();
    barrier_keep_dma();
  } else {                                       // VALU(t-1) | MFMA(t)
    start_dma(0);
    barrier_keep_dma();
    scores(0, kring);
    glds_wait();
    barrier_keep_dma();
    if (T_ > 1) {
      start_dma(1);
      softmax_first();
      barrier_keep_dma();
      pv(vring);
      scores(1, kring + TILE);
      glds_wait();
      barrier_keep_dma();
      for (int t = 2; t < T_; ++t) {
        start_dma(t);
        softmax_lazy(t - 1);
        barrier_keep_dma();
        pv(vring + ((t - 1) & 1) * TILE);
        scores(t, kring + (t % 3) * TILE);
        glds_wait();
        barrier_keep_dma();
      }
      softmax_lazy(T_ - 1);
      barrier_keep_dma();
      pv(vring + ((T_ - 1) & 1) * TILE);
      barrier_keep_dma();
    } else {
      softmax_first();
      barrier_keep_dma();
      pv(vring);
      barrier_keep_dma();
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

}  // namespace star
