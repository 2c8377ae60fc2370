// gemm_tq.h -- the q | k | v projection of a TEMPORAL attention fused with the attention itself, for the level-0 width (C = 320,
// 5 heads): BasicTransformerBlock's temporal branch (unet_v2v.py:479-489 -> MemoryEfficientCrossAttention.forward :158-195 on
// '(b h w) f c' tensors).  As two kernels the layer writes q | k | v -- 1.6 GB at cfg2's level 0 -- and reads it straight back
// for a 32 x 32 attention per pixel and head.  Here the A-stationary K = 320 kernel of gemm_as.h runs in TEMPORAL row order:
//   * a wave's 64 rows of A are 2 pixels x 32 frames (row -> token map of its A loads: token = frame * HW + pixel), resident in
//     160 AGPRs as there; W streams through the same two-slot LDS ring, its 64-row tiles re-ordered at load time to
//     (q_h, k_h, v_h) per head; the folded-LayerNorm epilogue of tile t rides in tile t + 1's MFMA shadow as there --
//   * but writes its packed 16-bit tile into the wave's private 8 KB staging block (swz_off layout) instead of HBM; behind the
//     q and k tiles the block is read back as MFMA fragments into registers, behind the v tile the attention of that head runs on
//     them with temporal_attn_kernel's arithmetic (attn.h: S^T = K Q^T, softmax in registers, V^T by ds_read_b64_tr_b16, O^T = V^T P^T)
//     -- the same values in the same order: bit-identical to the two-kernel path -- and O leaves through buffer stores.
// HBM traffic of the pair 4.3 GB -> 1.1 GB per layer.  F <= 32 frames per chunk (longer chunks keep the two kernels).
#pragma once
#include "gemm.h"
#include "attn.h"

namespace star {

struct TqParams {
  const void* A; const void* W; void* O;     // A: token rows [F*HW][lda]; W: [15 * 64][320], tiles (q_h, k_h, v_h) per head; O: [F*HW][ldo]
  const float* bias; const float* colsum; const float* rowab;   // folded LayerNorm operands, in W's row order / per token
  int lda, ldo, HW, F;
  float scale_log2e;
};

template <class T>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 1)
gemm_tq_kernel(const TqParams p) {
  constexpr int K = 320, KS = K / 16, SLAB = 64 * 128, WTILE = (K / 64) * SLAB;   // 40 KB per 64-row W tile
  constexpr int STG = 64 * 128, N = 960, NT = N / 64, HEADS = 5;
  char* smem = dyn_smem();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = wave_uniform(tid >> 6);
  char* stg = smem + 2 * WTILE + wv * STG;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * WTILE + 4 * STG);   // bias[N] | colsum[N]
  const int h2 = lane >> 5, lq = lane & 31;
  const int pix0 = (int)blockIdx.x * 8 + wv * 2;   // this wave's two pixels: pix0, pix0 + 1
  const T* __restrict__ Ag = (const T*)p.A;

  // ---- this lane's rows of A (row block rb = pixel, lane lq = frame) as MFMA B operands, as in gemm_as.h
  vec<T, 8> af[2][KS];
  float ra[2], rbv[2];
  const int fr = lq < p.F ? lq : p.F - 1;   // frames past the chunk: a valid row (masked as keys, never stored)
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int px = pix0 + rb < p.HW ? pix0 + rb : p.HW - 1;
    const size_t m = (size_t)fr * p.HW + px;
    const T* row = Ag + m * p.lda + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[rb][ks] = *reinterpret_cast<const vec<T, 8>*>(row + ks * 16);
    const vec<float, 2> ab = *reinterpret_cast<const vec<float, 2>*>(p.rowab + 2 * m);
    ra[rb] = ab[0]; rbv[rb] = ab[1];
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) STAR_AGPR_PIN(af[rb][ks]);
  for (int n = tid; n < N; n += 256) { bias_lds[n] = p.bias[n]; bias_lds[N + n] = p.colsum[n]; }

  // ---- W staging / fragment addresses (gemm_as.h)
  const int pos = tid & 7;
  uint32_t wo[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = ps * 32 + (tid >> 3);
    wo[ps] = (uint32_t)(r * K + (pos ^ ((r >> 1) & 7)) * 8) * 2u;
  }
  const char* Wb = (const char*)p.W;
  auto stage = [&](int t, int slot) STAR_ALWAYS_INLINE {
    const char* base = Wb + (size_t)t * 64 * K * 2;
    char* dst = smem + slot * WTILE + wv * 1024;
#pragma unroll
    for (int s = 0; s < K / 64; ++s)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) glds16_su(base + s * 128, wo[ps], dst + s * SLAB + ps * 4096);
  };
  const char* wfa[2][4];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int q = 0; q < 4; ++q) wfa[sl][q] = opaque(smem + sl * WTILE + lq * 128 + ((((2 * q + h2) & 7) ^ ((lq >> 1) & 7)) << 4));

  // ---- output: rows = tokens (frame lq of pixel pix0 + rb), 128 B per head; a lane owns 16-byte pieces of its row
  const BufRsrc ors = make_rsrc(p.O, 0xFFFF0000u);
  f32x16 acc_a[2][2], acc_b[2][2];
#define STAR_TQ_ACC(P) (*((P) == 0 ? &acc_a : &acc_b))

  // ---- epilogue pieces of tile t - 1, issued inside tile t's k loop: affine, pack, 8 bytes into the staging block (swz_off layout)
  f32x4 ecs[2], ecb[2];
  auto epi_load = [&](int tp, auto utag) STAR_ALWAYS_INLINE {
    constexpr int U = decltype(utag)::value;
    constexpr int cb = (U / 4) % 2, g = U % 4;
    const float* bl = bias_lds + tp * 64;
    const int nl = cb * 32 + 8 * g + 4 * h2;
    ecs[U & 1] = *reinterpret_cast<const f32x4*>(bl + N + nl); ecb[U & 1] = *reinterpret_cast<const f32x4*>(bl + nl);
  };
  auto epi_unit = [&](auto ptag, auto utag) STAR_ALWAYS_INLINE {
    constexpr int P = decltype(ptag)::value, U = decltype(utag)::value;
    constexpr int rb = U / 8, cb = (U / 4) % 2, g = U % 4;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ra[rb] * STAR_TQ_ACC(P)[rb][cb][g * 4 + e] + (rbv[rb] * ecs[U & 1][e] + ecb[U & 1][e]);
    vec<T, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
    const int row = rb * 32 + lq;
    *reinterpret_cast<vec<T, 4>*>(stg + swz_off(row, cb * 4 + g) + h2 * 8) = o;
  };
  constexpr int NU = 16;

  auto tile = [&](int t, auto slot_tag, auto drain_tag) STAR_ALWAYS_INLINE {
    constexpr int SL = decltype(slot_tag)::value;
    constexpr bool DRAIN = decltype(drain_tag)::value;
    // W tile t has landed.  Its DMA was issued at the start of tile t - 1; behind it only the 8 output stores of a head's attention
    // can be in flight (exactly 8: dead lanes store to an out-of-range offset), and only when the attention ran behind tile t - 1,
    // i.e. t % 3 == 1 from t = 4 on (behind tile 0 nothing has run yet): they retire after the DMA, so a counted wait leaves them in flight
    if (t % 3 == 1 && t >= 4) STAR_WAIT_VMCNT_N(8); else STAR_WAIT_VMCNT(0);
    barrier_keep_dma();
    if (t + 1 < NT) stage(t + 1, SL ^ 1);
    {
      f32x16 zero;
#pragma unroll
      for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) STAR_TQ_ACC(SL)[rb][cb] = zero;
    }
    vec<T, 8> wf[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) wf[0][cb] = *reinterpret_cast<const vec<T, 8>*>(wfa[SL][0] + cb * 4096);
    if constexpr (DRAIN) {
      epi_load(t - 1, std::integral_constant<int, 0>{});
      epi_load(t - 1, std::integral_constant<int, 1>{});
    }
    static_for<KS>([&](auto kstag) STAR_ALWAYS_INLINE {
      constexpr int ks = decltype(kstag)::value;
      if constexpr (ks + 1 < KS) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          wf[(ks + 1) & 1][cb] = *reinterpret_cast<const vec<T, 8>*>(wfa[SL][(ks + 1) & 3] + ((ks + 1) >> 2) * SLAB + cb * 4096);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) STAR_TQ_ACC(SL)[rb][cb] = mfma32<T>(wf[ks & 1][cb], af[rb][ks], STAR_TQ_ACC(SL)[rb][cb]);
      if constexpr (DRAIN && ks < NU) {
        epi_unit(std::integral_constant<int, SL ^ 1>{}, std::integral_constant<int, ks>{});
        if constexpr (ks + 2 < NU) epi_load(t - 1, std::integral_constant<int, ks + 2>{});
      }
      STAR_SCHED_FENCE();
    });
  };

  // ---- behind the q / k tiles: the staged tile back as MFMA fragments (row rb * 32 + lq, 16-byte chunk 2 ks + h2)
  vec<T, 8> qf[2][4], kf[2][4];
  auto read_frags = [&](vec<T, 8> (&dst)[2][4]) STAR_ALWAYS_INLINE {
    wave_lds_order();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dst[rb][ks] = *reinterpret_cast<const vec<T, 8>*>(stg + swz_off(rb * 32 + lq, ks * 2 + h2));
    wave_lds_order();
  };
  // ---- behind the v tile: the attention of head hd for this wave's two pixels (temporal_attn_kernel's arithmetic, NB = 1)
  const float c = p.scale_log2e;
  auto attention = [&](int hd) STAR_ALWAYS_INLINE {
    wave_lds_order();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sacc = mfma32<T>(kf[rb][ks], qf[rb][ks], sacc);
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * h2;
        if (key >= p.F) sacc[r] = -1e30f;
        mx = fmaxf(mx, sacc[r]);
      }
      mx = fmaxf(mx, shfl_xor(mx, 32)) * c;
      float ls = 0.f;
      vec<T, 8> pf[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        vec<T, 8> pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = fast_exp2(sacc[8 * u + e] * c - mx);
          ls += pv;
          pk[e] = from_f32<T>(pv);
        }
        pf[u] = pk;
      }
      ls += shfl_xor(ls, 32);
      const float linv = 1.0f / ls;
      f32x16 oacc[2];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const vec<T, 8> vf = load_vt_frag<T>(stg + rb * 4096, tt * 16, db, lane);
          oacc[db] = mfma32<T>(vf, pf[tt], oacc[db]);
        }
      // O row of (frame lq, pixel pix0 + rb): 16-byte pieces through the buffer descriptor; frames / pixels past the edge get an
      // out-of-range offset (the store count per head stays exact for the counted vmcnt wait in front of the next W tile)
      const bool live = lq < p.F && pix0 + rb < p.HW;
      const uint32_t rowoff = (uint32_t)(((size_t)lq * p.HW + (pix0 + rb)) * p.ldo + hd * 64) * 2u;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          uint32_t w0[2], w1[2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            vec<T, 4> o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = from_f32<T>(oacc[db][(2 * a + gg) * 4 + e] * linv);
            u32x2 pk = __builtin_bit_cast(u32x2, o4);
            if (gg == 0) { w0[0] = pk[0]; w0[1] = pk[1]; } else { w1[0] = pk[0]; w1[1] = pk[1]; }
          }
          const u32x2 s0 = permlane32_swap(w0[0], w1[0]);
          const u32x2 s1 = permlane32_swap(w0[1], w1[1]);
          u32x4 out;
          out[0] = s0[0]; out[1] = s1[0]; out[2] = s0[1]; out[3] = s1[1];
          buf_store16(ors, live ? rowoff + (uint32_t)(32 * db + 16 * a + 8 * h2) * 2u : GLDS_BUF_OOB, out);
        }
    }
    wave_lds_order();
  };
  // what the staging block holds once tile t's k loop (which carried tile t - 1's epilogue) is done
  auto after = [&](int t) STAR_ALWAYS_INLINE {
    const int done = t - 1, role = done % 3;
    if (role == 0) read_frags(qf);
    else if (role == 1) read_frags(kf);
    else attention(done / 3);
  };

  glds_wait();                       // the A / rowab loads are in registers before the first hand-counted LDS-DMA
  block_sync();                      // bias / colsum visible
  stage(0, 0);
  tile(0, std::integral_constant<int, 0>{}, std::false_type{});
  int t = 1;
  for (; t + 1 < NT; t += 2) {
    tile(t, std::integral_constant<int, 1>{}, std::true_type{});
    after(t);
    tile(t + 1, std::integral_constant<int, 0>{}, std::true_type{});
    after(t + 1);
  }
  // NT = 15 is odd: t == 15 here; the last tile (14, slot 0, accumulator set 0) drains serially
  static_for<NU>([&](auto u) STAR_ALWAYS_INLINE {
    epi_load(NT - 1, u);
    epi_unit(std::integral_constant<int, 0>{}, u);
  });
  attention(HEADS - 1);
}
#undef STAR_TQ_ACC

}  // namespace star
