// gemm_as.h -- A-stationary GEMM for the short-K (K = 320) layers of level 0 whose output is several times wider than their input:
// the fused q | k | v projection (N = 960) and the GEGLU projection (N = 2560) of the SpatialTransformer blocks, both with the
// LayerNorm folded in (unet_v2v.py:466-477,496-529).  In the tiled kernel of gemm.h such a layer is five K tiles between a cold
// prologue and a store-bound epilogue (profiles/r02_gemm_shortk_ablation.txt: the MFMAs are 7 % of the critical path).  Here
//   * a wave keeps its 64 rows x 320 k of A as MFMA operands IN REGISTERS for the whole row block (160 AGPRs; one wave per SIMD,
//     built with -mllvm -amdgpu-mfma-vgpr-form so that the accumulators stay in architectural VGPRs): A is read from HBM once, with
//     no LDS traffic and no K loop over it;
//   * W (L2-resident: 320 x N) streams through a two-slot LDS ring in 64-row tiles (40 KB: five 64-k slabs in the swizzled
//     128-byte-row layout of gemm.h), 8 B/clk/CU; one raw barrier per W tile = per 80 MFMAs of a wave;
//   * the epilogue (folded-LayerNorm affine, GEGLU with the erfc-form GELU, pack) works on the lane's own row (operand-swapped MFMA:
//     a lane owns one row of a 32-row block); the packed outputs cross a wave-private, XOR-swizzled LDS block (64 rows x 128 B) so
//     that they leave as whole 128-byte lines, 16 B per lane along a row (row-per-lane stores at a row stride are store-ISSUE
//     bound: the first build of this kernel spent 4.6k of its 9.2k cycles per tile in them), through buffer stores (out-of-range
//     rows dropped by the descriptor: the store count per tile is exact, so the LDS-DMA of the next W tile is awaited with a
//     COUNTED vmcnt instead of draining the stores).
// Same arithmetic as gemm_kernel's EPI_ROWAFF (+ EPI_GEGLU) flavours: k-steps accumulate in the same order, the epilogue
// expressions are the same.
#pragma once
#include "gemm.h"

namespace star {

// ABL 6 (round 4, for round 5's first call): everything but the GLOBAL STORES of the flush (the epilogue's VALU, the staging writes and
// the flush reads stay) -- vmcnt retires in order, so the per-tile wait for the next W tile's LDS-DMA also waits for the OLDER stores of
// the flush before it: if the store latency under load exceeds a tile time, that wait is the epilogue's unexplained 0.22 ms.
// ABL (bench builds, timing only, wrong results): 1 no epilogue, 2 no W staging / barrier after tile 0, 3 W fragments not re-read,
// 4 = 1 + 2 (the k loop alone: MFMAs + W fragment reads), 5 = 4 + 3 (MFMAs alone)
// ILV: the epilogue pieces are INTERLEAVED with the MFMAs of a k-step by sched_group_barrier patterns (1 MFMA, then a share of the
// step's VALU) instead of being left to the scheduler, which clusters them (ISA of the ILV = 0 GEGLU loop: 8 MFMAs back to back, then
// ~75 VALU with the matrix pipe idle -- 560 cycles per two k-steps where max(MFMA, VALU) is 380); GEGLU units are split in two halves
// (one per k-step) so that every step carries VALU work.  Same instructions, same arithmetic: bit-identical.
template <class T, int GEGLU, int ABL_ = 0, int ILV = 0>
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(256, 1)
gemm_astat_kernel(const GemmParams p) {
  constexpr int ABL = ABL_ >= 6 ? 0 : ABL_;     // 6: the product paths everywhere except the global stores
  constexpr bool NOSTORE = ABL_ == 6;
  constexpr int K = 320, KS = K / 16, SLAB = 64 * 128, WTILE = (K / 64) * SLAB;   // 40 KB per 64-row W tile
  constexpr int STG = 64 * 128;                 // per-wave staging block: 64 rows x 64 outputs (GEGLU: two W tiles fill it)
  // Round 5 (profiles/r05_cbench_astat_*.txt, r05_pmc_astat.txt): half of a wave's life in this kernel is an s_waitcnt, and it is vector
  // memory, not the LDS (SQ_WAIT_INST_LDS 4 %); WITHOUT its global stores the kernel is 23 % faster at N = 960 (tile 41).  Two cheap,
  // bit-identical measures ship: the flush's stores are NON-TEMPORAL (aux nt: the 1.6 GB output streams past the L2 instead of evicting
  // the W panel and the next rows of A; +4.5 % at N = 960, +1.7 % at N = 2560; write-through sc1 loses 2.6 %) and the FIRST round of
  // workgroups starts staggered (every workgroup does the same work, so the 256 CUs ran their A-panel prologues -- 42 MB at once, nobody
  // computing -- in lockstep: phase (blockIdx / 8) % 8 sleeps phase x 8128 cycles; +2.9 % / +0.3 %, longer delays lose; only launches of >= 1024 workgroups, where the last phase's 27 us are small).  The counted
  // per-tile wait is worth 2.3 % against vmcnt(0) (tile 46): store LATENCY is not what is left.
  // ABL_ 7 / 8 / 9: stores plain (the round-4 kernel's) / write-through / write-through + nt;  10: every per-tile wait is vmcnt(0);
  // 11: stagger unit from p.group_m (1, 2, 4, 8);  12: the round-4 kernel (plain stores, no stagger)
  constexpr int STPOL = (ABL_ == 7 || ABL_ == 12) ? 0 : ABL_ == 8 ? 16 : ABL_ == 9 ? 19 : 2;
  constexpr bool DRAIN_ALL = ABL_ == 10;
  constexpr bool STAGGER = ABL_ != 12;
  if constexpr (STAGGER) {
    // p.tiles_m (unused by this kernel otherwise) = the stagger window, set by the launcher from the device's CU count: the workgroups of the
    // FIRST round (one per CU), and only on launches of >= 4 rounds; 0 = no stagger (the window was a literal 256)
    if ((int)blockIdx.x < p.tiles_m) wave_sleep(((int)(blockIdx.x >> 3) & 7) * (ABL_ == 11 ? p.group_m : 1) * 127);
  }
  char* smem = dyn_smem();
  char* stg = smem + 2 * WTILE + wave_uniform((int)threadIdx.x >> 6) * STG;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * WTILE + 4 * STG);   // bias[N] | colsum[N]
  // ILV >= 3: the GELU polynomial as scalar v_fma_f32 instead of v_pk_fma_f32.  Measured (tools/probe/mfma_valu_overlap.hip,
  // profiles/r04_probe_mfma_valu_overlap.txt): a wave's v_fma_f32 ride in the shadow of its own MFMAs (4 per MFMA for free, 4.3
  // cycles each beyond), its v_pk_fma_f32 do NOT (MFMA + 16 + 4.4 cycles each, no overlap at all) -- packed fp32 only pays where no
  // MFMA is in flight.  ILV 3 = 1 + scalar, 4 = 0 + scalar, 5 = 2 + scalar.
  // ILV 6: the product placement with the units re-ordered so that the two row blocks of a column quad are CONSECUTIVE and share one
  // fetch of its column sums / biases: 16 instead of 32 ds_read_b128 per tile and wave.  (Per W tile a CU's LDS moves 160 KB of W
  // fragments + 128 KB of these broadcast reads + 40 KB of staging and flush + the 40 KB DMA fill: ~2100 LDS-array cycles against 2560
  // cycles of MFMA -- the W fragment reads the MFMAs wait for queue behind the epilogue's.)
  constexpr bool SCALAR = ILV >= 3 && ILV <= 5;
  constexpr bool SHARE = ILV == 6;
  constexpr int ILVP = ILV == 6 ? 0 : ILV >= 3 ? (ILV == 3 ? 1 : ILV == 4 ? 0 : 2) : ILV;   // the placement pattern
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = wave_uniform(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int mwg = blockIdx.x * 256;
  const int m0 = mwg + wv * 64;
  const int N = p.N;
  const T* __restrict__ Ag = (const T*)p.A;

  // ---- this lane's rows of A as MFMA B operands (a lane owns row rb*32 + lq of its wave's 64): af[rb][ks] = A[m][16 ks + 8 h2 ..]
  vec<T, 8> af[2][KS];
  float ra[2], rbv[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    int m = m0 + rb * 32 + lq;
    if (m > p.M - 1) m = p.M - 1;
    const T* row = Ag + (size_t)m * p.lda + h2 * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[rb][ks] = *reinterpret_cast<const vec<T, 8>*>(row + ks * 16);
    const vec<float, 2> ab = *reinterpret_cast<const vec<float, 2>*>(p.rowab + 2 * (size_t)m);
    ra[rb] = ab[0]; rbv[rb] = ab[1];
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) STAR_AGPR_PIN(af[rb][ks]);
  for (int n = tid; n < N; n += 256) { bias_lds[n] = p.bias[n]; bias_lds[N + n] = p.colsum[n]; }

  // ---- W staging: thread (ps, tid) copies chunk (tid & 7) ^ swizzle of tile row ps*32 + (tid >> 3), one 64-k slab per instruction
  const int pos = tid & 7;
  uint32_t wo[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = ps * 32 + (tid >> 3);
    wo[ps] = (uint32_t)(r * K + (pos ^ ((r >> 1) & 7)) * 8) * 2u;
  }
  const char* Wb = (const char*)p.W;
  auto stage = [&](int t, int slot) STAR_ALWAYS_INLINE {
    const char* base = Wb + (size_t)t * 64 * K * 2;   // wave-uniform
    char* dst = smem + slot * WTILE + wv * 1024;
#pragma unroll
    for (int s = 0; s < K / 64; ++s)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) glds16_su(base + s * 128, wo[ps], dst + s * SLAB + ps * 4096);
  };
  // fragment addresses (A operand of the swapped MFMA: row n = cb*32 + lq, 16-B chunk (2 ks + h2) of the 640-B row = chunk
  // (2 ks + h2) & 7 of slab ks >> 2); slab and column block go into the ds_read's immediate offset
  const char* wfa[2][4];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int q = 0; q < 4; ++q) wfa[sl][q] = opaque(smem + sl * WTILE + lq * 128 + ((((2 * q + h2) & 7) ^ ((lq >> 1) & 7)) << 4));

  // ---- output: per-workgroup buffer descriptor, rows at or past M are out of range and dropped
  const int n_out = GEGLU ? N / 2 : N;
  const int rows_here = p.M - mwg < 256 ? p.M - mwg : 256;
  const BufRsrc crs = make_rsrc((const char*)p.C + (size_t)mwg * p.ldc * 2, (uint32_t)((size_t)rows_here * p.ldc * 2));
  // LDS -> global: lane l stores 16-B chunk (l & 7) of row 8 i + (l >> 3), i = 0..7: one instruction = 8 whole 128-byte lines
  uint32_t srow_g = (uint32_t)(wv * 64 + (lane >> 3)) * (uint32_t)p.ldc * 2u + (uint32_t)(lane & 7) * 16u;
  const uint32_t srow_step = 8u * (uint32_t)p.ldc * 2u;
  const int srd = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // + i * 1024 (rows 8 i + .. keep row & 7)
  const int nt = N / 64;
  f32x16 acc_a[2][2], acc_b[2][2];   // [row block][column block] x two sets by tile parity: the MFMAs fill one set while the epilogue
                                     // drains the other (two separate arrays: one [2][2][2] array ended up in scratch memory)
#define STAR_ACC(P) (*((P) == 0 ? &acc_a : &acc_b))

  // ---- epilogue pieces, issued inside the NEXT tile's k loop (or after the last tile).  A unit = one quad of 4 consecutive output
  // columns of one row block: affine (+ GEGLU), pack, 8 bytes into the wave's staging block.
  f32x4 ecs[2], ecb[2], egs[2], egb[2];   // column sums / biases of the next two units (read from LDS TWO k-steps ahead of their use: one
                                          // step ahead the compiler's wait for them also drained the W fragment reads, 20 times per tile)
  auto epi_load = [&](int tp, auto utag) STAR_ALWAYS_INLINE {
    constexpr int U = decltype(utag)::value;
    constexpr int Q = SHARE ? U / 2 : U;                      // SHARE: unit U = (column quad U / 2, row block U % 2)
    constexpr int cb = GEGLU ? 0 : (Q / 4) % 2, g = Q % 4, S = Q & 1;
    const float* bl = bias_lds + tp * 64;
    const int nl = cb * 32 + 8 * g + 4 * h2;
    ecs[S] = *reinterpret_cast<const f32x4*>(bl + N + nl); ecb[S] = *reinterpret_cast<const f32x4*>(bl + nl);
    if constexpr (GEGLU != 0) { egs[S] = *reinterpret_cast<const f32x4*>(bl + N + nl + 32); egb[S] = *reinterpret_cast<const f32x4*>(bl + nl + 32); }
  };
  auto epi_unit = [&](auto ptag, auto utag) STAR_ALWAYS_INLINE {   // drains accumulator set P with the operands epi_load fetched
    constexpr int P = decltype(ptag)::value, U = decltype(utag)::value;
    constexpr int Q = SHARE ? U / 2 : U, S = Q & 1;
    constexpr int rb = SHARE ? U % 2 : U / (GEGLU ? 4 : 8), cb = GEGLU ? 0 : (Q / 4) % 2, g = Q % 4;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ra[rb] * STAR_ACC(P)[rb][cb][g * 4 + e] + (rbv[rb] * ecs[S][e] + ecb[S][e]);
    if constexpr (GEGLU != 0) {
      f32x4 gt;
#pragma unroll
      for (int e = 0; e < 4; ++e) gt[e] = ra[rb] * STAR_ACC(P)[rb][1][g * 4 + e] + (rbv[rb] * egs[S][e] + egb[S][e]);
      if constexpr (SCALAR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * gelu_erf(gt[e]);
      } else
      v = v * gelu_erf4(gt);
    }
    vec<T, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
    // staging block: row rb*32 + lq, 16-B chunk c = (output column) / 8, 8-byte half h2; chunk c lives at c ^ (row & 7)
    const int c = (GEGLU ? P * 4 : cb * 4) + g;
    const int row = rb * 32 + lq;
    *reinterpret_cast<vec<T, 4>*>(stg + row * 128 + ((c ^ (row & 7)) << 4) + h2 * 8) = o;
  };
  // GEGLU, ILV: half a unit (2 of the quad's 4 columns) per k-step; the packed pair of the first half waits in a register for the second
  vec<T, 2> o_lo;
  auto epi_half = [&](auto ptag, auto utag, auto htag) STAR_ALWAYS_INLINE {
    constexpr int P = decltype(ptag)::value, U = decltype(utag)::value, H = decltype(htag)::value;
    constexpr int rb = U / 4, g = U % 4;
    f32x2 v, gt;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      v[e] = ra[rb] * STAR_ACC(P)[rb][0][g * 4 + 2 * H + e] + (rbv[rb] * ecs[U & 1][2 * H + e] + ecb[U & 1][2 * H + e]);
      gt[e] = ra[rb] * STAR_ACC(P)[rb][1][g * 4 + 2 * H + e] + (rbv[rb] * egs[U & 1][2 * H + e] + egb[U & 1][2 * H + e]);
    }
    if constexpr (SCALAR) {
#pragma unroll
      for (int e = 0; e < 2; ++e) v[e] = v[e] * gelu_erf(gt[e]);
    } else
    v = v * gelu_erf2(gt);
    if constexpr (H == 0) {
      o_lo[0] = from_f32<T>(v[0]); o_lo[1] = from_f32<T>(v[1]);
    } else {
      vec<T, 4> o;
      o[0] = o_lo[0]; o[1] = o_lo[1]; o[2] = from_f32<T>(v[0]); o[3] = from_f32<T>(v[1]);
      const int c = P * 4 + g;
      const int row = rb * 32 + lq;
      *reinterpret_cast<vec<T, 4>*>(stg + row * 128 + ((c ^ (row & 7)) << 4) + h2 * 8) = o;
    }
  };
  constexpr int NU = GEGLU ? 8 : 16;   // units per tile
  // flush piece i of 8: 8 rows x 128 B leave the staging block as whole lines (DS operations of a wave execute in order: the
  // reads see the units' writes; only the compiler has to keep the order)
  u32x4 fl[2];                // two flush pieces in flight between their LDS read and their store
  auto flush_read = [&](int i, int slot) STAR_ALWAYS_INLINE { fl[slot] = *reinterpret_cast<const u32x4*>(stg + srd + i * 1024); };
  auto flush_store = [&](int tp, int i, int slot, bool live_all) STAR_ALWAYS_INLINE {
    const uint32_t col0 = (uint32_t)(GEGLU ? (tp >> 1) * 64 : tp * 64) * 2u;
    const bool live = live_all || (lane & 7) < 4;   // an odd GEGLU tile count leaves the upper half of the block stale
    if constexpr (NOSTORE) { asm volatile("" :: "v"(fl[slot])); (void)live; (void)col0; }
    else if constexpr (STPOL != 0) buf_store16_pol<STPOL>(crs, live ? srow_g + i * srow_step + col0 : GLDS_BUF_OOB, fl[slot]);
    else buf_store16(crs, live ? srow_g + i * srow_step + col0 : GLDS_BUF_OOB, fl[slot]);
  };

  auto tile = [&](int t, auto slot_tag, auto drain_tag) STAR_ALWAYS_INLINE {
    constexpr int SL = decltype(slot_tag)::value;
    constexpr bool DRAIN = decltype(drain_tag)::value && ABL != 1 && ABL < 4;   // compile time: the epilogue pieces must share basic blocks with the MFMAs
    // does THIS tile's k loop carry the stores of a flush?  (tile t drains tile t - 1; GEGLU flushes once two tiles are staged)
    const bool prev_flushed = GEGLU ? (SL == 1 && t >= 3) : (t >= 2);   // ... and did the PREVIOUS tile's?
    // W tile t has landed (the only vector-memory operations issued after its DMA are the 8 stores of the previous tile's
    // flush, if it had one; vmcnt retires in order), and every wave is done reading the other slot
    if ((ABL != 2 && ABL < 4) || t == 0) {
      if (prev_flushed && ABL != 1 && ABL < 4 && !NOSTORE && !DRAIN_ALL) STAR_WAIT_VMCNT_N(8); else STAR_WAIT_VMCNT(0);
      barrier_keep_dma();
      if (t + 1 < nt && ABL != 2 && ABL < 4) stage(t + 1, SL ^ 1);
    }
    {
      f32x16 zero;   // assigned as a WHOLE vector: element-wise zeroing gets SLP-packed into <2 x float> stores, which keep the
#pragma unroll       // accumulators from being promoted out of scratch memory
      for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) STAR_ACC(SL)[rb][cb] = zero;
    }
    constexpr bool FLUSH = DRAIN && (GEGLU == 0 || SL == 0);   // GEGLU: tile t - 1 odd completes a 64-column block
    // k-steps: the W fragments of step ks + 1 are read (2 ds_read_b128) ahead of the 4 MFMAs of step ks; each step also carries
    // one piece of the previous tile's epilogue in the shadow of its MFMAs
    vec<T, 8> wf[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) wf[0][cb] = *reinterpret_cast<const vec<T, 8>*>(wfa[SL][0] + cb * 4096);
    if constexpr (DRAIN) {
      epi_load(t - 1, std::integral_constant<int, 0>{});
      if constexpr (GEGLU == 0 && !SHARE) epi_load(t - 1, std::integral_constant<int, 1>{});
    }
    static_for<KS>([&](auto kstag) STAR_ALWAYS_INLINE {
      constexpr int ks = decltype(kstag)::value;
      if (ks + 1 < KS && ((ABL != 3 && ABL != 5) || ks == 0)) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          wf[(ks + 1) & 1][cb] = *reinterpret_cast<const vec<T, 8>*>(wfa[SL][(ks + 1) & 3] + ((ks + 1) >> 2) * SLAB + cb * 4096);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) STAR_ACC(SL)[rb][cb] = mfma32<T>(wf[ks & 1][cb], af[rb][ks], STAR_ACC(SL)[rb][cb]);
      if constexpr (DRAIN && ILVP == 1 && GEGLU != 0) {   // half a unit per step; the next unit's LDS operands are fetched behind the first half
        if constexpr (ks / 2 < NU) {
          epi_half(std::integral_constant<int, SL ^ 1>{}, std::integral_constant<int, ks / 2>{}, std::integral_constant<int, ks % 2>{});
          if constexpr (ks % 2 == 0 && ks / 2 + 1 < NU) epi_load(t - 1, std::integral_constant<int, ks / 2 + 1>{});
        }
      } else
      if constexpr (DRAIN) {   // one epilogue unit per step (GEGLU: per two steps); its LDS operands were fetched two steps earlier
        constexpr int STRIDE = GEGLU ? 2 : 1, LEAD = 2 / STRIDE;
        if constexpr (ks % STRIDE == 0 && ks / STRIDE < NU) {
          epi_unit(std::integral_constant<int, SL ^ 1>{}, std::integral_constant<int, ks / STRIDE>{});
          if constexpr (SHARE) {   // behind the first unit of a pair: the next pair's operands (the other slot)
            if constexpr ((ks / STRIDE) % 2 == 0 && ks / STRIDE + 2 < NU) epi_load(t - 1, std::integral_constant<int, ks / STRIDE + 2>{});
          } else
          if constexpr (ks / STRIDE + LEAD < NU) epi_load(t - 1, std::integral_constant<int, ks / STRIDE + LEAD>{});
        }
      }
      if constexpr (FLUSH) {   // steps 16..19: piece pair k is read while pair k - 1 is stored
        if constexpr (ks > 16) { flush_store(t - 1, 2 * (ks - 17), 0, true); flush_store(t - 1, 2 * (ks - 17) + 1, 1, true); }
        if constexpr (ks == 16) wave_lds_order();   // all sixteen units have written the block
        if constexpr (ks >= 16) { flush_read(2 * (ks - 16), 0); flush_read(2 * (ks - 16) + 1, 1); }
        if constexpr (ks == KS - 1) wave_lds_order();   // ... and it is read out before the next tile's units overwrite it
      }
      // ILV 1: this step's 4 MFMAs, each followed by a quarter of its VALU.  ILV 2 (GEGLU): whole units as in ILV 0, but the fence
      // between the unit's step and the next one is dropped and the pair's 8 MFMAs are each followed by an eighth of the unit
      // (two independent polynomial chains side by side: no s_nop between dependent v_pk_fma_f32)
      constexpr bool PAIR = ILVP == 2 && GEGLU != 0;
      if constexpr (DRAIN && ILVP != 0 && !PAIR && ks < (GEGLU ? 2 * NU : NU)) {
        constexpr int VPER = GEGLU ? (SCALAR ? 12 : 10) : (ILVP == 2 ? 4 : 6);
#pragma unroll
        for (int i = 0; i < 4; ++i) { STAR_SCHED_GROUP(0x008, 1, 0); STAR_SCHED_GROUP(0x002, VPER, 0); }
      }
      if constexpr (DRAIN && PAIR && ks % 2 == 1 && ks / 2 < NU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { STAR_SCHED_GROUP(0x008, 1, 0); STAR_SCHED_GROUP(0x002, SCALAR ? 12 : 10, 0); }
      }
      if constexpr (!(DRAIN && PAIR && ks % 2 == 0 && ks / 2 < NU && ks + 1 < KS)) STAR_SCHED_FENCE();
    });
    if constexpr (FLUSH) { flush_store(t - 1, 6, 0, true); flush_store(t - 1, 7, 1, true); }
  };
  glds_wait();                       // the A / rowab loads are in registers before the first hand-counted LDS-DMA
  block_sync();                      // bias / colsum slice visible
  stage(0, 0);
  tile(0, std::integral_constant<int, 0>{}, std::false_type{});     // nothing to drain yet
  int t = 1;
  for (; t + 1 < nt; t += 2) {
    tile(t, std::integral_constant<int, 1>{}, std::true_type{});
    tile(t + 1, std::integral_constant<int, 0>{}, std::true_type{});
  }
  if (t < nt) { tile(t, std::integral_constant<int, 1>{}, std::true_type{}); ++t; }
  // ---- drain the last tile (its accumulators sit in set (nt - 1) & 1)
  if constexpr (ABL != 1 && ABL < 4) {
    static_for<NU>([&](auto u) STAR_ALWAYS_INLINE {
      if constexpr (!SHARE || decltype(u)::value % 2 == 0) epi_load(nt - 1, u);
      if ((nt - 1) & 1) epi_unit(std::integral_constant<int, 1>{}, u); else epi_unit(std::integral_constant<int, 0>{}, u);
    });
    wave_lds_order();
#pragma unroll
    for (int i = 0; i < 8; ++i) { flush_read(i, 0); flush_store(nt - 1, i, 0, GEGLU == 0 || ((nt - 1) & 1) != 0); }
  } else {
    float z = 0.f;
    z = acc_a[0][0][0] + acc_a[0][1][0] + acc_a[1][0][0] + acc_a[1][1][0] + acc_b[0][0][0] + acc_b[0][1][0] + acc_b[1][0][0] + acc_b[1][1][0];
    if (z == 1234.5f) bias_lds[0] = z;
  }
  (void)n_out;
}
#undef STAR_ACC

}  // namespace star
