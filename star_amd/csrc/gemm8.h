// gemm8.h -- persistent, phase-interleaved MFMA GEMM / implicit-conv kernel (round 2).
//
// Same contract as gemm.h (C[M,N] = epilogue(A'[M,K] * W[N,K]^T), A' gathered by LDS-DMA, weights [N][K]), different
// schedule.  The 2-stage loop of gemm.h issues all LDS-DMA of a K tile in one burst at the top of the tile, by all eight
// waves at once, and both waves of a SIMD then wait for LDS and fight for the matrix pipe in lockstep: the pipe idles
// while the burst is issued (profiles/r01_gemm_ablation.txt: 1.7 us per 64-deep step where the MFMAs need 0.85 us).  Here
//
//  * 256 x 256 x 64 tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 2 x 2 quadrants of (64 rows x 32 cols);
//  * a K tile is FOUR phases, one quadrant x K = 64 each (8 MFMA 32x32x16 = 256 matrix-pipe cycles per wave);
//  * the K tile is staged as four 16 KB PARTS, each the rows ONE phase reads: A0 / A1 = the first / second 64 rows of
//    both wave rows, B0 / B1 = the first / second 32 columns of all four wave columns.  One part (2 LDS-DMA per
//    thread) is issued per phase, 3-6 phases ahead of its first read, into a 2 x 64 KB ring; never a burst;
//  * the two waves of a SIMD (w and w + 4: wave rows 0 and 1) run half a phase apart: while one issues its 8 MFMAs
//    (s_setprio 1) the other reads its fragments from LDS and issues its DMA; two raw s_barrier per phase keep the
//    antiphase, and neither drains vmcnt (LDS-DMA stays in flight across them);
//  * persistent workgroups (one per CU) walk their output tiles with the DMA stream running across tile boundaries:
//    the first K tiles of the next output tile are already landing while the epilogue of this one runs, which is what
//    the short-K layers (K = 320 / 640: 5 / 10 K tiles per output tile) need.  The epilogue is wave-private (LDS
//    staging block per wave, no workgroup barrier), so the two wave groups run theirs half a phase apart as well.
//
// LDS hazards, by construction (MI355X_MICROARCH.md "Two waves per SIMD" item 7; guide section 5 "Read a staged buffer one
// phase AFTER the wait").  In K tile t (ring slot t & 1) the load sections read A0 + B0 (phase 0), B1 (phase 1), A1
// (phase 2), B0 again (phase 3).  The load section of phase p of K tile t also issues
//      p0: A1(t+1)   p1: B0(t+1)   p2: A0(t+2)   p3: B1(t+2), then waits vmcnt(4)
// RAW: after p3's wait only A0/B1(t+2) (4 DMA per wave) are in flight, so every part of K tile t+1 has landed in this
//      wave's view; the other wave group passes the same wait one barrier later, and the first read of t+1 is two
//      barriers later.
// WAR: a part is re-staged at least two phases (four barriers) after the load section that last read it, for either
//      wave group: A0(t+2) over A0(t) (read p0) in p2; B1(t+2) over B1(t) (read p1) in p3; A1(t+1) over A1(t-1) (read p2
//      of t-1) in p0; B0(t+1) over B0(t-1) (read p3 of t-1) in p1.
// vmcnt counts stores too and retires in order: the epilogue's stores sit between the DMA of K tiles t+1 / t+2 and the
// next wait, which therefore also waits for them.  The bias slice of a tile travels by 4-byte LDS-DMA as well (no VGPR
// destination, so hipcc never drains the queue for it).
#pragma once
#include "gemm.h"

namespace star {

struct G8 {
  static constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
  static constexpr int PART = 16384, BUF = 4 * PART;
  static constexpr int OFF_A0 = 0, OFF_B0 = PART, OFF_B1 = 2 * PART, OFF_A1 = 3 * PART;
  static constexpr int EPI_PITCH = 136, EPI_WAVE = 16 * EPI_PITCH;     // 16 rows x (64 cols x 2 B + 8) per wave
  static constexpr int SMEM_EPI = 2 * BUF, SMEM_BIAS = SMEM_EPI + 8 * EPI_WAVE;
  static constexpr int SMEM_TOTAL = SMEM_BIAS + 8 * 256;              // + 64 fp32 bias values per wave: 150528 B
};

// 4-byte-per-lane LDS-DMA (bias slices): destination lds_wave_base + lane*4
STAR_DEV void glds4(const void* gsrc, void* lds_wave_base) {
#ifdef STAR_HOSTEMU
  memcpy((char*)lds_wave_base + lane_id() * 4, gsrc, 4);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
#endif
}

template <class T, int AMODE, bool RES, int ABL = 0>   // ABL: timing ablations (bench build only; results are garbage): 1 no DMA in the loop, 2 no fragment reads, 3 neither, 4 neither + no barriers, 5 neither + one barrier per phase, 6 DMA from a hot 64 KB region (always cache hits), 7 no A DMA, 8 no B DMA; 9 = a REAL variant (correct results): no stagger between the two wave groups (all eight waves in phase)
// RES: residual add in the epilogue (compile time: a run-time branch around the loads would make hipcc's vmcnt waits inexact)
STAR_GLOBAL void STAR_LAUNCH_BOUNDS(512, 2)
gemm8_kernel(const GemmParams p) {
  constexpr int BM = G8::BM, BN = G8::BN, BK = G8::BK;
  char* smem = dyn_smem();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_uniform(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int nk = p.K / BK;
  const int nblk = p.tiles_m * p.tiles_n;
  const int G = (int)gridDim.x;
  const int my_tiles = ((int)blockIdx.x < nblk) ? (nblk - 1 - (int)blockIdx.x) / G + 1 : 0;
  if (my_tiles == 0) return;

  // ---- position of one K tile in this workgroup's stream (all wave-uniform)
  struct Pos { int it, kt, tap, c0, m0, n0; bool valid; };
  // it-th output tile of this workgroup.  Virtual block id v = blockIdx + it * gridDim keeps v % 8 = blockIdx % 8 (the XCD;
  // gridDim is a multiple of 8 whenever a workgroup gets more than one tile); the remap gives the workgroups of one XCD
  // consecutive logical tiles, which share the A row panel (same bijection as gemm.h).
  auto enter_tile = [&](Pos& q) STAR_ALWAYS_INLINE {
    q.kt = 0; q.tap = 0; q.c0 = 0;
    q.valid = q.it < my_tiles;
    if (!q.valid) return;
    const int v = (int)blockIdx.x + q.it * G;
    const int qq = nblk >> 3, r = nblk & 7;
    const int xcd = v & 7, slot = v >> 3;
    const int id = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    const int tm = id / p.tiles_n;
    q.m0 = tm * BM;
    q.n0 = (id - tm * p.tiles_n) * BN;
  };
  auto advance = [&](Pos& q) STAR_ALWAYS_INLINE {
    if (!q.valid) return;
    ++q.kt;
    if constexpr (AMODE != A_PLAIN) { q.c0 += BK; if (q.c0 >= p.Cin) { q.c0 = 0; ++q.tap; } }
    if (q.kt == nk) { ++q.it; enter_tile(q); }
  };

  const T* __restrict__ Ag = (const T*)p.A;
  const T* __restrict__ Wg = (const T*)p.W;

  // ---- loader identity of this thread: DMA j (0/1) of a part covers part rows j*64 + (tid >> 3), 16-B chunk tid & 7
  const int lrow = tid >> 3;
  const int cc8 = ((tid & 7) ^ ((lrow >> 1) & 7)) * 8;   // source chunk (elements): the LDS image is swizzled through the source address
  const int b_row = (tid >> 8) * 64 + (lrow & 31);       // B parts: part row j*64 + lrow = wave column 2j + (tid >> 8), column lrow & 31

  // conv / temporal modes: coordinates of this thread's two rows of each A part, refreshed when the part's stream enters
  // an output tile (y | x << 16 packed; image index / frame)
  int st_yx[2][2], st_nb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) { st_yx[s][j] = 0; st_nb[s][j] = 0; }

  auto a_issue = [&](const Pos& q, const int seq, const int s, const int off) STAR_ALWAYS_INLINE {
    if (!q.valid) return;
    if (ABL == 1 || (ABL >= 3 && ABL <= 5) || ABL == 7) { if (seq >= 2) return; }
    char* dst = smem + (seq & 1) * G8::BUF + off;
    if constexpr (ABL == 6) {
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(Ag + (size_t)(j * 128 + s * 64 + lrow) * p.lda + cc8, dst + (size_t)(j * 512 + wave * 64) * 16);
      return;
    }
    if constexpr (AMODE != A_PLAIN) {
      if (q.kt == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int m = q.m0 + j * 128 + s * 64 + lrow;
          if (m > p.M - 1) m = p.M - 1;
          if constexpr (AMODE == A_TCONV3) {
            st_nb[s][j] = m / p.HW;
          } else {
            const int hw = p.Ho * p.Wo;
            const int nb = m / hw, rem = m - nb * hw;
            const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
            st_nb[s][j] = nb;
            st_yx[s][j] = ((yo * p.stride - p.pad_t) & 0xffff) | ((xo * p.stride - p.pad_l) << 16);
          }
        }
      }
    }
    int ky = 0, kx = 0;
    if constexpr (AMODE == A_CONV3X3 || AMODE == A_CONV3X3_UP) { ky = q.tap / 3; kx = q.tap - ky * 3; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const void* src;
      if constexpr (AMODE == A_PLAIN || AMODE == A_TCONV3) {
        int m = q.m0 + j * 128 + s * 64 + lrow;
        if (m > p.M - 1) m = p.M - 1;
        if constexpr (AMODE == A_PLAIN) {
          src = Ag + (size_t)m * p.lda + (cc8 + q.kt * BK);
        } else {
          const int f = st_nb[s][j] + q.tap - 1;
          src = (f >= 0 && f < p.F) ? (const void*)(Ag + ((ptrdiff_t)m + (ptrdiff_t)(q.tap - 1) * p.HW) * p.lda + (cc8 + q.c0)) : p.zero_page;
        }
      } else {
        const int y = (int)(short)(st_yx[s][j] & 0xffff) + ky, x = (st_yx[s][j] >> 16) + kx;
        if constexpr (AMODE == A_CONV3X3) {
          src = (y >= 0 && y < p.H && x >= 0 && x < p.Wd)
                    ? (const void*)(Ag + (((size_t)st_nb[s][j] * p.H + y) * p.Wd + x) * p.lda + (cc8 + q.c0)) : p.zero_page;
        } else {   // A_CONV3X3_UP: conv input U[y][x] = X[(y+crop)>>1][x>>1], U is (2H - 2 crop) x (2 Wd)
          src = (y >= 0 && y < 2 * p.H - 2 * p.up_crop && x >= 0 && x < 2 * p.Wd)
                    ? (const void*)(Ag + (((size_t)st_nb[s][j] * p.H + ((y + p.up_crop) >> 1)) * p.Wd + (x >> 1)) * p.lda + (cc8 + q.c0))
                    : p.zero_page;
        }
      }
      glds16(src, dst + (size_t)(j * 512 + wave * 64) * 16);
    }
  };
  auto b_issue = [&](const Pos& q, const int seq, const int s, const int off) STAR_ALWAYS_INLINE {
    if (!q.valid) return;
    if (ABL == 1 || (ABL >= 3 && ABL <= 5) || ABL == 8) { if (seq >= 2) return; }
    char* dst = smem + (seq & 1) * G8::BUF + off;
    if constexpr (ABL == 6) {
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16(Wg + (size_t)(j * 128 + s * 32 + b_row) * p.K + cc8, dst + (size_t)(j * 512 + wave * 64) * 16);
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = q.n0 + j * 128 + s * 32 + b_row;
      if (n > p.N - 1) n = p.N - 1;
      glds16(Wg + (size_t)n * p.K + (cc8 + q.kt * BK), dst + (size_t)(j * 512 + wave * 64) * 16);
    }
  };

  // ---- fragment read offsets: part row R, k-step ks -> R*128 + (((2*ks + fhalf) ^ ((R>>1)&7)) << 4); (R>>1)&7 depends on frow only
  const int fsw = (frow >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fo[ks] = (((2 * ks + fhalf) ^ fsw) << 4);
  const int a_row0 = (wr * 64 + frow) * 128;     // + i * 32 * 128
  const int b_row0 = (wc * 32 + frow) * 128;

  vec<T, 8> a[2][4], b[4];
  f32x16 acc[2][2][2];   // [A sub][B sub][row block i]
  auto zero_acc = [&]() STAR_ALWAYS_INLINE {
#pragma unroll
    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[sa][sb][i][r] = 0.f;
  };
  bool abl_first = true;   // ABL >= 2: fragments are read once and kept (same operand statistics, no LDS traffic)
  auto read_A = [&](const char* part) STAR_ALWAYS_INLINE {
    if (ABL >= 2 && ABL <= 5 && !abl_first) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[i][ks] = *reinterpret_cast<const vec<T, 8>*>(part + a_row0 + i * 4096 + fo[ks]);
  };
  auto read_B = [&](const char* part) STAR_ALWAYS_INLINE {
    if (ABL >= 2 && ABL <= 5 && !abl_first) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<const vec<T, 8>*>(part + b_row0 + fo[ks]);
  };
  auto mfma_q = [&](f32x16 (&ac)[2]) STAR_ALWAYS_INLINE {
    STAR_SETPRIO(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) ac[i] = mfma32<T>(b[ks], a[i][ks], ac[i]);   // swapped: rows = n, cols = m (epilogue layout of gemm.h)
    STAR_SETPRIO(0);
  };

  auto bar1 = [&]() STAR_ALWAYS_INLINE { if (ABL != 4) raw_barrier(); };
  auto bar2 = [&]() STAR_ALWAYS_INLINE { if (ABL != 4 && ABL != 5) raw_barrier(); };
  // ---- stream positions: cur = the K tile being multiplied, n1 / n2 = one / two K tiles ahead
  Pos cur{};
  cur.it = 0;
  enter_tile(cur);
  Pos n1 = cur; advance(n1);
  Pos n2 = n1; advance(n2);

  // ---- prologue: K tile 0 completely, A0 / B1 of K tile 1
  a_issue(cur, 0, 0, G8::OFF_A0); b_issue(cur, 0, 1, G8::OFF_B1); a_issue(cur, 0, 1, G8::OFF_A1); b_issue(cur, 0, 0, G8::OFF_B0);
  a_issue(n1, 1, 0, G8::OFF_A0); b_issue(n1, 1, 1, G8::OFF_B1);
  if (n1.valid) { STAR_WAIT_VMCNT(4); } else { STAR_WAIT_VMCNT(0); }
  raw_barrier();
  if (ABL != 9 && wr == 1) raw_barrier();   // wave row 1 runs half a phase behind wave row 0

  zero_acc();
  const int S = my_tiles * nk;
#pragma clang loop unroll(disable)
  for (int seq = 0; seq < S; ++seq) {
    const char* buf = smem + (seq & 1) * G8::BUF;
    // ---- phase 0: quadrant (A0, B0)
    read_B(buf + G8::OFF_B0);
    read_A(buf + G8::OFF_A0);
    a_issue(n1, seq + 1, 1, G8::OFF_A1);
    bar1();
    mfma_q(acc[0][0]);
    bar2();
    // ---- phase 1: (A0, B1)
    read_B(buf + G8::OFF_B1);
    if (cur.kt == 0) {   // this tile's bias slice (64 columns of this wave), by 4-byte LDS-DMA; covered by phase 3's wait
      const int n = cur.n0 + wc * 64 + lane;
      const void* src = ((p.epi & EPI_BIAS) && n < p.N) ? (const void*)(p.bias + n) : (const void*)((const char*)p.zero_page + lane * 4);
      glds4(src, smem + G8::SMEM_BIAS + wave * 256);
    }
    b_issue(n1, seq + 1, 0, G8::OFF_B0);
    bar1();
    mfma_q(acc[0][1]);
    bar2();
    // ---- phase 2: (A1, B1)
    read_A(buf + G8::OFF_A1);
    a_issue(n2, seq, 0, G8::OFF_A0);
    bar1();
    mfma_q(acc[1][1]);
    bar2();
    // ---- phase 3: (A1, B0); the next K tile has landed after this wait
    read_B(buf + G8::OFF_B0);
    b_issue(n2, seq, 1, G8::OFF_B1);
    if (n2.valid) { if (ABL == 7 || ABL == 8) { STAR_WAIT_VMCNT(2); } else { STAR_WAIT_VMCNT(4); } } else { STAR_WAIT_VMCNT(0); }
    bar1();
    mfma_q(acc[1][0]);
    bar2();

    if (cur.kt == nk - 1) {
      // ---------------------------------------------------------------- epilogue (wave-private, no workgroup barrier)
      // lane holds, for row i*32 + frow of an (A sub, B sub) quadrant: cols 8*g + 4*fhalf + (0..3), g = 0..3 (acc regs 4g..4g+3)
      const bool geglu = (p.epi & EPI_GEGLU) != 0;
      const int cpr = geglu ? 4 : 8;                          // 16-B chunks per output row of this wave
      const int out_n0 = geglu ? (cur.n0 + wc * 64) / 2 : (cur.n0 + wc * 64);
      const int N_out = geglu ? p.N / 2 : p.N;
      const float* bias_lds = reinterpret_cast<const float*>(smem + G8::SMEM_BIAS + wave * 256);
      char* my = smem + G8::SMEM_EPI + wave * G8::EPI_WAVE;
      const int row_base = cur.m0 + wr * 128;

      // every global access of the epilogue goes through a buffer descriptor and is UNCONDITIONAL: rows >= M fall outside the
      // descriptor's range, columns >= N_out (and the unused second chunk of the GEGLU layout) get an out-of-range offset.
      // With no exec-masked branch around a VMEM instruction hipcc's vmcnt waits are exact counts, so a residual load only
      // waits for what was issued before it and never for the stores of the previous step.
      const int rows_left = p.M - row_base;                    // <= 0 for a wave row that lies entirely below the matrix
      const int nrows = rows_left < 0 ? 0 : (rows_left < 128 ? rows_left : 128);
      const BufRsrc c_rs = make_rsrc((const T*)p.C + (size_t)row_base * p.ldc, (uint32_t)nrows * (uint32_t)p.ldc * 2u);
      const BufRsrc r_rs = make_rsrc(RES ? (const void*)((const T*)p.res + (size_t)row_base * p.ldr) : p.zero_page,
                                     RES ? (uint32_t)nrows * (uint32_t)p.ldr * 2u : 0u);
      uint32_t c_off[2], r_off[2];   // byte offsets of this lane's two chunks within a 16-row step (+ step * 16 * ld * 2)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = lane + 64 * u;
        const int row = q / cpr, cc = q - row * cpr;
        const int n = out_n0 + cc * 8;
        const bool ok = q < 16 * cpr && n < N_out;
        c_off[u] = ok ? (uint32_t)(row * p.ldc + n) * 2u : 0x80000000u;   // stays out of range after + step * stride (no 32-bit wrap)
        r_off[u] = ok ? (uint32_t)(row * p.ldr + n) * 2u : 0x80000000u;   // stays out of range after + step * stride (no 32-bit wrap)
      }
      const uint32_t c_step = (uint32_t)p.ldc * 32u, r_step = (uint32_t)p.ldr * 32u;
      const int rd_row[2] = {lane / cpr, (lane + 64) / cpr};
      u32x4 rv[2][2];
      if constexpr (RES) {
#pragma unroll
        for (int u = 0; u < 2; ++u) rv[0][u] = buf_load16(r_rs, r_off[u]);
      }
#pragma unroll
      for (int step = 0; step < 8; ++step) {   // 16 rows each: step = sa*4 + i*2 + h
        const int sa = step >> 2, i = (step >> 1) & 1, h = step & 1;
        if ((frow >> 4) == h) {
#pragma unroll
          for (int sb = 0; sb < 2; ++sb) {
            if (geglu && sb == 1) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nl = sb * 32 + 8 * g + 4 * fhalf;
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[sa][sb][i][g * 4 + e];
              v += *reinterpret_cast<const f32x4*>(bias_lds + nl);
              if (geglu) {
                f32x4 gt;
#pragma unroll
                for (int e = 0; e < 4; ++e) gt[e] = acc[sa][1][i][g * 4 + e];
                gt += *reinterpret_cast<const f32x4*>(bias_lds + 32 + nl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * gelu_erf(gt[e]);
              }
              vec<T, 4> o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
              *reinterpret_cast<vec<T, 4>*>(my + (frow & 15) * G8::EPI_PITCH + nl * 2) = o;
            }
          }
        }
        wave_lds_fence();
        if constexpr (RES) if (step + 1 < 8) {   // the next step's residual chunks, ahead of this step's stores
#pragma unroll
          for (int u = 0; u < 2; ++u) rv[(step + 1) & 1][u] = buf_load16(r_rs, r_off[u] + (uint32_t)(step + 1) * r_step);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = lane + 64 * u;
          const int cc = q - rd_row[u] * cpr;
          const char* src = my + (rd_row[u] & 15) * G8::EPI_PITCH + cc * 16;
          const vec<T, 4> lo = *reinterpret_cast<const vec<T, 4>*>(src);
          const vec<T, 4> hi = *reinterpret_cast<const vec<T, 4>*>(src + 8);
          vec<T, 8> ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ov[e] = lo[e]; ov[4 + e] = hi[e]; }
          if constexpr (RES) {
            const vec<T, 8> rr = __builtin_bit_cast(vec<T, 8>, rv[step & 1][u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = from_f32<T>(to_f32<T>(ov[e]) + to_f32<T>(rr[e]));
          }
          buf_store16(c_rs, c_off[u] + (uint32_t)step * c_step, __builtin_bit_cast(u32x4, ov));
        }
        wave_lds_fence();
      }
      zero_acc();
    }
    cur = n1; n1 = n2; advance(n2);
    abl_first = false;
  }
  if (ABL != 9 && wr == 0) raw_barrier();
}

}  // namespace star
