// prim.h -- the primitive layer every star_amd kernel is written against.
//
// Device build (hipcc, gfx950): thin wrappers over CDNA4 builtins
//   MFMA 32x32x16 / 16x16x32 (f16, bf16), global_load_lds (16 B LDS-DMA),
//   ds_read_b64_tr_b16, v_permlane32_swap, wave shuffles.
// Host-emulator build (-DSTAR_HOSTEMU, tools/hostemu): the same API on the SIMT
//   emulator, used only to test kernel index logic on machines without a GPU.
#pragma once
#include <cstdint>
#include <cstddef>
#include <type_traits>
#include <utility>

#ifdef STAR_HOSTEMU
#include "hostemu.h"
#define STAR_DEV inline
#define STAR_GLOBAL
#define STAR_LAUNCH_BOUNDS(...)
#define threadIdx (::star_emu::cur_fiber()->tid)
#define blockIdx (::star_emu::cur_block()->bid)
#define blockDim (::star_emu::cur_block()->bdim)
#define gridDim (::star_emu::cur_block()->gdim)
using dim3 = ::star_emu::Dim3;
typedef void* hipStream_t;
#else
#include <hip/hip_runtime.h>
#define STAR_DEV __device__ __forceinline__
#define STAR_GLOBAL __global__
#define STAR_LAUNCH_BOUNDS(...) __launch_bounds__(__VA_ARGS__)
#endif

// lambdas that are expanded at several call sites of one kernel: never outline them (a call spills the live accumulators)
#define STAR_ALWAYS_INLINE __attribute__((always_inline))

namespace star {

using f16 = _Float16;
using bf16 = __bf16;

template <class T, int N>
using vec = T __attribute__((ext_vector_type(N)));
using f32x4 = vec<float, 4>;
using f32x16 = vec<float, 16>;
using u32x4 = vec<uint32_t, 4>;
using u32x2 = vec<uint32_t, 2>;

// ---------------------------------------------------------------- conversions
template <class T>
STAR_DEV float to_f32(T v);
template <>
STAR_DEV float to_f32<f16>(f16 v) { return (float)v; }
template <>
STAR_DEV float to_f32<float>(float v) { return v; }
template <>
STAR_DEV float to_f32<bf16>(bf16 v) {
#ifdef STAR_HOSTEMU
  uint16_t b = __builtin_bit_cast(uint16_t, v);
  return __builtin_bit_cast(float, (uint32_t)b << 16);
#else
  return (float)v;
#endif
}
template <class T>
STAR_DEV T from_f32(float v);
template <>
STAR_DEV f16 from_f32<f16>(float v) { return (f16)v; }
template <>
STAR_DEV float from_f32<float>(float v) { return v; }
template <>
STAR_DEV bf16 from_f32<bf16>(float v) {  // round-to-nearest-even, NaN preserved
#ifndef STAR_HOSTEMU
  return (bf16)v;   // gfx950: v_cvt_pk_bf16_f32
#endif
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return __builtin_bit_cast(bf16, (uint16_t)((u >> 16) | 0x40));
  u += 0x7fffu + ((u >> 16) & 1u);
  return __builtin_bit_cast(bf16, (uint16_t)(u >> 16));
}

// 16-byte global store that does not allocate in the L2 (global_store_dwordx4 ... nt)
template <class V>
STAR_DEV void store_nt(V* dst, V v) {
#ifdef STAR_HOSTEMU
  *dst = v;
#else
  __builtin_nontemporal_store(v, dst);
#endif
}
// fp32 accumulate of a 16-bit pair: c + a.x b.x + a.y b.y (v_dot2_f32_f16 / v_dot2_f32_bf16) and c + a.x + a.y
template <class T>
STAR_DEV float dot2_acc(vec<T, 2> a, vec<T, 2> b, float c) {
#ifndef STAR_HOSTEMU
  if constexpr (__is_same(T, f16)) return __builtin_amdgcn_fdot2(a, b, c, false);
  else if constexpr (__is_same(T, bf16)) return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
  else
#endif
  return c + (to_f32<T>(a[0]) * to_f32<T>(b[0]) + to_f32<T>(a[1]) * to_f32<T>(b[1]));
}
template <class T>
STAR_DEV float dot2_one(vec<T, 2> a, float c) {
  vec<T, 2> one;
  one[0] = from_f32<T>(1.0f); one[1] = from_f32<T>(1.0f);
  return dot2_acc<T>(a, one, c);
}

STAR_DEV int lane_id() {
#ifdef STAR_HOSTEMU
  return ::star_emu::cur_fiber()->lane;
#else
  return (int)(threadIdx.x & 63);
#endif
}

// ---------------------------------------------------------------- workgroup
STAR_DEV char* dyn_smem() {
#ifdef STAR_HOSTEMU
  return ::star_emu::cur_block()->smem;
#else
  extern __shared__ __attribute__((aligned(16))) char star_smem_[];
  return star_smem_;
#endif
}
STAR_DEV void block_sync() {
#ifdef STAR_HOSTEMU
  ::star_emu::block_sync();
#else
  __syncthreads();
#endif
}

// ---------------------------------------------------------------- MFMA
// D = A(32x16) * B(16x32) + C.  lane l: a[j] = A[l&31][8*(l>>5)+j], b[j] = B[8*(l>>5)+j][l&31],
// c[r] = C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]   (verified: profiles/r01_probe_primitives.txt)
template <class T>
STAR_DEV f32x16 mfma32(vec<T, 8> a, vec<T, 8> b, f32x16 c) {
#ifdef STAR_HOSTEMU
  struct P { float a[8], b[8]; } mine;
  for (int j = 0; j < 8; ++j) { mine.a[j] = to_f32<T>(a[j]); mine.b[j] = to_f32<T>(b[j]); }
  auto st = ::star_emu::wave_exchange(&mine, sizeof(mine));
  const int l = lane_id();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float s = c[r];
    for (int k = 0; k < 16; ++k) {
      const P* pa = reinterpret_cast<const P*>(st[row + 32 * (k >> 3)]);
      const P* pb = reinterpret_cast<const P*>(st[col + 32 * (k >> 3)]);
      s += pa->a[k & 7] * pb->b[k & 7];
    }
    c[r] = s;
  }
  return c;
#else
  if constexpr (sizeof(T) == 2 && __is_same(T, bf16)) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// The same MFMA with its accumulator in ARCHITECTURAL registers (the "VGPR form").  hipcc selects one form per function (accumulation
// registers here), so a tile with more than 256 accumulators -- gemm.h's 4 x 5 scheduled tile keeps 64 of its 320 in v-registers -- names
// the form of those MFMAs itself; through the builtin the compiler copies the block into accumulation registers and back around every MFMA.
template <class T>
STAR_DEV void mfma32_vform(vec<T, 8> a, vec<T, 8> b, f32x16& c) {
#ifdef STAR_HOSTEMU
  c = mfma32<T>(a, b, c);
#else
  if constexpr (sizeof(T) == 2 && __is_same(T, bf16)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
}

// D = A(32x8) * B(8x32) + C (the pre-CDNA4 half-depth form).  lane l: a[j] = A[l&31][4*(l>>5)+j], b[j] = B[4*(l>>5)+j][l&31]
template <class T>
STAR_DEV f32x16 mfma32_k8(vec<T, 4> a, vec<T, 4> b, f32x16 c) {
#ifdef STAR_HOSTEMU
  struct P { float a[4], b[4]; } mine;
  for (int j = 0; j < 4; ++j) { mine.a[j] = to_f32<T>(a[j]); mine.b[j] = to_f32<T>(b[j]); }
  auto st = ::star_emu::wave_exchange(&mine, sizeof(mine));
  const int l = lane_id();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float s = c[r];
    for (int k = 0; k < 8; ++k) {
      const P* pa = reinterpret_cast<const P*>(st[row + 32 * (k >> 2)]);
      const P* pb = reinterpret_cast<const P*>(st[col + 32 * (k >> 2)]);
      s += pa->a[k & 3] * pb->b[k & 3];
    }
    c[r] = s;
  }
  return c;
#else
  if constexpr (__is_same(T, bf16)) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
  }
#endif
}

// D = A(16x32) * B(32x16) + C. lane l: a[j] = A[l&15][8*(l>>4)+j], b[j] = B[8*(l>>4)+j][l&15], c[r] = C[4*(l>>4)+r][l&15]
template <class T>
STAR_DEV f32x4 mfma16(vec<T, 8> a, vec<T, 8> b, f32x4 c) {
#ifdef STAR_HOSTEMU
  struct P { float a[8], b[8]; } mine;
  for (int j = 0; j < 8; ++j) { mine.a[j] = to_f32<T>(a[j]); mine.b[j] = to_f32<T>(b[j]); }
  auto st = ::star_emu::wave_exchange(&mine, sizeof(mine));
  const int l = lane_id();
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r, col = l & 15;
    float s = c[r];
    for (int k = 0; k < 32; ++k) {
      const P* pa = reinterpret_cast<const P*>(st[row + 16 * (k >> 3)]);
      const P* pb = reinterpret_cast<const P*>(st[col + 16 * (k >> 3)]);
      s += pa->a[k & 7] * pb->b[k & 7];
    }
    c[r] = s;
  }
  return c;
#else
  if constexpr (__is_same(T, bf16)) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// ---------------------------------------------------------------- LDS-DMA
// Asynchronous 16-byte-per-lane copy global -> LDS.  Destination is
// lds_wave_base + lane*16 (lds_wave_base must be wave-uniform); the source
// address is per lane.  Complete after glds_wait() + a barrier.
STAR_DEV void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef STAR_HOSTEMU
  struct P { const void* src; void* dst; } mine{gsrc, lds_wave_base};
  auto st = ::star_emu::wave_exchange(&mine, sizeof(mine));
  const int l = lane_id();
  // all lanes must agree on the base (wave-uniform) -- check against lane 0 of the live set
  const P* p0 = nullptr;
  for (int i = 0; i < 64 && !p0; ++i) {
    const P* q = reinterpret_cast<const P*>(st[i]);
    if (q->dst != (void*)~(uintptr_t)0) p0 = q;
  }
  if (p0 && p0->dst != lds_wave_base) { fprintf(stderr, "hostemu: glds16 LDS base is not wave-uniform\n"); abort(); }
  memcpy((char*)lds_wave_base + l * 16, gsrc, 16);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
// The same copy with the source split into a WAVE-UNIFORM base pointer and a 32-bit per-lane byte offset: the saddr form
// `global_load_lds_dwordx4 v_off, s[base:base+1]`.  hipcc never selects that form for the builtin inside a loop (its loop
// strength reduction turns base + offset into per-lane 64-bit pointers that cost 2 VGPRs and 1-2 VALU per copy and tile), so
// it is written out; M0 carries the LDS base.  The compiler does not see this asm's vmcnt traffic: a kernel that uses it
// must not have compiler-visible vector loads in flight around it (glds_wait() before the first call covers the prologue).
STAR_DEV void glds16_su(const void* ubase, uint32_t lane_off, void* lds_wave_base) {
#ifdef STAR_HOSTEMU
  glds16((const char*)ubase + lane_off, lds_wave_base);
#else
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds), "v"(lane_off), "s"(ubase) : "memory", "m0");
#endif
}
// glds16 written out (per-lane 64-bit source pointer), for kernels whose LDS-DMA must stay invisible to hipcc: one compiler-visible
// LDS-DMA anywhere in a kernel makes hipcc put s_waitcnt vmcnt(0) in front of the first LDS read behind every point where a DMA
// may be pending -- which drains the hand-counted asm copies that run TILES ahead as well (attn7.h)
STAR_DEV void glds16_v(const void* gsrc, void* lds_wave_base) {
#ifdef STAR_HOSTEMU
  glds16(gsrc, lds_wave_base);
#else
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds), "v"(gsrc) : "memory", "m0");
#endif
}
STAR_DEV void glds_wait() {
#ifndef STAR_HOSTEMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// counted wait on the vector-memory queue (LDS-DMA included): at most N of this wave's loads may still be in flight
#ifdef STAR_HOSTEMU
#define STAR_WAIT_VMCNT(N)
#else
#define STAR_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#endif
// same with a compile-time-computed count (inline asm needs a literal)
#ifdef STAR_HOSTEMU
#define STAR_WAIT_VMCNT_N(expr)
#else
#define STAR_WAIT_VMCNT_N(expr)                                                            \
  do {                                                                                     \
    constexpr int star_n_ = (expr);                                                        \
    static_assert(star_n_ >= 0 && star_n_ <= 63, "vmcnt count out of the counter's range"); \
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(star_n_) : "memory");                        \
  } while (0)
#endif
// workgroup barrier WITHOUT the vmcnt(0) drain that __syncthreads() implies while LDS-DMA is in flight: own LDS
// accesses are retired (lgkmcnt(0)), DMA stays in flight across the barrier
STAR_DEV void barrier_keep_dma() {
#ifdef STAR_HOSTEMU
  ::star_emu::block_sync();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#endif
}

// workgroup barrier with NO wait at all in front of it (raw s_barrier): LDS-DMA and this wave's own ds_reads stay in flight
// across it; the compiler still waits (lgkmcnt) before the first use of an LDS read's result.  Only for schedules whose
// LDS hazards are covered by construction (the epilogue staging of gemm.h).
STAR_DEV void raw_barrier() {
#ifdef STAR_HOSTEMU
  ::star_emu::block_sync();
#else
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}
// wave-private LDS hand-over between lanes of ONE wave (ds_write by some lanes, ds_read of the same bytes by others): DS
// operations of a wave execute in order, so only the compiler has to be kept from reordering them
STAR_DEV void wave_lds_fence() {
#ifdef STAR_HOSTEMU
  int v = 0;
  (void)::star_emu::wave_exchange(&v, 4);
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// a wait on the LDS / scalar-memory counter the compiler KNOWS about (it books builtin waits, not asm ones): placed where
// nothing can still be in flight, it keeps hipcc from inserting its own conservative lgkmcnt(0) behind freshly issued reads
#ifdef STAR_HOSTEMU
#define STAR_WAIT_LGKM0()
#else
#define STAR_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#endif
// ordering point between DS operations of ONE wave that touch the same bytes from different lanes, where no wait is wanted: the
// hardware executes a wave's DS operations in order, so on the device this only stops the compiler from reordering them; the
// emulator (lanes are fibers) needs the rendezvous
STAR_DEV void wave_lds_order() {
#ifdef STAR_HOSTEMU
  int v = 0;
  (void)::star_emu::wave_exchange(&v, 4);
#else
  asm volatile("" ::: "memory");
#endif
}
#ifdef STAR_HOSTEMU
#define STAR_SETPRIO(n)
#else
#define STAR_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

// ---------------------------------------------------------------- buffer (descriptor) access
// the wave gives up its issue slot for ~64 n cycles (n <= 127): start-phase staggering of resident workgroups
STAR_DEV void wave_sleep(int n) {
#ifndef STAR_HOSTEMU
  for (; n > 0; n -= 127) __builtin_amdgcn_s_sleep(127);
#else
  (void)n;
#endif
}
// 16-byte loads / stores through a buffer descriptor: lanes whose byte offset is >= the descriptor's range read zeros /
// store nothing, so row and column tails need no exec-masked branches (hipcc then counts every vmcnt wait exactly).
// base and bytes must be wave-uniform.
#ifdef STAR_HOSTEMU
struct BufRsrc { char* base; uint32_t bytes; };
STAR_DEV BufRsrc make_rsrc(const void* base, uint32_t bytes) { return BufRsrc{(char*)base, bytes}; }
STAR_DEV u32x4 buf_load16(BufRsrc r, uint32_t voff) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if ((uint64_t)voff + 16 <= r.bytes) memcpy(&v, r.base + voff, 16);
  return v;
}
STAR_DEV void buf_store16(BufRsrc r, uint32_t voff, u32x4 v) {
  if ((uint64_t)voff + 16 <= r.bytes) memcpy(r.base + voff, &v, 16);
}
template <int AUX> STAR_DEV void buf_store16_pol(BufRsrc r, uint32_t voff, u32x4 v) { buf_store16(r, voff, v); }
#else
using BufRsrc = __amdgpu_buffer_rsrc_t;
STAR_DEV BufRsrc make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
STAR_DEV u32x4 buf_load16(BufRsrc r, uint32_t voff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0); }
STAR_DEV void buf_store16(BufRsrc r, uint32_t voff, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, 0); }
// the same store with a cache policy (gfx94x / gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1): 2 = non-temporal (streamed past the L2's
// LRU), 16 = write-through to memory, 19 = all three
template <int AUX> STAR_DEV void buf_store16_pol(BufRsrc r, uint32_t voff, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, AUX); }
#endif

// LDS-DMA through a buffer descriptor (`buffer_load_dwordx4 v_off, s[rsrc], 0 offen lds`): like glds16, but the source is
// descriptor base + 32-bit lane offset and a lane whose offset is out of the descriptor's range writes ZEROS to its LDS
// slot -- padding taps of the implicit-GEMM gathers need no zero page and no 64-bit address select.
STAR_DEV void glds16_buf(BufRsrc r, uint32_t voff, void* lds_wave_base) {
#ifdef STAR_HOSTEMU
  struct P { void* dst; } mine{lds_wave_base};
  (void)::star_emu::wave_exchange(&mine, sizeof(mine));
  u32x4 v = buf_load16(r, voff);
  memcpy((char*)lds_wave_base + lane_id() * 16, &v, 16);
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0, 0, 0);
#endif
}
constexpr uint32_t GLDS_BUF_RANGE = 0xFFFF0000u;   // descriptor range used by the gathers: every valid offset is below it ...
constexpr uint32_t GLDS_BUF_OOB = 0xFFFFFFF0u;     // ... and this one is outside (reads as zeros)

// A hand-built buffer descriptor for LDS-DMA that must stay INVISIBLE to hipcc's vmcnt bookkeeping (the persistent GEMM of
// gemm_p.h keeps a hand-counted DMA stream running across output tiles): base and bytes wave-uniform, raw addressing (stride 0).
// A lane whose byte offset voff is >= bytes writes ZEROS to its LDS slot and touches no memory; soff is a wave-uniform extra
// offset (the K tile) that stays inside a row, so rows at or past the end of the matrix are out of range whether or not the
// hardware counts soff in the check.
#ifdef STAR_HOSTEMU
struct BufDesc { const char* base; uint32_t bytes; };
STAR_DEV BufDesc make_desc(const void* base, uint32_t bytes) { return BufDesc{(const char*)base, bytes}; }
STAR_DEV void glds16_desc(BufDesc d, uint32_t voff, uint32_t soff, void* lds_wave_base) {
  struct P { void* dst; } mine{lds_wave_base};
  (void)::star_emu::wave_exchange(&mine, sizeof(mine));
  u32x4 v = {0u, 0u, 0u, 0u};
  if ((uint64_t)voff + 16 <= d.bytes) memcpy(&v, d.base + soff + voff, 16);
  memcpy((char*)lds_wave_base + lane_id() * 16, &v, 16);
}
STAR_DEV void glds4_desc(BufDesc d, uint32_t voff, uint32_t soff, void* lds_wave_base) {
  struct P { void* dst; } mine{lds_wave_base};
  (void)::star_emu::wave_exchange(&mine, sizeof(mine));
  uint32_t v = 0;
  if ((uint64_t)voff + 4 <= d.bytes) memcpy(&v, d.base + soff + voff, 4);
  memcpy((char*)lds_wave_base + lane_id() * 4, &v, 4);
}
#else
using BufDesc = u32x4;
STAR_DEV BufDesc make_desc(const void* base, uint32_t bytes) {
  const uint64_t b = (uint64_t)(uintptr_t)base;
  BufDesc d;
  d[0] = (uint32_t)b; d[1] = (uint32_t)(b >> 32) & 0xffffu; d[2] = bytes; d[3] = 0x00020000u;
  return d;
}
STAR_DEV void glds16_desc(BufDesc d, uint32_t voff, uint32_t soff, void* lds_wave_base) {
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds), "v"(voff), "s"(d), "s"(soff) : "memory", "m0");
}
STAR_DEV void glds4_desc(BufDesc d, uint32_t voff, uint32_t soff, void* lds_wave_base) {   // 4 bytes per lane (bias slices)
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds), "v"(voff), "s"(d), "s"(soff) : "memory", "m0");
}
#endif

// ---------------------------------------------------------------- transpose read
// ds_read_b64_tr_b16: every lane passes the (8-byte aligned) LDS address of 4
// contiguous 16-bit elements P[lane][0..3]; within each 16-lane group lane i
// receives R[j] = P[group + 4*j + i/4][i%4]   (verified on MI355X by the probe).
template <class T>
STAR_DEV vec<T, 4> lds_read_tr(const void* lds_addr) {
#ifdef STAR_HOSTEMU
  uint64_t mine;
  memcpy(&mine, lds_addr, 8);
  if (((uintptr_t)lds_addr) & 7) { fprintf(stderr, "hostemu: lds_read_tr address not 8B aligned\n"); abort(); }
  auto st = ::star_emu::wave_exchange(&mine, 8);
  const int l = lane_id(), g = l & ~15, i = l & 15;
  vec<T, 4> r;
  for (int j = 0; j < 4; ++j) {
    uint16_t e;
    memcpy(&e, st[g + 4 * j + i / 4] + 2 * (i % 4), 2);
    r[j] = __builtin_bit_cast(T, e);
  }
  return r;
#else
  typedef short s4 __attribute__((ext_vector_type(4)));
  s4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)lds_addr);
  return __builtin_bit_cast(vec<T, 4>, t);
#endif
}

// ---------------------------------------------------------------- cross-lane
STAR_DEV float shfl_xor(float v, int mask) {
#ifdef STAR_HOSTEMU
  auto st = ::star_emu::wave_exchange(&v, 4);
  float r;
  memcpy(&r, st[lane_id() ^ mask], 4);
  return r;
#else
  return __shfl_xor(v, mask, 64);
#endif
}
STAR_DEV float shfl(float v, int src_lane) {
#ifdef STAR_HOSTEMU
  auto st = ::star_emu::wave_exchange(&v, 4);
  float r;
  memcpy(&r, st[src_lane & 63], 4);
  return r;
#else
  return __shfl(v, src_lane, 64);
#endif
}
// returns {r0, r1}: r0[l<32]=a[l], r0[l>=32]=b[l-32]; r1[l<32]=a[l+32], r1[l>=32]=b[l]
STAR_DEV u32x2 permlane32_swap(uint32_t a, uint32_t b) {
#ifdef STAR_HOSTEMU
  uint32_t mine[2] = {a, b};
  auto st = ::star_emu::wave_exchange(mine, 8);
  const int l = lane_id();
  uint32_t o[2];
  memcpy(o, st[l ^ 32], 8);
  u32x2 r;
  r[0] = l < 32 ? a : o[1];
  r[1] = l < 32 ? o[0] : b;
  return r;
#else
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  u32x2 o;
  o[0] = r[0];
  o[1] = r[1];
  return o;
#endif
}
// lane l <-> lane l^32 exchange through v_permlane32_swap (one VALU op, no LDS): returns the partner's value
STAR_DEV float xor32(float v) {
#ifdef STAR_HOSTEMU
  return shfl_xor(v, 32);
#else
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r0 = {own | lower's}, r1 = {upper's | own}
  const uint32_t o = (threadIdx.x & 32) ? r[0] : r[1];
  return __builtin_bit_cast(float, o);
#endif
}
// max / sum of a value with its lane^32 partner (both halves get the same result)
STAR_DEV float pair_max(float v) {
#ifdef STAR_HOSTEMU
  return fmaxf(v, shfl_xor(v, 32));
#else
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
#endif
}
STAR_DEV float pair_sum(float v) {
#ifdef STAR_HOSTEMU
  return v + shfl_xor(v, 32);
#else
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
#endif
}
// acc + a[0] + a[1] in fp32 (v_dot2c_f32_f16 / v_dot2c_f32_bf16 against packed ones): row sums of packed probabilities
template <class T>
STAR_DEV float dot2_ones(vec<T, 2> a, float acc) {
#ifdef STAR_HOSTEMU
  return acc + to_f32<T>(a[0]) + to_f32<T>(a[1]);
#else
  if constexpr (__is_same(T, bf16)) {
    vec<bf16, 2> one; one[0] = (bf16)1.0f; one[1] = (bf16)1.0f;
    return __builtin_amdgcn_fdot2_f32_bf16(a, one, acc, false);
  } else {
    vec<f16, 2> one; one[0] = (f16)1.0f; one[1] = (f16)1.0f;
    return __builtin_amdgcn_fdot2(a, one, acc, false);
  }
#endif
}
// hide a (loop-invariant) LDS address from the optimiser: it stays in its register instead of being rematerialised from its
// terms at every use
STAR_DEV const char* opaque(const char* p) {
#ifndef STAR_HOSTEMU
  auto l = (__attribute__((address_space(3))) const char*)p;
  asm volatile("" : "+v"(l));
  return (const char*)l;
#else
  return p;
#endif
}
// an integer the optimiser cannot see through (blocks loop-invariant code motion of everything derived from it)
STAR_DEV int opaque_int(int v) {
#ifdef STAR_HOSTEMU
  asm volatile("" : "+r"(v));
#else
  asm volatile("" : "+v"(v));
#endif
  return v;
}
// make a 32-bit value exist in a VGPR at this point of the instruction stream (no instruction): its producer cannot sink below,
// its consumers cannot rise above
template <class V>
STAR_DEV V pin_here(V v) {
#ifndef STAR_HOSTEMU
  if constexpr (sizeof(V) == 4) {
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    asm volatile("" : "+v"(u));
    return __builtin_bit_cast(V, u);
  } else {
    asm volatile("" : "+v"(v));
    return v;
  }
#else
  return v;
#endif
}
// tell the compiler a value is wave-uniform (v_readfirstlane); identity on the emulator
STAR_DEV int wave_uniform(int v) {
#ifdef STAR_HOSTEMU
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}
// wave-uniform "any lane has pred"
STAR_DEV bool wave_any(bool pred) {
#ifdef STAR_HOSTEMU
  int v = pred ? 1 : 0;
  auto st = ::star_emu::wave_exchange(&v, 4);
  bool any = false;
  for (int i = 0; i < 64; ++i) { int x; memcpy(&x, st[i], 4); if (x == 1) any = true; }
  return any;
#else
  return __any(pred ? 1 : 0) != 0;
#endif
}
// wave-uniform bit mask of the lanes with pred (v_cmp into an SGPR pair); the value is pinned so the compare is issued HERE and a
// later `if (mask)` is only s_cmp + s_cbranch
STAR_DEV uint64_t wave_ballot(bool pred) {
#ifdef STAR_HOSTEMU
  return wave_any(pred) ? 1ull : 0ull;
#else
  uint64_t m = __ballot(pred ? 1 : 0);
  asm volatile("" : "+s"(m));
  return m;
#endif
}
// packed 16-bit add (v_pk_add_f16; attn5.h's row sums); the emulator rounds each lane's sum to T exactly like the instruction does
template <class T>
STAR_DEV vec<T, 2> pk_add(vec<T, 2> a, vec<T, 2> b) {
#ifdef STAR_HOSTEMU
  vec<T, 2> r;
  r[0] = from_f32<T>(to_f32<T>(a[0]) + to_f32<T>(b[0]));
  r[1] = from_f32<T>(to_f32<T>(a[1]) + to_f32<T>(b[1]));
  return r;
#else
  return a + b;
#endif
}
// two fp32 -> packed f16, rounded TOWARD ZERO (v_cvt_pkrtz_f16_f32: one issue slot where the nearest-even v_cvt_pk_f16_f32 of gfx950
// takes two; attn5.h RTZ).  The emulator truncates the nearest-even result back when it overshot; inf / NaN pass through; a finite value
// beyond the f16 range truncates to the largest finite f16 like the instruction does.
template <class T>
STAR_DEV vec<T, 2> cvt_pkrtz(float a, float b) {
  static_assert(__is_same(T, f16), "");
#ifdef STAR_HOSTEMU
  auto one = [](float v) -> f16 {
    f16 r = (f16)v;
    if (v != v) return r;
    const float back = (float)r;
    const bool finite_in = v - v == 0.f;
    if (finite_in && (back - back != 0.f || __builtin_fabsf(back) > __builtin_fabsf(v))) {   // rounded away from zero (or to inf): one ulp back
      uint16_t u = __builtin_bit_cast(uint16_t, r);
      u = (uint16_t)(u - 1);
      r = __builtin_bit_cast(f16, u);
    }
    return r;
  };
  vec<T, 2> r; r[0] = one(a); r[1] = one(b);
  return r;
#else
  return __builtin_bit_cast(vec<T, 2>, __builtin_amdgcn_cvt_pkrtz(a, b));
#endif
}
// max of the 8 values of a 16-byte chunk (f16: a tree of v_pk_max_f16)
template <class T>
STAR_DEV float max8(vec<T, 8> v) {
#ifndef STAR_HOSTEMU
  if constexpr (__is_same(T, f16)) {
    vec<f16, 2> a, b, c, d;
    a[0] = v[0]; a[1] = v[1]; b[0] = v[2]; b[1] = v[3]; c[0] = v[4]; c[1] = v[5]; d[0] = v[6]; d[1] = v[7];
    a = __builtin_elementwise_max(a, b);
    c = __builtin_elementwise_max(c, d);
    a = __builtin_elementwise_max(a, c);
    return fmaxf((float)a[0], (float)a[1]);
  }
#endif
  float m = to_f32<T>(v[0]);
#pragma unroll
  for (int e = 1; e < 8; ++e) m = fmaxf(m, to_f32<T>(v[e]));
  return m;
}
// compile-time scheduling hint (LLVM sched_group_barrier): emit `n` instructions of class `mask` next
#ifdef STAR_HOSTEMU
#define STAR_SCHED_GROUP(mask, n, id)
#define STAR_SCHED_FENCE()
#else
#define STAR_SCHED_GROUP(mask, n, id) __builtin_amdgcn_sched_group_barrier((mask), (n), (id))
#define STAR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// move a value that only MFMAs read (A / B operand) into the accumulator half of the unified register file: MFMA operands are
// read from AGPRs in place, and the architectural VGPRs stay free for what vector instructions touch (one-wave-per-SIMD kernels)
#ifdef STAR_HOSTEMU
#define STAR_AGPR_PIN(x)
#define STAR_VGPR_PIN(x)
#else
#define STAR_AGPR_PIN(x) asm volatile("" : "+a"(x))
#define STAR_VGPR_PIN(x) asm volatile("" : "+v"(x))   // ... in architectural registers (a 320-accumulator tile keeps 64 of them there)
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... Is>
STAR_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
STAR_DEV void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

STAR_DEV float wave_sum(float v) {
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}
STAR_DEV float wave_max(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
  return v;
}

// ---------------------------------------------------------------- atomics / math
STAR_DEV void atomic_add(double* p, double v) {
#ifdef STAR_HOSTEMU
  *p += v;
#else
  atomicAdd(p, v);
#endif
}
STAR_DEV void atomic_add(float* p, float v) {
#ifdef STAR_HOSTEMU
  *p += v;
#else
  atomicAdd(p, v);
#endif
}
STAR_DEV float fast_exp2(float x) {
#ifdef STAR_HOSTEMU
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
STAR_DEV float fast_rcp(float x) {
#ifdef STAR_HOSTEMU
  return 1.0f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}

}  // namespace star

// ---------------------------------------------------------------- launch
namespace star {
// first failed launch since the last check (name of the kernel + HIP error string); defined in api.cpp
void rt_note_launch_error(const char* what);
}
#ifdef STAR_HOSTEMU
#define STAR_LAUNCH(kern, grid, block, smem, stream, ...) \
  ::star_emu::launch((grid), (block), (smem), [&]() { kern(__VA_ARGS__); })
#else
// The dynamic-LDS opt-in (> 64 KB) is a per-DEVICE property of the function object: it is cached per kernel instantiation, call
// site and device (a process may hold contexts on several GPUs: one table entry per device id below 64, no sharing; the cache is a racy-but-idempotent hint: at worst the attribute
// is set twice).  A refused attribute or launch is remembered in star::rt::launch_error (checked at the end of every C-ABI call:
// include/star_hip.h), never dropped.
#define STAR_LAUNCH(kern, grid, block, smem, stream, ...)                                              \
  do {                                                                                                  \
    if ((smem) > 65536) {                                                                               \
      static size_t star_attr_set_[64] = {0};                                                           \
      size_t star_uncached_ = 0;                                                                        \
      int star_dev_ = 0;                                                                                \
      (void)hipGetDevice(&star_dev_);                                                                   \
      /* one entry per visible device; a device id past the table is simply not cached (attribute set per launch) */ \
      size_t& star_cur_ = (star_dev_ >= 0 && star_dev_ < 64) ? star_attr_set_[star_dev_] : star_uncached_; \
      if ((size_t)(smem) > star_cur_) {                                                                 \
        if (hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem)) != hipSuccess) \
          ::star::rt_note_launch_error(#kern ": dynamic LDS size refused");                             \
        star_cur_ = (size_t)(smem);                                                                     \
      }                                                                                                 \
    }                                                                                                   \
    (void)hipGetLastError();   /* a stale error of an earlier call (e.g. a failed hipMalloc the pool recovered from) is not ours */ \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__);                           \
    if (hipPeekAtLastError() != hipSuccess) ::star::rt_note_launch_error(#kern);                        \
  } while (0)
#endif
