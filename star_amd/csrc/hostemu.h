// hostemu.h -- a small SIMT emulator used ONLY by tools/hostemu (development /
// CPU-side index-logic tests).  It is never compiled into libstar_hip.so and is
// not a CPU fallback: the product library requires a gfx950 device.
//
// Each thread of a workgroup runs as a fiber (hand-rolled x86-64 context
// switch); workgroups run one after another.  __syncthreads() and the wave
// collectives (MFMA, shuffles, permlane, LDS-DMA, transpose reads) are
// rendezvous points.  The lane layouts implemented here are the ones verified
// on a real MI355X by tools/probe/probe.hip (profiles/r01_probe_primitives.txt).
#pragma once
#ifndef STAR_HOSTEMU
#error "hostemu.h is only for -DSTAR_HOSTEMU builds"
#endif
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace star_emu {

struct Dim3 {
  unsigned x = 1, y = 1, z = 1;
  Dim3() = default;
  Dim3(unsigned x_, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct Wave {
  // double-buffered staging for collectives: 64 lanes x 256 B
  alignas(16) unsigned char stage[2][64][256];
  int arrived = 0;
  unsigned gen = 0;
  int alive = 0;
};

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  Dim3 tid;
  int flat = 0, lane = 0, wave = 0;
  bool done = false;
};

struct Block {
  Dim3 bid, bdim, gdim;
  char* smem = nullptr;
  size_t smem_bytes = 0;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int alive = 0;
  int bar_arrived = 0;
  unsigned bar_gen = 0;
  void* sched_sp = nullptr;
  void (*entry)(void*) = nullptr;
  void* entry_arg = nullptr;
};

Block*& cur_block();
Fiber*& cur_fiber();
void yield();
void block_sync();
// all alive lanes of the calling lane's wave deposit `bytes` (<=256) and get a
// pointer to the wave's 64x256B staging area valid until the next collective.
const unsigned char (*wave_exchange(const void* mine, int bytes))[256];
void run_grid(Dim3 grid, Dim3 block, size_t smem, void (*entry)(void*), void* arg);

template <class F>
void launch(Dim3 grid, Dim3 block, size_t smem, F&& f) {
  auto tramp = [](void* p) { (*static_cast<F*>(p))(); };
  run_grid(grid, block, smem, tramp, &f);
}

}  // namespace star_emu
