// SIMT emulator runtime (see star_amd/csrc/hostemu.h).  Test tooling only.
#define STAR_HOSTEMU 1
#include "../../star_amd/csrc/hostemu.h"

extern "C" void star_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl star_emu_switch
.type star_emu_switch,@function
star_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size star_emu_switch, .-star_emu_switch
)");

namespace star_emu {

static thread_local Block* g_block = nullptr;
static thread_local Fiber* g_fiber = nullptr;
Block*& cur_block() { return g_block; }
Fiber*& cur_fiber() { return g_fiber; }

static constexpr size_t kStack = 256 * 1024;

void yield() {
  Block* b = g_block;
  Fiber* f = g_fiber;
  star_emu_switch(&f->sp, b->sched_sp);
}

static void fiber_main() {
  Block* b = g_block;
  Fiber* f = g_fiber;
  b->entry(b->entry_arg);
  f->done = true;
  b->alive--;
  b->waves[f->wave].alive--;
  // a thread that exits releases barriers it would otherwise block
  if (b->alive > 0 && b->bar_arrived >= b->alive && b->bar_arrived > 0) {
    b->bar_arrived = 0;
    b->bar_gen++;
  }
  Wave& w = b->waves[f->wave];
  if (w.alive > 0 && w.arrived >= w.alive && w.arrived > 0) {
    w.arrived = 0;
    w.gen++;
  }
  star_emu_switch(&f->sp, b->sched_sp);
  fprintf(stderr, "hostemu: resumed a finished fiber\n");
  abort();
}

void block_sync() {
  Block* b = g_block;
  unsigned my = b->bar_gen;
  b->bar_arrived++;
  if (b->bar_arrived >= b->alive) {
    b->bar_arrived = 0;
    b->bar_gen++;
    return;
  }
  while (b->bar_gen == my) yield();
}

const unsigned char (*wave_exchange(const void* mine, int bytes))[256] {
  Block* b = g_block;
  Fiber* f = g_fiber;
  Wave& w = b->waves[f->wave];
  assert(bytes <= 256);
  unsigned my = w.gen;
  int par = my & 1;
  memcpy(w.stage[par][f->lane], mine, bytes);
  w.arrived++;
  if (w.arrived >= w.alive) {
    w.arrived = 0;
    w.gen++;
  } else {
    while (w.gen == my) yield();
  }
  return w.stage[par];
}

static thread_local std::vector<char*> g_stacks;

void run_grid(Dim3 grid, Dim3 block, size_t smem, void (*entry)(void*), void* arg) {
  const int nthreads = block.x * block.y * block.z;
  const int nwaves = (nthreads + 63) / 64;
  while ((int)g_stacks.size() < nthreads) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, kStack)) abort();
    g_stacks.push_back((char*)p);
  }
  Block blk;
  blk.bdim = block;
  blk.gdim = grid;
  blk.entry = entry;
  blk.entry_arg = arg;
  blk.smem_bytes = smem;
  void* sm = nullptr;
  if (posix_memalign(&sm, 256, smem ? smem : 256)) abort();
  blk.smem = (char*)sm;
  blk.fibers.resize(nthreads);
  blk.waves.resize(nwaves);
  Block* saved_b = g_block;
  Fiber* saved_f = g_fiber;
  g_block = &blk;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blk.bid = Dim3(bx, by, bz);
        blk.alive = nthreads;
        blk.bar_arrived = 0;
        blk.bar_gen = 0;
        memset(blk.smem, 0xEE, smem ? smem : 256);  // poison LDS
        for (int w = 0; w < nwaves; ++w) {
          blk.waves[w].arrived = 0;
          blk.waves[w].gen = 0;
          blk.waves[w].alive = (w == nwaves - 1) ? nthreads - 64 * w : 64;
          memset(blk.waves[w].stage, 0xFF, sizeof(blk.waves[w].stage));
        }
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = blk.fibers[t];
          f.flat = t;
          f.lane = t & 63;
          f.wave = t >> 6;
          f.tid = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          f.done = false;
          f.stack = g_stacks[t];
          // initial frame: 6 callee-saved regs + return address into fiber_main
          uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
          uintptr_t* sp = (uintptr_t*)(top - 16);  // slot holding the return address (16B aligned)
          sp[0] = (uintptr_t)&fiber_main;
          sp -= 6;
          for (int i = 0; i < 6; ++i) sp[i] = 0;
          f.sp = sp;
        }
        int remaining = nthreads;
        while (remaining > 0) {
          int progressed = 0;
          for (int t = 0; t < nthreads; ++t) {
            Fiber& f = blk.fibers[t];
            if (f.done) continue;
            g_fiber = &f;
            star_emu_switch(&blk.sched_sp, f.sp);
            ++progressed;
            if (f.done) --remaining;
          }
          if (!progressed) break;
        }
      }
  free(blk.smem);
  g_block = saved_b;
  g_fiber = saved_f;
}

}  // namespace star_emu
