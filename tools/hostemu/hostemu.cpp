// SIMT emulator runtime (see star_amd/csrc/hostemu.h).  Test tooling only.
#define STAR_HOSTEMU 1
#include "../../star_amd/csrc/hostemu.h"
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

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

// one workgroup, start to finish, on the calling host thread (its fibers' stacks, the Block and its LDS are that thread's own)
static void run_block(Block& blk, Dim3 bid, Dim3 block, size_t smem, int nthreads, int nwaves) {
  blk.bid = bid;
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

// the workgroups `next` hands out, one after another, on the calling host thread
static void run_blocks(Dim3 grid, Dim3 block, size_t smem, void (*entry)(void*), void* arg, std::atomic<long long>& next, long long total) {
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
  for (long long i = next.fetch_add(1); i < total; i = next.fetch_add(1)) {
    const unsigned bx = (unsigned)(i % grid.x), by = (unsigned)((i / grid.x) % grid.y), bz = (unsigned)(i / ((long long)grid.x * grid.y));
    run_block(blk, Dim3(bx, by, bz), block, smem, nthreads, nwaves);
  }
  free(blk.smem);
  g_block = saved_b;
  g_fiber = saved_f;
}

// Host threads per launch: workgroups of a grid are independent (the kernels have no inter-workgroup communication: no atomics,
// fixed-order reductions), so they are dealt to STAR_EMU_THREADS host threads (default: the host's cores, at most 8; 1 = the
// sequential order of rounds 1-3).  Every thread owns its Block, LDS and fiber stacks (thread_local above).
static int emu_threads() {
  static const int n = [] {
    const char* e = std::getenv("STAR_EMU_THREADS");
    int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    if (!e && v > 8) v = 8;
    return v < 1 ? 1 : v;
  }();
  return n;
}

// persistent workers (created on first use, never joined: they sleep on a condition variable between launches and die with the process)
struct Job {
  Dim3 grid, block;
  size_t smem = 0;
  void (*entry)(void*) = nullptr;
  void* arg = nullptr;
  std::atomic<long long>* next = nullptr;
  long long total = 0;
};
struct Pool {
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  Job job;
  unsigned long long gen = 0;
  int want = 0, running = 0;     // workers that should take the current job / have not finished it yet
  int nworkers = 0;
};
static Pool& pool() { static Pool* p = new Pool; return *p; }
static std::mutex& launch_mutex() { static std::mutex* m = new std::mutex; return *m; }   // one emulated launch at a time per process

static void worker_main() {
  Pool& P = pool();
  unsigned long long seen = 0;
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> lk(P.m);
      P.cv_job.wait(lk, [&] { return P.gen != seen && P.want > 0; });
      seen = P.gen;
      --P.want;
      j = P.job;
    }
    run_blocks(j.grid, j.block, j.smem, j.entry, j.arg, *j.next, j.total);
    {
      std::lock_guard<std::mutex> lk(P.m);
      if (--P.running == 0) P.cv_done.notify_all();
    }
  }
}

void run_grid(Dim3 grid, Dim3 block, size_t smem, void (*entry)(void*), void* arg) {
  const long long total = (long long)grid.x * grid.y * grid.z;
  std::atomic<long long> next{0};
  int nt = emu_threads();
  if (g_block != nullptr) nt = 1;                     // (a launch from inside a kernel does not exist; stay safe)
  if ((long long)nt > total) nt = (int)total;
  if (nt <= 1) {
    run_blocks(grid, block, smem, entry, arg, next, total);
    return;
  }
  std::lock_guard<std::mutex> launch(launch_mutex());   // contexts on several Python threads share the workers
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.m);
    while (P.nworkers < nt - 1) { std::thread(worker_main).detach(); ++P.nworkers; }
    P.job = Job{grid, block, smem, entry, arg, &next, total};
    P.want = P.running = nt - 1;
    ++P.gen;
  }
  P.cv_job.notify_all();
  run_blocks(grid, block, smem, entry, arg, next, total);
  std::unique_lock<std::mutex> lk(P.m);
  P.cv_done.wait(lk, [&] { return P.running == 0; });
}

}  // namespace star_emu
