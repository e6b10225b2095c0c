// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <mutex>

// x86-64 SysV context switch: save callee-saved regs on the old stack, swap sp.
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

namespace emu {

thread_local Worker* g_worker = nullptr;
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static void fiber_exit_bookkeeping(Worker* w, Fiber& f) {
  BlockState& b = w->blk;
  f.done = true;
  b.progress++;
  b.active--;
  if (b.active > 0 && b.arrived == b.active) {  // others are waiting at a block barrier
    b.arrived = 0;
    b.gen++;
  }
  WaveState& wv = b.waves[f.lin / kWave];
  wv.active--;
  if (wv.active > 0 && wv.arrived == wv.active) {
    wv.arrived = 0;
    wv.gen++;
  }
}

static void fiber_entry() {
  Worker* w = g_worker;
  Fiber& f = w->fibers[w->cur];
  (*w->body)();
  fiber_exit_bookkeeping(w, f);
  void* dummy;
  emu_switch(&dummy, w->sched_sp);
  abort();  // never resumed
}

void yield_to_sched() {
  Worker* w = g_worker;
  Fiber& f = w->fibers[w->cur];
  emu_switch(&f.sp, w->sched_sp);
}

void block_barrier() {
  BlockState& b = g_worker->blk;
  b.progress++;
  b.arrived++;
  unsigned gen = b.gen;
  if (b.arrived == b.active) {
    b.arrived = 0;
    b.gen++;
    return;
  }
  while (b.gen == gen) yield_to_sched();
}

void wave_barrier() {
  Worker* w = g_worker;
  WaveState& wv = w->blk.waves[w->fibers[w->cur].lin / kWave];
  w->blk.progress++;
  wv.arrived++;
  unsigned gen = wv.gen;
  if (wv.arrived == wv.active) {
    wv.arrived = 0;
    wv.gen++;
    return;
  }
  while (wv.gen == gen) yield_to_sched();
}

static void init_fiber(Fiber& f) {
  if (!f.stack) {
    f.stack = static_cast<char*>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (f.stack == MAP_FAILED) { perror("mmap"); abort(); }
  }
  uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
  void** sp = reinterpret_cast<void**>(top - 16);
  *sp = reinterpret_cast<void*>(&fiber_entry);  // return address for the first switch
  sp -= 6;                                       // r15,r14,r13,r12,rbx,rbp
  for (int i = 0; i < 6; ++i) sp[i] = nullptr;
  f.sp = sp;
  f.done = false;
}

static void run_block(Worker* w, dim3 grid, dim3 block, dim3 bid) {
  const int n = block.x * block.y * block.z;
  if ((int)w->fibers.size() < n) w->fibers.resize(n);
  BlockState& b = w->blk;
  b.nthreads = n;
  b.active = n;
  b.arrived = 0;
  b.gen = 0;
  b.progress = 0;
  const int nw = (n + kWave - 1) / kWave;
  b.waves.assign(nw, WaveState());
  for (int i = 0; i < n; ++i) {
    Fiber& f = w->fibers[i];
    init_fiber(f);
    f.lin = i;
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    b.waves[i / kWave].active++;
  }
  blockIdx = bid;
  blockDim = block;
  gridDim = grid;
  int remaining = n;
  while (remaining > 0) {
    unsigned long before = b.progress;
    remaining = 0;
    for (int i = 0; i < n; ++i) {
      Fiber& f = w->fibers[i];
      if (f.done) continue;
      w->cur = i;
      threadIdx = f.tid;
      emu_switch(&w->sched_sp, f.sp);
      if (!f.done) remaining++;
    }
    if (remaining > 0 && b.progress == before) {
      fprintf(stderr, "hip_emu: deadlock in block (%u,%u,%u): %d fibers stuck (divergent barrier?)\n", bid.x, bid.y, bid.z, remaining);
      abort();
    }
  }
}

static void free_worker(Worker* w) {
  for (auto& f : w->fibers)
    if (f.stack) munmap(f.stack, kStack);
  free(w->dyn_smem);
  delete w;
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  if (nblocks <= 0) return;
  const int n = block.x * block.y * block.z;
  if (n <= 0 || n > 1024) { fprintf(stderr, "hip_emu: bad block size %d\n", n); abort(); }
  static int max_workers = [] {
    const char* e = getenv("CMDA_EMU_THREADS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  int nworkers = (int)std::min<long>(nblocks, max_workers);
  std::atomic<long> next{0};
  auto work = [&]() {
    Worker* w = new Worker();
    g_worker = w;
    w->body = &body;
    w->dyn_smem_cap = smem + 64;
    w->dyn_smem = static_cast<char*>(aligned_alloc(64, (w->dyn_smem_cap + 63) / 64 * 64));
    for (;;) {
      long i = next.fetch_add(1);
      if (i >= nblocks) break;
      dim3 bid(i % grid.x, (i / grid.x) % grid.y, i / ((long)grid.x * grid.y));
      run_block(w, grid, block, bid);
    }
    g_worker = nullptr;
    free_worker(w);
  };
  if (nworkers == 1) {
    // run on a fresh thread anyway so `static thread_local` LDS never aliases the caller's
    std::thread t(work);
    t.join();
  } else {
    std::vector<std::thread> ts;
    for (int i = 0; i < nworkers; ++i) ts.emplace_back(work);
    for (auto& t : ts) t.join();
  }
}

}  // namespace emu
