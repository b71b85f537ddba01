// hipemu runtime: fibers, rendezvous, grid loop (see hip/hip_runtime.h in this directory).  TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>

namespace hipemu {

Block *g_blk = nullptr;
size_t g_dyn_shared_bytes = 0;
static const size_t kStack = 256 * 1024;

[[noreturn]] static void die(const char *what) {
  std::fprintf(stderr, "hipemu: %s\n", what);
  std::abort();
}

// Fiber switch without system calls (swapcontext saves / restores the signal mask with two syscalls per switch, which
// dominated the run time): the callee-saved registers go on the old stack, the stack pointers are exchanged.  x86-64 SysV.
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch, .-hipemu_switch\n");

static void *g_sched_sp = nullptr;

void yield() { hipemu_switch(&self().sp, g_sched_sp); }

// A rendezvous completes when every lane that is still alive has arrived.  Lanes that returned from the kernel no longer
// count (their exit re-evaluates pending rendezvous in the scheduler).
void block_barrier() {
  Block &b = *g_blk;
  const long gen = b.gen;
  ++b.arrived;
  while (b.gen == gen) {
    if (b.arrived >= b.alive) {
      b.arrived = 0;
      ++b.gen;
      break;
    }
    yield();
  }
}

void wave_barrier() {
  Block &b = *g_blk;
  const int w = wave_of();
  const long gen = b.w_gen[w];
  ++b.w_arrived[w];
  while (b.w_gen[w] == gen) {
    if (b.w_arrived[w] >= b.w_alive[w]) {
      b.w_arrived[w] = 0;
      ++b.w_gen[w];
      break;
    }
    yield();
  }
}

void check_same_kind(int kind) {
  Block &b = *g_blk;
  const int w = wave_of(), base = w * 64, n = (int)std::min<size_t>(64, b.lanes.size() - base);
  for (int l = 0; l < n; ++l) {
    if (b.lanes[base + l].done) continue;
    int k;
    std::memcpy(&k, b.xchg.data() + ((size_t)w * 64 + l) * 64 + 60, 4);
    if (k != kind) die("lanes of one wave met in different collectives (a collective inside divergent control flow?)");
  }
}

static bool glds_late() {
  static const int mode = [] {
    const char *e = std::getenv("HIPEMU_GLDS");
    return (e && std::strcmp(e, "late") == 0) ? 1 : 0;
  }();
  return mode == 1;
}
void glds_issue(const void *src, void *dst_lane, int bytes) {
  if (!glds_late()) {
    std::memcpy(dst_lane, src, (size_t)bytes);
    return;
  }
  self().vm.push_back({src, dst_lane, bytes});
}
// s_waitcnt vmcnt(n): the oldest operations complete until at most n are outstanding (loads return in order)
void vmcnt_wait(int n) {
  std::vector<PendingDma> &q = self().vm;
  const int done = (int)q.size() - n;
  if (done <= 0) return;
  for (int i = 0; i < done; ++i) std::memcpy(q[i].dst, q[i].src, (size_t)q[i].bytes);
  q.erase(q.begin(), q.begin() + done);
}

static void trampoline() {
  Block &b = *g_blk;
  b.body();
  Lane &me = self();
  me.done = true;
  --b.alive;
  --b.w_alive[flat_tid() >> 6];
  hipemu_switch(&me.sp, g_sched_sp);
  die("a finished lane was resumed");
}

static void run_block(Block &b) {
  const int n = (int)(b.bdim.x * b.bdim.y * b.bdim.z);
  const int nw = (n + 63) / 64;
  b.alive = n;
  b.arrived = 0;
  b.gen = 0;
  b.w_arrived.assign(nw, 0);
  b.w_gen.assign(nw, 0);
  b.w_alive.assign(nw, 0);
  for (int t = 0; t < n; ++t) ++b.w_alive[t >> 6];
  b.xchg.assign((size_t)nw * 64 * 64, 0);
  if ((int)b.lanes.size() < n) b.lanes.resize(n);
  for (int t = 0; t < n; ++t) {
    Lane &l = b.lanes[t];
    if (l.stack.size() != kStack) l.stack.resize(kStack);
    l.done = false;
    l.vm.clear();
    l.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
    // initial frame: six zeroed callee-saved registers, then the trampoline as the address `ret` jumps to; after that `ret`
    // the stack pointer is 8 below a 16-byte boundary, as at any function entry
    uintptr_t top = ((uintptr_t)l.stack.data() + l.stack.size()) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                       // return address slot of the trampoline (it never returns)
    *--sp = (void *)trampoline;
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    l.sp = (void *)sp;
  }
  // round-robin until every lane has returned; a full pass in which nobody makes progress cannot be detected cheaply, so a
  // generous pass limit guards against a rendezvous that can never complete (e.g. a collective inside divergent control flow)
  long passes = 0;
  while (b.alive > 0) {
    for (int t = 0; t < n; ++t) {
      if (b.lanes[t].done) continue;
      b.cur = t;
      hipemu_switch(&g_sched_sp, b.lanes[t].sp);
      // a lane that just finished may complete a rendezvous the others are waiting in: they re-check when resumed
    }
    if (++passes > 200000000L) die("a block made no progress (collective in divergent code, or a missing barrier partner)");
  }
}

void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  if (g_blk) die("nested launch");
  static Block b;   // keeps the fiber stacks between launches
  g_dyn_shared_bytes = shmem;
  b.bdim = block;
  b.gdim = grid;
  b.body = body;
  g_blk = &b;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        b.bid = dim3(x, y, z);
        run_block(b);
      }
  g_blk = nullptr;
}

}  // namespace hipemu
