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

void yield() { swapcontext(&self().ctx, &g_blk->sched); }

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

static void trampoline() {
  Block &b = *g_blk;
  b.body();
  Lane &me = self();
  me.done = true;
  --b.alive;
  --b.w_alive[flat_tid() >> 6];
  swapcontext(&me.ctx, &b.sched);
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
    l.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
    getcontext(&l.ctx);
    l.ctx.uc_stack.ss_sp = l.stack.data();
    l.ctx.uc_stack.ss_size = l.stack.size();
    l.ctx.uc_link = nullptr;
    makecontext(&l.ctx, (void (*)())trampoline, 0);
  }
  // round-robin until every lane has returned; a full pass in which nobody makes progress cannot be detected cheaply, so a
  // generous pass limit guards against a rendezvous that can never complete (e.g. a collective inside divergent control flow)
  long passes = 0;
  while (b.alive > 0) {
    for (int t = 0; t < n; ++t) {
      if (b.lanes[t].done) continue;
      b.cur = t;
      swapcontext(&b.sched, &b.lanes[t].ctx);
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
