// simt.cpp -- see simt.hpp (oracle/cuda_on_cpu, TEST INFRASTRUCTURE)
//
// Cooperative scheduling: the CUDA threads of a block are fibers (ucontext) that run ONE AT A TIME, in thread-index order,
// each until it reaches a synchronisation point (__syncthreads, __syncwarp, a shuffle, a ballot) or returns; then the next
// one runs.  That is a legal SIMT schedule, it is deterministic, and it keeps the warp-synchronous idioms of the sources
// correct (e.g. "every lane reads its right neighbour's slot, then writes its own" without a barrier in between: lanes in
// ascending order do exactly what lock-step lanes do).
#include "simt.hpp"

#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>

namespace cuoc {

uint3 t_threadIdx, t_blockIdx;
dim3 t_blockDim, t_gridDim;
BlockState *t_block = nullptr;
int t_linear_tid = 0;

namespace {

struct Rendezvous {
  unsigned arrived = 0;
  unsigned long generation = 0;
  uint32_t slot[32] = {0}, snapshot[32] = {0};
  unsigned pred_bits = 0, pred_snapshot = 0;
  unsigned participants = 0;  // the lanes that took part in the completed round (a lane may return right after it)
};
struct WarpSync {
  unsigned exited = 0;
  std::map<unsigned, Rendezvous> by_mask;
};
struct Fiber {
  ucontext_t ctx;
  std::unique_ptr<char[]> stack;
  uint3 tid;
  int linear = 0;
  bool done = false;
};
struct Scheduler {
  ucontext_t main_ctx;
  std::vector<Fiber> fibers;
  std::vector<WarpSync> warps;
  int current = -1;
  const std::function<void()> *body = nullptr;
};
Scheduler *g_sched = nullptr;
constexpr size_t kStackBytes = 256 * 1024;

inline WarpSync &my_warp() { return g_sched->warps[t_linear_tid >> 5]; }

void yield() {  // back to the scheduler; it resumes this fiber on its next turn
  Fiber &f = g_sched->fibers[g_sched->current];
  swapcontext(&f.ctx, &g_sched->main_ctx);
}

inline void complete(Rendezvous &r, unsigned need) {
  for (int i = 0; i < 32; i++) r.snapshot[i] = r.slot[i];
  r.pred_snapshot = r.pred_bits & need;
  r.participants = need;
  r.pred_bits = 0;
  r.arrived = 0;
  r.generation++;
}

// every lane of `mask` that is still running meets here; returns once all have arrived
Rendezvous &meet(unsigned mask, uint32_t value, int pred) {
  WarpSync &w = my_warp();
  const int lane = t_linear_tid & 31;
  Rendezvous &r = w.by_mask[mask];
  r.slot[lane] = value;
  if (pred) r.pred_bits |= 1u << lane;
  r.arrived |= 1u << lane;
  const unsigned long gen = r.generation;
  const unsigned need = mask & ~w.exited;
  if ((r.arrived & need) == need) complete(r, need);
  while (r.generation == gen) yield();
  return r;
}

void fiber_entry() {
  Scheduler &s = *g_sched;
  (*s.body)();
  Fiber &f = s.fibers[s.current];
  f.done = true;
  {  // this thread is gone: nobody waits for it any more
    WarpSync &w = s.warps[f.linear >> 5];
    w.exited |= 1u << (f.linear & 31);
    for (auto &kv : w.by_mask) {
      Rendezvous &r = kv.second;
      const unsigned need = kv.first & ~w.exited;
      if (r.arrived != 0 && (r.arrived & need) == need) complete(r, need);
    }
  }
  BlockState *b = t_block;
  b->alive--;
  if (b->alive > 0 && b->waiting == b->alive) {
    b->waiting = 0;
    b->generation++;
  }
  swapcontext(&f.ctx, &s.main_ctx);  // never resumed
}

}  // namespace

void syncthreads() {
  BlockState *b = t_block;
  const unsigned long gen = b->generation;
  if (++b->waiting == b->alive) {
    b->waiting = 0;
    b->generation++;
  }
  while (b->generation == gen) yield();
}

void syncwarp(unsigned mask) { meet(mask, 0u, 0); }

unsigned ballot(unsigned mask, int pred) { return meet(mask, 0u, pred).pred_snapshot; }

uint32_t shfl_bits(unsigned mask, uint32_t v, int src_lane) {
  Rendezvous &r = meet(mask, v, 0);
  const bool valid = src_lane >= 0 && src_lane < 32 && ((r.participants >> src_lane) & 1u);
  return valid ? r.snapshot[src_lane] : v;
}

float atomic_add(float *p, float v) {
  const float old = *p;
  *p = old + v;
  return old;
}
int atomic_add(int *p, int v) {
  const int old = *p;
  *p = old + v;
  return old;
}

void run_block(dim3 grid, dim3 block, uint3 bidx, size_t shared_bytes, const std::function<void()> &body) {
  const int n = (int)(block.x * block.y * block.z);
  BlockState bs;
  bs.alive = n;
  bs.dyn_shared.assign(shared_bytes + 64, 0);
  Scheduler s;
  s.body = &body;
  s.fibers.resize(n);
  s.warps.resize((n + 31) / 32);
  for (size_t w = 0; w < s.warps.size(); w++)  // lanes beyond the block size never run
    for (int l = 0; l < 32; l++)
      if ((int)w * 32 + l >= n) s.warps[w].exited |= 1u << l;
  for (int t = 0; t < n; t++) {
    Fiber &f = s.fibers[t];
    f.linear = t;
    f.tid = uint3{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
    f.stack.reset(new char[kStackBytes]);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.get();
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  Scheduler *outer = g_sched;
  g_sched = &s;
  t_block = &bs;
  t_blockDim = block;
  t_gridDim = grid;
  t_blockIdx = bidx;
  unsigned long rounds = 0;
  while (bs.alive > 0) {
    if (++rounds > 50000000ul) { std::fprintf(stderr, "cuoc: no progress in block (%u, %u, %u): a barrier that not every thread reaches?\n", bidx.x, bidx.y, bidx.z); std::abort(); }
    for (int t = 0; t < n; t++) {
      Fiber &f = s.fibers[t];
      if (f.done) continue;
      s.current = t;
      t_linear_tid = t;
      t_threadIdx = f.tid;
      swapcontext(&s.main_ctx, &f.ctx);
    }
  }
  g_sched = outer;
}

}  // namespace cuoc
