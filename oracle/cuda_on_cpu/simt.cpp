// simt.cpp -- see simt.hpp (oracle/cuda_on_cpu, TEST INFRASTRUCTURE)
#include "simt.hpp"

#include <map>

namespace cuoc {

thread_local uint3 t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;
thread_local BlockState *t_block = nullptr;
thread_local int t_linear_tid = 0;

namespace {

struct Rendezvous {
  unsigned arrived = 0;
  unsigned long generation = 0;
  uint32_t slot[32] = {0}, snapshot[32] = {0};
  unsigned pred_bits = 0, pred_snapshot = 0;
  unsigned participants = 0;  // the lanes that took part in the completed round (a lane may exit right after it)
};
struct WarpSync {
  std::mutex m;
  std::condition_variable cv;
  unsigned exited = 0;
  std::map<unsigned, Rendezvous> by_mask;
};
thread_local WarpSync *t_warp = nullptr;
std::mutex g_atomic;

inline void complete(Rendezvous &r, unsigned need) {
  for (int i = 0; i < 32; i++) r.snapshot[i] = r.slot[i];
  r.pred_snapshot = r.pred_bits & need;
  r.participants = need;
  r.pred_bits = 0;
  r.arrived = 0;
  r.generation++;
}

// every lane of `mask` that is still running meets here; returns once all have arrived
Rendezvous &meet(unsigned mask, uint32_t value, int pred, std::unique_lock<std::mutex> &lk) {
  WarpSync &w = *t_warp;
  const int lane = t_linear_tid & 31;
  Rendezvous &r = w.by_mask[mask];
  r.slot[lane] = value;
  if (pred) r.pred_bits |= 1u << lane;
  r.arrived |= 1u << lane;
  const unsigned long gen = r.generation;
  const unsigned need = mask & ~w.exited;
  if ((r.arrived & need) == need) {
    complete(r, need);
    w.cv.notify_all();
  } else {
    w.cv.wait(lk, [&] { return r.generation != gen; });
  }
  return r;
}

}  // namespace

void syncthreads() {
  BlockState *b = t_block;
  std::unique_lock<std::mutex> lk(b->m);
  const unsigned long gen = b->generation;
  if (++b->waiting == b->alive) {
    b->waiting = 0;
    b->generation++;
    b->cv.notify_all();
  } else {
    b->cv.wait(lk, [&] { return b->generation != gen; });
  }
}

void syncwarp(unsigned mask) {
  std::unique_lock<std::mutex> lk(t_warp->m);
  meet(mask, 0u, 0, lk);
}

unsigned ballot(unsigned mask, int pred) {
  std::unique_lock<std::mutex> lk(t_warp->m);
  return meet(mask, 0u, pred, lk).pred_snapshot;
}

uint32_t shfl_bits(unsigned mask, uint32_t v, int src_lane) {
  std::unique_lock<std::mutex> lk(t_warp->m);
  Rendezvous &r = meet(mask, v, 0, lk);
  const bool valid = src_lane >= 0 && src_lane < 32 && ((r.participants >> src_lane) & 1u);
  return valid ? r.snapshot[src_lane] : v;
}

float atomic_add(float *p, float v) {
  std::lock_guard<std::mutex> g(g_atomic);
  const float old = *p;
  *p = old + v;
  return old;
}
int atomic_add(int *p, int v) {
  std::lock_guard<std::mutex> g(g_atomic);
  const int old = *p;
  *p = old + v;
  return old;
}

void run_block(dim3 grid, dim3 block, uint3 bidx, size_t shared_bytes, const std::function<void()> &body) {
  const int n = (int)(block.x * block.y * block.z);
  BlockState bs;
  bs.alive = n;
  bs.dyn_shared.assign(shared_bytes + 64, 0);
  std::vector<WarpSync> warps((n + 31) / 32);
  for (size_t w = 0; w < warps.size(); w++) {  // lanes beyond the block size never run
    const int first = (int)w * 32;
    for (int l = 0; l < 32; l++)
      if (first + l >= n) warps[w].exited |= 1u << l;
  }
  std::vector<std::thread> threads;
  threads.reserve(n);
  for (int t = 0; t < n; t++) {
    threads.emplace_back([&, t] {
      t_block = &bs;
      t_linear_tid = t;
      t_warp = &warps[t >> 5];
      t_blockDim = block;
      t_gridDim = grid;
      t_blockIdx = bidx;
      t_threadIdx = uint3{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
      body();
      {  // this thread is gone: nobody waits for it any more
        WarpSync &w = *t_warp;
        std::lock_guard<std::mutex> g(w.m);
        w.exited |= 1u << (t & 31);
        for (auto &kv : w.by_mask) {
          Rendezvous &r = kv.second;
          const unsigned need = kv.first & ~w.exited;
          if (r.arrived != 0 && (r.arrived & need) == need) complete(r, need);
        }
        w.cv.notify_all();
      }
      {
        std::lock_guard<std::mutex> g(bs.m);
        bs.alive--;
        if (bs.alive > 0 && bs.waiting == bs.alive) {
          bs.waiting = 0;
          bs.generation++;
          bs.cv.notify_all();
        }
      }
    });
  }
  for (auto &th : threads) th.join();
}

}  // namespace cuoc
