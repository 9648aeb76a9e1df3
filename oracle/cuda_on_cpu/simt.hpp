// simt.hpp -- a CUDA block as a group of cooperatively scheduled fibers (oracle/cuda_on_cpu, TEST INFRASTRUCTURE).
//
// The reference's kernels (/root/reference/curobo/_src/curobolib/kernels) are CUDA C++.  There is no nvcc / NVRTC and no
// NVIDIA GPU here, but the kernels only use a small part of the execution model: threadIdx / blockIdx / blockDim,
// __syncthreads, __syncwarp(mask), __shfl_*_sync, __ballot_sync, atomicAdd, static and dynamic shared memory.  This
// runtime gives them exactly that on the CPU: launch() runs the blocks of a grid one after the other; every CUDA thread
// of a block is a fiber; fibers run one at a time in thread-index order, each up to its next synchronisation point;
// __syncthreads is a barrier over the threads of the block that have not returned yet; warp-level primitives meet on a
// per-warp rendezvous keyed by the mask.  Returned threads never block a barrier (as on the GPU).  Deterministic.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "cuda_runtime.h"

namespace cuoc {

struct BlockState {
  int alive = 0, waiting = 0;
  unsigned long generation = 0;
  std::vector<unsigned char> dyn_shared;
};

extern BlockState *t_block;
extern int t_linear_tid;

// dynamic shared memory of the running block (`extern __shared__ T name[]` in the sources is rewritten to
// `T *name = cuoc::dyn_shared<T>();` by the build recipe: C++ has no array of unknown bound with static storage)
template <typename T>
inline T *dyn_shared() { return reinterpret_cast<T *>(t_block->dyn_shared.data()); }

void run_block(dim3 grid, dim3 block, uint3 bidx, size_t shared_bytes, const std::function<void()> &kernel_body);

template <typename F>
void launch(dim3 grid, dim3 block, size_t shared_bytes, F &&body) {
  const std::function<void()> fn = body;
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) run_block(grid, block, uint3{x, y, z}, shared_bytes, fn);
}

}  // namespace cuoc
