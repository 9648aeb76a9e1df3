// How many workgroups does a gfx950 CU co-schedule as a function of dynamic LDS size and block size?
// Each workgroup spins ~30 us and stamps its start / end wall clock; we count how many started
// before the first one finished.   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe lds_occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) spin(long long *stamps, int spin_ticks) {
  extern __shared__ float smem[];
  const long long t0 = wall_clock64();
  smem[threadIdx.x] = (float)t0;
  __syncthreads();
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = wall_clock64();
  }
}

int main() {
  const int nblocks = 4096;
  long long *d;
  hipMalloc(&d, nblocks * 2 * sizeof(long long));
  std::vector<long long> h(nblocks * 2);
  const int threads_list[] = {256, 512, 576, 1024};
  const int lds_list[] = {16, 32, 40, 48, 53, 56, 64, 65, 72, 80, 96, 128, 160};
  for (int threads : threads_list)
    for (int kb : lds_list) {
      const size_t lds = (size_t)kb * 1024;
      if (hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        printf("threads %4d lds %3d KB: cannot set attribute\n", threads, kb);
        continue;
      }
      int api = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, (const void *)spin, threads, lds);
      hipLaunchKernelGGL(spin, dim3(nblocks), dim3(threads), lds, 0, d, 3000);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); continue; }
      hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      long long first_end = h[1];
      for (int i = 0; i < nblocks; i++) first_end = std::min(first_end, h[2 * i + 1]);
      int started = 0;
      for (int i = 0; i < nblocks; i++) started += h[2 * i] < first_end;
      printf("threads %4d lds %3d KB: %4d concurrent workgroups (%.2f per CU of 256); occupancy API says %d/CU\n", threads, kb,
             started, started / 256.0, api);
    }
  return 0;
}
