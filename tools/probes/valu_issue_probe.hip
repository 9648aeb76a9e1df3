// VALU issue rate probe (gfx950): wave64 v_fma_f32 / v_pk_fma_f32 / mixed VALU+SALU streams at 1..8 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_issue_probe.hip -o /tmp/valu_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) probe(float *out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
  int s = iters;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 independent scalar fmas per iteration
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 1) {  // 4 independent packed fmas (the same 8 fmas of work) per iteration
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
    } else if (MODE == 2) {  // 8 DEPENDENT scalar fmas (one chain)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   : "+v"(x0) : "v"(a), "v"(b));
    } else if (MODE == 3) {  // 8 scalar fmas interleaved with 8 SALU adds
      asm volatile("v_fma_f32 %0, %0, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %9, %10\n s_add_u32 %8, %8, 1\n"
                   "v_fma_f32 %2, %2, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %9, %10\n s_add_u32 %8, %8, 1\n"
                   "v_fma_f32 %4, %4, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %9, %10\n s_add_u32 %8, %8, 1\n"
                   "v_fma_f32 %6, %6, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %9, %10\n s_add_u32 %8, %8, 1\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(s) : "v"(a), "v"(b) : "scc");
    } else if (MODE == 4) {  // 8 independent v_add_f32 (two-operand VOP2)
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + s;
}

template <int MODE>
void run(const char *name, int per_iter_valu, float *out) {
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * 4 * wps;  // one wavefront per workgroup; wps wavefronts per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cycles = ms * 1e-3 * 2.4e9;  // per SIMD (all run in parallel), at the nominal clock
    const double valu = (double)iters * per_iter_valu * wps;
    printf("%-34s %d wavefronts/SIMD: %.3f ms, %.2f SIMD-cycles per VALU wave-instruction\n", name, wps, ms, cycles / valu);
    fflush(stdout);
  }
}

int main() {
  float *out; hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  run<0>("8 independent v_fma_f32", 8, out);
  run<4>("8 independent v_add_f32", 8, out);
  run<1>("4 independent v_pk_fma_f32", 4, out);
  run<2>("8 dependent v_fma_f32", 8, out);
  run<3>("8 v_fma_f32 + 8 s_add_u32", 8, out);
  return 0;
}
