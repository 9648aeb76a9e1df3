// runtime.cpp -- error plumbing shared by every C-ABI entry point.
#include "common.hpp"

#include <atomic>

namespace curobo_hip {

static thread_local char g_err[512] = "";
static std::atomic<int> g_debug_sync{0};

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Launch errors are sticky per thread in HIP; hipGetLastError() also clears them.  With debug
// sync on, the stream is drained so asynchronous faults surface at the offending launch
// (reference runtime.debug: cuda_core_backend/launch_helper.py:13-19).  Never used while a
// stream is capturing (synchronising a capturing stream is an error), so capture stays legal.
int check_launch(const char *what, hipStream_t stream) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && g_debug_sync.load(std::memory_order_relaxed)) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)
      e = hipStreamSynchronize(stream);
  }
  if (e != hipSuccess) return set_error(CUROBO_HIP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return CUROBO_HIP_OK;
}

}  // namespace curobo_hip

CUROBO_EXPORT const char *curobo_hip_last_error(void) { return curobo_hip::g_err; }
CUROBO_EXPORT int curobo_hip_abi_version(void) { return 7; }  // 7: curobo_hip_mesh.sign_rule + cell lists, curobo_hip_mesh_set.flags, curobo_hip_mesh_cells_*; 6: curobo_hip_rollout_fused_shape_id / _set_shapes_enabled (compile-time shapes of the fused launch); 3: round-3 additions (mesh BVH, seed-IK LDS query); 4: RNEA launches with a transposition scratch; 5: curobo_hip_mesh_set.num_envs, the queued mesh launch
CUROBO_EXPORT void curobo_hip_set_debug_sync(int enabled) { curobo_hip::g_debug_sync.store(enabled ? 1 : 0); }
