// fused_shapes.hpp -- compile-time shapes of the fused rollout launch (rollout_fused.hip).
//
// The generic kernel reads every dimension (horizon, dof, links, spheres, pairs, workgroup size ...) from its arguments:
// loop trip counts, the divisors of the flat-index decodes and the LDS carve are run-time values, so every phase carries
// integer divisions, loop control and address arithmetic (28 % of the instruction stream was scalar, profiles/r04_d).
// With the dimensions known at compile time the same source unrolls to straight-line code: 64.9 -> 52.9 us per
// 1024-trajectory launch on the C2 workload, bit-identical outputs (profiles/r05_b_*).  The reference does the same with
// NVRTC templates per robot (kinematics_forward_kernel.cuh:126 `N_LINKS`, cuda_core_backend/kernel_cache.py:161-235).
//
// A shape is a FusedShape<...> instantiation; the kernel turns it into __builtin_assume facts on the run-time arguments, and
// the host dispatch (fused_shape_launch) takes a specialised instantiation only when EVERY run-time dimension equals the
// shape's -- anything else runs the generic kernel, so a stale table costs speed, never correctness.
// tests/test_fused_shapes.py holds the table to the packaged robots (dimensions + the lane-list lengths of
// curobo_hip_self_lane_lists_host).  One translation unit per shape (curobo_amd/build.py: -DCUROBO_FUSED_SHAPE_TU=<id>).
//
//           H = padded horizon, NK = knots, D = dof, L = links, S = spheres, P = pairs, C = link-chain length,
//           LEN0 / LEN1 = lane-list entries per lane (passes 0 / 1), NT = threads per workgroup,
//           NCUB / NVOX = cuboid / voxel-grid slots of the scene (-1: not part of the shape)
#pragma once

namespace curobo_hip {

struct FusedShapeDyn { static constexpr bool kStatic = false; };

template <int H_, int NK_, int D_, int L_, int S_, int P_, int C_, int LEN0_, int LEN1_, int NT_, int NCUB_, int NVOX_, int PLAIN_ = 0>
struct FusedShape {
  static constexpr bool kStatic = true;
  static constexpr int kH = H_, kNK = NK_, kD = D_, kL = L_, kS = S_, kP = P_, kC = C_, kLen0 = LEN0_, kLen1 = LEN1_, kNT = NT_,
                       kNCub = NCUB_, kNVox = NVOX_;
  // PLAIN: the launch form of an optimiser iteration is part of the shape too -- self + scene collision with the speed
  // metric, one environment, longest-first dispatch, nothing materialised (no position / sphere outputs, no profile
  // stamps): the branches on those arguments fold away (53.8 -> 51.0 us on the C2 workload).  Any other form of the same
  // dimensions takes the next shape in the list, or the generic kernel.
  static constexpr bool kPlain = PLAIN_ != 0;
};

}  // namespace curobo_hip

//                                      H  NK  D   L   S    P   C  LEN0 LEN1  NT NCUB NVOX PLAIN
// 1: Franka, 12 knots x 2 (BASELINE C2: 256 seeds x 32-step horizon), a scene of four cuboid slots (the C2 world's), plain launch
#define CUROBO_FUSED_SHAPE_1 FusedShape<33, 12, 7, 13, 65, 818, 88, 13, 0, 512, 4, 0, 1>
#define CUROBO_FUSED_SHAPE_1_KERNELS(K) K(3, 3, 1, false)
// 2: Franka, 12 knots x 2, any scene, plain launch; K(.., true) = the plain form of the full trajectory-optimisation cost set
//    (tool pose + c-space STATE on top of the collision terms, no torque limits, no per-term outputs: fused_plain_terms)
#define CUROBO_FUSED_SHAPE_2 FusedShape<33, 12, 7, 13, 65, 818, 88, 13, 0, 512, -1, -1, 1>
#define CUROBO_FUSED_SHAPE_2_KERNELS(K) K(3, 3, 1, false) K(3, 3, 3, false) K(3, 3, 1, true)
// 3: Franka, 12 knots x 2, any scene, any launch form; with the optional trajopt terms (tool pose, c-space STATE)
#define CUROBO_FUSED_SHAPE_3 FusedShape<33, 12, 7, 13, 65, 818, 88, 13, 0, 512, -1, -1, 0>
#define CUROBO_FUSED_SHAPE_3_KERNELS(K) K(3, 3, 1, false) K(3, 3, 1, true) K(3, 3, 3, false)
// 4: Franka, 12 knots x 4 (BASELINE C5: horizon 64, one world per problem), any scene, any launch form
#define CUROBO_FUSED_SHAPE_4 FusedShape<65, 12, 7, 13, 65, 818, 88, 13, 0, 1024, -1, -1, 0>
#define CUROBO_FUSED_SHAPE_4_KERNELS(K) K(3, 3, 1, false) K(3, 3, 3, false)
// 5: UR10e, 12 knots x 2 (BASELINE C3: ESDF world), any scene, plain launch
#define CUROBO_FUSED_SHAPE_5 FusedShape<33, 12, 6, 10, 20, 83, 55, 5, 0, 512, -1, -1, 1>
#define CUROBO_FUSED_SHAPE_5_KERNELS(K) K(3, 3, 2, false) K(3, 3, 1, false)
// 6: UR10e, 12 knots x 2, any scene, any launch form
#define CUROBO_FUSED_SHAPE_6 FusedShape<33, 12, 6, 10, 20, 83, 55, 5, 0, 512, -1, -1, 0>
#define CUROBO_FUSED_SHAPE_6_KERNELS(K) K(3, 3, 2, false) K(3, 3, 1, false)
// 99: a shape compiled at RUN TIME for the robot / horizon at hand (curobo_amd/backends/fused_jit.py: hipcc on this file's
// translation unit with the shape and its kernel list on the command line, the object loaded and registered with
// curobo_hip_rollout_fused_register_shape) -- what the reference does for every kernel with NVRTC
// (cuda_core_backend/kernel_cache.py:161-235)
#ifdef CUROBO_FUSED_JIT_SHAPE
#define CUROBO_FUSED_SHAPE_99 CUROBO_FUSED_JIT_SHAPE
#define CUROBO_FUSED_SHAPE_99_KERNELS(K) CUROBO_FUSED_JIT_KERNELS(K)  // (-D'CUROBO_FUSED_JIT_KERNELS(K)=K(3, 3, 1, false) ...')
#endif
#define CUROBO_FUSED_NUM_SHAPES 6
// (the list the main translation unit walks, most specific first)
#define CUROBO_FUSED_FOR_EACH_SHAPE(X) X(1) X(2) X(3) X(4) X(5) X(6)
