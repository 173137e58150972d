// tpamd_launch.h -- the seam between the engine's host code (tpamd_capi.hip) and the
// specialised sweep kernels, which are compiled one (joint count, extra rows) instance per
// translation unit (tpamd_sweep_inst.hip with -DTPAMD_INST_D / -DTPAMD_INST_E) so that the
// build runs in parallel: a single instance takes 10-18 s of compiler time, nine of them in one
// translation unit made every build 2.5 minutes.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/tpamd.h"
#include "tpamd_kernels.h"

namespace tpamd {

// (D, E) pairs with a specialised sweep kernel: joint-space paths with 3..8 and 14 joints,
// Cartesian paths (two extra B-only rows) with 6 and 7.
#define TPAMD_SWEEP_INSTANCES(X) X(3, 0) X(4, 0) X(5, 0) X(6, 0) X(7, 0) X(8, 0) X(14, 0) X(6, 2) X(7, 2)

// One workgroup of 128 threads per path on `st`; the dynamic LDS follows from the stride N.
template <int D, int E>
void launch_sweep_joint(int B, hipStream_t st, int N, int max_loops, const JointSource &src,
                        const Workspace &ws, const tpamd_path_outputs *out);
// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KB for the instance (per device).
template <int D, int E>
hipError_t configure_sweep_joint();
template <int D, int E>
hipError_t sweep_joint_attributes(hipFuncAttributes *attr);

#define TPAMD_DECLARE_SWEEP(D, E)                                                                \
  template <>                                                                                    \
  void launch_sweep_joint<D, E>(int B, hipStream_t st, int N, int max_loops, const JointSource &src, \
                                const Workspace &ws, const tpamd_path_outputs *out);             \
  template <>                                                                                    \
  hipError_t configure_sweep_joint<D, E>();                                                      \
  template <>                                                                                    \
  hipError_t sweep_joint_attributes<D, E>(hipFuncAttributes * attr);
TPAMD_SWEEP_INSTANCES(TPAMD_DECLARE_SWEEP)
#undef TPAMD_DECLARE_SWEEP

}  // namespace tpamd
