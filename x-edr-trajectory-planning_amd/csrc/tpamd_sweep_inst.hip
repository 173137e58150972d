// tpamd_sweep_inst.hip -- one instance of the specialised sweep kernel k_sweep_joint<D, E>
// (tpamd_sweep_joint.h) per translation unit: compile with -DTPAMD_INST_D=<joints>
// -DTPAMD_INST_E=<extra rows>. See tpamd_launch.h.
#include "tpamd_launch.h"
#include "tpamd_sweep_joint.h"

#if !defined(TPAMD_INST_D) || !defined(TPAMD_INST_E)
#error "compile with -DTPAMD_INST_D=<joints> -DTPAMD_INST_E=<0|2>"
#endif

namespace tpamd {

template <>
void launch_sweep_joint<TPAMD_INST_D, TPAMD_INST_E>(int B, hipStream_t st, int N, int max_loops,
                                                    const JointSource &src, const Workspace &ws,
                                                    const tpamd_path_outputs *out) {
  hipLaunchKernelGGL((k_sweep_joint<TPAMD_INST_D, TPAMD_INST_E>), dim3(B), dim3(128),
                     (sweep_joint_lds_bytes<TPAMD_INST_D, TPAMD_INST_E>(N)), st, N, max_loops, src, ws,
                     out->time, out->s, out->sd, out->sdd, out->last_extremal_index,
                     out->max_time_increment, out->status, out->qd, out->qdd);
}

template <>
hipError_t configure_sweep_joint<TPAMD_INST_D, TPAMD_INST_E>() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_joint<TPAMD_INST_D, TPAMD_INST_E>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <>
hipError_t sweep_joint_attributes<TPAMD_INST_D, TPAMD_INST_E>(hipFuncAttributes *attr) {
  return hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&k_sweep_joint<TPAMD_INST_D, TPAMD_INST_E>));
}

}  // namespace tpamd
