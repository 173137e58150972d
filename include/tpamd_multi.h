/*
 * tpamd_multi.h -- one process, several MI355X: a batch of independent paths sharded over the
 * devices of one node, with ONE RCCL gather over xGMI at the end (BASELINE.json north_star,
 * SURVEY.md section 8e). libtpamd_multi.so = this layer on top of libtpamd.so + librccl.so.
 *
 * The reference has no counterpart (it is single-threaded, single-process:
 * trajectory_planning/ holds no thread, process or device boundary); what is sharded is the
 * batched form of PathTimingTrajectory::ComputeTimingProfile (path_timing_trajectory.cc:307-475)
 * that tpamd_time_joint_paths_* performs, and every path's result is the single-device result,
 * bit for bit. bench.py does the same across processes through torch.distributed; this is the
 * C++-side equivalent for host code that stays in one process (the mirror's BatchPathTiming).
 *
 * Shape: paths are cut into contiguous blocks, one per device
 * (tpamd_shard_bounds / tpamd_shard_bounds_balanced in tpamd.h); one host thread per device
 * uploads its block and solves it on that device's engine -- no data-path collective in the
 * solve. Results then either go straight back to the caller's host arrays, each device over its
 * own PCIe link (no collective at all), or stay on the devices and the chosen payload is gathered
 * into buffers on the root device: ncclGather (rccl.h:745) when the blocks are equal, one group of
 * ncclSend / ncclRecv (rccl.h:700, :722) when they are not. Communicators come from
 * ncclCommInitAll (rccl.h:236), one per device, all owned by this process.
 */
#ifndef TPAMD_MULTI_H_
#define TPAMD_MULTI_H_

#include "tpamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tpamd_multi tpamd_multi;

#define TPAMD_E_RCCL (-6) /* an RCCL call failed */

/* What the root receives per path (the three payloads of bench.py --gather):
 *   COMPACT  sd, sdd [N] + ds, time_start: the root rebuilds time from sd with the solver's own
 *            operations (tpamd_rebuild_time_device; bit-identical to the solve's time) and
 *            s = s_start + i ds is an arithmetic sequence          -> 16 N + 16 bytes
 *   PROFILE  time, sd, sdd [N]                                      -> 24 N bytes
 *   FULL     time, sd, sdd [N] and q [N][D] (north_star's t, s', q) -> 24 N + 8 N D bytes
 * qd / qdd always stay on the device that produced them. */
#define TPAMD_GATHER_COMPACT 0
#define TPAMD_GATHER_PROFILE 1
#define TPAMD_GATHER_FULL 2

/* Engines, streams and (for more than one device, or when force_rccl != 0) RCCL communicators for
 * the given devices; device_ordinals NULL = 0 .. num_devices-1. Device 0 of the set is the root. */
int tpamd_multi_create(int num_devices, const int *device_ordinals, int force_rccl, tpamd_multi **out);
void tpamd_multi_destroy(tpamd_multi *multi);
int tpamd_multi_num_devices(const tpamd_multi *multi);
int tpamd_multi_device(const tpamd_multi *multi, int k);     /* ordinal of device k of the set */
tpamd_engine *tpamd_multi_engine(tpamd_multi *multi, int k); /* its engine */
int tpamd_multi_uses_rccl(const tpamd_multi *multi);

/* Bytes one path adds to the gather for a payload (what config.gather.bytes_per_path reports). */
size_t tpamd_gather_bytes_per_path(int payload, int num_samples, int num_dofs);

/* Sharded solve of one joint-space batch given as HOST arrays (tpamd_joint_inputs of B paths).
 *   shard_begin [num_devices + 1]: block k = [shard_begin[k], shard_begin[k+1]) goes to device k;
 *               NULL = tpamd_shard_bounds (equal blocks).
 *   host_out    != NULL: every device copies the outputs of its block into these host arrays
 *               ([B]-shaped, the layout of tpamd_time_joint_paths_host) -- no collective.
 *   root_out    != NULL: DEVICE pointers on the root device, [B]-shaped; the payload's arrays
 *               (COMPACT: sd, sdd, and time rebuilt on the root; PROFILE: time, sd, sdd; FULL:
 *               + q) are gathered there with one ncclGather / one send-recv group. status and
 *               last_extremal_index, if given, are gathered as well (4 bytes per path each).
 *               The call returns when the gather has landed.
 * Either, both or neither may be given. One host thread per device. */
int tpamd_multi_time_joint_paths_host(tpamd_multi *multi, const tpamd_joint_batch *batch,
                                      const tpamd_joint_inputs *in, const int32_t *shard_begin,
                                      const tpamd_path_outputs *host_out, int payload,
                                      const tpamd_path_outputs *root_out);

/* Sharded solve of several joint groups (a mixed-DOF, ragged batch bucketed by the caller, as
 * tpamd_time_joint_groups_host takes it): group_device[g] names the device (index into the set)
 * that solves group g; every device runs its groups side by side on its engine's lanes, one host
 * thread per device, results straight back to the host arrays. No collective. */
int tpamd_multi_time_joint_groups_host(tpamd_multi *multi, int num_groups,
                                       const tpamd_joint_batch *batches,
                                       const tpamd_joint_inputs *inputs,
                                       const tpamd_path_outputs *outputs, const int32_t *group_device);

#ifdef __cplusplus
}
#endif
#endif /* TPAMD_MULTI_H_ */
