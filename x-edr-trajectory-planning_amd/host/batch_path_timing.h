// BatchPathTiming -- the addition to the reference API: many independent
// TimeableJointSplinePath objects timed by ONE engine call (B >> 1). Each path gets the
// result PathTimingTrajectory::ComputeTimingProfile would have produced for it
// (path_timing_trajectory.cc:307-475, new-path case: s from 0, zero start velocity
// unless the path carries an initial velocity).
#ifndef TPAMD_HOST_BATCH_PATH_TIMING_H_
#define TPAMD_HOST_BATCH_PATH_TIMING_H_

#include <memory>
#include <vector>

#include "timeable_path_joint_spline.h"

namespace trajectory_planning {

// Results are packed path after path without padding: path b owns
// time/s/sd/sdd[sample_offset[b] .. sample_offset[b+1]) and
// q/qd/qdd[joint_offset[b] .. joint_offset[b+1]) (row-major [n_b][D_b]). For a uniform
// batch that is the dense [B][N] / [B][N][D] layout.
struct BatchTimingResult {
  int num_samples = 0, num_dofs = 0;         // maxima over the batch (= the common values if uniform)
  std::vector<int32_t> status;               // [B] TPAMD_PATH_* (0 = solved)
  std::vector<int32_t> last_extremal_index;  // [B]
  std::vector<int32_t> samples_per_path, dofs_per_path;  // [B]
  std::vector<size_t> sample_offset, joint_offset;        // [B+1]
  std::vector<double> time, s, sd, sdd;
  std::vector<double> q, qd, qdd;
};

class BatchPathTiming {
 public:
  // Paths may differ in num_dofs, num_path_samples and waypoint count (BASELINE.json configs[4]):
  // they are grouped by (dofs, control points, constraint safety[, ceil(samples / bucket)]) with
  // per-path sample counts, and ALL groups of a device are solved by one engine call, side by side
  // (tpamd_time_joint_groups_host; within a group the sweep takes the paths longest first).
  Status SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths);
  // HIP devices that share the batch: the paths are cut into contiguous blocks of
  // roughly equal cost (samples x rows^2, tpamd_shard_bounds_balanced), one block, one engine from
  // the pool and one host thread per device; results go straight into the caller's result, nothing
  // travels between devices. Default: the default device (TPAMD_DEVICE, else 0) alone.
  Status SetDevices(const std::vector<int> &devices);
  // Sample-count bucket of ragged batches: 0 (default) = one group per (dofs, control points,
  // safety), its stride the largest sample count; w > 0 = also by ceil(samples / w), as
  // SURVEY.md 8d sketches with w = 512. Measured on MI355X (DESIGN.md, configs[4]): the coarser
  // the faster -- every group costs four launches and the runtime has four hardware queues.
  void SetSampleBucket(int samples) { sample_bucket_ = samples > 0 ? samples : 0; }
  // Times every path starting at path parameter 0 and time `time_start_sec`.
  Status ComputeTimingProfiles(double time_start_sec, BatchTimingResult *result);

 private:
  Status ComputeBlock(int device, size_t lo, size_t hi, double time_start_sec, BatchTimingResult *r) const;
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths_;
  std::vector<int> devices_;
  int sample_bucket_ = 0;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_BATCH_PATH_TIMING_H_
