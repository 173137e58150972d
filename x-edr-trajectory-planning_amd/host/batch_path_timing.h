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
  // they are grouped by (dofs, control points, constraint safety), one engine call per group with
  // per-path sample counts.
  Status SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths);
  // Times every path starting at path parameter 0 and time `time_start_sec`.
  Status ComputeTimingProfiles(double time_start_sec, BatchTimingResult *result);

 private:
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths_;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_BATCH_PATH_TIMING_H_
