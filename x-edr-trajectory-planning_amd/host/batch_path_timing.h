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

struct BatchTimingResult {
  int num_samples = 0, num_dofs = 0;
  std::vector<int32_t> status;               // [B] TPAMD_PATH_* (0 = solved)
  std::vector<int32_t> last_extremal_index;  // [B]
  std::vector<double> time, s, sd, sdd;      // [B][N]
  std::vector<double> q, qd, qdd;            // [B][N][D]
};

class BatchPathTiming {
 public:
  // All paths must share num_dofs, num_path_samples and the number of waypoints.
  Status SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths);
  // Times every path starting at path parameter 0 and time `time_start_sec`.
  Status ComputeTimingProfiles(double time_start_sec, BatchTimingResult *result);

 private:
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths_;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_BATCH_PATH_TIMING_H_
