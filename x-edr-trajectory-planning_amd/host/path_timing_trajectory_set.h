// PathTimingTrajectorySet -- B PathTimingTrajectory planners (path_timing_trajectory.h:91-186) with
// TimeableJointSplinePath paths of one shape whose state stays ON THE DEVICE between Plan calls
// (include/tpamd.h tpamd_planner_set_*). Where PathTimingTrajectory::PlanBatch ships every
// planner's window history up and down on each call (about 200 MB each way for 1024 planners),
// a Plan call here moves 24 bytes per planner up and one 56-byte record down; the histories, the
// window loop, the resampling in time and the erase / append bookkeeping of
// path_timing_trajectory.cc:540-577, :660-684 run on the device. Every planner ends in exactly the
// state Plan(start, time_horizon) would have left a PathTimingTrajectory in.
//
// The waypoint fit and online path edits stay on the host (TimeableJointSplinePath::SetWaypoints,
// SwitchToWaypointPath: O(waypoints)); SetPath uploads the resulting spline and its state.
// Trajectories come down only when asked for (GetTrajectory).
#ifndef TPAMD_HOST_PATH_TIMING_TRAJECTORY_SET_H_
#define TPAMD_HOST_PATH_TIMING_TRAJECTORY_SET_H_

#include <memory>
#include <vector>

#include "engine_handle.h"
#include "path_timing_trajectory.h"
#include "timeable_path_joint_spline.h"

namespace trajectory_planning {

// What TrajectoryPlanner's getters return for one planner (trajectory_planner.h:81-110).
struct PlannedTrajectory {
  std::vector<double> time, path_parameter, path_parameter_derivative, second_path_parameter_derivative;
  std::vector<double> positions, velocities, accelerations;   // [samples][dofs], packed
};

class PathTimingTrajectorySet {
 public:
  // All planners share the planner options, the path options (dofs, samples, delta may differ per
  // path: it is taken from each path) and the number of control points of their splines.
  PathTimingTrajectorySet(const PathTimingTrajectoryOptions &options, size_t num_planners,
                          size_t num_control_points, double constraint_safety = 0.8, int device = -1);
  ~PathTimingTrajectorySet();
  PathTimingTrajectorySet(const PathTimingTrajectorySet &) = delete;
  PathTimingTrajectorySet &operator=(const PathTimingTrajectorySet &) = delete;

  Status status() const { return init_status_; }      // construction outcome (no GPU: not ok)
  size_t size() const { return num_planners_; }
  // SetPath for one planner / for planners 0..paths.size()-1: the path must be kNewPath (after
  // SetWaypoints) or kModifiedPath (after SwitchToWaypointPath); its spline, limits, sampling
  // distance and initial velocity go to the device.
  Status SetPath(size_t planner, const TimeableJointSplinePath &path);
  Status SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths);
  void Reset(size_t planner);
  // Plan(start, time_horizon) for every planner; one status per planner.
  std::vector<Status> Plan(Time start, Duration time_horizon);
  std::vector<Status> Plan(const std::vector<Time> &start, const std::vector<Duration> &time_horizon);

  // State after the last Plan, from the summary record (no trajectory download).
  size_t GetNumTimeSamples(size_t planner) const { return (size_t)summary_[planner].num_samples; }
  Time GetStartTime(size_t planner) const { return ::tpamd::compat::FromUnixNanos(summary_[planner].start_time_ns); }
  Time GetEndTime(size_t planner) const { return ::tpamd::compat::FromUnixNanos(summary_[planner].end_time_ns); }
  Time GetFinalDecelStart(size_t planner) const {
    return ::tpamd::compat::FromUnixNanos(summary_[planner].final_decel_start_ns);
  }
  Time GetNextPlanStartTime(size_t planner, Time target_time) const {
    return std::min(GetEndTime(planner), std::max(target_time, GetStartTime(planner)));
  }
  bool IsTrajectoryAtEnd(size_t planner) const {      // trajectory_planner.h:103-110
    const int st = summary_[planner].path_state;
    return st != 1 && st != 2 && summary_[planner].target_reached != 0;
  }
  int WindowsOfLastPlan(size_t planner) const { return summary_[planner].windows; }
  // The planner's trajectory (GetTime, GetPositions, ...): one download of its samples.
  Status GetTrajectory(size_t planner, PlannedTrajectory *out) const;
  // Bytes the last Plan call moved over PCIe, both directions.
  size_t LastPlanBytesOverPcie() const;
  size_t DeviceBytes() const;

 private:
  const PathTimingTrajectoryOptions options_;
  const size_t num_planners_, num_control_points_;
  const double constraint_safety_;
  Status init_status_;
  ::tpamd::EngineLease lease_;
  tpamd_planner_set *set_ = nullptr;
  std::vector<tpamd_planner_summary> summary_;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_PATH_TIMING_TRAJECTORY_SET_H_
