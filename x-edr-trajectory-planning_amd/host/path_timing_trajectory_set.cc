#include "path_timing_trajectory_set.h"

#include <algorithm>

namespace trajectory_planning {

using ::tpamd::compat::DeadlineExceededError;
using ::tpamd::compat::FailedPreconditionError;
using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;
using ::tpamd::compat::OutOfRangeError;

PathTimingTrajectorySet::PathTimingTrajectorySet(const PathTimingTrajectoryOptions &options, size_t num_planners,
                                                 size_t num_control_points, double constraint_safety, int device)
    : options_(options), num_planners_(num_planners), num_control_points_(num_control_points),
      constraint_safety_(constraint_safety), summary_(num_planners) {
  lease_ = ::tpamd::acquire_engine(device);
  if (!lease_) { init_status_ = InternalError("no GPU engine"); return; }
  tpamd_planner_set_config cfg{};
  cfg.num_planners = (int32_t)num_planners; cfg.num_dofs = (int32_t)options.GetNumDofs();
  cfg.num_samples = (int32_t)options.GetNumPathSamples(); cfg.num_points = (int32_t)num_control_points;
  cfg.history_capacity = 0; cfg.trajectory_capacity = 0;
  cfg.sampling_method =
      options.GetTimeSamplingMethod() == PathTimingTrajectoryOptions::TimeSamplingMethod::kUniformlyInTime ? 0 : 1;
  cfg.max_planning_iterations = options.GetMaxPlanningIterations();
  cfg.constraint_safety = constraint_safety;
  cfg.max_initial_velocity_error = options.GetMaxInitialVelocityError();
  cfg.time_step_ns = options.GetTimeStep().nanos();
  const int rc = tpamd_planner_set_create(lease_.get(), &cfg, &set_);
  if (rc != 0) init_status_ = InternalError(tpamd_error_string(rc));
}

PathTimingTrajectorySet::~PathTimingTrajectorySet() {
  if (set_) tpamd_planner_set_destroy(set_);     // before the engine goes back to the pool
}

namespace {
int StateCode(TimeablePath::State s) {
  switch (s) {
    case TimeablePath::State::kNewPath: return 1;
    case TimeablePath::State::kModifiedPath: return 2;
    case TimeablePath::State::kPathWasSampled: return 3;
    default: return 0;
  }
}
}  // namespace

Status PathTimingTrajectorySet::SetPath(size_t planner, const TimeableJointSplinePath &path) {
  if (!init_status_.ok()) return init_status_;
  if (planner >= num_planners_) return InvalidArgumentError("no such planner");
  if (path.NumDofs() != options_.GetNumDofs()) return InvalidArgumentError("Path and planner disagree on the number of dofs");
  if (path.NumPathSamples() != options_.GetNumPathSamples())
    return InvalidArgumentError("Path and planner disagree on the number of path samples");
  if ((size_t)path.num_control_points() != num_control_points_)
    return InvalidArgumentError("the set holds splines of one size (control points)");
  if (path.options().constraint_safety() != constraint_safety_) return InvalidArgumentError("constraint safety differs");
  const int state = StateCode(path.GetState());
  if (state != 1 && state != 2) return FailedPreconditionError("SetWaypoints / SwitchToWaypointPath first");
  const int32_t id = (int32_t)planner, st = state;
  const double delta = path.GetPathSamplingDistance();
  const int rc = tpamd_planner_set_upload_paths(set_, 1, &id, path.knots().data(), path.packed_control_points().data(),
                                                path.GetMaxJointVelocity().data(), path.GetMaxJointAcceleration().data(),
                                                &delta, path.GetInitialVelocity().data(), &st);
  if (rc != 0) return InternalError(tpamd_error_string(rc));
  summary_[planner].path_state = state;
  return OkStatus();
}

Status PathTimingTrajectorySet::SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths) {
  if (!init_status_.ok()) return init_status_;
  if (paths.size() > num_planners_) return InvalidArgumentError("more paths than planners");
  const size_t n = paths.size(), D = options_.GetNumDofs(), P = num_control_points_;
  std::vector<double> knots(n * (P + 3)), cps(n * P * D), vmax(n * D), amax(n * D), dl(n), iv(n * D);
  std::vector<int32_t> st(n);
  for (size_t k = 0; k < n; k++) {
    const TimeableJointSplinePath &p = *paths[k];
    if (p.NumDofs() != D || p.NumPathSamples() != options_.GetNumPathSamples() ||
        (size_t)p.num_control_points() != P || p.options().constraint_safety() != constraint_safety_)
      return InvalidArgumentError("path does not have the shape of the set");
    st[k] = StateCode(p.GetState());
    if (st[k] != 1 && st[k] != 2) return FailedPreconditionError("SetWaypoints / SwitchToWaypointPath first");
    std::copy(p.knots().begin(), p.knots().end(), knots.begin() + k * (P + 3));
    std::copy(p.packed_control_points().begin(), p.packed_control_points().end(), cps.begin() + k * P * D);
    for (size_t d = 0; d < D; d++) {
      vmax[k * D + d] = p.GetMaxJointVelocity()[d];
      amax[k * D + d] = p.GetMaxJointAcceleration()[d];
      iv[k * D + d] = p.GetInitialVelocity()[d];
    }
    dl[k] = p.GetPathSamplingDistance();
  }
  const int rc = tpamd_planner_set_upload_paths(set_, (int)n, nullptr, knots.data(), cps.data(), vmax.data(), amax.data(),
                                                dl.data(), iv.data(), st.data());
  if (rc != 0) return InternalError(tpamd_error_string(rc));
  for (size_t k = 0; k < n; k++) summary_[k].path_state = st[k];
  return OkStatus();
}

void PathTimingTrajectorySet::Reset(size_t planner) {
  if (!set_ || planner >= num_planners_) return;
  const int32_t id = (int32_t)planner;
  tpamd_planner_set_reset(set_, 1, &id);
  summary_[planner] = tpamd_planner_summary{};
}

std::vector<Status> PathTimingTrajectorySet::Plan(Time start, Duration time_horizon) {
  return Plan(std::vector<Time>(num_planners_, start), std::vector<Duration>(num_planners_, time_horizon));
}

std::vector<Status> PathTimingTrajectorySet::Plan(const std::vector<Time> &start,
                                                  const std::vector<Duration> &time_horizon) {
  std::vector<Status> result(num_planners_, OkStatus());
  if (!init_status_.ok() || start.size() != num_planners_ || time_horizon.size() != num_planners_) {
    const Status st = init_status_.ok() ? InvalidArgumentError("one start time and horizon per planner") : init_status_;
    std::fill(result.begin(), result.end(), st);
    return result;
  }
  std::vector<int64_t> s(num_planners_), h(num_planners_);
  for (size_t b = 0; b < num_planners_; b++) {
    s[b] = ::tpamd::compat::ToUnixNanos(start[b]);
    h[b] = time_horizon[b].nanos();
  }
  const int rc = tpamd_planner_set_plan(set_, s.data(), h.data(), summary_.data());
  if (rc != 0) {
    std::fill(result.begin(), result.end(), InternalError(tpamd_error_string(rc)));
    return result;
  }
  for (size_t b = 0; b < num_planners_; b++) {
    switch (summary_[b].status) {
      case TPAMD_PLAN_OK: break;
      case TPAMD_PLAN_FAILED_PRECONDITION: result[b] = FailedPreconditionError("No path set / nothing to connect to."); break;
      case TPAMD_PLAN_OUT_OF_RANGE: result[b] = OutOfRangeError("start outside the previous plan"); break;
      case TPAMD_PLAN_INVALID_ARGUMENT:
        result[b] = InvalidArgumentError("start time / duration / initial velocity not acceptable"); break;
      case TPAMD_PLAN_DEADLINE_EXCEEDED: result[b] = DeadlineExceededError("Reached maximum number of planning loops"); break;
      default: result[b] = InternalError("Error optimizing path parameter"); break;
    }
  }
  return result;
}

Status PathTimingTrajectorySet::GetTrajectory(size_t planner, PlannedTrajectory *out) const {
  if (!init_status_.ok()) return init_status_;
  if (planner >= num_planners_ || !out) return InvalidArgumentError("no such planner");
  const size_t n = (size_t)summary_[planner].num_samples, D = options_.GetNumDofs();
  out->time.resize(n); out->path_parameter.resize(n); out->path_parameter_derivative.resize(n);
  out->second_path_parameter_derivative.resize(n);
  out->positions.resize(n * D); out->velocities.resize(n * D); out->accelerations.resize(n * D);
  if (n == 0) return OkStatus();
  const int rc = tpamd_planner_set_download_trajectory(
      set_, (int)planner, 0, (int)n, out->time.data(), out->path_parameter.data(), out->path_parameter_derivative.data(),
      out->second_path_parameter_derivative.data(), out->positions.data(), out->velocities.data(),
      out->accelerations.data());
  return rc == 0 ? OkStatus() : InternalError(tpamd_error_string(rc));
}

size_t PathTimingTrajectorySet::LastPlanBytesOverPcie() const {
  size_t up = 0, down = 0;
  tpamd_planner_set_last_plan_bytes(set_, &up, &down);
  return up + down;
}

size_t PathTimingTrajectorySet::DeviceBytes() const { return tpamd_planner_set_device_bytes(set_); }

}  // namespace trajectory_planning
