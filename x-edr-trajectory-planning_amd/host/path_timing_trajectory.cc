#include "path_timing_trajectory.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <tuple>

#include "engine_handle.h"
#include "timeable_path_joint_spline.h"

namespace trajectory_planning {

using ::tpamd::compat::DeadlineExceededError;
using ::tpamd::compat::FailedPreconditionError;
using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;
using ::tpamd::compat::OutOfRangeError;
using ::tpamd::compat::Seconds;
using ::tpamd::compat::StatusOr;

namespace {
constexpr int64_t kNsecsPerSec = 1000000000;
// trajectory_planning/time.h:22-29 (truncating conversion through int64 nanoseconds)
Time TimeFromSec(double seconds) { return ::tpamd::compat::FromUnixNanos((int64_t)(seconds * kNsecsPerSec)); }
double TimeToSec(Time t) { return (double)::tpamd::compat::ToUnixNanos(t) / (double)kNsecsPerSec; }
}  // namespace

PathTimingTrajectory::PathTimingTrajectory(const PathTimingTrajectoryOptions &options)
    : options_(options),
      time_step_sec_((double)options.GetTimeStep().nanos() / (double)kNsecsPerSec) {
  Reset();
}

void PathTimingTrajectory::ResetDerived() {
  initial_plan_ = false;
  path_time_start_ = 0.0;
  path_start_ = 0.0;
  path_start_velocity_ = 0.0;
  path_start_acceleration_ = 0.0;
  path_horizon_ = 0.0;
  planned_to_end_ = true;
  final_decel_start_ = TimeFromSec(0.0);
  time_at_path_samples_.clear();
  path_parameter_at_path_samples_.clear();
  path_velocity_at_path_samples_.clear();
  path_acceleration_at_path_samples_.clear();
  position_at_path_samples_.clear();
  velocity_at_path_samples_.clear();
  acceleration_at_path_samples_.clear();
}

void PathTimingTrajectory::ClampToTimeStepMultiple(Time *time) {
  const int64_t loop_multiple = (int64_t)std::round(TimeToSec(*time) / time_step_sec_);
  *time = TimeFromSec(loop_multiple * time_step_sec_);
}

Status PathTimingTrajectory::SetPath(std::shared_ptr<TimeablePath> path) {
  if (path == nullptr) return InvalidArgumentError("Path is nullptr.");
  if (path->NumDofs() != options_.GetNumDofs())
    return InvalidArgumentError("Path and planner disagree on the number of dofs");
  if (path->NumPathSamples() != options_.GetNumPathSamples())
    return InvalidArgumentError("Path and planner disagree on the number of path samples");
  path_ = path;
  return OkStatus();
}

Time PathTimingTrajectory::GetNextPlanStartTime(Time target_time) {
  return std::min(end_time_, std::max(target_time, start_time_));
}

StatusOr<int> PathTimingTrajectory::GetTimeOffsetAfter(Time time) const {
  const double time_sec = TimeToSec(time);
  if (time_.empty()) return FailedPreconditionError("No samples yet.");
  if (time_sec < time_.front()) return OutOfRangeError("time < start_time.");
  const auto it = std::upper_bound(time_.begin(), time_.end(), time_sec);
  if (it == time_.end()) return InternalError("time >= end_time_.");
  return (int)(it - time_.begin());
}

// path_timing_trajectory.cc:477-500
void PathTimingTrajectory::UpdatePathTrackingStatus() {
  target_reached_ = false;
  planned_to_end_ = false;
  if (!initial_plan_) {
    path_horizon_ = 0;
    path_start_ = 0;
    return;
  }
  planned_to_end_ = path_->CloseToEnd(path_horizon_);
  if (planned_to_end_) {
    if (path_->GetState() != TimeablePath::State::kNewPath &&
        path_->GetState() != TimeablePath::State::kModifiedPath) {
      target_reached_ = true;
    } else {
      path_horizon_ = 0.0;
      path_time_start_ = 0.0;
      path_start_ = 0.0;
      path_start_velocity_ = 0.0;
      path_start_acceleration_ = 0.0;
      planned_to_end_ = false;
    }
  }
}

// path_timing_trajectory.cc:502-538
Status PathTimingTrajectory::HandleTimeArguments(Time start) {
  if (initial_plan_ && start > end_time_ + Seconds(time_step_sec_))
    return OutOfRangeError("start > end of previous plan");
  if (!initial_plan_) {
    start_time_ = start;
    end_time_ = start;
    path_start_ = 0.0;
  } else {
    if (start > end_time_) return InvalidArgumentError("Start time must be < end time");
    if (start < start_time_) return InvalidArgumentError("Start time must be >= previous start time");
    start_time_ = start;
  }
  return OkStatus();
}

// One timing window (path_timing_trajectory.cc:307-475), phase 1: where the window starts.
Status PathTimingTrajectory::BeginWindow(Time start, Duration target_duration, Window *w) {
  const double start_sec = TimeToSec(start);
  if (path_ == nullptr) return FailedPreconditionError("No path set");
  if (target_duration <= Seconds(0)) return InvalidArgumentError("Duration must be positive");
  const size_t N = options_.GetNumPathSamples(), D = options_.GetNumDofs();
  w->old_state = path_->GetState();
  w->offset = 0;
  if (w->old_state == TimeablePath::State::kNewPath) {
    path_start_ = 0.0;
    path_start_velocity_ = 0.0;
    path_start_acceleration_ = 0.0;
    path_time_start_ = start_sec;
  } else {
    const int num = (int)time_at_path_samples_.size();
    if (num == 0) return FailedPreconditionError("no previous window to connect to");
    const int lb = (int)(std::lower_bound(time_at_path_samples_.begin(), time_at_path_samples_.end(),
                                          start_sec) - time_at_path_samples_.begin());
    w->offset = std::clamp(lb - 1, 0, num - 1);
    path_start_ = path_parameter_at_path_samples_[w->offset];
    path_start_velocity_ = path_velocity_at_path_samples_[w->offset];
    path_time_start_ = time_at_path_samples_[w->offset];
  }
  w->delta = path_->GetPathSamplingDistance();
  path_horizon_ = path_start_ + w->delta * (path_->GetNumPathSamples() - 1);
  w->q.assign(N * D, 0.0); w->q1.assign(N * D, 0.0); w->q2.assign(N * D, 0.0);
  w->qd.assign(N * D, 0.0); w->qdd.assign(N * D, 0.0);
  w->t.assign(N, 0.0); w->s.assign(N, 0.0); w->sd.assign(N, 0.0); w->sdd.assign(N, 0.0); w->sd2.assign(N, 0.0);
  w->status = -1;
  return OkStatus();
}

// Phase 2 (after the path has been sampled): least-squares projection of the requested
// initial velocity on the start tangent (path_timing_trajectory.cc:360-377).
Status PathTimingTrajectory::ProjectStartVelocity(const Window &w) {
  const size_t D = options_.GetNumDofs();
  if (w.old_state == TimeablePath::State::kModifiedPath || w.old_state == TimeablePath::State::kNewPath) {
    const VectorXd &d0 = path_->GetFirstPathDerivativeAt(0);
    const double nrm2 = d0.squaredNorm();
    if (nrm2 > 100 * std::numeric_limits<double>::epsilon())
      path_start_velocity_ = std::max(path_->GetInitialVelocity().dot(d0) / nrm2, 0.0);
    double max_err = 0.0;
    for (size_t d = 0; d < D; d++)
      max_err = std::max(max_err, std::fabs(d0[d] * path_start_velocity_ - path_->GetInitialVelocity()[d]));
    if (max_err > options_.GetMaxInitialVelocityError())
      return InvalidArgumentError("Could not satisfy initial velocity (probably not parallel to initial tangent)");
  }
  return OkStatus();
}

// Windows of TimeableJointSplinePath planners with equal shapes: ONE sampling call and ONE
// solve for all of them (B = ids.size()).
void PathTimingTrajectory::SolveJointWindows(const std::vector<PathTimingTrajectory *> &planners,
                                             std::vector<Window> *windows, const std::vector<size_t> &ids,
                                             std::vector<Status> *status) {
  tpamd_engine *engine = ::tpamd::shared_engine();
  if (!engine) {
    for (size_t id : ids) (*status)[id] = InternalError("no GPU engine");
    return;
  }
  const size_t B = ids.size();
  auto *first = dynamic_cast<TimeableJointSplinePath *>(planners[ids[0]]->path_.get());
  const size_t D = first->NumDofs(), N = first->NumPathSamples(), P = first->num_control_points();
  std::vector<double> knots(B * (P + 3)), cps(B * P * D), vmax(B * D), amax(B * D), ps(B), dl(B), sd0(B),
      sdd0(B), t0(B);
  for (size_t g = 0; g < B; g++) {
    PathTimingTrajectory *pl = planners[ids[g]];
    auto *joint = dynamic_cast<TimeableJointSplinePath *>(pl->path_.get());
    std::copy(joint->knots().begin(), joint->knots().end(), knots.begin() + g * (P + 3));
    std::copy(joint->packed_control_points().begin(), joint->packed_control_points().end(),
              cps.begin() + g * P * D);
    for (size_t d = 0; d < D; d++) {
      vmax[g * D + d] = joint->GetMaxJointVelocity()[d];
      amax[g * D + d] = joint->GetMaxJointAcceleration()[d];
    }
    ps[g] = pl->path_start_; dl[g] = (*windows)[ids[g]].delta;
  }
  // sampling first: the start velocity is projected on q'(0) before the solve
  std::vector<double> q(B * N * D), q1(B * N * D), q2(B * N * D);
  {
    ::tpamd::EngineGuard guard;
    const int rc = tpamd_sample_joint_paths_host(engine, (int)B, (int)D, (int)N, (int)P, knots.data(), cps.data(),
                                                 ps.data(), dl.data(), q.data(), q1.data(), q2.data());
    if (rc != 0) {
      for (size_t id : ids) (*status)[id] = InternalError(tpamd_error_string(rc));
      return;
    }
  }
  std::vector<char> live(B, 1);
  for (size_t g = 0; g < B; g++) {
    PathTimingTrajectory *pl = planners[ids[g]];
    auto *joint = dynamic_cast<TimeableJointSplinePath *>(pl->path_.get());
    joint->AdoptSamples(pl->path_start_, &q[g * N * D], &q1[g * N * D], &q2[g * N * D]);
    const Status st = pl->ProjectStartVelocity((*windows)[ids[g]]);
    if (!st.ok()) { (*status)[ids[g]] = st; live[g] = 0; }
    sd0[g] = pl->path_start_velocity_; sdd0[g] = pl->path_start_acceleration_; t0[g] = pl->path_time_start_;
  }
  std::vector<double> t(B * N), s(B * N), sd(B * N), sdd(B * N), sd2(B * N), qd(B * N * D), qdd(B * N * D),
      dtmax(B);
  std::vector<int32_t> lei(B, 0), st(B, -1);
  const int max_solver_loops = (int)std::max<size_t>(100, 10 * N);
  tpamd_joint_batch batch{(int)B, (int)D, (int)N, (int)P, max_solver_loops, 0, first->options().constraint_safety()};
  tpamd_joint_inputs in{knots.data(), cps.data(), vmax.data(), amax.data(), ps.data(), dl.data(),
                        sd0.data(), sdd0.data(), t0.data(), nullptr};
  tpamd_path_outputs out{t.data(), s.data(), sd.data(), sdd.data(), q.data(), qd.data(), qdd.data(),
                         lei.data(), dtmax.data(), st.data(), sd2.data()};
  {
    ::tpamd::EngineGuard guard;
    const int rc = tpamd_time_joint_paths_host(engine, &batch, &in, &out);
    if (rc != 0) {
      for (size_t id : ids) (*status)[id] = InternalError(tpamd_error_string(rc));
      return;
    }
  }
  for (size_t g = 0; g < B; g++) {
    if (!live[g]) continue;
    Window &w = (*windows)[ids[g]];
    w.status = st[g];
    if (st[g] >= 2 && st[g] <= 6) { (*status)[ids[g]] = InternalError("Error setting up optimization problem"); continue; }
    if (st[g] != 0) { (*status)[ids[g]] = InternalError("Error optimizing path parameter"); continue; }
    auto cp = [&](std::vector<double> &dst, const std::vector<double> &src, size_t n) {
      std::copy_n(src.begin() + g * n, n, dst.begin());
    };
    cp(w.t, t, N); cp(w.s, s, N); cp(w.sd, sd, N); cp(w.sdd, sdd, N); cp(w.sd2, sd2, N);
    cp(w.q, q, N * D); cp(w.qd, qd, N * D); cp(w.qdd, qdd, N * D);
    w.last_extremal_index = lei[g]; w.max_time_increment = dtmax[g];
    PathTimingTrajectory *pl = planners[ids[g]];
    pl->profile_.AdoptSolution((int)N, (int)(2 * D), pl->path_start_, pl->path_horizon_, w.t.data(), w.s.data(),
                               w.sd.data(), w.sdd.data(), w.sd2.data(), w.last_extremal_index,
                               w.max_time_increment);
  }
}

// Any other TimeablePath: its own SamplePath / ConstraintSetup, rows through the solver
// mirror (which runs on the GPU), epilogue on the host (path_timing_trajectory.cc:458-472).
Status PathTimingTrajectory::SolveWindowOnHost(Window *w) {
  const size_t N = options_.GetNumPathSamples(), D = options_.GetNumDofs();
  Status st = path_->SamplePath(path_start_);
  if (!st.ok()) return st;
  st = path_->ConstraintSetup();
  if (!st.ok()) return st;
  st = ProjectStartVelocity(*w);
  if (!st.ok()) return st;
  const int max_solver_loops = (int)std::max<size_t>(100, 10 * N);
  if (!profile_.InitSolver((int)N, (int)path_->NumConstraints())) return InternalError("Error initializing solver.");
  profile_.SetMaxNumSolverLoops(max_solver_loops);
  if (!profile_.SetupProblem(path_->GetConstraints(), path_start_, path_horizon_, path_start_velocity_,
                             path_start_acceleration_, path_time_start_))
    return InternalError("Error setting up optimization problem");
  if (!profile_.OptimizePathParameter()) return InternalError("Error optimizing path parameter");
  const VectorXd &amax = path_->GetMaxJointAcceleration();
  for (size_t i = 0; i < N; i++) {
    const double v = profile_.GetPathVelocity()[i], a = profile_.GetPathAcceleration()[i];
    for (size_t d = 0; d < D; d++) {
      w->q[i * D + d] = path_->GetPathPositionAt(i)[d];
      const double d1 = path_->GetFirstPathDerivativeAt(i)[d], d2 = path_->GetSecondPathDerivativeAt(i)[d];
      w->qd[i * D + d] = d1 * v;
      w->qdd[i * D + d] = std::min(std::max(d1 * a + d2 * (v * v), -amax[d]), amax[d]);
    }
  }
  w->status = 0;
  return OkStatus();
}

// Phase 3: drop what the new window replaces, then append it (path_timing_trajectory.cc:418-456).
Status PathTimingTrajectory::EndWindow(Window *w) {
  const size_t N = options_.GetNumPathSamples(), D = options_.GetNumDofs();
  auto cut = [&](std::vector<double> &v, size_t stride) { v.resize((size_t)w->offset * stride); };
  cut(time_at_path_samples_, 1); cut(path_parameter_at_path_samples_, 1);
  cut(path_velocity_at_path_samples_, 1); cut(path_acceleration_at_path_samples_, 1);
  cut(position_at_path_samples_, D); cut(velocity_at_path_samples_, D); cut(acceleration_at_path_samples_, D);
  auto app = [](std::vector<double> &v, const double *p, size_t n) { v.insert(v.end(), p, p + n); };
  app(time_at_path_samples_, profile_.GetTimeSamples().data(), N);
  app(path_parameter_at_path_samples_, profile_.GetPathParameter().data(), N);
  app(path_velocity_at_path_samples_, profile_.GetPathVelocity().data(), N);
  app(path_acceleration_at_path_samples_, profile_.GetPathAcceleration().data(), N);
  app(position_at_path_samples_, w->q.data(), N * D);
  app(velocity_at_path_samples_, w->qd.data(), N * D);
  app(acceleration_at_path_samples_, w->qdd.data(), N * D);
  return OkStatus();
}

// path_timing_trajectory.cc:686-695
int PathTimingTrajectory::TimeAtPathSamplesLowerIndex(int starting_index, double time) const {
  const int n = (int)time_at_path_samples_.size();
  for (int index = starting_index; index < n - 1; ++index)
    if (time_at_path_samples_[index + 1] > time) return index;
  return n - 1;
}

// path_timing_trajectory.cc:709-753 for ONE query time (host control flow of Plan; the resampling
// of whole trajectories runs on the GPU with the same a + t (b - a) blend).
PathTimingTrajectory::InterpolationResult PathTimingTrajectory::InterpolateAtTime(double time_sec,
                                                                                  int lower_index) const {
  const size_t D = options_.GetNumDofs();
  InterpolationResult r;
  r.lower_index = TimeAtPathSamplesLowerIndex(lower_index, time_sec);
  const int lower = r.lower_index;
  const int upper = std::min<int>((int)time_at_path_samples_.size() - 1, lower + 1);
  const double *tm = time_at_path_samples_.data();
  const double at = std::abs(tm[upper] - tm[lower]) < std::numeric_limits<double>::epsilon()
                        ? 0.5
                        : (time_sec - tm[lower]) / (tm[upper] - tm[lower]);
  auto lerp = [at](double a, double b) { return a + at * (b - a); };
  const VectorXd &amax = path_->GetMaxJointAcceleration();
  r.position = VectorXd(D); r.velocity = VectorXd(D); r.acceleration = VectorXd(D);
  for (size_t d = 0; d < D; d++) {
    r.position[d] = lerp(position_at_path_samples_[lower * D + d], position_at_path_samples_[upper * D + d]);
    r.velocity[d] = lerp(velocity_at_path_samples_[lower * D + d], velocity_at_path_samples_[upper * D + d]);
    const double a = lerp(acceleration_at_path_samples_[lower * D + d], acceleration_at_path_samples_[upper * D + d]);
    r.acceleration[d] = std::min(std::max(a, -amax[d]), amax[d]);
  }
  r.path_parameter = lerp(path_parameter_at_path_samples_[lower], path_parameter_at_path_samples_[upper]);
  r.path_parameter_derivative = lerp(path_velocity_at_path_samples_[lower], path_velocity_at_path_samples_[upper]);
  r.second_path_parameter_derivative =
      lerp(path_acceleration_at_path_samples_[lower], path_acceleration_at_path_samples_[upper]);
  return r;
}

// path_timing_trajectory.cc:868-880
void PathTimingTrajectory::EraseSamplesUntil(int offset) {
  if (offset <= 0) return;
  auto drop = [&](auto &v) { v.erase(v.begin(), v.begin() + std::min<size_t>((size_t)offset, v.size())); };
  drop(time_); drop(path_parameter_); drop(path_parameter_derivative_);
  drop(second_path_parameter_derivative_); drop(positions_); drop(velocities_); drop(accelerations_);
}

// path_timing_trajectory.cc:540-575
void PathTimingTrajectory::EraseTrajectoryBefore(Time time) {
  const double time_sec = TimeToSec(time);
  if (time_.empty() || time_sec < time_.front()) return;
  switch (options_.GetTimeSamplingMethod()) {
    case PathTimingTrajectoryOptions::TimeSamplingMethod::kSkipSamplesCloserThanTimeStep: {
      // number of samples with a time stamp < time_sec (GetSampleCountUntil, :41-44)
      int smaller = (int)(std::lower_bound(time_.begin(), time_.end(), time_sec) - time_.begin());
      smaller = std::min<int>(smaller, (int)time_.size() - 1);   // the reference reads time_[smaller] unguarded
      const InterpolationResult at_time = InterpolateAtTime(time_sec, std::max(smaller, 0));
      if (time_[smaller] < time_sec + GetMinTimeDeltaToKeep()) EraseSamplesUntil(smaller);
      else EraseSamplesUntil(smaller - 1);
      // the first sample sits exactly at the requested time
      time_.front() = time_sec;
      positions_.front() = at_time.position;
      velocities_.front() = at_time.velocity;
      accelerations_.front() = at_time.acceleration;
      path_parameter_.front() = at_time.path_parameter;
      path_parameter_derivative_.front() = at_time.path_parameter_derivative;
      second_path_parameter_derivative_.front() = at_time.second_path_parameter_derivative;
    } break;
    case PathTimingTrajectoryOptions::TimeSamplingMethod::kUniformlyInTime: {
      const int offset = std::min<int>((int)std::round((time_sec - time_.front()) / time_step_sec_),
                                       (int)time_.size() - 1);
      EraseSamplesUntil(offset);
    } break;
  }
}

// path_timing_trajectory.cc:579-630: everything before the window loop.
Status PathTimingTrajectory::PlanPrologue(Time start, Duration time_horizon, bool *needs_windows) {
  *needs_windows = false;
  if (path_ == nullptr) return FailedPreconditionError("No path set.");
  if (Status st = HandleTimeArguments(start); !st.ok()) return st;
  UpdatePathTrackingStatus();
  const bool planned_enough = (path_->GetState() != TimeablePath::State::kNewPath) &&
                              (path_->GetState() != TimeablePath::State::kModifiedPath) &&
                              (final_decel_start_ >= start + time_horizon);
  if (!time_.empty() && planned_enough) {
    // Already planned far enough: only drop what lies before `start`.
    EraseTrajectoryBefore(start);
    return OkStatus();
  }
  if (initial_plan_) {
    auto offset_or = GetTimeOffsetAfter(start);
    if (!offset_or.ok()) return offset_or.status();
    const int offset = *offset_or;
    auto keep = [&](auto &v) { v.erase(v.begin() + offset, v.end()); };
    keep(time_); keep(path_parameter_); keep(path_parameter_derivative_);
    keep(second_path_parameter_derivative_); keep(positions_); keep(velocities_); keep(accelerations_);
  }
  *needs_windows = true;
  return OkStatus();
}

// path_timing_trajectory.cc:662-684: resample and bookkeeping after the window loop.
Status PathTimingTrajectory::PlanEpilogue(Time start) {
  if (Status st = ResampleTrajectory(TimeToSec(start)); !st.ok()) return st;
  initial_plan_ = true;
  if (!time_.empty()) {
    end_time_ = TimeFromSec(time_.back());
    ClampToTimeStepMultiple(&end_time_);
    final_decel_start_ = TimeFromSec(profile_.GetTimeSamples()[profile_.GetLastExtremalIndex()]);
    ClampToTimeStepMultiple(&final_decel_start_);
  } else {
    end_time_ = start_time_;
    final_decel_start_ = end_time_;
  }
  target_reached_ = planned_to_end_;
  return OkStatus();
}

// path_timing_trajectory.cc:579-684
Status PathTimingTrajectory::Plan(Time start, Duration time_horizon) {
  return PlanBatch({this}, start, time_horizon)[0];
}

std::vector<Status> PathTimingTrajectory::PlanBatch(const std::vector<PathTimingTrajectory *> &planners,
                                                    Time start, Duration time_horizon) {
  const size_t P = planners.size();
  std::vector<Status> status(P, OkStatus());
  std::vector<char> looping(P, 0), finish(P, 0);
  std::vector<Time> loop_start(P, start);
  std::vector<int> loop(P, 0);
  std::vector<Window> windows(P);
  for (size_t i = 0; i < P; i++) {
    if (planners[i] == nullptr) { status[i] = InvalidArgumentError("null planner"); continue; }
    bool needs = false;
    status[i] = planners[i]->PlanPrologue(start, time_horizon, &needs);
    // the loop condition of path_timing_trajectory.cc:632 is tested before the first window
    if (status[i].ok() && needs) { looping[i] = !planners[i]->planned_to_end_; finish[i] = 1; }
  }
  for (;;) {
    // phase 1 on the host, then group the joint-spline windows by shape
    std::map<std::tuple<size_t, size_t, size_t, double>, std::vector<size_t>> groups;
    std::vector<size_t> foreign;
    bool any = false;
    for (size_t i = 0; i < P; i++) {
      if (!looping[i]) continue;
      any = true;
      PathTimingTrajectory *pl = planners[i];
      status[i] = pl->BeginWindow(loop_start[i], start + time_horizon - loop_start[i], &windows[i]);
      if (!status[i].ok()) { looping[i] = 0; finish[i] = 0; continue; }
      if (auto *joint = dynamic_cast<TimeableJointSplinePath *>(pl->path_.get()))
        groups[std::make_tuple(joint->NumDofs(), joint->NumPathSamples(), (size_t)joint->num_control_points(),
                               joint->options().constraint_safety())].push_back(i);
      else
        foreign.push_back(i);
    }
    if (!any) break;
    for (const auto &kv : groups) SolveJointWindows(planners, &windows, kv.second, &status);
    for (size_t i : foreign) status[i] = planners[i]->SolveWindowOnHost(&windows[i]);
    // phase 3 and the loop bookkeeping of path_timing_trajectory.cc:640-660
    for (size_t i = 0; i < P; i++) {
      if (!looping[i]) continue;
      PathTimingTrajectory *pl = planners[i];
      if (status[i].ok()) status[i] = pl->EndWindow(&windows[i]);
      if (!status[i].ok()) { looping[i] = 0; finish[i] = 0; continue; }
      const int N = (int)pl->options_.GetNumPathSamples();
      const int decel_start = std::max(pl->profile_.GetLastExtremalIndex(), N / 2);
      pl->final_decel_start_ = TimeFromSec(pl->profile_.GetTimeSamples()[decel_start]);
      pl->planned_to_end_ = pl->path_->CloseToEnd(pl->path_horizon_);
      const bool time_horizon_reached =
          (pl->profile_.GetTimeSamples()[N - 1] - TimeToSec(start)) > time_horizon / Seconds(1);
      if (loop[i] >= pl->options_.GetMaxPlanningIterations()) {
        status[i] = DeadlineExceededError("Reached maximum number of planning loops");
        looping[i] = 0; finish[i] = 0;
        continue;
      }
      loop_start[i] = pl->final_decel_start_;
      loop[i]++;
      if (pl->planned_to_end_ || time_horizon_reached) looping[i] = 0;
    }
  }
  for (size_t i = 0; i < P; i++)
    if (finish[i] && status[i].ok()) status[i] = planners[i]->PlanEpilogue(start);
  return status;
}

Status PathTimingTrajectory::ResampleTrajectory(double start_sec) {
  switch (options_.GetTimeSamplingMethod()) {
    case PathTimingTrajectoryOptions::TimeSamplingMethod::kUniformlyInTime:
      return ResampleEquidistantlyInTime(start_sec);
    case PathTimingTrajectoryOptions::TimeSamplingMethod::kSkipSamplesCloserThanTimeStep:
      ResampleSkippingSamplesCloserThanTimeStep(start_sec);
      return OkStatus();
  }
  return OkStatus();
}

// path_timing_trajectory.cc:755-783 on the GPU (tpamd_resample_uniform_host).
Status PathTimingTrajectory::ResampleEquidistantlyInTime(double start_sec) {
  const size_t D = options_.GetNumDofs();
  const int S = (int)time_at_path_samples_.size();
  if (S < 2) return InternalError("nothing to resample");
  const double duration = time_at_path_samples_.back() - start_sec;
  const int M = (int)(std::ceil(duration / time_step_sec_) + 1);
  if (M < 1) return InternalError("negative trajectory duration");
  tpamd_engine *engine = ::tpamd::shared_engine();
  if (!engine) return InternalError("no GPU engine");
  std::vector<double> ot(M), os(M), osd(M), osdd(M), oq((size_t)M * D), oqd((size_t)M * D), oqdd((size_t)M * D);
  int32_t count = 0;
  tpamd_resample_args a{};
  a.num_paths = 1; a.num_samples = S; a.num_dofs = (int)D; a.max_out = M;
  a.time = time_at_path_samples_.data(); a.s = path_parameter_at_path_samples_.data();
  a.sd = path_velocity_at_path_samples_.data(); a.sdd = path_acceleration_at_path_samples_.data();
  a.q = position_at_path_samples_.data(); a.qd = velocity_at_path_samples_.data();
  a.qdd = acceleration_at_path_samples_.data();
  a.max_acceleration = path_->GetMaxJointAcceleration().data();
  a.start_sec = &start_sec; a.time_step = time_step_sec_; a.status = nullptr;
  a.out_time = ot.data(); a.out_s = os.data(); a.out_sd = osd.data(); a.out_sdd = osdd.data();
  a.out_q = oq.data(); a.out_qd = oqd.data(); a.out_qdd = oqdd.data(); a.count = &count;
  {
    ::tpamd::EngineGuard guard;
    const int rc = tpamd_resample_uniform_host(engine, &a);
    if (rc != 0) return InternalError(tpamd_error_string(rc));
  }
  time_.assign(ot.begin(), ot.end());
  path_parameter_.assign(os.begin(), os.end());
  path_parameter_derivative_.assign(osd.begin(), osd.end());
  second_path_parameter_derivative_.assign(osdd.begin(), osdd.end());
  positions_.assign(M, VectorXd(D)); velocities_.assign(M, VectorXd(D)); accelerations_.assign(M, VectorXd(D));
  for (int i = 0; i < M; i++)
    for (size_t d = 0; d < D; d++) {
      positions_[i][d] = oq[(size_t)i * D + d];
      velocities_[i][d] = oqd[(size_t)i * D + d];
      accelerations_[i][d] = oqdd[(size_t)i * D + d];
    }
  return OkStatus();
}

// path_timing_trajectory.cc:785-836 on the GPU (tpamd_resample_skip_host).
void PathTimingTrajectory::ResampleSkippingSamplesCloserThanTimeStep(double start_sec) {
  const size_t D = options_.GetNumDofs();
  const int S = (int)time_at_path_samples_.size();
  time_.clear(); positions_.clear(); velocities_.clear(); accelerations_.clear();
  path_parameter_.clear(); path_parameter_derivative_.clear(); second_path_parameter_derivative_.clear();
  tpamd_engine *engine = ::tpamd::shared_engine();
  if (!engine || S < 2) return;
  const int cap = S + 1;
  std::vector<double> ot(cap), os(cap), osd(cap), osdd(cap), oq((size_t)cap * D), oqd((size_t)cap * D),
      oqdd((size_t)cap * D);
  int32_t count = 0;
  tpamd_resample_args a{};
  a.num_paths = 1; a.num_samples = S; a.num_dofs = (int)D; a.max_out = cap;
  a.time = time_at_path_samples_.data(); a.s = path_parameter_at_path_samples_.data();
  a.sd = path_velocity_at_path_samples_.data(); a.sdd = path_acceleration_at_path_samples_.data();
  a.q = position_at_path_samples_.data(); a.qd = velocity_at_path_samples_.data();
  a.qdd = acceleration_at_path_samples_.data();
  a.max_acceleration = path_->GetMaxJointAcceleration().data();
  a.start_sec = &start_sec; a.time_step = time_step_sec_; a.status = nullptr;
  a.out_time = ot.data(); a.out_s = os.data(); a.out_sd = osd.data(); a.out_sdd = osdd.data();
  a.out_q = oq.data(); a.out_qd = oqd.data(); a.out_qdd = oqdd.data(); a.count = &count;
  {
    ::tpamd::EngineGuard guard;
    if (tpamd_resample_skip_host(engine, &a) != 0) return;
  }
  const int M = std::min<int>(count, cap);
  time_.assign(ot.begin(), ot.begin() + M);
  path_parameter_.assign(os.begin(), os.begin() + M);
  path_parameter_derivative_.assign(osd.begin(), osd.begin() + M);
  second_path_parameter_derivative_.assign(osdd.begin(), osdd.begin() + M);
  positions_.assign(M, VectorXd(D)); velocities_.assign(M, VectorXd(D)); accelerations_.assign(M, VectorXd(D));
  for (int i = 0; i < M; i++)
    for (size_t d = 0; d < D; d++) {
      positions_[i][d] = oq[(size_t)i * D + d];
      velocities_[i][d] = oqd[(size_t)i * D + d];
      accelerations_[i][d] = oqdd[(size_t)i * D + d];
    }
}

}  // namespace trajectory_planning
