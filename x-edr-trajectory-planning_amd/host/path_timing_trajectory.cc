#include "path_timing_trajectory.h"

#include <algorithm>
#include <cmath>
#include <limits>

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

// One timing window (path_timing_trajectory.cc:307-475).
Status PathTimingTrajectory::ComputeTimingProfile(Time start, Duration target_duration) {
  const double start_sec = TimeToSec(start);
  if (path_ == nullptr) return FailedPreconditionError("No path set");
  if (target_duration <= Seconds(0)) return InvalidArgumentError("Duration must be positive");
  const size_t N = options_.GetNumPathSamples(), D = options_.GetNumDofs();
  const TimeablePath::State old_path_state = path_->GetState();
  int path_samples_offset = 0;
  if (old_path_state == TimeablePath::State::kNewPath) {
    path_start_ = 0.0;
    path_start_velocity_ = 0.0;
    path_start_acceleration_ = 0.0;
    path_time_start_ = start_sec;
  } else {
    const int num = (int)time_at_path_samples_.size();
    if (num == 0) return FailedPreconditionError("no previous window to connect to");
    const int lb = (int)(std::lower_bound(time_at_path_samples_.begin(), time_at_path_samples_.end(),
                                          start_sec) - time_at_path_samples_.begin());
    path_samples_offset = std::clamp(lb - 1, 0, num - 1);
    path_start_ = path_parameter_at_path_samples_[path_samples_offset];
    path_start_velocity_ = path_velocity_at_path_samples_[path_samples_offset];
    path_time_start_ = time_at_path_samples_[path_samples_offset];
  }
  const double delta = path_->GetPathSamplingDistance();
  path_horizon_ = path_start_ + delta * (path_->GetNumPathSamples() - 1);

  auto *joint = dynamic_cast<TimeableJointSplinePath *>(path_.get());
  std::vector<double> q(N * D), q1(N * D), q2(N * D);
  if (joint != nullptr) {
    // sampling first: the start velocity is projected on q'(0) before the solve
    tpamd_engine *engine = ::tpamd::shared_engine();
    if (!engine) return InternalError("no GPU engine");
    ::tpamd::EngineGuard guard;
    const int rc = tpamd_sample_joint_paths_host(engine, 1, (int)D, (int)N, joint->num_control_points(),
                                                 joint->knots().data(),
                                                 joint->packed_control_points().data(), &path_start_,
                                                 &delta, q.data(), q1.data(), q2.data());
    if (rc != 0) return InternalError(tpamd_error_string(rc));
    joint->AdoptSamples(path_start_, q.data(), q1.data(), q2.data());
  } else {
    Status st = path_->SamplePath(path_start_);
    if (!st.ok()) return st;
    st = path_->ConstraintSetup();
    if (!st.ok()) return st;
  }

  if (old_path_state == TimeablePath::State::kModifiedPath ||
      old_path_state == TimeablePath::State::kNewPath) {
    // least-squares projection of the requested initial velocity on the start tangent
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

  const int max_solver_loops = (int)std::max<size_t>(100, 10 * N);
  std::vector<double> qd(N * D), qdd(N * D);
  if (joint != nullptr) {
    tpamd_engine *engine = ::tpamd::shared_engine();
    ::tpamd::EngineGuard guard;
    tpamd_joint_batch batch{1, (int)D, (int)N, joint->num_control_points(), max_solver_loops, 0,
                            joint->options().constraint_safety()};
    tpamd_joint_inputs in{joint->knots().data(), joint->packed_control_points().data(),
                          joint->GetMaxJointVelocity().data(), joint->GetMaxJointAcceleration().data(),
                          &path_start_, &delta, &path_start_velocity_, &path_start_acceleration_,
                          &path_time_start_, nullptr};
    std::vector<double> t(N), s(N), sd(N), sdd(N), sd2(N);
    int32_t lei = 0, status = -1;
    double dtmax = 0;
    tpamd_path_outputs out{t.data(), s.data(), sd.data(), sdd.data(), q.data(), qd.data(), qdd.data(),
                           &lei, &dtmax, &status, sd2.data()};
    const int rc = tpamd_time_joint_paths_host(engine, &batch, &in, &out);
    if (rc != 0) return InternalError(tpamd_error_string(rc));
    if (status >= 2 && status <= 6) return InternalError("Error setting up optimization problem");
    if (status != 0) return InternalError("Error optimizing path parameter");
    profile_.AdoptSolution((int)N, (int)(2 * D), path_start_, path_horizon_, t.data(), s.data(),
                           sd.data(), sdd.data(), sd2.data(), lei, dtmax);
  } else {
    if (!profile_.InitSolver((int)N, (int)path_->NumConstraints()))
      return InternalError("Error initializing solver.");
    profile_.SetMaxNumSolverLoops(max_solver_loops);
    if (!profile_.SetupProblem(path_->GetConstraints(), path_start_, path_horizon_, path_start_velocity_,
                               path_start_acceleration_, path_time_start_))
      return InternalError("Error setting up optimization problem");
    if (!profile_.OptimizePathParameter()) return InternalError("Error optimizing path parameter");
    // epilogue on the host for foreign path types (path_timing_trajectory.cc:458-472)
    const VectorXd &amax = path_->GetMaxJointAcceleration();
    for (size_t i = 0; i < N; i++) {
      const double v = profile_.GetPathVelocity()[i], a = profile_.GetPathAcceleration()[i];
      for (size_t d = 0; d < D; d++) {
        q[i * D + d] = path_->GetPathPositionAt(i)[d];
        const double d1 = path_->GetFirstPathDerivativeAt(i)[d], d2 = path_->GetSecondPathDerivativeAt(i)[d];
        qd[i * D + d] = d1 * v;
        qdd[i * D + d] = std::min(std::max(d1 * a + d2 * (v * v), -amax[d]), amax[d]);
      }
    }
  }

  // Drop what the new window replaces, then append it (path_timing_trajectory.cc:418-456).
  auto cut = [&](std::vector<double> &v, size_t stride) { v.resize((size_t)path_samples_offset * stride); };
  cut(time_at_path_samples_, 1); cut(path_parameter_at_path_samples_, 1);
  cut(path_velocity_at_path_samples_, 1); cut(path_acceleration_at_path_samples_, 1);
  cut(position_at_path_samples_, D); cut(velocity_at_path_samples_, D); cut(acceleration_at_path_samples_, D);
  auto app = [](std::vector<double> &v, const double *p, size_t n) { v.insert(v.end(), p, p + n); };
  app(time_at_path_samples_, profile_.GetTimeSamples().data(), N);
  app(path_parameter_at_path_samples_, profile_.GetPathParameter().data(), N);
  app(path_velocity_at_path_samples_, profile_.GetPathVelocity().data(), N);
  app(path_acceleration_at_path_samples_, profile_.GetPathAcceleration().data(), N);
  app(position_at_path_samples_, q.data(), N * D);
  app(velocity_at_path_samples_, qd.data(), N * D);
  app(acceleration_at_path_samples_, qdd.data(), N * D);
  return OkStatus();
}

// path_timing_trajectory.cc:579-684
Status PathTimingTrajectory::Plan(Time start, Duration time_horizon) {
  const double start_sec = TimeToSec(start);
  if (path_ == nullptr) return FailedPreconditionError("No path set.");
  if (Status st = HandleTimeArguments(start); !st.ok()) return st;
  UpdatePathTrackingStatus();
  const bool planned_enough = (path_->GetState() != TimeablePath::State::kNewPath) &&
                              (path_->GetState() != TimeablePath::State::kModifiedPath) &&
                              (final_decel_start_ >= start + time_horizon);
  if (!time_.empty() && planned_enough) {
    // Already planned far enough: drop the uniformly sampled part before `start`
    // (kUniformlyInTime branch of EraseTrajectoryBefore, path_timing_trajectory.cc:568-573).
    if (start_sec >= time_.front()) {
      const int offset = std::min<int>((int)std::round((start_sec - time_.front()) / time_step_sec_),
                                       (int)time_.size() - 1);
      auto drop = [&](auto &v) { v.erase(v.begin(), v.begin() + offset); };
      drop(time_); drop(path_parameter_); drop(path_parameter_derivative_);
      drop(second_path_parameter_derivative_); drop(positions_); drop(velocities_); drop(accelerations_);
    }
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
  Time loop_start_time = start;
  bool time_horizon_reached = false;
  const int N = (int)options_.GetNumPathSamples();
  for (int loop = 0; !planned_to_end_ && !time_horizon_reached; loop++) {
    if (Status st = ComputeTimingProfile(loop_start_time, start + time_horizon - loop_start_time); !st.ok())
      return st;
    const int decel_start = std::max(profile_.GetLastExtremalIndex(), N / 2);
    final_decel_start_ = TimeFromSec(profile_.GetTimeSamples()[decel_start]);
    planned_to_end_ = path_->CloseToEnd(path_horizon_);
    time_horizon_reached = (profile_.GetTimeSamples()[N - 1] - TimeToSec(start)) > time_horizon / Seconds(1);
    if (loop >= options_.GetMaxPlanningIterations())
      return DeadlineExceededError("Reached maximum number of planning loops");
    loop_start_time = final_decel_start_;
  }
  if (Status st = ResampleTrajectory(start_sec); !st.ok()) return st;
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

// path_timing_trajectory.cc:785-836 (host-side: SURVEY.md section 8f item 2).
void PathTimingTrajectory::ResampleSkippingSamplesCloserThanTimeStep(double start_sec) {
  const size_t D = options_.GetNumDofs();
  const int S = (int)time_at_path_samples_.size();
  auto lower_index = [&](int from, double t) {
    for (int i = from; i < S - 1; ++i) if (time_at_path_samples_[i + 1] > t) return i;
    return S - 1;
  };
  time_.clear(); positions_.clear(); velocities_.clear(); accelerations_.clear();
  path_parameter_.clear(); path_parameter_derivative_.clear(); second_path_parameter_derivative_.clear();
  const int lo = lower_index(0, start_sec);
  const int up = std::min(S - 1, lo + 1);
  const double span = time_at_path_samples_[up] - time_at_path_samples_[lo];
  const double at = std::fabs(span) < std::numeric_limits<double>::epsilon()
                        ? 0.5 : (start_sec - time_at_path_samples_[lo]) / span;
  auto lerp = [&](double a, double b) { return a + at * (b - a); };
  auto row = [&](const std::vector<double> &v, int i) { return VectorXd(v.data() + (size_t)i * D, D); };
  const VectorXd &amax = path_->GetMaxJointAcceleration();
  VectorXd p0(D), v0(D), a0(D);
  for (size_t d = 0; d < D; d++) {
    p0[d] = lerp(position_at_path_samples_[(size_t)lo * D + d], position_at_path_samples_[(size_t)up * D + d]);
    v0[d] = lerp(velocity_at_path_samples_[(size_t)lo * D + d], velocity_at_path_samples_[(size_t)up * D + d]);
    a0[d] = std::min(std::max(lerp(acceleration_at_path_samples_[(size_t)lo * D + d],
                                   acceleration_at_path_samples_[(size_t)up * D + d]), -amax[d]), amax[d]);
  }
  time_.push_back(start_sec); positions_.push_back(p0); velocities_.push_back(v0); accelerations_.push_back(a0);
  path_parameter_.push_back(lerp(path_parameter_at_path_samples_[lo], path_parameter_at_path_samples_[up]));
  path_parameter_derivative_.push_back(lerp(path_velocity_at_path_samples_[lo], path_velocity_at_path_samples_[up]));
  second_path_parameter_derivative_.push_back(
      lerp(path_acceleration_at_path_samples_[lo], path_acceleration_at_path_samples_[up]));
  const double keep = GetMinTimeDeltaToKeep();
  for (int i = lo + 1; i < S; ++i) {
    if (std::fabs(time_at_path_samples_[i] - time_.back()) < keep) continue;
    time_.push_back(time_at_path_samples_[i]);
    positions_.push_back(row(position_at_path_samples_, i));
    velocities_.push_back(row(velocity_at_path_samples_, i));
    accelerations_.push_back(row(acceleration_at_path_samples_, i));
    path_parameter_.push_back(path_parameter_at_path_samples_[i]);
    path_parameter_derivative_.push_back(path_velocity_at_path_samples_[i]);
    second_path_parameter_derivative_.push_back(path_acceleration_at_path_samples_[i]);
  }
  positions_.back() = row(position_at_path_samples_, S - 1);
  velocities_.back().setZero();
  accelerations_.back().setZero();
}

}  // namespace trajectory_planning
