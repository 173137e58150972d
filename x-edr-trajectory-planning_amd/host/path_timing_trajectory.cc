#include "path_timing_trajectory.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <tuple>

#include "engine_handle.h"
#include "timeable_path_cartesian_spline.h"
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

// All windows a Plan() call needs, for the TimeableJointSplinePath planners of one shape, chained
// on the device (tpamd_plan_joint_windows_host: path_timing_trajectory.cc:628-660 around
// ComputeTimingProfile :307-475). The planners' window histories go up once, the extended
// histories and each planner's last window come back once.
void PathTimingTrajectory::PlanJointWindowsOnDevice(const std::vector<PathTimingTrajectory *> &planners,
                                                    const std::vector<size_t> &ids, Time start,
                                                    Duration time_horizon, std::vector<Status> *status) {
  ::tpamd::EngineLease lease = ::tpamd::acquire_engine(planners[ids[0]]->device_);
  tpamd_engine *engine = lease.get();
  if (!engine) {
    for (size_t id : ids) (*status)[id] = InternalError("no GPU engine");
    return;
  }
  const size_t B = ids.size();
  auto *first = dynamic_cast<TimeableJointSplinePath *>(planners[ids[0]]->path_.get());
  const size_t D = first->NumDofs(), N = first->NumPathSamples(), P = first->num_control_points();
  std::vector<double> knots(B * (P + 3)), cps(B * P * D), vmax(B * D), amax(B * D), dl(B), iv(B * D);
  std::vector<int64_t> start_ns(B, ::tpamd::compat::ToUnixNanos(start)), horizon_ns(B, time_horizon.nanos());
  std::vector<int32_t> path_state(B), pte(B), count(B);
  size_t max_count = 0;
  int max_iterations = 0;
  double max_velocity_error = 0.0;
  for (size_t g = 0; g < B; g++) {
    PathTimingTrajectory *pl = planners[ids[g]];
    auto *joint = dynamic_cast<TimeableJointSplinePath *>(pl->path_.get());
    std::copy(joint->knots().begin(), joint->knots().end(), knots.begin() + g * (P + 3));
    std::copy(joint->packed_control_points().begin(), joint->packed_control_points().end(),
              cps.begin() + g * P * D);
    for (size_t d = 0; d < D; d++) {
      vmax[g * D + d] = joint->GetMaxJointVelocity()[d];
      amax[g * D + d] = joint->GetMaxJointAcceleration()[d];
      iv[g * D + d] = joint->GetInitialVelocity()[d];
    }
    dl[g] = joint->GetPathSamplingDistance();
    switch (joint->GetState()) {
      case TimeablePath::State::kNewPath: path_state[g] = 1; break;
      case TimeablePath::State::kModifiedPath: path_state[g] = 2; break;
      case TimeablePath::State::kPathWasSampled: path_state[g] = 3; break;
      default: path_state[g] = 0; break;
    }
    pte[g] = pl->planned_to_end_ ? 1 : 0;
    count[g] = (int32_t)pl->time_at_path_samples_.size();
    max_count = std::max(max_count, pl->time_at_path_samples_.size());
    // the engine call takes one limit for the group; planners of one shape share their options in
    // every use we know of, the strictest one is applied otherwise
    max_iterations = g == 0 ? pl->options_.GetMaxPlanningIterations()
                            : std::min(max_iterations, pl->options_.GetMaxPlanningIterations());
    max_velocity_error = g == 0 ? pl->options_.GetMaxInitialVelocityError()
                                : std::min(max_velocity_error, pl->options_.GetMaxInitialVelocityError());
  }
  std::vector<double> ph(B), wps(B), wsd0(B), wt0(B), wdtm(B);
  std::vector<int64_t> fds(B), loop_start(B);
  std::vector<int32_t> wlei(B), st(B), windows(B), loop_count(B), looping(B);
  std::vector<double> wt(B * N), ws(B * N), wsd(B * N), wsdd(B * N), wsd2(B * N), wq(B * N * D), wq1(B * N * D),
      wq2(B * N * D);
  for (size_t g = 0; g < B; g++) {
    ph[g] = planners[ids[g]]->path_horizon_;
    fds[g] = ::tpamd::compat::ToUnixNanos(planners[ids[g]]->final_decel_start_);
  }
  size_t cap = max_count + 8 * N;
  std::vector<double> ht, hs, hsd, hsdd, hq, hqd, hqdd;
  auto pack_history = [&](size_t new_cap, bool from_planners) {
    std::vector<double> nt(B * new_cap), ns(B * new_cap), nsd(B * new_cap), nsdd(B * new_cap),
        nq(B * new_cap * D), nqd(B * new_cap * D), nqdd(B * new_cap * D);
    for (size_t g = 0; g < B; g++) {
      const size_t c = (size_t)count[g];
      if (from_planners) {
        PathTimingTrajectory *pl = planners[ids[g]];
        std::copy_n(pl->time_at_path_samples_.begin(), c, nt.begin() + g * new_cap);
        std::copy_n(pl->path_parameter_at_path_samples_.begin(), c, ns.begin() + g * new_cap);
        std::copy_n(pl->path_velocity_at_path_samples_.begin(), c, nsd.begin() + g * new_cap);
        std::copy_n(pl->path_acceleration_at_path_samples_.begin(), c, nsdd.begin() + g * new_cap);
        std::copy_n(pl->position_at_path_samples_.begin(), c * D, nq.begin() + g * new_cap * D);
        std::copy_n(pl->velocity_at_path_samples_.begin(), c * D, nqd.begin() + g * new_cap * D);
        std::copy_n(pl->acceleration_at_path_samples_.begin(), c * D, nqdd.begin() + g * new_cap * D);
      } else {
        std::copy_n(ht.begin() + g * cap, c, nt.begin() + g * new_cap);
        std::copy_n(hs.begin() + g * cap, c, ns.begin() + g * new_cap);
        std::copy_n(hsd.begin() + g * cap, c, nsd.begin() + g * new_cap);
        std::copy_n(hsdd.begin() + g * cap, c, nsdd.begin() + g * new_cap);
        std::copy_n(hq.begin() + g * cap * D, c * D, nq.begin() + g * new_cap * D);
        std::copy_n(hqd.begin() + g * cap * D, c * D, nqd.begin() + g * new_cap * D);
        std::copy_n(hqdd.begin() + g * cap * D, c * D, nqdd.begin() + g * new_cap * D);
      }
    }
    ht.swap(nt); hs.swap(ns); hsd.swap(nsd); hsdd.swap(nsdd); hq.swap(nq); hqd.swap(nqd); hqdd.swap(nqdd);
    cap = new_cap;
  };
  pack_history(cap, true);
  std::vector<int32_t> total_windows(B, 0);
  std::vector<char> have_window(B, 0);
  std::vector<double> kt(B * N), ks(B * N), ksd(B * N), ksdd(B * N), ksd2(B * N), kq(B * N * D), kq1(B * N * D),
      kq2(B * N * D), kps(B), ksd0(B), kt0(B), kdtm(B);
  std::vector<int32_t> klei(B);
  for (int round = 0, resume = 0;; round++) {
    tpamd_plan_args a{};
    a.num_planners = (int32_t)B; a.num_dofs = (int32_t)D; a.num_samples = (int32_t)N; a.num_points = (int32_t)P;
    a.history_capacity = (int32_t)cap;
    a.max_planning_iterations = max_iterations;
    a.constraint_safety = first->options().constraint_safety();
    a.max_initial_velocity_error = max_velocity_error;
    a.knots = knots.data(); a.control_points = cps.data(); a.max_velocity = vmax.data();
    a.max_acceleration = amax.data(); a.delta = dl.data(); a.initial_velocity = iv.data();
    a.start_ns = start_ns.data(); a.horizon_ns = horizon_ns.data();
    a.path_state = path_state.data(); a.planned_to_end = pte.data(); a.history_count = count.data();
    a.history_time = ht.data(); a.history_s = hs.data(); a.history_sd = hsd.data(); a.history_sdd = hsdd.data();
    a.history_q = hq.data(); a.history_qd = hqd.data(); a.history_qdd = hqdd.data();
    a.path_horizon = ph.data(); a.final_decel_start_ns = fds.data();
    a.window_time = wt.data(); a.window_s = ws.data(); a.window_sd = wsd.data(); a.window_sdd = wsdd.data();
    a.window_sd2 = wsd2.data(); a.window_q = wq.data(); a.window_q1 = wq1.data(); a.window_q2 = wq2.data();
    a.window_path_start = wps.data(); a.window_sd_start = wsd0.data(); a.window_time_start = wt0.data();
    a.window_last_extremal_index = wlei.data(); a.window_max_time_increment = wdtm.data();
    a.status = st.data(); a.windows = windows.data();
    a.resume = resume; a.loop_start_ns = loop_start.data(); a.loop_count = loop_count.data();
    a.looping = looping.data();
    int rc;
    {
      rc = tpamd_plan_joint_windows_host(engine, &a);
    }
    if (rc != 0) {
      for (size_t id : ids) (*status)[id] = InternalError(tpamd_error_string(rc));
      return;
    }
    bool more = false;
    for (size_t g = 0; g < B; g++) {
      if (windows[g] > 0) {       // keep the planner's most recent window
        have_window[g] = 1;
        total_windows[g] += windows[g];
        std::copy_n(wt.begin() + g * N, N, kt.begin() + g * N); std::copy_n(ws.begin() + g * N, N, ks.begin() + g * N);
        std::copy_n(wsd.begin() + g * N, N, ksd.begin() + g * N); std::copy_n(wsdd.begin() + g * N, N, ksdd.begin() + g * N);
        std::copy_n(wsd2.begin() + g * N, N, ksd2.begin() + g * N);
        std::copy_n(wq.begin() + g * N * D, N * D, kq.begin() + g * N * D);
        std::copy_n(wq1.begin() + g * N * D, N * D, kq1.begin() + g * N * D);
        std::copy_n(wq2.begin() + g * N * D, N * D, kq2.begin() + g * N * D);
        kps[g] = wps[g]; ksd0[g] = wsd0[g]; kt0[g] = wt0[g]; kdtm[g] = wdtm[g]; klei[g] = wlei[g];
      }
      if (st[g] == TPAMD_PLAN_MORE) more = true;
    }
    if (!more || round > 64) break;
    pack_history(2 * cap, false);   // room for more windows, then continue where the loop stopped
    resume = 1;
  }
  for (size_t g = 0; g < B; g++) {
    PathTimingTrajectory *pl = planners[ids[g]];
    auto *joint = dynamic_cast<TimeableJointSplinePath *>(pl->path_.get());
    if (have_window[g]) {
      const size_t c = (size_t)count[g];
      pl->time_at_path_samples_.assign(ht.begin() + g * cap, ht.begin() + g * cap + c);
      pl->path_parameter_at_path_samples_.assign(hs.begin() + g * cap, hs.begin() + g * cap + c);
      pl->path_velocity_at_path_samples_.assign(hsd.begin() + g * cap, hsd.begin() + g * cap + c);
      pl->path_acceleration_at_path_samples_.assign(hsdd.begin() + g * cap, hsdd.begin() + g * cap + c);
      pl->position_at_path_samples_.assign(hq.begin() + g * cap * D, hq.begin() + (g * cap + c) * D);
      pl->velocity_at_path_samples_.assign(hqd.begin() + g * cap * D, hqd.begin() + (g * cap + c) * D);
      pl->acceleration_at_path_samples_.assign(hqdd.begin() + g * cap * D, hqdd.begin() + (g * cap + c) * D);
      pl->path_start_ = kps[g];
      pl->path_start_velocity_ = ksd0[g];
      pl->path_time_start_ = kt0[g];
      pl->path_horizon_ = ph[g];
      pl->planned_to_end_ = pte[g] != 0;
      pl->final_decel_start_ = ::tpamd::compat::FromUnixNanos(fds[g]);
      joint->AdoptSamples(kps[g], &kq[g * N * D], &kq1[g * N * D], &kq2[g * N * D]);
      pl->profile_.AdoptSolution((int)N, (int)(2 * D), kps[g], ph[g], &kt[g * N], &ks[g * N], &ksd[g * N],
                                 &ksdd[g * N], &ksd2[g * N], klei[g], kdtm[g]);
    }
    switch (st[g]) {
      case TPAMD_PLAN_OK: break;
      case TPAMD_PLAN_FAILED_PRECONDITION:
        (*status)[ids[g]] = FailedPreconditionError("no previous window to connect to"); break;
      case TPAMD_PLAN_INVALID_ARGUMENT:
        (*status)[ids[g]] = InvalidArgumentError(
            "Duration must be positive / could not satisfy initial velocity (probably not parallel to "
            "initial tangent)");
        break;
      case TPAMD_PLAN_DEADLINE_EXCEEDED:
        (*status)[ids[g]] = DeadlineExceededError("Reached maximum number of planning loops"); break;
      case TPAMD_PLAN_MORE:   // (64 doublings of the history buffer were not enough: not a solver failure)
        (*status)[ids[g]] = ::tpamd::compat::ResourceExhaustedError(
            "window history still growing after 64 extensions of its buffer"); break;
      default: (*status)[ids[g]] = InternalError("Error optimizing path parameter"); break;
    }
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

// One window each for planners whose path is a TimeableCartesianSplinePath: the path samples its
// pose splines on the GPU and runs the user's path-IK callback (SamplePath), the Jacobian callback
// is evaluated at every sample, and then everything that follows in ComputeTimingProfile --
// path derivatives, the 2D + 2 constraint rows, the solver, the epilogue -- is ONE engine call per
// group of equal (dofs, samples, safety): tpamd_time_cartesian_paths_host.
void PathTimingTrajectory::SolveCartesianWindows(const std::vector<PathTimingTrajectory *> &planners,
                                                 const std::vector<size_t> &ids, std::vector<Window> *windows,
                                                 std::vector<Status> *status) {
  std::map<std::tuple<int, size_t, size_t, double>, std::vector<size_t>> groups;
  for (size_t i : ids) {
    PathTimingTrajectory *pl = planners[i];
    auto *path = dynamic_cast<TimeableCartesianSplinePath *>(pl->path_.get());
    Status st = path->SamplePath(pl->path_start_);
    if (st.ok()) st = pl->ProjectStartVelocity((*windows)[i]);
    (*status)[i] = st;
    if (st.ok())
      groups[std::make_tuple(pl->device_, path->NumDofs(), path->NumPathSamples(),
                             path->options().constraint_safety())].push_back(i);
  }
  for (const auto &kv : groups) {
    const std::vector<size_t> &g = kv.second;
    const size_t B = g.size(), D = std::get<1>(kv.first), N = std::get<2>(kv.first);
    std::vector<double> q, J, vmax(B * D), amax(B * D), vt(B), vr(B), ps(B), dl(B), sd0(B), sdd0(B), t0(B);
    q.reserve(B * N * D); J.reserve(B * N * 6 * D);
    bool packed = true;
    for (size_t k = 0; k < B && packed; k++) {
      PathTimingTrajectory *pl = planners[g[k]];
      auto *path = dynamic_cast<TimeableCartesianSplinePath *>(pl->path_.get());
      const Status st = path->PackSampledWindow(&q, &J);      // the user's Jacobian callback, once per sample
      if (!st.ok()) { for (size_t i : g) (*status)[i] = st; packed = false; break; }
      for (size_t d = 0; d < D; d++) {
        vmax[k * D + d] = path->GetMaxJointVelocity()[d];
        amax[k * D + d] = path->GetMaxJointAcceleration()[d];
      }
      vt[k] = path->max_translational_velocity(); vr[k] = path->max_rotational_velocity();
      ps[k] = pl->path_start_; dl[k] = path->GetPathSamplingDistance();
      sd0[k] = pl->path_start_velocity_; sdd0[k] = pl->path_start_acceleration_; t0[k] = pl->path_time_start_;
    }
    if (!packed) continue;
    std::vector<double> t(B * N), s(B * N), sd(B * N), sdd(B * N), sd2(B * N), qd(B * N * D), qdd(B * N * D), dtm(B);
    std::vector<int32_t> st(B, -1), lei(B, 0);
    tpamd_cartesian_batch batch{(int)B, (int)D, (int)N, 0, std::get<3>(kv.first)};
    tpamd_cartesian_inputs in{q.data(), J.data(), vmax.data(), amax.data(), vt.data(), vr.data(), ps.data(),
                              dl.data(), sd0.data(), sdd0.data(), t0.data()};
    tpamd_path_outputs out{t.data(), s.data(), sd.data(), sdd.data(), nullptr, qd.data(), qdd.data(), lei.data(),
                           dtm.data(), st.data(), sd2.data()};
    int rc;
    {
      ::tpamd::EngineLease lease = ::tpamd::acquire_engine(planners[g[0]]->device_);
      rc = lease ? tpamd_time_cartesian_paths_host(lease.get(), &batch, &in, &out) : TPAMD_E_NO_DEVICE;
    }
    for (size_t k = 0; k < B; k++) {
      const size_t i = g[k];
      PathTimingTrajectory *pl = planners[i];
      if (rc != 0) { (*status)[i] = InternalError(tpamd_error_string(rc)); continue; }
      if (st[k] != 0) { (*status)[i] = InternalError("Error optimizing path parameter"); continue; }
      pl->profile_.AdoptSolution((int)N, (int)(2 * D + 2), pl->path_start_, pl->path_horizon_, &t[k * N], &s[k * N],
                                 &sd[k * N], &sdd[k * N], &sd2[k * N], lei[k], dtm[k]);
      Window &w = (*windows)[i];
      std::copy_n(q.begin() + k * N * D, N * D, w.q.begin());
      std::copy_n(qd.begin() + k * N * D, N * D, w.qd.begin());
      std::copy_n(qdd.begin() + k * N * D, N * D, w.qdd.begin());
      w.status = 0;
    }
  }
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
  // TimeableJointSplinePath planners: all windows of this call chained on the device, one engine
  // call per shape group. Other TimeablePath types: the window loop on the host below.
  {
    std::map<std::tuple<int, size_t, size_t, size_t, double, int, double>, std::vector<size_t>> groups;
    for (size_t i = 0; i < P; i++) {
      if (!finish[i]) continue;
      if (auto *joint = dynamic_cast<TimeableJointSplinePath *>(planners[i]->path_.get())) {
        // (planners whose options differ in the loop limit or the start-velocity tolerance are
        // solved by separate engine calls: one value of each per call)
        groups[std::make_tuple(planners[i]->device_, joint->NumDofs(), joint->NumPathSamples(),
                               (size_t)joint->num_control_points(), joint->options().constraint_safety(),
                               planners[i]->options_.GetMaxPlanningIterations(),
                               planners[i]->options_.GetMaxInitialVelocityError())].push_back(i);
        looping[i] = 0;            // handled here, not by the host loop
      }
    }
    for (const auto &kv : groups) {
      PlanJointWindowsOnDevice(planners, kv.second, start, time_horizon, &status);
      for (size_t i : kv.second)
        if (!status[i].ok()) finish[i] = 0;
    }
  }
  for (;;) {
    std::vector<size_t> foreign, cartesian;
    bool any = false;
    for (size_t i = 0; i < P; i++) {
      if (!looping[i]) continue;
      any = true;
      PathTimingTrajectory *pl = planners[i];
      status[i] = pl->BeginWindow(loop_start[i], start + time_horizon - loop_start[i], &windows[i]);
      if (!status[i].ok()) { looping[i] = 0; finish[i] = 0; continue; }
      if (dynamic_cast<TimeableCartesianSplinePath *>(pl->path_.get())) cartesian.push_back(i);
      else foreign.push_back(i);
    }
    if (!any) break;
    for (size_t i : foreign) status[i] = planners[i]->SolveWindowOnHost(&windows[i]);
    if (!cartesian.empty()) SolveCartesianWindows(planners, cartesian, &windows, &status);
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
  ::tpamd::EngineLease lease = ::tpamd::acquire_engine(device_);
  tpamd_engine *engine = lease.get();
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
  ::tpamd::EngineLease lease = ::tpamd::acquire_engine(device_);
  tpamd_engine *engine = lease.get();
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
