#include "timeable_path_joint_spline.h"

#include <algorithm>
#include <cmath>
#include <limits>

#include "engine_handle.h"
#include "spline_edit.h"

namespace trajectory_planning {

using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;

namespace {
constexpr double kSmall = 1e-4;  // timeable_path_joint_spline.cc:33

// Corner offset for rounding a polyline corner (splines/spline_utils.cc:25-45).
VectorXd CornerOffset(const VectorXd &from, const VectorXd &to, double radius) {
  const size_t n = from.size();
  VectorXd delta(n), offset(n);
  for (size_t i = 0; i < n; i++) delta[i] = to[i] - from[i];
  const double norm = delta.norm();
  for (size_t i = 0; i < n; i++) offset[i] = norm > 1e-6 ? delta[i] / norm : 0.0;
  const double kSpacing = 4.0;  // spline_utils.h:44-46
  if (norm > kSpacing * radius) {
    for (size_t i = 0; i < n; i++) offset[i] = offset[i] * radius;
  } else {
    for (size_t i = 0; i < n; i++) offset[i] = offset[i] * (1.0 / kSpacing) * norm;
  }
  return offset;
}

VectorXd Plus(const VectorXd &a, const VectorXd &b) {
  VectorXd r(a.size());
  for (size_t i = 0; i < a.size(); i++) r[i] = a[i] + b[i];
  return r;
}
}  // namespace

TimeableJointSplinePath::TimeableJointSplinePath(const JointPathOptions &options)
    : options_(options) {
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  path_position_.assign(N, VectorXd(D));
  first_path_derivative_.assign(N, VectorXd(D));
  second_path_derivative_.assign(N, VectorXd(D));
  constraints_.resize(N);
  for (auto &c : constraints_) c.resize((int)(2 * D));
  max_joint_velocity_ = VectorXd(D);
  max_joint_acceleration_ = VectorXd(D);
  initial_velocity_ = VectorXd(D);
  Reset();
}

void TimeableJointSplinePath::Reset() {
  waypoints_.clear();
  control_points_.clear();
  packed_control_points_.clear();
  knots_.clear();
  path_state_ = State::kNoPath;
  parameter_start_ = std::numeric_limits<double>::quiet_NaN();
  parameter_end_ = std::numeric_limits<double>::quiet_NaN();
  initial_velocity_.setZero();
}

Status TimeableJointSplinePath::SetMaxJointVelocity(Span<const double> v) {
  if (v.size() != options_.num_dofs()) return InvalidArgumentError("max_velocity has the wrong dimension");
  max_joint_velocity_ = VectorXd(v.data(), v.size());
  return OkStatus();
}

Status TimeableJointSplinePath::SetMaxJointAcceleration(Span<const double> a) {
  if (a.size() != options_.num_dofs()) return InvalidArgumentError("max_acceleration has the wrong dimension");
  max_joint_acceleration_ = VectorXd(a.data(), a.size());
  return OkStatus();
}

Status TimeableJointSplinePath::SetInitialVelocity(Span<const double> v) {
  if (v.size() != NumDofs()) return InvalidArgumentError("Velocity dimension doesn't match number of dofs.");
  initial_velocity_ = VectorXd(v.data(), v.size());
  return OkStatus();
}

bool TimeableJointSplinePath::CloseToEnd(double parameter) const {
  return knots_.empty() || parameter >= knots_.back() - kSmall;
}

Status TimeableJointSplinePath::SetWaypoints(Span<const VectorXd> waypoints) {
  for (const auto &wp : waypoints)
    if (wp.size() != options_.num_dofs()) return InvalidArgumentError("waypoint has the wrong dimension");
  waypoints_.assign(waypoints.begin(), waypoints.end());
  path_state_ = State::kNewPath;
  return FitSplineToWaypoints();
}

// Waypoints -> 3W-2 control points with rounded corners (splines/spline_utils.cc:47-102),
// uniform degree-2 knots by running accumulation (splines/bspline_base.cc:356-381),
// scaled by the control polygon length (timeable_path_joint_spline.cc:252-292).
void TimeableJointSplinePath::PolyLineToControlPoints(const std::vector<VectorXd> &waypoints, double radius,
                                                      std::vector<VectorXd> *control_points) {
  const size_t W = waypoints.size();
  auto &cp = *control_points;
  if (W == 1) {
    cp.assign(4, waypoints.front());
    return;
  }
  cp.assign(3 * W - 2, VectorXd(waypoints.front().size()));
  for (size_t i = 0; i < W; i++) cp[3 * i] = waypoints[i];
  for (size_t i = 1; i + 1 < W; i++) {
    const size_t k = 3 * i;
    cp[k + 1] = Plus(cp[k], CornerOffset(cp[k], cp[k + 3], radius));
    cp[k - 1] = Plus(cp[k], CornerOffset(cp[k], cp[k - 3], radius));
  }
  cp[1] = Plus(cp[0], CornerOffset(cp[0], cp[3], radius));
  const size_t sz = cp.size();
  cp[sz - 2] = Plus(cp[sz - 1], CornerOffset(cp[sz - 1], cp[sz - 4], radius));
}

void TimeableJointSplinePath::PackControlPoints() {
  const size_t P = control_points_.size(), D = options_.num_dofs();
  packed_control_points_.resize(P * D);
  for (size_t i = 0; i < P; i++)
    for (size_t d = 0; d < D; d++) packed_control_points_[i * D + d] = control_points_[i][d];
}

Status TimeableJointSplinePath::FitSplineToWaypoints() {
  if (waypoints_.empty()) return InvalidArgumentError("Control point vector empty.");
  const size_t D = options_.num_dofs();
  PolyLineToControlPoints(waypoints_, options_.rounding(), &control_points_);
  const size_t P = control_points_.size();
  const size_t nk = P + kSplineOrder + 1;
  knots_.assign(nk, 0.0);
  const double spacing = (1.0 / (nk - 2.0 * (kSplineOrder + 1.0) + 1.0)) * (1.0 - 0.0);
  for (size_t i = kSplineOrder + 1; i < nk - kSplineOrder - 1; i++) knots_[i] = knots_[i - 1] + spacing;
  for (size_t i = nk - kSplineOrder - 1; i < nk; i++) knots_[i] = 1.0;
  double length = 0.0;
  for (size_t i = 0; i + 1 < P; i++) {
    VectorXd diff(D);
    for (size_t d = 0; d < D; d++) diff[d] = control_points_[i + 1][d] - control_points_[i][d];
    length += diff.norm();
  }
  const double weighted = std::max(length * 1.0, 0.1);
  for (double &k : knots_) k *= weighted;
  PackControlPoints();
  return OkStatus();
}

// timeable_path_joint_spline.cc:209-250
Status TimeableJointSplinePath::SwitchToWaypointPath(const double keep_path_until,
                                                     Span<const VectorXd> waypoints) {
  for (const auto &wp : waypoints)
    if (wp.size() != options_.num_dofs()) return InvalidArgumentError("waypoint has the wrong dimension");
  if (knots_.empty()) return ::tpamd::compat::FailedPreconditionError("No path to switch from.");
  path_state_ = State::kModifiedPath;
  // the reference allocates twice the initial knot count, at least 100 (:262-268)
  const int capacity = std::max<int>(2 * (int)knots_.size() + 3 * (int)waypoints.size() + 8, 100);
  EditableBSpline spline;
  if (Status st = spline.Init(kSplineOrder, capacity, {knots_.data(), knots_.size()},
                              {control_points_.data(), control_points_.size()});
      !st.ok())
    return st;
  if (Status st = spline.TruncateSplineAt(keep_path_until); !st.ok()) return st;
  VectorXd switch_position(NumDofs());
  if (Status st = spline.EvalCurve(keep_path_until, &switch_position); !st.ok()) return st;
  const auto projection = ProjectPointOnPath(waypoints, switch_position);
  if (!projection.ok()) return projection.status();
  std::vector<VectorXd> new_waypoints;
  new_waypoints.reserve(waypoints.size() + 1);
  // the projected point becomes the first waypoint unless it coincides with the switch position
  constexpr double kEpsilon = 1e-3;
  double inf_norm = 0.0;
  for (size_t d = 0; d < NumDofs(); d++)
    inf_norm = std::max(inf_norm, std::fabs(switch_position[d] - (*projection).projected_point[d]));
  if (inf_norm > kEpsilon) new_waypoints.push_back((*projection).projected_point);
  const int first_waypoint = (*projection).line_parameter >= 0 ? (*projection).waypoint_index + 1
                                                               : (*projection).waypoint_index;
  for (size_t i = (size_t)first_waypoint; i < waypoints.size(); i++) new_waypoints.push_back(waypoints[i]);
  if (new_waypoints.empty()) return InvalidArgumentError("No waypoints left after the switch position.");
  std::vector<VectorXd> extra;
  PolyLineToControlPoints(new_waypoints, options_.rounding(), &extra);
  if (Status st = spline.ExtendWithControlPoints({extra.data(), extra.size()}); !st.ok()) return st;
  knots_ = spline.knots();
  control_points_ = spline.control_points();
  PackControlPoints();
  return OkStatus();
}

void TimeableJointSplinePath::AdoptSamples(double path_start, const double *q, const double *q1,
                                           const double *q2) {
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  parameter_start_ = path_start;
  parameter_end_ = N * options_.delta_parameter();  // sic: timeable_path_joint_spline.cc:296
  for (size_t i = 0; i < N; i++)
    for (size_t d = 0; d < D; d++) {
      path_position_[i][d] = q[i * D + d];
      first_path_derivative_[i][d] = q1[i * D + d];
      second_path_derivative_[i][d] = q2[i * D + d];
    }
  path_state_ = State::kPathWasSampled;
}

Status TimeableJointSplinePath::SamplePath(const double path_start) {
  if (knots_.empty()) return ::tpamd::compat::FailedPreconditionError("Call SetWaypoints first.");
  ::tpamd::EngineLease lease = ::tpamd::acquire_engine();
  tpamd_engine *engine = lease.get();
  if (!engine) return InternalError("no GPU engine");
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  std::vector<double> q(N * D), q1(N * D), q2(N * D);
  const double delta = options_.delta_parameter();
  {
    const int rc = tpamd_sample_joint_paths_host(engine, 1, (int)D, (int)N, num_control_points(),
                                                 knots_.data(), packed_control_points_.data(),
                                                 &path_start, &delta, q.data(), q1.data(), q2.data());
    if (rc != 0) return InternalError(tpamd_error_string(rc));
  }
  AdoptSamples(path_start, q.data(), q1.data(), q2.data());
  return OkStatus();
}

// timeable_path_joint_spline.cc:320-343 (only needed when a caller wants the rows; the
// fused engine call forms them on the device).
Status TimeableJointSplinePath::ConstraintSetup() {
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  const double safety = options_.constraint_safety();
  for (size_t idx = 0; idx < N; idx++) {
    auto &c = constraints_[idx];
    for (size_t dof = 0; dof < D; dof++) {
      const double d1 = first_path_derivative_[idx][dof];
      c.a_coefficient((int)dof) = d1;
      c.b_coefficient((int)dof) = second_path_derivative_[idx][dof];
      c.upper((int)dof) = max_joint_acceleration_[dof] * safety;
      c.lower((int)dof) = -max_joint_acceleration_[dof] * safety;
      c.a_coefficient((int)(D + dof)) = 0.0;
      c.b_coefficient((int)(D + dof)) = d1 * d1;
      const double v = max_joint_velocity_[dof] * safety;
      c.upper((int)(D + dof)) = v * v;
      c.lower((int)(D + dof)) = 0.0;
    }
  }
  return OkStatus();
}

}  // namespace trajectory_planning
