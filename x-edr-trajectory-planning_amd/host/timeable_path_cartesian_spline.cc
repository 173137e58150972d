#include "timeable_path_cartesian_spline.h"

#include <algorithm>
#include <cmath>
#include <limits>

#include "engine_handle.h"
#include "spline_edit.h"
#include "timeable_path_joint_spline.h"

namespace trajectory_planning {

using ::tpamd::compat::AngleAxisd;
using ::tpamd::compat::FailedPreconditionError;
using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;
using ::tpamd::compat::UnimplementedError;

namespace {
constexpr double kSmall = 1e-4;   // timeable_path_cartesian_spline.cc:37

// timeable_path_cartesian_spline.cc:39-68: forward differences of the IK solution; the last
// first derivative and the two outer second derivatives are zero.
void ComputePathDerivatives(const std::vector<VectorXd> &path, double delta_parameter,
                            std::vector<VectorXd> *first_derivative, std::vector<VectorXd> *second_derivative) {
  const int n = (int)path.size();
  const size_t D = path[0].size();
  const double inv = 1.0 / delta_parameter;
  auto &d1 = *first_derivative;
  auto &d2 = *second_derivative;
  for (int i = 0; i < n - 1; i++)
    for (size_t d = 0; d < D; d++) d1[i][d] = inv * (path[i + 1][d] - path[i][d]);
  d1[n - 1].setZero();
  for (int i = 1; i < n - 1; i++)
    for (size_t d = 0; d < D; d++) d2[i][d] = inv * (d1[i + 1][d] - d1[i][d]);
  d2[0].setZero();
  d2[n - 1].setZero();
}

// splines/spline_utils.cc:104-148
Pose3d CornerOffset(const Pose3d &delta, double translation_radius, double rotation_radius) {
  Pose3d offset;
  constexpr double kMinRadius = 1e-6;
  if (translation_radius < kMinRadius || rotation_radius < kMinRadius) return offset;
  const double translation_norm = delta.translation().norm();
  AngleAxisd delta_rotation(delta.quaternion());
  const double rotation_angle = delta_rotation.angle;
  const double inf = std::numeric_limits<double>::infinity();
  const double pct_trans = translation_norm == 0.0 ? inf : translation_radius / translation_norm;
  const double pct_rot = rotation_angle == 0.0 ? inf : rotation_radius / rotation_angle;
  double pct = std::min(pct_trans, pct_rot);
  constexpr double kMinWaypointSpacingFactor = 4.0;   // spline_utils.h:44-46
  if (pct > (1.0 / kMinWaypointSpacingFactor)) pct = (1.0 / kMinWaypointSpacingFactor);
  offset.translation() = delta.translation() * pct;
  delta_rotation.angle *= pct;
  offset.setQuaternion(delta_rotation.toQuaternion());
  return offset;
}
}  // namespace

// splines/spline_utils.cc:150-204
void PolyLineToBspline3Waypoints(const std::vector<Pose3d> &corners, double translation_radius,
                                 double rotational_radius, std::vector<Pose3d> *output) {
  auto &out = *output;
  if (corners.size() == 1) {
    out.assign(4, corners.front());
    return;
  }
  out.assign(3 * corners.size() - 2, Pose3d());
  for (size_t i = 0; i < corners.size(); i++) out[3 * i] = corners[i];
  for (size_t i = 1; i + 1 < corners.size(); i++) {
    const size_t k = 3 * i;
    out[k + 1] = out[k] * CornerOffset(out[k].inverse() * out[k + 3], translation_radius, rotational_radius);
    out[k - 1] = out[k] * CornerOffset(out[k].inverse() * out[k - 3], translation_radius, rotational_radius);
  }
  out[1] = out[0] * CornerOffset(out[0].inverse() * out[3], translation_radius, rotational_radius);
  const size_t sz = out.size();
  out[sz - 2] = out[sz - 1] * CornerOffset(out[sz - 1].inverse() * out[sz - 4], translation_radius, rotational_radius);
}

TimeableCartesianSplinePath::TimeableCartesianSplinePath(const CartesianPathOptions &options)
    : options_(options), num_constraints_(options.num_dofs() * 2 + 2) {
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  path_ik_func_ = options.GetPathIKFunc();
  jacobian_func_ = options.GetJacobianFunc();
  path_position_.assign(N, VectorXd(D));
  first_path_derivative_.assign(N, VectorXd(D));
  second_path_derivative_.assign(N, VectorXd(D));
  constraints_.resize(N);
  for (auto &c : constraints_) c.resize((int)num_constraints_);
  max_joint_velocity_ = VectorXd(D);
  max_joint_acceleration_ = VectorXd(D);
  initial_velocity_ = VectorXd(D);
  Reset();
}

void TimeableCartesianSplinePath::Reset() {
  joint_waypoints_.clear();
  pose_waypoints_.clear();
  joint_control_points_.clear();
  pose_control_points_.clear();
  path_ik_positions_.clear();
  path_state_ = State::kNoPath;
  parameter_start_ = -1.0;
  parameter_end_ = -1.0;
}

Status TimeableCartesianSplinePath::SetWaypoints(Span<const Pose3d> pose_waypoints,
                                                 Span<const VectorXd> joint_waypoints) {
  path_state_ = State::kNewPath;
  if (options_.num_path_samples() < 3) return FailedPreconditionError("need at least 3 path samples");
  if (!path_ik_func_ || !jacobian_func_)
    return FailedPreconditionError("Need a path IK function and a Jacobian function set in Options.");
  if (joint_waypoints.size() != pose_waypoints.size())
    return InvalidArgumentError("'joint_waypoints' and 'pose_waypoints' have different sizes.");
  if (joint_waypoints.empty()) return InvalidArgumentError("no waypoints");
  for (const auto &wp : joint_waypoints)
    if (wp.size() != options_.num_dofs()) return InvalidArgumentError("Dimension error in a joint waypoint.");
  joint_waypoints_.assign(joint_waypoints.begin(), joint_waypoints.end());
  pose_waypoints_.assign(pose_waypoints.begin(), pose_waypoints.end());
  path_ik_positions_.clear();
  return FitSplineToWaypoints();
}

Status TimeableCartesianSplinePath::SwitchToWaypointPath(double, Span<const Pose3d>, Span<const VectorXd>) {
  return UnimplementedError(
      "TimeableCartesianSplinePath::SwitchToWaypointPath needs knot insertion into the quaternion "
      "spline (BSplineQ), which this mirror does not carry (SURVEY.md section 2)");
}

Status TimeableCartesianSplinePath::SetMaxJointVelocity(Span<const double> v) {
  if (v.size() != options_.num_dofs()) return InvalidArgumentError("max_velocity has the wrong dimension");
  max_joint_velocity_ = VectorXd(v.data(), v.size());
  return OkStatus();
}

Status TimeableCartesianSplinePath::SetMaxJointAcceleration(Span<const double> a) {
  if (a.size() != options_.num_dofs()) return InvalidArgumentError("max_acceleration has the wrong dimension");
  max_joint_acceleration_ = VectorXd(a.data(), a.size());
  return OkStatus();
}

Status TimeableCartesianSplinePath::SetInitialVelocity(Span<const double> v) {
  if (v.size() != NumDofs()) return InvalidArgumentError("Velocity dimension doesn't match number of dofs.");
  initial_velocity_ = VectorXd(v.data(), v.size());
  return OkStatus();
}

Status TimeableCartesianSplinePath::SetMaxCartesianVelocity(double max_translational_velocity,
                                                            double max_rotational_velocity) {
  if (max_translational_velocity <= 0 || max_rotational_velocity <= 0)
    return InvalidArgumentError("Velocity limits must be positive");
  max_rotational_velocity_ = max_rotational_velocity;
  max_translational_velocity_ = max_translational_velocity;
  return OkStatus();
}

Status TimeableCartesianSplinePath::SetTranslationRounding(double translation_rounding) {
  if (translation_rounding <= 0.0) return InvalidArgumentError("translation_rounding needs to be greater than zero");
  options_.set_translation_rounding(translation_rounding);
  return OkStatus();
}

Status TimeableCartesianSplinePath::SetRotationRounding(double rotation_rounding) {
  if (rotation_rounding <= 0.0) return InvalidArgumentError("rotation_rounding needs to be greater than zero");
  options_.set_rounding(rotation_rounding);
  return OkStatus();
}

bool TimeableCartesianSplinePath::CloseToEnd(double parameter) const {
  return knots_.empty() || parameter >= knots_.back() - kSmall;
}

// timeable_path_cartesian_spline.cc:415-482
Status TimeableCartesianSplinePath::FitSplineToWaypoints() {
  TimeableJointSplinePath::PolyLineToControlPoints(joint_waypoints_, options_.rounding(), &joint_control_points_);
  PolyLineToBspline3Waypoints(pose_waypoints_, options_.translation_rounding(), options_.rounding(),
                              &pose_control_points_);
  const size_t P = joint_control_points_.size();
  const size_t nk = P + kSplineOrder + 1;
  // uniform knots by running accumulation (splines/bspline_base.cc:356-381)
  knots_.assign(nk, 0.0);
  const double spacing = (1.0 / (nk - 2.0 * (kSplineOrder + 1.0) + 1.0)) * (1.0 - 0.0);
  for (size_t i = kSplineOrder + 1; i < nk - kSplineOrder - 1; i++) knots_[i] = knots_[i - 1] + spacing;
  for (size_t i = nk - kSplineOrder - 1; i < nk; i++) knots_[i] = 1.0;
  // knot scaling by the control polygon length: translation + translation, the rotation length is
  // summed but not used (sic, :436-438)
  double translation_length = 0.0;
  for (size_t i = 0; i + 1 < pose_control_points_.size(); i++)
    translation_length += (pose_control_points_[i + 1].translation() - pose_control_points_[i].translation()).norm();
  constexpr double kMinimumFinalKnotValue = 0.1;
  constexpr double kPathParameterPerPolygonLength = 10.0;
  const double weighted_length = std::max(translation_length + translation_length, kMinimumFinalKnotValue);
  for (double &k : knots_) k *= weighted_length * kPathParameterPerPolygonLength;
  packed_translation_.resize(3 * P);
  packed_rotation_.resize(4 * P);
  for (size_t i = 0; i < P; i++) {
    for (int d = 0; d < 3; d++) packed_translation_[3 * i + d] = pose_control_points_[i].translation()[d];
    const Quaterniond &q = pose_control_points_[i].quaternion();
    packed_rotation_[4 * i] = q.w; packed_rotation_[4 * i + 1] = q.x;
    packed_rotation_[4 * i + 2] = q.y; packed_rotation_[4 * i + 3] = q.z;
  }
  return OkStatus();
}

int TimeableCartesianSplinePath::PathIkIndex(const double path_parameter) const {
  return (int)std::round(path_parameter / options_.delta_parameter());
}

double TimeableCartesianSplinePath::PathIkParameter(const int index) const {
  return index * options_.delta_parameter();
}

// timeable_path_cartesian_spline.cc:484-549
Status TimeableCartesianSplinePath::SamplePath(const double path_start) {
  if (knots_.empty()) return FailedPreconditionError("Call SetWaypoints first.");
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  const double delta = options_.delta_parameter();
  const double path_horizon = path_start + delta * (N - 1);
  const int horizon_ik_upper_index = PathIkIndex(path_horizon);
  const int current_ik_upper_index = (int)path_ik_positions_.size() - 1;
  if (horizon_ik_upper_index >= current_ik_upper_index) {
    // Pose and joint targets for the part of the path that has no IK solution yet, re-evaluating
    // the last solved sample (the first sample, parameter 0, twice on the first call).
    const int num_new_samples = horizon_ik_upper_index - current_ik_upper_index + 1;
    sampled_pose_targets_.assign(num_new_samples, Pose3d());
    sampled_joint_targets_.assign(num_new_samples, VectorXd(D));
    // pose targets on the GPU: the translation spline and the quaternion spline at
    // PathIkParameter(i) = i * delta for i = 0 .. horizon (the parameter of sample i is formed as
    // 0 + i * delta there, the reference's product); the ones already solved are dropped
    std::vector<double> poses((size_t)(horizon_ik_upper_index + 1) * 7);
    {
      ::tpamd::EngineLease lease = ::tpamd::acquire_engine();
      if (!lease) return InternalError("no GPU engine");
      const double zero = 0.0;
      const int rc = tpamd_sample_pose_splines_host(lease.get(), 1, horizon_ik_upper_index + 1,
                                                    (int)joint_control_points_.size(), knots_.data(),
                                                    packed_translation_.data(), packed_rotation_.data(), &zero,
                                                    &delta, poses.data());
      if (rc != 0) return InternalError(tpamd_error_string(rc));
    }
    // joint targets on the host (BSplineT::EvalCurve, D values per sample)
    EditableBSpline joint_spline;
    if (Status st = joint_spline.Init(kSplineOrder, (int)knots_.size(), {knots_.data(), knots_.size()},
                                      {joint_control_points_.data(), joint_control_points_.size()});
        !st.ok())
      return st;
    for (int i = current_ik_upper_index; i <= horizon_ik_upper_index; ++i) {
      const int k = std::max(i, 0);
      const double parameter = i < 0 ? 0.0 : PathIkParameter(i);
      const int new_sample_index = i - current_ik_upper_index;
      if (parameter < knots_.back() - delta) {
        if (Status st = joint_spline.EvalCurve(parameter, &sampled_joint_targets_[new_sample_index]); !st.ok())
          return st;
        const double *p = &poses[(size_t)k * 7];
        sampled_pose_targets_[new_sample_index] =
            Pose3d(Quaterniond(p[3], p[4], p[5], p[6]), Vector3d(p[0], p[1], p[2]));
      } else {
        sampled_pose_targets_[new_sample_index] = pose_control_points_.back();
        sampled_joint_targets_[new_sample_index] = joint_control_points_.back();
      }
    }
    const VectorXd initial_value =
        current_ik_upper_index > 0 ? path_ik_positions_[current_ik_upper_index] : sampled_joint_targets_.front();
    new_ik_path_.clear();
    const Status ik_status = path_ik_func_(initial_value, sampled_pose_targets_, sampled_joint_targets_, &new_ik_path_);
    // appended unconditionally from the second sample on (the first matches the initial conditions)
    if (!new_ik_path_.empty())
      path_ik_positions_.insert(path_ik_positions_.end(), new_ik_path_.begin() + 1, new_ik_path_.end());
    if (!ik_status.ok()) return ik_status;
  }
  const int path_start_index = PathIkIndex(path_start);
  if (path_start_index < 0 || horizon_ik_upper_index - path_start_index != (int)N - 1 ||
      horizon_ik_upper_index >= (int)path_ik_positions_.size())
    return InternalError("IK solution does not cover the sampled window");   // the reference CHECKs
  std::copy(path_ik_positions_.begin() + path_start_index, path_ik_positions_.begin() + horizon_ik_upper_index + 1,
            path_position_.begin());
  ComputePathDerivatives(path_position_, delta, &first_path_derivative_, &second_path_derivative_);
  path_state_ = State::kPathWasSampled;
  parameter_start_ = path_start;
  parameter_end_ = path_horizon;
  return OkStatus();
}

// timeable_path_cartesian_spline.cc:551-595
Status TimeableCartesianSplinePath::ConstraintSetup() {
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  const double safety = options_.constraint_safety();
  Matrix6Xd jacobian(6, D);
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
    jacobian.setZero();
    if (Status st = jacobian_func_(path_position_[idx], &jacobian); !st.ok()) return st;
    double vd[6];
    for (int r = 0; r < 6; r++) {
      double acc = 0.0;
      for (size_t dof = 0; dof < D; dof++) acc += jacobian(r, dof) * first_path_derivative_[idx][dof];
      vd[r] = acc;
    }
    const int ct = (int)(2 * D), cr = (int)(2 * D + 1);
    c.a_coefficient(ct) = 0.0;
    c.b_coefficient(ct) = (vd[0] * vd[0] + vd[1] * vd[1]) + vd[2] * vd[2];
    c.upper(ct) = max_translational_velocity_ * max_translational_velocity_;
    c.lower(ct) = -c.upper(ct);
    c.a_coefficient(cr) = 0.0;
    c.b_coefficient(cr) = (vd[3] * vd[3] + vd[4] * vd[4]) + vd[5] * vd[5];
    c.upper(cr) = max_rotational_velocity_ * max_rotational_velocity_;
    c.lower(cr) = -c.upper(cr);
  }
  return OkStatus();
}

Status TimeableCartesianSplinePath::PackSampledWindow(std::vector<double> *q, std::vector<double> *J) const {
  const size_t N = options_.num_path_samples(), D = options_.num_dofs();
  Matrix6Xd jacobian(6, D);
  for (size_t idx = 0; idx < N; idx++) {
    q->insert(q->end(), path_position_[idx].begin(), path_position_[idx].end());
    jacobian.setZero();
    if (Status st = jacobian_func_(path_position_[idx], &jacobian); !st.ok()) return st;
    J->insert(J->end(), jacobian.data(), jacobian.data() + 6 * D);
  }
  return OkStatus();
}

}  // namespace trajectory_planning
