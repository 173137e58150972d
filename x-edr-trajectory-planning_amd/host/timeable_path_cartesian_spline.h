// Host mirror of trajectory_planning/timeable_path_cartesian_spline.h: CartesianPathOptions
// (:32-60, with the reference's PathIKFunc / JacobianFunc typedefs :35-42) and
// TimeableCartesianSplinePath (:62-192), the TimeablePath a PathTimingTrajectory accepts through
// SetPath for Cartesian-space motions.
//
// Where the work runs: the waypoint fit (O(waypoints)) on the host; the pose targets SamplePath
// evaluates before the IK callback (:484-503) on the GPU (tpamd_sample_pose_splines_host); the
// path-IK and Jacobian callbacks are the user's std::functions and run on the host, as in the
// reference (:508-510, :576); everything after them -- path derivatives (:39-68), the C = 2D + 2
// constraint rows (:551-595), the solver and the planner epilogue -- on the GPU when the path is
// planned through PathTimingTrajectory::Plan / PlanBatch (tpamd_time_cartesian_paths_host, many
// planners per call). SamplePath / ConstraintSetup / GetConstraints also work on their own, for
// callers that drive a TimeOptimalPathProfile themselves.
//
// Not mirrored: SwitchToWaypointPath (:75-78) needs knot insertion into the quaternion spline
// (BSplineQ, out of scope: SURVEY.md section 2); it returns an Unimplemented status.
#ifndef TPAMD_HOST_TIMEABLE_PATH_CARTESIAN_SPLINE_H_
#define TPAMD_HOST_TIMEABLE_PATH_CARTESIAN_SPLINE_H_

#include <functional>
#include <vector>

#include "timeable_path.h"

namespace trajectory_planning {

using ::tpamd::compat::Matrix6Xd;
using ::tpamd::compat::Pose3d;
using ::tpamd::compat::Quaterniond;
using ::tpamd::compat::Vector3d;

class CartesianPathOptions : public PathOptions<CartesianPathOptions> {
 public:
  // (initial_value, pose_targets, joint_targets, ik_result)
  typedef std::function<Status(const VectorXd &, const std::vector<Pose3d> &, const std::vector<VectorXd> &,
                               std::vector<VectorXd> *)>
      PathIKFunc;
  typedef std::function<Status(const VectorXd &, Matrix6Xd *)> JacobianFunc;
  double translation_rounding() const { return translational_rounding_; }
  CartesianPathOptions &set_translation_rounding(double rounding) { translational_rounding_ = rounding; return *this; }
  PathIKFunc GetPathIKFunc() const { return path_ik_func_; }
  CartesianPathOptions &set_path_ik_func(PathIKFunc path_ik) { path_ik_func_ = std::move(path_ik); return *this; }
  JacobianFunc GetJacobianFunc() const { return jacobian_func_; }
  CartesianPathOptions &set_jacobian_func(JacobianFunc jacobian) { jacobian_func_ = std::move(jacobian); return *this; }

 private:
  double translational_rounding_ = 0.05;   // timeable_path_cartesian_spline.h:54
  PathIKFunc path_ik_func_;
  JacobianFunc jacobian_func_;
};

class TimeableCartesianSplinePath : public TimeablePath {
 public:
  // (the reference CHECKs the callbacks and num_path_samples >= 3 in its constructor; here a path
  // built without them reports FailedPrecondition from SetWaypoints / SamplePath)
  explicit TimeableCartesianSplinePath(const CartesianPathOptions &options);

  Status SetWaypoints(Span<const Pose3d> pose_waypoints, Span<const VectorXd> joint_waypoints);
  Status SwitchToWaypointPath(double keep_path_until, Span<const Pose3d> pose_waypoints,
                              Span<const VectorXd> joint_waypoints);
  Status SetMaxCartesianVelocity(double max_translational_velocity, double max_rotational_velocity);
  double GetTranslationRounding() const { return options_.translation_rounding(); }
  double GetRotationRounding() const { return options_.rounding(); }
  Status SetTranslationRounding(double translation_rounding);
  Status SetRotationRounding(double rotation_rounding);

  Status SetMaxJointVelocity(Span<const double> max_velocity) override;
  Status SetMaxJointAcceleration(Span<const double> max_acceleration) override;
  const VectorXd &GetMaxJointVelocity() const override { return max_joint_velocity_; }
  const VectorXd &GetMaxJointAcceleration() const override { return max_joint_acceleration_; }
  Status SetInitialVelocity(Span<const double> velocity) override;
  const VectorXd &GetInitialVelocity() const override { return initial_velocity_; }
  bool CloseToEnd(double parameter) const override;
  State GetState() const override { return path_state_; }
  Status SamplePath(double path_start) override;
  Status ConstraintSetup() override;
  const std::vector<TimeOptimalPathProfile::Constraint> &GetConstraints() const override { return constraints_; }
  size_t NumConstraints() const override { return num_constraints_; }
  size_t NumDofs() const override { return options_.num_dofs(); }
  size_t NumPathSamples() const override { return options_.num_path_samples(); }
  void Reset() override;
  const VectorXd &GetPathStart() const override { return joint_waypoints_.front(); }
  const VectorXd &GetPathEnd() const override { return path_position_.back(); }
  const std::vector<VectorXd> &GetJointWaypoints() const { return joint_waypoints_; }
  const std::vector<Pose3d> &GetPoseWaypoints() const { return pose_waypoints_; }
  double GetParameterStart() const override { return parameter_start_; }
  double GetParameterEnd() const override { return parameter_end_; }
  const VectorXd &GetPathPositionAt(size_t n) const override { return path_position_.at(n); }
  const VectorXd &GetFirstPathDerivativeAt(size_t n) const override { return first_path_derivative_.at(n); }
  const VectorXd &GetSecondPathDerivativeAt(size_t n) const override { return second_path_derivative_.at(n); }
  const std::vector<VectorXd> &GetSplineIKPosition() const { return path_ik_positions_; }
  int GetNumPathSamples() const override { return (int)options_.num_path_samples(); }
  double GetPathSamplingDistance() const override { return options_.delta_parameter(); }
  int PathIkIndex(double path_parameter) const;     // :671-674
  double PathIkParameter(int index) const;          // :676-678

  // Engine-facing accessors (the planner's fused Cartesian call).
  const CartesianPathOptions &options() const { return options_; }
  const std::vector<double> &knots() const { return knots_; }
  double max_translational_velocity() const { return max_translational_velocity_; }
  double max_rotational_velocity() const { return max_rotational_velocity_; }
  // The sampled window as the engine takes it: ik_positions [N][D] and the Jacobian callback's
  // result at every sample [N][6][D] (row-major), appended to q / J.
  Status PackSampledWindow(std::vector<double> *q, std::vector<double> *J) const;

 private:
  Status FitSplineToWaypoints();

  static constexpr int kSplineOrder = 2;
  CartesianPathOptions options_;
  const size_t num_constraints_;
  State path_state_ = State::kNoPath;
  std::vector<VectorXd> joint_waypoints_, joint_control_points_;
  std::vector<Pose3d> pose_waypoints_, pose_control_points_;
  std::vector<double> knots_;
  std::vector<double> packed_translation_, packed_rotation_;   // [P][3], [P][4] (w, x, y, z)
  std::vector<Pose3d> sampled_pose_targets_;
  std::vector<VectorXd> sampled_joint_targets_;
  std::vector<VectorXd> path_position_, first_path_derivative_, second_path_derivative_;
  std::vector<VectorXd> path_ik_positions_;     // the IK solution at 0, delta, 2 delta, ...
  std::vector<VectorXd> new_ik_path_;
  std::vector<TimeOptimalPathProfile::Constraint> constraints_;
  VectorXd max_joint_velocity_, max_joint_acceleration_, initial_velocity_;
  double max_translational_velocity_ = 0.0, max_rotational_velocity_ = 0.0;
  CartesianPathOptions::PathIKFunc path_ik_func_;
  CartesianPathOptions::JacobianFunc jacobian_func_;
  double parameter_start_ = -1.0, parameter_end_ = -1.0;
};

// splines/spline_utils.cc:104-204: corner rounding of a polyline of poses (3W - 2 control poses).
void PolyLineToBspline3Waypoints(const std::vector<Pose3d> &corners, double translation_radius,
                                 double rotational_radius, std::vector<Pose3d> *output);

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_TIMEABLE_PATH_CARTESIAN_SPLINE_H_
