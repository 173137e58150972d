// Host mirror of trajectory_planning/timeable_path_joint_spline.h:
// TimeableJointSplinePath keeps the reference's interface. The waypoint -> control
// point fit stays on the host (O(W)); SamplePath runs on the GPU
// (tpamd_sample_joint_paths_host), and the planner / BatchPathTiming use the fused
// engine call that never materialises the constraint rows.
#ifndef TPAMD_HOST_TIMEABLE_PATH_JOINT_SPLINE_H_
#define TPAMD_HOST_TIMEABLE_PATH_JOINT_SPLINE_H_

#include <vector>

#include "timeable_path.h"

namespace trajectory_planning {

class JointPathOptions : public PathOptions<JointPathOptions> {};

class TimeableJointSplinePath : public TimeablePath {
 public:
  explicit TimeableJointSplinePath(const JointPathOptions &options);

  Status SetWaypoints(Span<const VectorXd> waypoints);
  // timeable_path_joint_spline.h:40-47 / .cc:209-250: keep the path up to `keep_path_until`,
  // then continue along `waypoints` (the spline is truncated there and extended with the new
  // rounded control polygon; the state becomes kModifiedPath). Host-side edit: the next
  // SamplePath / Plan / batch call picks up the new knots and control points.
  Status SwitchToWaypointPath(double keep_path_until, Span<const VectorXd> waypoints);

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
  const std::vector<TimeOptimalPathProfile::Constraint> &GetConstraints() const override {
    return constraints_;
  }
  size_t NumConstraints() const override { return 2 * options_.num_dofs(); }
  size_t NumDofs() const override { return options_.num_dofs(); }
  size_t NumPathSamples() const override { return options_.num_path_samples(); }
  void Reset() override;
  const VectorXd &GetPathStart() const override { return path_position_.front(); }
  const VectorXd &GetPathEnd() const override { return path_position_.back(); }
  const std::vector<VectorXd> &GetWaypoints() const { return waypoints_; }
  double GetParameterStart() const override { return parameter_start_; }
  double GetParameterEnd() const override { return parameter_end_; }
  const VectorXd &GetPathPositionAt(size_t n) const override { return path_position_.at(n); }
  const VectorXd &GetFirstPathDerivativeAt(size_t n) const override {
    return first_path_derivative_.at(n);
  }
  const VectorXd &GetSecondPathDerivativeAt(size_t n) const override {
    return second_path_derivative_.at(n);
  }
  int GetNumPathSamples() const override { return (int)options_.num_path_samples(); }
  double GetPathSamplingDistance() const override { return options_.delta_parameter(); }

  // Engine-facing accessors (packed inputs of the C-ABI joint form).
  const JointPathOptions &options() const { return options_; }
  const std::vector<double> &knots() const { return knots_; }
  const std::vector<double> &packed_control_points() const { return packed_control_points_; }
  int num_control_points() const { return (int)control_points_.size(); }
  // Installs samples computed by a fused engine call (planner / batch front ends).
  void AdoptSamples(double path_start, const double *q, const double *q1, const double *q2);
  // splines/spline_utils.cc:47-102 (PolyLineToBspline3Waypoints, vector variant): W corners ->
  // 3W - 2 control points with rounded corners. Also used by the Cartesian path's joint spline.
  static void PolyLineToControlPoints(const std::vector<VectorXd> &waypoints, double radius,
                                      std::vector<VectorXd> *control_points);

 private:
  Status FitSplineToWaypoints();
  void PackControlPoints();

  static constexpr int kSplineOrder = 2;
  const JointPathOptions options_;
  State path_state_ = State::kNoPath;
  std::vector<VectorXd> waypoints_, control_points_;
  std::vector<double> packed_control_points_;  // [P][D]
  std::vector<double> knots_;
  std::vector<VectorXd> path_position_, first_path_derivative_, second_path_derivative_;
  std::vector<TimeOptimalPathProfile::Constraint> constraints_;
  VectorXd max_joint_velocity_, max_joint_acceleration_, initial_velocity_;
  double parameter_start_, parameter_end_;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_TIMEABLE_PATH_JOINT_SPLINE_H_
