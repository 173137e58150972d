// Host mirror of trajectory_planning/trajectory_planner.h: TrajectoryPlannerOptions
// (:29-55) and the abstract TrajectoryPlanner (:59-160).
#ifndef TPAMD_HOST_TRAJECTORY_PLANNER_H_
#define TPAMD_HOST_TRAJECTORY_PLANNER_H_

#include <memory>
#include <vector>

#include "compat.h"
#include "timeable_path.h"

namespace trajectory_planning {

using ::tpamd::compat::Duration;
using ::tpamd::compat::Time;

template <typename DerivedOptions>
class TrajectoryPlannerOptions {
 public:
  size_t GetNumDofs() const { return num_dofs_; }
  Duration GetTimeStep() const { return time_step_; }
  DerivedOptions &SetNumDofs(size_t n) { num_dofs_ = n; return static_cast<DerivedOptions &>(*this); }
  DerivedOptions &SetTimeStep(Duration d) { time_step_ = d; return static_cast<DerivedOptions &>(*this); }

 protected:
  Duration time_step_;
  size_t num_dofs_ = 0;
};

class TrajectoryPlanner {
 public:
  TrajectoryPlanner() = default;
  virtual ~TrajectoryPlanner() = default;

  void Reset() {
    ResetBase();
    ResetDerived();
  }
  virtual Status Plan(Time start, Duration time_horizon) = 0;
  size_t GetNumTimeSamples() const { return time_.size(); }
  Time GetStartTime() const { return start_time_; }
  Time GetEndTime() const { return end_time_; }
  const std::vector<double> &GetTime() const { return time_; }
  const std::vector<VectorXd> &GetPositions() const { return positions_; }
  const std::vector<VectorXd> &GetVelocities() const { return velocities_; }
  const std::vector<VectorXd> &GetAccelerations() const { return accelerations_; }
  const std::vector<double> &GetPathParameters() const { return path_parameter_; }
  const std::vector<double> &GetPathParameterDerivatives() const { return path_parameter_derivative_; }
  const std::vector<double> &GetSecondPathParameterDerivatives() const {
    return second_path_parameter_derivative_;
  }
  virtual bool IsTrajectoryAtEnd() const {
    const bool path_unchanged = path_ == nullptr ||
                                (path_->GetState() != TimeablePath::State::kModifiedPath &&
                                 path_->GetState() != TimeablePath::State::kNewPath);
    return path_unchanged && target_reached_;
  }
  virtual Status SetPath(std::shared_ptr<TimeablePath> path) = 0;

 protected:
  virtual void ResetDerived() = 0;
  void ResetBase() {
    if (path_ != nullptr) path_->Reset();
    start_time_ = ::tpamd::compat::FromUnixSeconds(0.0);
    end_time_ = ::tpamd::compat::FromUnixSeconds(0.0);
    time_.clear();
    path_parameter_.clear();
    path_parameter_derivative_.clear();
    second_path_parameter_derivative_.clear();
    positions_.clear();
    velocities_.clear();
    accelerations_.clear();
    target_reached_ = false;
  }

  std::shared_ptr<TimeablePath> path_;
  Time start_time_, end_time_;
  std::vector<double> time_, path_parameter_, path_parameter_derivative_,
      second_path_parameter_derivative_;
  std::vector<VectorXd> positions_, velocities_, accelerations_;
  bool target_reached_ = false;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_TRAJECTORY_PLANNER_H_
