// Host mirror of trajectory_planning/timeable_path.h: PathOptions (:44-90) and the
// abstract TimeablePath interface (:92-166) the planner drives.
#ifndef TPAMD_HOST_TIMEABLE_PATH_H_
#define TPAMD_HOST_TIMEABLE_PATH_H_

#include <cstddef>
#include <vector>

#include "compat.h"
#include "time_optimal_path_timing.h"

namespace trajectory_planning {

using ::tpamd::compat::Span;
using ::tpamd::compat::Status;
using ::tpamd::compat::VectorXd;

template <typename DerivedOptions>
class PathOptions {
 public:
  double constraint_safety() const { return constraint_safety_; }
  DerivedOptions &set_constraint_safety(double v) { constraint_safety_ = v; return self(); }
  double rounding() const { return rounding_; }
  DerivedOptions &set_rounding(double v) { rounding_ = v; return self(); }
  size_t num_dofs() const { return num_dofs_; }
  DerivedOptions &set_num_dofs(size_t v) { num_dofs_ = v; return self(); }
  size_t num_path_samples() const { return num_path_samples_; }
  DerivedOptions &set_num_path_samples(size_t v) { num_path_samples_ = v; return self(); }
  double delta_parameter() const { return delta_parameter_; }
  DerivedOptions &set_delta_parameter(double v) { delta_parameter_ = v; return self(); }

 private:
  DerivedOptions &self() { return static_cast<DerivedOptions &>(*this); }
  // defaults of timeable_path.h:79-89
  double constraint_safety_ = 0.8;
  double rounding_ = 0.2;
  size_t num_dofs_ = 0;
  size_t num_path_samples_ = 500;
  double delta_parameter_ = 0.005;
};

class TimeablePath {
 public:
  enum class State { kNoPath, kNewPath, kModifiedPath, kPathWasSampled };
  virtual ~TimeablePath() = default;

  virtual Status SetMaxJointVelocity(Span<const double> max_velocity) = 0;
  virtual Status SetMaxJointAcceleration(Span<const double> max_acceleration) = 0;
  virtual const VectorXd &GetMaxJointVelocity() const = 0;
  virtual const VectorXd &GetMaxJointAcceleration() const = 0;
  virtual Status SetInitialVelocity(Span<const double> velocity) = 0;
  virtual const VectorXd &GetInitialVelocity() const = 0;
  virtual bool CloseToEnd(double parameter) const = 0;
  virtual State GetState() const = 0;
  virtual Status SamplePath(double path_start) = 0;
  virtual int GetNumPathSamples() const = 0;
  virtual double GetPathSamplingDistance() const = 0;
  virtual Status ConstraintSetup() = 0;
  virtual const std::vector<TimeOptimalPathProfile::Constraint> &GetConstraints() const = 0;
  virtual size_t NumConstraints() const = 0;
  virtual size_t NumDofs() const = 0;
  virtual size_t NumPathSamples() const = 0;
  virtual void Reset() = 0;
  virtual const VectorXd &GetPathStart() const = 0;
  virtual const VectorXd &GetPathEnd() const = 0;
  virtual double GetParameterStart() const = 0;
  virtual double GetParameterEnd() const = 0;
  virtual const VectorXd &GetPathPositionAt(size_t n) const = 0;
  virtual const VectorXd &GetFirstPathDerivativeAt(size_t n) const = 0;
  virtual const VectorXd &GetSecondPathDerivativeAt(size_t n) const = 0;
};

inline constexpr const char *ToString(const TimeablePath::State state) {
  switch (state) {
    case TimeablePath::State::kNoPath: return "kNoPath";
    case TimeablePath::State::kNewPath: return "kNewPath";
    case TimeablePath::State::kModifiedPath: return "kModifiedPath";
    case TimeablePath::State::kPathWasSampled: return "kPathWasSampled";
  }
  return "Invalid TimeablePath::State enum value";
}

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_TIMEABLE_PATH_H_
