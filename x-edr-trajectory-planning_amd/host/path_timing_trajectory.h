// Host mirror of trajectory_planning/path_timing_trajectory.h: PathTimingTrajectoryOptions
// (:33-77) and PathTimingTrajectory (:91-186). ComputeTimingProfile (one "timing") runs on
// the GPU: fused (spline sampling -> rows -> solve -> epilogue) for a
// TimeableJointSplinePath, SamplePath/ConstraintSetup + rows solve for any other
// TimeablePath. The uniform-in-time resample runs on the GPU as well.
#ifndef TPAMD_HOST_PATH_TIMING_TRAJECTORY_H_
#define TPAMD_HOST_PATH_TIMING_TRAJECTORY_H_

#include <memory>
#include <vector>

#include "time_optimal_path_timing.h"
#include "timeable_path.h"
#include "trajectory_planner.h"

namespace trajectory_planning {

class PathTimingTrajectoryOptions : public TrajectoryPlannerOptions<PathTimingTrajectoryOptions> {
 public:
  enum class TimeSamplingMethod { kUniformlyInTime, kSkipSamplesCloserThanTimeStep };
  size_t GetNumPathSamples() const { return num_path_samples_; }
  PathTimingTrajectoryOptions &SetNumPathSamples(size_t n) { num_path_samples_ = n; return *this; }
  double GetMaxInitialVelocityError() const { return max_initial_velocity_error_; }
  PathTimingTrajectoryOptions &SetMaxInitialVelocityError(double e) { max_initial_velocity_error_ = e; return *this; }
  PathTimingTrajectoryOptions &SetMaxPlanningLoops(int n) { max_planning_iterations_ = n; return *this; }
  int GetMaxPlanningIterations() const { return max_planning_iterations_; }
  PathTimingTrajectoryOptions &SetTimeSamplingMethod(TimeSamplingMethod m) { time_sampling_method_ = m; return *this; }
  TimeSamplingMethod GetTimeSamplingMethod() const { return time_sampling_method_; }

 private:
  // defaults of path_timing_trajectory.h:72-76
  size_t num_path_samples_ = 1000;
  double max_initial_velocity_error_ = 1e-2;
  int max_planning_iterations_ = 200;
  TimeSamplingMethod time_sampling_method_ = TimeSamplingMethod::kUniformlyInTime;
};

class PathTimingTrajectory : public TrajectoryPlanner {
 public:
  explicit PathTimingTrajectory(const PathTimingTrajectoryOptions &options);
  Status Plan(Time start, Duration time_horizon) override;
  size_t NumTimeSamples() const { return time_.size(); }
  Time GetFinalDecelStart() const { return final_decel_start_; }
  Time GetNextPlanStartTime(Time target_time);
  Status SetPath(std::shared_ptr<TimeablePath> path) override;
  void SetProfileDebugVerbosity(int level) { TimeOptimalPathProfile::SetDebugVerbosity(level); }
  const PathTimingTrajectoryOptions &GetOptions() const { return options_; }
  const TimeOptimalPathProfile &GetProfile() const { return profile_; }

 protected:
  void ResetDerived() override;

 private:
  void UpdatePathTrackingStatus();
  Status HandleTimeArguments(Time start);
  Status ComputeTimingProfile(Time start, Duration target_duration);
  void ClampToTimeStepMultiple(Time *time);
  ::tpamd::compat::StatusOr<int> GetTimeOffsetAfter(Time time) const;
  Status ResampleTrajectory(double start_sec);
  Status ResampleEquidistantlyInTime(double start_sec);
  void ResampleSkippingSamplesCloserThanTimeStep(double start_sec);
  double GetMinTimeDeltaToKeep() const { return 0.95 * time_step_sec_; }

  const PathTimingTrajectoryOptions options_;
  const double time_step_sec_;
  Time final_decel_start_;
  double path_horizon_ = 0, path_time_start_ = 0, path_start_ = 0, path_start_velocity_ = 0,
         path_start_acceleration_ = 0;
  TimeOptimalPathProfile profile_;
  bool initial_plan_ = false, planned_to_end_ = false;
  // values at the path samples of all windows planned so far
  std::vector<double> time_at_path_samples_, path_parameter_at_path_samples_,
      path_velocity_at_path_samples_, path_acceleration_at_path_samples_;
  std::vector<double> position_at_path_samples_, velocity_at_path_samples_,
      acceleration_at_path_samples_;  // [samples][D], packed
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_PATH_TIMING_TRAJECTORY_H_
