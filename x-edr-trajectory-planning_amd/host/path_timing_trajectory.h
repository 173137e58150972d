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
  // Plan() for many planners at once: in every iteration of the
  // receding-horizon loop (path_timing_trajectory.cc:632-660) the planning windows of all
  // planners that still need one are sampled and solved by ONE engine call per group of
  // equal (dofs, samples, control points, safety). Each planner ends in exactly the state
  // Plan(start, time_horizon) would have left it in; the result holds one status per planner.
  static std::vector<Status> PlanBatch(const std::vector<PathTimingTrajectory *> &planners, Time start,
                                       Duration time_horizon);
  // The HIP device this planner's engine calls run on (default: TPAMD_DEVICE, else 0). PlanBatch
  // groups planners by device as well, so planners of different devices are solved side by side.
  void SetDevice(int device) { device_ = device; }
  int GetDevice() const { return device_; }
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
  // One planning window (path_timing_trajectory.cc:307-475) in three phases, so that the
  // engine calls between them can serve many planners at once.
  struct Window {
    int offset = 0;                          // path_samples_offset
    TimeablePath::State old_state = TimeablePath::State::kNoPath;
    double delta = 0.0;
    std::vector<double> q, q1, q2, qd, qdd, t, s, sd, sdd, sd2;
    int32_t last_extremal_index = 0, status = -1;
    double max_time_increment = 0.0;
  };
  Status BeginWindow(Time start, Duration target_duration, Window *w);   // up to the path sampling
  Status ProjectStartVelocity(const Window &w);                          // after sampling, :360-377
  Status SolveWindowOnHost(Window *w);                                   // foreign TimeablePath types
  static void SolveCartesianWindows(const std::vector<PathTimingTrajectory *> &planners,
                                    const std::vector<size_t> &ids, std::vector<Window> *windows,
                                    std::vector<Status> *status);       // TimeableCartesianSplinePath
  Status EndWindow(Window *w);                                           // adopt + append, :418-456
  // Plan() split around its window loop
  Status PlanPrologue(Time start, Duration time_horizon, bool *needs_windows);
  Status PlanEpilogue(Time start);
  static void PlanJointWindowsOnDevice(const std::vector<PathTimingTrajectory *> &planners,
                                       const std::vector<size_t> &ids, Time start, Duration time_horizon,
                                       std::vector<Status> *status);
  void ClampToTimeStepMultiple(Time *time);
  // path_timing_trajectory.h (reference) InterpolationResult, InterpolateAtTime :709-753
  struct InterpolationResult {
    int lower_index = 0;
    VectorXd position, velocity, acceleration;
    double path_parameter = 0, path_parameter_derivative = 0, second_path_parameter_derivative = 0;
  };
  InterpolationResult InterpolateAtTime(double time_sec, int lower_index) const;
  int TimeAtPathSamplesLowerIndex(int starting_index, double time) const;
  void EraseTrajectoryBefore(Time time);   // :540-575, both time sampling methods
  void EraseSamplesUntil(int offset);      // :868-880
  ::tpamd::compat::StatusOr<int> GetTimeOffsetAfter(Time time) const;
  Status ResampleTrajectory(double start_sec);
  Status ResampleEquidistantlyInTime(double start_sec);
  void ResampleSkippingSamplesCloserThanTimeStep(double start_sec);
  double GetMinTimeDeltaToKeep() const { return 0.95 * time_step_sec_; }

  const PathTimingTrajectoryOptions options_;
  const double time_step_sec_;
  int device_ = -1;   // HIP device this planner's engine calls use (-1: the default device)
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
