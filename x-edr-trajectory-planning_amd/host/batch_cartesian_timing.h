// BatchCartesianTiming -- many Cartesian-space paths timed by ONE engine call per group.
//
// A TimeableCartesianSplinePath samples its pose splines, runs the user's path-IK callback
// (timeable_path_cartesian_spline.cc:508-510) and then does arithmetic only:
// ComputePathDerivatives (:39-68), ConstraintSetup (:551-595, which calls the user's
// Jacobian callback per sample, :576), the solver and the planner epilogue. This class
// takes over from the point where the IK solution exists: per path the IK positions
// (path_position_) and the Jacobian callback. The callbacks run on the host, everything
// after them on the GPU (tpamd_time_cartesian_paths_host).
#ifndef TPAMD_HOST_BATCH_CARTESIAN_TIMING_H_
#define TPAMD_HOST_BATCH_CARTESIAN_TIMING_H_

#include <functional>
#include <vector>

#include "batch_path_timing.h"

namespace trajectory_planning {

// Stands where the reference has JacobianFunction
// (timeable_path_cartesian_spline.h:39-42, absl::Status(const VectorXd&, Matrix6Xd*)): the
// 6 x D Jacobian is written row-major into `jacobian`.
using CartesianJacobianFunction = std::function<Status(const VectorXd &joints, double *jacobian)>;

struct CartesianPathSamples {
  std::vector<VectorXd> ik_positions;     // N samples of D joints (path_position_)
  CartesianJacobianFunction jacobian;     // jacobian_func_
  VectorXd max_joint_velocity, max_joint_acceleration;   // D each
  double max_translational_velocity = 0.0, max_rotational_velocity = 0.0;
  double delta_parameter = 0.0;           // CartesianPathOptions::delta_parameter
  double constraint_safety = 0.8;         // CartesianPathOptions::constraint_safety
  double path_start = 0.0;                // SamplePath(path_start)
  double start_velocity = 0.0;            // path_start_velocity_ (sd at the first sample)
};

class BatchCartesianTiming {
 public:
  // Paths may differ in joint count and sample count; they are grouped by
  // (dofs, samples, constraint safety), one engine call per group.
  Status SetPaths(std::vector<CartesianPathSamples> paths);
  // HIP devices that share the batch: contiguous blocks of roughly equal cost (samples x rows^2),
  // one engine from the pool and one host thread per device (the Jacobian callbacks of a block run
  // on its thread: they must be callable concurrently for different paths). Default: the default
  // device alone.
  Status SetDevices(const std::vector<int> &devices);
  // Results in the packed layout of BatchTimingResult; q holds the IK positions.
  Status ComputeTimingProfiles(double time_start_sec, BatchTimingResult *result);

 private:
  Status ComputeBlock(int device, size_t lo, size_t hi, double time_start_sec, BatchTimingResult *r) const;
  std::vector<CartesianPathSamples> paths_;
  std::vector<int> devices_;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_BATCH_CARTESIAN_TIMING_H_
