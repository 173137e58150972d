// Host mirror of the reference's TimeOptimalPathProfile
// (trajectory_planning/time_optimal_path_timing.h:36-357): same class name, nested
// Constraint, method names, argument meaning and bool/Status conventions; the solve
// itself (InitSolver/SetupProblem/OptimizePathParameter) is one call into the MI355X
// engine with a batch of one. Array types come from compat.h.
#ifndef TPAMD_HOST_TIME_OPTIMAL_PATH_TIMING_H_
#define TPAMD_HOST_TIME_OPTIMAL_PATH_TIMING_H_

#include <cstdint>
#include <limits>
#include <vector>

#include "compat.h"

namespace trajectory_planning {

namespace absl_like = ::tpamd::compat;

class TimeOptimalPathProfile {
 public:
  using Scalar = double;  // fp64 only, as the reference (time_optimal_path_timing.h:38-41)
  using ArrayX = ::tpamd::compat::ArrayXd;

  enum DebugVerbosity { kNoOutput = 0, kMainAlgorithm, kExtremalLoop, kExtremalControl,
                        kExtremalDetail, kAll };

  // Rows lower <= A*sdd + B*sd^2 <= upper of one path sample
  // (time_optimal_path_timing.h:65-102). Stored column-wise.
  class Constraint {
   public:
    void resize(int size) {
      size_ = size;
      data_.assign(4 * (size_t)size, 0.0);
    }
    int size() const { return size_; }
    Scalar &a_coefficient(int i) { return data_[i]; }
    const Scalar &a_coefficient(int i) const { return data_[i]; }
    Scalar &b_coefficient(int i) { return data_[size_ + i]; }
    const Scalar &b_coefficient(int i) const { return data_[size_ + i]; }
    Scalar &lower(int i) { return data_[2 * size_ + i]; }
    const Scalar &lower(int i) const { return data_[2 * size_ + i]; }
    Scalar &upper(int i) { return data_[3 * size_ + i]; }
    const Scalar &upper(int i) const { return data_[3 * size_ + i]; }
    // column access (contiguous, size() entries)
    Scalar *a_coefficient() { return data_.data(); }
    const Scalar *a_coefficient() const { return data_.data(); }
    Scalar *b_coefficient() { return data_.data() + size_; }
    const Scalar *b_coefficient() const { return data_.data() + size_; }
    Scalar *lower() { return data_.data() + 2 * size_; }
    const Scalar *lower() const { return data_.data() + 2 * size_; }
    Scalar *upper() { return data_.data() + 3 * size_; }
    const Scalar *upper() const { return data_.data() + 3 * size_; }

   private:
    int size_ = 0;
    std::vector<Scalar> data_;
  };

  TimeOptimalPathProfile() = default;
  ~TimeOptimalPathProfile() = default;

  // Host-only no-op: the engine has no debug text output.
  static void SetDebugVerbosity(int level);

  bool InitSolver(int num_samples, int num_constraints);
  bool SetupProblem(const std::vector<Constraint> &constraints, Scalar s_start, Scalar s_end,
                    Scalar sd_start, Scalar sdd_start, Scalar time_start);
  void SetMaxNumSolverLoops(int num_loops);
  bool OptimizePathParameter();

  const ArrayX &GetTimeSamples() const { return time_; }
  const ArrayX &GetPathParameter() const { return s_; }
  const ArrayX &GetPathVelocity() const { return sd_; }
  const ArrayX &GetPathAcceleration() const { return sdd_; }

  bool GetPathParameterAndDerivatives(Scalar t, Scalar *s, Scalar *sd, Scalar *sdd) const;
  Scalar GetTotalDuration() const { return time_[time_.size() - 1] - time_[0]; }
  Scalar GetEndTime() const { return time_[time_.size() - 1]; }
  Scalar GetStartTime() const { return time_[0]; }
  bool GetPreviousDiscreteValues(Scalar t, Scalar *sk, Scalar *sdk, Scalar *sddk,
                                 Scalar *tk) const;
  int GetPreviousIndex(Scalar t) const;
  Scalar GetMaxTimeIncrement() const;
  // The reference's cross-check of the LP (time_optimal_path_timing.cc:1010-1103, public "for
  // testability" at .h:198-201; its only callers are tests): every pairwise intersection of
  // row bounds, validated against all rows. O(C^3) for one sample on the host -- a checking
  // tool next to the GPU LP below, never on the solve path.
  void FindMaxSd2BruteForce(const Constraint &constr, Scalar *sd2max, Scalar *sddmax,
                            Scalar *sd2zero) const;
  // One LP on the GPU (FindMaxSd2Simplex, time_optimal_path_timing.cc:1149-1363).
  void FindMaxSd2Simplex(const Constraint &constr, Scalar *sd2max, Scalar *sddmax,
                         Scalar *sd2zero);
  int GetLastExtremalIndex() const { return last_extremal_index_; }
  ::tpamd::compat::Status SolutionSatisfiesConstraints();

  // Engine status of the last solve (TPAMD_PATH_*), 0 when solved.
  int last_status() const { return last_status_; }

  // For PathTimingTrajectory: install a solution computed by a fused engine call.
  void AdoptSolution(int num_samples, int num_constraints, Scalar s_start, Scalar s_end,
                     const Scalar *time, const Scalar *s, const Scalar *sd, const Scalar *sdd,
                     const Scalar *sd2, int last_extremal_index, Scalar dt_max);

 private:
  enum SolverState { kInvalidState = 0, kAllocated, kProblemDefined, kProblemSolved };
  static constexpr Scalar kTiny = std::numeric_limits<Scalar>::epsilon() * 1e5;
  int SampleIndexFromTime(Scalar t) const;
  void FinishSolvedState();

  SolverState solver_state_ = kInvalidState;
  int num_constraints_ = 0, num_samples_ = 0;
  Scalar s_start_ = 0, s_end_ = 0, sd_start_ = 0, sdd_start_ = 0, time_start_ = 0;
  Scalar ds_ = 0, inv_ds_ = 0, dt_max_ = 0;
  int max_num_loops_ = 100;  // time_optimal_path_timing.h:339
  int low_idx_ = 0, high_idx_ = 0, last_extremal_index_ = 0, last_status_ = 0;
  std::vector<Scalar> rows_a_, rows_b_, rows_lo_, rows_hi_;  // [N][C]
  ArrayX time_, s_, sd_, sdd_, sd2_;
};

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_TIME_OPTIMAL_PATH_TIMING_H_
