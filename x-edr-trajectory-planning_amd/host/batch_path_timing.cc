#include "batch_path_timing.h"

#include <algorithm>
#include <limits>

#include "engine_handle.h"

namespace trajectory_planning {

using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;

Status BatchPathTiming::SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths) {
  if (paths.empty()) return InvalidArgumentError("no paths");
  for (const auto &p : paths) {
    if (!p) return InvalidArgumentError("null path");
    if (p->NumDofs() != paths[0]->NumDofs() || p->NumPathSamples() != paths[0]->NumPathSamples() ||
        p->num_control_points() != paths[0]->num_control_points() || p->num_control_points() < 3)
      return InvalidArgumentError("paths of one batch must share dofs, samples and control point count");
  }
  paths_ = paths;
  return OkStatus();
}

Status BatchPathTiming::ComputeTimingProfiles(double time_start_sec, BatchTimingResult *r) {
  if (paths_.empty()) return InvalidArgumentError("SetPaths first");
  tpamd_engine *engine = ::tpamd::shared_engine();
  if (!engine) return InternalError("no GPU engine");
  const size_t B = paths_.size(), D = paths_[0]->NumDofs(), N = paths_[0]->NumPathSamples();
  const size_t P = paths_[0]->num_control_points();
  std::vector<double> knots(B * (P + 3)), cps(B * P * D), vmax(B * D), amax(B * D), ps(B, 0.0), dl(B),
      sd0(B, 0.0), sdd0(B, 0.0), t0(B, time_start_sec);
  for (size_t b = 0; b < B; b++) {
    const auto &p = *paths_[b];
    std::copy(p.knots().begin(), p.knots().end(), knots.begin() + b * (P + 3));
    std::copy(p.packed_control_points().begin(), p.packed_control_points().end(), cps.begin() + b * P * D);
    for (size_t d = 0; d < D; d++) {
      vmax[b * D + d] = p.GetMaxJointVelocity()[d];
      amax[b * D + d] = p.GetMaxJointAcceleration()[d];
    }
    dl[b] = p.GetPathSamplingDistance();
  }
  // start velocity: projection of the requested joint velocity on q'(0)
  // (path_timing_trajectory.cc:360-372); needs the first sample only
  {
    std::vector<double> q(B * D), q1(B * D), q2(B * D);
    ::tpamd::EngineGuard guard;
    const int rc = tpamd_sample_joint_paths_host(engine, (int)B, (int)D, 1, (int)P, knots.data(), cps.data(),
                                                 ps.data(), dl.data(), q.data(), q1.data(), q2.data());
    if (rc != 0) return InternalError(tpamd_error_string(rc));
    for (size_t b = 0; b < B; b++) {
      double nrm2 = 0, dot = 0;
      for (size_t d = 0; d < D; d++) {
        nrm2 += q1[b * D + d] * q1[b * D + d];
        dot += paths_[b]->GetInitialVelocity()[d] * q1[b * D + d];
      }
      if (nrm2 > 100 * std::numeric_limits<double>::epsilon()) sd0[b] = std::max(dot / nrm2, 0.0);
    }
  }
  r->num_samples = (int)N; r->num_dofs = (int)D;
  r->status.assign(B, -1); r->last_extremal_index.assign(B, 0);
  r->time.resize(B * N); r->s.resize(B * N); r->sd.resize(B * N); r->sdd.resize(B * N);
  r->q.resize(B * N * D); r->qd.resize(B * N * D); r->qdd.resize(B * N * D);
  tpamd_joint_batch batch{(int)B, (int)D, (int)N, (int)P, 0, 0, paths_[0]->options().constraint_safety()};
  tpamd_joint_inputs in{knots.data(), cps.data(), vmax.data(), amax.data(), ps.data(), dl.data(),
                        sd0.data(), sdd0.data(), t0.data()};
  tpamd_path_outputs out{r->time.data(), r->s.data(), r->sd.data(), r->sdd.data(), r->q.data(),
                         r->qd.data(), r->qdd.data(), r->last_extremal_index.data(), nullptr,
                         r->status.data(), nullptr};
  ::tpamd::EngineGuard guard;
  const int rc = tpamd_time_joint_paths_host(engine, &batch, &in, &out);
  if (rc != 0) return InternalError(tpamd_error_string(rc));
  return OkStatus();
}

}  // namespace trajectory_planning
