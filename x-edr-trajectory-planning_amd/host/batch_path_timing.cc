#include "batch_path_timing.h"

#include <algorithm>
#include <limits>
#include <map>
#include <tuple>

#include "engine_handle.h"

namespace trajectory_planning {

using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;

Status BatchPathTiming::SetPaths(const std::vector<std::shared_ptr<TimeableJointSplinePath>> &paths) {
  if (paths.empty()) return InvalidArgumentError("no paths");
  for (const auto &p : paths) {
    if (!p) return InvalidArgumentError("null path");
    if (p->num_control_points() < 3) return InvalidArgumentError("path without a spline");
  }
  paths_ = paths;
  return OkStatus();
}

namespace {
struct GroupKey {
  size_t dofs, points;
  double safety;
  bool operator<(const GroupKey &o) const {
    return std::tie(dofs, points, safety) < std::tie(o.dofs, o.points, o.safety);
  }
};
}  // namespace

Status BatchPathTiming::ComputeTimingProfiles(double time_start_sec, BatchTimingResult *r) {
  if (paths_.empty()) return InvalidArgumentError("SetPaths first");
  tpamd_engine *engine = ::tpamd::shared_engine();
  if (!engine) return InternalError("no GPU engine");
  const size_t Bt = paths_.size();
  r->status.assign(Bt, -1); r->last_extremal_index.assign(Bt, 0);
  r->samples_per_path.resize(Bt); r->dofs_per_path.resize(Bt);
  r->sample_offset.assign(Bt + 1, 0); r->joint_offset.assign(Bt + 1, 0);
  r->num_samples = 0; r->num_dofs = 0;
  std::map<GroupKey, std::vector<size_t>> groups;
  for (size_t b = 0; b < Bt; b++) {
    const auto &p = *paths_[b];
    const size_t n = p.NumPathSamples(), d = p.NumDofs();
    r->samples_per_path[b] = (int32_t)n; r->dofs_per_path[b] = (int32_t)d;
    r->sample_offset[b + 1] = r->sample_offset[b] + n;
    r->joint_offset[b + 1] = r->joint_offset[b] + n * d;
    r->num_samples = std::max(r->num_samples, (int)n);
    r->num_dofs = std::max(r->num_dofs, (int)d);
    groups[GroupKey{d, (size_t)p.num_control_points(), p.options().constraint_safety()}].push_back(b);
  }
  r->time.resize(r->sample_offset[Bt]); r->s.resize(r->sample_offset[Bt]);
  r->sd.resize(r->sample_offset[Bt]); r->sdd.resize(r->sample_offset[Bt]);
  r->q.resize(r->joint_offset[Bt]); r->qd.resize(r->joint_offset[Bt]); r->qdd.resize(r->joint_offset[Bt]);

  for (const auto &kv : groups) {
    const std::vector<size_t> &ids = kv.second;
    const size_t B = ids.size(), D = kv.first.dofs, P = kv.first.points;
    size_t N = 0;
    bool ragged = false;
    for (size_t b : ids) {
      ragged = ragged || (N != 0 && paths_[b]->NumPathSamples() != N);
      N = std::max(N, paths_[b]->NumPathSamples());
    }
    std::vector<double> knots(B * (P + 3)), cps(B * P * D), vmax(B * D), amax(B * D), ps(B, 0.0), dl(B),
        sd0(B, 0.0), sdd0(B, 0.0), t0(B, time_start_sec);
    std::vector<int32_t> ns(B);
    for (size_t g = 0; g < B; g++) {
      const auto &p = *paths_[ids[g]];
      std::copy(p.knots().begin(), p.knots().end(), knots.begin() + g * (P + 3));
      std::copy(p.packed_control_points().begin(), p.packed_control_points().end(), cps.begin() + g * P * D);
      for (size_t d = 0; d < D; d++) {
        vmax[g * D + d] = p.GetMaxJointVelocity()[d];
        amax[g * D + d] = p.GetMaxJointAcceleration()[d];
      }
      dl[g] = p.GetPathSamplingDistance();
      ns[g] = (int32_t)p.NumPathSamples();
    }
    // start velocity: projection of the requested joint velocity on q'(0)
    // (path_timing_trajectory.cc:360-372); needs the first sample only
    {
      std::vector<double> q(B * D), q1(B * D), q2(B * D);
      ::tpamd::EngineGuard guard;
      const int rc = tpamd_sample_joint_paths_host(engine, (int)B, (int)D, 1, (int)P, knots.data(), cps.data(),
                                                   ps.data(), dl.data(), q.data(), q1.data(), q2.data());
      if (rc != 0) return InternalError(tpamd_error_string(rc));
      for (size_t g = 0; g < B; g++) {
        double nrm2 = 0, dot = 0;
        for (size_t d = 0; d < D; d++) {
          nrm2 += q1[g * D + d] * q1[g * D + d];
          dot += paths_[ids[g]]->GetInitialVelocity()[d] * q1[g * D + d];
        }
        if (nrm2 > 100 * std::numeric_limits<double>::epsilon()) sd0[g] = std::max(dot / nrm2, 0.0);
      }
    }
    std::vector<double> time(B * N), s(B * N), sd(B * N), sdd(B * N), q(B * N * D), qd(B * N * D), qdd(B * N * D);
    std::vector<int32_t> status(B, -1), lei(B, 0);
    tpamd_joint_batch batch{(int)B, (int)D, (int)N, (int)P, 0, 0, kv.first.safety};
    tpamd_joint_inputs in{knots.data(), cps.data(), vmax.data(), amax.data(), ps.data(), dl.data(),
                          sd0.data(), sdd0.data(), t0.data(), ragged ? ns.data() : nullptr};
    tpamd_path_outputs out{time.data(), s.data(), sd.data(), sdd.data(), q.data(), qd.data(), qdd.data(),
                           lei.data(), nullptr, status.data(), nullptr};
    {
      ::tpamd::EngineGuard guard;
      const int rc = tpamd_time_joint_paths_host(engine, &batch, &in, &out);
      if (rc != 0) return InternalError(tpamd_error_string(rc));
    }
    for (size_t g = 0; g < B; g++) {   // padded group layout -> packed result
      const size_t b = ids[g], n = (size_t)ns[g];
      r->status[b] = status[g]; r->last_extremal_index[b] = lei[g];
      if (status[g] != 0) continue;
      std::copy_n(time.begin() + g * N, n, r->time.begin() + r->sample_offset[b]);
      std::copy_n(s.begin() + g * N, n, r->s.begin() + r->sample_offset[b]);
      std::copy_n(sd.begin() + g * N, n, r->sd.begin() + r->sample_offset[b]);
      std::copy_n(sdd.begin() + g * N, n, r->sdd.begin() + r->sample_offset[b]);
      std::copy_n(q.begin() + g * N * D, n * D, r->q.begin() + r->joint_offset[b]);
      std::copy_n(qd.begin() + g * N * D, n * D, r->qd.begin() + r->joint_offset[b]);
      std::copy_n(qdd.begin() + g * N * D, n * D, r->qdd.begin() + r->joint_offset[b]);
    }
  }
  return OkStatus();
}

}  // namespace trajectory_planning
