#include "batch_cartesian_timing.h"

#include <algorithm>
#include <map>
#include <thread>
#include <tuple>

#include "engine_handle.h"

namespace trajectory_planning {

using ::tpamd::compat::InternalError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;

Status BatchCartesianTiming::SetPaths(std::vector<CartesianPathSamples> paths) {
  if (paths.empty()) return InvalidArgumentError("no paths");
  for (const auto &p : paths) {
    if (p.ik_positions.size() < 3) return InvalidArgumentError("a path needs at least 3 samples");
    const size_t D = p.ik_positions[0].size();
    if (D == 0 || p.max_joint_velocity.size() != D || p.max_joint_acceleration.size() != D)
      return InvalidArgumentError("limits must have one entry per joint");
    for (const auto &q : p.ik_positions)
      if (q.size() != D) return InvalidArgumentError("IK positions of one path must share a size");
    if (!p.jacobian) return InvalidArgumentError("missing Jacobian callback");
    if (!(p.delta_parameter > 0.0)) return InvalidArgumentError("delta_parameter must be positive");
  }
  paths_ = std::move(paths);
  return OkStatus();
}

Status BatchCartesianTiming::SetDevices(const std::vector<int> &devices) {
  const int visible = ::tpamd::device_count();
  for (size_t i = 0; i < devices.size(); i++) {
    if (devices[i] < 0 || devices[i] >= visible) return InvalidArgumentError("no such device");
    for (size_t j = 0; j < i; j++)
      if (devices[j] == devices[i]) return InvalidArgumentError("device listed twice");
  }
  devices_ = devices;
  return OkStatus();
}

Status BatchCartesianTiming::ComputeTimingProfiles(double time_start_sec, BatchTimingResult *r) {
  if (paths_.empty()) return InvalidArgumentError("SetPaths first");
  const size_t Bt = paths_.size();
  r->status.assign(Bt, -1); r->last_extremal_index.assign(Bt, 0);
  r->samples_per_path.resize(Bt); r->dofs_per_path.resize(Bt);
  r->sample_offset.assign(Bt + 1, 0); r->joint_offset.assign(Bt + 1, 0);
  r->num_samples = 0; r->num_dofs = 0;
  std::vector<double> cost(Bt);
  for (size_t b = 0; b < Bt; b++) {
    const size_t n = paths_[b].ik_positions.size(), d = paths_[b].ik_positions[0].size();
    r->samples_per_path[b] = (int32_t)n; r->dofs_per_path[b] = (int32_t)d;
    r->sample_offset[b + 1] = r->sample_offset[b] + n;
    r->joint_offset[b + 1] = r->joint_offset[b] + n * d;
    r->num_samples = std::max(r->num_samples, (int)n);
    r->num_dofs = std::max(r->num_dofs, (int)d);
    cost[b] = (double)n * (2.0 * d + 2.0) * (2.0 * d + 2.0);
  }
  r->time.resize(r->sample_offset[Bt]); r->s.resize(r->sample_offset[Bt]);
  r->sd.resize(r->sample_offset[Bt]); r->sdd.resize(r->sample_offset[Bt]);
  r->q.resize(r->joint_offset[Bt]); r->qd.resize(r->joint_offset[Bt]); r->qdd.resize(r->joint_offset[Bt]);
  const std::vector<int> devices = devices_.empty() ? std::vector<int>{::tpamd::default_device()} : devices_;
  const int nd = (int)devices.size();
  std::vector<int32_t> begin(nd + 1);
  tpamd_shard_bounds_balanced((int)Bt, cost.data(), nd, begin.data());
  std::vector<Status> st(nd, OkStatus());
  std::vector<std::thread> threads;
  for (int k = 1; k < nd; k++)
    threads.emplace_back([&, k] { st[k] = ComputeBlock(devices[k], begin[k], begin[k + 1], time_start_sec, r); });
  st[0] = ComputeBlock(devices[0], begin[0], begin[1], time_start_sec, r);
  for (auto &t : threads) t.join();
  for (const Status &s : st)
    if (!s.ok()) return s;
  return OkStatus();
}

Status BatchCartesianTiming::ComputeBlock(int device, size_t lo, size_t hi, double time_start_sec,
                                          BatchTimingResult *r) const {
  if (hi <= lo) return OkStatus();
  ::tpamd::EngineLease lease = ::tpamd::acquire_engine(device);
  tpamd_engine *engine = lease.get();
  if (!engine) return InternalError("no GPU engine");
  std::map<std::tuple<size_t, size_t, double>, std::vector<size_t>> groups;
  for (size_t b = lo; b < hi; b++)
    groups[std::make_tuple(paths_[b].ik_positions[0].size(), paths_[b].ik_positions.size(),
                           paths_[b].constraint_safety)].push_back(b);

  for (const auto &kv : groups) {
    const std::vector<size_t> &ids = kv.second;
    const size_t B = ids.size(), D = std::get<0>(kv.first), N = std::get<1>(kv.first);
    std::vector<double> q(B * N * D), J(B * N * 6 * D), vmax(B * D), amax(B * D), vt(B), vr(B), ps(B),
        dl(B), sd0(B), sdd0(B, 0.0), t0(B, time_start_sec);
    for (size_t g = 0; g < B; g++) {
      const CartesianPathSamples &p = paths_[ids[g]];
      for (size_t i = 0; i < N; i++) {
        std::copy(p.ik_positions[i].begin(), p.ik_positions[i].end(), q.begin() + (g * N + i) * D);
        // the user's callback, once per sample, as ConstraintSetup does (:576)
        const Status st = p.jacobian(p.ik_positions[i], &J[(g * N + i) * 6 * D]);
        if (!st.ok()) return st;
      }
      for (size_t d = 0; d < D; d++) {
        vmax[g * D + d] = p.max_joint_velocity[d];
        amax[g * D + d] = p.max_joint_acceleration[d];
      }
      vt[g] = p.max_translational_velocity; vr[g] = p.max_rotational_velocity;
      ps[g] = p.path_start; dl[g] = p.delta_parameter; sd0[g] = p.start_velocity;
    }
    std::vector<double> time(B * N), s(B * N), sd(B * N), sdd(B * N), qd(B * N * D), qdd(B * N * D);
    std::vector<int32_t> status(B, -1), lei(B, 0);
    tpamd_cartesian_batch batch{(int)B, (int)D, (int)N, 0, std::get<2>(kv.first)};
    tpamd_cartesian_inputs in{q.data(), J.data(), vmax.data(), amax.data(), vt.data(), vr.data(),
                              ps.data(), dl.data(), sd0.data(), sdd0.data(), t0.data()};
    tpamd_path_outputs out{time.data(), s.data(), sd.data(), sdd.data(), nullptr, qd.data(), qdd.data(),
                           lei.data(), nullptr, status.data(), nullptr};
    {
      const int rc = tpamd_time_cartesian_paths_host(engine, &batch, &in, &out);
      if (rc != 0) return InternalError(tpamd_error_string(rc));
    }
    for (size_t g = 0; g < B; g++) {
      const size_t b = ids[g];
      r->status[b] = status[g]; r->last_extremal_index[b] = lei[g];
      std::copy_n(q.begin() + g * N * D, N * D, r->q.begin() + r->joint_offset[b]);
      if (status[g] != 0) continue;
      std::copy_n(time.begin() + g * N, N, r->time.begin() + r->sample_offset[b]);
      std::copy_n(s.begin() + g * N, N, r->s.begin() + r->sample_offset[b]);
      std::copy_n(sd.begin() + g * N, N, r->sd.begin() + r->sample_offset[b]);
      std::copy_n(sdd.begin() + g * N, N, r->sdd.begin() + r->sample_offset[b]);
      std::copy_n(qd.begin() + g * N * D, N * D, r->qd.begin() + r->joint_offset[b]);
      std::copy_n(qdd.begin() + g * N * D, N * D, r->qdd.begin() + r->joint_offset[b]);
    }
  }
  return OkStatus();
}

}  // namespace trajectory_planning
