#include "batch_path_timing.h"

#include <algorithm>
#include <limits>
#include <map>
#include <thread>
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
  size_t dofs, points, bucket;
  double safety;
  bool operator<(const GroupKey &o) const {
    return std::tie(dofs, points, bucket, safety) < std::tie(o.dofs, o.points, o.bucket, o.safety);
  }
};
// host arrays of one group, in the layout of the C-ABI's joint form
struct GroupArrays {
  std::vector<size_t> ids;
  size_t B = 0, D = 0, P = 0, N = 0;
  bool ragged = false;
  double safety = 0.8;
  std::vector<double> knots, cps, vmax, amax, ps, dl, sd0, sdd0, t0;
  std::vector<int32_t> ns;
  std::vector<double> time, s, sd, sdd, q, qd, qdd;
  std::vector<int32_t> status, lei;
};
}  // namespace

Status BatchPathTiming::SetDevices(const std::vector<int> &devices) {
  const int visible = ::tpamd::device_count();
  for (size_t i = 0; i < devices.size(); i++) {
    if (devices[i] < 0 || devices[i] >= visible) return InvalidArgumentError("no such device");
    for (size_t j = 0; j < i; j++)
      if (devices[j] == devices[i]) return InvalidArgumentError("device listed twice");
  }
  devices_ = devices;
  return OkStatus();
}

Status BatchPathTiming::ComputeTimingProfiles(double time_start_sec, BatchTimingResult *r) {
  if (paths_.empty()) return InvalidArgumentError("SetPaths first");
  const size_t Bt = paths_.size();
  r->status.assign(Bt, -1); r->last_extremal_index.assign(Bt, 0);
  r->samples_per_path.resize(Bt); r->dofs_per_path.resize(Bt);
  r->sample_offset.assign(Bt + 1, 0); r->joint_offset.assign(Bt + 1, 0);
  r->num_samples = 0; r->num_dofs = 0;
  std::vector<double> cost(Bt);
  for (size_t b = 0; b < Bt; b++) {
    const auto &p = *paths_[b];
    const size_t n = p.NumPathSamples(), d = p.NumDofs();
    r->samples_per_path[b] = (int32_t)n; r->dofs_per_path[b] = (int32_t)d;
    r->sample_offset[b + 1] = r->sample_offset[b] + n;
    r->joint_offset[b + 1] = r->joint_offset[b] + n * d;
    r->num_samples = std::max(r->num_samples, (int)n);
    r->num_dofs = std::max(r->num_dofs, (int)d);
    cost[b] = (double)n * (2.0 * d) * (2.0 * d);            // samples x rows^2
  }
  r->time.resize(r->sample_offset[Bt]); r->s.resize(r->sample_offset[Bt]);
  r->sd.resize(r->sample_offset[Bt]); r->sdd.resize(r->sample_offset[Bt]);
  r->q.resize(r->joint_offset[Bt]); r->qd.resize(r->joint_offset[Bt]); r->qdd.resize(r->joint_offset[Bt]);

  const std::vector<int> devices = devices_.empty() ? std::vector<int>{::tpamd::default_device()} : devices_;
  const int nd = (int)devices.size();
  std::vector<int32_t> begin(nd + 1);
  tpamd_shard_bounds_balanced((int)Bt, cost.data(), nd, begin.data());
  // one host thread per device; each block writes its own part of the (packed) result
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

Status BatchPathTiming::ComputeBlock(int device, size_t lo, size_t hi, double time_start_sec,
                                     BatchTimingResult *r) const {
  if (hi <= lo) return OkStatus();
  ::tpamd::EngineLease lease = ::tpamd::acquire_engine(device);
  tpamd_engine *engine = lease.get();
  if (!engine) return InternalError("no GPU engine");
  std::map<GroupKey, std::vector<size_t>> keyed;
  for (size_t b = lo; b < hi; b++) {
    const auto &p = *paths_[b];
    const size_t n = p.NumPathSamples();
    const size_t bucket = sample_bucket_ > 0 ? (n + sample_bucket_ - 1) / sample_bucket_ : 0;
    keyed[GroupKey{p.NumDofs(), (size_t)p.num_control_points(), bucket, p.options().constraint_safety()}]
        .push_back(b);
  }
  std::vector<GroupArrays> groups;
  groups.reserve(keyed.size());
  for (const auto &kv : keyed) {
    groups.emplace_back();
    GroupArrays &g = groups.back();
    g.ids = kv.second;
    g.B = g.ids.size(); g.D = kv.first.dofs; g.P = kv.first.points; g.safety = kv.first.safety;
    for (size_t b : g.ids) {
      g.ragged = g.ragged || (g.N != 0 && paths_[b]->NumPathSamples() != g.N);
      g.N = std::max(g.N, paths_[b]->NumPathSamples());
    }
    const size_t B = g.B, D = g.D, P = g.P, N = g.N;
    g.knots.resize(B * (P + 3)); g.cps.resize(B * P * D); g.vmax.resize(B * D); g.amax.resize(B * D);
    g.ps.assign(B, 0.0); g.dl.resize(B); g.sd0.assign(B, 0.0); g.sdd0.assign(B, 0.0); g.t0.assign(B, time_start_sec);
    g.ns.resize(B);
    for (size_t i = 0; i < B; i++) {
      const auto &p = *paths_[g.ids[i]];
      std::copy(p.knots().begin(), p.knots().end(), g.knots.begin() + i * (P + 3));
      std::copy(p.packed_control_points().begin(), p.packed_control_points().end(), g.cps.begin() + i * P * D);
      for (size_t d = 0; d < D; d++) {
        g.vmax[i * D + d] = p.GetMaxJointVelocity()[d];
        g.amax[i * D + d] = p.GetMaxJointAcceleration()[d];
      }
      g.dl[i] = p.GetPathSamplingDistance();
      g.ns[i] = (int32_t)p.NumPathSamples();
    }
    // start velocity: projection of the requested joint velocity on q'(0)
    // (path_timing_trajectory.cc:360-372); needs the first sample only
    {
      std::vector<double> q(B * D), q1(B * D), q2(B * D);
      const int rc = tpamd_sample_joint_paths_host(engine, (int)B, (int)D, 1, (int)P, g.knots.data(), g.cps.data(),
                                                   g.ps.data(), g.dl.data(), q.data(), q1.data(), q2.data());
      if (rc != 0) return InternalError(tpamd_error_string(rc));
      for (size_t i = 0; i < B; i++) {
        double nrm2 = 0, dot = 0;
        for (size_t d = 0; d < D; d++) {
          nrm2 += q1[i * D + d] * q1[i * D + d];
          dot += paths_[g.ids[i]]->GetInitialVelocity()[d] * q1[i * D + d];
        }
        if (nrm2 > 100 * std::numeric_limits<double>::epsilon()) g.sd0[i] = std::max(dot / nrm2, 0.0);
      }
    }
    g.time.resize(B * N); g.s.resize(B * N); g.sd.resize(B * N); g.sdd.resize(B * N);
    g.q.resize(B * N * D); g.qd.resize(B * N * D); g.qdd.resize(B * N * D);
    g.status.assign(B, -1); g.lei.assign(B, 0);
  }
  // all groups of this device in ONE call: they run side by side on the engine's lanes
  std::vector<tpamd_joint_batch> batches;
  std::vector<tpamd_joint_inputs> ins;
  std::vector<tpamd_path_outputs> outs;
  for (GroupArrays &g : groups) {
    batches.push_back(tpamd_joint_batch{(int)g.B, (int)g.D, (int)g.N, (int)g.P, 0, 0, g.safety});
    ins.push_back(tpamd_joint_inputs{g.knots.data(), g.cps.data(), g.vmax.data(), g.amax.data(), g.ps.data(),
                                     g.dl.data(), g.sd0.data(), g.sdd0.data(), g.t0.data(),
                                     g.ragged ? g.ns.data() : nullptr});
    outs.push_back(tpamd_path_outputs{g.time.data(), g.s.data(), g.sd.data(), g.sdd.data(), g.q.data(),
                                      g.qd.data(), g.qdd.data(), g.lei.data(), nullptr, g.status.data(), nullptr});
  }
  {
    const int rc = tpamd_time_joint_groups_host(engine, (int)groups.size(), batches.data(), ins.data(), outs.data());
    if (rc != 0) return InternalError(tpamd_error_string(rc));
  }
  for (const GroupArrays &g : groups) {     // padded group layout -> packed result
    const size_t N = g.N, D = g.D;
    for (size_t i = 0; i < g.B; i++) {
      const size_t b = g.ids[i], n = (size_t)g.ns[i];
      r->status[b] = g.status[i]; r->last_extremal_index[b] = g.lei[i];
      if (g.status[i] != 0) continue;
      std::copy_n(g.time.begin() + i * N, n, r->time.begin() + r->sample_offset[b]);
      std::copy_n(g.s.begin() + i * N, n, r->s.begin() + r->sample_offset[b]);
      std::copy_n(g.sd.begin() + i * N, n, r->sd.begin() + r->sample_offset[b]);
      std::copy_n(g.sdd.begin() + i * N, n, r->sdd.begin() + r->sample_offset[b]);
      std::copy_n(g.q.begin() + i * N * D, n * D, r->q.begin() + r->joint_offset[b]);
      std::copy_n(g.qd.begin() + i * N * D, n * D, r->qd.begin() + r->joint_offset[b]);
      std::copy_n(g.qdd.begin() + i * N * D, n * D, r->qdd.begin() + r->joint_offset[b]);
    }
  }
  return OkStatus();
}

}  // namespace trajectory_planning
