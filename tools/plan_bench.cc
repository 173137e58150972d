// Receding-horizon planning throughput (run ON the GPU box through tools/gpu_plan_bench.py):
// B planners with 7-joint paths of 10 random waypoints, N = 1000 path samples, 4 ms time step,
// 750 ms horizon, replanning every 200 ms until every planner has reached the end of its path --
// the settings of the reference's planner tests (path_timing_trajectory_test.cc:62-66). Three ways:
//   set     PathTimingTrajectorySet: planner state resident on the device
//   batch   PathTimingTrajectory::PlanBatch: histories travel up and down on every call
//   oracle  the CPU restatement of Plan, one planner per OpenMP thread (the checker; timed as the
//           CPU baseline, as bench.py does for the single-window solve)
// One JSON line with Plan calls per second for each, the PCIe bytes per call of the set, and
// whether the set's final trajectories equal the oracle's bit for bit.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../oracle/tp_oracle.h"
#include "../x-edr-trajectory-planning_amd/host/path_timing_trajectory.h"
#include "../x-edr-trajectory-planning_amd/host/path_timing_trajectory_set.h"

using namespace trajectory_planning;
using tpamd::compat::FromUnixNanos;
using tpamd::compat::Milliseconds;

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? std::atoi(argv[1]) : 1024;
  const int D = argc > 2 ? std::atoi(argv[2]) : 7;
  const int N = 1000, W = 10, P = 3 * W - 2;
  const int threads = argc > 3 ? std::atoi(argv[3]) : 16;
  const bool run_batch = argc > 4 ? std::atoi(argv[4]) != 0 : true;
  const double windows_per_path = argc > 5 ? std::atof(argv[5]) : 3.0;
  const int64_t kMs = 1000000;
  unsigned long long seed = 20240607;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
  PathTimingTrajectoryOptions opt;
  opt.SetNumDofs(D).SetNumPathSamples(N).SetTimeStep(Milliseconds(4));
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths(B);
  for (int b = 0; b < B; b++) {
    std::vector<VectorXd> wps;
    for (int i = 0; i < W; i++) { VectorXd v(D); for (int d = 0; d < D; d++) v[d] = 4.0 * rnd() - 2.0; wps.push_back(v); }
    auto probe = std::make_shared<TimeableJointSplinePath>(JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
    probe->SetWaypoints({wps.data(), wps.size()});
    // a window covers 1 / windows_per_path of the path
    const double delta = probe->knots().back() / (windows_per_path * (N - 1));
    paths[b] = std::make_shared<TimeableJointSplinePath>(
        JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta));
    std::vector<double> vmax(D), amax(D);
    for (int d = 0; d < D; d++) { vmax[d] = 1.0 + rnd(); amax[d] = 2.0 + 2.0 * rnd(); }
    paths[b]->SetMaxJointVelocity({vmax.data(), vmax.size()});
    paths[b]->SetMaxJointAcceleration({amax.data(), amax.size()});
    paths[b]->SetWaypoints({wps.data(), wps.size()});
  }
  // ---- device-resident set
  PathTimingTrajectorySet set(opt, B, P);
  if (!set.status().ok()) { std::printf("{\"error\": \"%s\"}\n", set.status().ToString().c_str()); return 1; }
  set.SetPaths(paths);
  { // warm-up on a throw-away set (kernel load, workspace growth)
    PathTimingTrajectorySet warm(opt, B, P);
    warm.SetPaths(paths);
    warm.Plan(FromUnixNanos(0), Milliseconds(750));
  }
  int calls_set = 0;
  size_t max_bytes = 0;
  long long windows_set = 0;
  std::vector<std::vector<int64_t>> starts_of_call;     // replayed on the oracle
  double t0 = now();
  for (int64_t start = 0;; start += 200 * kMs) {
    std::vector<tpamd::compat::Time> starts(B);
    for (int b = 0; b < B; b++) starts[b] = set.GetNumTimeSamples(b) ? set.GetNextPlanStartTime(b, FromUnixNanos(start)) : FromUnixNanos(start);
    set.Plan(starts, std::vector<tpamd::compat::Duration>(B, Milliseconds(750)));
    starts_of_call.emplace_back(B);
    for (int b = 0; b < B; b++) starts_of_call.back()[b] = tpamd::compat::ToUnixNanos(starts[b]);
    calls_set++;
    max_bytes = std::max(max_bytes, set.LastPlanBytesOverPcie());
    bool all = true;
    for (int b = 0; b < B; b++) { all = all && set.IsTrajectoryAtEnd(b); windows_set += set.WindowsOfLastPlan(b); }
    if (all || calls_set > 500) break;
  }
  const double t_set = now() - t0;
  // ---- PlanBatch with host-side planner state
  double t_batch = 0.0;
  int calls_batch = 0;
  if (run_batch) {
    std::vector<std::unique_ptr<PathTimingTrajectory>> planners;
    std::vector<PathTimingTrajectory *> ptrs;
    std::vector<std::shared_ptr<TimeableJointSplinePath>> paths2(B);
    for (int b = 0; b < B; b++) {
      paths2[b] = std::make_shared<TimeableJointSplinePath>(paths[b]->options());
      paths2[b]->SetMaxJointVelocity({paths[b]->GetMaxJointVelocity().data(), (size_t)D});
      paths2[b]->SetMaxJointAcceleration({paths[b]->GetMaxJointAcceleration().data(), (size_t)D});
      paths2[b]->SetWaypoints({paths[b]->GetWaypoints().data(), paths[b]->GetWaypoints().size()});
      planners.push_back(std::make_unique<PathTimingTrajectory>(opt));
      planners.back()->SetPath(paths2[b]);
      ptrs.push_back(planners.back().get());
    }
    t0 = now();
    for (int64_t start = 0;; start += 200 * kMs) {
      // (one start time for all, as PlanBatch takes it: clamp to the earliest end time)
      int64_t s = start;
      for (int b = 0; b < B; b++)
        if (planners[b]->GetNumTimeSamples()) s = std::min<int64_t>(s, tpamd::compat::ToUnixNanos(planners[b]->GetEndTime()));
      PathTimingTrajectory::PlanBatch(ptrs, FromUnixNanos(s), Milliseconds(750));
      calls_batch++;
      bool all = true;
      for (int b = 0; b < B; b++) all = all && planners[b]->IsTrajectoryAtEnd();
      if (all || calls_batch > 60) break;     // (bounded: this path moves ~200 MB per call)
    }
    t_batch = now() - t0;
  }
  // ---- CPU oracle, one planner per thread
  std::vector<tpo_planner *> oracles(B);
  for (int b = 0; b < B; b++) {
    oracles[b] = tpo_planner_create(D, N, paths[b]->GetPathSamplingDistance(), 0.8, 4 * kMs, 0, 200, 1e-2);
    tpo_planner_set_limits(oracles[b], paths[b]->GetMaxJointVelocity().data(), paths[b]->GetMaxJointAcceleration().data());
    tpo_planner_set_spline(oracles[b], paths[b]->knots().data(), (int)paths[b]->knots().size(),
                           paths[b]->packed_control_points().data(), paths[b]->num_control_points(), TPO_PATH_NEW);
  }
  // a bounded sample of the same workload: the first `sample` planners, the very Plan calls the
  // set made for them (same start times)
  const int sample = std::min(B, 8 * threads);
  long long calls_oracle = 0;
  t0 = now();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+ : calls_oracle)
  for (int b = 0; b < sample; b++) {
    for (int n = 0; n < calls_set; n++) {
      tpo_planner_plan(oracles[b], starts_of_call[n][b], 750 * kMs);
      calls_oracle++;
    }
  }
  const double t_oracle = now() - t0;
  // the set's final trajectories against the oracle's (sample planners)
  bool same = true;
  for (int b = 0; b < sample; b++) {
    PlannedTrajectory tr;
    set.GetTrajectory(b, &tr);
    const int M = tpo_planner_num_samples(oracles[b]);
    if ((int)tr.time.size() != M) { same = false; continue; }
    for (int i = 0; i < M; i++) same = same && tr.time[i] == tpo_planner_time(oracles[b])[i];
    for (int i = 0; i < M * D; i++) same = same && tr.positions[i] == tpo_planner_positions(oracles[b])[i] &&
                                           tr.accelerations[i] == tpo_planner_accelerations(oracles[b])[i];
  }
  std::printf("{\"planners\": %d, \"dofs\": %d, \"path_samples\": %d, \"time_step_ms\": 4, \"horizon_ms\": 750, "
              "\"replan_every_ms\": 200, \"windows_per_path\": %.1f, \"set_plan_calls\": %d, \"set_windows\": %lld, \"set_seconds\": %.4f, "
              "\"set_planner_plans_per_s\": %.1f, \"set_ms_per_plan_call\": %.3f, \"set_pcie_bytes_per_plan_call\": %zu, "
              "\"set_device_MB\": %.1f, \"batch_plan_calls\": %d, \"batch_seconds\": %.4f, "
              "\"batch_planner_plans_per_s\": %.1f, \"oracle_threads\": %d, \"oracle_sample_planners\": %d, "
              "\"oracle_seconds\": %.4f, \"oracle_planner_plans_per_s\": %.1f, "
              "\"set_equals_oracle_on_sample\": %s}\n",
              B, D, N, windows_per_path, calls_set, windows_set, t_set, (double)B * calls_set / t_set, 1e3 * t_set / calls_set, max_bytes,
              set.DeviceBytes() / 1e6, calls_batch, t_batch, calls_batch ? (double)B * calls_batch / t_batch : 0.0, threads,
              sample, t_oracle, (double)calls_oracle / t_oracle, same ? "true" : "false");
  return 0;
}
