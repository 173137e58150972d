// GPU test of the host mirror of the reference's C++ API (run by tests/test_gpu_host_api.py).
// The oracle is linked here only as the checker.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../oracle/tp_oracle.h"
#include "../../x-edr-trajectory-planning_amd/host/batch_cartesian_timing.h"
#include "../../x-edr-trajectory-planning_amd/host/batch_path_timing.h"
#include "../../x-edr-trajectory-planning_amd/host/path_timing_trajectory.h"
#include "../../x-edr-trajectory-planning_amd/host/path_timing_trajectory_set.h"
#include "../../x-edr-trajectory-planning_amd/host/time_optimal_path_timing.h"
#include "../../x-edr-trajectory-planning_amd/host/timeable_path_cartesian_spline.h"
#include "../../x-edr-trajectory-planning_amd/host/timeable_path_joint_spline.h"

using namespace trajectory_planning;
using tpamd::compat::FromUnixSeconds;
using tpamd::compat::Milliseconds;
using tpamd::compat::Seconds;

static int g_fail = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
  } while (0)

using Constraint = TimeOptimalPathProfile::Constraint;
constexpr double kTiny = 2.220446049250313e-16 * 1e5;

// circle scenario of the reference's solver test (time_optimal_path_timing_test.cc:125-158)
static std::vector<Constraint> CircleRows(int n, double s0, double s1, double R, double vmax, double amax) {
  std::vector<Constraint> c(n);
  const double ds = (s1 - s0) / (n - 1);
  for (int i = 0; i < n; i++) {
    const double s = i * ds + s0;
    c[i].resize(4);
    c[i].a_coefficient(0) = -R * std::sin(s); c[i].a_coefficient(1) = R * std::cos(s);
    c[i].a_coefficient(2) = 0; c[i].a_coefficient(3) = 0;
    c[i].b_coefficient(0) = -R * std::cos(s); c[i].b_coefficient(1) = -R * std::sin(s);
    c[i].b_coefficient(2) = std::pow(R * std::sin(s), 2); c[i].b_coefficient(3) = std::pow(R * std::cos(s), 2);
    c[i].upper(0) = amax; c[i].upper(1) = amax; c[i].upper(2) = vmax * vmax; c[i].upper(3) = vmax * vmax;
    c[i].lower(0) = -amax; c[i].lower(1) = -amax; c[i].lower(2) = 0; c[i].lower(3) = 0;
  }
  return c;
}

static void TestProfileAgainstOracle() {
  for (int n : {50, 51}) {
    for (double sd0 : {0.0, 0.1}) {
      auto rows = CircleRows(n, 0.0, M_PI, 2.0, 1.2, 1.0);
      TimeOptimalPathProfile opt;
      CHECK(opt.InitSolver(n, 4));
      CHECK(opt.SetupProblem(rows, 0.0, M_PI, sd0, 0.0, 0.0));
      CHECK(opt.OptimizePathParameter());
      CHECK(opt.SolutionSatisfiesConstraints().ok());
      double s, sd, sdd;
      CHECK(opt.GetPathParameterAndDerivatives(0.0, &s, &sd, &sdd));
      CHECK(sd == sd0);   // time_optimal_path_timing_test.cc:538-541
      // oracle on the same rows
      std::vector<double> A(n * 4), B(n * 4), lo(n * 4), hi(n * 4);
      for (int i = 0; i < n; i++)
        for (int c = 0; c < 4; c++) {
          A[i * 4 + c] = rows[i].a_coefficient(c); B[i * 4 + c] = rows[i].b_coefficient(c);
          lo[i * 4 + c] = rows[i].lower(c); hi[i * 4 + c] = rows[i].upper(c);
        }
      tpo_profile *p = tpo_profile_create(n, 4);
      CHECK(tpo_profile_setup(p, A.data(), B.data(), lo.data(), hi.data(), 0.0, M_PI, sd0, 0.0, 0.0) == 0);
      CHECK(tpo_profile_optimize(p) == 0);
      for (int i = 0; i < n; i++) {
        CHECK(opt.GetTimeSamples()[i] == tpo_profile_time(p)[i]);
        CHECK(opt.GetPathParameter()[i] == tpo_profile_s(p)[i]);
        CHECK(opt.GetPathVelocity()[i] == tpo_profile_sd(p)[i]);
        CHECK(opt.GetPathAcceleration()[i] == tpo_profile_sdd(p)[i]);
      }
      CHECK(opt.GetLastExtremalIndex() == tpo_profile_last_extremal_index(p));
      CHECK(opt.GetMaxTimeIncrement() == tpo_profile_max_time_increment(p));
      const double T = opt.GetEndTime();
      for (int k = -2; k <= 102; k++) {
        const double t = T * k / 100.0;
        double os, osd, osdd;
        CHECK(opt.GetPathParameterAndDerivatives(t, &s, &sd, &sdd));
        CHECK(tpo_profile_query(p, t, &os, &osd, &osdd) == 1);
        CHECK(s == os && sd == osd && sdd == osdd);
        CHECK(opt.GetPreviousIndex(t) == tpo_profile_previous_index(p, t));
      }
      tpo_profile_destroy(p);
    }
  }
  // failure modes keep the reference's bool convention
  auto rows = CircleRows(20, 0.0, M_PI, 2.0, 1.2, 1.0);
  TimeOptimalPathProfile opt;
  CHECK(!opt.OptimizePathParameter());                 // not set up
  CHECK(opt.InitSolver(20, 4));
  CHECK(!opt.SetupProblem(rows, M_PI, M_PI, 0, 0, 0));  // s_start >= s_end
  CHECK(!opt.SetupProblem(rows, 0, M_PI, -1, 0, 0));    // sd_start < 0
  rows[3].upper(0) = rows[3].lower(0);
  CHECK(!opt.SetupProblem(rows, 0, M_PI, 0, 0, 0));     // lower >= upper
  // one LP through the class API
  Constraint c;
  c.resize(2);
  c.a_coefficient(0) = 1; c.b_coefficient(0) = 0; c.lower(0) = -1; c.upper(0) = 1;
  c.a_coefficient(1) = 0; c.b_coefficient(1) = 1; c.lower(1) = 0; c.upper(1) = 4;
  double sd2max, sddmax, sd2zero;
  opt.FindMaxSd2Simplex(c, &sd2max, &sddmax, &sd2zero);
  CHECK(sd2max == 4.0 && sd2zero == 4.0);
}

static std::shared_ptr<TimeableJointSplinePath> MakePath(int N, const std::vector<VectorXd> &wps,
                                                         double vmax, double amax, double *delta_out) {
  const size_t D = wps[0].size();
  // first fit with a provisional option set to learn the spline length, as a user would
  // pick delta_parameter from the path length
  auto probe = std::make_shared<TimeableJointSplinePath>(
      JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
  probe->SetWaypoints({wps.data(), wps.size()});
  const double delta = probe->knots().back() / (N - 1);
  auto path = std::make_shared<TimeableJointSplinePath>(
      JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta));
  std::vector<double> v(D, vmax), a(D, amax);
  CHECK(path->SetMaxJointVelocity({v.data(), v.size()}).ok());
  CHECK(path->SetMaxJointAcceleration({a.data(), a.size()}).ok());
  CHECK(path->SetWaypoints({wps.data(), wps.size()}).ok());
  if (delta_out) *delta_out = delta;
  return path;
}

static void TestJointPathAndPlanner() {
  // path_timing_trajectory_test.cc:112-173: waypoints (1,2,3), (-1,-2,-3), (1,2,3)
  const std::vector<VectorXd> wps = {VectorXd{1, 2, 3}, VectorXd{-1, -2, -3}, VectorXd{1, 2, 3}};
  const int N = 1000;
  double delta = 0;
  auto path = MakePath(N, wps, 1.0, 2.0, &delta);
  CHECK(path->GetState() == TimeablePath::State::kNewPath);
  // fit and sampling against the oracle
  std::vector<double> w(9), cps(7 * 3), knots(10);
  for (int i = 0; i < 3; i++) for (int d = 0; d < 3; d++) w[i * 3 + d] = wps[i][d];
  CHECK(tpo_joint_fit_spline(w.data(), 3, 3, 0.2, cps.data(), knots.data()) == 7);
  CHECK(path->num_control_points() == 7);
  for (int k = 0; k < 10; k++) CHECK(path->knots()[k] == knots[k]);
  for (int k = 0; k < 21; k++) CHECK(path->packed_control_points()[k] == cps[k]);
  CHECK(path->SamplePath(0.0).ok());
  CHECK(path->ConstraintSetup().ok());
  std::vector<double> q(N * 3), q1(N * 3), q2(N * 3);
  CHECK(tpo_joint_sample_path(knots.data(), 10, cps.data(), 7, 3, 0.0, delta, N, q.data(), q1.data(), q2.data()) == 0);
  for (int i = 0; i < N; i++)
    for (int d = 0; d < 3; d++) {
      CHECK(path->GetPathPositionAt(i)[d] == q[i * 3 + d]);
      CHECK(path->GetFirstPathDerivativeAt(i)[d] == q1[i * 3 + d]);
      CHECK(path->GetSecondPathDerivativeAt(i)[d] == q2[i * 3 + d]);
    }
  CHECK(path->GetConstraints()[10].a_coefficient(1) == q1[10 * 3 + 1]);
  CHECK(path->GetConstraints()[10].b_coefficient(3 + 1) == q1[10 * 3 + 1] * q1[10 * 3 + 1]);
  CHECK(path->GetConstraints()[10].upper(3) == (1.0 * 0.8) * (1.0 * 0.8));
  CHECK(path->GetParameterEnd() == N * delta);   // sic, quirk Q7

  for (auto method : {PathTimingTrajectoryOptions::TimeSamplingMethod::kUniformlyInTime,
                      PathTimingTrajectoryOptions::TimeSamplingMethod::kSkipSamplesCloserThanTimeStep}) {
    auto opts = PathTimingTrajectoryOptions().SetNumDofs(3).SetNumPathSamples(N).SetTimeStep(Milliseconds(4))
                    .SetTimeSamplingMethod(method);
    auto run = [&](double t0, std::vector<double> *times, std::vector<VectorXd> *pos) {
      auto p = MakePath(N, wps, 1.0, 2.0, nullptr);
      PathTimingTrajectory planner(opts);
      CHECK(planner.SetPath(p).ok());
      const auto st = planner.Plan(FromUnixSeconds(t0), Seconds(1000.0));
      if (!st.ok()) std::printf("plan: %s\n", st.ToString().c_str());
      CHECK(st.ok());
      CHECK(planner.IsTrajectoryAtEnd());
      CHECK(planner.GetNumTimeSamples() > 100);
      // end position = last waypoint, end velocity = 0 (path_timing_trajectory_test.cc:167-172)
      for (int d = 0; d < 3; d++) {
        CHECK(std::fabs(planner.GetPositions().back()[d] - wps.back()[d]) < 1e-10);
        CHECK(planner.GetVelocities().back()[d] == 0.0);
      }
      for (const auto &v : planner.GetVelocities()) CHECK(v.maxAbs() <= 1.0 * 0.8 + 1e-9);
      for (const auto &a : planner.GetAccelerations()) CHECK(a.maxAbs() <= 2.0 + 1e-12);
      *times = planner.GetTime();
      *pos = planner.GetPositions();
      // Reset + replan reproduces the trajectory exactly (:533-545)
      planner.Reset();
      CHECK(p->GetState() == TimeablePath::State::kNoPath);
      CHECK(p->SetWaypoints({wps.data(), wps.size()}).ok());
      CHECK(planner.Plan(FromUnixSeconds(t0), Seconds(1000.0)).ok());
      CHECK(planner.GetTime() == *times);
      CHECK(planner.GetPositions().size() == pos->size());
      for (size_t i = 0; i < pos->size(); i++) CHECK(planner.GetPositions()[i] == (*pos)[i]);
    };
    std::vector<double> t_a, t_b;
    std::vector<VectorXd> p_a, p_b;
    run(0.0, &t_a, &p_a);
    run(100.0, &t_b, &p_b);
    // invariance to the start time within 1e-10 (:254-296)
    CHECK(t_a.size() == t_b.size());
    for (size_t i = 0; i < std::min(t_a.size(), t_b.size()); i++) {
      CHECK(std::fabs((t_b[i] - 100.0) - t_a[i]) < 1e-9);
      for (int d = 0; d < 3; d++) CHECK(std::fabs(p_a[i][d] - p_b[i][d]) < 1e-10);
    }
    if (method == PathTimingTrajectoryOptions::TimeSamplingMethod::kUniformlyInTime)
      for (size_t i = 1; i < t_a.size(); i++) CHECK(std::fabs(t_a[i] - t_a[i - 1] - 0.004) < 1e-12);
  }
  // mismatched path is rejected (path_timing_trajectory.cc:850-861)
  PathTimingTrajectory planner(PathTimingTrajectoryOptions().SetNumDofs(4).SetNumPathSamples(N).SetTimeStep(Milliseconds(4)));
  CHECK(!planner.SetPath(path).ok());
  CHECK(!planner.Plan(FromUnixSeconds(0), Seconds(1)).ok());
}

static void TestBatch() {
  const int B = 24, D = 7, N = 500, W = 6;
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths;
  unsigned long long seed = 12345;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
  std::vector<double> deltas(B);
  for (int b = 0; b < B; b++) {
    std::vector<VectorXd> wps;
    for (int i = 0; i < W; i++) {
      VectorXd v(D);
      for (int d = 0; d < D; d++) v[d] = 4.0 * rnd() - 2.0;
      wps.push_back(v);
    }
    paths.push_back(MakePath(N, wps, 1.0 + rnd(), 2.0 + 2.0 * rnd(), &deltas[b]));
  }
  BatchPathTiming batch;
  CHECK(batch.SetPaths(paths).ok());
  BatchTimingResult r;
  CHECK(batch.ComputeTimingProfiles(1.5, &r).ok());
  const int P = 3 * W - 2;
  for (int b = 0; b < B; b++) {
    CHECK(r.status[b] == 0);
    std::vector<double> t(N), s(N), sd(N), sdd(N), q(N * D), qd(N * D), qdd(N * D);
    int lei = 0;
    const int rc = tpo_time_joint_path(paths[b]->knots().data(), P + 3, paths[b]->packed_control_points().data(),
                                       P, D, paths[b]->GetMaxJointVelocity().data(),
                                       paths[b]->GetMaxJointAcceleration().data(), 0.8, 0.0, deltas[b], N,
                                       0.0, 0.0, 1.5, nullptr, t.data(), s.data(), sd.data(), sdd.data(),
                                       q.data(), qd.data(), qdd.data(), &lei);
    CHECK(rc == 0);
    CHECK(lei == r.last_extremal_index[b]);
    for (int i = 0; i < N; i++) {
      CHECK(r.time[(size_t)b * N + i] == t[i]);
      CHECK(r.sd[(size_t)b * N + i] == sd[i]);
      CHECK(r.sdd[(size_t)b * N + i] == sdd[i]);
    }
    for (int i = 0; i < N * D; i++) {
      CHECK(r.q[(size_t)b * N * D + i] == q[i]);
      CHECK(r.qd[(size_t)b * N * D + i] == qd[i]);
      CHECK(r.qdd[(size_t)b * N * D + i] == qdd[i]);
    }
  }
}

// BASELINE.json configs[4] in miniature: 6/7/14 DOF, different sample and waypoint counts in one
// SetPaths call; every path must equal its own single-path oracle run bit for bit.
static void TestMixedBatch() {
  const int B = 30;
  const int dofs[3] = {6, 7, 14};
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths;
  unsigned long long seed = 777;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
  std::vector<double> deltas(B);
  std::vector<int> Ns(B), Ws(B), Ds(B);
  for (int b = 0; b < B; b++) {
    Ds[b] = dofs[b % 3];
    Ns[b] = 100 + (int)(rnd() * 500);
    Ws[b] = (b % 5 == 0) ? 4 : 6;
    std::vector<VectorXd> wps;
    for (int i = 0; i < Ws[b]; i++) {
      VectorXd v(Ds[b]);
      for (int d = 0; d < Ds[b]; d++) v[d] = 4.0 * rnd() - 2.0;
      wps.push_back(v);
    }
    paths.push_back(MakePath(Ns[b], wps, 1.0 + rnd(), 2.0 + 2.0 * rnd(), &deltas[b]));
  }
  BatchPathTiming batch;
  CHECK(batch.SetPaths(paths).ok());
  BatchTimingResult r;
  CHECK(batch.ComputeTimingProfiles(0.25, &r).ok());
  CHECK(r.num_dofs == 14);
  for (int b = 0; b < B; b++) {
    const int N = Ns[b], D = Ds[b], P = 3 * Ws[b] - 2;
    CHECK(r.samples_per_path[b] == N && r.dofs_per_path[b] == D);
    CHECK(r.sample_offset[b + 1] - r.sample_offset[b] == (size_t)N);
    std::vector<double> t(N), s(N), sd(N), sdd(N), q(N * D), qd(N * D), qdd(N * D);
    int lei = 0;
    const int rc = tpo_time_joint_path(paths[b]->knots().data(), P + 3, paths[b]->packed_control_points().data(),
                                       P, D, paths[b]->GetMaxJointVelocity().data(),
                                       paths[b]->GetMaxJointAcceleration().data(), 0.8, 0.0, deltas[b], N,
                                       0.0, 0.0, 0.25, nullptr, t.data(), s.data(), sd.data(), sdd.data(),
                                       q.data(), qd.data(), qdd.data(), &lei);
    CHECK(rc == r.status[b]);
    if (rc != 0) continue;
    CHECK(lei == r.last_extremal_index[b]);
    const size_t so = r.sample_offset[b], jo = r.joint_offset[b];
    for (int i = 0; i < N; i++) {
      CHECK(r.time[so + i] == t[i]);
      CHECK(r.s[so + i] == s[i]);
      CHECK(r.sd[so + i] == sd[i]);
      CHECK(r.sdd[so + i] == sdd[i]);
    }
    for (int i = 0; i < N * D; i++) {
      CHECK(r.q[jo + i] == q[i]);
      CHECK(r.qd[jo + i] == qd[i]);
      CHECK(r.qdd[jo + i] == qdd[i]);
    }
  }
}

// Cartesian batch: joint-spline samples stand in for the IK solution, the Jacobian callback
// is a smooth function of q with a unit block (the reference's tests use test doubles for both,
// path_timing_trajectory_test.cc:552-587). Every path must equal the oracle's Cartesian path.
static void TestCartesianBatch() {
  const int B = 10;
  unsigned long long seed = 4242;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
  std::vector<CartesianPathSamples> paths(B);
  for (int b = 0; b < B; b++) {
    const int D = (b % 2) ? 7 : 6, N = 200 + 50 * (b % 3), W = 5;
    std::vector<VectorXd> wps;
    for (int i = 0; i < W; i++) {
      VectorXd v(D);
      for (int d = 0; d < D; d++) v[d] = 4.0 * rnd() - 2.0;
      wps.push_back(v);
    }
    double delta = 0;
    auto jp = MakePath(N, wps, 1.0, 2.0, &delta);
    CHECK(jp->SamplePath(0.0).ok());
    CartesianPathSamples &p = paths[b];
    for (int i = 0; i < N; i++) p.ik_positions.push_back(jp->GetPathPositionAt(i));
    p.jacobian = [D](const VectorXd &q, double *J) {
      for (int r = 0; r < 6; r++)
        for (int d = 0; d < D; d++)
          J[r * D + d] = 0.25 * std::sin(q[d] * (r + 1.0) + 0.37 * d) + ((r == d % 6) ? 1.0 : 0.0);
      return ::tpamd::compat::OkStatus();
    };
    p.max_joint_velocity = VectorXd(D, 1.0 + rnd());
    p.max_joint_acceleration = VectorXd(D, 2.0 + 2.0 * rnd());
    p.max_translational_velocity = 0.6 + 0.9 * rnd();
    p.max_rotational_velocity = 0.8 + 1.2 * rnd();
    p.delta_parameter = delta;
  }
  const std::vector<CartesianPathSamples> copy = paths;
  BatchCartesianTiming batch;
  CHECK(batch.SetPaths(paths).ok());
  BatchTimingResult r;
  CHECK(batch.ComputeTimingProfiles(2.0, &r).ok());
  for (int b = 0; b < B; b++) {
    const CartesianPathSamples &p = copy[b];
    const int N = (int)p.ik_positions.size(), D = (int)p.ik_positions[0].size();
    std::vector<double> q(N * D), J(N * 6 * D), t(N), s(N), sd(N), sdd(N), qd(N * D), qdd(N * D);
    for (int i = 0; i < N; i++) {
      for (int d = 0; d < D; d++) q[i * D + d] = p.ik_positions[i][d];
      CHECK(p.jacobian(p.ik_positions[i], &J[i * 6 * D]).ok());
    }
    int lei = 0;
    const int rc = tpo_time_cartesian_path(q.data(), J.data(), N, D, p.max_joint_velocity.data(),
                                           p.max_joint_acceleration.data(), p.max_translational_velocity,
                                           p.max_rotational_velocity, 0.8, 0.0, p.delta_parameter, 0.0, 0.0, 2.0,
                                           t.data(), s.data(), sd.data(), sdd.data(), qd.data(), qdd.data(), &lei);
    CHECK(rc == r.status[b]);
    if (rc != 0) continue;
    CHECK(lei == r.last_extremal_index[b]);
    const size_t so = r.sample_offset[b], jo = r.joint_offset[b];
    for (int i = 0; i < N; i++) {
      CHECK(r.time[so + i] == t[i]);
      CHECK(r.sd[so + i] == sd[i]);
      CHECK(r.sdd[so + i] == sdd[i]);
    }
    for (int i = 0; i < N * D; i++) {
      CHECK(r.q[jo + i] == q[i]);
      CHECK(r.qd[jo + i] == qd[i]);
      CHECK(r.qdd[jo + i] == qdd[i]);
    }
  }
}

// PlanBatch: several planners advance through their receding-horizon windows together
// (one engine call per iteration and shape group); each must end exactly where its own
// Plan() would have ended.
static void TestPlanBatch() {
  const int K = 6;
  unsigned long long seed = 99;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
  struct Spec { int D, N; std::vector<VectorXd> wps; double vmax, amax; };
  std::vector<Spec> specs(K);
  for (int k = 0; k < K; k++) {
    Spec &sp = specs[k];
    sp.D = (k == K - 1) ? 5 : 7;                 // the last planner is alone in its shape group
    sp.N = (k == K - 1) ? 300 : 400;
    const int W = 4 + (k % 2) * 3;               // long paths need several windows
    for (int i = 0; i < W; i++) {
      VectorXd v(sp.D);
      for (int d = 0; d < sp.D; d++) v[d] = 6.0 * rnd() - 3.0;
      sp.wps.push_back(v);
    }
    sp.vmax = 1.0 + rnd(); sp.amax = 2.0 + 2.0 * rnd();
  }
  auto make = [&](const Spec &sp, std::shared_ptr<TimeableJointSplinePath> *path_out) {
    // a short sampling distance: one window covers only a part of the path
    auto probe = std::make_shared<TimeableJointSplinePath>(
        JointPathOptions().set_num_dofs(sp.D).set_num_path_samples(sp.N));
    probe->SetWaypoints({sp.wps.data(), sp.wps.size()});
    const double delta = 0.4 * probe->knots().back() / (sp.N - 1);
    auto path = std::make_shared<TimeableJointSplinePath>(
        JointPathOptions().set_num_dofs(sp.D).set_num_path_samples(sp.N).set_delta_parameter(delta));
    std::vector<double> v(sp.D, sp.vmax), a(sp.D, sp.amax);
    CHECK(path->SetMaxJointVelocity({v.data(), v.size()}).ok());
    CHECK(path->SetMaxJointAcceleration({a.data(), a.size()}).ok());
    CHECK(path->SetWaypoints({sp.wps.data(), sp.wps.size()}).ok());
    *path_out = path;
    auto planner = std::make_unique<PathTimingTrajectory>(
        PathTimingTrajectoryOptions().SetNumDofs(sp.D).SetNumPathSamples(sp.N).SetTimeStep(Milliseconds(4)));
    CHECK(planner->SetPath(path).ok());
    return planner;
  };
  std::vector<std::shared_ptr<TimeableJointSplinePath>> paths_a(K), paths_b(K);
  std::vector<std::unique_ptr<PathTimingTrajectory>> batch, single;
  std::vector<PathTimingTrajectory *> ptrs;
  for (int k = 0; k < K; k++) {
    batch.push_back(make(specs[k], &paths_a[k]));
    single.push_back(make(specs[k], &paths_b[k]));
    ptrs.push_back(batch.back().get());
  }
  const auto st = PathTimingTrajectory::PlanBatch(ptrs, FromUnixSeconds(10.0), Seconds(1000.0));
  for (int k = 0; k < K; k++) {
    CHECK(st[k].ok());
    CHECK(single[k]->Plan(FromUnixSeconds(10.0), Seconds(1000.0)).ok());
    CHECK(batch[k]->IsTrajectoryAtEnd());
    CHECK(batch[k]->GetTime() == single[k]->GetTime());
    CHECK(batch[k]->GetTime().size() > 50);
    CHECK(batch[k]->GetPositions().size() == single[k]->GetPositions().size());
    for (size_t i = 0; i < std::min(batch[k]->GetPositions().size(), single[k]->GetPositions().size()); i++) {
      CHECK(batch[k]->GetPositions()[i] == single[k]->GetPositions()[i]);
      CHECK(batch[k]->GetVelocities()[i] == single[k]->GetVelocities()[i]);
      CHECK(batch[k]->GetAccelerations()[i] == single[k]->GetAccelerations()[i]);
    }
    // the end of the path is reached with zero velocity
    for (int d = 0; d < specs[k].D; d++) {
      CHECK(std::fabs(batch[k]->GetPositions().back()[d] - specs[k].wps.back()[d]) < 1e-9);
      CHECK(batch[k]->GetVelocities().back()[d] == 0.0);
    }
  }
  // a short horizon: planners stop after the horizon is covered and can be continued
  std::vector<std::unique_ptr<PathTimingTrajectory>> b2, s2;
  std::vector<std::shared_ptr<TimeableJointSplinePath>> pa(K), pb(K);
  std::vector<PathTimingTrajectory *> p2;
  for (int k = 0; k < K; k++) { b2.push_back(make(specs[k], &pa[k])); s2.push_back(make(specs[k], &pb[k])); p2.push_back(b2.back().get()); }
  const auto st2 = PathTimingTrajectory::PlanBatch(p2, FromUnixSeconds(0.0), Seconds(0.5));
  for (int k = 0; k < K; k++) {
    CHECK(st2[k].ok());
    CHECK(s2[k]->Plan(FromUnixSeconds(0.0), Seconds(0.5)).ok());
    CHECK(b2[k]->GetTime() == s2[k]->GetTime());
    CHECK(b2[k]->IsTrajectoryAtEnd() == s2[k]->IsTrajectoryAtEnd());
    CHECK(b2[k]->GetFinalDecelStart() == s2[k]->GetFinalDecelStart());
  }
  const auto st3 = PathTimingTrajectory::PlanBatch(p2, FromUnixSeconds(0.2), Seconds(1000.0));
  for (int k = 0; k < K; k++) {
    CHECK(st3[k].ok());
    CHECK(s2[k]->Plan(FromUnixSeconds(0.2), Seconds(1000.0)).ok());
    CHECK(b2[k]->GetTime() == s2[k]->GetTime());
    CHECK(b2[k]->IsTrajectoryAtEnd());
  }
}

// Plan / PlanBatch against the oracle's restatement of PathTimingTrajectory::Plan
// (oracle/tp_oracle_plan.c): same windows, same resampled trajectory, same bookkeeping, bit for
// bit -- for the reference's planner fixture (path_timing_trajectory_test.cc:112-173: three
// waypoints, 1000 path samples at the default sampling distance so that the path needs several
// windows, 4 ms, 750 ms horizon, a replan every 200 ms) with both time sampling methods
// (including the "already planned enough" erase of path_timing_trajectory.cc:540-575), and for
// random multi-window 7-joint planners advanced together by PlanBatch.
static void ComparePlannerWithOracle(const PathTimingTrajectory &pl, const tpo_planner *o, int D) {
  const int M = tpo_planner_num_samples(o);
  CHECK((int)pl.GetTime().size() == M);
  if ((int)pl.GetTime().size() != M) return;
  const double *t = tpo_planner_time(o), *q = tpo_planner_positions(o), *v = tpo_planner_velocities(o),
               *a = tpo_planner_accelerations(o), *s = tpo_planner_path_parameter(o);
  int bad = 0;
  for (int i = 0; i < M; i++) {
    if (pl.GetTime()[i] != t[i]) bad++;
    if (pl.GetPathParameters()[i] != s[i]) bad++;
    for (int d = 0; d < D; d++) {
      if (pl.GetPositions()[i][d] != q[(size_t)i * D + d]) bad++;
      if (pl.GetVelocities()[i][d] != v[(size_t)i * D + d]) bad++;
      if (pl.GetAccelerations()[i][d] != a[(size_t)i * D + d]) bad++;
    }
  }
  CHECK(bad == 0);
  CHECK(tpamd::compat::ToUnixNanos(pl.GetEndTime()) == tpo_planner_end_time(o));
  CHECK(tpamd::compat::ToUnixNanos(pl.GetFinalDecelStart()) == tpo_planner_final_decel_start(o));
  CHECK(pl.IsTrajectoryAtEnd() == (tpo_planner_target_reached(o) != 0));
}

static void TestPlanAgainstOracle() {
  using Method = PathTimingTrajectoryOptions::TimeSamplingMethod;
  const int64_t kMs = 1000000;
  for (Method method : {Method::kUniformlyInTime, Method::kSkipSamplesCloserThanTimeStep}) {
    const int D = 3, N = 1000;
    auto path = std::make_shared<TimeableJointSplinePath>(JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
    PathTimingTrajectory planner(PathTimingTrajectoryOptions().SetTimeStep(Milliseconds(4)).SetNumDofs(D)
                                     .SetNumPathSamples(N).SetTimeSamplingMethod(method));
    CHECK(planner.SetPath(path).ok());
    std::vector<VectorXd> wps;
    for (double sgn : {1.0, -1.0, 1.0}) { VectorXd w(D); w[0] = sgn; w[1] = 2 * sgn; w[2] = 3 * sgn; wps.push_back(w); }
    CHECK(path->SetWaypoints({wps.data(), wps.size()}).ok());
    std::vector<double> vmax(D, 1.0), amax(D, 2.0);
    CHECK(path->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
    CHECK(path->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
    tpo_planner *o = tpo_planner_create(D, N, path->options().delta_parameter(), path->options().constraint_safety(),
                                        4 * kMs, method == Method::kUniformlyInTime ? 0 : 1, 10000, 1e-3);
    tpo_planner_set_limits(o, vmax.data(), amax.data());
    tpo_planner_set_spline(o, path->knots().data(), (int)path->knots().size(), path->packed_control_points().data(),
                           (int)path->num_control_points(), TPO_PATH_NEW);
    int64_t start = 0;
    int loops = 0, windows = 0, erase_only = 0;
    while (!planner.IsTrajectoryAtEnd() && loops < 200) {
      const bool ok = planner.Plan(tpamd::compat::FromUnixNanos(start), Milliseconds(750)).ok();
      const int rc = tpo_planner_plan(o, start, 750 * kMs);
      CHECK(ok == (rc == TPO_PLAN_OK));
      if (!ok || rc != TPO_PLAN_OK) break;
      ComparePlannerWithOracle(planner, o, D);
      windows += tpo_planner_windows(o);
      erase_only += tpo_planner_windows(o) == 0;
      start = std::min<int64_t>(tpo_planner_end_time(o), start + 200 * kMs);
      loops++;
    }
    CHECK(planner.IsTrajectoryAtEnd());
    CHECK(loops > 5 && windows >= 3);
    for (int d = 0; d < D; d++) {
      CHECK(planner.GetVelocities().back()[d] == 0.0);
      CHECK(std::fabs(planner.GetPositions().back()[d] - wps.back()[d]) < 1e-9);
    }
    // the whole path in one call, then a later start well inside the planned horizon: the
    // "already planned enough" branch only erases (for kSkip: interpolated first sample)
    auto path2 = std::make_shared<TimeableJointSplinePath>(JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
    PathTimingTrajectory planner2(PathTimingTrajectoryOptions().SetTimeStep(Milliseconds(4)).SetNumDofs(D)
                                      .SetNumPathSamples(N).SetTimeSamplingMethod(method));
    CHECK(planner2.SetPath(path2).ok());
    CHECK(path2->SetWaypoints({wps.data(), wps.size()}).ok());
    CHECK(path2->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
    CHECK(path2->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
    tpo_planner *o2 = tpo_planner_create(D, N, path2->options().delta_parameter(), path2->options().constraint_safety(),
                                         4 * kMs, method == Method::kUniformlyInTime ? 0 : 1, 10000, 1e-3);
    tpo_planner_set_limits(o2, vmax.data(), amax.data());
    tpo_planner_set_spline(o2, path2->knots().data(), (int)path2->knots().size(),
                           path2->packed_control_points().data(), (int)path2->num_control_points(), TPO_PATH_NEW);
    CHECK(planner2.Plan(tpamd::compat::FromUnixNanos(0), Seconds(1.0e6)).ok());
    CHECK(tpo_planner_plan(o2, 0, (int64_t)1000000 * 1000 * kMs) == TPO_PLAN_OK);
    ComparePlannerWithOracle(planner2, o2, D);
    for (int64_t st : {(int64_t)501 * kMs, (int64_t)1337 * kMs + 12345}) {
      CHECK(planner2.Plan(tpamd::compat::FromUnixNanos(st), Milliseconds(100)).ok());
      CHECK(tpo_planner_plan(o2, st, 100 * kMs) == TPO_PLAN_OK);
      CHECK(tpo_planner_windows(o2) == 0);
      ComparePlannerWithOracle(planner2, o2, D);
      if (method == Method::kSkipSamplesCloserThanTimeStep)   // first sample interpolated at the start time
        CHECK(planner2.GetTime().front() == (double)st / 1e9);
    }
    tpo_planner_destroy(o);
    tpo_planner_destroy(o2);
  }
  // several 7-joint planners with random waypoints, advanced together
  {
    const int K = 5, D = 7, N = 400;
    unsigned long long seed = 4242;
    auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
    std::vector<std::shared_ptr<TimeableJointSplinePath>> paths(K);
    std::vector<std::unique_ptr<PathTimingTrajectory>> planners;
    std::vector<PathTimingTrajectory *> ptrs;
    std::vector<tpo_planner *> oracles(K);
    for (int k = 0; k < K; k++) {
      std::vector<VectorXd> wps;
      for (int i = 0; i < 4 + k % 3; i++) { VectorXd v(D); for (int d = 0; d < D; d++) v[d] = 5.0 * rnd() - 2.5; wps.push_back(v); }
      auto probe = std::make_shared<TimeableJointSplinePath>(JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
      probe->SetWaypoints({wps.data(), wps.size()});
      // the last planner needs some 25 windows in one call: more than the first history the
      // mirror sends up can take, so the engine call is resumed (TPAMD_PLAN_MORE)
      const double delta = (k == K - 1 ? 0.05 : 0.35) * probe->knots().back() / (N - 1);
      paths[k] = std::make_shared<TimeableJointSplinePath>(
          JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta));
      std::vector<double> vmax(D), amax(D);
      for (int d = 0; d < D; d++) { vmax[d] = 1.0 + rnd(); amax[d] = 2.0 + 2.0 * rnd(); }
      CHECK(paths[k]->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
      CHECK(paths[k]->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
      CHECK(paths[k]->SetWaypoints({wps.data(), wps.size()}).ok());
      const bool skip = (k % 2) == 1;
      planners.push_back(std::make_unique<PathTimingTrajectory>(
          PathTimingTrajectoryOptions().SetNumDofs(D).SetNumPathSamples(N).SetTimeStep(Milliseconds(4))
              .SetTimeSamplingMethod(skip ? Method::kSkipSamplesCloserThanTimeStep : Method::kUniformlyInTime)));
      CHECK(planners.back()->SetPath(paths[k]).ok());
      ptrs.push_back(planners.back().get());
      oracles[k] = tpo_planner_create(D, N, delta, paths[k]->options().constraint_safety(), 4 * kMs, skip ? 1 : 0,
                                      10000, 1e-3);
      tpo_planner_set_limits(oracles[k], vmax.data(), amax.data());
      tpo_planner_set_spline(oracles[k], paths[k]->knots().data(), (int)paths[k]->knots().size(),
                             paths[k]->packed_control_points().data(), (int)paths[k]->num_control_points(),
                             TPO_PATH_NEW);
    }
    int64_t start = 3 * 1000 * kMs;
    for (int round = 0; round < 4; round++) {
      const auto st = PathTimingTrajectory::PlanBatch(ptrs, tpamd::compat::FromUnixNanos(start),
                                                      round == 3 ? Seconds(1000.0) : Milliseconds(600));
      for (int k = 0; k < K; k++) {
        const int rc = tpo_planner_plan(oracles[k], start, round == 3 ? (int64_t)1000 * 1000 * kMs : 600 * kMs);
        CHECK(st[k].ok() == (rc == TPO_PLAN_OK));
        if (st[k].ok() && rc == TPO_PLAN_OK) ComparePlannerWithOracle(*planners[k], oracles[k], D);
      }
      start += 150 * kMs;
    }
    for (int k = 0; k < K; k++) { CHECK(planners[k]->IsTrajectoryAtEnd()); tpo_planner_destroy(oracles[k]); }
  }
}

// A TimeableCartesianSplinePath behind PathTimingTrajectory::SetPath (timeable_path_cartesian_spline.h:
// 62-192), planned to its end by receding-horizon Plan calls, as the reference's Cartesian planner
// tests do with test doubles for the kinematics (path_timing_trajectory_test.cc:548-937: 7 fake
// joints = xyz + rotation vector + one null-space joint). The fake IK here is a pure function of the
// targets, so the IK solution of a sample does not depend on when it was computed; the oracle's
// planner is given the finished IK table and the Jacobian callback's values on it, replays the same
// Plan calls, and every replanning step must agree bit for bit. Rotation-vector extraction and the
// pose corner rounding restate Eigen / eigenmath algorithms: unpinned at the ulp level, but both
// sides of this comparison see the same table.
static void TestCartesianSplinePathPlanning() {
  using Method = PathTimingTrajectoryOptions::TimeSamplingMethod;
  using tpamd::compat::AngleAxisd;
  using tpamd::compat::Matrix6Xd;
  using tpamd::compat::Pose3d;
  using tpamd::compat::Quaterniond;
  using tpamd::compat::Vector3d;
  const int64_t kMs = 1000000;
  const int D = 7, N = 400;
  auto fake_ik = [D](const VectorXd &, const std::vector<Pose3d> &poses, const std::vector<VectorXd> &joints,
                     std::vector<VectorXd> *result) -> Status {
    result->clear();
    for (size_t i = 0; i < poses.size(); i++) {
      VectorXd q(D);
      for (int d = 0; d < 3; d++) q[d] = poses[i].translation()[d];
      const AngleAxisd aa(poses[i].quaternion());
      for (int d = 0; d < 3; d++) q[3 + d] = aa.axis[d] * aa.angle;
      q[6] = joints[i][6];
      result->push_back(q);
    }
    return tpamd::compat::OkStatus();
  };
  auto fake_jacobian = [D](const VectorXd &q, Matrix6Xd *J) -> Status {
    for (int r = 0; r < 6; r++)
      for (int d = 0; d < D; d++) (*J)(r, d) = 0.2 * std::sin(q[d] * (r + 1.0) + 0.31 * d) + (r == d ? 1.0 : 0.0);
    return tpamd::compat::OkStatus();
  };
  auto make_pose = [](double x, double y, double z, double ax, double ay, double az, double angle) {
    AngleAxisd aa;
    const double n = std::sqrt(ax * ax + ay * ay + az * az);
    aa.axis = Vector3d(ax / n, ay / n, az / n);
    aa.angle = angle;
    return Pose3d(aa.toQuaternion(), Vector3d(x, y, z));
  };
  for (Method method : {Method::kUniformlyInTime, Method::kSkipSamplesCloserThanTimeStep}) {
    const std::vector<Pose3d> poses = {make_pose(0.3, 0.0, 0.4, 0, 0, 1, 0.1), make_pose(0.5, 0.25, 0.6, 0, 1, 0, 0.7),
                                       make_pose(0.2, 0.5, 0.3, 1, 0, 0, 0.4), make_pose(0.45, 0.1, 0.5, 0, 0, 1, 1.0)};
    std::vector<VectorXd> joints;
    for (size_t i = 0; i < poses.size(); i++) {
      VectorXd q(D);
      for (int d = 0; d < 3; d++) q[d] = poses[i].translation()[d];
      const AngleAxisd aa(poses[i].quaternion());
      for (int d = 0; d < 3; d++) q[3 + d] = aa.axis[d] * aa.angle;
      q[6] = 0.2 * (double)i - 0.3;
      joints.push_back(q);
    }
    // delta so that the path needs several windows (the knot vector depends on the waypoints only)
    CartesianPathOptions probe_opt;
    probe_opt.set_num_dofs(D).set_num_path_samples(N);
    probe_opt.set_path_ik_func(fake_ik).set_jacobian_func(fake_jacobian);
    TimeableCartesianSplinePath probe(probe_opt);
    CHECK(probe.SetWaypoints({poses.data(), poses.size()}, {joints.data(), joints.size()}).ok());
    const double kend = probe.knots().back();
    const double delta = 0.4 * kend / (N - 1);
    CartesianPathOptions opt;
    opt.set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta);
    opt.set_path_ik_func(fake_ik).set_jacobian_func(fake_jacobian);
    auto path = std::make_shared<TimeableCartesianSplinePath>(opt);
    PathTimingTrajectory planner(PathTimingTrajectoryOptions().SetTimeStep(Milliseconds(4)).SetNumDofs(D)
                                     .SetNumPathSamples(N).SetTimeSamplingMethod(method));
    CHECK(planner.SetPath(path).ok());
    std::vector<double> vmax = {0.6, 0.5, 0.7, 1.0, 0.9, 1.1, 0.8}, amax = {1.5, 1.2, 1.8, 2.5, 2.0, 3.0, 2.2};
    CHECK(path->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
    CHECK(path->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
    CHECK(path->SetMaxCartesianVelocity(0.35, 0.9).ok());
    CHECK(!path->SetMaxCartesianVelocity(0.0, 1.0).ok());
    CHECK(path->SetWaypoints({poses.data(), poses.size()}, {joints.data(), joints.size()}).ok());
    CHECK(path->GetState() == TimeablePath::State::kNewPath);
    CHECK(!path->SwitchToWaypointPath(0.1, {poses.data(), poses.size()}, {joints.data(), joints.size()}).ok());
    CHECK(path->GetState() == TimeablePath::State::kNewPath);
    // the mirror first, recording every step's trajectory
    struct Step { int64_t start; std::vector<double> t, s; std::vector<double> pos, vel, acc; bool ok; };
    std::vector<Step> steps;
    int64_t start = 0;
    while (!planner.IsTrajectoryAtEnd() && steps.size() < 100) {
      Step st;
      st.start = start;
      st.ok = planner.Plan(tpamd::compat::FromUnixNanos(start), Milliseconds(750)).ok();
      CHECK(st.ok);
      if (!st.ok) break;
      st.t.assign(planner.GetTime().begin(), planner.GetTime().end());
      st.s.assign(planner.GetPathParameters().begin(), planner.GetPathParameters().end());
      for (size_t i = 0; i < planner.GetPositions().size(); i++)
        for (int d = 0; d < D; d++) {
          st.pos.push_back(planner.GetPositions()[i][d]);
          st.vel.push_back(planner.GetVelocities()[i][d]);
          st.acc.push_back(planner.GetAccelerations()[i][d]);
        }
      steps.push_back(st);
      const int64_t end_ns = tpamd::compat::ToUnixNanos(planner.GetEndTime());
      start = std::min<int64_t>(end_ns, start + 200 * kMs);
    }
    CHECK(planner.IsTrajectoryAtEnd());
    CHECK(steps.size() > 4);
    // end of the motion: the last joint waypoint (through the fake IK), at rest
    for (int d = 0; d < D; d++) {
      CHECK(planner.GetVelocities().back()[d] == 0.0);
      CHECK(std::fabs(planner.GetPositions().back()[d] - joints.back()[d]) < 1e-6);
    }
    // Cartesian speed limit along the trajectory (safety factor applies to joint limits only)
    // the oracle on the finished IK table
    const std::vector<VectorXd> &table = path->GetSplineIKPosition();
    const int M = (int)table.size();
    CHECK(M > N);
    std::vector<double> tq((size_t)M * D), tJ((size_t)M * 6 * D);
    Matrix6Xd J(6, D);
    for (int i = 0; i < M; i++) {
      for (int d = 0; d < D; d++) tq[(size_t)i * D + d] = table[i][d];
      fake_jacobian(table[i], &J);
      std::copy(J.data(), J.data() + 6 * D, tJ.begin() + (size_t)i * 6 * D);
    }
    tpo_planner *o = tpo_planner_create(D, N, delta, path->options().constraint_safety(), 4 * kMs,
                                        method == Method::kUniformlyInTime ? 0 : 1, 10000, 1e-3);
    tpo_planner_set_limits(o, vmax.data(), amax.data());
    tpo_planner_set_ik_table(o, tq.data(), tJ.data(), M, path->knots().back(), 0.35, 0.9, TPO_PATH_NEW);
    int windows = 0;
    for (const Step &st : steps) {
      const int rc = tpo_planner_plan(o, st.start, 750 * kMs);
      CHECK(rc == TPO_PLAN_OK);
      if (rc != TPO_PLAN_OK) break;
      windows += tpo_planner_windows(o);
      const int Mo = tpo_planner_num_samples(o);
      CHECK(Mo == (int)st.t.size());
      if (Mo != (int)st.t.size()) break;
      for (int i = 0; i < Mo; i++) {
        CHECK(st.t[i] == tpo_planner_time(o)[i]);
        CHECK(st.s[i] == tpo_planner_path_parameter(o)[i]);
      }
      for (int i = 0; i < Mo * D; i++) {
        CHECK(st.pos[i] == tpo_planner_positions(o)[i]);
        CHECK(st.vel[i] == tpo_planner_velocities(o)[i]);
        CHECK(st.acc[i] == tpo_planner_accelerations(o)[i]);
      }
    }
    CHECK(windows >= 3);
    CHECK(tpo_planner_target_reached(o) == 1);
    tpo_planner_destroy(o);
    // SamplePath / ConstraintSetup on their own give the rows the planner's fused call forms on
    // the device: the two Cartesian rows have A = 0, lower = -upper (timeable_path_cartesian_spline.cc:578-592)
    CHECK(path->SamplePath(0.0).ok());
    CHECK(path->ConstraintSetup().ok());
    const auto &rows = path->GetConstraints();
    CHECK((int)rows.size() == N && rows[5].size() == 2 * D + 2);
    CHECK(rows[5].a_coefficient(2 * D) == 0.0 && rows[5].lower(2 * D) == -rows[5].upper(2 * D));
    CHECK(rows[5].upper(2 * D) == 0.35 * 0.35 && rows[5].upper(2 * D + 1) == 0.9 * 0.9);
    CHECK(path->PathIkIndex(path->PathIkParameter(17)) == 17);
  }
  // several Cartesian planners of two shapes in one PlanBatch call equal one Plan each
  {
    std::vector<std::shared_ptr<TimeableCartesianSplinePath>> paths;
    std::vector<std::unique_ptr<PathTimingTrajectory>> batch, single;
    std::vector<PathTimingTrajectory *> ptrs;
    for (int k = 0; k < 4; k++) {
      const int Nk = (k % 2) ? 300 : 260;
      for (int copy = 0; copy < 2; copy++) {
        CartesianPathOptions opt;
        opt.set_num_dofs(D).set_num_path_samples(Nk).set_delta_parameter(0.02 + 0.004 * k);
        opt.set_path_ik_func(fake_ik).set_jacobian_func(fake_jacobian);
        auto path = std::make_shared<TimeableCartesianSplinePath>(opt);
        std::vector<double> vmax(D, 0.7 + 0.1 * k), amax(D, 1.5 + 0.2 * k);
        CHECK(path->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
        CHECK(path->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
        CHECK(path->SetMaxCartesianVelocity(0.3 + 0.05 * k, 0.8).ok());
        std::vector<Pose3d> poses = {make_pose(0.1 * k, 0.0, 0.4, 0, 0, 1, 0.1), make_pose(0.5, 0.2 + 0.05 * k, 0.6, 0, 1, 0, 0.5),
                                     make_pose(0.2, 0.5, 0.3 + 0.02 * k, 1, 0, 0, 0.3)};
        std::vector<VectorXd> joints;
        for (size_t i = 0; i < poses.size(); i++) {
          VectorXd q(D);
          for (int d = 0; d < 3; d++) q[d] = poses[i].translation()[d];
          const AngleAxisd aa(poses[i].quaternion());
          for (int d = 0; d < 3; d++) q[3 + d] = aa.axis[d] * aa.angle;
          q[6] = 0.1 * (double)i;
          joints.push_back(q);
        }
        CHECK(path->SetWaypoints({poses.data(), poses.size()}, {joints.data(), joints.size()}).ok());
        auto pl = std::make_unique<PathTimingTrajectory>(
            PathTimingTrajectoryOptions().SetTimeStep(Milliseconds(4)).SetNumDofs(D).SetNumPathSamples(Nk));
        CHECK(pl->SetPath(path).ok());
        if (copy == 0) { ptrs.push_back(pl.get()); batch.push_back(std::move(pl)); }
        else single.push_back(std::move(pl));
      }
    }
    for (int round = 0; round < 3; round++) {
      const auto start = tpamd::compat::FromUnixNanos((int64_t)round * 150 * kMs);
      const auto st = PathTimingTrajectory::PlanBatch(ptrs, start, Milliseconds(500));
      for (size_t k = 0; k < ptrs.size(); k++) {
        CHECK(st[k].ok());
        CHECK(single[k]->Plan(start, Milliseconds(500)).ok());
        CHECK(batch[k]->GetTime() == single[k]->GetTime());
        CHECK(batch[k]->GetPositions().size() == single[k]->GetPositions().size());
        for (size_t i = 0; i < batch[k]->GetPositions().size(); i++) {
          CHECK(batch[k]->GetPositions()[i] == single[k]->GetPositions()[i]);
          CHECK(batch[k]->GetAccelerations()[i] == single[k]->GetAccelerations()[i]);
        }
      }
    }
  }
}

// PathTimingTrajectorySet: planners whose state lives on the device (tpamd_planner_set_*). Every
// planner must equal the oracle's Plan bit for bit at every replanning step -- window loop,
// resampling in time (both methods), the "planned enough" erase branch, a new path after the end
// of the old one, error statuses -- while a Plan call moves only a few bytes per planner.
static void TestPlannerSet() {
  using Method = PathTimingTrajectoryOptions::TimeSamplingMethod;
  const int64_t kMs = 1000000;
  const int K = 6, D = 7, N = 400, W = 5, P = 3 * W - 2;
  for (Method method : {Method::kUniformlyInTime, Method::kSkipSamplesCloserThanTimeStep}) {
    const bool skip = method == Method::kSkipSamplesCloserThanTimeStep;
    const int64_t step_ns = skip ? 4 * kMs : 1 * kMs;      // 1 ms: the long path outgrows 4096 trajectory samples
    unsigned long long seed = skip ? 9001 : 31337;
    auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
    PathTimingTrajectoryOptions opt;
    opt.SetNumDofs(D).SetNumPathSamples(N).SetTimeStep(tpamd::compat::Nanoseconds(step_ns)).SetTimeSamplingMethod(method);
    PathTimingTrajectorySet set(opt, K + 1, P);             // the last planner never gets a path
    CHECK(set.status().ok());
    if (!set.status().ok()) return;
    std::vector<std::shared_ptr<TimeableJointSplinePath>> paths(K);
    std::vector<tpo_planner *> oracles(K);
    auto make_path = [&](int k, double fraction) {
      std::vector<VectorXd> wps;
      for (int i = 0; i < W; i++) { VectorXd v(D); for (int d = 0; d < D; d++) v[d] = 5.0 * rnd() - 2.5; wps.push_back(v); }
      auto probe = std::make_shared<TimeableJointSplinePath>(JointPathOptions().set_num_dofs(D).set_num_path_samples(N));
      probe->SetWaypoints({wps.data(), wps.size()});
      const double delta = fraction * probe->knots().back() / (N - 1);
      auto path = std::make_shared<TimeableJointSplinePath>(
          JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta));
      std::vector<double> vmax(D), amax(D);
      for (int d = 0; d < D; d++) { vmax[d] = 1.0 + rnd(); amax[d] = 2.0 + 2.0 * rnd(); }
      CHECK(path->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
      CHECK(path->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
      CHECK(path->SetWaypoints({wps.data(), wps.size()}).ok());
      (void)k;
      return path;
    };
    auto oracle_set_path = [&](tpo_planner *o, const TimeableJointSplinePath &p, int state) {
      tpo_planner_set_limits(o, p.GetMaxJointVelocity().data(), p.GetMaxJointAcceleration().data());
      tpo_planner_set_spline(o, p.knots().data(), (int)p.knots().size(), p.packed_control_points().data(),
                             p.num_control_points(), state);
    };
    for (int k = 0; k < K; k++) {
      // the last one needs ~25 windows in a single call: its history outgrows the initial 8 N samples
      paths[k] = make_path(k, k == K - 1 ? 0.05 : 0.3 + 0.05 * k);
      oracles[k] = tpo_planner_create(D, N, paths[k]->GetPathSamplingDistance(), 0.8, step_ns, skip ? 1 : 0, 200, 1e-2);
      oracle_set_path(oracles[k], *paths[k], TPO_PATH_NEW);
    }
    CHECK(set.SetPaths(paths).ok());
    auto compare = [&](int k) {
      tpo_planner *o = oracles[k];
      const int M = tpo_planner_num_samples(o);
      CHECK((int)set.GetNumTimeSamples(k) == M);
      CHECK(tpamd::compat::ToUnixNanos(set.GetEndTime(k)) == tpo_planner_end_time(o));
      CHECK(tpamd::compat::ToUnixNanos(set.GetFinalDecelStart(k)) == tpo_planner_final_decel_start(o));
      CHECK(set.IsTrajectoryAtEnd(k) == (tpo_planner_target_reached(o) != 0 && tpo_planner_path_state(o) == TPO_PATH_SAMPLED));
      CHECK(set.WindowsOfLastPlan(k) == tpo_planner_windows(o));
      PlannedTrajectory tr;
      CHECK(set.GetTrajectory(k, &tr).ok());
      if ((int)tr.time.size() != M) return;
      for (int i = 0; i < M; i++) {
        CHECK(tr.time[i] == tpo_planner_time(o)[i]);
        CHECK(tr.path_parameter[i] == tpo_planner_path_parameter(o)[i]);
        CHECK(tr.path_parameter_derivative[i] == tpo_planner_path_velocity(o)[i]);
        CHECK(tr.second_path_parameter_derivative[i] == tpo_planner_path_acceleration(o)[i]);
      }
      for (int i = 0; i < M * D; i++) {
        CHECK(tr.positions[i] == tpo_planner_positions(o)[i]);
        CHECK(tr.velocities[i] == tpo_planner_velocities(o)[i]);
        CHECK(tr.accelerations[i] == tpo_planner_accelerations(o)[i]);
      }
    };
    auto code_of = [](const Status &st) {
      using tpamd::compat::StatusCode;
      switch (st.code()) {
        case StatusCode::kOk: return (int)TPO_PLAN_OK;
        case StatusCode::kFailedPrecondition: return (int)TPO_PLAN_FAILED_PRECONDITION;
        case StatusCode::kOutOfRange: return (int)TPO_PLAN_OUT_OF_RANGE;
        case StatusCode::kInvalidArgument: return (int)TPO_PLAN_INVALID_ARGUMENT;
        case StatusCode::kDeadlineExceeded: return (int)TPO_PLAN_DEADLINE_EXCEEDED;
        default: return (int)TPO_PLAN_INTERNAL;
      }
    };
    int64_t start = 3 * 1000 * kMs;
    size_t max_bytes = 0;
    for (int round = 0; round < 7; round++) {
      const int64_t horizon = round == 4 ? (int64_t)1000 * 1000 * kMs : 600 * kMs;
      const auto st = set.Plan(tpamd::compat::FromUnixNanos(start), tpamd::compat::Nanoseconds(horizon));
      max_bytes = std::max(max_bytes, set.LastPlanBytesOverPcie());
      CHECK(st[K].code() == tpamd::compat::StatusCode::kFailedPrecondition);    // no path set
      for (int k = 0; k < K; k++) {
        const int rc = tpo_planner_plan(oracles[k], start, horizon);
        CHECK(code_of(st[k]) == rc);
        if (rc == TPO_PLAN_OK) compare(k);
      }
      // rounds 5, 6: everything is planned to the end -- only the erase branch runs
      if (round >= 5) for (int k = 0; k < K; k++) CHECK(set.WindowsOfLastPlan(k) == 0);
      start += round >= 4 ? 137 * kMs + 12345 : 150 * kMs;
    }
    for (int k = 0; k < K; k++) CHECK(set.IsTrajectoryAtEnd(k));
    // a Plan call moves a few bytes per planner, not the histories
    CHECK(max_bytes < (size_t)(K + 1) * 100 + 4096);
    // error statuses: a start beyond the previous plan's end, and one before its start
    {
      std::vector<tpamd::compat::Time> starts(K + 1, tpamd::compat::FromUnixNanos(start));
      std::vector<tpamd::compat::Duration> hor(K + 1, Milliseconds(600));
      starts[0] = tpamd::compat::FromUnixNanos(tpo_planner_end_time(oracles[0]) + 50 * kMs);
      starts[1] = tpamd::compat::FromUnixNanos(1000 * kMs);
      const auto st = set.Plan(starts, hor);
      for (int k = 0; k < K; k++) {
        const int rc = tpo_planner_plan(oracles[k], tpamd::compat::ToUnixNanos(starts[k]), 600 * kMs);
        CHECK(code_of(st[k]) == rc);
        if (rc == TPO_PLAN_OK) compare(k);
      }
      CHECK(!st[0].ok() && !st[1].ok());
    }
    // New paths after the old ones were finished. Planner 2 keeps its planner state (the reference's
    // use: SetWaypoints on the path object the planner holds; UpdatePathTrackingStatus then resets
    // the path bookkeeping, :488-497) -- same sampling distance as before, which the oracle object
    // fixes at creation. Planner 4 is Reset first and gets a path with another sampling distance.
    {
      const double delta2 = paths[2]->GetPathSamplingDistance();
      std::vector<VectorXd> wps;
      for (int i = 0; i < W; i++) { VectorXd v(D); for (int d = 0; d < D; d++) v[d] = 4.0 * rnd() - 2.0; wps.push_back(v); }
      auto path = std::make_shared<TimeableJointSplinePath>(
          JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(delta2));
      std::vector<double> vmax(D, 1.3), amax(D, 2.6);
      CHECK(path->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
      CHECK(path->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
      CHECK(path->SetWaypoints({wps.data(), wps.size()}).ok());
      paths[2] = path;
      CHECK(set.SetPath(2, *paths[2]).ok());
      oracle_set_path(oracles[2], *paths[2], TPO_PATH_NEW);
      paths[4] = make_path(4, 0.4);
      set.Reset(4);
      CHECK(set.SetPath(4, *paths[4]).ok());
      tpo_planner_destroy(oracles[4]);
      oracles[4] = tpo_planner_create(D, N, paths[4]->GetPathSamplingDistance(), 0.8, step_ns, skip ? 1 : 0, 200, 1e-2);
      oracle_set_path(oracles[4], *paths[4], TPO_PATH_NEW);
    }
    start += 500 * kMs;
    for (int round = 0; round < 3; round++) {
      const int64_t horizon = round == 2 ? (int64_t)1000 * 1000 * kMs : 700 * kMs;
      std::vector<tpamd::compat::Time> starts(K + 1, tpamd::compat::FromUnixNanos(start));
      std::vector<tpamd::compat::Duration> hor(K + 1, tpamd::compat::Nanoseconds(horizon));
      const auto st = set.Plan(starts, hor);
      for (int k : {2, 4}) {
        const int rc = tpo_planner_plan(oracles[k], start, horizon);
        CHECK(code_of(st[k]) == rc && rc == TPO_PLAN_OK);
        if (rc == TPO_PLAN_OK) compare(k);
      }
      start += 180 * kMs;
    }
    CHECK(set.IsTrajectoryAtEnd(2) && set.IsTrajectoryAtEnd(4));
    std::printf("planner set (%s): %zu bytes over PCIe per Plan call at most, %.1f MB on the device\n",
                skip ? "skip" : "uniform", max_bytes, set.DeviceBytes() / 1e6);
    for (int k = 0; k < K; k++) tpo_planner_destroy(oracles[k]);
  }
}

// SwitchToWaypointPath (timeable_path_joint_spline.cc:209-250) while a plan is being followed,
// as in the reference's SwitchToNewJointWaypointPathWorks (path_timing_trajectory_test.cc:298-420):
// plan, switch to a new waypoint path at a parameter ahead of the robot, carry the current
// velocity over as the initial velocity, keep replanning to the end. The HIP-backed planner and
// the oracle's Plan see the same edited spline (state kModifiedPath) and must agree bit for bit;
// the trajectory must stay continuous across the switch and end at the new last waypoint.
static void TestSwitchPathPlanning() {
  using Method = PathTimingTrajectoryOptions::TimeSamplingMethod;
  const int64_t kMs = 1000000;
  const int D = 3, N = 1000;
  auto V3 = [](double x, double y, double z) { VectorXd v(3); v[0] = x; v[1] = y; v[2] = z; return v; };
  auto path = std::make_shared<TimeableJointSplinePath>(
      JointPathOptions().set_num_dofs(D).set_num_path_samples(N).set_delta_parameter(0.001));
  PathTimingTrajectory planner(PathTimingTrajectoryOptions().SetTimeStep(Milliseconds(4)).SetNumDofs(D)
                                   .SetNumPathSamples(N).SetTimeSamplingMethod(Method::kUniformlyInTime));
  CHECK(planner.SetPath(path).ok());
  const std::vector<VectorXd> wps = {V3(1, 2, 3), V3(-1, -2, -3), V3(0.5, 1.0, 1.5)};
  const std::vector<VectorXd> new_wps = {V3(1, 2, 3), V3(0.5, 1.0, 5.5)};
  CHECK(path->SetWaypoints({wps.data(), wps.size()}).ok());
  std::vector<double> vmax(D, 1.0), amax(D, 2.0);
  CHECK(path->SetMaxJointVelocity({vmax.data(), vmax.size()}).ok());
  CHECK(path->SetMaxJointAcceleration({amax.data(), amax.size()}).ok());
  tpo_planner *o = tpo_planner_create(D, N, 0.001, path->options().constraint_safety(), 4 * kMs, 0, 10000, 1e-3);
  tpo_planner_set_limits(o, vmax.data(), amax.data());
  tpo_planner_set_spline(o, path->knots().data(), (int)path->knots().size(), path->packed_control_points().data(),
                         (int)path->num_control_points(), TPO_PATH_NEW);
  int64_t start = 0;
  for (int loop = 0; loop < 3; loop++) {      // follow the first path for a while
    CHECK(planner.Plan(tpamd::compat::FromUnixNanos(start), Milliseconds(750)).ok());
    CHECK(tpo_planner_plan(o, start, 750 * kMs) == TPO_PLAN_OK);
    ComparePlannerWithOracle(planner, o, D);
    start += 200 * kMs;
  }
  CHECK(!planner.IsTrajectoryAtEnd());
  // the sample at the next start time: its velocity is carried over, the switch happens ahead of it
  size_t k = 0;
  while (k + 1 < planner.GetTime().size() && planner.GetTime()[k] < (double)start / 1e9 - 1e-9) k++;
  const VectorXd v_now = planner.GetVelocities()[k];
  const VectorXd q_now = planner.GetPositions()[k];
  const double s_keep = planner.GetPathParameters()[std::min(k + 60, planner.GetPathParameters().size() - 1)];
  CHECK(path->SwitchToWaypointPath(s_keep, {new_wps.data(), new_wps.size()}).ok());
  CHECK(path->GetState() == TimeablePath::State::kModifiedPath);
  CHECK(path->SetInitialVelocity({v_now.data(), v_now.size()}).ok());
  tpo_planner_set_spline(o, path->knots().data(), (int)path->knots().size(), path->packed_control_points().data(),
                         (int)path->num_control_points(), TPO_PATH_MODIFIED);
  tpo_planner_set_initial_velocity(o, v_now.data());
  int loops = 0;
  bool first = true;
  while (!planner.IsTrajectoryAtEnd() && loops < 100) {
    const bool ok = planner.Plan(tpamd::compat::FromUnixNanos(start), Milliseconds(750)).ok();
    const int rc = tpo_planner_plan(o, start, 750 * kMs);
    CHECK(ok && rc == TPO_PLAN_OK);
    if (!ok || rc != TPO_PLAN_OK) break;
    ComparePlannerWithOracle(planner, o, D);
    if (first) {                               // continuity across the switch
      for (int d = 0; d < D; d++) {
        CHECK(std::fabs(planner.GetPositions().front()[d] - q_now[d]) < 1e-6);
        CHECK(std::fabs(planner.GetVelocities().front()[d] - v_now[d]) < 2e-3);
      }
      first = false;
    }
    start = std::min<int64_t>(tpo_planner_end_time(o), start + 200 * kMs);
    loops++;
  }
  CHECK(planner.IsTrajectoryAtEnd() && loops > 1);
  for (int d = 0; d < D; d++) {
    CHECK(planner.GetVelocities().back()[d] == 0.0);
    CHECK(std::fabs(planner.GetPositions().back()[d] - new_wps.back()[d]) < 1e-9);
  }
  tpo_planner_destroy(o);
}

int main() {
  TestProfileAgainstOracle();
  TestJointPathAndPlanner();
  TestBatch();
  TestMixedBatch();
  TestCartesianBatch();
  TestPlanBatch();
  TestPlanAgainstOracle();
  TestSwitchPathPlanning();
  TestCartesianSplinePathPlanning();
  TestPlannerSet();
  if (g_fail == 0) std::printf("ALL OK\n");
  else std::printf("%d CHECKS FAILED\n", g_fail);
  return g_fail == 0 ? 0 : 1;
}
