// CPU-only checks of the host mirror's path edits (run by tests/test_host_cpu.py): the
// reference's own knot-insertion / truncation / extension / projection tests restated as data
// (splines/bspline_test.cc:1317-1642, path_tools_test.cc:41-110), SwitchToWaypointPath's
// invariants, and the brute-force LP of the mirror against the oracle's. No engine call.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../oracle/tp_oracle.h"
#include "../../x-edr-trajectory-planning_amd/host/spline_edit.h"
#include "../../x-edr-trajectory-planning_amd/host/timeable_path_cartesian_spline.h"
#include "../../x-edr-trajectory-planning_amd/host/time_optimal_path_timing.h"
#include "../../x-edr-trajectory-planning_amd/host/timeable_path_joint_spline.h"

using namespace trajectory_planning;
using tpamd::compat::StatusCode;

static int g_fail = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
  } while (0)

static VectorXd V2(double x, double y) { VectorXd v(2); v[0] = x; v[1] = y; return v; }
static bool Near(const VectorXd &a, const VectorXd &b, double tol) {
  double n = 0, m = 0;
  for (size_t i = 0; i < a.size(); i++) { n += (a[i] - b[i]) * (a[i] - b[i]); m += b[i] * b[i]; }
  return std::sqrt(n) <= tol * std::max(1.0, std::sqrt(m));
}
static std::vector<double> UniformKnots(int num_points, int degree) {
  std::vector<double> k(num_points + degree + 1);
  tpo_make_uniform_knots(num_points, degree, 0.0, 1.0, k.data());
  return k;
}

static void TestInsertKnot() {   // bspline_test.cc:1346-1436
  const double r3 = std::sqrt(3.0) * 0.5;
  const std::vector<VectorXd> points = {V2(-1.5, -0.5), V2(-1.1, -0.1), V2(-0.5, -r3), V2(0.5, -r3),
                                        V2(1, 0), V2(0.5, r3), V2(-0.5, r3), V2(-1.0, -0.0)};
  for (int degree = 1; degree <= 3; ++degree) {
    for (int mult = 1; mult <= degree; ++mult) {
      const auto knots = UniformKnots((int)points.size(), degree);
      const int nk = (int)knots.size();
      const double to_insert[3] = {0.5 * knots[0] + 0.5 * knots[degree + 1], 0.5 * knots.front() + 0.5 * knots.back(),
                                   0.5 * knots[nk - degree - 2] + 0.5 * knots.back()};
      for (double the_knot : to_insert) {
        EditableBSpline ref, mod;
        CHECK(ref.Init(degree, nk + mult, {knots.data(), knots.size()}, {points.data(), points.size()}).ok());
        CHECK(mod.Init(degree, nk + mult, {knots.data(), knots.size()}, {points.data(), points.size()}).ok());
        CHECK(mod.InsertKnotAndUpdateControlPoints(the_knot, mult).ok());
        CHECK((int)mod.knots().size() == nk + mult);
        CHECK((int)mod.control_points().size() == (int)points.size() + mult);
        for (double u = knots.front(); u <= knots.back(); u += 0.001) {
          VectorXd a, b;
          CHECK(ref.EvalCurve(u, &a).ok() && mod.EvalCurve(u, &b).ok());
          if (!Near(b, a, 1e-12)) { CHECK(false); break; }
        }
      }
    }
  }
  // InsertKnotFailsForInvalidInput (:1317-1344): outside the knot range, capacity, multiplicity
  const auto knots = UniformKnots(8, 2);
  EditableBSpline s;
  CHECK(s.Init(2, (int)knots.size() + 1, {knots.data(), knots.size()}, {points.data(), points.size()}).ok());
  CHECK(s.InsertKnotAndUpdateControlPoints(-0.1, 1).code() == StatusCode::kInvalidArgument);
  CHECK(s.InsertKnotAndUpdateControlPoints(1.0, 1).code() == StatusCode::kInvalidArgument);
  CHECK(s.InsertKnotAndUpdateControlPoints(0.5, 4).code() == StatusCode::kInvalidArgument);
  CHECK(s.InsertKnotAndUpdateControlPoints(0.5, 2).code() == StatusCode::kFailedPrecondition);
}

static void TestTruncate() {   // bspline_test.cc:1438-1543
  const std::vector<VectorXd> points = {V2(0, 0), V2(0, 1), V2(1, 1), V2(1, 0), V2(2, 0), V2(2, 1)};
  const auto knots = UniformKnots(6, 3);
  EditableBSpline ref;
  CHECK(ref.Init(3, 20, {knots.data(), knots.size()}, {points.data(), points.size()}).ok());
  {
    EditableBSpline s = ref;
    CHECK(s.TruncateSplineAt(knots.back() + 1).ok() && s.knots() == ref.knots());
    CHECK(s.TruncateSplineAt(knots.back()).ok() && s.knots() == ref.knots());
    CHECK(s.control_points().size() == points.size());
  }
  for (double at : {knots.front(), knots.front() - 1.0}) {
    EditableBSpline s = ref;
    VectorXd v;
    CHECK(s.TruncateSplineAt(at).ok());
    CHECK(s.knots().empty() && s.control_points().empty());
    CHECK(s.EvalCurve(0.0, &v).code() == StatusCode::kOutOfRange);
  }
  for (double end : {0.01, 0.1, 0.3333, 0.6, 0.9, 0.999}) {
    EditableBSpline s = ref;
    VectorXd expected;
    CHECK(ref.EvalCurve(end, &expected).ok());
    CHECK(s.TruncateSplineAt(end).ok());
    CHECK(Near(s.control_points().back(), expected, 1e-7));
    CHECK(s.knots().back() == end);
    CHECK((int)s.control_points().size() == EditableBSpline::NumPoints((int)s.knots().size(), 3));
    for (double u = s.knots().front(); u <= s.knots().back(); u += 0.001) {
      VectorXd a, b;
      CHECK(ref.EvalCurve(u, &a).ok() && s.EvalCurve(u, &b).ok());
      if (!Near(b, a, 1e-7)) { CHECK(false); break; }
    }
  }
}

static void TestExtend() {   // bspline_test.cc:1545-1642
  const std::vector<VectorXd> points = {V2(0, 0), V2(0, 1), V2(1, 1), V2(1, 0), V2(2, 0), V2(2, 1)};
  const std::vector<VectorXd> extra = {V2(3, 1), V2(3, 0), V2(4, 0), V2(4, 1), V2(4, 2), V2(4, 3)};
  {
    const std::vector<VectorXd> four(points.begin(), points.begin() + 4);
    const auto k3 = UniformKnots(4, 3);
    EditableBSpline s;
    CHECK(s.Init(3, 100, {k3.data(), k3.size()}, {four.data(), four.size()}).ok());
    CHECK(s.ExtendWithControlPoints({four.data(), four.size()}).code() == StatusCode::kUnimplemented);
    const auto k2 = UniformKnots(4, 2);
    CHECK(s.Init(2, 10, {k2.data(), k2.size()}, {four.data(), four.size()}).ok());
    CHECK(s.ExtendWithControlPoints({four.data(), 1}).code() == StatusCode::kUnimplemented);
    CHECK(s.ExtendWithControlPoints({four.data(), four.size()}).code() == StatusCode::kFailedPrecondition);
  }
  const auto knots = UniformKnots(6, 2);
  EditableBSpline ref;
  CHECK(ref.Init(2, 50, {knots.data(), knots.size()}, {points.data(), points.size()}).ok());
  for (size_t count = 2; count < extra.size(); ++count) {
    EditableBSpline s = ref;
    CHECK(s.ExtendWithControlPoints({extra.data(), count}).ok());
    CHECK((int)s.control_points().size() == 6 + (int)count);
    CHECK((int)s.knots().size() == EditableBSpline::NumKnots(6 + (int)count, 2));
    for (size_t i = 1; i < s.knots().size(); i++) CHECK(s.knots()[i] >= s.knots()[i - 1]);
    // the old section is unchanged in value and slope (finite differences on both curves)
    for (double u = knots.front(); u < knots.back() - 0.01; u += 0.01) {
      VectorXd a, b, a2, b2;
      CHECK(ref.EvalCurve(u, &a).ok() && s.EvalCurve(u, &b).ok());
      CHECK(ref.EvalCurve(u + 1e-6, &a2).ok() && s.EvalCurve(u + 1e-6, &b2).ok());
      if (!Near(b, a, 1e-12)) { CHECK(false); break; }
      for (int d = 0; d < 2; d++) CHECK(std::fabs((a2[d] - a[d]) - (b2[d] - b[d])) < 1e-12);
    }
    // the new section joins smoothly: one-sided slopes at the joint agree
    const double uj = knots.back(), h = 1e-6;
    VectorXd l0, l1, r0, r1;
    CHECK(s.EvalCurve(uj - h, &l0).ok() && s.EvalCurve(uj, &l1).ok() && s.EvalCurve(uj + h, &r1).ok());
    r0 = l1;
    for (int d = 0; d < 2; d++) CHECK(std::fabs((l1[d] - l0[d]) - (r1[d] - r0[d])) < 1e-4 * h * 1e3);
    VectorXd end;
    CHECK(s.EvalCurve(s.knots().back(), &end).ok());
    CHECK(Near(end, extra[count - 1], 1e-12));
  }
}

static void TestProjectPointOnPath() {   // path_tools_test.cc:41-110
  std::vector<VectorXd> none;
  CHECK(ProjectPointOnPath({none.data(), none.size()}, V2(0, 0)).status().code() == StatusCode::kInvalidArgument);
  {
    std::vector<VectorXd> wrong = {VectorXd(3), VectorXd(3)};
    CHECK(ProjectPointOnPath({wrong.data(), wrong.size()}, V2(0, 0)).status().code() == StatusCode::kInvalidArgument);
  }
  {
    std::vector<VectorXd> one = {V2(1, 1)};
    const auto r = ProjectPointOnPath({one.data(), one.size()}, V2(1, 1));
    CHECK(r.ok() && (*r).waypoint_index == 0 && (*r).distance_to_path == 0.0 && (*r).line_parameter == 0.0);
  }
  std::vector<VectorXd> two = {V2(1, 1), V2(2, 2)};
  {
    const auto r = ProjectPointOnPath({two.data(), two.size()}, V2(1, 1));
    CHECK(r.ok() && (*r).waypoint_index == 0 && (*r).distance_to_path == 0.0 && (*r).line_parameter == 0.0);
  }
  {
    const auto r = ProjectPointOnPath({two.data(), two.size()}, V2(2, 2));
    CHECK(r.ok() && (*r).waypoint_index == 0 && (*r).distance_to_path == 0.0 && (*r).line_parameter == 1.0);
  }
  {
    std::vector<VectorXd> three = {V2(1, 1), V2(2, 2), V2(-3, -3)};
    const double t = 0.4;
    const VectorXd proj = V2(2 + t * (-5), 2 + t * (-5));
    const VectorXd point = V2(proj[0] + 0.1, proj[1] - 0.1);
    const auto r = ProjectPointOnPath({three.data(), three.size()}, point);
    CHECK(r.ok() && (*r).waypoint_index == 1);
    CHECK(std::fabs((*r).distance_to_path - std::sqrt(0.02)) < 1e-15);
    CHECK(std::fabs((*r).line_parameter - t) < 1e-15);
    CHECK(Near((*r).projected_point, proj, 1e-12));
  }
}

static void TestSwitchToWaypointPath() {   // timeable_path_joint_spline.cc:209-250
  const int D = 3;
  auto V3 = [](double x, double y, double z) { VectorXd v(3); v[0] = x; v[1] = y; v[2] = z; return v; };
  TimeableJointSplinePath path(JointPathOptions().set_num_dofs(D).set_num_path_samples(100));
  const std::vector<VectorXd> wps = {V3(1, 2, 3), V3(-1, -2, -3), V3(0.5, 1.0, 1.5)};
  const std::vector<VectorXd> new_wps = {V3(1, 2, 3), V3(0.5, 1.0, 5.5)};   // path_timing_trajectory_test.cc:361-362
  CHECK(path.SwitchToWaypointPath(0.5, {new_wps.data(), new_wps.size()}).code() == StatusCode::kFailedPrecondition);
  CHECK(path.SetWaypoints({wps.data(), wps.size()}).ok());
  const std::vector<double> old_knots = path.knots();
  std::vector<VectorXd> old_points;
  for (int i = 0; i < path.num_control_points(); i++) {
    VectorXd p(D);
    for (int d = 0; d < D; d++) p[d] = path.packed_control_points()[i * D + d];
    old_points.push_back(p);
  }
  EditableBSpline before, after;
  CHECK(before.Init(2, 200, {old_knots.data(), old_knots.size()}, {old_points.data(), old_points.size()}).ok());
  const double keep = 0.37 * old_knots.back();
  CHECK(path.SwitchToWaypointPath(keep, {new_wps.data(), new_wps.size()}).ok());
  CHECK(path.GetState() == TimeablePath::State::kModifiedPath);
  std::vector<VectorXd> new_points;
  for (int i = 0; i < path.num_control_points(); i++) {
    VectorXd p(D);
    for (int d = 0; d < D; d++) p[d] = path.packed_control_points()[i * D + d];
    new_points.push_back(p);
  }
  CHECK((int)path.knots().size() == path.num_control_points() + 3);
  CHECK(after.Init(2, 200, {path.knots().data(), path.knots().size()}, {new_points.data(), new_points.size()}).ok());
  // the kept part of the path is unchanged, the new end is the last new waypoint
  for (double u = 0.0; u < keep; u += keep / 200) {
    VectorXd a, b;
    CHECK(before.EvalCurve(u, &a).ok() && after.EvalCurve(u, &b).ok());
    if (!Near(b, a, 1e-10)) { CHECK(false); break; }
  }
  VectorXd end;
  CHECK(after.EvalCurve(path.knots().back(), &end).ok());
  CHECK(Near(end, new_wps.back(), 1e-12));
  CHECK(path.knots().back() > keep);
  CHECK(!path.CloseToEnd(keep));
}

static void TestBruteForceLp() {   // time_optimal_path_timing.cc:1010-1103 on the mirror vs the oracle's
  unsigned long long seed = 7;
  auto rnd = [&]() { seed = seed * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(seed >> 11) / 9007199254740992.0; };
  TimeOptimalPathProfile prof;
  for (int it = 0; it < 300; it++) {
    const int C = 2 + (it % 29);
    TimeOptimalPathProfile::Constraint c;
    c.resize(C);
    for (int i = 0; i < C; i++) {
      c.a_coefficient(i) = (it % 7 == 0 && i % 2) ? 0.0 : 200 * rnd() - 100;
      c.b_coefficient(i) = 200 * rnd() - 100;
      c.lower(i) = -10 * rnd();
      c.upper(i) = 10 * rnd();
    }
    double m, x, z, om, ox, oz;
    prof.FindMaxSd2BruteForce(c, &m, &x, &z);
    tpo_find_max_sd2_bruteforce(c.a_coefficient(), c.b_coefficient(), c.lower(), c.upper(), C, &om, &ox, &oz);
    CHECK(m == om && x == ox && z == oz);
  }
}

// The reference's corner-rounding tests for poses (splines/spline_utils_test.cc:31-146) as data:
// tests/golden/spline_utils_golden.json, flattened to numbers by tests/test_host_cpu.py. The mirror's
// pose variant (host/timeable_path_cartesian_spline.cc) must reproduce every pinned control pose
// (translation and rotation about (1, 2, 3) / |(1, 2, 3)|), and the vector variant
// (TimeableJointSplinePath::PolyLineToControlPoints) the translations of the rotation-free cases.
static void TestCornerRoundingFixtures(const char *file) {
  using tpamd::compat::AngleAxisd;
  using tpamd::compat::Pose3d;
  using tpamd::compat::Quaterniond;
  using tpamd::compat::Vector3d;
  FILE *f = std::fopen(file, "r");
  CHECK(f != nullptr);
  if (!f) return;
  const double an = std::sqrt(14.0);
  auto make = [&](const double *v) {
    AngleAxisd aa;
    aa.axis = Vector3d(1.0 / an, 2.0 / an, 3.0 / an);
    aa.angle = v[3];
    return Pose3d(aa.toQuaternion(), Vector3d(v[0], v[1], v[2]));
  };
  auto close = [](const Pose3d &a, const Pose3d &b) {
    double e = 0.0;
    for (int d = 0; d < 3; d++) e = std::max(e, std::fabs(a.translation()[d] - b.translation()[d]));
    const Quaterniond &p = a.quaternion(), &q = b.quaternion();
    const double dot = std::fabs(p.w * q.w + p.x * q.x + p.y * q.y + p.z * q.z);   // q and -q are one rotation
    return e < 1e-9 && std::fabs(dot - 1.0) < 1e-12;
  };
  int ncases = 0, checked = 0;
  CHECK(std::fscanf(f, "%d", &ncases) == 1);
  for (int c = 0; c < ncases; c++) {
    int nc = 0, ncp = 0, nexp = 0;
    double tr = 0, rr = 0;
    CHECK(std::fscanf(f, "%d %lf %lf %d", &nc, &tr, &rr, &ncp) == 4);
    std::vector<Pose3d> corners;
    std::vector<VectorXd> vec_corners;
    bool rotation_free = true;
    for (int i = 0; i < nc; i++) {
      double v[4];
      CHECK(std::fscanf(f, "%lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3]) == 4);
      corners.push_back(make(v));
      vec_corners.push_back(VectorXd{v[0], v[1], v[2]});
      rotation_free = rotation_free && v[3] == 0.0;
    }
    std::vector<Pose3d> out;
    PolyLineToBspline3Waypoints(corners, tr, rr, &out);
    CHECK((int)out.size() == ncp);
    for (int i = 0; i < nc && (int)out.size() == ncp && nc > 1; i++) CHECK(close(out[3 * i], corners[i]));
    if (nc == 1) for (const Pose3d &p : out) CHECK(close(p, corners[0]));
    std::vector<VectorXd> vout;
    if (rotation_free) TimeableJointSplinePath::PolyLineToControlPoints(vec_corners, tr, &vout);
    CHECK(std::fscanf(f, "%d", &nexp) == 1);
    for (int k = 0; k < nexp; k++) {
      int idx = 0;
      double v[4];
      CHECK(std::fscanf(f, "%d %lf %lf %lf %lf", &idx, &v[0], &v[1], &v[2], &v[3]) == 5);
      if (idx < (int)out.size()) CHECK(close(out[idx], make(v)));
      if (rotation_free && idx < (int)vout.size())
        for (int d = 0; d < 3; d++) CHECK(std::fabs(vout[idx][d] - v[d]) < 1e-9);
      checked++;
    }
    if (nc == 2 && tr == 0.0) { CHECK(close(out[1], corners[0])); CHECK(close(out[2], corners[1])); }   // ZeroRadius
  }
  std::fclose(f);
  CHECK(ncases == 5 && checked == 10);
}

int main(int argc, char **argv) {
  if (argc > 1) TestCornerRoundingFixtures(argv[1]);
  TestInsertKnot();
  TestTruncate();
  TestExtend();
  TestProjectPointOnPath();
  TestSwitchToWaypointPath();
  TestBruteForceLp();
  if (g_fail == 0) std::printf("ALL OK\n");
  else std::printf("%d CHECKS FAILED\n", g_fail);
  return g_fail == 0 ? 0 : 1;
}
