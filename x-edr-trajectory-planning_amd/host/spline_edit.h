// Host-side edits of a B-spline path that is already being followed:
// knot insertion, truncation and extension of the reference's BSplineT
// (trajectory_planning/splines/bspline.h:222-511) for VectorXd control points. These are O(P)
// operations on a few dozen control points and stay on the host, as in the reference; their
// result (knots + control points) is what the batched GPU entry points take as input.
#ifndef TPAMD_HOST_SPLINE_EDIT_H_
#define TPAMD_HOST_SPLINE_EDIT_H_

#include <vector>

#include "compat.h"

namespace trajectory_planning {

using ::tpamd::compat::Span;
using ::tpamd::compat::Status;
using ::tpamd::compat::VectorXd;

class EditableBSpline {
 public:
  // splines/bspline_base.h: NumKnots / NumPoints / MinNumKnots
  static int NumKnots(int num_points, int degree) { return num_points + degree + 1; }
  static int NumPoints(int num_knots, int degree) { return num_knots - degree - 1; }
  static int MinNumKnots(int degree) { return 2 * (degree + 1); }

  // knot_capacity bounds the number of knots the edits may produce (the reference allocates
  // its arrays once in Init, splines/bspline_base.cc:40-75).
  Status Init(int degree, int knot_capacity, Span<const double> knots, Span<const VectorXd> points);
  int degree() const { return degree_; }
  const std::vector<double> &knots() const { return knots_; }
  const std::vector<VectorXd> &control_points() const { return points_; }
  bool empty() const { return knots_.empty(); }
  double umin() const { return umin_; }
  double umax() const { return umax_; }

  size_t KnotSpan(double u) const;                                     // bspline_base.cc:218-246
  Status EvalCurve(double u, VectorXd *value) const;                   // bspline.h:512-536
  // bspline.h:222-232 through the closed form of :236-272 (NURBS A5.1). The reference's default
  // variant (:274-402) solves the same `degree` x dim equations "curve unchanged at `degree`
  // parameter values" numerically (Eigen colPivHouseholderQr) for the same unknowns; both keep
  // the curve and differ by that solver's rounding.
  Status InsertKnotAndUpdateControlPoints(double knot, int multiplicity);
  Status TruncateSplineAt(double u_end);                               // bspline.h:404-428
  Status ExtendWithControlPoints(Span<const VectorXd> points);         // bspline.h:430-511

 private:
  Status CanInsertKnot(double knot, int multiplicity) const;           // bspline_base.cc:166-195
  void Basis(size_t span, double u, double *N) const;                  // bspline_base.cc:249-265
  int degree_ = 0, knot_capacity_ = 0;
  std::vector<double> knots_;
  std::vector<VectorXd> points_;
  double umin_ = 0, umax_ = 0;
};

// path_tools.h:26-100: the point of a polyline closest to `point`.
struct ProjectedPointResult {
  int waypoint_index = 0;
  double distance_to_path = 0;
  double line_parameter = 0;
  VectorXd projected_point;
};
::tpamd::compat::StatusOr<ProjectedPointResult> ProjectPointOnPath(Span<const VectorXd> waypoints,
                                                                   const VectorXd &point);

}  // namespace trajectory_planning

#endif  // TPAMD_HOST_SPLINE_EDIT_H_
