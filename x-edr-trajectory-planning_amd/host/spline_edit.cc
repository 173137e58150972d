#include "spline_edit.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace trajectory_planning {

using ::tpamd::compat::FailedPreconditionError;
using ::tpamd::compat::InvalidArgumentError;
using ::tpamd::compat::OkStatus;
using ::tpamd::compat::OutOfRangeError;
using ::tpamd::compat::StatusOr;
using ::tpamd::compat::UnimplementedError;

Status EditableBSpline::Init(int degree, int knot_capacity, Span<const double> knots,
                             Span<const VectorXd> points) {
  if (degree < 0) return InvalidArgumentError("degree must be >= 0");
  if ((int)knots.size() < MinNumKnots(degree)) return OutOfRangeError("too few knots");
  if ((int)knots.size() > knot_capacity) return OutOfRangeError("Too many knots");
  if ((int)points.size() != NumPoints((int)knots.size(), degree))
    return InvalidArgumentError("Wrong number of control points");
  for (size_t i = 1; i < knots.size(); i++)
    if (knots[i] < knots[i - 1]) return InvalidArgumentError("Knot vector must be non-decreasing");
  degree_ = degree;
  knot_capacity_ = knot_capacity;
  knots_.assign(knots.begin(), knots.end());
  points_.assign(points.begin(), points.end());
  umin_ = knots_.front();
  umax_ = knots_.back();
  return OkStatus();
}

// lower_bound with a "<=" comparator over knots[degree .. num_knots-degree): the first knot
// strictly greater than u, minus one; u equal to the last knot maps to the last span.
size_t EditableBSpline::KnotSpan(double u) const {
  const int nk = (int)knots_.size();
  if (nk == 0) return 0;
  if (u == knots_[nk - 1]) return (size_t)(nk - degree_ - 2);
  int lo = degree_, hi = nk - degree_;
  while (lo < hi) {
    const int mid = lo + (hi - lo) / 2;
    if (knots_[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return (size_t)(lo - 1);
}

void EditableBSpline::Basis(size_t span, double u, double *N) const {
  std::vector<double> left(degree_ + 1), right(degree_ + 1);
  N[0] = 1.0;
  for (int j = 1; j <= degree_; j++) {
    left[j] = u - knots_[span + 1 - j];
    right[j] = knots_[span + j] - u;
  }
  for (int j = 1; j <= degree_; j++) {
    double saved = 0.0;
    for (int r = 0; r < j; r++) {
      const double tmp = N[r] / (right[r + 1] + left[j - r]);
      N[r] = saved + right[r + 1] * tmp;
      saved = left[j - r] * tmp;
    }
    N[j] = saved;
  }
}

Status EditableBSpline::EvalCurve(double u, VectorXd *value) const {
  if (knots_.empty() || u < umin_ || u > umax_) return OutOfRangeError("Spline parameter outside the valid range");
  const size_t span = KnotSpan(u);
  std::vector<double> N(degree_ + 1);
  Basis(span, u, N.data());
  const size_t dim = points_.front().size();
  *value = VectorXd(dim);
  value->setZero();
  for (int i = 0; i <= degree_; i++) {
    const VectorXd &p = points_[span - degree_ + i];
    for (size_t d = 0; d < dim; d++) (*value)[d] += N[i] * p[d];
  }
  return OkStatus();
}

Status EditableBSpline::CanInsertKnot(double knot, int multiplicity) const {
  if (multiplicity > degree_ + 1) return InvalidArgumentError("Knot multiplicity > degree + 1 not supported.");
  if ((int)knots_.size() + multiplicity > knot_capacity_) return FailedPreconditionError("Knot capacity too small");
  if (knots_.size() < 2) return FailedPreconditionError("Set initial knot vector first.");
  if (knot <= knots_.front() || knot >= knots_.back())
    return InvalidArgumentError("knot not in range of current knots");
  return OkStatus();
}

Status EditableBSpline::InsertKnotAndUpdateControlPoints(double knot, int multiplicity) {
  if (Status st = CanInsertKnot(knot, multiplicity); !st.ok()) return st;
  if (degree_ < 1) return UnimplementedError("Not implemented for splines of degree 0.");
  const size_t dim = points_.front().size();
  for (int m = 0; m < multiplicity; m++) {
    const int span = (int)KnotSpan(knot);
    // new `degree` control points for the refined knot vector (bspline.h:262-266)
    std::vector<VectorXd> fresh(degree_, VectorXd(dim));
    for (int i = 0; i < degree_; i++) {
      const int k = span + i - degree_ + 1;
      const double alpha = (knot - knots_[k]) / (knots_[k + degree_] - knots_[k]);
      for (size_t d = 0; d < dim; d++) fresh[i][d] = alpha * points_[k][d] + (1.0 - alpha) * points_[k - 1][d];
    }
    // points span .. end move up by one; span-degree+1 .. span take the new values
    points_.insert(points_.begin() + span, points_[span]);
    for (int i = 0; i < degree_; i++) points_[span - degree_ + 1 + i] = fresh[i];
    knots_.insert(knots_.begin() + span + 1, knot);       // InsertKnotIntoKnotVector, bspline_base.cc:197-213
  }
  return OkStatus();
}

Status EditableBSpline::TruncateSplineAt(double u_end) {
  if (u_end >= umax_) return OkStatus();
  if (u_end <= umin_) {             // the curve is cleared
    umin_ = std::numeric_limits<double>::infinity();
    umax_ = -std::numeric_limits<double>::infinity();
    knots_.clear();
    points_.clear();
    return OkStatus();
  }
  // degree+1 equal knots decouple the curve at u_end; the second section is dropped
  if (Status st = InsertKnotAndUpdateControlPoints(u_end, degree_ + 1); !st.ok()) return st;
  const int span = (int)KnotSpan(u_end);
  knots_.resize(span + 1);
  points_.resize(NumPoints((int)knots_.size(), degree_));
  umax_ = u_end;
  return OkStatus();
}

Status EditableBSpline::ExtendWithControlPoints(Span<const VectorXd> points) {
  if (degree_ != 2) return UnimplementedError("Only implemented for 2nd order splines.");
  const int num_knots = (int)knots_.size(), num_points = (int)points_.size();
  const int new_num_points = num_points + (int)points.size();
  const int added_knots = NumKnots((int)points.size() + 1, degree_) - 2 * degree_;
  const int new_num_knots = num_knots + added_knots;
  if (num_knots < MinNumKnots(degree_)) return FailedPreconditionError("Spline is empty or invalid.");
  if (new_num_knots > knot_capacity_ || new_num_points > NumPoints(knot_capacity_, degree_))
    return FailedPreconditionError("Knot capacity too small to append all points.");
  if (points.size() < 2) return UnimplementedError("Only implemented for >= 2 points.");
  const double u_join = knots_[num_knots - 1];
  // uniform knots for the new control points with the density of the existing ones
  const double old_knot_range = knots_[num_knots - 1] - knots_[0];
  const int old_inner = num_knots - 2 * degree_ - 1;
  const int new_inner = new_num_knots - 2 * degree_ - 1;
  const double new_knot_range = (old_knot_range * new_inner) / old_inner;
  const int lin_start = num_knots - degree_ - 1;
  const int lin_size = (new_num_knots - degree_) - lin_start;
  knots_.resize(new_num_knots);
  // Eigen setLinSpaced(size, low, high): low + i * (high - low) / (size - 1), the last one = high
  for (int i = 0; i < lin_size; i++) {
    const double step = (lin_size > 1) ? (new_knot_range - old_knot_range) / (lin_size - 1) : 0.0;
    knots_[lin_start + i] = (i == lin_size - 1) ? new_knot_range : old_knot_range + i * step;
  }
  for (int i = 0; i <= degree_; i++) knots_[new_num_knots - degree_ - 1 + i] = knots_[0] + new_knot_range;
  umax_ = knots_[new_num_knots - 1];
  // the old end point moves so that the curve still passes through it at u_join (:486-503)
  const int modified = num_points - 1;
  points_.resize(new_num_points, VectorXd(points_.front().size()));
  const size_t span = KnotSpan(u_join);
  double N[3];
  Basis(span, u_join, N);
  if (!(N[1] > 0)) return FailedPreconditionError("degenerate joint: basis[1] must be > 0");
  const size_t dim = points_.front().size();
  VectorXd moved(dim);
  for (size_t d = 0; d < dim; d++)
    moved[d] = 1.0 / N[1] * (points_[modified][d] - N[0] * points_[modified - 1][d]);
  points_[modified] = moved;
  for (size_t i = 0; i < points.size(); i++) points_[num_points + i] = points[i];
  return OkStatus();
}

// Closest point of the segment a-b (eigenmath::DistanceFromLineSegment is outside the
// reference tree; restated: orthogonal projection with the parameter limited to the segment's
// end, a negative parameter is kept so that callers can tell "before the first waypoint",
// timeable_path_joint_spline.cc:232-239).
StatusOr<ProjectedPointResult> ProjectPointOnPath(Span<const VectorXd> waypoints, const VectorXd &point) {
  if (waypoints.empty()) return InvalidArgumentError("No waypoints.");
  ProjectedPointResult res;
  const size_t dim = point.size();
  for (const auto &wp : waypoints)
    if (wp.size() != dim) return InvalidArgumentError("Invalid number of joints");
  auto dist = [&](const VectorXd &a) {
    double s = 0.0;
    for (size_t d = 0; d < dim; d++) s += (a[d] - point[d]) * (a[d] - point[d]);
    return std::sqrt(s);
  };
  if (waypoints.size() == 1) {
    res.distance_to_path = dist(waypoints[0]);
    res.projected_point = waypoints[0];
    return res;
  }
  res.distance_to_path = std::numeric_limits<double>::max();
  for (size_t i = 0; i + 1 < waypoints.size(); i++) {
    const VectorXd &a = waypoints[i], &b = waypoints[i + 1];
    double ab2 = 0.0, ap_ab = 0.0;
    for (size_t d = 0; d < dim; d++) {
      ab2 += (b[d] - a[d]) * (b[d] - a[d]);
      ap_ab += (point[d] - a[d]) * (b[d] - a[d]);
    }
    double t = ab2 > 0.0 ? ap_ab / ab2 : 0.0;
    if (t > 1.0) t = 1.0;
    const double tc = t < 0.0 ? 0.0 : t;
    VectorXd c(dim);
    for (size_t d = 0; d < dim; d++) c[d] = a[d] + tc * (b[d] - a[d]);
    const double dd = dist(c);
    if (dd < res.distance_to_path) {
      res.distance_to_path = dd;
      res.line_parameter = t;
      res.waypoint_index = (int)i;
    }
  }
  const VectorXd &a = waypoints[res.waypoint_index], &b = waypoints[res.waypoint_index + 1];
  res.projected_point = VectorXd(dim);
  for (size_t d = 0; d < dim; d++) res.projected_point[d] = a[d] + res.line_parameter * (b[d] - a[d]);
  return res;
}

}  // namespace trajectory_planning
