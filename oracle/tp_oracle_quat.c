/*
 * tp_oracle_quat.c -- CPU restatement of the quaternion B-spline evaluation of the reference
 * (test infrastructure, like the rest of oracle/): splines/bsplineq.cc
 *   NormalizeIfNecessaryAndEnsurePositiveReal  :98-108
 *   QuatLog :112-125, QuatExp :127-134, QuatPower :136-146
 *   BSplineQ::EvalCurve :223-244 with UpdateCumulativeBasis :309-317
 * and the pose sampling of TimeableCartesianSplinePath::SamplePath
 * (timeable_path_cartesian_spline.cc:484-503; BSplineT::EvalCurve splines/bspline.h:512-536).
 *
 * Quaternions are [w, x, y, z]. The reference computes with Eigen 3.4.0 (absent from this
 * image): Quaternion product / inverse / normalize and VectorBlock::stableNorm /
 * stableNormalized are restated from Eigen's published scalar algorithms
 * (Eigen/src/Geometry/Quaternion.h quat_product, QuaternionBase::inverse, ::normalize;
 * Eigen/src/Core/StableNorm.h, Dot.h stableNormalized). Eigen's vectorised builds sum in a
 * different order: ulp-level parity with the reference binary is NOT pinned. What pins this
 * file: the reference's Mathematica table for QuatExp, its Exp/Log round trips, and its
 * slerp-equivalence tests (splines/bsplineq_test.cc:99-198, :309-344, :805-867), reproduced in
 * tests/test_oracle_quat.py at the reference's tolerances.
 */
#include <math.h>
#include <string.h>

#include "tp_oracle.h"

#define EIGEN_DUMMY_PRECISION 1e-12
#define TPO_MAX_DEGREE 15 /* as tp_oracle.c */ /* NumTraits<double>::dummy_precision() */

static double sq_norm4(const double *q) { return q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0]; }

/* Eigen quat_product (scalar path), a * b */
static void quat_mul(const double *a, const double *b, double *out) {
  double r[4];
  r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  memcpy(out, r, sizeof(r));
}

/* QuaternionBase::inverse: conjugate / squaredNorm (zero quaternion for n2 == 0) */
static void quat_inverse(const double *q, double *out) {
  const double n2 = sq_norm4(q);
  if (n2 > 0.0) {
    out[0] = q[0] / n2; out[1] = -q[1] / n2; out[2] = -q[2] / n2; out[3] = -q[3] / n2;
  } else {
    out[0] = out[1] = out[2] = out[3] = 0.0;
  }
}

/* StableNorm.h for a 3-vector: one block, scaled by its largest magnitude */
static double stable_norm3(const double *v) {
  double mx = fabs(v[0]);
  if (fabs(v[1]) > mx) mx = fabs(v[1]);
  if (fabs(v[2]) > mx) mx = fabs(v[2]);
  if (!(mx > 0.0)) return mx;           /* zero (or NaN) vector */
  const double inv = 1.0 / mx;
  const double a = v[0] * inv, b = v[1] * inv, c = v[2] * inv;
  return mx * sqrt(a * a + b * b + c * c);
}

/* Dot.h stableNormalized: (v / w) / sqrt(|v / w|^2) with w = max |v_i|; v itself if that is 0 */
static void stable_normalized3(const double *v, double *out) {
  double w = fabs(v[0]);
  if (fabs(v[1]) > w) w = fabs(v[1]);
  if (fabs(v[2]) > w) w = fabs(v[2]);
  const double a = v[0] / w, b = v[1] / w, c = v[2] / w;
  const double z = a * a + b * b + c * c;
  if (z > 0.0) {
    const double s = sqrt(z);
    out[0] = a / s; out[1] = b / s; out[2] = c / s;
  } else {
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
  }
}

/* bsplineq.cc:98-108 */
static void normalize_if_necessary_positive_real(double *q) {
  if (q[0] < 0) { q[0] *= -1.0; q[1] *= -1.0; q[2] *= -1.0; q[3] *= -1.0; }
  if (fabs(sq_norm4(q) - 1.0) > EIGEN_DUMMY_PRECISION) {
    const double n = sqrt(sq_norm4(q));     /* Quaternion::normalize: coeffs / norm */
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
  }
}

/* bsplineq.cc:112-125 */
void tpo_quat_log(const double *q, double *out) {
  const double squared_norm_q = sq_norm4(q);
  const double norm_v = stable_norm3(q + 1);
  double r[4];
  r[0] = 0.5 * log(squared_norm_q);
  if (norm_v > EIGEN_DUMMY_PRECISION) {
    double n[3];
    stable_normalized3(q + 1, n);
    const double ang = atan2(norm_v, q[0]);
    r[1] = n[0] * ang; r[2] = n[1] * ang; r[3] = n[2] * ang;
  } else {
    r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
  }
  memcpy(out, r, sizeof(r));
}

/* bsplineq.cc:127-134 */
void tpo_quat_exp(const double *q, double *out) {
  const double norm_v = stable_norm3(q + 1);
  double r[4], n[3];
  r[0] = cos(norm_v);
  stable_normalized3(q + 1, n);
  const double s = sin(norm_v);
  r[1] = n[0] * s; r[2] = n[1] * s; r[3] = n[2] * s;
  const double e = exp(q[0]);
  r[0] *= e; r[1] *= e; r[2] *= e; r[3] *= e;
  memcpy(out, r, sizeof(r));
}

/* bsplineq.cc:136-146: q^p = exp(p log q) */
void tpo_quat_power(const double *q, double power, double *out) {
  double in[4], l[4];
  memcpy(in, q, sizeof(in));
  normalize_if_necessary_positive_real(in);
  tpo_quat_log(in, l);
  l[0] *= power; l[1] *= power; l[2] *= power; l[3] *= power;
  tpo_quat_exp(l, out);
}

/* bsplineq.cc:223-244; points [num_points][4]. Returns 1 if u is outside the knot range. */
int tpo_bsplineq_eval_curve(const double *knots, int num_knots, int degree, const double *points,
                            double u, double *quat) {
  if (u < knots[0] || u > knots[num_knots - 1]) return 1;
  const int span = tpo_knot_span(knots, num_knots, degree, u);
  double basis[TPO_MAX_DEGREE + 1], cum[TPO_MAX_DEGREE + 1];
  tpo_basis(knots, span, degree, u, basis);
  /* :309-317: cumulative basis functions that are not identically 0 or 1 */
  if (degree >= 1) {
    cum[degree - 1] = basis[degree];
    for (int i = degree - 2; i >= 0; --i) cum[i] = cum[i + 1] + basis[i + 1];
  }
  double q[4];
  memcpy(q, points + (size_t)(span - degree) * 4, sizeof(q));
  for (int i = 0; i < degree; ++i) {
    double inv[4], rel[4], pw[4];
    quat_inverse(points + (size_t)(span - degree + i) * 4, inv);
    quat_mul(inv, points + (size_t)(span - degree + i + 1) * 4, rel);
    tpo_quat_power(rel, cum[i], pw);
    quat_mul(q, pw, q);
  }
  normalize_if_necessary_positive_real(q);
  memcpy(quat, q, sizeof(q));
  return 0;
}

/* The pose targets TimeableCartesianSplinePath::SamplePath hands to its IK callback
 * (timeable_path_cartesian_spline.cc:484-503), for the parameters path_start + i*delta:
 * translation spline (degree-2 BSpline3d) and rotation spline (degree-2 BSplineQ) on a shared
 * knot vector; beyond knots.back() - delta the last control pose is repeated.
 * poses [N][7] = (tx, ty, tz, qw, qx, qy, qz). */
int tpo_sample_pose_spline(const double *knots, int num_knots, const double *translation_points,
                           const double *rotation_points, int num_points, double path_start,
                           double delta, int N, double *poses) {
  const double kend = knots[num_knots - 1];
  for (int i = 0; i < N; i++) {
    const double parameter = path_start + i * delta;
    double *out = poses + (size_t)i * 7;
    if (parameter < kend - delta) {
      if (tpo_eval_curve(knots, num_knots, 2, translation_points, 3, parameter, out) != 0) return 1;
      if (tpo_bsplineq_eval_curve(knots, num_knots, 2, rotation_points, parameter, out + 3) != 0) return 1;
    } else {
      memcpy(out, translation_points + (size_t)(num_points - 1) * 3, sizeof(double) * 3);
      memcpy(out + 3, rotation_points + (size_t)(num_points - 1) * 4, sizeof(double) * 4);
    }
  }
  return 0;
}
