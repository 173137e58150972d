/*
 * tp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See tp_oracle.h for the pinning status and the usage rule.
 *
 * The arithmetic below keeps the reference's operation order (no FMA
 * contraction: build with -ffp-contract=off) because comparisons against
 * kTiny decide control flow in the solver.
 */
#include "tp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TPO_MAX_DEGREE 15

static int tiny(double v) { return fabs(v) < TPO_KTINY; }

/* ======================================================================= */
/*                                 splines                                  */
/* ======================================================================= */

/* splines/bspline_base.cc:218-246. The reference runs std::lower_bound with a
 * "<=" comparator over knots[degree .. num_knots-degree): that yields the first
 * knot strictly greater than u; the span is one less. u equal to the last knot
 * maps to the last span (num_points-1). */
int tpo_knot_span(const double *knots, int num_knots, int degree, double u) {
  if (num_knots == 0) return 0;
  if (u == knots[num_knots - 1]) return num_knots - degree - 2;
  int lo = degree, hi = num_knots - degree;
  while (lo < hi) {
    int mid = lo + (hi - lo) / 2;
    if (knots[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

/* splines/bspline_base.cc:249-265 (NURBS book A2.2) */
void tpo_basis(const double *knots, int span, int p, double u, double *basis) {
  double left[TPO_MAX_DEGREE + 1], right[TPO_MAX_DEGREE + 1];
  basis[0] = 1.0;
  for (int j = 1; j <= p; j++) {
    left[j] = u - knots[span + 1 - j];
    right[j] = knots[span + j] - u;
  }
  for (int j = 1; j <= p; j++) {
    double saved = 0.0;
    for (int r = 0; r < j; r++) {
      double tmp = basis[r] / (right[r + 1] + left[j - r]);
      basis[r] = saved + right[r + 1] * tmp;
      saved = left[j - r] * tmp;
    }
    basis[j] = saved;
  }
}

/* splines/bspline_base.cc:268-348 (NURBS book A2.3). ndu holds basis functions
 * in the upper triangle and knot differences in the lower one. */
void tpo_basis_and_derivatives(const double *knots, int span, int p, int der,
                               double u, double *ders) {
  double ndu[TPO_MAX_DEGREE + 1][TPO_MAX_DEGREE + 1];
  double a[2][TPO_MAX_DEGREE + 1];
  double left[TPO_MAX_DEGREE + 1], right[TPO_MAX_DEGREE + 1];
  const int w = p + 1;
  memset(a, 0, sizeof(a));
  ndu[0][0] = 1.0;
  for (int j = 1; j <= p; j++) {
    left[j] = u - knots[span + 1 - j];
    right[j] = knots[span + j] - u;
    double saved = 0.0;
    for (int r = 0; r < j; r++) {
      ndu[j][r] = right[r + 1] + left[j - r];
      const double tmp = ndu[r][j - 1] / ndu[j][r];
      ndu[r][j] = saved + right[r + 1] * tmp;
      saved = left[j - r] * tmp;
    }
    ndu[j][j] = saved;
  }
  for (int j = 0; j <= p; j++) ders[0 * w + j] = ndu[j][p];

  for (int r = 0; r <= p; r++) {
    int s1 = 0, s2 = 1;
    a[0][0] = 1.0;
    for (int k = 1; k <= der; k++) {
      double d = 0.0;
      const int rk = r - k, pk = p - k;
      int j1, j2;
      if (r >= k) {
        a[s2][0] = a[s1][0] / ndu[pk + 1][rk];
        d = a[s2][0] * ndu[rk][pk];
      }
      j1 = (rk >= -1) ? 1 : -rk;
      j2 = (r - 1 <= pk) ? k - 1 : p - r;
      for (int j = j1; j <= j2; j++) {
        a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j];
        d += a[s2][j] * ndu[rk + j][pk];
      }
      if (r <= pk) {
        a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r];
        d += a[s2][k] * ndu[r][pk];
      }
      ders[k * w + r] = d;
      int t = s1; s1 = s2; s2 = t;
    }
  }
  double factor = (double)p;
  for (int k = 1; k <= der; k++) {
    for (int j = 0; j <= p; j++) ders[k * w + j] *= factor;
    factor *= (double)(p - k);
  }
}

/* splines/bspline.h:515-537 (A3.1) */
int tpo_eval_curve(const double *knots, int num_knots, int degree,
                   const double *points, int dim, double u, double *value) {
  if (u < knots[0] || u > knots[num_knots - 1]) return 1;
  double basis[TPO_MAX_DEGREE + 1];
  const int span = tpo_knot_span(knots, num_knots, degree, u);
  tpo_basis(knots, span, degree, u, basis);
  for (int d = 0; d < dim; d++) value[d] = 0.0;
  for (int i = 0; i <= degree; i++) {
    const double *pt = points + (size_t)(span - degree + i) * dim;
    for (int d = 0; d < dim; d++) value[d] += basis[i] * pt[d];
  }
  return 0;
}

/* splines/bspline.h:540-568 (A3.2). The reference asserts der <= degree inside
 * UpdateBasisAndDerivatives (bspline_base.cc:270); here that is error 2. */
int tpo_eval_curve_and_derivatives(const double *knots, int num_knots, int degree,
                                   const double *points, int dim, double u,
                                   int nvalues, double *values) {
  if (nvalues <= 0) return 3;
  if (u < knots[0] || u > knots[num_knots - 1]) return 1;
  const int der = nvalues - 1;
  if (der > degree) return 2;
  double ders[(TPO_MAX_DEGREE + 1) * (TPO_MAX_DEGREE + 1)];
  const int span = tpo_knot_span(knots, num_knots, degree, u);
  tpo_basis_and_derivatives(knots, span, degree, der, u, ders);
  for (int k = 0; k <= der; k++) {
    double *v = values + (size_t)k * dim;
    for (int d = 0; d < dim; d++) v[d] = 0.0;
    for (int j = 0; j <= degree; j++) {
      const double *pt = points + (size_t)(span - degree + j) * dim;
      const double b = ders[k * (degree + 1) + j];
      for (int d = 0; d < dim; d++) v[d] += b * pt[d];
    }
  }
  return 0;
}

/* splines/bspline_base.cc:350-381 */
int tpo_make_uniform_knots(int num_points, int degree, double low, double high,
                           double *knots) {
  if (high <= low) return 1;
  if (num_points < degree + 1) return 2;
  const int nknots = num_points + degree + 1;
  for (int i = 0; i <= degree; i++) knots[i] = low;
  const double denom = nknots - 2.0 * (degree + 1.0) + 1.0;
  const double spacing = (1.0 / denom) * (high - low);
  for (int i = degree + 1; i < nknots - degree - 1; i++)
    knots[i] = knots[i - 1] + spacing;
  for (int i = nknots - degree - 1; i < nknots; i++) knots[i] = high;
  return 0;
}

/* Euclidean norm, sequential sum. (Eigen's norm() uses a build-dependent
 * packet reduction order; this host-side pre-processing is an INPUT generator
 * for the hot path, so the order is fixed here and documented in DESIGN.md.) */
static double vec_norm(const double *v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  return sqrt(s);
}

/* splines/spline_utils.cc:25-45 */
static void corner_offset(const double *delta, int D, double radius, double *offset) {
  const double kMinNorm = 1e-6;
  const double kMinWaypointSpacingFactor = 4.0; /* spline_utils.h:44-46 */
  const double norm = vec_norm(delta, D);
  if (norm > kMinNorm) {
    for (int d = 0; d < D; d++) offset[d] = delta[d] / norm;
  } else {
    for (int d = 0; d < D; d++) offset[d] = 0.0;
  }
  if (norm > kMinWaypointSpacingFactor * radius) {
    for (int d = 0; d < D; d++) offset[d] = offset[d] * radius;
  } else {
    for (int d = 0; d < D; d++)
      offset[d] = offset[d] * (1.0 / kMinWaypointSpacingFactor) * norm;
  }
}

/* splines/spline_utils.cc:47-102 */
int tpo_polyline_to_bspline3_waypoints(const double *corners, int W, int D,
                                       double radius, double *out) {
  if (W == 1) {
    for (int k = 0; k < 4; k++) memcpy(out + (size_t)k * D, corners, sizeof(double) * D);
    return 4;
  }
  const int P = 3 * W - 2;
  double *delta = (double *)malloc(sizeof(double) * 2 * D);
  double *offset = delta + D;
  for (int i = 0; i < W; i++)
    memcpy(out + (size_t)(3 * i) * D, corners + (size_t)i * D, sizeof(double) * D);
#define PT(k) (out + (size_t)(k) * D)
  for (int i = 1; i < W - 1; i++) {
    const int k = 3 * i, knext = 3 * (i + 1), klast = 3 * (i - 1);
    for (int d = 0; d < D; d++) delta[d] = PT(knext)[d] - PT(k)[d];
    corner_offset(delta, D, radius, offset);
    for (int d = 0; d < D; d++) PT(k + 1)[d] = PT(k)[d] + offset[d];
    for (int d = 0; d < D; d++) delta[d] = PT(klast)[d] - PT(k)[d];
    corner_offset(delta, D, radius, offset);
    for (int d = 0; d < D; d++) PT(k - 1)[d] = PT(k)[d] + offset[d];
  }
  for (int d = 0; d < D; d++) delta[d] = PT(3)[d] - PT(0)[d];
  corner_offset(delta, D, radius, offset);
  for (int d = 0; d < D; d++) PT(1)[d] = PT(0)[d] + offset[d];
  for (int d = 0; d < D; d++) delta[d] = PT(P - 4)[d] - PT(P - 1)[d];
  corner_offset(delta, D, radius, offset);
  for (int d = 0; d < D; d++) PT(P - 2)[d] = PT(P - 1)[d] + offset[d];
#undef PT
  free(delta);
  return P;
}

/* ======================================================================= */
/*                             joint-space path                             */
/* ======================================================================= */

/* timeable_path_joint_spline.cc:252-292 (kSplineOrder = 2, .h:93) */
int tpo_joint_fit_spline(const double *waypoints, int W, int D, double rounding,
                         double *control_points, double *knots) {
  const int P = tpo_polyline_to_bspline3_waypoints(waypoints, W, D, rounding, control_points);
  const int nk = P + 2 + 1;
  tpo_make_uniform_knots(P, 2, 0.0, 1.0, knots);
  double length = 0.0;
  double *diff = (double *)malloc(sizeof(double) * D);
  for (int i = 0; i < P - 1; i++) {
    for (int d = 0; d < D; d++)
      diff[d] = control_points[(size_t)(i + 1) * D + d] - control_points[(size_t)i * D + d];
    length += vec_norm(diff, D);
  }
  free(diff);
  const double kMinimumFinalKnotValue = 0.1;
  const double kPathParameterPerPolygonLength = 1.0;
  double weighted = length * kPathParameterPerPolygonLength;
  if (weighted < kMinimumFinalKnotValue) weighted = kMinimumFinalKnotValue;
  for (int i = 0; i < nk; i++) knots[i] *= weighted;
  return P;
}

/* timeable_path_joint_spline.cc:294-318 */
int tpo_joint_sample_path(const double *knots, int num_knots,
                          const double *control_points, int num_points, int D,
                          double path_start, double delta, int N, double *q,
                          double *q1, double *q2) {
  const double k0 = knots[0], kend = knots[num_knots - 1];
  double *vals = (double *)malloc(sizeof(double) * 3 * D);
  for (int idx = 0; idx < N; idx++) {
    const double parameter = path_start + idx * delta;
    double *pq = q + (size_t)idx * D, *pq1 = q1 + (size_t)idx * D, *pq2 = q2 + (size_t)idx * D;
    if (parameter < kend + delta) {
      double u = parameter;
      if (u < k0) u = k0;      /* std::clamp */
      if (kend < u) u = kend;
      int rc = tpo_eval_curve_and_derivatives(knots, num_knots, 2, control_points, D, u, 3, vals);
      if (rc != 0) { free(vals); return rc; }
      memcpy(pq, vals, sizeof(double) * D);
      memcpy(pq1, vals + D, sizeof(double) * D);
      memcpy(pq2, vals + 2 * D, sizeof(double) * D);
    } else {
      memcpy(pq, control_points + (size_t)(num_points - 1) * D, sizeof(double) * D);
      for (int d = 0; d < D; d++) { pq1[d] = 0.0; pq2[d] = 0.0; }
    }
  }
  free(vals);
  return 0;
}

/* timeable_path_joint_spline.cc:320-343 */
void tpo_joint_constraint_setup(const double *q1, const double *q2, int N, int D,
                                const double *vmax, const double *amax,
                                double safety, double *A, double *B,
                                double *lower, double *upper) {
  const int C = 2 * D;
  for (int idx = 0; idx < N; idx++) {
    double *a = A + (size_t)idx * C, *b = B + (size_t)idx * C;
    double *lo = lower + (size_t)idx * C, *hi = upper + (size_t)idx * C;
    const double *d1 = q1 + (size_t)idx * D, *d2 = q2 + (size_t)idx * D;
    for (int dof = 0; dof < D; dof++) {
      a[dof] = d1[dof];
      b[dof] = d2[dof];
      hi[dof] = amax[dof] * safety;
      lo[dof] = -amax[dof] * safety;
      a[D + dof] = 0.0;
      b[D + dof] = d1[dof] * d1[dof];             /* std::pow(x, 2) */
      const double v = vmax[dof] * safety;
      hi[D + dof] = v * v;                         /* std::pow(x, 2) */
      lo[D + dof] = 0.0;
    }
  }
}

/* timeable_path_cartesian_spline.cc:39-68 */
void tpo_cartesian_path_derivatives(const double *q, int N, int D, double delta,
                                    double *q1, double *q2) {
  const double inv = 1.0 / delta;
  for (int i = 0; i < N - 1; i++)
    for (int d = 0; d < D; d++)
      q1[(size_t)i * D + d] = inv * (q[(size_t)(i + 1) * D + d] - q[(size_t)i * D + d]);
  for (int d = 0; d < D; d++) q1[(size_t)(N - 1) * D + d] = 0.0;
  for (int i = 1; i < N - 1; i++)
    for (int d = 0; d < D; d++)
      q2[(size_t)i * D + d] = inv * (q1[(size_t)(i + 1) * D + d] - q1[(size_t)i * D + d]);
  for (int d = 0; d < D; d++) { q2[d] = 0.0; q2[(size_t)(N - 1) * D + d] = 0.0; }
}

/* timeable_path_cartesian_spline.cc:551-595 (jq1 = jacobian * q1 per sample) */
void tpo_cartesian_constraint_setup(const double *q1, const double *q2,
                                    const double *jq1, int N, int D,
                                    const double *vmax, const double *amax,
                                    double max_trans_vel, double max_rot_vel,
                                    double safety, double *A, double *B,
                                    double *lower, double *upper) {
  const int C = 2 * D + 2;
  for (int idx = 0; idx < N; idx++) {
    double *a = A + (size_t)idx * C, *b = B + (size_t)idx * C;
    double *lo = lower + (size_t)idx * C, *hi = upper + (size_t)idx * C;
    const double *d1 = q1 + (size_t)idx * D, *d2 = q2 + (size_t)idx * D;
    const double *v6 = jq1 + (size_t)idx * 6;
    for (int dof = 0; dof < D; dof++) {
      a[dof] = d1[dof];
      b[dof] = d2[dof];
      hi[dof] = amax[dof] * safety;
      lo[dof] = -amax[dof] * safety;
      a[D + dof] = 0.0;
      b[D + dof] = d1[dof] * d1[dof];
      const double v = vmax[dof] * safety;
      hi[D + dof] = v * v;
      lo[D + dof] = 0.0;
    }
    a[2 * D] = 0.0;
    b[2 * D] = (v6[0] * v6[0] + v6[1] * v6[1]) + v6[2] * v6[2];
    hi[2 * D] = max_trans_vel * max_trans_vel;
    lo[2 * D] = -hi[2 * D];
    a[2 * D + 1] = 0.0;
    b[2 * D + 1] = (v6[3] * v6[3] + v6[4] * v6[4]) + v6[5] * v6[5];
    hi[2 * D + 1] = max_rot_vel * max_rot_vel;
    lo[2 * D + 1] = -hi[2 * D + 1];
  }
}

/* ======================================================================= */
/*                                  solver                                  */
/* ======================================================================= */

enum { ST_INVALID = 0, ST_ALLOCATED, ST_DEFINED, ST_SOLVED };
enum { CT_NOTSET = 0, CT_UPPER = 1, CT_LOWER = 2 };

struct tpo_profile {
  int state;
  int C, N;
  double s_start, sd_start, sdd_start, time_start, s_end;
  double *A, *B, *lo, *hi; /* [N][C] */
  /* boundary curve (time_optimal_path_timing.h:225-255) */
  double *sd2_max, *sdd_max, *sdd_min, *sd2_zero;
  uint8_t *at_sdd0, *type;
  int max_loops;
  double *time, *sd2, *sd, *s, *sdd;
  double ds, inv_ds;
  double dt_max;
  int low_idx, high_idx;
  int last_extremal_index;
  int loops_used;
  /* deferred boundary fixes (index_value_work_) */
  int *fix_idx; double *fix_val; int nfix;
};

tpo_profile *tpo_profile_create(int N, int C) {
  tpo_profile *p = (tpo_profile *)calloc(1, sizeof(*p));
  if (!p) return NULL;
  p->N = N; p->C = C; p->max_loops = 100;
  const size_t nc = (size_t)(N > 0 ? N : 1) * (size_t)(C > 0 ? C : 1);
  const size_t n = (size_t)(N > 0 ? N : 1);
  p->A = (double *)calloc(nc, sizeof(double));
  p->B = (double *)calloc(nc, sizeof(double));
  p->lo = (double *)calloc(nc, sizeof(double));
  p->hi = (double *)calloc(nc, sizeof(double));
  p->sd2_max = (double *)calloc(n, sizeof(double));
  p->sdd_max = (double *)calloc(n, sizeof(double));
  p->sdd_min = (double *)calloc(n, sizeof(double));
  p->sd2_zero = (double *)calloc(n, sizeof(double));
  p->at_sdd0 = (uint8_t *)calloc(n, 1);
  p->type = (uint8_t *)calloc(n, 1);
  p->time = (double *)calloc(n, sizeof(double));
  p->sd2 = (double *)calloc(n, sizeof(double));
  p->sd = (double *)calloc(n, sizeof(double));
  p->s = (double *)calloc(n, sizeof(double));
  p->sdd = (double *)calloc(n, sizeof(double));
  p->fix_idx = (int *)calloc(n, sizeof(int));
  p->fix_val = (double *)calloc(n, sizeof(double));
  p->state = ST_ALLOCATED;
  return p;
}

void tpo_profile_destroy(tpo_profile *p) {
  if (!p) return;
  free(p->A); free(p->B); free(p->lo); free(p->hi);
  free(p->sd2_max); free(p->sdd_max); free(p->sdd_min); free(p->sd2_zero);
  free(p->at_sdd0); free(p->type);
  free(p->time); free(p->sd2); free(p->sd); free(p->s); free(p->sdd);
  free(p->fix_idx); free(p->fix_val);
  free(p);
}

void tpo_profile_set_max_loops(tpo_profile *p, int loops) { p->max_loops = loops; }

/* .cc:161-203 then .cc:535-576 */
int tpo_profile_setup(tpo_profile *p, const double *A, const double *B,
                      const double *lower, const double *upper, double s_start,
                      double s_end, double sd_start, double sdd_start,
                      double time_start) {
  const int N = p->N, C = p->C;
  for (int i = 0; i < N; i++) {
    double mx = -DBL_MAX;
    for (int c = 0; c < C; c++) {
      const double w = upper[(size_t)i * C + c] - lower[(size_t)i * C + c];
      if (w > mx) mx = w;
    }
    if (mx <= 0) return TPO_ERR_INFEASIBLE_BOUNDS;
  }
  if (s_start >= s_end) return TPO_ERR_S_RANGE;
  if (sd_start < 0) return TPO_ERR_SD_START_NEG;
  const size_t nc = (size_t)N * C;
  memcpy(p->A, A, nc * sizeof(double));
  memcpy(p->B, B, nc * sizeof(double));
  memcpy(p->lo, lower, nc * sizeof(double));
  memcpy(p->hi, upper, nc * sizeof(double));
  p->s_end = s_end; p->s_start = s_start; p->sd_start = sd_start;
  p->sdd_start = sdd_start; p->time_start = time_start;
  /* IsSetupValid */
  for (size_t k = 0; k < nc; k++)
    if (p->lo[k] >= p->hi[k]) return TPO_ERR_LOWER_GE_UPPER;
  if (N < 2) return TPO_ERR_TOO_FEW_SAMPLES;
  /* SetSetupDone */
  p->ds = (s_end - s_start) / (N - 1);
  p->inv_ds = 1.0 / p->ds;
  for (int i = 0; i < N; i++) p->s[i] = p->ds * i + s_start;
  p->s[N - 1] = s_end;
  p->state = ST_DEFINED;
  return TPO_OK;
}

/* .cc:624-636 on an explicit row set */
static int rows_valid(const double *A, const double *B, const double *lo,
                      const double *hi, int C, double sdd, double sd2) {
  for (int i = 0; i < C; i++) {
    const double v = A[i] * sdd + B[i] * sd2;
    if (v + TPO_KTINY < lo[i] || v - TPO_KTINY > hi[i]) return 0;
  }
  return 1;
}

/* .cc:638-666 */
double tpo_find_sdd_max(const double *A, const double *B, const double *lo,
                        const double *hi, int C, double sd2) {
  double sdd = -DBL_MAX; /* numeric_limits::lowest() */
  for (int i = 0; i < C; i++) {
    if (!tiny(A[i])) {
      double sddi = (lo[i] - B[i] * sd2) / A[i];
      if ((sddi > sdd) && rows_valid(A, B, lo, hi, C, sddi, sd2)) sdd = sddi;
      sddi = (hi[i] - B[i] * sd2) / A[i];
      if ((sddi > sdd) && rows_valid(A, B, lo, hi, C, sddi, sd2)) sdd = sddi;
    }
  }
  if (sdd == -DBL_MAX) sdd = 0;
  return sdd;
}

/* .cc:668-695 */
double tpo_find_sdd_min(const double *A, const double *B, const double *lo,
                        const double *hi, int C, double sd2) {
  double sdd = DBL_MAX;
  for (int i = 0; i < C; i++) {
    if (!tiny(A[i])) {
      double sddi = (lo[i] - B[i] * sd2) / A[i];
      if ((sddi < sdd) && rows_valid(A, B, lo, hi, C, sddi, sd2)) sdd = sddi;
      sddi = (hi[i] - B[i] * sd2) / A[i];
      if ((sddi < sdd) && rows_valid(A, B, lo, hi, C, sddi, sd2)) sdd = sddi;
    }
  }
  if (sdd == DBL_MAX) sdd = 0;
  return sdd;
}

#define ROWS(p, i) (p)->A + (size_t)(i) * (p)->C, (p)->B + (size_t)(i) * (p)->C, \
                   (p)->lo + (size_t)(i) * (p)->C, (p)->hi + (size_t)(i) * (p)->C, (p)->C

static int derivs_valid(const tpo_profile *p, int idx, double sdd, double sd2) {
  return rows_valid(ROWS(p, idx), sdd, sd2);
}
static double sdd_max_at(const tpo_profile *p, int idx, double sd2) {
  return tpo_find_sdd_max(ROWS(p, idx), sd2);
}
static double sdd_min_at(const tpo_profile *p, int idx, double sd2) {
  return tpo_find_sdd_min(ROWS(p, idx), sd2);
}

/* .cc:954-981 */
static int intersect(double A1, double B1, double e1, double A2, double B2,
                     double e2, double *sdd, double *sp2) {
  const double det = A1 * B2 - B1 * A2;
  if (tiny(det)) {
    if (tiny(A1)) {
      *sdd = 0;
      if (tiny(B1)) return 0;
      *sp2 = e1 / B1;
      return 1;
    }
    return 0;
  }
  const double inv_det = 1.0 / det;
  *sdd = (B2 * e1 - B1 * e2) * inv_det;
  *sp2 = (-A2 * e1 + A1 * e2) * inv_det;
  return 1;
}

/* .cc:1010-1103 */
void tpo_find_max_sd2_bruteforce(const double *A, const double *B,
                                 const double *lo, const double *hi, int C,
                                 double *sd2max, double *sddmax, double *sd2zero) {
  *sd2max = 0; *sddmax = 0; *sd2zero = TPO_KMAXSD2;
  for (int c = 0; c < C; c++) {
    if (B[c] > TPO_KTINY) {
      const double tmp = hi[c] / B[c];
      if (tmp < *sd2zero) *sd2zero = tmp;
    } else if (B[c] < -TPO_KTINY) {
      const double tmp = lo[c] / B[c];
      if (tmp < *sd2zero) *sd2zero = tmp;
    }
  }
  for (int c1 = 0; c1 < C; c1++) {
    for (int c2 = c1 + 1; c2 < C; c2++) {
      const double e1[4] = {hi[c1], hi[c1], lo[c1], lo[c1]};
      const double e2[4] = {hi[c2], lo[c2], hi[c2], lo[c2]};
      for (int k = 0; k < 4; k++) {
        double sd2, sdd;
        if (intersect(A[c1], B[c1], e1[k], A[c2], B[c2], e2[k], &sdd, &sd2)) {
          if ((sd2 > *sd2max) && rows_valid(A, B, lo, hi, C, sdd, sd2)) {
            *sd2max = sd2; *sddmax = sdd;
          }
        }
      }
    }
  }
  if (0 == *sd2max || *sd2max > TPO_KMAXSD2) { *sd2max = TPO_KMAXSD2; *sddmax = 0; }
  if (0 == *sd2zero) *sd2zero = TPO_KMAXSD2;
}

/* .cc:1105-1147 : KKT sign test for a pair of active rows */
static int pair_is_optimal(const double *A, const double *B, int first, int second,
                           int first_type, int second_type) {
  const double denom = A[second] * B[first] - A[first] * B[second];
  if (fabs(denom) < TPO_KTINY) return 0;
  const double t1 = denom * A[first];
  const double t2 = denom * (-A[second]);
  if (first_type == CT_UPPER) {
    if (second_type == CT_UPPER) return t1 <= 0 && t2 <= 0;
    return t1 >= 0 && t2 <= 0;
  }
  if (second_type == CT_UPPER) return t1 <= 0 && t2 >= 0;
  return t1 >= 0 && t2 >= 0;
}

typedef struct { int index, type; double slope; } active_t;

/* remove the first (index,type) match, keeping order (.cc:1250-1255, :1349-1354) */
static void set_erase(int *set_idx, int *set_type, int *n, int index, int type) {
  for (int k = 0; k < *n; k++) {
    if (set_idx[k] == index && set_type[k] == type) {
      for (int m = k; m + 1 < *n; m++) { set_idx[m] = set_idx[m + 1]; set_type[m] = set_type[m + 1]; }
      (*n)--;
      return;
    }
  }
}

/* .cc:1149-1363 */
void tpo_find_max_sd2_simplex(const double *A, const double *B, const double *lo,
                              const double *hi, int C, double *sd2max,
                              double *sddmax, double *sd2zero) {
  *sd2max = 0; *sddmax = 0;
  int *set_idx = (int *)malloc(sizeof(int) * 4 * (C > 0 ? C : 1));
  int *set_type = set_idx + 2 * (C > 0 ? C : 1);
  active_t *act = (active_t *)malloc(sizeof(active_t) * (2 * (C > 0 ? C : 1) + 1));
  int nset = 2 * C, nact = 0;
  for (int c = 0; c < C; c++) {
    set_idx[2 * c] = c; set_type[2 * c] = CT_UPPER;
    set_idx[2 * c + 1] = c; set_type[2 * c + 1] = CT_LOWER;
  }
  /* step 2: walk along sdd = 0 */
  double sd2 = DBL_MAX, sdd = 0.0;
  for (int idx = 0; idx < C; idx++) {
    if (fabs(B[idx]) < TPO_KTINY) continue;
    if (B[idx] > TPO_KTINY) {
      const double invB = 1.0 / B[idx];
      const double tmp = hi[idx] * invB;
      if (tmp < (sd2 + TPO_KTINY) && tmp > 0) {
        if (tmp < sd2 - TPO_KTINY) nact = 0;
        act[nact].index = idx; act[nact].type = CT_UPPER;
        act[nact].slope = fabs(A[idx] * invB); nact++;
        sd2 = tmp;
      }
    } else if (B[idx] < -TPO_KTINY) {
      const double invB = 1.0 / B[idx];
      const double tmp = lo[idx] * invB;
      if (tmp < (sd2 + TPO_KTINY) && tmp > 0) {
        if (tmp < sd2 - TPO_KTINY) nact = 0;
        act[nact].index = idx; act[nact].type = CT_LOWER;
        act[nact].slope = fabs(A[idx] * invB); nact++;
        sd2 = tmp;
      }
    }
  }
  if (sd2 > TPO_KMAXSD2 || nact == 0) {
    *sd2zero = TPO_KMAXSD2; *sd2max = TPO_KMAXSD2; *sddmax = 0.0;
    goto done;
  }
  *sd2zero = sd2;
  {
    active_t search = act[0];
    if (nact >= 2) {
      for (int first = 0; first < nact; first++) {
        if (act[first].slope < search.slope) search = act[first];
        for (int second = first + 1; second < nact; second++) {
          if (pair_is_optimal(A, B, act[first].index, act[second].index,
                              act[first].type, act[second].type)) {
            *sd2max = sd2; *sddmax = 0.0;
            goto done;
          }
        }
      }
    }
    for (int k = 0; k < nact; k++) set_erase(set_idx, set_type, &nset, act[k].index, act[k].type);

    for (int loop = 0; loop < C; loop++) {
      if (fabs(A[search.index]) < TPO_KTINY) {
        *sd2max = sd2; *sddmax = sdd;
        goto done;
      }
      /* search line sdd = a + b*sd2 */
      const double invA = 1.0 / A[search.index];
      const double b = -B[search.index] * invA;
      const double a = (search.type == CT_UPPER) ? hi[search.index] * invA
                                                 : lo[search.index] * invA;
      nact = 0;
      double next_sd2 = DBL_MAX, next_sdd = 0.0;
      for (int k = 0; k < nset; k++) {
        const int c = set_idx[k];
        const double Bc = A[c] * b + B[c];
        if (fabs(Bc) < TPO_KTINY) continue;
        const double invB = 1.0 / Bc;
        const double lim = (set_type[k] == CT_UPPER) ? hi[c] : lo[c];
        const double tmp = (lim - A[c] * a) * invB;
        if (tmp < (next_sd2 + TPO_KTINY) && tmp > sd2) {
          if (tmp < next_sd2 - TPO_KTINY) nact = 0;
          act[nact].index = c; act[nact].type = set_type[k];
          act[nact].slope = fabs(A[c] * invB); nact++;
          next_sd2 = tmp;
          next_sdd = a + b * next_sd2;
        }
      }
      if (nact == 0) {
        *sd2max = *sd2zero; *sddmax = 0.0;
        goto done;
      }
      act[nact++] = search;
      search.index = -1; search.type = CT_NOTSET; search.slope = DBL_MAX;
      for (int first = 0; first < nact; first++) {
        if (first != nact - 1 && act[first].slope < search.slope) search = act[first];
        for (int second = first + 1; second < nact; second++) {
          if (pair_is_optimal(A, B, act[first].index, act[second].index,
                              act[first].type, act[second].type)) {
            *sd2max = next_sd2; *sddmax = next_sdd;
            if (next_sd2 > TPO_KMAXSD2) { *sd2max = TPO_KMAXSD2; *sddmax = 0.0; }
            goto done;
          }
        }
      }
      nact--; /* drop the previous search direction */
      for (int k = 0; k < nact; k++) set_erase(set_idx, set_type, &nset, act[k].index, act[k].type);
      sd2 = next_sd2; sdd = next_sdd;
    }
    *sd2max = *sd2zero; *sddmax = 0.0;
  }
done:
  free(set_idx);
  free(act);
}

/* Debug counters (single-threaded use only): [0] forward steps, [1] backward steps,
 * [2] boundary-exceed validity checks, [3] intersection fix-ups, [4] critical-point
 * searches, [5] samples scanned by those searches. */
static long long g_counters[8];
void tpo_debug_counters(long long *out, int reset) {
  for (int i = 0; i < 8; i++) { out[i] = g_counters[i]; if (reset) g_counters[i] = 0; }
}

/* .cc:753-767 */
static void forward_step(const tpo_profile *p, int index, double sd2, double *sdd, double *sd2next) {
  g_counters[0]++;
  *sdd = sdd_max_at(p, index, sd2);
  *sd2next = sd2 + 2.0 * p->ds * (*sdd);
}
static void backward_step(const tpo_profile *p, int index, double sd2, double *sdd, double *sd2prev) {
  g_counters[1]++;
  *sdd = sdd_min_at(p, index, sd2);
  *sd2prev = sd2 - 2.0 * p->ds * (*sdd);
}

/* .cc:1365-1487 */
int tpo_profile_calculate_boundary(tpo_profile *p) {
  const int N = p->N;
  for (int i = 0; i < N; i++) {
    double sd2max = 0, sddmax = 0, sd2zero = 0;
    tpo_find_max_sd2_simplex(ROWS(p, i), &sd2max, &sddmax, &sd2zero);
    p->sd2_max[i] = sd2max;
    p->sd2_zero[i] = sd2zero;
    p->sdd_max[i] = sdd_max_at(p, i, sd2max);
    p->sdd_min[i] = sdd_min_at(p, i, sd2max);
    p->at_sdd0[i] = fabs(p->sd2_max[i] - p->sd2_zero[i]) < TPO_KTINY;
  }
  p->type[0] = TPO_BND_NONE;
  p->type[N - 1] = TPO_BND_NONE;
  p->nfix = 0;
  for (int i = 1; i < N - 1; i++) {
    if (!p->at_sdd0[i - 1] && p->at_sdd0[i] && !p->at_sdd0[i + 1]) {
      p->sd2_max[i - 1] = p->sd2_zero[i - 1];
      p->sdd_max[i - 1] = sdd_max_at(p, i - 1, p->sd2_max[i - 1]);
      p->sdd_min[i - 1] = sdd_min_at(p, i - 1, p->sd2_max[i - 1]);
      p->sd2_max[i + 1] = p->sd2_zero[i + 1];
      p->sdd_max[i + 1] = sdd_max_at(p, i + 1, p->sd2_max[i + 1]);
      p->sdd_min[i + 1] = sdd_max_at(p, i + 1, p->sd2_max[i + 1]); /* sic: .cc:1394-1395 */
    }
    const double sd2p = (p->sd2_max[i + 1] - p->sd2_max[i]) / (p->ds);
    const double sd2p_min = 2 * p->sdd_min[i];
    const double sd2p_max = 2 * p->sdd_max[i];
    const int sink_or_source = (sd2p < sd2p_min) || (sd2p > sd2p_max);
    const int skipped_sdd = (p->sdd_max[i] > 0) && (p->sdd_min[i + 1] < 0);
    const int skipped_sd2 = (p->sd2_max[i] > p->sd2_max[i - 1] - TPO_KTINY) &&
                            (p->sd2_max[i] > p->sd2_max[i + 1] - TPO_KTINY);
    if ((skipped_sd2 || skipped_sdd) && sink_or_source) {
      double tmp, fw, bw;
      forward_step(p, i - 1, p->sd2_max[i - 1], &tmp, &fw);
      backward_step(p, i + 1, p->sd2_max[i + 1], &tmp, &bw);
      double m = p->sd2_zero[i];
      if (fw < m) m = fw;   /* std::min({a,b,c}) keeps the first of equal values */
      if (bw < m) m = bw;
      const double v = (0.0 < m) ? m : 0.0; /* std::max(Scalar{0}, m) */
      p->fix_idx[p->nfix] = i; p->fix_val[p->nfix] = v; p->nfix++;
    }
  }
  for (int k = 0; k < p->nfix; k++) {
    const int index = p->fix_idx[k];
    const double value = p->fix_val[k];
    p->sd2_max[index] = value;
    p->sdd_max[index] = sdd_max_at(p, index, value);
    p->sdd_min[index] = sdd_min_at(p, index, value);
    if (index > 0) {
      p->sd2_max[index - 1] = p->sd2_zero[index - 1];
      p->sdd_max[index - 1] = sdd_max_at(p, index - 1, p->sd2_max[index - 1]);
      p->sdd_min[index - 1] = sdd_min_at(p, index - 1, p->sd2_max[index - 1]);
    }
    if (index < N - 1) {
      p->sd2_max[index + 1] = p->sd2_zero[index + 1];
      p->sdd_max[index + 1] = sdd_max_at(p, index + 1, p->sd2_max[index + 1]);
      p->sdd_min[index + 1] = sdd_min_at(p, index + 1, p->sd2_max[index + 1]);
    }
  }
  for (int i = 1; i < N - 1; i++) {
    const double sd2p = (p->sd2_max[i + 1] - p->sd2_max[i]) / (p->ds);
    const double sd2p_min = 2 * p->sdd_min[i];
    const double sd2p_max = 2 * p->sdd_max[i];
    p->type[i] = TPO_BND_NONE;
    if (sd2p < sd2p_min) p->type[i] = TPO_BND_SINK;
    else if (sd2p > sd2p_max) p->type[i] = TPO_BND_SOURCE;
    if ((sd2p <= sd2p_max) && (sd2p >= sd2p_min)) p->type[i] = TPO_BND_TRAJECTORY;
  }
  return 1;
}

/* .cc:722-751 */
static void sdd_at_intersection(tpo_profile *p, int index) {
  const int N = p->N;
  g_counters[3]++;
  double cand[3]; int n = 0;
  if (index > 0 && index < N - 1) cand[n++] = 0.25 / p->ds * (p->sd2[index + 1] - p->sd2[index - 1]);
  if (index < N - 1) cand[n++] = 0.5 / p->ds * (p->sd2[index + 1] - p->sd2[index]);
  if (index > 0) cand[n++] = 0.5 / p->ds * (p->sd2[index] - p->sd2[index - 1]);
  p->sdd[index] = 0.0;
  for (int k = 0; k < n; k++) {
    if (derivs_valid(p, index, cand[k], p->sd2[index])) { p->sdd[index] = cand[k]; return; }
  }
}

/* .cc:769-857 */
static int add_forward_extremal(tpo_profile *p, int idx_lo) {
  const int N = p->N;
  int idx = idx_lo;
  double sd2tmp = NAN, sddtmp = NAN;
  for (; idx < N - 2; idx++) {
    const int on_boundary = tiny(p->sd2[idx] - p->sd2_max[idx]);
    if (on_boundary && ((p->type[idx] & TPO_BND_TRAJECTORY) && (idx < N - 1) &&
                        (p->type[idx + 1] & TPO_BND_TRAJECTORY))) {
      sd2tmp = p->sd2_max[idx + 1];
      sddtmp = 0.5 * (sd2tmp - p->sd2[idx]) / p->ds;
    } else {
      forward_step(p, idx, p->sd2[idx], &sddtmp, &sd2tmp);
    }
    if (!isnan(p->sd2[idx + 1]) && (p->sd2[idx + 1] < sd2tmp)) {
      sdd_at_intersection(p, idx);
      return N - 1;
    }
    if (sd2tmp > p->sd2_max[idx + 1]) {
      const double sdd_bound = 0.5 * (p->sd2_max[idx + 1] - p->sd2[idx]) / p->ds;
      const int deriv_invalid = !derivs_valid(p, idx, sdd_bound, p->sd2_max[idx]);
      g_counters[2]++;
      const int type_invalid = p->type[idx + 1] & TPO_BND_SINK;
      if (type_invalid || deriv_invalid) return idx;
      sd2tmp = p->sd2_max[idx + 1];
      sddtmp = sdd_bound;
    }
    if (sd2tmp < 0) {
      sd2tmp = 0.0;
      if (idx <= 1) sddtmp = 0.0; else sddtmp = -p->sd2[idx - 1] / p->ds;
    }
    p->sd2[idx + 1] = sd2tmp;
    p->sdd[idx] = sddtmp;
  }
  return N - 1;
}

/* .cc:859-952 */
static int add_backward_extremal(tpo_profile *p, int idx_hi) {
  const int N = p->N;
  int idx = idx_hi;
  double sd2tmp = NAN, sddtmp = NAN;
  for (; idx > 1; idx--) {
    const int on_boundary = tiny(p->sd2[idx] - p->sd2_max[idx]);
    if (on_boundary && ((p->type[idx] & TPO_BND_TRAJECTORY) && (idx > 0) &&
                        (p->type[idx - 1] & TPO_BND_TRAJECTORY))) {
      sd2tmp = p->sd2_max[idx - 1];
      sddtmp = 0.5 * (p->sd2[idx] - sd2tmp) / p->ds;
    } else {
      backward_step(p, idx, p->sd2[idx], &sddtmp, &sd2tmp);
    }
    if (!isnan(p->sd2[idx - 1]) && (p->sd2[idx - 1] < sd2tmp)) {
      sdd_at_intersection(p, idx);
      return 0;
    }
    if (sd2tmp > p->sd2_max[idx - 1]) {
      const double sdd_bound = 0.5 * (p->sd2[idx] - p->sd2_max[idx - 1]) / p->ds;
      const int deriv_invalid = !derivs_valid(p, idx, sdd_bound, p->sd2[idx]);
      g_counters[2]++;
      const int type_invalid = p->type[idx - 1] & TPO_BND_SOURCE;
      const int is_connecting = (idx_hi != (N - 1));
      if ((type_invalid || deriv_invalid) && !is_connecting) return idx;
      sd2tmp = p->sd2_max[idx - 1];
      sddtmp = sdd_bound;
    }
    if (sd2tmp < 0) {
      sd2tmp = 0.0;
      if (idx < N - 1) sddtmp = p->sd2[idx + 1] / p->ds; else sddtmp = 0.0;
    }
    p->sd2[idx - 1] = sd2tmp;
    p->sdd[idx] = sddtmp;
  }
  return 0;
}

/* .cc:697-720 */
static int next_critical_point(const tpo_profile *p, int idx_lo, int idx_hi) {
  int crit = -1;
  g_counters[4]++;
  for (int idx = idx_lo + 1; idx <= idx_hi; idx++) {
    g_counters[5]++;
    if (crit < 0) {
      if ((p->type[idx] & TPO_BND_SOURCE) || (p->type[idx] & TPO_BND_TRAJECTORY)) crit = idx;
    } else {
      if (p->sd2_max[idx] == p->sd2_zero[0]) crit = idx; /* sic: index 0, .cc:710 */
    }
    if ((crit > 0) && (!isnan(p->sd2[idx]))) return crit;
  }
  return -1;
}

/* .cc:287-490 */
int tpo_profile_optimize(tpo_profile *p) {
  if (p->state != ST_DEFINED) return TPO_ERR_NOT_SOLVED;
  const int N = p->N;
  for (int i = 0; i < N; i++) { p->sd2[i] = NAN; p->sdd[i] = NAN; }
  p->sd2[0] = p->sd_start * p->sd_start;
  p->sd2[N - 1] = 0;
  p->dt_max = 0.0;
  p->loops_used = 0;
  tpo_profile_calculate_boundary(p);

  int iforw_lo = 0, iback_hi = N - 1, iback_lo, iforw_hi, icrit, icrit_lo, icrit_hi;
  iback_lo = add_backward_extremal(p, iback_hi);
  iforw_hi = add_forward_extremal(p, iforw_lo);
  icrit_hi = iback_lo;
  if ((iforw_hi < icrit_hi) && ((icrit_hi < N - 2) && (icrit_hi >= 2))) {
    p->sd2[icrit_hi] = NAN;
    icrit_hi++;
    iback_lo++;
  }
  icrit_lo = iforw_hi;
  for (int loop = 0; loop < p->max_loops; loop++) {
    if (iforw_hi >= icrit_hi) break;
    p->loops_used = loop + 1;
    icrit = next_critical_point(p, icrit_lo, icrit_hi);
    if (icrit < 0 || icrit >= N) icrit = (int)(0.5 * (icrit_lo + icrit_hi));
    if (icrit > 0 && icrit < N - 1) p->sd2[icrit] = p->sd2_max[icrit];
    if (icrit < 1) return TPO_ERR_CRIT_INDEX_ZERO; /* reference reads sd2_max[-1] here */
    if (p->sd2_max[icrit - 1] <= p->sd2_max[icrit]) {
      iback_hi = icrit - 1;
      p->sd2[icrit - 1] = p->sd2_max[icrit - 1];
    } else {
      iback_hi = icrit;
    }
    {
      /* debug only: what running the two extremals of a loop concurrently would save */
      const long long b0 = g_counters[1], f0 = g_counters[0];
      iback_lo = add_backward_extremal(p, iback_hi);
      iforw_lo = icrit;
      iforw_hi = add_forward_extremal(p, iforw_lo);
      const long long cb = g_counters[1] - b0, cf = g_counters[0] - f0;
      g_counters[6] += (cb > cf) ? cb : cf;
      g_counters[7] += (iback_hi == icrit);
    }
    if (iback_lo > icrit_lo) return TPO_ERR_NO_CONNECTION;
    icrit_lo = iforw_hi;
  }
  for (int idx = 0; idx < N; idx++) {
    if (isnan(p->sd2[idx])) return TPO_ERR_NAN_SD2;
    if (isnan(p->sdd[idx])) sdd_at_intersection(p, idx);
  }
  if (derivs_valid(p, 0, p->sdd_start, p->sd2[0])) p->sdd[0] = p->sdd_start;
  for (int i = 0; i < N; i++) p->sd[i] = sqrt(p->sd2[i]);
  if (p->sd2[N - 1] != 0) return TPO_ERR_NONZERO_END;

  p->last_extremal_index = (1 > N - 2) ? 1 : N - 2;
  while (p->last_extremal_index >= 1) {
    if (p->sdd[p->last_extremal_index] > 0.0 ||
        fabs(p->sd2[p->last_extremal_index] - p->sd2_max[p->last_extremal_index]) < TPO_KTINY)
      break;
    p->last_extremal_index--;
  }
  p->time[0] = p->time_start;
  for (int idx = 1; idx < N; idx++) {
    if ((p->sd2[idx - 1] > 0) || (p->sd2[idx] > 0)) {
      const double dt = 2.0 * p->ds / (p->sd[idx - 1] + p->sd[idx]);
      p->time[idx] = p->time[idx - 1] + dt;
      p->dt_max = (dt > p->dt_max) ? dt : p->dt_max; /* std::max(dt, dt_max_) */
    } else {
      p->time[idx] = p->time[idx - 1];
      p->sdd[idx - 1] = 0;
      p->sdd[idx] = 0;
    }
  }
  p->low_idx = 0;
  while ((p->time[p->low_idx] == p->time[p->low_idx + 1]) && (p->low_idx < N - 2)) p->low_idx++;
  p->high_idx = N - 1;
  while ((p->high_idx >= 1) && (p->time[p->high_idx] == p->time[p->high_idx - 1]) &&
         (p->high_idx >= p->low_idx))
    p->high_idx--;
  p->state = ST_SOLVED;
  return TPO_OK;
}

const double *tpo_profile_time(const tpo_profile *p) { return p->time; }
const double *tpo_profile_s(const tpo_profile *p) { return p->s; }
const double *tpo_profile_sd(const tpo_profile *p) { return p->sd; }
const double *tpo_profile_sdd(const tpo_profile *p) { return p->sdd; }
const double *tpo_profile_sd2(const tpo_profile *p) { return p->sd2; }
const double *tpo_profile_sd2_max(const tpo_profile *p) { return p->sd2_max; }
const double *tpo_profile_sdd_max_for_sd2_max(const tpo_profile *p) { return p->sdd_max; }
const double *tpo_profile_sdd_min_for_sd2_max(const tpo_profile *p) { return p->sdd_min; }
const double *tpo_profile_sd2_max_for_sdd0(const tpo_profile *p) { return p->sd2_zero; }
const uint8_t *tpo_profile_boundary_type(const tpo_profile *p) { return p->type; }
int tpo_profile_last_extremal_index(const tpo_profile *p) { return p->last_extremal_index; }
double tpo_profile_max_time_increment(const tpo_profile *p) {
  return p->state == ST_SOLVED ? p->dt_max : -1.0;
}
int tpo_profile_num_loops_used(const tpo_profile *p) { return p->loops_used; }

/* .cc:492-518 */
int tpo_profile_constraint_violations(const tpo_profile *p) {
  if (p->state != ST_SOLVED) return -1;
  int count = 0;
  for (int i = 0; i < p->N; i++) {
    const double *A = p->A + (size_t)i * p->C, *B = p->B + (size_t)i * p->C;
    const double *lo = p->lo + (size_t)i * p->C, *hi = p->hi + (size_t)i * p->C;
    for (int c = 0; c < p->C; c++) {
      const double v = A[c] * p->sdd[i] + B[c] * p->sd2[i];
      if ((v + TPO_KTINY < lo[c]) || (v - TPO_KTINY > hi[c])) count++;
    }
  }
  return count;
}

/* .cc:1497-1524 */
static int sample_index_from_time(const tpo_profile *p, double t) {
  const int N = p->N;
  if (t <= p->time[0]) return 0;
  if (t >= p->time[N - 1]) return N - 2;
  int low = p->low_idx, high = p->high_idx, mid = (low + high) / 2;
  while (t < p->time[mid] || t >= p->time[mid + 1]) {
    if (t < p->time[mid]) high = mid; else low = mid;
    mid = (low + high) / 2;
  }
  while ((mid < N - 1) && (p->time[mid] == p->time[mid + 1])) mid++;
  return mid;
}

/* .cc:1549-1627 */
int tpo_profile_query(const tpo_profile *p, double t, double *s, double *sd, double *sdd) {
  if (p->state != ST_SOLVED) return 0;
  const int N = p->N;
  if (t <= p->time[0]) {
    *s = p->s_start; *sd = p->sd[0];
    *sdd = 0.5 * p->inv_ds * (p->sd2[1] - p->sd2[0]);
    return 1;
  }
  if (t >= p->time[N - 1]) {
    *s = p->s_end; *sd = p->sd[N - 1]; *sdd = 0.0;
    return 1;
  }
  const int k = sample_index_from_time(p, t);
  if (p->time[k] == p->time[k + 1]) {
    *s = p->s[k + 1]; *sd = p->sd[k + 1];
    *sdd = 0.5 * p->inv_ds * (p->sd2[k + 1] - p->sd2[k]);
    return 1;
  }
  const double dt = t - p->time[k];
  double ds = 0;
  if (k > (N - 2)) return 0;
  const double sda = p->sd[k], sdb = p->sd[k + 1];
  const double sd2a = p->sd2[k], sd2b = p->sd2[k + 1];
  if (sda > 0 || sdb > 0) {
    ds = sda * dt + dt * dt * 0.25 * p->inv_ds * (sd2b - sd2a);
    if (ds > p->ds) ds = p->ds;
    if (dt < 0 || ds < 0) return 0;
    const double cand = p->s[k] + ds;
    *s = (p->s[k + 1] < cand) ? p->s[k + 1] : cand; /* std::min(a, b) */
    const double sqrt_arg = sd2a + ds * p->inv_ds * (sd2b - sd2a);
    *sd = sqrt(sqrt_arg);
    *sdd = 0.5 * p->inv_ds * (sd2b - sd2a);
  } else {
    *s = p->s[k] + (p->s[k + 1] - p->s[k]) * dt / (p->time[k + 1] - p->time[k]);
    *sd = 0.0; *sdd = 0.0;
  }
  return 1;
}

/* .cc:1629-1645 */
int tpo_profile_previous_index(const tpo_profile *p, double t) {
  if (p->state != ST_SOLVED) return -1;
  if (t < p->time[0]) return -1;
  if (t > p->time[p->N - 1]) return p->N - 1;
  return sample_index_from_time(p, t);
}

/* ======================================================================= */
/*                       planner epilogue and resample                      */
/* ======================================================================= */

/* path_timing_trajectory.cc:458-472 */
void tpo_epilogue(const double *q1, const double *q2, int N, int D,
                  const double *sd, const double *sdd, const double *amax,
                  double *qd, double *qdd) {
  for (int i = 0; i < N; i++) {
    const double v = sd[i], a = sdd[i];
    const double v2 = v * v; /* eigenmath::Square */
    for (int d = 0; d < D; d++) {
      const size_t k = (size_t)i * D + d;
      qd[k] = q1[k] * v;
      double acc = q1[k] * a + q2[k] * v2;
      if (acc < -amax[d]) acc = -amax[d]; /* cwiseMax(-amax) */
      if (acc > amax[d]) acc = amax[d];   /* cwiseMin(amax) */
      qdd[k] = acc;
    }
  }
}

int tpo_resample_uniform_count(double end_time, double start_sec, double time_step) {
  const double duration = end_time - start_sec;
  return (int)(ceil(duration / time_step) + 1);
}

/* eigenmath::InterpolateLinear is not in the reference tree (external
 * x_edr_eigenmath 1.0.0): restated as a + t*(b-a); PARITY UNPINNED at the ulp
 * level for the resampled outputs. */
static double lerp(double t, double a, double b) { return a + t * (b - a); }

/* path_timing_trajectory.cc:686-695 */
static int lower_index_from(const double *time, int N, int start, double t) {
  for (int index = start; index < N - 1; ++index)
    if (time[index + 1] > t) return index;
  return N - 1;
}

/* path_timing_trajectory.cc:755-783 with :709-753 */
int tpo_resample_uniform(const double *time, const double *s, const double *sd,
                         const double *sdd, const double *q, const double *qd,
                         const double *qdd, int N, int D, double start_sec,
                         double time_step, const double *amax, int max_out,
                         double *ot, double *os, double *osd, double *osdd,
                         double *oq, double *oqd, double *oqdd) {
  const int M = tpo_resample_uniform_count(time[N - 1], start_sec, time_step);
  int lower = 0;
  const int lim = (M < max_out) ? M : max_out;
  for (int i = 0; i < lim; i++) {
    const double t = start_sec + time_step * i;
    lower = lower_index_from(time, N, lower, t);
    const int upper = (N - 1 < lower + 1) ? N - 1 : lower + 1;
    const double at = (fabs(time[upper] - time[lower]) < DBL_EPSILON)
                          ? 0.5
                          : (t - time[lower]) / (time[upper] - time[lower]);
    ot[i] = t;
    for (int d = 0; d < D; d++) {
      const size_t kl = (size_t)lower * D + d, ku = (size_t)upper * D + d, ko = (size_t)i * D + d;
      oq[ko] = lerp(at, q[kl], q[ku]);
      oqd[ko] = lerp(at, qd[kl], qd[ku]);
      double a = lerp(at, qdd[kl], qdd[ku]);
      if (a < -amax[d]) a = -amax[d];
      if (a > amax[d]) a = amax[d];
      oqdd[ko] = a;
    }
    os[i] = lerp(at, s[lower], s[upper]);
    osd[i] = lerp(at, sd[lower], sd[upper]);
    osdd[i] = lerp(at, sdd[lower], sdd[upper]);
  }
  if (M >= 1 && M <= max_out) {
    for (int d = 0; d < D; d++) {
      oq[(size_t)(M - 1) * D + d] = q[(size_t)(N - 1) * D + d];
      oqd[(size_t)(M - 1) * D + d] = 0.0;
      oqdd[(size_t)(M - 1) * D + d] = 0.0;
    }
  }
  return M;
}

/* ResampleSkippingSamplesCloserThanTimeStep, path_timing_trajectory.cc:785-836: the
 * first output is interpolated at start_sec (InterpolateAtTime :709-753), then every
 * path sample at least min_delta (= 0.95 time_step, :893-900) after the last kept one is
 * taken as it is; the last output gets the end position and zero derivatives. */
int tpo_resample_skip(const double *time, const double *s, const double *sd,
                      const double *sdd, const double *q, const double *qd,
                      const double *qdd, int N, int D, double start_sec,
                      double min_delta, const double *amax, int max_out,
                      double *ot, double *os, double *osd, double *osdd,
                      double *oq, double *oqd, double *oqdd) {
  const int lower = lower_index_from(time, N, 0, start_sec);
  const int upper = (N - 1 < lower + 1) ? N - 1 : lower + 1;
  const double at = (fabs(time[upper] - time[lower]) < DBL_EPSILON)
                        ? 0.5
                        : (start_sec - time[lower]) / (time[upper] - time[lower]);
  int M = 0;
  double last = start_sec;
  if (M < max_out) {
    ot[0] = start_sec;
    os[0] = lerp(at, s[lower], s[upper]);
    osd[0] = lerp(at, sd[lower], sd[upper]);
    osdd[0] = lerp(at, sdd[lower], sdd[upper]);
    for (int d = 0; d < D; d++) {
      const size_t kl = (size_t)lower * D + d, ku = (size_t)upper * D + d;
      oq[d] = lerp(at, q[kl], q[ku]);
      oqd[d] = lerp(at, qd[kl], qd[ku]);
      double a = lerp(at, qdd[kl], qdd[ku]);
      if (a < -amax[d]) a = -amax[d];
      if (a > amax[d]) a = amax[d];
      oqdd[d] = a;
    }
  }
  M = 1;
  for (int i = lower + 1; i < N; i++) {
    if (fabs(time[i] - last) < min_delta) continue;
    last = time[i];
    if (M < max_out) {
      ot[M] = time[i]; os[M] = s[i]; osd[M] = sd[i]; osdd[M] = sdd[i];
      memcpy(oq + (size_t)M * D, q + (size_t)i * D, sizeof(double) * D);
      memcpy(oqd + (size_t)M * D, qd + (size_t)i * D, sizeof(double) * D);
      memcpy(oqdd + (size_t)M * D, qdd + (size_t)i * D, sizeof(double) * D);
    }
    M++;
  }
  if (M <= max_out) {
    for (int d = 0; d < D; d++) {
      oq[(size_t)(M - 1) * D + d] = q[(size_t)(N - 1) * D + d];
      oqd[(size_t)(M - 1) * D + d] = 0.0;
      oqdd[(size_t)(M - 1) * D + d] = 0.0;
    }
  }
  return M;
}

/* ======================================================================= */
/*                     whole hot path for one joint path                    */
/* ======================================================================= */

int tpo_time_joint_path(const double *knots, int num_knots,
                        const double *control_points, int num_points, int D,
                        const double *vmax, const double *amax, double safety,
                        double path_start, double delta, int N, double sd_start,
                        double sdd_start, double time_start, tpo_profile *work,
                        double *t, double *s, double *sd, double *sdd, double *q,
                        double *qd, double *qdd, int *last_extremal_index) {
  const int C = 2 * D;
  tpo_profile *p = work ? work : tpo_profile_create(N, C);
  double *q1 = (double *)malloc(sizeof(double) * 2 * (size_t)N * D);
  double *q2 = q1 + (size_t)N * D;
  double *rows = (double *)malloc(sizeof(double) * 4 * (size_t)N * C);
  double *A = rows, *B = rows + (size_t)N * C, *lo = B + (size_t)N * C, *hi = lo + (size_t)N * C;
  int rc = tpo_joint_sample_path(knots, num_knots, control_points, num_points, D,
                                 path_start, delta, N, q, q1, q2);
  if (rc == 0) {
    tpo_joint_constraint_setup(q1, q2, N, D, vmax, amax, safety, A, B, lo, hi);
    const int loops = (100 > 10 * N) ? 100 : 10 * N; /* path_timing_trajectory.cc:398-400 */
    tpo_profile_set_max_loops(p, loops);
    const double horizon = path_start + delta * (N - 1); /* :340-341 */
    rc = tpo_profile_setup(p, A, B, lo, hi, path_start, horizon, sd_start, sdd_start, time_start);
    if (rc == TPO_OK) rc = tpo_profile_optimize(p);
    if (rc == TPO_OK) {
      memcpy(t, p->time, sizeof(double) * N);
      memcpy(s, p->s, sizeof(double) * N);
      memcpy(sd, p->sd, sizeof(double) * N);
      memcpy(sdd, p->sdd, sizeof(double) * N);
      tpo_epilogue(q1, q2, N, D, p->sd, p->sdd, amax, qd, qdd);
      if (last_extremal_index) *last_extremal_index = p->last_extremal_index;
    }
  } else {
    rc = 100 + rc;
  }
  free(q1); free(rows);
  if (!work) tpo_profile_destroy(p);
  return rc;
}

int tpo_time_joint_batch(int B, const double *knots, int num_knots,
                         const double *control_points, int num_points, int D,
                         const double *vmax, const double *amax, double safety,
                         const double *path_start, const double *delta, int N,
                         const double *sd_start, const double *time_start,
                         int nthreads, double *t, double *s, double *sd,
                         double *sdd, double *q, double *qd, double *qdd,
                         int *last_extremal_index, int *status) {
  int failed = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : failed)
  {
    tpo_profile *work = tpo_profile_create(N, 2 * D);
#pragma omp for schedule(dynamic, 4)
    for (int b = 0; b < B; b++) {
      int lei = 0;
      const int rc = tpo_time_joint_path(
          knots + (size_t)b * num_knots, num_knots,
          control_points + (size_t)b * num_points * D, num_points, D,
          vmax + (size_t)b * D, amax + (size_t)b * D, safety, path_start[b],
          delta[b], N, sd_start[b], 0.0, time_start[b], work,
          t + (size_t)b * N, s + (size_t)b * N, sd + (size_t)b * N,
          sdd + (size_t)b * N, q + (size_t)b * N * D, qd + (size_t)b * N * D,
          qdd + (size_t)b * N * D, &lei);
      work->state = ST_ALLOCATED;
      status[b] = rc;
      if (last_extremal_index) last_extremal_index[b] = lei;
      if (rc != TPO_OK) failed++;
    }
    tpo_profile_destroy(work);
  }
  return failed;
}

/* jacobian * first_path_derivative (timeable_path_cartesian_spline.cc:577): J is
 * [N][6][D] row-major, the product is accumulated over the dofs in index order.
 * Eigen's own summation order for this 6xD product is an implementation detail
 * (it may differ in the last ulp); parity on these two rows is 1e-6 relative. */
void tpo_cartesian_jacobian_times_q1(const double *J, const double *q1, int N, int D,
                                     double *jq1) {
  for (int i = 0; i < N; i++)
    for (int r = 0; r < 6; r++) {
      double acc = 0.0;
      for (int d = 0; d < D; d++) acc += J[((size_t)i * 6 + r) * D + d] * q1[(size_t)i * D + d];
      jq1[(size_t)i * 6 + r] = acc;
    }
}

/* The Cartesian path's share of ComputeTimingProfile after the IK callback has run:
 * ComputePathDerivatives (:39-68, called from SamplePath :541-542), ConstraintSetup
 * (:551-595), then the solver and the planner epilogue exactly as for a joint path
 * (path_timing_trajectory.cc:340-341, :398-400, :458-472). */
int tpo_time_cartesian_path(const double *q, const double *J, int N, int D,
                            const double *vmax, const double *amax, double max_trans_vel,
                            double max_rot_vel, double safety, double path_start,
                            double delta, double sd_start, double sdd_start,
                            double time_start, double *t, double *s, double *sd,
                            double *sdd, double *qd, double *qdd,
                            int *last_extremal_index) {
  const int C = 2 * D + 2;
  tpo_profile *p = tpo_profile_create(N, C);
  double *q1 = (double *)malloc(sizeof(double) * (2 * (size_t)N * D + 6 * (size_t)N));
  double *q2 = q1 + (size_t)N * D, *jq1 = q2 + (size_t)N * D;
  double *rows = (double *)malloc(sizeof(double) * 4 * (size_t)N * C);
  double *A = rows, *B = rows + (size_t)N * C, *lo = B + (size_t)N * C, *hi = lo + (size_t)N * C;
  tpo_cartesian_path_derivatives(q, N, D, delta, q1, q2);
  tpo_cartesian_jacobian_times_q1(J, q1, N, D, jq1);
  tpo_cartesian_constraint_setup(q1, q2, jq1, N, D, vmax, amax, max_trans_vel, max_rot_vel,
                                 safety, A, B, lo, hi);
  tpo_profile_set_max_loops(p, (100 > 10 * N) ? 100 : 10 * N);
  int rc = tpo_profile_setup(p, A, B, lo, hi, path_start, path_start + delta * (N - 1),
                             sd_start, sdd_start, time_start);
  if (rc == TPO_OK) rc = tpo_profile_optimize(p);
  if (rc == TPO_OK) {
    memcpy(t, p->time, sizeof(double) * N);
    memcpy(s, p->s, sizeof(double) * N);
    memcpy(sd, p->sd, sizeof(double) * N);
    memcpy(sdd, p->sdd, sizeof(double) * N);
    tpo_epilogue(q1, q2, N, D, p->sd, p->sdd, amax, qd, qdd);
    if (last_extremal_index) *last_extremal_index = p->last_extremal_index;
  }
  free(q1); free(rows);
  tpo_profile_destroy(p);
  return rc;
}

int tpo_time_cartesian_batch(int B, const double *q, const double *J, int N, int D,
                             const double *vmax, const double *amax,
                             const double *max_trans_vel, const double *max_rot_vel,
                             double safety, const double *path_start, const double *delta,
                             const double *sd_start, const double *time_start, int nthreads,
                             double *t, double *s, double *sd, double *sdd, double *qd,
                             double *qdd, int *last_extremal_index, int *status) {
  int failed = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4) reduction(+ : failed)
  for (int b = 0; b < B; b++) {
    int lei = 0;
    const int rc = tpo_time_cartesian_path(
        q + (size_t)b * N * D, J + (size_t)b * N * 6 * D, N, D, vmax + (size_t)b * D,
        amax + (size_t)b * D, max_trans_vel[b], max_rot_vel[b], safety, path_start[b], delta[b],
        sd_start[b], 0.0, time_start[b], t + (size_t)b * N, s + (size_t)b * N, sd + (size_t)b * N,
        sdd + (size_t)b * N, qd + (size_t)b * N * D, qdd + (size_t)b * N * D, &lei);
    status[b] = rc;
    if (last_extremal_index) last_extremal_index[b] = lei;
    if (rc != TPO_OK) failed++;
  }
  return failed;
}
