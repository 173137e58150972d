/*
 * tp_oracle_plan.c -- CPU restatement of PathTimingTrajectory::Plan for a joint-space spline
 * path (test infrastructure, like the rest of oracle/; the product never links it).
 *
 * Follows, statement by statement, trajectory_planning/ of the reference:
 *   Plan                      path_timing_trajectory.cc:579-684
 *   HandleTimeArguments       :502-538        UpdatePathTrackingStatus   :477-500
 *   EraseTrajectoryBefore     :540-575 (both sampling methods; GetSampleCountUntil :41-44)
 *   GetTimeOffsetAfter        :289-305        ComputeTimingProfile       :307-475
 *   InterpolateAtTime         :709-753        TimeAtPathSamplesLowerIndex :686-695
 *   ResampleEquidistantlyInTime :755-783      ResampleSkipping...        :785-836
 *   ClampToTimeStepMultiple   :229-233        GetMinTimeDeltaToKeep      :893-900
 *   time.h:22-29  TimeFromSec truncates seconds*1e9 to int64 nanoseconds, TimeToSec divides.
 *   TimeableJointSplinePath   timeable_path_joint_spline.cc: CloseToEnd :142-144 (kSmall 1e-4),
 *                             SamplePath :294-318 (state -> kPathWasSampled), state enum
 *                             timeable_path.h:94-103.
 * The single-window pieces (sampling, rows, solver, epilogue) are the functions of tp_oracle.c.
 *
 * absl::Time / absl::Duration are held as int64 nanoseconds; absl::Seconds(double) is rounded
 * to the nearest nanosecond (abseil keeps quarter nanoseconds: differences below 1 ns are
 * outside what this restatement pins). eigenmath::InterpolateLinear is restated as
 * a + t (b - a), as in tpo_resample_uniform (ulp-level parity unpinned).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tp_oracle.h"

#define NSECS_PER_SEC 1000000000LL

typedef struct {
  double *v;
  size_t n, cap;
} dvec;

static void dv_reserve(dvec *a, size_t cap) {
  if (cap > a->cap) {
    size_t c = a->cap ? a->cap : 64;
    while (c < cap) c *= 2;
    a->v = (double *)realloc(a->v, c * sizeof(double));
    a->cap = c;
  }
}
static void dv_resize(dvec *a, size_t n) { dv_reserve(a, n); a->n = n; }
static void dv_append(dvec *a, const double *p, size_t n) {
  dv_reserve(a, a->n + n);
  memcpy(a->v + a->n, p, n * sizeof(double));
  a->n += n;
}
static void dv_erase_front(dvec *a, size_t n) {   /* erase [0, n) */
  if (n > a->n) n = a->n;
  memmove(a->v, a->v + n, (a->n - n) * sizeof(double));
  a->n -= n;
}
static void dv_free(dvec *a) { free(a->v); a->v = NULL; a->n = a->cap = 0; }

struct tpo_planner {
  /* options (PathTimingTrajectoryOptions, JointPathOptions) */
  int D, N;
  double delta, safety;
  int64_t time_step_ns;
  int sampling_method; /* 0 kUniformlyInTime, 1 kSkipSamplesCloserThanTimeStep */
  int max_planning_iterations;
  double max_initial_velocity_error;
  double time_step_sec;
  /* path (TimeableJointSplinePath) */
  int path_state; /* TPO_PATH_* */
  int num_points, num_knots;
  double *knots, *cps, *vmax, *amax, *initial_velocity;
  double *q, *q1, *q2; /* last SamplePath */
  /* path (TimeableCartesianSplinePath after its IK callback): the IK solution at every multiple
   * of delta along the whole path and the Jacobian there; see tpo_planner_set_ik_table */
  int table_len;
  double *table_q, *table_J, table_end;
  double max_trans_vel, max_rot_vel;
  /* planner state (path_timing_trajectory.h members) */
  int initial_plan, planned_to_end, target_reached;
  double path_time_start, path_start, path_start_velocity, path_start_acceleration, path_horizon;
  int64_t start_time, end_time, final_decel_start;
  dvec t_ps, s_ps, sd_ps, sdd_ps, q_ps, qd_ps, qdd_ps;      /* *_at_path_samples_ */
  dvec time, s, sd, sdd, pos, vel, acc;                     /* the resampled trajectory */
  tpo_profile *profile;
  int profile_solved;
  int windows; /* ComputeTimingProfile calls of the last Plan */
};

static int64_t time_from_sec(double seconds) { return (int64_t)(seconds * (double)NSECS_PER_SEC); }
static double time_to_sec(int64_t t) { return (double)t / (double)NSECS_PER_SEC; }
static int64_t seconds_to_duration(double s) { return (int64_t)llround(s * 1e9); }

tpo_planner *tpo_planner_create(int D, int N, double delta, double safety, int64_t time_step_ns,
                                int sampling_method, int max_planning_iterations,
                                double max_initial_velocity_error) {
  tpo_planner *p = (tpo_planner *)calloc(1, sizeof(tpo_planner));
  p->D = D; p->N = N; p->delta = delta; p->safety = safety;
  p->time_step_ns = time_step_ns;
  p->sampling_method = sampling_method;
  p->max_planning_iterations = max_planning_iterations;
  p->max_initial_velocity_error = max_initial_velocity_error;
  /* constructor :206-211: time_step_sec_ = TimeToSec(FromUnixDuration(time step)) */
  p->time_step_sec = time_to_sec(time_step_ns);
  p->vmax = (double *)calloc((size_t)D, sizeof(double));
  p->amax = (double *)calloc((size_t)D, sizeof(double));
  p->initial_velocity = (double *)calloc((size_t)D, sizeof(double));
  p->q = (double *)calloc((size_t)3 * N * D, sizeof(double));
  p->q1 = p->q + (size_t)N * D;
  p->q2 = p->q1 + (size_t)N * D;
  p->profile = tpo_profile_create(N, 2 * D);
  p->path_state = TPO_PATH_NONE;
  /* ResetDerived :213-227 */
  p->planned_to_end = 1;
  p->final_decel_start = time_from_sec(0.0);
  return p;
}

void tpo_planner_destroy(tpo_planner *p) {
  if (!p) return;
  free(p->knots); free(p->cps); free(p->vmax); free(p->amax); free(p->initial_velocity); free(p->q);
  free(p->table_q); free(p->table_J);
  dv_free(&p->t_ps); dv_free(&p->s_ps); dv_free(&p->sd_ps); dv_free(&p->sdd_ps);
  dv_free(&p->q_ps); dv_free(&p->qd_ps); dv_free(&p->qdd_ps);
  dv_free(&p->time); dv_free(&p->s); dv_free(&p->sd); dv_free(&p->sdd);
  dv_free(&p->pos); dv_free(&p->vel); dv_free(&p->acc);
  tpo_profile_destroy(p->profile);
  free(p);
}

void tpo_planner_set_limits(tpo_planner *p, const double *vmax, const double *amax) {
  memcpy(p->vmax, vmax, sizeof(double) * p->D);
  memcpy(p->amax, amax, sizeof(double) * p->D);
}

void tpo_planner_set_initial_velocity(tpo_planner *p, const double *v) {
  memcpy(p->initial_velocity, v, sizeof(double) * p->D);
}

/* The spline a TimeableJointSplinePath holds after SetWaypoints (state kNewPath) or after
 * SwitchToWaypointPath (state kModifiedPath). */
void tpo_planner_set_spline(tpo_planner *p, const double *knots, int num_knots, const double *cps,
                            int num_points, int state) {
  free(p->knots); free(p->cps);
  p->knots = (double *)malloc(sizeof(double) * num_knots);
  p->cps = (double *)malloc(sizeof(double) * (size_t)num_points * p->D);
  memcpy(p->knots, knots, sizeof(double) * num_knots);
  memcpy(p->cps, cps, sizeof(double) * (size_t)num_points * p->D);
  p->num_knots = num_knots; p->num_points = num_points;
  p->path_state = state;
}

/* What a TimeableCartesianSplinePath holds once its path-IK callback has run over the whole
 * path (timeable_path_cartesian_spline.cc:484-538: path_ik_positions_, one sample per multiple of
 * delta_parameter, PathIkIndex = round(parameter / delta) :671-674) plus the Jacobian the
 * Jacobian callback returns at each of them (:576) and the Cartesian velocity limits. The IK and
 * Jacobian callbacks are user code; the planner then does arithmetic only: SamplePath copies a
 * segment of the table and differentiates it (:39-68, :527-542), ConstraintSetup adds the two
 * Cartesian rows (:551-595). path_end = knots.back() (CloseToEnd :411-413). */
void tpo_planner_set_ik_table(tpo_planner *p, const double *ik_positions, const double *jacobians,
                              int num_table_samples, double path_end, double max_trans_vel,
                              double max_rot_vel, int state) {
  const size_t D = (size_t)p->D;
  free(p->table_q); free(p->table_J);
  p->table_q = (double *)malloc(sizeof(double) * (size_t)num_table_samples * D);
  p->table_J = (double *)malloc(sizeof(double) * (size_t)num_table_samples * 6 * D);
  memcpy(p->table_q, ik_positions, sizeof(double) * (size_t)num_table_samples * D);
  memcpy(p->table_J, jacobians, sizeof(double) * (size_t)num_table_samples * 6 * D);
  p->table_len = num_table_samples;
  p->table_end = path_end;
  p->max_trans_vel = max_trans_vel;
  p->max_rot_vel = max_rot_vel;
  p->path_state = state;
  tpo_profile_destroy(p->profile);
  p->profile = tpo_profile_create(p->N, 2 * p->D + 2);
}

/* timeable_path_joint_spline.cc:142-144 / timeable_path_cartesian_spline.cc:411-413 */
static int close_to_end(const tpo_planner *p, double parameter) {
  const double kSmall = 1e-4;
  if (p->table_q) return parameter >= p->table_end - kSmall;
  return p->num_knots == 0 || parameter >= p->knots[p->num_knots - 1] - kSmall;
}

/* :229-233 */
static void clamp_to_time_step_multiple(const tpo_planner *p, int64_t *t) {
  const int64_t loop_multiple = (int64_t)round(time_to_sec(*t) / p->time_step_sec);
  *t = time_from_sec((double)loop_multiple * p->time_step_sec);
}

/* :686-695 */
static int lower_index_at(const tpo_planner *p, int starting_index, double time) {
  const int n = (int)p->t_ps.n;
  for (int index = starting_index; index < n - 1; ++index)
    if (p->t_ps.v[index + 1] > time) return index;
  return n - 1;
}

static double lerp_ref(double t, double a, double b) { return a + t * (b - a); }

/* :709-753. Outputs: position/velocity/acceleration [D] and the three path scalars. */
static int interpolate_at_time(const tpo_planner *p, double time_sec, int lower_index, double *pos,
                               double *vel, double *acc, double *s, double *sd, double *sdd) {
  const int D = p->D;
  const int lower = lower_index_at(p, lower_index, time_sec);
  const int n = (int)p->t_ps.n;
  const int upper = (n - 1 < lower + 1) ? n - 1 : lower + 1;
  const double *tm = p->t_ps.v;
  const double at = (fabs(tm[upper] - tm[lower]) < 2.220446049250313e-16)
                        ? 0.5
                        : (time_sec - tm[lower]) / (tm[upper] - tm[lower]);
  for (int d = 0; d < D; d++) {
    pos[d] = lerp_ref(at, p->q_ps.v[(size_t)lower * D + d], p->q_ps.v[(size_t)upper * D + d]);
    vel[d] = lerp_ref(at, p->qd_ps.v[(size_t)lower * D + d], p->qd_ps.v[(size_t)upper * D + d]);
    double a = lerp_ref(at, p->qdd_ps.v[(size_t)lower * D + d], p->qdd_ps.v[(size_t)upper * D + d]);
    if (a < -p->amax[d]) a = -p->amax[d];   /* cwiseMax(-a_max) then cwiseMin(a_max) */
    if (a > p->amax[d]) a = p->amax[d];
    acc[d] = a;
  }
  *s = lerp_ref(at, p->s_ps.v[lower], p->s_ps.v[upper]);
  *sd = lerp_ref(at, p->sd_ps.v[lower], p->sd_ps.v[upper]);
  *sdd = lerp_ref(at, p->sdd_ps.v[lower], p->sdd_ps.v[upper]);
  return lower;
}

static void erase_samples_until(tpo_planner *p, int offset) { /* :868-880 */
  const size_t D = (size_t)p->D;
  if (offset < 0) offset = 0;
  dv_erase_front(&p->time, (size_t)offset);
  dv_erase_front(&p->s, (size_t)offset);
  dv_erase_front(&p->sd, (size_t)offset);
  dv_erase_front(&p->sdd, (size_t)offset);
  dv_erase_front(&p->pos, (size_t)offset * D);
  dv_erase_front(&p->vel, (size_t)offset * D);
  dv_erase_front(&p->acc, (size_t)offset * D);
}

/* :540-575 */
static void erase_trajectory_before(tpo_planner *p, int64_t time) {
  const double time_sec = time_to_sec(time);
  if (p->time.n == 0 || time_sec < p->time.v[0]) return;
  const int D = p->D;
  if (p->sampling_method == 1) {
    /* GetSampleCountUntil :41-44: lower_bound = number of samples with a time stamp < time_sec */
    int smaller = 0;
    {
      int lo = 0, hi = (int)p->time.n;
      while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (p->time.v[mid] < time_sec) lo = mid + 1; else hi = mid;
      }
      smaller = lo;
      /* the reference reads time_[smaller] below without a guard; start <= the last sample in
       * every call Plan makes */
      if (smaller > (int)p->time.n - 1) smaller = (int)p->time.n - 1;
    }
    double *pos = (double *)malloc(sizeof(double) * 3 * D), *vel = pos + D, *acc = vel + D;
    double s, sd, sdd;
    interpolate_at_time(p, time_sec, smaller > 0 ? smaller : 0, pos, vel, acc, &s, &sd, &sdd);
    if (p->time.v[smaller] < time_sec + 0.95 * p->time_step_sec) erase_samples_until(p, smaller);
    else erase_samples_until(p, smaller - 1);
    p->time.v[0] = time_sec;
    memcpy(p->pos.v, pos, sizeof(double) * D);
    memcpy(p->vel.v, vel, sizeof(double) * D);
    memcpy(p->acc.v, acc, sizeof(double) * D);
    p->s.v[0] = s; p->sd.v[0] = sd; p->sdd.v[0] = sdd;
    free(pos);
  } else {
    int offset = (int)round((time_sec - p->time.v[0]) / p->time_step_sec);
    if (offset > (int)p->time.n - 1) offset = (int)p->time.n - 1;
    erase_samples_until(p, offset);
  }
}

/* :307-475 */
static int compute_timing_profile(tpo_planner *p, int64_t start, int64_t target_duration) {
  const double start_sec = time_to_sec(start);
  const int D = p->D, N = p->N;
  if (p->knots == NULL && p->table_q == NULL) return TPO_PLAN_FAILED_PRECONDITION;
  if (target_duration <= 0) return TPO_PLAN_INVALID_ARGUMENT;
  const int old_state = p->path_state;
  int offset = 0;
  if (old_state == TPO_PATH_NEW) {
    p->path_start = 0.0;
    p->path_start_velocity = 0.0;
    p->path_start_acceleration = 0.0;
    p->path_time_start = start_sec;
  } else {
    const int num = (int)p->t_ps.n;
    if (num == 0) return TPO_PLAN_FAILED_PRECONDITION; /* CHECK(!time_at_path_samples_.empty()) */
    int lo = 0, hi = num;                              /* lower_bound */
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (p->t_ps.v[mid] < start_sec) lo = mid + 1; else hi = mid;
    }
    offset = lo - 1;
    if (offset < 0) offset = 0;
    if (offset > num - 1) offset = num - 1;
    p->path_start = p->s_ps.v[offset];
    p->path_start_velocity = p->sd_ps.v[offset];
    p->path_time_start = p->t_ps.v[offset];
  }
  p->path_horizon = p->path_start + p->delta * (N - 1);
  const int C = p->table_q ? 2 * D + 2 : 2 * D;
  double *rows = (double *)malloc(sizeof(double) * 4 * (size_t)N * C);
  double *A = rows, *B = A + (size_t)N * C, *lo_ = B + (size_t)N * C, *hi_ = lo_ + (size_t)N * C;
  if (p->table_q) {
    /* SamplePath :527-542: the segment of the IK table that starts at PathIkIndex(path_start) */
    const int first = (int)round(p->path_start / p->delta);
    const int last = (int)round(p->path_horizon / p->delta);
    if (first < 0 || last - first != N - 1 || last >= p->table_len) { free(rows); return TPO_PLAN_INTERNAL; }
    memcpy(p->q, p->table_q + (size_t)first * D, sizeof(double) * (size_t)N * D);
    tpo_cartesian_path_derivatives(p->q, N, D, p->delta, p->q1, p->q2);
    double *jq1 = (double *)malloc(sizeof(double) * 6 * (size_t)N);
    tpo_cartesian_jacobian_times_q1(p->table_J + (size_t)first * 6 * D, p->q1, N, D, jq1);
    tpo_cartesian_constraint_setup(p->q1, p->q2, jq1, N, D, p->vmax, p->amax, p->max_trans_vel,
                                   p->max_rot_vel, p->safety, A, B, lo_, hi_);
    free(jq1);
  } else {
    if (tpo_joint_sample_path(p->knots, p->num_knots, p->cps, p->num_points, D, p->path_start, p->delta,
                              N, p->q, p->q1, p->q2) != 0) {
      free(rows);
      return TPO_PLAN_INTERNAL;
    }
    tpo_joint_constraint_setup(p->q1, p->q2, N, D, p->vmax, p->amax, p->safety, A, B, lo_, hi_);
  }
  p->path_state = TPO_PATH_SAMPLED;
  if (old_state == TPO_PATH_MODIFIED || old_state == TPO_PATH_NEW) {
    /* :360-393: least-squares start velocity along the start tangent; Eigen's squaredNorm and
     * dot are summed in index order here */
    double nrm2 = 0.0, dot = 0.0;
    for (int d = 0; d < D; d++) nrm2 += p->q1[d] * p->q1[d];
    if (nrm2 > 100 * 2.220446049250313e-16) {
      for (int d = 0; d < D; d++) dot += p->initial_velocity[d] * p->q1[d];
      const double v = dot / nrm2;
      p->path_start_velocity = (v > 0.0) ? v : 0.0;
    }
    double max_err = 0.0;
    for (int d = 0; d < D; d++) {
      const double e = fabs(p->q1[d] * p->path_start_velocity - p->initial_velocity[d]);
      if (e > max_err) max_err = e;
    }
    if (max_err > p->max_initial_velocity_error) { free(rows); return TPO_PLAN_INVALID_ARGUMENT; }
  }
  const int loops = (100 > 10 * N) ? 100 : 10 * N;
  tpo_profile_set_max_loops(p->profile, loops);
  if (tpo_profile_setup(p->profile, A, B, lo_, hi_, p->path_start, p->path_horizon,
                        p->path_start_velocity, p->path_start_acceleration,
                        p->path_time_start) != TPO_OK) {
    free(rows);
    return TPO_PLAN_INTERNAL;
  }
  const int rc = tpo_profile_optimize(p->profile);
  free(rows);
  if (rc != TPO_OK) return TPO_PLAN_INTERNAL;
  p->profile_solved = 1;
  p->windows++;
  /* erase what was replanned, append the new profile (:418-472) */
  dv_resize(&p->t_ps, (size_t)offset); dv_resize(&p->s_ps, (size_t)offset);
  dv_resize(&p->sd_ps, (size_t)offset); dv_resize(&p->sdd_ps, (size_t)offset);
  dv_resize(&p->q_ps, (size_t)offset * D); dv_resize(&p->qd_ps, (size_t)offset * D);
  dv_resize(&p->qdd_ps, (size_t)offset * D);
  dv_append(&p->t_ps, tpo_profile_time(p->profile), (size_t)N);
  dv_append(&p->s_ps, tpo_profile_s(p->profile), (size_t)N);
  dv_append(&p->sd_ps, tpo_profile_sd(p->profile), (size_t)N);
  dv_append(&p->sdd_ps, tpo_profile_sdd(p->profile), (size_t)N);
  double *qd = (double *)malloc(sizeof(double) * 2 * (size_t)N * D), *qdd = qd + (size_t)N * D;
  tpo_epilogue(p->q1, p->q2, N, D, tpo_profile_sd(p->profile), tpo_profile_sdd(p->profile), p->amax, qd,
               qdd);
  dv_append(&p->q_ps, p->q, (size_t)N * D);
  dv_append(&p->qd_ps, qd, (size_t)N * D);
  dv_append(&p->qdd_ps, qdd, (size_t)N * D);
  free(qd);
  return TPO_PLAN_OK;
}

/* :755-836 through the single-trajectory functions of tp_oracle.c */
static void resample_trajectory(tpo_planner *p, double start_sec) {
  const int D = p->D, S = (int)p->t_ps.n;
  int cap;
  if (p->sampling_method == 0) cap = tpo_resample_uniform_count(p->t_ps.v[S - 1], start_sec, p->time_step_sec);
  else cap = S + 1;
  if (cap < 1) cap = 1;
  dv_resize(&p->time, (size_t)cap); dv_resize(&p->s, (size_t)cap); dv_resize(&p->sd, (size_t)cap);
  dv_resize(&p->sdd, (size_t)cap);
  dv_resize(&p->pos, (size_t)cap * D); dv_resize(&p->vel, (size_t)cap * D); dv_resize(&p->acc, (size_t)cap * D);
  int M;
  if (p->sampling_method == 0)
    M = tpo_resample_uniform(p->t_ps.v, p->s_ps.v, p->sd_ps.v, p->sdd_ps.v, p->q_ps.v, p->qd_ps.v, p->qdd_ps.v,
                             S, D, start_sec, p->time_step_sec, p->amax, cap, p->time.v, p->s.v, p->sd.v,
                             p->sdd.v, p->pos.v, p->vel.v, p->acc.v);
  else
    M = tpo_resample_skip(p->t_ps.v, p->s_ps.v, p->sd_ps.v, p->sdd_ps.v, p->q_ps.v, p->qd_ps.v, p->qdd_ps.v, S,
                          D, start_sec, 0.95 * p->time_step_sec, p->amax, cap, p->time.v, p->s.v, p->sd.v,
                          p->sdd.v, p->pos.v, p->vel.v, p->acc.v);
  if (M > cap) M = cap;
  dv_resize(&p->time, (size_t)M); dv_resize(&p->s, (size_t)M); dv_resize(&p->sd, (size_t)M);
  dv_resize(&p->sdd, (size_t)M);
  dv_resize(&p->pos, (size_t)M * D); dv_resize(&p->vel, (size_t)M * D); dv_resize(&p->acc, (size_t)M * D);
}

/* :579-684 */
int tpo_planner_plan(tpo_planner *p, int64_t start, int64_t time_horizon) {
  const double start_sec = time_to_sec(start);
  const size_t D = (size_t)p->D;
  p->windows = 0;
  if (p->knots == NULL && p->table_q == NULL) return TPO_PLAN_FAILED_PRECONDITION;
  /* HandleTimeArguments :502-538 */
  if (p->initial_plan && start > p->end_time + seconds_to_duration(p->time_step_sec))
    return TPO_PLAN_OUT_OF_RANGE;
  if (!p->initial_plan) {
    p->start_time = start;
    p->end_time = start;
    p->path_start = 0.0;
  } else {
    if (start > p->end_time) return TPO_PLAN_INVALID_ARGUMENT;
    if (start < p->start_time) return TPO_PLAN_INVALID_ARGUMENT;
    p->start_time = start;
  }
  /* UpdatePathTrackingStatus :477-500 */
  p->target_reached = 0;
  p->planned_to_end = 0;
  if (!p->initial_plan) {
    p->path_horizon = 0;
    p->path_start = 0;
  } else {
    p->planned_to_end = close_to_end(p, p->path_horizon);
    if (p->planned_to_end) {
      if (p->path_state != TPO_PATH_NEW && p->path_state != TPO_PATH_MODIFIED) {
        p->target_reached = 1;
      } else {
        p->path_horizon = 0.0;
        p->path_time_start = 0.0;
        p->path_start = 0.0;
        p->path_start_velocity = 0.0;
        p->path_start_acceleration = 0.0;
        p->planned_to_end = 0;
      }
    }
  }
  const int planned_enough = (p->path_state != TPO_PATH_NEW) && (p->path_state != TPO_PATH_MODIFIED) &&
                             (p->final_decel_start >= start + time_horizon);
  if (p->time.n != 0 && planned_enough) {
    erase_trajectory_before(p, start);
    return TPO_PLAN_OK;
  }
  if (p->initial_plan) {
    /* GetTimeOffsetAfter :289-305 (upper_bound) */
    if (p->time.n == 0) return TPO_PLAN_FAILED_PRECONDITION;
    if (start_sec < p->time.v[0]) return TPO_PLAN_OUT_OF_RANGE;
    int lo = 0, hi = (int)p->time.n;
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (!(start_sec < p->time.v[mid])) lo = mid + 1; else hi = mid;
    }
    if (lo == (int)p->time.n) return TPO_PLAN_INTERNAL;
    const size_t offset = (size_t)lo;
    dv_resize(&p->time, offset); dv_resize(&p->s, offset); dv_resize(&p->sd, offset);
    dv_resize(&p->sdd, offset);
    dv_resize(&p->pos, offset * D); dv_resize(&p->vel, offset * D); dv_resize(&p->acc, offset * D);
  }
  int64_t loop_start_time = start;
  int time_horizon_reached = 0;
  for (int loop = 0; !p->planned_to_end && !time_horizon_reached; loop++) {
    const int rc = compute_timing_profile(p, loop_start_time, start + time_horizon - loop_start_time);
    if (rc != TPO_PLAN_OK) return rc;
    const int lei = tpo_profile_last_extremal_index(p->profile);
    const int decel_start = (lei > p->N / 2) ? lei : p->N / 2;
    const double *t = tpo_profile_time(p->profile);
    p->final_decel_start = time_from_sec(t[decel_start]);
    p->planned_to_end = close_to_end(p, p->path_horizon);
    time_horizon_reached = (t[p->N - 1] - time_to_sec(start)) > (double)time_horizon / (double)NSECS_PER_SEC;
    if (loop >= p->max_planning_iterations) return TPO_PLAN_DEADLINE_EXCEEDED;
    loop_start_time = p->final_decel_start;
  }
  resample_trajectory(p, start_sec);
  p->initial_plan = 1;
  if (p->time.n != 0) {
    p->end_time = time_from_sec(p->time.v[p->time.n - 1]);
    clamp_to_time_step_multiple(p, &p->end_time);
    const int decel_start = tpo_profile_last_extremal_index(p->profile);
    p->final_decel_start = time_from_sec(tpo_profile_time(p->profile)[decel_start]);
    clamp_to_time_step_multiple(p, &p->final_decel_start);
  } else {
    p->end_time = p->start_time;
    p->final_decel_start = p->end_time;
  }
  p->target_reached = p->planned_to_end;
  return TPO_PLAN_OK;
}

int tpo_planner_num_samples(const tpo_planner *p) { return (int)p->time.n; }
const double *tpo_planner_time(const tpo_planner *p) { return p->time.v; }
const double *tpo_planner_positions(const tpo_planner *p) { return p->pos.v; }
const double *tpo_planner_velocities(const tpo_planner *p) { return p->vel.v; }
const double *tpo_planner_accelerations(const tpo_planner *p) { return p->acc.v; }
const double *tpo_planner_path_parameter(const tpo_planner *p) { return p->s.v; }
const double *tpo_planner_path_velocity(const tpo_planner *p) { return p->sd.v; }
const double *tpo_planner_path_acceleration(const tpo_planner *p) { return p->sdd.v; }
int64_t tpo_planner_end_time(const tpo_planner *p) { return p->end_time; }
int64_t tpo_planner_final_decel_start(const tpo_planner *p) { return p->final_decel_start; }
int tpo_planner_target_reached(const tpo_planner *p) { return p->target_reached; }
int tpo_planner_windows(const tpo_planner *p) { return p->windows; }
int tpo_planner_path_state(const tpo_planner *p) { return p->path_state; }
