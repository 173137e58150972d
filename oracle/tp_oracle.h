/*
 * tp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded restatement of the reference algorithm for the
 * time-optimal path-timing hot path of theteamatx/x-edr-trajectory-planning.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the shipped engine (x-edr-trajectory-planning_amd/csrc)
 * never links or calls it.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - B-spline evaluation: PINNED by the reference's Mathematica golden tables
 *     (tests/golden/bspline_golden.json, from splines/bspline_test.cc:97-726).
 *   - 2-variable LP (FindMaxSd2Simplex): PINNED by the reference's five literal
 *     30-row LPs (tests/golden/lp_regression.json) against the restated
 *     brute-force solver (the reference's own cross-check) and against an
 *     independent LP solver (scipy HiGHS) in tests/.
 *   - Full solver output s(t), sd(t), sdd(t): PARITY UNPINNED by stored
 *     reference numbers. The reference's tests hold no expected solver outputs
 *     (only property checks, reproduced in tests/), and the reference cannot be
 *     built in this image (needs Eigen 3.4, abseil, eigenmath; no stand-ins are
 *     written for them).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * trajectory_planning/ of the reference tree).
 *
 * Layout conventions: constraint rows are four arrays A, B, lower, upper of
 * shape [N][C] (sample-major). Path samples q, q1 (=dq/ds), q2 (=d2q/ds2) are
 * [N][D].
 */
#ifndef TP_ORACLE_H_
#define TP_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* time_optimal_path_timing.h:275-279 */
#define TPO_KTINY (2.220446049250313e-16 * 1e5)
#define TPO_KMAXSD2 1e6

/* Boundary classification, time_optimal_path_timing.h:226-231 */
enum { TPO_BND_NONE = 0, TPO_BND_SOURCE = 1, TPO_BND_SINK = 2, TPO_BND_TRAJECTORY = 4 };

/* Per-path status. 0 = success. Order follows the reference's checks
 * (time_optimal_path_timing.cc:165-193, :554-576, :383-391, :400-403, :422-428). */
enum {
  TPO_OK = 0,
  TPO_ERR_INFEASIBLE_BOUNDS = 2, /* some sample has max(upper-lower) <= 0 (.cc:174-182) */
  TPO_ERR_S_RANGE = 3,           /* s_start >= s_end (.cc:185) */
  TPO_ERR_SD_START_NEG = 4,      /* sd_start < 0 (.cc:190) */
  TPO_ERR_LOWER_GE_UPPER = 5,    /* any lower >= upper (.cc:557) */
  TPO_ERR_TOO_FEW_SAMPLES = 6,   /* N < 2 (.cc:568-571) */
  TPO_ERR_NO_CONNECTION = 7,     /* "Could not connect from critical point" (.cc:383-391) */
  TPO_ERR_NAN_SD2 = 8,           /* residual NaN in sd2 (.cc:400-403) */
  TPO_ERR_NONZERO_END = 9,       /* non-zero terminal velocity (.cc:422-428) */
  TPO_ERR_CRIT_INDEX_ZERO = 10,  /* reference would read sd2_max[-1] (.cc:361,:372): UB there */
  TPO_ERR_NOT_SOLVED = 11
};

/* ---------------------------------------------------------------- splines */

/* splines/bspline_base.cc:218-246 */
int tpo_knot_span(const double *knots, int num_knots, int degree, double u);
/* splines/bspline_base.cc:249-265; basis[0..p] */
void tpo_basis(const double *knots, int span, int p, double u, double *basis);
/* splines/bspline_base.cc:268-348; ders is (der+1) x (p+1), row-major */
void tpo_basis_and_derivatives(const double *knots, int span, int p, int der,
                               double u, double *ders);
/* splines/bspline.h:515-537; points [num_points][dim]; returns 0 ok, 1 out of range */
int tpo_eval_curve(const double *knots, int num_knots, int degree,
                   const double *points, int dim, double u, double *value);
/* splines/bspline.h:540-568; values [(nvalues)][dim] */
int tpo_eval_curve_and_derivatives(const double *knots, int num_knots, int degree,
                                   const double *points, int dim, double u,
                                   int nvalues, double *values);
/* splines/bspline_base.cc:356-381 */
int tpo_make_uniform_knots(int num_points, int degree, double low, double high,
                           double *knots /* num_points+degree+1 */);
/* splines/spline_utils.cc:25-45, :47-102 ; out [(3W-2)][D] (or 4 x D when W==1).
 * Returns number of control points written. */
int tpo_polyline_to_bspline3_waypoints(const double *corners, int W, int D,
                                       double radius, double *out);

/* --------------------------------------------------------- joint-space path */

/* timeable_path_joint_spline.cc:252-292. Writes control points [(3W-2)][D] and
 * knots [(3W-2)+3]. Returns the number of control points. */
int tpo_joint_fit_spline(const double *waypoints, int W, int D, double rounding,
                         double *control_points, double *knots);
/* timeable_path_joint_spline.cc:294-318 */
int tpo_joint_sample_path(const double *knots, int num_knots,
                          const double *control_points, int num_points, int D,
                          double path_start, double delta, int N, double *q,
                          double *q1, double *q2);
/* timeable_path_joint_spline.cc:320-343; C = 2D */
void tpo_joint_constraint_setup(const double *q1, const double *q2, int N, int D,
                                const double *vmax, const double *amax,
                                double safety, double *A, double *B,
                                double *lower, double *upper);
/* timeable_path_cartesian_spline.cc:39-68 */
void tpo_cartesian_path_derivatives(const double *q, int N, int D, double delta,
                                    double *q1, double *q2);
/* timeable_path_cartesian_spline.cc:551-595; C = 2D+2; jq1 is J*q1, [N][6] */
void tpo_cartesian_constraint_setup(const double *q1, const double *q2,
                                    const double *jq1, int N, int D,
                                    const double *vmax, const double *amax,
                                    double max_trans_vel, double max_rot_vel,
                                    double safety, double *A, double *B,
                                    double *lower, double *upper);

/* ------------------------------------------------------------------ solver */

typedef struct tpo_profile tpo_profile;

/* InitSolver, time_optimal_path_timing.cc:135-159 */
tpo_profile *tpo_profile_create(int num_samples, int num_constraints);
void tpo_profile_destroy(tpo_profile *p);
/* SetMaxNumSolverLoops, .cc:205-207 */
void tpo_profile_set_max_loops(tpo_profile *p, int loops);
/* SetupProblem + SetSetupDone + IsSetupValid, .cc:161-203, :535-576 */
int tpo_profile_setup(tpo_profile *p, const double *A, const double *B,
                      const double *lower, const double *upper, double s_start,
                      double s_end, double sd_start, double sdd_start,
                      double time_start);
/* OptimizePathParameter, .cc:287-490. Returns a TPO_* status. */
int tpo_profile_optimize(tpo_profile *p);
/* CalculateBoundary only (.cc:1365-1487); for stage-wise parity tests. */
int tpo_profile_calculate_boundary(tpo_profile *p);

const double *tpo_profile_time(const tpo_profile *p);
const double *tpo_profile_s(const tpo_profile *p);
const double *tpo_profile_sd(const tpo_profile *p);
const double *tpo_profile_sdd(const tpo_profile *p);
const double *tpo_profile_sd2(const tpo_profile *p);
const double *tpo_profile_sd2_max(const tpo_profile *p);
const double *tpo_profile_sdd_max_for_sd2_max(const tpo_profile *p);
const double *tpo_profile_sdd_min_for_sd2_max(const tpo_profile *p);
const double *tpo_profile_sd2_max_for_sdd0(const tpo_profile *p);
const uint8_t *tpo_profile_boundary_type(const tpo_profile *p);
int tpo_profile_last_extremal_index(const tpo_profile *p);
double tpo_profile_max_time_increment(const tpo_profile *p);
int tpo_profile_num_loops_used(const tpo_profile *p);
/* SolutionSatisfiesConstraints, .cc:492-518; returns number of violations, -1 if unsolved */
int tpo_profile_constraint_violations(const tpo_profile *p);
/* GetPathParameterAndDerivatives, .cc:1549-1627; returns 1 on success */
int tpo_profile_query(const tpo_profile *p, double t, double *s, double *sd,
                      double *sdd);
/* GetPreviousIndex, .cc:1629-1645 */
int tpo_profile_previous_index(const tpo_profile *p, double t);

/* FindMaxSd2Simplex (.cc:1149-1363) and FindMaxSd2BruteForce (.cc:1010-1103)
 * on one constraint set of C rows. */
void tpo_find_max_sd2_simplex(const double *A, const double *B,
                              const double *lower, const double *upper, int C,
                              double *sd2max, double *sddmax, double *sd2zero);
void tpo_find_max_sd2_bruteforce(const double *A, const double *B,
                                 const double *lower, const double *upper, int C,
                                 double *sd2max, double *sddmax, double *sd2zero);
/* FindSddMax / FindSddMin (.cc:638-695) on one constraint set. */
double tpo_find_sdd_max(const double *A, const double *B, const double *lower,
                        const double *upper, int C, double sd2);
double tpo_find_sdd_min(const double *A, const double *B, const double *lower,
                        const double *upper, int C, double sd2);

/* Debug counters of the sweep (single-threaded use only); see tp_oracle.c. */
void tpo_debug_counters(long long *out8, int reset);

/* ------------------------------------------------- planner epilogue/resample */

/* path_timing_trajectory.cc:458-472 : qd = q1*sd ; qdd = clamp(q1*sdd + q2*sd^2) */
void tpo_epilogue(const double *q1, const double *q2, int N, int D,
                  const double *sd, const double *sdd, const double *amax,
                  double *qd, double *qdd);
/* ResampleEquidistantlyInTime, path_timing_trajectory.cc:755-783 with
 * InterpolateAtTime :709-753 and TimeAtPathSamplesLowerIndex :686-695.
 * Returns the number of uniform samples (ceil(T/dt)+1); writes at most max_out.
 * Outputs: t[M], s[M], sd[M], sdd[M], q[M][D], qd[M][D], qdd[M][D]. */
int tpo_resample_uniform(const double *time, const double *s, const double *sd,
                         const double *sdd, const double *q, const double *qd,
                         const double *qdd, int N, int D, double start_sec,
                         double time_step, const double *amax, int max_out,
                         double *ot, double *os, double *osd, double *osdd,
                         double *oq, double *oqd, double *oqdd);
/* ResampleSkippingSamplesCloserThanTimeStep, path_timing_trajectory.cc:785-836;
 * min_delta = GetMinTimeDeltaToKeep() = 0.95 * time step (path_timing_trajectory.cc:893-900).
 * Returns the number of output samples; writes at most max_out. */
int tpo_resample_skip(const double *time, const double *s, const double *sd,
                      const double *sdd, const double *q, const double *qd,
                      const double *qdd, int N, int D, double start_sec,
                      double min_delta, const double *amax, int max_out,
                      double *ot, double *os, double *osd, double *osdd,
                      double *oq, double *oqd, double *oqdd);
/* Number of uniform samples ResampleEquidistantlyInTime would produce (:756-757). */
int tpo_resample_uniform_count(double end_time, double start_sec, double time_step);

/* ------------------------------------------- whole hot path, one joint path */

/* One "timing" as PathTimingTrajectory::ComputeTimingProfile runs it for a new
 * joint path (path_timing_trajectory.cc:307-475): SamplePath -> ConstraintSetup
 * -> InitSolver/SetupProblem (s in [path_start, path_start+delta*(N-1)],
 * max loops = max(100, 10N)) -> OptimizePathParameter -> epilogue.
 * Outputs (caller-allocated): t,s,sd,sdd [N]; q,qd,qdd [N][D]. Returns TPO_*.
 * work may be NULL (allocates) or a tpo_profile created for (N, 2D). */
int tpo_time_joint_path(const double *knots, int num_knots,
                        const double *control_points, int num_points, int D,
                        const double *vmax, const double *amax, double safety,
                        double path_start, double delta, int N, double sd_start,
                        double sdd_start, double time_start, tpo_profile *work,
                        double *t, double *s, double *sd, double *sdd, double *q,
                        double *qd, double *qdd, int *last_extremal_index);

/* Batched driver for the CPU baseline: B uniform-shape paths, packed
 * [B][...] arrays, nthreads OpenMP threads (1 = the reference's execution
 * model). status[B]. Returns number of failed paths. */
int tpo_time_joint_batch(int B, const double *knots, int num_knots,
                         const double *control_points, int num_points, int D,
                         const double *vmax, const double *amax, double safety,
                         const double *path_start, const double *delta, int N,
                         const double *sd_start, const double *time_start,
                         int nthreads, double *t, double *s, double *sd,
                         double *sdd, double *q, double *qd, double *qdd,
                         int *last_extremal_index, int *status);

/* Cartesian path after the IK callback (see tp_oracle.c): q [N][D], J [N][6][D]. */
void tpo_cartesian_jacobian_times_q1(const double *J, const double *q1, int N, int D,
                                     double *jq1);
int tpo_time_cartesian_path(const double *q, const double *J, int N, int D,
                            const double *vmax, const double *amax, double max_trans_vel,
                            double max_rot_vel, double safety, double path_start,
                            double delta, double sd_start, double sdd_start,
                            double time_start, double *t, double *s, double *sd,
                            double *sdd, double *qd, double *qdd,
                            int *last_extremal_index);
int tpo_time_cartesian_batch(int B, const double *q, const double *J, int N, int D,
                             const double *vmax, const double *amax,
                             const double *max_trans_vel, const double *max_rot_vel,
                             double safety, const double *path_start, const double *delta,
                             const double *sd_start, const double *time_start, int nthreads,
                             double *t, double *s, double *sd, double *sdd, double *qd,
                             double *qdd, int *last_extremal_index, int *status);

/* ----------------------------------------------------- quaternion B-splines */

/* splines/bsplineq.cc: QuatLog :112-125, QuatExp :127-134, QuatPower :136-146, BSplineQ::EvalCurve
 * :223-244 (tp_oracle_quat.c). Quaternions are [w, x, y, z]; points [num_points][4]. */
void tpo_quat_log(const double *q, double *out);
void tpo_quat_exp(const double *q, double *out);
void tpo_quat_power(const double *q, double power, double *out);
int tpo_bsplineq_eval_curve(const double *knots, int num_knots, int degree, const double *points,
                            double u, double *quat);
/* pose targets of TimeableCartesianSplinePath::SamplePath (timeable_path_cartesian_spline.cc:
 * 484-503): poses [N][7] = (tx, ty, tz, qw, qx, qy, qz) */
int tpo_sample_pose_spline(const double *knots, int num_knots, const double *translation_points,
                           const double *rotation_points, int num_points, double path_start,
                           double delta, int N, double *poses);

/* ------------------------------------------------ receding-horizon planner */

/* PathTimingTrajectory::Plan (path_timing_trajectory.cc:579-684) for ONE planner with a
 * TimeableJointSplinePath, restated in tp_oracle_plan.c on top of the single-window functions
 * above. Times and durations are int64 nanoseconds (trajectory_planning/time.h:22-29). */
typedef struct tpo_planner tpo_planner;
enum { TPO_PATH_NONE = 0, TPO_PATH_NEW = 1, TPO_PATH_MODIFIED = 2, TPO_PATH_SAMPLED = 3 };  /* timeable_path.h:94-103 */
enum {
  TPO_PLAN_OK = 0,
  TPO_PLAN_FAILED_PRECONDITION = 1, /* no path (.cc:582-584), nothing to connect to */
  TPO_PLAN_OUT_OF_RANGE = 2,        /* start beyond the previous plan (.cc:503-510, :296-298) */
  TPO_PLAN_INVALID_ARGUMENT = 3,    /* time order (.cc:517-535), duration (:313-317), start velocity (:387-392) */
  TPO_PLAN_INTERNAL = 4,            /* solver set-up / optimisation failed (.cc:394-417), :299-303 */
  TPO_PLAN_DEADLINE_EXCEEDED = 5    /* planning-loop limit (.cc:655-658) */
};
/* sampling_method: 0 kUniformlyInTime, 1 kSkipSamplesCloserThanTimeStep */
tpo_planner *tpo_planner_create(int D, int N, double delta, double safety, int64_t time_step_ns,
                                int sampling_method, int max_planning_iterations,
                                double max_initial_velocity_error);
void tpo_planner_destroy(tpo_planner *p);
void tpo_planner_set_limits(tpo_planner *p, const double *vmax, const double *amax);
void tpo_planner_set_initial_velocity(tpo_planner *p, const double *v);
/* the spline of the path and its state (TPO_PATH_NEW after SetWaypoints, TPO_PATH_MODIFIED
 * after SwitchToWaypointPath) */
void tpo_planner_set_spline(tpo_planner *p, const double *knots, int num_knots, const double *cps,
                            int num_points, int state);
/* instead of a spline: the IK table of a TimeableCartesianSplinePath (tp_oracle_plan.c):
 * ik_positions [M][D] at parameters 0, delta, 2 delta, ..., jacobians [M][6][D] row-major */
void tpo_planner_set_ik_table(tpo_planner *p, const double *ik_positions, const double *jacobians,
                              int num_table_samples, double path_end, double max_trans_vel,
                              double max_rot_vel, int state);
int tpo_planner_plan(tpo_planner *p, int64_t start_ns, int64_t time_horizon_ns);
int tpo_planner_num_samples(const tpo_planner *p);
const double *tpo_planner_time(const tpo_planner *p);
const double *tpo_planner_positions(const tpo_planner *p);     /* [M][D] */
const double *tpo_planner_velocities(const tpo_planner *p);
const double *tpo_planner_accelerations(const tpo_planner *p);
const double *tpo_planner_path_parameter(const tpo_planner *p);
const double *tpo_planner_path_velocity(const tpo_planner *p);
const double *tpo_planner_path_acceleration(const tpo_planner *p);
int64_t tpo_planner_end_time(const tpo_planner *p);
int64_t tpo_planner_final_decel_start(const tpo_planner *p);
int tpo_planner_target_reached(const tpo_planner *p);
int tpo_planner_windows(const tpo_planner *p);   /* timing windows computed by the last plan */
int tpo_planner_path_state(const tpo_planner *p);

#ifdef __cplusplus
}
#endif
#endif /* TP_ORACLE_H_ */
