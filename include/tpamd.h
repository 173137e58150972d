/*
 * tpamd.h -- C-ABI of the MI355X (gfx950) batched time-optimal path-timing engine.
 *
 * The reference (theteamatx/x-edr-trajectory-planning) has no FFI: its boundary
 * is the C++ class API of trajectory_planning/. Every entry point below names
 * the reference interface (file:line under trajectory_planning/) whose work it
 * performs for a BATCH of independent paths. The C++ mirror classes in
 * x-edr-trajectory-planning_amd/host/ call these with B = 1 (drop-in) or B >> 1
 * (BatchPathTiming); INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *  - All arithmetic is fp64 (time_optimal_path_timing.h:38-41).
 *  - "_device" entry points take DEVICE pointers, enqueue work on the given HIP
 *    stream (hipStream_t passed as void*) and return without synchronising.
 *    "_host" entry points take HOST pointers, copy in, run, copy out and
 *    synchronise before returning.
 *  - Return value: 0 on success, a negative TPAMD_E_* code for call-level errors
 *    (bad arguments, HIP failure). Per-path solver outcomes are written to the
 *    status[] array (TPAMD_PATH_*), in the reference's order of checks.
 *  - No exceptions cross this boundary. An engine handle is not thread-safe;
 *    distinct engines may be used from distinct threads.
 */
#ifndef TPAMD_H_
#define TPAMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPAMD_VERSION 200

/* call-level errors */
#define TPAMD_E_INVALID_ARGUMENT (-1)
#define TPAMD_E_HIP (-2)
#define TPAMD_E_UNSUPPORTED (-3)
#define TPAMD_E_NO_DEVICE (-4)
#define TPAMD_E_STALE (-5) /* engine-held state belongs to a different solve */

/* per-path status (status[b]); 0 = solved. Codes follow the reference's checks:
 * SetupProblem (time_optimal_path_timing.cc:165-193), IsSetupValid (:554-576),
 * OptimizePathParameter (:383-391, :400-403, :422-428). */
#define TPAMD_PATH_OK 0
#define TPAMD_PATH_INFEASIBLE_BOUNDS 2
#define TPAMD_PATH_S_RANGE 3
#define TPAMD_PATH_SD_START_NEGATIVE 4
#define TPAMD_PATH_LOWER_GE_UPPER 5
#define TPAMD_PATH_TOO_FEW_SAMPLES 6
#define TPAMD_PATH_NO_CONNECTION 7
#define TPAMD_PATH_NAN_SD2 8
#define TPAMD_PATH_NONZERO_END 9
#define TPAMD_PATH_CRIT_INDEX_ZERO 10 /* reference reads sd2_max[-1] here (.cc:361,:372) */

typedef struct tpamd_engine tpamd_engine;

/* Engine lifetime. An engine owns a device workspace that grows on demand
 * (outside any timed region once warmed up) and is bound to one HIP device. */
int tpamd_engine_create(int device_ordinal, tpamd_engine **out);
void tpamd_engine_destroy(tpamd_engine *engine);
int tpamd_version(void);
const char *tpamd_error_string(int code);
/* HIP devices visible to the library (0: none, the engine has no CPU fallback). */
int tpamd_device_count(void);
/* Pre-size the workspace for batches up to (num_paths, num_samples, num_rows). */
int tpamd_engine_reserve(tpamd_engine *engine, int num_paths, int num_samples,
                         int num_rows);
/* Pipelined modes for streams of joint-space solves (tpamd_time_joint_paths_device). The engine
 * keeps two workspaces, used alternately.
 *   1: the front stage of a solve (set-up and the sampling / LP kernel, which also writes out->q)
 *      runs on a stream of the engine, so that it overlaps the extremal sweep of the PREVIOUS
 *      solve; the sweep and every other output stay on the caller's stream, in order.
 *   2: the sweep runs on one of two engine streams as well, ordered behind the call's position in
 *      the caller's stream and behind its own front stage, so that it can start while the slowest
 *      paths of the previous solve are still running. The caller's stream is ordered behind the
 *      PREVIOUS solve when a call returns; tpamd_engine_fence orders it behind all of them.
 * Contract while a mode is on: the inputs of a call (and its out->q buffer) must be ready when the
 * call is made -- the front stage is NOT ordered behind earlier work on the caller's stream -- and
 * must stay untouched until the caller's stream has passed the call (mode 2: the fence). Only
 * tpamd_time_joint_paths_device is pipelined: the _host entry points and every other solve
 * (rows, Cartesian, planning windows, joint groups) run in order on the caller's stream and are
 * ordered against the engine's streams where they share a workspace with pipelined solves. Calls
 * captured into a HIP graph run unpipelined; fence and synchronise the engine before capturing
 * (a captured stream cannot wait for work outside the capture). 0 = off (default); changing the
 * mode waits for the engine's streams. */
int tpamd_engine_set_pipelining(tpamd_engine *engine, int mode);
/* Make hip_stream wait for every solve issued so far (needed in mode 2 before the outputs of the
 * last solve are used; harmless otherwise). */
int tpamd_engine_fence(tpamd_engine *engine, void *hip_stream);
/* Bytes of device workspace currently held. */
size_t tpamd_engine_workspace_bytes(const tpamd_engine *engine);

/* ------------------------------------------------------------------------
 * Form (i): joint-space degree-2 B-spline paths.
 * Performs, per path, what PathTimingTrajectory::ComputeTimingProfile runs for a
 * TimeableJointSplinePath (path_timing_trajectory.cc:307-475):
 *   SamplePath          timeable_path_joint_spline.cc:294-318
 *                       (BSplineBase::KnotSpan bspline_base.cc:218-246,
 *                        UpdateBasisAndDerivatives :268-348,
 *                        BSplineT::EvalCurveAndDerivatives bspline.h:540-568)
 *   ConstraintSetup     timeable_path_joint_spline.cc:320-343 (C = 2D rows)
 *   InitSolver/SetupProblem/SetSetupDone  time_optimal_path_timing.cc:135-203,:535-576
 *                       with s_start = path_start, s_end = path_start + delta*(N-1)
 *   OptimizePathParameter                  time_optimal_path_timing.cc:287-490
 *   epilogue qd = q'*sd, qdd = clamp(q'*sdd + q''*sd^2, +-a_max)  path_timing_trajectory.cc:458-472
 * ------------------------------------------------------------------------ */
typedef struct tpamd_joint_batch {
  int32_t num_paths;        /* B */
  int32_t num_dofs;         /* D, 1..16 */
  int32_t num_samples;      /* N, 3..8192 (JointPathOptions::num_path_samples) */
  int32_t num_points;       /* P control points per path; P+3 knots */
  int32_t max_solver_loops; /* <=0: max(100, 10*N) as path_timing_trajectory.cc:398-400 */
  int32_t reserved;
  double constraint_safety; /* PathOptions::constraint_safety, timeable_path.h:80 */
} tpamd_joint_batch;

typedef struct tpamd_joint_inputs {
  const double *knots;          /* [B][P+3] */
  const double *control_points; /* [B][P][D] */
  const double *max_velocity;   /* [B][D]  TimeablePath::SetMaxJointVelocity */
  const double *max_acceleration; /* [B][D] TimeablePath::SetMaxJointAcceleration */
  const double *path_start;     /* [B]  SamplePath(path_start) */
  const double *delta;          /* [B]  PathOptions::delta_parameter */
  const double *sd_start;       /* [B]  SetupProblem sd_start */
  const double *sdd_start;      /* [B]  SetupProblem sdd_start; NULL = 0 */
  const double *time_start;     /* [B]  SetupProblem time_start */
  /* Ragged batches (BASELINE.json configs[4]): samples of each path, 3 <= n[b] <=
   * num_samples; NULL = every path has num_samples. All [B][N]-shaped arrays keep the
   * stride num_samples; entries beyond n[b] are not written. The sweep then takes the paths
   * longest first (a device-side counting sort), whatever their order in the batch. */
  const int32_t *num_samples_per_path; /* [B] or NULL */
} tpamd_joint_inputs;

typedef struct tpamd_path_outputs {
  double *time;  /* [B][N] GetTimeSamples()      */
  double *s;     /* [B][N] GetPathParameter()    */
  double *sd;    /* [B][N] GetPathVelocity()     */
  double *sdd;   /* [B][N] GetPathAcceleration() */
  double *q;     /* [B][N][D] GetPathPositionAt(i); may be NULL */
  double *qd;    /* [B][N][D] velocity_at_path_samples_; may be NULL */
  double *qdd;   /* [B][N][D] acceleration_at_path_samples_; may be NULL */
  int32_t *last_extremal_index; /* [B] GetLastExtremalIndex(); may be NULL */
  double *max_time_increment;   /* [B] GetMaxTimeIncrement(); may be NULL */
  int32_t *status;              /* [B] TPAMD_PATH_* */
  double *sd2;                  /* [B][N] squared path velocity sd2_ (sd = sqrt(sd2)); may be NULL */
} tpamd_path_outputs;

int tpamd_time_joint_paths_device(tpamd_engine *engine, const tpamd_joint_batch *batch,
                                  const tpamd_joint_inputs *in,
                                  const tpamd_path_outputs *out, void *hip_stream);
int tpamd_time_joint_paths_host(tpamd_engine *engine, const tpamd_joint_batch *batch,
                                const tpamd_joint_inputs *in,
                                const tpamd_path_outputs *out);

/* Several joint-space batches solved CONCURRENTLY (BASELINE.json configs[4]: a mixed 6/7/14-joint
 * batch with 500..4000 samples per path). The kernels are specialised on the joint count, and a
 * launch's LDS is sized by its sample stride, so a mixed batch is bucketed by the caller into
 * groups of one (num_dofs, num_points, stride) each -- BatchPathTiming buckets by
 * (D, P, ceil(N / 512)) -- and every group is what tpamd_time_joint_paths_* takes. Here the groups
 * of one call run side by side: the groups are taken heaviest first (stride x joints) and dealt
 * to the engine's lanes (a stream and a workspace of the engine's own each; four, the number of
 * hardware queues the runtime uses), the heaviest on the highest-priority lane; their sampling/LP
 * kernels run one after another in that order, their sweeps overlap. The
 * lanes fork from hip_stream's position at the call and hip_stream waits for all of them before it
 * goes on, so for the caller the call behaves like one solve on hip_stream. Within a ragged group
 * the sweep takes the paths longest first (as tpamd_time_joint_paths_* does). Never pipelined.
 * Same per-path results as separate calls, bit for bit. batches/inputs/outputs: [num_groups]. */
int tpamd_time_joint_groups_device(tpamd_engine *engine, int num_groups,
                                   const tpamd_joint_batch *batches,
                                   const tpamd_joint_inputs *inputs,
                                   const tpamd_path_outputs *outputs, void *hip_stream);
int tpamd_time_joint_groups_host(tpamd_engine *engine, int num_groups,
                                 const tpamd_joint_batch *batches, const tpamd_joint_inputs *inputs,
                                 const tpamd_path_outputs *outputs);

/* Stand-alone batched TimeableJointSplinePath::SamplePath
 * (timeable_path_joint_spline.cc:294-318): q, q' = dq/ds, q'' = d2q/ds2 at
 * path_start + i*delta, i < N, as [B][N][D] host arrays (GetPathPositionAt,
 * GetFirstPathDerivativeAt, GetSecondPathDerivativeAt). */
int tpamd_sample_joint_paths_host(tpamd_engine *engine, int num_paths, int num_dofs,
                                  int num_samples, int num_points, const double *knots,
                                  const double *control_points, const double *path_start,
                                  const double *delta, double *q, double *q1, double *q2);

/* ------------------------------------------------------------------------
 * Form (ii): explicit constraint rows  lower <= A*sdd + B*sd^2 <= upper.
 * Batched TimeOptimalPathProfile::InitSolver + SetupProblem +
 * OptimizePathParameter (time_optimal_path_timing.h:118-150). Row arrays are
 * [B][N][C] (sample-major, row-minor): Constraint::a_coefficient/b_coefficient/
 * lower/upper (time_optimal_path_timing.h:65-102). q/qd/qdd outputs are unused.
 * ------------------------------------------------------------------------ */
typedef struct tpamd_rows_batch {
  int32_t num_paths;        /* B */
  int32_t num_samples;      /* N */
  int32_t num_rows;         /* C, 1..64 */
  int32_t max_solver_loops; /* <=0: 100 (time_optimal_path_timing.h:339) */
} tpamd_rows_batch;

typedef struct tpamd_rows_inputs {
  const double *a;      /* [B][N][C] */
  const double *b;      /* [B][N][C] */
  const double *lower;  /* [B][N][C] */
  const double *upper;  /* [B][N][C] */
  const double *s_start, *s_end, *sd_start, *sdd_start, *time_start; /* [B] each */
} tpamd_rows_inputs;

int tpamd_optimize_rows_device(tpamd_engine *engine, const tpamd_rows_batch *batch,
                               const tpamd_rows_inputs *in,
                               const tpamd_path_outputs *out, void *hip_stream);
int tpamd_optimize_rows_host(tpamd_engine *engine, const tpamd_rows_batch *batch,
                             const tpamd_rows_inputs *in,
                             const tpamd_path_outputs *out);

/* ------------------------------------------------------------------------
 * Form (iii): Cartesian-space paths after the IK callback (BASELINE.json configs[3]).
 * Replaces, for B paths at once, the arithmetic of TimeableCartesianSplinePath that
 * follows path_ik_func_: ComputePathDerivatives (timeable_path_cartesian_spline.cc:39-68,
 * called from SamplePath :541-542), ConstraintSetup (:551-595, C = 2D+2 rows), then
 * the solver and the planner epilogue as for joint paths. The pose-spline sampling, the
 * IK callback and the Jacobian callback are user std::functions (:508-510, :576) and stay
 * on the host: their results are the inputs here. J*q' is summed over the dofs in index
 * order. max_solver_loops <= 0: max(100, 10 N).
 * ------------------------------------------------------------------------ */
typedef struct tpamd_cartesian_batch {
  int32_t num_paths;        /* B */
  int32_t num_dofs;         /* D, 1..16 */
  int32_t num_samples;      /* N */
  int32_t max_solver_loops;
  double constraint_safety; /* CartesianPathOptions::constraint_safety */
} tpamd_cartesian_batch;

typedef struct tpamd_cartesian_inputs {
  const double *ik_positions;  /* [B][N][D]    path_position_ (IK solution per sample) */
  const double *jacobians;     /* [B][N][6][D] jacobian_func_(path_position_[i]), row-major */
  const double *max_velocity;      /* [B][D] */
  const double *max_acceleration;  /* [B][D] */
  const double *max_translational_velocity; /* [B] */
  const double *max_rotational_velocity;    /* [B] */
  const double *path_start, *delta, *sd_start, *sdd_start, *time_start; /* [B] each; sdd_start may be NULL */
} tpamd_cartesian_inputs;

/* out->q, if given, receives a copy of ik_positions. */
int tpamd_time_cartesian_paths_device(tpamd_engine *engine, const tpamd_cartesian_batch *batch,
                                      const tpamd_cartesian_inputs *in,
                                      const tpamd_path_outputs *out, void *hip_stream);
int tpamd_time_cartesian_paths_host(tpamd_engine *engine, const tpamd_cartesian_batch *batch,
                                    const tpamd_cartesian_inputs *in,
                                    const tpamd_path_outputs *out);

/* Pose targets of Cartesian-space paths: what TimeableCartesianSplinePath::SamplePath evaluates
 * before it calls the IK callback (timeable_path_cartesian_spline.cc:484-503), for B paths at
 * once: the degree-2 translation spline (BSplineT::EvalCurve, splines/bspline.h:512-536) and the
 * degree-2 quaternion spline (BSplineQ::EvalCurve, splines/bsplineq.cc:223-244: cumulative basis
 * :309-317, QuatPower = exp(p log q) :112-146) on a shared knot vector, at
 * path_start[b] + i * delta[b], i < N; beyond knots.back() - delta the last control pose is
 * repeated. knots [B][P+3], translation_points [B][P][3], rotation_points [B][P][4] as
 * (w, x, y, z), poses [B][N][7] = (tx, ty, tz, qw, qx, qy, qz). The IK and Jacobian callbacks
 * stay with the caller; their results go to tpamd_time_cartesian_paths_*. */
int tpamd_sample_pose_splines_device(tpamd_engine *engine, int num_paths, int num_samples,
                                     int num_points, const double *knots,
                                     const double *translation_points,
                                     const double *rotation_points, const double *path_start,
                                     const double *delta, double *poses, void *hip_stream);
int tpamd_sample_pose_splines_host(tpamd_engine *engine, int num_paths, int num_samples,
                                   int num_points, const double *knots,
                                   const double *translation_points, const double *rotation_points,
                                   const double *path_start, const double *delta, double *poses);

/* ------------------------------------------------------------------------
 * Receding-horizon planning: the window loop of PathTimingTrajectory::Plan
 * (path_timing_trajectory.cc:628-660 around ComputeTimingProfile :307-475) for B planners with
 * TimeableJointSplinePath paths of one shape, CHAINED ON THE DEVICE. Per iteration and planner:
 * the window start is looked up in the planner's window history (the sample before the loop's
 * start time, :328-339; a new path starts at 0), the path is sampled and solved from there, the
 * start velocity of a new or modified path is projected on the start tangent (:360-393), the
 * window replaces the tail of the history (:418-472), and the loop continues from the start of
 * the final deceleration -- max(last_extremal_index, N/2) (:639-646), converted with the
 * nanosecond truncation of trajectory_planning/time.h:22-29 -- until the path's end is planned
 * (CloseToEnd, timeable_path_joint_spline.cc:142-144) or the time horizon is covered (:650-652).
 * Only the number of planners still looping crosses PCIe per iteration; histories and results
 * cross once per call. Host pointers; in/out arrays carry planner state between calls.
 * ------------------------------------------------------------------------ */
#define TPAMD_PLAN_OK 0
#define TPAMD_PLAN_FAILED_PRECONDITION 1 /* nothing to connect to (:324) */
#define TPAMD_PLAN_OUT_OF_RANGE 2
#define TPAMD_PLAN_INVALID_ARGUMENT 3    /* non-positive duration (:313-317), start velocity (:387-392) */
#define TPAMD_PLAN_INTERNAL 4            /* solver set-up / optimisation failed (:394-417) */
#define TPAMD_PLAN_DEADLINE_EXCEEDED 5   /* planning-loop limit (:655-658) */
#define TPAMD_PLAN_MORE 100              /* history_capacity exhausted: call again with more room */

typedef struct tpamd_plan_args {
  int32_t num_planners, num_dofs, num_samples, num_points;
  int32_t history_capacity;        /* samples per planner the history arrays hold (their stride) */
  int32_t max_planning_iterations; /* PathTimingTrajectoryOptions::GetMaxPlanningIterations */
  double constraint_safety;        /* PathOptions::constraint_safety */
  double max_initial_velocity_error;
  const double *knots, *control_points;          /* [B][P+3], [B][P][D] */
  const double *max_velocity, *max_acceleration; /* [B][D] */
  const double *delta;                           /* [B] PathOptions::delta_parameter */
  const double *initial_velocity;                /* [B][D] TimeablePath::GetInitialVelocity */
  const int64_t *start_ns, *horizon_ns;          /* [B] Plan(start, time_horizon) */
  /* planner state, in/out */
  int32_t *path_state;      /* [B] 1 kNewPath, 2 kModifiedPath, 3 kPathWasSampled */
  int32_t *planned_to_end;  /* [B] in: planned_to_end_ after UpdatePathTrackingStatus */
  int32_t *history_count;   /* [B] size of time_at_path_samples_ */
  double *history_time, *history_s, *history_sd, *history_sdd; /* [B][capacity] *_at_path_samples_ */
  double *history_q, *history_qd, *history_qdd;                /* [B][capacity][D] */
  double *path_horizon;             /* [B] path_horizon_ */
  int64_t *final_decel_start_ns;    /* [B] final_decel_start_ as left by the loop (:645-646) */
  /* the last window each planner solved in this call: profile_ and the path's samples (out) */
  double *window_time, *window_s, *window_sd, *window_sdd, *window_sd2; /* [B][N] */
  double *window_q, *window_q1, *window_q2;                             /* [B][N][D] */
  double *window_path_start, *window_sd_start, *window_time_start;      /* [B] path_start_, path_start_velocity_, path_time_start_ */
  int32_t *window_last_extremal_index;  /* [B] */
  double *window_max_time_increment;    /* [B] */
  int32_t *status;   /* [B] TPAMD_PLAN_* */
  int32_t *windows;  /* [B] windows solved in this call (0: the outputs above are untouched) */
  /* Continuation after TPAMD_PLAN_MORE: the loop state of every planner, written by every call;
   * with resume != 0 it is read back in (call again with larger history arrays, same other
   * arguments), and only planners with looping[b] != 0 go on. */
  int32_t resume;
  int64_t *loop_start_ns;  /* [B] loop_start_time of the next window (:659) */
  int32_t *loop_count;     /* [B] windows counted by the loop so far (:633) */
  int32_t *looping;        /* [B] */
} tpamd_plan_args;

int tpamd_plan_joint_windows_host(tpamd_engine *engine, const tpamd_plan_args *args);

/* ------------------------------------------------------------------------
 * Planner sets: B PathTimingTrajectory planners (path_timing_trajectory.h:91-186) with
 * TimeableJointSplinePath paths of one shape whose WHOLE state lives on the device between Plan
 * calls -- the spline and limits, the window history (*_at_path_samples_), the planner scalars
 * (path_horizon_, planned_to_end_, final_decel_start_, start/end time, initial_plan_ ...), the
 * profile of the last window, and the resampled trajectory (time_, positions_, ...).
 * tpamd_planner_set_plan is Plan(start, time_horizon) (path_timing_trajectory.cc:579-684) for all
 * of them at once, entirely on the device: HandleTimeArguments :502-538, UpdatePathTrackingStatus
 * :477-500, the "planned enough" branch with EraseTrajectoryBefore :540-577 (both sampling
 * methods), the truncation at GetTimeOffsetAfter :604-621, the window loop :628-660 chained as in
 * tpamd_plan_joint_windows_host, ResampleTrajectory :755-836 and the bookkeeping of :662-684.
 * A Plan call moves 16 bytes per planner up (the two time arguments) and one status + summary
 * record down, plus a few words per window iteration; trajectories come down only when asked for
 * (tpamd_planner_set_download_trajectory). The mirror's PathTimingTrajectorySet wraps this.
 * History and trajectory buffers grow on the device when a planner needs more room.
 * ------------------------------------------------------------------------ */
typedef struct tpamd_planner_set tpamd_planner_set;

typedef struct tpamd_planner_set_config {
  int32_t num_planners, num_dofs, num_samples, num_points;
  int32_t history_capacity;        /* samples per planner to start with; <= 0: 8 * num_samples */
  int32_t trajectory_capacity;     /* resampled samples per planner to start with; <= 0: 4096 */
  int32_t sampling_method;         /* 0 kUniformlyInTime, 1 kSkipSamplesCloserThanTimeStep */
  int32_t max_planning_iterations; /* PathTimingTrajectoryOptions::GetMaxPlanningIterations */
  double constraint_safety;        /* PathOptions::constraint_safety */
  double max_initial_velocity_error;
  int64_t time_step_ns;            /* TrajectoryPlannerOptions::GetTimeStep */
} tpamd_planner_set_config;

typedef struct tpamd_planner_summary {
  int64_t end_time_ns, final_decel_start_ns, start_time_ns; /* GetEndTime, GetFinalDecelStart, GetStartTime */
  int32_t num_samples;     /* GetNumTimeSamples */
  int32_t target_reached;  /* with path_state: IsTrajectoryAtEnd */
  int32_t planned_to_end;
  int32_t windows;         /* timing windows solved by this Plan call */
  int32_t path_state;      /* TimeablePath::State after the call (3 = kPathWasSampled) */
  int32_t history_count;   /* size of time_at_path_samples_ */
  int32_t status;          /* TPAMD_PLAN_* */
  int32_t reserved;
} tpamd_planner_summary;

int tpamd_planner_set_create(tpamd_engine *engine, const tpamd_planner_set_config *config,
                             tpamd_planner_set **out);
void tpamd_planner_set_destroy(tpamd_planner_set *set);
/* The paths of `count` planners (ids[count], or planners 0..count-1 if ids is NULL) after
 * SetWaypoints (path_state 1 = kNewPath) or SwitchToWaypointPath (2 = kModifiedPath), which stay on
 * the host (O(waypoints) spline edits): knots [count][P+3], control_points [count][P][D],
 * max_velocity / max_acceleration / initial_velocity [count][D], delta [count]. Host pointers. */
int tpamd_planner_set_upload_paths(tpamd_planner_set *set, int count, const int32_t *ids,
                                   const double *knots, const double *control_points,
                                   const double *max_velocity, const double *max_acceleration,
                                   const double *delta, const double *initial_velocity,
                                   const int32_t *path_state);
/* TrajectoryPlanner::Reset for the listed planners (ids NULL: all): no path, no plan. */
int tpamd_planner_set_reset(tpamd_planner_set *set, int count, const int32_t *ids);
/* Plan(start, time_horizon) for every planner: start_ns / horizon_ns [B] host arrays;
 * summary [B] (host, may be NULL) receives one record per planner, status included. */
int tpamd_planner_set_plan(tpamd_planner_set *set, const int64_t *start_ns, const int64_t *horizon_ns,
                           tpamd_planner_summary *summary);
/* The trajectory of one planner after its last Plan: samples first .. first + count - 1 of
 * GetTime / GetPathParameters / ...Derivatives [count] and GetPositions / GetVelocities /
 * GetAccelerations [count][D] into host arrays (any may be NULL). */
int tpamd_planner_set_download_trajectory(tpamd_planner_set *set, int planner, int first, int count,
                                          double *time, double *s, double *sd, double *sdd,
                                          double *q, double *qd, double *qdd);
/* Bytes the last tpamd_planner_set_plan call moved over PCIe (host to device, device to host). */
void tpamd_planner_set_last_plan_bytes(const tpamd_planner_set *set, size_t *host_to_device,
                                       size_t *device_to_host);
size_t tpamd_planner_set_device_bytes(const tpamd_planner_set *set);

/* Batched TimeOptimalPathProfile::FindMaxSd2Simplex (time_optimal_path_timing.cc:1149-1363)
 * on num_lps independent constraint sets of C rows each ([num_lps][C] arrays);
 * outputs sd2max/sddmax/sd2zero [num_lps]. Host pointers. */
int tpamd_find_max_sd2_host(tpamd_engine *engine, int num_lps, int num_rows,
                            const double *a, const double *b, const double *lower,
                            const double *upper, double *sd2max, double *sddmax,
                            double *sd2zero);

/* ------------------------------------------------------------------------
 * s(t) query: batched TimeOptimalPathProfile::GetPathParameterAndDerivatives
 * (time_optimal_path_timing.cc:1549-1627) on solved profiles. For each path b
 * and query k: t_query[b][k] -> s, sd, sdd, ok [B][K]. time/s/sd/sd2 are the [B][N]
 * outputs of ONE solve (tpamd_path_outputs.time/.s/.sd/.sd2); the per-path ds, s_start and
 * s_end are recovered from the s rows. sd2 may be NULL: the engine then uses the copy of
 * sd2_ it keeps from its LAST solve, and returns TPAMD_E_STALE unless `time` is that solve's
 * output array and the shape matches (a later solve into other buffers invalidates it).
 * Every path has num_samples samples (no ragged batches). status may be NULL. Device pointers.
 * ------------------------------------------------------------------------ */
int tpamd_query_device(tpamd_engine *engine, int num_paths, int num_samples,
                       int num_queries, const double *time, const double *s,
                       const double *sd, const double *sd2, const int32_t *status,
                       const double *t_query, double *out_s, double *out_sd, double *out_sdd,
                       int32_t *ok, void *hip_stream);

/* ------------------------------------------------------------------------
 * Time samples rebuilt from the velocities of solved paths: time[i] = time[i-1] +
 * 2 ds / (sd[i-1] + sd[i]) (0 across a stationary pair), summed left to right as
 * TimeOptimalPathProfile::OptimizePathParameter does (time_optimal_path_timing.cc:447-455) --
 * bit-identical to the `time` output of the solve that produced sd. For the root of a multi-GPU
 * job: shards send (sd, sdd) and the two scalars (ds, time_start) per path, the root rebuilds
 * time. num_shards blocks of paths_per_shard paths each; block r keeps its arrays at
 * base + r * shard_stride (in doubles): sd [paths_per_shard][num_samples] at sd, ds and
 * time_start [paths_per_shard] at ds / time_start (pointers into shard 0). time_out is
 * [num_shards * paths_per_shard][num_samples], densely packed. num_samples_per_path (per global
 * path index) may be NULL. ds of a path = (s_end - s_start) / (n - 1) with
 * s_end = path_start + delta (n - 1) (path_timing_trajectory.cc:340-341). Device pointers.
 * ------------------------------------------------------------------------ */
int tpamd_rebuild_time_device(tpamd_engine *engine, int num_shards, int paths_per_shard,
                              int num_samples, size_t shard_stride, const double *sd,
                              const double *ds, const double *time_start,
                              const int32_t *num_samples_per_path, double *time_out,
                              void *hip_stream);

/* ------------------------------------------------------------------------
 * Sharding a batch of independent paths over several devices (the arithmetic of
 * sharding.shard_bounds / balanced_bounds, which bench.py uses across processes): contiguous
 * blocks of path indices, block k = [begin[k], begin[k+1]). begin has num_shards + 1 entries.
 * tpamd_shard_bounds: sizes differ by at most one. tpamd_shard_bounds_balanced: blocks of roughly
 * equal total cost for per-path costs (ragged batches: samples x rows^2), every block non-empty
 * while paths last. Host arithmetic only (no device needed).
 * ------------------------------------------------------------------------ */
void tpamd_shard_bounds(int num_paths, int num_shards, int32_t *begin);
void tpamd_shard_bounds_balanced(int num_paths, const double *cost, int num_shards, int32_t *begin);

/* ------------------------------------------------------------------------
 * Uniform-in-time resample: PathTimingTrajectory::ResampleEquidistantlyInTime
 * (path_timing_trajectory.cc:755-783) with InterpolateAtTime (:709-753) for a
 * batch of solved paths. Output row b holds count[b] = ceil((t_end-start)/dt)+1
 * samples, written at [b][0..count) of arrays with stride max_out; if
 * count[b] > max_out only max_out samples are written (count still reports the
 * full number). Device pointers.
 * ------------------------------------------------------------------------ */
typedef struct tpamd_resample_args {
  int32_t num_paths, num_samples, num_dofs, max_out;
  const double *time, *s, *sd, *sdd; /* [B][N] */
  const double *q, *qd, *qdd;        /* [B][N][D] */
  const double *max_acceleration;    /* [B][D] */
  const double *start_sec;           /* [B] */
  double time_step;
  const int32_t *status;             /* [B] paths with status != 0 are skipped; may be NULL */
  double *out_time, *out_s, *out_sd, *out_sdd; /* [B][max_out] */
  double *out_q, *out_qd, *out_qdd;            /* [B][max_out][D] */
  int32_t *count;                              /* [B] */
} tpamd_resample_args;

int tpamd_resample_uniform_device(tpamd_engine *engine, const tpamd_resample_args *args,
                                  void *hip_stream);
/* Same with HOST pointers in args (copies in, runs, copies out, synchronises). */
int tpamd_resample_uniform_host(tpamd_engine *engine, const tpamd_resample_args *args);

/* PathTimingTrajectory::ResampleSkippingSamplesCloserThanTimeStep
 * (path_timing_trajectory.cc:785-836; TimeSamplingMethod::kSkipSamplesCloserThanTimeStep):
 * the first output is interpolated at start_sec, then every path sample at least
 * 0.95 * time_step (GetMinTimeDeltaToKeep, :893-900) after the last kept one is taken
 * unchanged; the last output gets the end position and zero derivatives. Same argument
 * struct; count[b] = number of outputs of path b (at most num_samples + 1). */
int tpamd_resample_skip_device(tpamd_engine *engine, const tpamd_resample_args *args,
                               void *hip_stream);
int tpamd_resample_skip_host(tpamd_engine *engine, const tpamd_resample_args *args);

/* ------------------------------------------------------------------------
 * Debug/inspection: copy the boundary curve of the LAST solve to host arrays
 * [B][N] (Boundary::sd2_max, sdd_max_for_sd2_max, sdd_min_for_sd2_max,
 * sd2_max_for_sdd0, type; time_optimal_path_timing.h:225-255) and the squared
 * velocity sd2_. Any pointer may be NULL. Synchronises the device.
 * ------------------------------------------------------------------------ */
int tpamd_debug_copy_boundary(tpamd_engine *engine, int num_paths, int num_samples,
                              double *sd2_max, double *sdd_max, double *sdd_min,
                              double *sd2_zero, uint8_t *type, double *sd2);
/* The specialised joint-space sweep kernels run CalculateBoundary's passes 2-4 themselves and
 * keep sdd_max/sdd_min/type on chip; switch this on BEFORE a solve to have them stored for
 * tpamd_debug_copy_boundary as well (off by default: 17 bytes per sample less HBM traffic). */
void tpamd_debug_keep_boundary(tpamd_engine *engine, int on);

/* Diagnostic builds (-DTPAMD_DIAG) only: per-path counters of the specialised sweep kernel,
 * [B][64] int64: slots 0..31 of the backward wave, 32..63 of the forward wave (meaning of a
 * slot: csrc/tpamd_sweep_joint.h, JointSweep::diag). The product build leaves them zero. */
int tpamd_debug_copy_diag(tpamd_engine *engine, int num_paths, long long *out);
/* Registers per lane of the two hot kernels of the 7-joint path as the loaded code object
 * reports them (hipFuncGetAttributes): which = 0 the sampling/LP kernel, 1 the sweep kernel.
 * The pipelined modes rely on 2 x sweep + 1 x sampling/LP <= 512 (one SIMD's register file);
 * tests/test_gpu_configs.py checks it. Negative: error code. */
int tpamd_debug_kernel_vgprs(tpamd_engine *engine, int which);

/* Per-kernel launch durations for bench.py: HIP events recorded on the launch stream
 * around each kernel (enable = 1) or around the sweep kernel only (enable = 2: two events
 * per solve, so that the measurement barely disturbs the timed region); 0 switches it off.
 * tpamd_profile_mean_ms returns the mean duration in milliseconds over the launches since
 * the last reset (0 if none). */
void tpamd_profile_reset(tpamd_engine *engine);
void tpamd_profile_enable(tpamd_engine *engine, int enable);
double tpamd_profile_mean_ms(tpamd_engine *engine, int kernel_index, int *num_launches);
const char *tpamd_profile_kernel_name(int kernel_index);
int tpamd_profile_num_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* TPAMD_H_ */
