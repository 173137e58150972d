// tpamd_planner_set.h -- B receding-horizon planners whose whole state lives on the device
// (include/tpamd.h tpamd_planner_set_*): the path (spline, limits), the window history
// (*_at_path_samples_), the planner scalars and the resampled trajectory of every
// PathTimingTrajectory stay in HBM between Plan calls; a Plan call moves the two time arguments up
// and one status + summary record per planner down.
//
// PathTimingTrajectory::Plan (path_timing_trajectory.cc:579-684) per planner, one thread each
// where it is control flow:
//   k_pset_prologue  HandleTimeArguments :502-538, UpdatePathTrackingStatus :477-500, the
//                    "already planned enough" branch with EraseTrajectoryBefore :540-577, and the
//                    truncation at GetTimeOffsetAfter :289-305 / :604-621
//   window loop      k_plan_begin / set-up / K1 / k_plan_project / sweep / k_plan_end /
//                    k_plan_append (tpamd_kernels.h), as in tpamd_plan_joint_windows_host
//   resample         k_resample / k_resample_skip over the histories (:755-836)
//   k_pset_epilogue  :662-684: end_time_, final_decel_start_ clamped to time-step multiples,
//                    target_reached_
// Times are int64 nanoseconds; TimeFromSec truncates seconds * 1e9, TimeToSec divides by 1e9
// (trajectory_planning/time.h:22-29).
#pragma once

#include "tpamd_kernels.h"

namespace tpamd {

enum { kPsetIdle = 0, kPsetEraseOnly = 1, kPsetWindows = 2 };

struct PlannerSetState {
  int B, N, D, K, cap, tcap, method, max_iterations;
  double time_step_sec;
  long long time_step_duration_ns;      // absl::Seconds(time_step_sec_): llround(s * 1e9)
  // path
  const double *knots;                  // [B][K]
  const double *amax;                   // [B][D]
  int *path_state;                      // [B]
  int *has_path;                        // [B]
  // history (*_at_path_samples_)
  int *count;
  double *h_time, *h_s, *h_sd, *h_sdd, *h_q, *h_qd, *h_qdd;
  // planner scalars
  int *initial_plan, *planned_to_end, *target_reached;
  double *path_horizon, *path_start, *path_start_velocity, *path_time_start;
  long long *start_time_ns, *end_time_ns, *final_decel_start_ns;
  // profile_ of the last window each planner solved
  const double *w_time;                 // [B][N]
  const int *w_lei;                     // [B]
  // resampled trajectory (time_, positions_, ...): samples t_first .. t_first + t_count - 1
  int *t_first, *t_count;
  double *t_time, *t_s, *t_sd, *t_sdd, *t_q, *t_qd, *t_qdd;
  // this call
  const long long *start_ns, *horizon_ns;
  int *mode, *status, *active, *finish;
  int *num_active;                      // [2]: planners that loop; "a history is full" flag
  int *resample_skip;                   // [B] != 0: no resample for this planner
  double *start_sec;                    // [B]
  int *resample_count;                  // [B]
};

__device__ __forceinline__ long long pset_time_from_sec(double s) { return (long long)(s * 1e9); }
__device__ __forceinline__ double pset_time_to_sec(long long t) { return (double)t / 1e9; }

// path_timing_trajectory.cc:686-695 on a planner's history
__device__ __forceinline__ int pset_lower_index(const double *ht, int n, int starting_index, double time) {
  for (int index = starting_index; index < n - 1; ++index)
    if (ht[index + 1] > time) return index;
  return n - 1;
}

// Everything of Plan() before the window loop. One thread per planner.
static __global__ void k_pset_prologue(PlannerSetState S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  const int D = S.D;
  S.mode[b] = kPsetIdle;
  S.active[b] = 0;
  S.finish[b] = 0;
  S.status[b] = kPlanOk;
  S.resample_skip[b] = 1;
  const long long start = S.start_ns[b], horizon = S.horizon_ns[b];
  const double start_sec = pset_time_to_sec(start);
  S.start_sec[b] = start_sec;
  if (!S.has_path[b]) { S.status[b] = kPlanFailedPrecondition; return; }     // :582-584
  // HandleTimeArguments :502-538
  if (S.initial_plan[b] && start > S.end_time_ns[b] + S.time_step_duration_ns) { S.status[b] = kPlanOutOfRange; return; }
  if (!S.initial_plan[b]) {
    S.start_time_ns[b] = start;
    S.end_time_ns[b] = start;
    S.path_start[b] = 0.0;
  } else {
    if (start > S.end_time_ns[b] || start < S.start_time_ns[b]) { S.status[b] = kPlanInvalidArgument; return; }
    S.start_time_ns[b] = start;
  }
  // UpdatePathTrackingStatus :477-500
  const int state = S.path_state[b];
  const bool fresh = (state == 1) || (state == 2);      // kNewPath / kModifiedPath
  int target_reached = 0, planned_to_end = 0;
  if (!S.initial_plan[b]) {
    S.path_horizon[b] = 0.0;
    S.path_start[b] = 0.0;
  } else {
    const double kend = S.knots[(size_t)b * S.K + S.K - 1];
    planned_to_end = S.path_horizon[b] >= kend - 1e-4;    // CloseToEnd
    if (planned_to_end) {
      if (!fresh) {
        target_reached = 1;
      } else {
        S.path_horizon[b] = 0.0; S.path_time_start[b] = 0.0; S.path_start[b] = 0.0;
        S.path_start_velocity[b] = 0.0;
        planned_to_end = 0;
      }
    }
  }
  S.target_reached[b] = target_reached;
  S.planned_to_end[b] = planned_to_end;
  double *tt = S.t_time + (size_t)b * S.tcap;
  const int first = S.t_first[b], tn = S.t_count[b];
  const bool planned_enough = !fresh && (S.final_decel_start_ns[b] >= start + horizon);
  if (tn > 0 && planned_enough) {
    // EraseTrajectoryBefore(start) :540-577, then done
    S.mode[b] = kPsetEraseOnly;
    const double *tm = tt + first;
    if (start_sec < tm[0]) return;
    int offset;
    if (S.method == 1) {       // kSkipSamplesCloserThanTimeStep
      int lo = 0, hi = tn;     // samples with a time stamp < start_sec (lower_bound)
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (tm[mid] < start_sec) lo = mid + 1; else hi = mid;
      }
      int smaller = min(lo, tn - 1);
      // InterpolateAtTime(start_sec, smaller) over the history (:709-753)
      const int hn = S.count[b];
      const double *ht = S.h_time + (size_t)b * S.cap;
      const int lower = pset_lower_index(ht, hn, max(smaller, 0), start_sec);
      const int upper = min(hn - 1, lower + 1);
      const double at = (fabs(ht[upper] - ht[lower]) < DBL_EPSILON) ? 0.5 : (start_sec - ht[lower]) / (ht[upper] - ht[lower]);
      offset = (tm[smaller] < start_sec + 0.95 * S.time_step_sec) ? smaller : smaller - 1;
      const int nf = first + max(min(offset, tn), 0);        // EraseSamplesUntil(offset) :868-880
      const size_t hl = (size_t)b * S.cap + lower, hu = (size_t)b * S.cap + upper, o = (size_t)b * S.tcap + nf;
      S.t_time[o] = start_sec;
      S.t_s[o] = lerp_ref(at, S.h_s[hl], S.h_s[hu]);
      S.t_sd[o] = lerp_ref(at, S.h_sd[hl], S.h_sd[hu]);
      S.t_sdd[o] = lerp_ref(at, S.h_sdd[hl], S.h_sdd[hu]);
      for (int d = 0; d < D; d++) {
        const double am = S.amax[(size_t)b * D + d];
        S.t_q[o * D + d] = lerp_ref(at, S.h_q[hl * D + d], S.h_q[hu * D + d]);
        S.t_qd[o * D + d] = lerp_ref(at, S.h_qd[hl * D + d], S.h_qd[hu * D + d]);
        const double a = lerp_ref(at, S.h_qdd[hl * D + d], S.h_qdd[hu * D + d]);
        S.t_qdd[o * D + d] = fmin(fmax(a, -am), am);
      }
    } else {
      offset = min((int)round((start_sec - tm[0]) / S.time_step_sec), tn - 1);
    }
    if (offset > 0) {
      const int drop = min(offset, tn);
      S.t_first[b] = first + drop;
      S.t_count[b] = tn - drop;
    }
    return;
  }
  if (S.initial_plan[b]) {
    // GetTimeOffsetAfter(start) :289-305, then everything from there on is dropped (:604-621)
    if (tn == 0) { S.status[b] = kPlanFailedPrecondition; return; }
    const double *tm = tt + first;
    if (start_sec < tm[0]) { S.status[b] = kPlanOutOfRange; return; }
    int lo = 0, hi = tn;       // upper_bound
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (tm[mid] <= start_sec) lo = mid + 1; else hi = mid;
    }
    if (lo == tn) { S.status[b] = kPlanInternal; return; }
    S.t_count[b] = lo;
  }
  S.mode[b] = kPsetWindows;
  S.finish[b] = 1;
  S.resample_skip[b] = 0;
  if (!planned_to_end) {           // the loop condition of :632 before the first window
    S.active[b] = 1;
    atomicAdd(S.num_active, 1);
  }
}

// "does every looping planner's history have room for one more window?" (count + N <= cap)
static __global__ void k_pset_check_capacity(PlannerSetState S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  if (S.active[b] && S.count[b] + S.N > S.cap) S.num_active[1] = 1;
}

// planners whose window loop ended in an error do not resample (Plan returned before :660)
static __global__ void k_pset_before_resample(PlannerSetState S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  if (S.finish[b] && S.status[b] != kPlanOk) { S.finish[b] = 0; S.resample_skip[b] = 1; }
  if (S.finish[b] && S.count[b] < 2) {
    S.resample_skip[b] = 1;
    if (S.method == 0) { S.status[b] = kPlanInternal; S.finish[b] = 0; }   // "nothing to resample"
    else { S.t_first[b] = 0; S.t_count[b] = 0; S.resample_count[b] = 0; }   // kSkip: cleared, no error
  }
}

// :662-684 after the resample. resample_count[b] = samples the resample produced (it may exceed
// tcap: the host then grows the trajectory arrays and repeats the resample and this kernel).
static __global__ void k_pset_epilogue(PlannerSetState S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  if (!S.finish[b]) return;
  if (!S.resample_skip[b]) {
    const int M = S.resample_count[b];
    if (S.method == 0 && M < 1) { S.status[b] = kPlanInternal; return; }   // negative trajectory duration
    if (M > S.tcap) atomicMax(&S.num_active[1], M);      // does not fit: the host grows the buffers and repeats
    S.t_first[b] = 0;
    S.t_count[b] = min(M, S.tcap);
  }
  S.initial_plan[b] = 1;
  const int tn = S.t_count[b];
  if (tn > 0) {
    const double step = S.time_step_sec;
    long long end = pset_time_from_sec(S.t_time[(size_t)b * S.tcap + S.t_first[b] + tn - 1]);
    end = pset_time_from_sec((double)(long long)round(pset_time_to_sec(end) / step) * step);
    S.end_time_ns[b] = end;
    long long fd = pset_time_from_sec(S.w_time[(size_t)b * S.N + S.w_lei[b]]);
    fd = pset_time_from_sec((double)(long long)round(pset_time_to_sec(fd) / step) * step);
    S.final_decel_start_ns[b] = fd;
  } else {
    S.end_time_ns[b] = S.start_time_ns[b];
    S.final_decel_start_ns[b] = S.end_time_ns[b];
  }
  S.target_reached[b] = S.planned_to_end[b];
}

// rows of `count[b]` valid entries (x width doubles) from a buffer of stride old_cap to one of
// stride new_cap (history / trajectory growth)
static __global__ void k_pset_regrow(int B, int old_cap, int new_cap, int width, const int *first, const int *count,
                                     const double *src, double *dst) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int f = first ? first[b] : 0;
  const int n = (f + count[b]) * width;
  if (i >= n) return;
  dst[(size_t)b * new_cap * width + i] = src[(size_t)b * old_cap * width + i];
}

// summary record per planner (what the mirror's getters need without a trajectory download)
struct PlannerSummaryDev {
  long long end_time_ns, final_decel_start_ns, start_time_ns;
  int num_samples, target_reached, planned_to_end, windows, path_state, history_count, status, pad;
};
static __global__ void k_pset_summary(PlannerSetState S, const int *windows, PlannerSummaryDev *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.B) return;
  PlannerSummaryDev r;
  r.end_time_ns = S.end_time_ns[b]; r.final_decel_start_ns = S.final_decel_start_ns[b];
  r.start_time_ns = S.start_time_ns[b];
  r.num_samples = S.t_count[b]; r.target_reached = S.target_reached[b]; r.planned_to_end = S.planned_to_end[b];
  r.windows = windows[b]; r.path_state = S.path_state[b]; r.history_count = S.count[b]; r.status = S.status[b];
  r.pad = 0;
  out[b] = r;
}

}  // namespace tpamd
