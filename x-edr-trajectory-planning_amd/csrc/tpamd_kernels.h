// tpamd_kernels.h -- HIP kernels of the batched path-timing engine (gfx950).
//
// Pipeline for one batch (all on one stream):
//   k_setup_*          one thread per path: ds, limits, setup validation
//   k_sample_lp_joint  one thread per (path, sample): degree-2 B-spline q,q',q'',
//   / k_lp_rows        constraint rows, LP boundary point, FindSddMax/Min
//   k_boundary_detect  one thread per (path, sample): isolated points and skipped
//                      maxima of the boundary curve (CalculateBoundary pass 2)
//   k_boundary_final   one thread per (path, sample): deferred fixes (pass 3) and
//                      sink/source/trajectory classification (pass 4)
//   k_sweep            one 64-lane wave per path: backward/forward extremals,
//                      switching-point loop, sqrt, time integration
//   k_epilogue         one thread per (path, sample): qd, qdd
#pragma once

#include "tpamd_device.h"

namespace tpamd {

// ------------------------------------------------------------------ workspace
struct Workspace {
  // per path
  double *ds, *s_start, *s_end, *sd_start, *sdd_start, *t_start, *delta;
  uint32_t *err_bits;
  // Ragged batches: number of samples of each path (null: every path has N samples).
  // Arrays keep the common stride N; only the first ns[b] samples of path b are used.
  const int32_t *ns;
  const double *amax;          // [B][D] joint acceleration limits (fused epilogue), or null
  // Ragged batches: path handled by sweep workgroup k, longest path first (k_order_paths), so
  // that the launch's tail is made of short paths; null: workgroup k handles path k.
  const int32_t *order;
  double *lim;  // [B][2][C] lower then upper (joint mode)
  // per (path, sample)
  // joint mode: one record of R = 2D+2 doubles per sample,
  //   [q'_0, q''_0, ..., q'_{D-1}, q''_{D-1}, (pad, pad)]
  // so that the sweep streams one contiguous, 16-byte aligned record per step. The first 2D
  // entries are written by k_sample_lp_joint; in the LDS copy of a record the pad holds the
  // final sd2_max and the type bits (JointSweep::store_tile).
  double *q12;
  double *m0, *z0, *X0, *Y0, *Xz, *Yz;  // pass-1 boundary, [B][N]
  uint8_t *at0;                          // sd2_max_at_sdd0
  uint8_t *fix_flag;
  double *fix_val;
  double *m, *X, *Y;  // final boundary
  uint8_t *type;
  double *sd2;
  long long *diag;  // [B][64] cycle counters; filled only by -DTPAMD_DIAG builds
  double *sd2_out;  // optional caller copy of sd2 ([B][N]); may be null
  int keep_boundary;  // the fused boundary passes also store sdd_max/sdd_min/type (debug copy)
};

__device__ __forceinline__ int path_samples(const Workspace &ws, int b, int N) {
  return ws.ns ? min(ws.ns[b], N) : N;
}
__device__ __forceinline__ int path_of_block(const Workspace &ws, int block) {
  return ws.order ? ws.order[block] : block;
}

// Longest-first processing order of a ragged batch: a counting sort of the paths by sample count
// in bins of kOrderBin samples (the cost of a path's sweep grows with its sample count; inside a
// bin the order does not matter). One block of 1024 threads; order[k] = path of workgroup k.
// The hardware hands workgroups to the CUs in index order as slots free up, so a launch in this
// order is a longest-processing-time-first schedule: the kernel no longer ends with a 4000-sample
// path that started late on an otherwise idle machine.
constexpr int kOrderBins = 1024;
static __global__ void __launch_bounds__(1024) k_order_paths(int B, int N, const int32_t *ns, int32_t *order) {
  __shared__ int hist[kOrderBins];
  __shared__ int scan[kOrderBins];
  const int tid = threadIdx.x;
  const int shift = (N > 8192) ? 5 : 3;          // bins of 8 samples cover N <= 8192
  hist[tid] = 0;
  __syncthreads();
  for (int b = tid; b < B; b += 1024) {
    const int n = min(max(ns[b], 0), N);
    atomicAdd(&hist[min((N - n) >> shift, kOrderBins - 1)], 1);
  }
  __syncthreads();
  // exclusive prefix sum over the bins (Hillis-Steele, 10 rounds)
  scan[tid] = hist[tid];
  __syncthreads();
  for (int off = 1; off < kOrderBins; off <<= 1) {
    const int v = (tid >= off) ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  hist[tid] = scan[tid] - hist[tid];              // first position of the bin
  __syncthreads();
  for (int b = tid; b < B; b += 1024) {
    const int n = min(max(ns[b], 0), N);
    order[atomicAdd(&hist[min((N - n) >> shift, kOrderBins - 1)], 1)] = b;
  }
}

struct JointSource {
  const double *q12;  // [B][N][2D+E+2] records
  const double *lim;  // [B][2][2D+E]
  int D;
  int E = 0;          // extra B-only rows (Cartesian paths: 2)
  __device__ __forceinline__ int rows() const { return 2 * D + E; }
  __device__ __forceinline__ int b_only_from() const { return D; }   // rows D.. have A = 0
  static constexpr bool kJoint = true;
  __device__ __forceinline__ int stride() const { return 2 * D + E + 2; }
  __device__ __forceinline__ JointRowsAt at(int b, int N, int idx) const {
    JointRowsAt r;
    r.q12 = q12 + ((size_t)b * N + idx) * stride();
    r.lim_lo = lim + (size_t)b * 2 * rows();
    r.lim_hi = r.lim_lo + rows();
    r.D = D;
    r.E = E;
    return r;
  }
};

struct GenericSource {
  const double *A, *B, *LO, *HI;  // [B][N][C]
  int C;
  __device__ __forceinline__ int rows() const { return C; }
  __device__ __forceinline__ int b_only_from() const { return -1; }
  static constexpr bool kJoint = false;
  __device__ __forceinline__ GlobalRowsAt at(int b, int N, int idx) const {
    const size_t o = ((size_t)b * N + idx) * C;
    GlobalRowsAt r;
    r.A = A + o; r.B = B + o; r.LO = LO + o; r.HI = HI + o;
    return r;
  }
};

// --------------------------------------------------------------------- setup
// Joint mode: InitSolver/SetupProblem checks that do not need per-sample data
// (time_optimal_path_timing.cc:165-193, :554-576) plus the limit rows of
// timeable_path_joint_spline.cc:327-339. s_end = path_start + delta*(N-1) as
// path_timing_trajectory.cc:340-341.
static __global__ void k_setup_joint(int B, int N, int D, double safety, const double *vmax,
                              const double *amax, const double *path_start,
                              const double *delta, const double *sd_start,
                              const double *sdd_start, const double *t_start, Workspace ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int C = 2 * D;
  double *lo = ws.lim + (size_t)b * 2 * C, *hi = lo + C;
  // every load of the thread is issued before the first use (the kernel sits at the head of the
  // front stage's chain: its latency is part of every pipelined step); the row widths come from
  // the registers, not from what was just stored
  const double s0 = path_start[b], dl = delta[b], sd0 = sd_start[b], t0 = t_start[b];
  const double sdd0 = sdd_start ? sdd_start[b] : 0.0;
  const int Nb = path_samples(ws, b, N);
  const bool too_many = ws.ns && ws.ns[b] > N;
  double maxw = -DBL_MAX;
  bool lower_ge_upper = false;
  for (int d0 = 0; d0 < D; d0 += 8) {
    double am[8], vm[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int d = min(d0 + u, D - 1);
      am[u] = amax[(size_t)b * D + d];
      vm[u] = vmax[(size_t)b * D + d];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int d = d0 + u;
      if (d < D) {
        const double a = am[u] * safety;
        const double na = -am[u] * safety;
        const double v = vm[u] * safety;
        const double vv = v * v;
        hi[d] = a;
        lo[d] = na;
        hi[D + d] = vv;
        lo[D + d] = 0.0;
        const double wa = a - na, wv = vv - 0.0;
        if (wa > maxw) maxw = wa;
        if (wv > maxw) maxw = wv;
        if (na >= a || 0.0 >= vv) lower_ge_upper = true;
      }
    }
  }
  const double s1 = s0 + dl * (Nb - 1);
  uint32_t bits = 0;
  if (maxw <= 0) bits |= kErrInfeasible;
  if (s0 >= s1) bits |= kErrSRange;
  if (sd0 < 0) bits |= kErrSdStartNeg;
  if (lower_ge_upper) bits |= kErrLowerGeUpper;
  if (Nb < 2 || too_many) bits |= kErrTooFew;   // count outside [2, stride]
  ws.err_bits[b] = bits;
  ws.s_start[b] = s0;
  ws.s_end[b] = s1;
  ws.ds[b] = (s1 - s0) / (Nb - 1);
  ws.sd_start[b] = sd0;
  ws.sdd_start[b] = sdd0;
  ws.t_start[b] = t0;
  ws.delta[b] = dl;
}

// One wave that does nothing for `ticks` of the 100 MHz real-time clock (front-stage delay of the
// pipelined mode, see solve_joint).
static __global__ void k_delay(int ticks) {
  unsigned long long t0, t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  do {
    __builtin_amdgcn_s_sleep(8);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  } while ((long long)(t - t0) < ticks);
}

static __global__ void k_setup_rows(int B, int N, const double *s_start, const double *s_end,
                             const double *sd_start, const double *sdd_start,
                             const double *t_start, Workspace ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  uint32_t bits = 0;
  if (s_start[b] >= s_end[b]) bits |= kErrSRange;
  if (sd_start[b] < 0) bits |= kErrSdStartNeg;
  if (N < 2) bits |= kErrTooFew;
  ws.err_bits[b] = bits;  // per-sample bits are OR-ed in by k_lp_rows
  ws.s_start[b] = s_start[b];
  ws.s_end[b] = s_end[b];
  ws.ds[b] = (s_end[b] - s_start[b]) / (N - 1);
  ws.sd_start[b] = sd_start[b];
  ws.sdd_start[b] = sdd_start ? sdd_start[b] : 0.0;
  ws.t_start[b] = t_start[b];
}

// ------------------------------------------------- pass 1, common tail per sample
template <int WORDS, bool JOINT, class R>
__device__ __forceinline__ void boundary_point(const R &r, int C, size_t o, Workspace ws) {
  double sd2max, sddmax, sd2zero;
  lp_find_max_sd2<WORDS>(r, C, &sd2max, &sddmax, &sd2zero);
  double x0, y0;
  if (JOINT) find_sdd_both_joint(r, C / 2, sd2max, &x0, &y0);
  else find_sdd_both(r, C, sd2max, &x0, &y0);
  ws.m0[o] = sd2max;
  ws.z0[o] = sd2zero;
  ws.X0[o] = x0;
  ws.Y0[o] = y0;
  // FindSddMax/Min at sd2_max_for_sdd0 (Xz, Yz) are needed only next to isolated points
  // and deferred fixes (.cc:1386-1395, :1440-1447): k_boundary_zfit / k_boundary_final
  // evaluate them there, as the reference does.
  ws.at0[o] = fabs(sd2max - sd2zero) < kTiny;
}

// ---------------------------------------------------- K1 (joint): sample + LP
// grid = (ceil(N/TPB), B), block = TPB. Dynamic LDS:
//   knots[P+3] | control points [P][D] | lim_lo[2D] | lim_hi[2D] | q'[D][TPB] | q''[D][TPB]
// SamplePath: timeable_path_joint_spline.cc:294-318; EvalCurveAndDerivatives:
// splines/bspline.h:540-568; ConstraintSetup: timeable_path_joint_spline.cc:320-343;
// CalculateBoundary pass 1: time_optimal_path_timing.cc:1366-1377.
// DT > 0: joint count fixed at compile time (q', q'' of the thread's sample also stay in
// registers for FindSddMax/Min); DT = 0: any D.
template <int WORDS, int DT>
__device__ __forceinline__ void sample_lp_joint_body(int N, int D_rt, int P, const double *knots_g,
                                                     const double *cps_g, double *q_out,
                                                     const Workspace &ws) {
  extern __shared__ double lds[];
#ifdef TPAMD_K1_PRIO
  __builtin_amdgcn_s_setprio(TPAMD_K1_PRIO);   // A/B: instruction-issue priority of this kernel's waves
#endif
#ifdef TPAMD_K1_STUDY
  // study build: start / end clock of every block of this kernel (ws.diag[2 * block], [2 * block + 1])
  if (threadIdx.x == 0) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    reinterpret_cast<unsigned long long *>(ws.diag)[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = t | ((unsigned long long)(xcc & 15u) << 60);
  }
#endif
  const int TPB = blockDim.x;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int K = P + 3;
  const int D = DT ? DT : D_rt;
  const int C = 2 * D;
  // a block that lies wholly behind the end of a (ragged) path has nothing to do
  if ((int)(blockIdx.x * TPB) >= path_samples(ws, b, N)) return;
  double q1r[DT ? DT : 1], q2r[DT ? DT : 1];
  double *s_knots = lds;
  double *s_cp = s_knots + K;
  double *s_lo = s_cp + P * D;
  double *s_hi = s_lo + C;
  double *s_Q1 = s_hi + C;
  double *s_Q2 = s_Q1 + (size_t)D * TPB;
  for (int k = tid; k < K; k += TPB) s_knots[k] = knots_g[(size_t)b * K + k];
  for (int k = tid; k < P * D; k += TPB) s_cp[k] = cps_g[(size_t)b * P * D + k];
  for (int k = tid; k < 2 * C; k += TPB) s_lo[k] = ws.lim[(size_t)b * 2 * C + k];
  __syncthreads();
  const int i0 = blockIdx.x * TPB;
  const int i = i0 + tid;
  const int Nb = path_samples(ws, b, N);
  const bool live = i < Nb;
  const size_t o = (size_t)b * N + i;

  const double path_start = ws.s_start[b];
  const double delta = ws.delta[b];
  const double k0 = s_knots[0], kend = s_knots[K - 1];
  const double parameter = path_start + i * delta;
  double *Q1 = s_Q1 + tid, *Q2 = s_Q2 + tid;
  if (!live) {
    // (stays for the block's record store below)
  } else if (parameter < kend + delta) {
    double u = parameter;
    if (u < k0) u = k0;
    if (kend < u) u = kend;
    const int span = knot_span_deg2(s_knots, K, u);
    double ders[3][3];
    basis_ders_deg2(s_knots, span, u, ders);
    const double *p0 = s_cp + (size_t)(span - 2) * D, *p1 = p0 + D, *p2 = p1 + D;
#pragma unroll
    for (int d = 0; d < D; d++) {
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      v0 += ders[0][0] * p0[d]; v0 += ders[0][1] * p1[d]; v0 += ders[0][2] * p2[d];
      v1 += ders[1][0] * p0[d]; v1 += ders[1][1] * p1[d]; v1 += ders[1][2] * p2[d];
      v2 += ders[2][0] * p0[d]; v2 += ders[2][1] * p1[d]; v2 += ders[2][2] * p2[d];
      if (q_out) q_out[o * D + d] = v0;
      Q1[d * TPB] = v1;
      Q2[d * TPB] = v2;
      if (DT) { q1r[d] = v1; q2r[d] = v2; }
    }
  } else {
    const double *pl = s_cp + (size_t)(P - 1) * D;
#pragma unroll
    for (int d = 0; d < D; d++) {
      if (q_out) q_out[o * D + d] = pl[d];
      Q1[d * TPB] = 0.0;
      Q2[d * TPB] = 0.0;
      if (DT) { q1r[d] = 0.0; q2r[d] = 0.0; }
    }
  }
  // The block's records -- [q'_d, q''_d] * D | pad, one contiguous run of (D + 1) 16-byte pairs
  // per sample -- written by consecutive threads from the LDS columns: whole cache lines per
  // store instruction (a thread storing its own record touches 64 lines per instruction, and the
  // partially written lines reach HBM more than once). The pad pair is written too (zeros; the
  // sweep fills it in its LDS copy only) so that the lines are complete.
  __syncthreads();
  {
    const int nvalid = min(TPB, Nb - i0);
    double2 *recs = reinterpret_cast<double2 *>(ws.q12 + ((size_t)b * N + i0) * (C + 2));
    const int pairs = D + 1;
    for (int c = tid; c < nvalid * pairs; c += TPB) {
      const int sm = c / pairs, pr = c - sm * pairs;
      recs[c] = (pr < D) ? make_double2(s_Q1[pr * TPB + sm], s_Q2[pr * TPB + sm]) : make_double2(0.0, 0.0);
    }
  }
  if (!live) return;
  LdsRowsJoint r;
  r.Q1 = Q1; r.Q2 = Q2; r.lim_lo = s_lo; r.lim_hi = s_hi; r.stride = TPB; r.D = D;
  if (DT) {
    double sd2max, sddmax, sd2zero, x0, y0;
    lp_find_max_sd2<WORDS, LdsRowsJoint, DT, 2 * DT>(r, C, &sd2max, &sddmax, &sd2zero, /*b_only_from=*/D,
                                                     q1r, q2r);
    find_sdd_both_joint_screened<(DT ? DT : 1)>(q1r, q2r, Q1, Q2, TPB, s_hi, sd2max, &x0, &y0);
    ws.m0[o] = sd2max;
    ws.z0[o] = sd2zero;
    ws.X0[o] = x0;
    ws.Y0[o] = y0;
    ws.at0[o] = fabs(sd2max - sd2zero) < kTiny;
  } else {
    boundary_point<WORDS, true>(r, C, o, ws);
  }
#ifdef TPAMD_K1_STUDY
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    reinterpret_cast<unsigned long long *>(ws.diag)[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = t;
  }
#endif
}

#ifdef TPAMD_K1_STUDY   // (the stamps must not change the register footprint the study is about)
#define TPAMD_K1_STUDY_REGS __attribute__((amdgpu_waves_per_eu(5, 5)))
#else
#define TPAMD_K1_STUDY_REGS
#endif
template <int WORDS, int DT>
__global__ void TPAMD_K1_STUDY_REGS k_sample_lp_joint(int N, int D_rt, int P, const double *knots_g,
                                  const double *cps_g, double *q_out, Workspace ws) {
  sample_lp_joint_body<WORDS, DT>(N, D_rt, P, knots_g, cps_g, q_out, ws);
}
// The same kernel with the whole register file of two waves per SIMD at its disposal: at D = 14 the
// LDS columns allow two waves per SIMD anyway, and the default budget (128 VGPRs: a kernel without launch bounds
// must fit 1024 threads per block) spills 58 registers. Blocks of at most 256 threads.
template <int WORDS, int DT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_sample_lp_joint_wide(int N, int D_rt, int P, const double *knots_g, const double *cps_g,
                       double *q_out, Workspace ws) {
  sample_lp_joint_body<WORDS, DT>(N, D_rt, P, knots_g, cps_g, q_out, ws);
}

// Sampling only (TimeableJointSplinePath::SamplePath as a stand-alone call,
// timeable_path_joint_spline.cc:294-318): q, q', q'' as [B][N][D] arrays.
// grid = (ceil(N/TPB), B); dynamic LDS: knots[P+3] | control points [P][D].
static __global__ void k_sample_only(int N, int D, int P, const double *knots_g, const double *cps_g,
                              const double *path_start, const double *delta_g, double *q,
                              double *q1, double *q2) {
  extern __shared__ double lds[];
  const int TPB = blockDim.x, tid = threadIdx.x, b = blockIdx.y, K = P + 3;
  double *s_knots = lds, *s_cp = lds + K;
  for (int k = tid; k < K; k += TPB) s_knots[k] = knots_g[(size_t)b * K + k];
  for (int k = tid; k < P * D; k += TPB) s_cp[k] = cps_g[(size_t)b * P * D + k];
  __syncthreads();
  const int i = blockIdx.x * TPB + tid;
  if (i >= N) return;
  const size_t o = ((size_t)b * N + i) * D;
  const double delta = delta_g[b];
  const double k0 = s_knots[0], kend = s_knots[K - 1];
  const double parameter = path_start[b] + i * delta;
  if (parameter < kend + delta) {
    double u = parameter;
    if (u < k0) u = k0;
    if (kend < u) u = kend;
    const int span = knot_span_deg2(s_knots, K, u);
    double ders[3][3];
    basis_ders_deg2(s_knots, span, u, ders);
    const double *p0 = s_cp + (size_t)(span - 2) * D, *p1 = p0 + D, *p2 = p1 + D;
    for (int d = 0; d < D; d++) {
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      v0 += ders[0][0] * p0[d]; v0 += ders[0][1] * p1[d]; v0 += ders[0][2] * p2[d];
      v1 += ders[1][0] * p0[d]; v1 += ders[1][1] * p1[d]; v1 += ders[1][2] * p2[d];
      v2 += ders[2][0] * p0[d]; v2 += ders[2][1] * p1[d]; v2 += ders[2][2] * p2[d];
      q[o + d] = v0; q1[o + d] = v1; q2[o + d] = v2;
    }
  } else {
    const double *pl = s_cp + (size_t)(P - 1) * D;
    for (int d = 0; d < D; d++) { q[o + d] = pl[d]; q1[o + d] = 0.0; q2[o + d] = 0.0; }
  }
}


// ------------------------------------------------------------ pose-spline sampling
// The pose targets TimeableCartesianSplinePath::SamplePath evaluates before it calls the IK
// callback (timeable_path_cartesian_spline.cc:484-503), for B paths at once: the translation
// spline (degree-2 BSpline3d, BSplineT::EvalCurve splines/bspline.h:512-536) and the rotation
// spline (degree-2 BSplineQ::EvalCurve splines/bsplineq.cc:223-244 with the cumulative basis
// :309-317 and QuatPower = exp(p log q) :112-146) on a shared knot vector, at
// parameter = path_start + i * delta; beyond knots.back() - delta the last control pose is
// repeated (:488, :499-502). Quaternions are [w, x, y, z]; poses [B][N][7] = (t | q).
// The operations follow the reference in order; log / atan2 / sin / cos / exp come
// from the device math library, so results agree to rounding, not bit for bit.
struct Quat { double w, x, y, z; };

__device__ __forceinline__ Quat quat_mul(const Quat &a, const Quat &b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__device__ __forceinline__ double quat_sqnorm(const Quat &q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
__device__ __forceinline__ Quat quat_inverse(const Quat &q) {
  const double n2 = quat_sqnorm(q);
  Quat r = {0.0, 0.0, 0.0, 0.0};
  if (n2 > 0.0) { r.w = q.w / n2; r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2; }
  return r;
}
// bsplineq.cc:98-108
__device__ __forceinline__ void quat_normalize_positive_real(Quat &q) {
  if (q.w < 0) { q.w *= -1.0; q.x *= -1.0; q.y *= -1.0; q.z *= -1.0; }
  if (fabs(quat_sqnorm(q) - 1.0) > 1e-12) {
    const double n = sqrt(quat_sqnorm(q));
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
  }
}
// Eigen stableNorm / stableNormalized of the vector part (restated as in the oracle)
__device__ __forceinline__ double vec3_stable_norm(double x, double y, double z) {
  double mx = fabs(x);
  if (fabs(y) > mx) mx = fabs(y);
  if (fabs(z) > mx) mx = fabs(z);
  if (!(mx > 0.0)) return mx;
  const double inv = 1.0 / mx;
  const double a = x * inv, b = y * inv, c = z * inv;
  return mx * sqrt(a * a + b * b + c * c);
}
__device__ __forceinline__ void vec3_stable_normalized(double &x, double &y, double &z) {
  double w = fabs(x);
  if (fabs(y) > w) w = fabs(y);
  if (fabs(z) > w) w = fabs(z);
  const double a = x / w, b = y / w, c = z / w;
  const double zz = a * a + b * b + c * c;
  if (zz > 0.0) {
    const double s = sqrt(zz);
    x = a / s; y = b / s; z = c / s;
  }
}
// bsplineq.cc:136-146 with QuatLog :112-125 and QuatExp :127-134
__device__ __forceinline__ Quat quat_power(Quat q, double power) {
  quat_normalize_positive_real(q);
  Quat l;
  {
    const double nv = vec3_stable_norm(q.x, q.y, q.z);
    l.w = 0.5 * log(quat_sqnorm(q));
    if (nv > 1e-12) {
      double nx = q.x, ny = q.y, nz = q.z;
      vec3_stable_normalized(nx, ny, nz);
      const double ang = atan2(nv, q.w);
      l.x = nx * ang; l.y = ny * ang; l.z = nz * ang;
    } else {
      l.x = q.x; l.y = q.y; l.z = q.z;
    }
  }
  l.w *= power; l.x *= power; l.y *= power; l.z *= power;
  Quat r;
  {
    const double nv = vec3_stable_norm(l.x, l.y, l.z);
    double nx = l.x, ny = l.y, nz = l.z;
    r.w = cos(nv);
    vec3_stable_normalized(nx, ny, nz);
    const double sn = sin(nv);
    r.x = nx * sn; r.y = ny * sn; r.z = nz * sn;
    const double e = exp(l.w);
    r.w *= e; r.x *= e; r.y *= e; r.z *= e;
  }
  return r;
}

// grid = (ceil(N/TPB), B); dynamic LDS: knots[P+3] | translation [P][3] | rotation [P][4]
static __global__ void k_sample_pose_splines(int N, int P, const double *knots_g, const double *trans_g,
                                      const double *rot_g, const double *path_start,
                                      const double *delta_g, double *poses) {
  extern __shared__ double lds[];
  const int TPB = blockDim.x, tid = threadIdx.x, b = blockIdx.y, K = P + 3;
  double *s_knots = lds, *s_t = lds + K, *s_r = s_t + 3 * P;
  for (int k = tid; k < K; k += TPB) s_knots[k] = knots_g[(size_t)b * K + k];
  for (int k = tid; k < 3 * P; k += TPB) s_t[k] = trans_g[(size_t)b * 3 * P + k];
  for (int k = tid; k < 4 * P; k += TPB) s_r[k] = rot_g[(size_t)b * 4 * P + k];
  __syncthreads();
  const int i = blockIdx.x * TPB + tid;
  if (i >= N) return;
  double *out = poses + ((size_t)b * N + i) * 7;
  const double delta = delta_g[b];
  const double kend = s_knots[K - 1];
  const double parameter = path_start[b] + i * delta;
  if (!(parameter < kend - delta) || parameter < s_knots[0]) {
    // past the end: the last control pose (a parameter below the first knot is the caller's
    // error in the reference; it gets the last pose here as well rather than garbage)
    for (int d = 0; d < 3; d++) out[d] = s_t[3 * (P - 1) + d];
    for (int d = 0; d < 4; d++) out[3 + d] = s_r[4 * (P - 1) + d];
    return;
  }
  const int span = knot_span_deg2(s_knots, K, parameter);
  double ders[3][3];
  basis_ders_deg2(s_knots, span, parameter, ders);   // ders[0][*]: the basis of NURBS A2.2
  const double b0 = ders[0][0], b1 = ders[0][1], b2 = ders[0][2];
  const double *t0 = s_t + 3 * (span - 2);
  for (int d = 0; d < 3; d++) {
    double v = 0.0;
    v += b0 * t0[d]; v += b1 * t0[3 + d]; v += b2 * t0[6 + d];
    out[d] = v;
  }
  const double cum1 = b2, cum0 = cum1 + b1;          // bsplineq.cc:313-316
  const double *r0 = s_r + 4 * (span - 2);
  const Quat p0 = {r0[0], r0[1], r0[2], r0[3]}, p1 = {r0[4], r0[5], r0[6], r0[7]},
             p2 = {r0[8], r0[9], r0[10], r0[11]};
  Quat q = p0;
  q = quat_mul(q, quat_power(quat_mul(quat_inverse(p0), p1), cum0));
  q = quat_mul(q, quat_power(quat_mul(quat_inverse(p1), p2), cum1));
  quat_normalize_positive_real(q);
  out[3] = q.w; out[4] = q.x; out[5] = q.y; out[6] = q.z;
}

// ------------------------------------------- Cartesian paths: rows from IK output
// One thread per (path, sample). ComputePathDerivatives
// (timeable_path_cartesian_spline.cc:39-68): forward differences of the IK positions,
// q'[N-1] = 0, q''[0] = q''[N-1] = 0. ConstraintSetup (:551-595): the 2D joint rows plus
// two rows bounding |(J q')_{1..3}|^2 and |(J q')_{4..6}|^2 with lower = -upper. The
// Jacobians [B][N][6][D] are evaluated by the caller's jacobian_func_ (:576); J q' is
// accumulated over the dofs in index order. Rows go to A/Bm/LO/HI [B][N][2D+2]; the
// (q', q'') pairs go to the sample's record for k_epilogue.
static __global__ void k_cartesian_rows(int N, int D, double safety, const double *q_g,
                                 const double *J_g, const double *vmax, const double *amax,
                                 const double *vtrans, const double *vrot, double *A,
                                 double *Bm, double *LO, double *HI, Workspace ws) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int C = 2 * D + 2;
  const size_t o = (size_t)b * N + i;
  const double inv = 1.0 / ws.delta[b];
  const double *q = q_g + o * D;
  const double *J = J_g + o * 6 * D;
  double *rec = ws.q12 + o * (2 * D + 2);
  double *a = A + o * C, *bb = Bm + o * C, *lo = LO + o * C, *hi = HI + o * C;
  double v6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int d = 0; d < D; d++) {
    double q1 = 0.0, q2 = 0.0;
    if (i < N - 1) {
      q1 = inv * (q[D + d] - q[d]);
      if (i >= 1) {
        const double q1n = (i + 1 < N - 1) ? inv * (q[2 * D + d] - q[D + d]) : 0.0;
        q2 = inv * (q1n - q1);
      }
    }
    rec[2 * d] = q1;
    rec[2 * d + 1] = q2;
    const double am = amax[(size_t)b * D + d] * safety;
    const double vm = vmax[(size_t)b * D + d] * safety;
    a[d] = q1;       bb[d] = q2;          hi[d] = am;          lo[d] = -am;
    a[D + d] = 0.0;  bb[D + d] = q1 * q1; hi[D + d] = vm * vm; lo[D + d] = 0.0;
#pragma unroll
    for (int r = 0; r < 6; r++) v6[r] += J[r * D + d] * q1;
  }
  const double vt = vtrans[b], vr = vrot[b];
  a[2 * D] = 0.0;
  bb[2 * D] = (v6[0] * v6[0] + v6[1] * v6[1]) + v6[2] * v6[2];
  hi[2 * D] = vt * vt;
  lo[2 * D] = -(vt * vt);
  a[2 * D + 1] = 0.0;
  bb[2 * D + 1] = (v6[3] * v6[3] + v6[4] * v6[4]) + v6[5] * v6[5];
  hi[2 * D + 1] = vr * vr;
  lo[2 * D + 1] = -(vr * vr);
}

// Setup for Cartesian paths: s_end = path_start + delta (N-1) as
// path_timing_trajectory.cc:340-341; the limit rows of
// timeable_path_cartesian_spline.cc:559-592 (C = 2D+2 per path, constant along the path) and
// the SetupProblem / IsSetupValid checks on them (time_optimal_path_timing.cc:174-175, :557).
static __global__ void k_setup_cartesian(int B, int N, int D, double safety, const double *vmax,
                                  const double *amax, const double *vtrans, const double *vrot,
                                  const double *path_start, const double *delta,
                                  const double *sd_start, const double *sdd_start,
                                  const double *t_start, Workspace ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int C = 2 * D + 2;
  double *lo = ws.lim + (size_t)b * 2 * C, *hi = lo + C;
  for (int d = 0; d < D; d++) {
    const double am = amax[(size_t)b * D + d] * safety;
    const double vm = vmax[(size_t)b * D + d] * safety;
    hi[d] = am;          lo[d] = -am;
    hi[D + d] = vm * vm; lo[D + d] = 0.0;
  }
  const double vt = vtrans[b], vr = vrot[b];
  hi[2 * D] = vt * vt;     lo[2 * D] = -(vt * vt);
  hi[2 * D + 1] = vr * vr; lo[2 * D + 1] = -(vr * vr);
  double maxw = -DBL_MAX;
  bool lower_ge_upper = false;
  for (int c = 0; c < C; c++) {
    const double w = hi[c] - lo[c];
    if (w > maxw) maxw = w;
    if (lo[c] >= hi[c]) lower_ge_upper = true;
  }
  const double s0 = path_start[b];
  const double s1 = s0 + delta[b] * (N - 1);
  uint32_t bits = 0;
  if (maxw <= 0) bits |= kErrInfeasible;
  if (s0 >= s1) bits |= kErrSRange;
  if (sd_start[b] < 0) bits |= kErrSdStartNeg;
  if (lower_ge_upper) bits |= kErrLowerGeUpper;
  if (N < 2) bits |= kErrTooFew;
  ws.err_bits[b] = bits;
  ws.s_start[b] = s0;
  ws.s_end[b] = s1;
  ws.ds[b] = (s1 - s0) / (N - 1);
  ws.sd_start[b] = sd_start[b];
  ws.sdd_start[b] = sdd_start ? sdd_start[b] : 0.0;
  ws.t_start[b] = t_start[b];
  ws.delta[b] = delta[b];
}

// K1 for Cartesian paths with the joint count fixed at compile time: the arithmetic of
// k_cartesian_rows, but the rows stay on chip -- (q', q'') pairs and the two Cartesian B
// values go to the sample's record ([q'_d, q''_d]*D | bt, br | m, type), the LP reads them
// from LDS, FindSddMax/Min from registers. grid = (ceil(N/TPB), B); dynamic LDS:
//   lim_lo[C] | lim_hi[C] | q'[D][TPB] | q''[D][TPB] | extras[2][TPB]
template <int WORDS, int D>
__global__ void k_cartesian_lp(int N, const double *q_g, const double *J_g, Workspace ws) {
  extern __shared__ double lds[];
  constexpr int C = 2 * D + 2;
  const int TPB = blockDim.x;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  double *s_lo = lds, *s_hi = s_lo + C;
  double *s_Q1 = s_hi + C, *s_Q2 = s_Q1 + (size_t)D * TPB, *s_X = s_Q2 + (size_t)D * TPB;
  for (int k = tid; k < 2 * C; k += TPB) s_lo[k] = ws.lim[(size_t)b * 2 * C + k];
  const int i0 = blockIdx.x * TPB;
  const int i = i0 + tid;
  const bool live = i < N;
  const size_t o = (size_t)b * N + (live ? i : N - 1);
  const double inv = 1.0 / ws.delta[b];
  const double *q = q_g + o * D;
  double *Q1 = s_Q1 + tid, *Q2 = s_Q2 + tid, *X = s_X + tid;
  double q1r[D], q2r[D];
#pragma unroll
  for (int d = 0; d < D; d++) {
    double q1 = 0.0, q2 = 0.0;
    if (live && i < N - 1) {
      q1 = inv * (q[D + d] - q[d]);
      if (i >= 1) {
        const double q1n = (i + 1 < N - 1) ? inv * (q[2 * D + d] - q[D + d]) : 0.0;
        q2 = inv * (q1n - q1);
      }
    }
    q1r[d] = q1; q2r[d] = q2;
    Q1[d * TPB] = q1;
  }
  __syncthreads();
  // J q' (timeable_path_cartesian_spline.cc:551-560), one thread per (sample, Cartesian row): the
  // block's Jacobians are one contiguous run of 6 D doubles per sample, so consecutive threads
  // read consecutive rows (a thread per sample would stride 48 D bytes: 64 cache lines per load
  // instruction). Each dot product adds its D terms in the reference's order, starting from 0.
  // The six results of a sample go back to its owner through the LDS that holds q'' afterwards.
  {
    double *V6 = s_Q2;                        // [6][TPB] <= [D][TPB]
    const int nvalid = min(TPB, N - i0);
    const double *Jb = J_g + ((size_t)b * N + i0) * 6 * D;
    for (int p = tid; p < 6 * nvalid; p += TPB) {
      const int sm = p / 6, r = p - 6 * sm;
      const double *Jr = Jb + (size_t)p * D;
      double acc = 0.0;
      if (D % 2 == 0) {
        const double2 *J2 = reinterpret_cast<const double2 *>(Jr);
        double2 jj[D / 2 ? D / 2 : 1];
#pragma unroll
        for (int h = 0; h < D / 2; h++) jj[h] = J2[h];
#pragma unroll
        for (int h = 0; h < D / 2; h++) {
          acc += jj[h].x * s_Q1[(2 * h) * TPB + sm];
          acc += jj[h].y * s_Q1[(2 * h + 1) * TPB + sm];
        }
      } else {
        double jj[D];
#pragma unroll
        for (int d = 0; d < D; d++) jj[d] = Jr[d];
#pragma unroll
        for (int d = 0; d < D; d++) acc += jj[d] * s_Q1[d * TPB + sm];
      }
      V6[r * TPB + sm] = acc;
    }
  }
  __syncthreads();
  double v6[6];
#pragma unroll
  for (int r = 0; r < 6; r++) v6[r] = s_Q2[r * TPB + tid];
  __syncthreads();                            // everyone has its six values: q'' may go there now
#pragma unroll
  for (int d = 0; d < D; d++) Q2[d * TPB] = q2r[d];
  const double bt = (v6[0] * v6[0] + v6[1] * v6[1]) + v6[2] * v6[2];
  const double br = (v6[3] * v6[3] + v6[4] * v6[4]) + v6[5] * v6[5];
  X[0] = bt;
  X[TPB] = br;
  // the block's records ([q'_d, q''_d] * D | bt, br | pad: D + 2 pairs per sample, contiguous)
  // written by consecutive threads from the LDS columns, whole cache lines per instruction
  __syncthreads();
  {
    const int nvalid = min(TPB, N - i0);
    double2 *recs = reinterpret_cast<double2 *>(ws.q12 + ((size_t)b * N + i0) * (C + 2));
    constexpr int pairs = D + 2;
    for (int c = tid; c < nvalid * pairs; c += TPB) {
      const int sm = c / pairs, pr = c - sm * pairs;
      double2 v = make_double2(0.0, 0.0);
      if (pr < D) v = make_double2(s_Q1[pr * TPB + sm], s_Q2[pr * TPB + sm]);
      else if (pr == D) v = make_double2(s_X[sm], s_X[TPB + sm]);
      recs[c] = v;
    }
  }
  if (!live) return;
  LdsRowsJoint r;
  r.Q1 = Q1; r.Q2 = Q2; r.lim_lo = s_lo; r.lim_hi = s_hi; r.stride = TPB; r.D = D;
  r.E = 2; r.X = X;
  double sd2max, sddmax, sd2zero, x0 = 0.0, y0 = 0.0;
  lp_find_max_sd2<WORDS, LdsRowsJoint, D, 2 * D + 2>(r, C, &sd2max, &sddmax, &sd2zero, /*b_only_from=*/D,
                                                     q1r, q2r);
  {
    // the two Cartesian rows have A = 0: like the velocity rows they only gate the result
    const double vt = bt * sd2max, vr = br * sd2max;
    const bool ok = !(vt + kTiny < s_lo[2 * D] || vt - kTiny > s_hi[2 * D]) &&
                    !(vr + kTiny < s_lo[2 * D + 1] || vr - kTiny > s_hi[2 * D + 1]);
    if (ok) find_sdd_both_joint_screened<D>(q1r, q2r, Q1, Q2, TPB, s_hi, sd2max, &x0, &y0);
  }
  ws.m0[o] = sd2max;
  ws.z0[o] = sd2zero;
  ws.X0[o] = x0;
  ws.Y0[o] = y0;
  ws.at0[o] = fabs(sd2max - sd2zero) < kTiny;
}

// ------------------------------------------------ K1 (rows): validation + LP
// grid = (ceil(N/TPB), B). Dynamic LDS: A,B,LO,HI each [C][TPB].
template <int WORDS>
__global__ void k_lp_rows(int N, int C, const double *Ag, const double *Bg, const double *LOg,
                          const double *HIg, Workspace ws) {
  extern __shared__ double lds[];
  const int TPB = blockDim.x;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  double *s_A = lds, *s_B = s_A + (size_t)C * TPB, *s_LO = s_B + (size_t)C * TPB,
         *s_HI = s_LO + (size_t)C * TPB;
  const int i0 = blockIdx.x * TPB;
  const int nvalid = min(TPB, path_samples(ws, b, N) - i0);
  const size_t base = ((size_t)b * N + i0) * C;
  for (int k = tid; k < nvalid * C; k += TPB) {
    const int s = k / C, c = k - s * C;
    s_A[c * TPB + s] = Ag[base + k];
    s_B[c * TPB + s] = Bg[base + k];
    s_LO[c * TPB + s] = LOg[base + k];
    s_HI[c * TPB + s] = HIg[base + k];
  }
  __syncthreads();
  if (tid >= nvalid) return;
  const size_t o = (size_t)b * N + i0 + tid;
  LdsRows r;
  r.A = s_A + tid; r.B = s_B + tid; r.LO = s_LO + tid; r.HI = s_HI + tid;
  r.stride = TPB; r.lim_stride = TPB;
  double maxw = -DBL_MAX;
  bool lge = false;
  for (int c = 0; c < C; c++) {
    const double w = r.hi(c) - r.lo(c);
    if (w > maxw) maxw = w;
    if (r.lo(c) >= r.hi(c)) lge = true;
  }
  uint32_t bits = 0;
  if (maxw <= 0) bits |= kErrInfeasible;
  if (lge) bits |= kErrLowerGeUpper;
  if (bits) atomicOr(&ws.err_bits[b], bits);
  boundary_point<WORDS, false>(r, C, o, ws);
}

// Stand-alone batched LP (tpamd_find_max_sd2_host): one thread per LP.
template <int WORDS>
__global__ void k_lp_only(int num, int C, const double *Ag, const double *Bg,
                          const double *LOg, const double *HIg, double *sd2max,
                          double *sddmax, double *sd2zero) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  GlobalRowsAt r;
  r.A = Ag + (size_t)i * C; r.B = Bg + (size_t)i * C;
  r.LO = LOg + (size_t)i * C; r.HI = HIg + (size_t)i * C;
  lp_find_max_sd2<WORDS>(r, C, &sd2max[i], &sddmax[i], &sd2zero[i]);
}

// ------------------------------------------- K1b: CalculateBoundary pass 2
// time_optimal_path_timing.cc:1381-1431. The reference walks i = 1..N-2 and
// (a) rewrites the neighbours of isolated points in place, (b) tests sample i
// against the state left by iterations <= i. Both are functions of pass-1 data
// only, so every i is evaluated independently here:
//   iso(i)   = !at[i-1] && at[i] && !at[i+1]
//   element j as seen by iteration k: rewritten as "left of iso(j+1)" if
//   j+1 <= k, else as "right of iso(j-1)" if j-1 <= k, else untouched.
//   A rewritten element has sd2_max = sd2_max_for_sdd0 and sdd_max = FindSddMax
//   there; its sdd_min is FindSddMin there for a left rewrite and FindSddMax
//   there for a right rewrite (the reference's line :1394-1395).
__device__ __forceinline__ bool iso_at(const uint8_t *at, int N, int i) {
  return (i >= 1) && (i <= N - 2) && !at[i - 1] && at[i] && !at[i + 1];
}

static __global__ void k_boundary_detect(int stride, Workspace ws) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = path_samples(ws, b, stride);
  if (k >= N) return;
  const size_t pb = (size_t)b * stride;
  if (k < 1 || k > N - 2) {
    ws.fix_flag[pb + k] = 0;
    return;
  }
  const uint8_t *at = ws.at0 + pb;
  const double *m0 = ws.m0 + pb, *z0 = ws.z0 + pb, *X0 = ws.X0 + pb, *Y0 = ws.Y0 + pb,
               *Xz = ws.Xz + pb, *Yz = ws.Yz + pb;
  const double ds = ws.ds[b];
  const bool iso_k = iso_at(at, N, k), iso_km1 = iso_at(at, N, k - 1),
             iso_km2 = iso_at(at, N, k - 2);
  // element k-1
  const bool l_mod = iso_k || iso_km2;
  const double m_l = l_mod ? z0[k - 1] : m0[k - 1];
  const double fsmax_l = l_mod ? Xz[k - 1] : X0[k - 1];  // FindSddMax(k-1, m_l)
  // element k
  const double m_c = iso_km1 ? z0[k] : m0[k];
  const double X_c = iso_km1 ? Xz[k] : X0[k];
  const double Y_c = iso_km1 ? Xz[k] : Y0[k];
  // element k+1
  const double m_r = iso_k ? z0[k + 1] : m0[k + 1];
  const double Y_r = iso_k ? Xz[k + 1] : Y0[k + 1];
  const double fsmin_r = iso_k ? Yz[k + 1] : Y0[k + 1];  // FindSddMin(k+1, m_r)

  const double sd2p = (m_r - m_c) / ds;
  const double sd2p_min = 2 * Y_c;
  const double sd2p_max = 2 * X_c;
  const bool sink_or_source = (sd2p < sd2p_min) || (sd2p > sd2p_max);
  const bool skipped_sdd = (X_c > 0) && (Y_r < 0);
  const bool skipped_sd2 = (m_c > m_l - kTiny) && (m_c > m_r - kTiny);
  uint8_t flag = 0;
  double val = 0.0;
  if ((skipped_sd2 || skipped_sdd) && sink_or_source) {
    const double fw = m_l + 2.0 * ds * fsmax_l;  // OneForwardExtremalStep(k-1, m_l), .cc:753-759
    const double bw = m_r - 2.0 * ds * fsmin_r;  // OneBackwardExtremalStep(k+1, m_r), .cc:761-767
    double mn = z0[k];
    if (fw < mn) mn = fw;
    if (bw < mn) mn = bw;
    val = (0.0 < mn) ? mn : 0.0;
    flag = 1;
  }
  ws.fix_flag[pb + k] = flag;
  ws.fix_val[pb + k] = val;
}

// ---------------------------------- K1c: CalculateBoundary passes 3 and 4
// time_optimal_path_timing.cc:1432-1484. The deferred list is applied in
// ascending index order, each entry k writing k, then k-1, then k+1; the last
// writer of element j is therefore entry j+1 (sdd0 values), else entry j (its
// value), else entry j-1 (sdd0 values), else the pass-2 state.
__device__ __forceinline__ double final_m(const Workspace &ws, size_t pb, int N, int j) {
  const uint8_t *ff = ws.fix_flag + pb;
  if (j + 1 <= N - 2 && ff[j + 1]) return ws.z0[pb + j];
  if (ff[j]) return ws.fix_val[pb + j];
  if (j >= 1 && ff[j - 1]) return ws.z0[pb + j];
  const uint8_t *at = ws.at0 + pb;
  if (iso_at(at, N, j + 1) || iso_at(at, N, j - 1)) return ws.z0[pb + j];
  return ws.m0[pb + j];
}

// FindSddMax and FindSddMin of one sample by a whole wave: candidates over lanes, every
// lane validates its candidate against all rows, wave reductions keep the extremes (the
// value find_sdd_both returns: its visiting order and pruning do not change the result;
// a NaN candidate is never selected there because it fails both comparisons).
// b_only_from >= 0: rows b_only_from .. C-1 have A = 0 (velocity / Cartesian rows of a
// joint-structured sample): they yield no candidates and their validity test does not
// involve the candidate, so it is made once for the whole sample (as find_sdd_both_joint).
template <class R>
__device__ __forceinline__ void wave_find_sdd_both(const R &r, int C, double sd2, int lane,
                                                   double *sdd_max, double *sdd_min,
                                                   int b_only_from = -1) {
  // lane k < C keeps row k in registers (one parallel load phase); the validity loop then
  // reads row k of every lane's candidate through readlane instead of re-loading it
  const bool has = lane < C;
  const double a_m = has ? r.a(lane) : 0.0, b_m = has ? r.b(lane) : 0.0;
  const double lo_m = has ? r.lo(lane) : 0.0, hi_m = has ? r.hi(lane) : 0.0;
  const int Cc = (b_only_from >= 0) ? b_only_from : C;      // rows that depend on the candidate
  bool fixed_bad = false;
  if (b_only_from >= 0 && has && lane >= b_only_from) {
    const double v = b_m * sd2;
    fixed_bad = (v + kTiny < lo_m) || (v - kTiny > hi_m);
  }
  const bool none = __any(fixed_bad);
  double smax = -DBL_MAX, smin = DBL_MAX;
  for (int c0 = 0; c0 < 2 * Cc && !none; c0 += 64) {
    const int c = c0 + lane;
    const int i = min(c >> 1, Cc - 1);
    const double A = __shfl(a_m, i, 64);
    const double bs = __shfl(b_m, i, 64) * sd2;
    // both shuffles outside any lane-dependent branch: a shuffle reads only active lanes
    const double hi_i = __shfl(hi_m, i, 64), lo_i = __shfl(lo_m, i, 64);
    const double lim = (c & 1) ? hi_i : lo_i;
    const double sddi = (lim - bs) / A;
    bool bad = (c >= 2 * Cc) || is_tiny(A) || !(fabs(sddi) <= DBL_MAX);
    for (int k = 0; k < Cc; k++) {
      const double v = wave_bcast_const(a_m, k) * sddi + wave_bcast_const(b_m, k) * sd2;
      const double lo_k = wave_bcast_const(lo_m, k), hi_k = wave_bcast_const(hi_m, k);
      const bool under = v + kTiny < lo_k, over = v - kTiny > hi_k;
      bad = bad || under || over;
    }
    if (!bad) {
      if (sddi > smax) smax = sddi;
      if (sddi < smin) smin = sddi;
    }
  }
  smax = wave_max_f64(smax);
  smin = wave_min_f64(smin);
  if (smax == -DBL_MAX) smax = 0;
  if (smin == DBL_MAX) smin = 0;
  *sdd_max = smax;
  *sdd_min = smin;
}

// Same for small constraint sets, four samples per wave: each 16-lane group owns one sample
// (rows in lanes 0..C-1 of the group, C <= 16, and its <= 16 candidates one per lane); rows are
// broadcast inside the group with shuffles. `valid` is false for a group without a sample.
// Row registers of group16_find_sdd_both: lane gl of a 16-lane group holds row gl of the
// group's sample (zeros beyond C / for a group without a sample).
struct Group16Rows {
  double a, b, lo, hi;
};
template <class R>
__device__ __forceinline__ Group16Rows group16_load_rows(const R &r, bool valid, int C, int lane) {
  const int gl = lane & 15;
  const bool has = valid && gl < C;
  Group16Rows g;
  g.a = has ? r.a(gl) : 0.0;
  g.b = has ? r.b(gl) : 0.0;
  g.lo = has ? r.lo(gl) : 0.0;
  g.hi = has ? r.hi(gl) : 0.0;
  return g;
}
__device__ __forceinline__ void group16_find_sdd_both_rows(const Group16Rows &g, bool valid, int C,
                                                           double sd2, int lane, double *sdd_max,
                                                           double *sdd_min, int b_only_from) {
  const int gl = lane & 15, gb = lane & 48;
  const bool has = valid && gl < C;
  const double a_m = g.a, b_m = g.b, lo_m = g.lo, hi_m = g.hi;
  const int Cc = (b_only_from >= 0) ? b_only_from : C;
  bool fixed_bad = false;
  if (b_only_from >= 0 && has && gl >= b_only_from) {
    const double v = b_m * sd2;
    fixed_bad = (v + kTiny < lo_m) || (v - kTiny > hi_m);
  }
  const unsigned long long fb = __ballot(fixed_bad);
  const bool none = ((fb >> gb) & 0xFFFFull) != 0ull;
  const int c = gl;
  const int i = min(c >> 1, Cc - 1);
  const double A = __shfl(a_m, gb + i, 64);
  const double bs = __shfl(b_m, gb + i, 64) * sd2;
  const double hi_i = __shfl(hi_m, gb + i, 64), lo_i = __shfl(lo_m, gb + i, 64);
  const double lim = (c & 1) ? hi_i : lo_i;
  const double sddi = (lim - bs) / A;
  bool bad = none || !valid || (c >= 2 * Cc) || is_tiny(A) || !(fabs(sddi) <= DBL_MAX);
  for (int k = 0; k < Cc; k++) {
    const double a_k = __shfl(a_m, gb + k, 64), b_k = __shfl(b_m, gb + k, 64);
    const double lo_k = __shfl(lo_m, gb + k, 64), hi_k = __shfl(hi_m, gb + k, 64);
    const double v = a_k * sddi + b_k * sd2;
    const bool under = v + kTiny < lo_k, over = v - kTiny > hi_k;
    bad = bad || under || over;
  }
  double smax = bad ? -DBL_MAX : sddi, smin = bad ? DBL_MAX : sddi;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    const double o1 = __shfl_xor(smax, off, 64), o2 = __shfl_xor(smin, off, 64);
    smax = (o1 > smax) ? o1 : smax;
    smin = (o2 < smin) ? o2 : smin;
  }
  if (smax == -DBL_MAX) smax = 0;
  if (smin == DBL_MAX) smin = 0;
  *sdd_max = smax;
  *sdd_min = smin;
}
template <class R>
__device__ __forceinline__ void group16_find_sdd_both(const R &r, bool valid, int C, double sd2,
                                                      int lane, double *sdd_max, double *sdd_min,
                                                      int b_only_from) {
  const Group16Rows g = group16_load_rows(r, valid, C, lane);
  group16_find_sdd_both_rows(g, valid, C, sd2, lane, sdd_max, sdd_min, b_only_from);
}

// Whether group16_find_sdd_both applies to a source.
template <class Source>
__device__ __forceinline__ bool fits_group16(const Source &src) {
  const int C = src.rows();
  const int Cc = (src.b_only_from() >= 0) ? src.b_only_from() : C;
  return C <= 16 && 2 * Cc <= 16;
}

// CalculateBoundary pass 2, first half (.cc:1386-1395): the two neighbours of an isolated
// point take sd2_max_for_sdd0 as their boundary value and FindSddMax/Min there. Those are
// the only places (besides the deferred fixes, see k_boundary_final) where Xz/Yz are read,
// so they are evaluated here, one wave per such sample, instead of for every sample in K1.
template <class Source>
__global__ void __launch_bounds__(256) k_boundary_zfit(int stride, Source src, Workspace ws) {
  __shared__ int s_count;
  __shared__ int s_list[256];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int j = blockIdx.x * 256 + tid;
  const int N = path_samples(ws, b, stride);
  const size_t pb = (size_t)b * stride;
  const uint8_t *at = ws.at0 + pb;
  if (tid == 0) s_count = 0;
  __syncthreads();
  if (j < N && (iso_at(at, N, j - 1) || iso_at(at, N, j + 1))) s_list[atomicAdd(&s_count, 1)] = tid;
  __syncthreads();
  const int count = s_count;
  const int lane = tid & 63;
  if (fits_group16(src)) {
    for (int it0 = (tid >> 6) * 4; it0 < count; it0 += 16) {     // four samples per wave
      const int it = it0 + (lane >> 4);
      const bool valid = it < count;
      const int jj = blockIdx.x * 256 + s_list[valid ? it : 0];
      const auto r = src.at(b, stride, jj);
      double x, y;
      group16_find_sdd_both(r, valid, src.rows(), ws.z0[pb + jj], lane, &x, &y, src.b_only_from());
      if (valid && (lane & 15) == 0) { ws.Xz[pb + jj] = x; ws.Yz[pb + jj] = y; }
    }
    return;
  }
  for (int it = tid >> 6; it < count; it += 4) {
    const int jj = blockIdx.x * 256 + s_list[it];
    const auto r = src.at(b, stride, jj);
    double x, y;
    wave_find_sdd_both(r, src.rows(), ws.z0[pb + jj], lane, &x, &y, src.b_only_from());
    if (lane == 0) { ws.Xz[pb + jj] = x; ws.Yz[pb + jj] = y; }
  }
}

// Block = 256 threads, one per sample. The few samples whose boundary value was replaced
// in pass 3 (by a new value, or by sd2_max_for_sdd0 next to a fix) need FindSddMax/Min
// there; they are collected in LDS and each is
// evaluated by a whole wave (the per-thread version made every wave holding one such
// sample run the full candidate loop).
template <class Source>
__global__ void __launch_bounds__(256) k_boundary_final(int stride, Source src, Workspace ws) {
  __shared__ int s_count;
  __shared__ int s_list[256];
  __shared__ double s_X[256], s_Y[256], s_at[256];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int j = blockIdx.x * 256 + tid;
  const int N = path_samples(ws, b, stride);
  const bool active = j < N;
  const size_t pb = (size_t)b * stride;
  const uint8_t *ff = ws.fix_flag + pb;
  const uint8_t *at = ws.at0 + pb;
  double m = 0.0, X = 0.0, Y = 0.0;
  bool refit = false;
  if (tid == 0) s_count = 0;
  __syncthreads();
  if (active) {
    const bool f_next = (j + 1 <= N - 2) && ff[j + 1];
    const bool f_self = ff[j];
    const bool f_prev = (j >= 1) && ff[j - 1];
    if (f_next || (!f_self && f_prev)) {
      m = ws.z0[pb + j];
      if (iso_at(at, N, j - 1) || iso_at(at, N, j + 1)) {   // k_boundary_zfit was here
        X = ws.Xz[pb + j]; Y = ws.Yz[pb + j];
      } else {
        refit = true;
      }
    } else if (f_self) {
      m = ws.fix_val[pb + j];
      refit = true;
    } else if (iso_at(at, N, j + 1)) {
      m = ws.z0[pb + j]; X = ws.Xz[pb + j]; Y = ws.Yz[pb + j];
    } else if (iso_at(at, N, j - 1)) {
      m = ws.z0[pb + j]; X = ws.Xz[pb + j]; Y = ws.Xz[pb + j];  // sic, .cc:1394-1395
    } else {
      m = ws.m0[pb + j]; X = ws.X0[pb + j]; Y = ws.Y0[pb + j];
    }
    if (refit) {
      s_list[atomicAdd(&s_count, 1)] = tid;
      s_at[tid] = m;
    }
  }
  __syncthreads();
  {
    const int count = s_count;
    const int lane = tid & 63;
    if (fits_group16(src)) {
      for (int it0 = (tid >> 6) * 4; it0 < count; it0 += 16) {   // four samples per wave
        const int it = it0 + (lane >> 4);
        const bool valid = it < count;
        const int t = s_list[valid ? it : 0];
        const auto r = src.at(b, stride, blockIdx.x * 256 + t);
        double x, y;
        group16_find_sdd_both(r, valid, src.rows(), s_at[t], lane, &x, &y, src.b_only_from());
        if (valid && (lane & 15) == 0) { s_X[t] = x; s_Y[t] = y; }
      }
    } else {
      for (int it = tid >> 6; it < count; it += 4) {
        const int t = s_list[it];
        const int jj = blockIdx.x * 256 + t;
        const auto r = src.at(b, stride, jj);
        double x, y;
        wave_find_sdd_both(r, src.rows(), s_at[t], lane, &x, &y, src.b_only_from());
        if (lane == 0) { s_X[t] = x; s_Y[t] = y; }
      }
    }
  }
  __syncthreads();
  if (!active) return;
  if (refit) { X = s_X[tid]; Y = s_Y[tid]; }
  ws.m[pb + j] = m;
  ws.X[pb + j] = X;
  ws.Y[pb + j] = Y;
  uint8_t type = kBndNone;
  if (j >= 1 && j <= N - 2) {
    const double m_next = final_m(ws, pb, N, j + 1);
    const double ds = ws.ds[b];
    const double sd2p = (m_next - m) / ds;
    const double sd2p_min = 2 * Y;
    const double sd2p_max = 2 * X;
    if (sd2p < sd2p_min) type = kBndSink;
    else if (sd2p > sd2p_max) type = kBndSource;
    if ((sd2p <= sd2p_max) && (sd2p >= sd2p_min)) type = kBndTrajectory;
  }
  // bit 3 is not part of the reference's classification: it caches the comparison
  // NextCriticalPoint makes against sd2_max_for_sdd0[0] (.cc:710) for the sweep kernel.
  if (m == ws.z0[pb]) type |= kBndEqualsZ00;
  ws.type[pb + j] = type;
}

// ------------------------------------------------------------- K2: the sweep
// One 64-lane wave per path. All solver scalars are wave-uniform (every lane
// computes the same values); lanes split the candidate/row work of
// FindSddMax/FindSddMin/AreDerivativesValid and the index ranges of the
// search/fill/scan loops. sd2_ and sdd_ live in LDS.
template <class Source>
struct Sweep {
  Source src;
  int b, N, stride, C, lane;   // N samples of this path; arrays use the batch stride
  double ds;
  double *sd2, *sdd;   // LDS [N]
  const double *m;     // final sd2_max [N]
  const uint8_t *type; // final classification [N]

  __device__ __forceinline__ void put_sd2(int i, double v) {
    if (lane == 0) sd2[i] = v;
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void put_sdd(int i, double v) {
    if (lane == 0) sdd[i] = v;
    __builtin_amdgcn_wave_barrier();
  }

  // FindSddMax (want_max) / FindSddMin, time_optimal_path_timing.cc:638-695:
  // candidates are spread over lanes, each lane validates its candidate against
  // all rows, a wave reduction keeps the extreme valid candidate.
  __device__ double find_sdd(int idx, double s2, bool want_max) const {
    const auto r = src.at(b, stride, idx);
    double best = want_max ? -DBL_MAX : DBL_MAX;
    for (int c = lane; c < 2 * C; c += 64) {
      const int i = c >> 1;
      const double A = r.a(i);
      if (!is_tiny(A)) {
        const double lim = (c & 1) ? r.hi(i) : r.lo(i);
        const double sddi = (lim - r.b(i) * s2) / A;
        const bool better = want_max ? (sddi > best) : (sddi < best);
        if (better && rows_valid(r, C, sddi, s2)) best = sddi;
      }
    }
    if (want_max) {
      best = wave_max_f64(best);
      if (best == -DBL_MAX) best = 0;
    } else {
      best = wave_min_f64(best);
      if (best == DBL_MAX) best = 0;
    }
    return best;
  }

  // AreDerivativesValid (.cc:624-636) with rows spread over lanes.
  __device__ bool derivs_valid(int idx, double sddv, double s2) const {
    const auto r = src.at(b, stride, idx);
    bool bad = false;
    for (int i = lane; i < C; i += 64) {
      const double v = r.a(i) * sddv + r.b(i) * s2;
      if (v + kTiny < r.lo(i) || v - kTiny > r.hi(i)) bad = true;
    }
    return !__any(bad);
  }

  // ComputeSddAtIntersection, .cc:722-751
  __device__ void sdd_at_intersection(int index) {
    double cand[3];
    int n = 0;
    if (index > 0 && index < N - 1) cand[n++] = 0.25 / ds * (sd2[index + 1] - sd2[index - 1]);
    if (index < N - 1) cand[n++] = 0.5 / ds * (sd2[index + 1] - sd2[index]);
    if (index > 0) cand[n++] = 0.5 / ds * (sd2[index] - sd2[index - 1]);
    double res = 0.0;
    const double s2 = sd2[index];
    for (int k = 0; k < n; k++) {
      if (derivs_valid(index, cand[k], s2)) { res = cand[k]; break; }
    }
    put_sdd(index, res);
  }

  // AddForwardExtremal, .cc:769-857
  __device__ int add_forward(int idx_lo) {
    const double two_ds = 2.0 * ds;
    for (int idx = idx_lo; idx < N - 2; idx++) {
      const double cur = sd2[idx];
      const double m_i = m[idx], m_n = m[idx + 1];
      const uint8_t t_i = type[idx], t_n = type[idx + 1];
      const bool on_boundary = is_tiny(cur - m_i);
      double sd2tmp, sddtmp;
      if (on_boundary && (t_i & kBndTrajectory) && (t_n & kBndTrajectory)) {
        sd2tmp = m_n;
        sddtmp = 0.5 * (sd2tmp - cur) / ds;
      } else {
        sddtmp = find_sdd(idx, cur, true);
        sd2tmp = cur + two_ds * sddtmp;
      }
      const double nxt = sd2[idx + 1];
      if (!isnan(nxt) && (nxt < sd2tmp)) {
        sdd_at_intersection(idx);
        return N - 1;
      }
      if (sd2tmp > m_n) {
        const double sdd_bound = 0.5 * (m_n - cur) / ds;
        const bool deriv_invalid = !derivs_valid(idx, sdd_bound, m_i);
        const bool type_invalid = t_n & kBndSink;
        if (type_invalid || deriv_invalid) return idx;
        sd2tmp = m_n;
        sddtmp = sdd_bound;
      }
      if (sd2tmp < 0) {
        sd2tmp = 0.0;
        if (idx <= 1) sddtmp = 0.0; else sddtmp = -sd2[idx - 1] / ds;
      }
      put_sd2(idx + 1, sd2tmp);
      put_sdd(idx, sddtmp);
    }
    return N - 1;
  }

  // AddBackwardExtremal, .cc:859-952
  __device__ int add_backward(int idx_hi) {
    const double two_ds = 2.0 * ds;
    for (int idx = idx_hi; idx > 1; idx--) {
      const double cur = sd2[idx];
      const double m_i = m[idx], m_p = m[idx - 1];
      const uint8_t t_i = type[idx], t_p = type[idx - 1];
      const bool on_boundary = is_tiny(cur - m_i);
      double sd2tmp, sddtmp;
      if (on_boundary && (t_i & kBndTrajectory) && (t_p & kBndTrajectory)) {
        sd2tmp = m_p;
        sddtmp = 0.5 * (cur - sd2tmp) / ds;
      } else {
        sddtmp = find_sdd(idx, cur, false);
        sd2tmp = cur - two_ds * sddtmp;
      }
      const double prv = sd2[idx - 1];
      if (!isnan(prv) && (prv < sd2tmp)) {
        sdd_at_intersection(idx);
        return 0;
      }
      if (sd2tmp > m_p) {
        const double sdd_bound = 0.5 * (cur - m_p) / ds;
        const bool deriv_invalid = !derivs_valid(idx, sdd_bound, cur);
        const bool type_invalid = t_p & kBndSource;
        const bool is_connecting = (idx_hi != (N - 1));
        if ((type_invalid || deriv_invalid) && !is_connecting) return idx;
        sd2tmp = m_p;
        sddtmp = sdd_bound;
      }
      if (sd2tmp < 0) {
        sd2tmp = 0.0;
        if (idx < N - 1) sddtmp = sd2[idx + 1] / ds; else sddtmp = 0.0;
      }
      put_sd2(idx - 1, sd2tmp);
      put_sdd(idx, sddtmp);
    }
    return 0;
  }

  // NextCriticalPoint, .cc:697-720, as two wave-parallel scans:
  //  1. first idx in (lo, hi] classified source or trajectory -> c0 (none: -1);
  //  2. first idx >= c0 whose sd2 is already set -> e (none: -1); the answer is
  //     the last idx in (c0, e] with sd2_max[idx] == sd2_max_for_sdd0[0]
  //     (sic, index 0, .cc:710), else c0.
  __device__ int next_critical_point(int idx_lo, int idx_hi, double z00) const {
    int c0 = -1;
    for (int base = idx_lo + 1; base <= idx_hi && c0 < 0; base += 64) {
      const int idx = base + lane;
      const bool hit = (idx <= idx_hi) && (type[idx] & (kBndSource | kBndTrajectory));
      const unsigned long long mask = __ballot(hit);
      if (mask) c0 = base + __ffsll((long long)mask) - 1;
    }
    if (c0 < 0) return -1;
    int crit = c0;
    for (int base = c0; base <= idx_hi; base += 64) {
      const int idx = base + lane;
      const bool in = idx <= idx_hi;
      const bool set = in && !isnan(sd2[idx]);
      const bool isol = in && (idx > c0) && (m[idx] == z00);
      const unsigned long long mset = __ballot(set);
      unsigned long long miso = __ballot(isol);
      if (mset) {
        const int e = __ffsll((long long)mset) - 1;  // lane of the first set sample
        if (e < 63) miso &= (2ull << e) - 1ull;
        if (miso) crit = base + 63 - __clzll((long long)miso);
        return crit;
      }
      if (miso) crit = base + 63 - __clzll((long long)miso);
    }
    return -1;
  }
};

// Ordered prefix sum over the 64 lanes: returns carry + d_0 + d_1 + ... + d_lane, added
// strictly left to right (the reference's time_[i] = time_[i-1] + dt, .cc:453-454). Lane L
// takes lane L-1's running sum through a wave_shr:1 DPP move and adds its own increment;
// after step j lanes 0..j hold their final value.
__device__ __forceinline__ double wave_ordered_prefix(double d, double carry) {
  double t = 0.0;
  const int clo = __double2loint(carry), chi = __double2hiint(carry);
#pragma unroll
  for (int j = 0; j < 64; j++) {
    const int lo = __builtin_amdgcn_update_dpp(clo, __double2loint(t), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(chi, __double2hiint(t), 0x138, 0xf, 0xf, false);
    t = __hiloint2double(hi, lo) + d;
  }
  return t;
}

// Common tail of the sweep kernels (time_optimal_path_timing.cc:398-477): NaN
// check, sdd fill-in at extremal intersections, start acceleration, sqrt,
// last_extremal_index_, time integration, outputs. `status` is the outcome of the
// switching-point loop (0, 7 or 10). sd2 is an LDS array [N]; sdd is either an LDS
// array (copy_sdd: written out at the end) or the output row itself.
// The tail is run by ONE wave (the whole block in the generic kernel, wave 0 in the two-wave
// kernel): stores of some lanes must be visible to loads of others, nothing more.
__device__ __forceinline__ void tail_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Returns the path's final status.
template <class Source>
__device__ __forceinline__ int sweep_tail(const Source &src, const Workspace &ws, int b, int N, int stride,
                           int lane, int status, double *sd2, double *sdd,
                           bool copy_sdd, double *t_out, double *s_out, double *sd_out,
                           double *sdd_out, int32_t *lei_out, double *dtmax_out,
                           int32_t *status_out) {
  const size_t pb = (size_t)b * stride;
  const double ds = ws.ds[b];
  const double *m = ws.m + pb;
  const int C = src.rows();
  // NaN check and sdd fill-in (.cc:398-411); every index is independent.
  if (status == 0) {
    bool has_nan = false;
    for (int idx = lane; idx < N; idx += 64)
      if (isnan(sd2[idx])) has_nan = true;
    if (__any(has_nan)) status = 8;
  }
  if (status == 0) {
    // four independent loads per lane in flight (sdd is a global array in the joint kernels)
    for (int base = 0; base < N; base += 256) {
      double cur[4];
#pragma unroll
      for (int u = 0; u < 4; u++) cur[u] = sdd[min(base + 64 * u + lane, N - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = base + 64 * u + lane;
        if (idx < N && isnan(cur[u])) {
          // ComputeSddAtIntersection (.cc:722-751) for this sample alone
          const auto r = src.at(b, stride, idx);
          const double s2 = sd2[idx];
          const bool has_next = idx < N - 1, has_prev = idx > 0;
          double res = 0.0;
          bool done = false;
          if (has_next && has_prev) {
            const double c = 0.25 / ds * (sd2[idx + 1] - sd2[idx - 1]);
            if (rows_valid(r, C, c, s2)) { res = c; done = true; }
          }
          if (!done && has_next) {
            const double c = 0.5 / ds * (sd2[idx + 1] - s2);
            if (rows_valid(r, C, c, s2)) { res = c; done = true; }
          }
          if (!done && has_prev) {
            const double c = 0.5 / ds * (s2 - sd2[idx - 1]);
            if (rows_valid(r, C, c, s2)) { res = c; done = true; }
          }
          sdd[idx] = res;
        }
      }
    }
    tail_sync();
    // Enforce the start acceleration if admissible (.cc:413-416): rows over lanes.
    {
      const double sdd_start = ws.sdd_start[b];
      const auto r0 = src.at(b, stride, 0);
      const double s20 = sd2[0];
      bool bad = false;
      for (int i = lane; i < C; i += 64) {
        const double v = r0.a(i) * sdd_start + r0.b(i) * s20;
        if (v + kTiny < r0.lo(i) || v - kTiny > r0.hi(i)) bad = true;
      }
      if (!__any(bad) && lane == 0) sdd[0] = sdd_start;
    }
    tail_sync();
    if (sd2[N - 1] != 0) status = 9;
  }
  if (status != 0) {
    if (lane == 0) {
      status_out[b] = status;
      if (lei_out) lei_out[b] = 0;
      if (dtmax_out) dtmax_out[b] = -1.0;
    }
    return status;
  }

  // last_extremal_index_ (.cc:430-445): scan down from N-2 for sdd > 0 or a
  // sample on the boundary curve.
  int lei = 0;
  {
    const int start = (1 > N - 2) ? 1 : N - 2;
    bool found = false;
    for (int top = start; top >= 1 && !found; top -= 256) {
      double a[4], mm[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {          // unconditional, independent loads
        const int ic = max(top - 64 * u - lane, 0);
        a[u] = sdd[ic];
        mm[u] = m[ic];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = top - 64 * u - lane;
        const bool hit = (idx >= 1) & ((a[u] > 0.0) | (fabs(sd2[max(idx, 0)] - mm[u]) < kTiny));
        const unsigned long long mask = __ballot(hit);
        if (mask && !found) { lei = top - 64 * u - (__ffsll((long long)mask) - 1); found = true; }
      }
    }
  }

  // sd = sqrt(sd2) (.cc:420) and the time integral (.cc:447-467): dt of 64 samples in
  // parallel, then the ordered prefix sum.
  const double t0 = ws.t_start[b];
  const double s0 = ws.s_start[b], s1 = ws.s_end[b];
  double tprev = t0;  // time_[base-1]
  double dtmax = 0.0;
  for (int base = 0; base < N; base += 64) {
    const int idx = base + lane;
    double dt = 0.0;
    bool zero_pair = false;
    double sdv = 0.0, s2 = 0.0;
    if (idx < N) {
      s2 = sd2[idx];
      sdv = sqrt(s2);
      if (idx >= 1) {
        const double s2p = sd2[idx - 1];
        if ((s2p > 0) || (s2 > 0)) dt = 2.0 * ds / (sqrt(s2p) + sdv);
        else zero_pair = true;
      }
    }
    const double t = wave_ordered_prefix(dt, tprev);
    tprev = __shfl(t, 63, 64);
    if (idx < N) {
      t_out[pb + idx] = t;
      sd_out[pb + idx] = sdv;
      s_out[pb + idx] = (idx == N - 1) ? s1 : ds * idx + s0;
      ws.sd2[pb + idx] = s2;
      if (ws.sd2_out) ws.sd2_out[pb + idx] = s2;
      if (dt > dtmax) dtmax = dt;
    }
    // zero acceleration across stationary pairs (.cc:463-465); both stores write 0
    if (zero_pair) { sdd[idx - 1] = 0; sdd[idx] = 0; }
  }
  dtmax = wave_max_f64(dtmax);
  if (copy_sdd) {
    tail_sync();
    for (int idx = lane; idx < N; idx += 64) sdd_out[pb + idx] = sdd[idx];
  }
  if (lane == 0) {
    status_out[b] = 0;
    if (lei_out) lei_out[b] = lei;
    if (dtmax_out) dtmax_out[b] = dtmax;
  }
  return 0;
}

// Dynamic LDS: sd2[N] | sdd[N] | dt[64]
template <class Source>
__global__ void __launch_bounds__(64)
k_sweep(int stride, int max_loops, Source src, Workspace ws, double *t_out, double *s_out,
        double *sd_out, double *sdd_out, int32_t *lei_out, double *dtmax_out,
        int32_t *status_out) {
  extern __shared__ double lds[];
  const int b = path_of_block(ws, blockIdx.x);
  const int lane = threadIdx.x;
  const int N = path_samples(ws, b, stride);
  const size_t pb = (size_t)b * stride;
  const uint32_t bits = ws.err_bits[b];
  if (bits & kErrSkip) return;          // not part of this solve: outputs stay as they are
  if (bits) {
    if (lane == 0) {
      status_out[b] = status_from_bits(bits);
      if (lei_out) lei_out[b] = 0;
      if (dtmax_out) dtmax_out[b] = -1.0;
    }
    return;
  }
  Sweep<Source> S;
  S.src = src; S.b = b; S.N = N; S.stride = stride; S.C = src.rows(); S.lane = lane;
  S.ds = ws.ds[b];
  S.sd2 = lds; S.sdd = lds + N;
  S.m = ws.m + pb; S.type = ws.type + pb;
  double *sd2 = S.sd2, *sdd = S.sdd;
  const double sd_start = ws.sd_start[b];

  for (int i = lane; i < N; i += 64) { sd2[i] = qnan(); sdd[i] = qnan(); }
  __syncthreads();
  if (lane == 0) { sd2[0] = sd_start * sd_start; sd2[N - 1] = 0; }
  __syncthreads();

  int status = 0;
  int iforw_lo = 0, iback_hi = N - 1, iback_lo, iforw_hi, icrit, icrit_lo, icrit_hi;
  iback_lo = S.add_backward(iback_hi);
  iforw_hi = S.add_forward(iforw_lo);
  icrit_hi = iback_lo;
  if ((iforw_hi < icrit_hi) && ((icrit_hi < N - 2) && (icrit_hi >= 2))) {
    S.put_sd2(icrit_hi, qnan());
    icrit_hi++;
    iback_lo++;
  }
  icrit_lo = iforw_hi;
  const double z00 = ws.z0[pb];
  if (max_loops <= 0) max_loops = max(100, 10 * N);   // path_timing_trajectory.cc:398-400
  for (int loop = 0; loop < max_loops; loop++) {
    if (iforw_hi >= icrit_hi) break;
    icrit = S.next_critical_point(icrit_lo, icrit_hi, z00);
    if (icrit < 0 || icrit >= N) icrit = (int)(0.5 * (icrit_lo + icrit_hi));
    if (icrit > 0 && icrit < N - 1) S.put_sd2(icrit, S.m[icrit]);
    if (icrit < 1) { status = 10; break; }
    if (S.m[icrit - 1] <= S.m[icrit]) {
      iback_hi = icrit - 1;
      S.put_sd2(icrit - 1, S.m[icrit - 1]);
    } else {
      iback_hi = icrit;
    }
    iback_lo = S.add_backward(iback_hi);
    iforw_lo = icrit;
    iforw_hi = S.add_forward(iforw_lo);
    if (iback_lo > icrit_lo) { status = 7; break; }
    icrit_lo = iforw_hi;
  }
  __syncthreads();
  sweep_tail(src, ws, b, N, stride, lane, status, lds, lds + N, /*copy_sdd=*/true, t_out, s_out,
             sd_out, sdd_out, lei_out, dtmax_out, status_out);
}

// -------------------------------------------------------------- K3: epilogue
// path_timing_trajectory.cc:458-472. One thread per (path, sample, joint): the reads of
// the (q', q'') pairs and the writes of qd/qdd are contiguous across a wave.
static __global__ void k_epilogue(int B, int N, int D, const double *rec, const double *sd,
                           const double *sdd, const double *amax, const int32_t *status,
                           const int32_t *ns, double *qd, double *qdd) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * D;
  if (e >= total) return;
  const size_t o = e / D;           // (path, sample)
  const int d = (int)(e - o * D);
  const int b = (int)(o / N);
  if (status[b] != 0) return;
  if (ns && (int)(o - (size_t)b * N) >= ns[b]) return;
  const double v = sd[o], a = sdd[o];
  const double v2 = v * v;
  const double2 pr = *reinterpret_cast<const double2 *>(rec + o * (2 * D + 2) + 2 * d);
  const double am = amax[(size_t)b * D + d];
  if (qd) qd[e] = pr.x * v;
  if (qdd) {
    double acc = pr.x * a + pr.y * v2;
    if (acc < -am) acc = -am;
    if (acc > am) acc = am;
    qdd[e] = acc;
  }
}


// -------------------------------------------------------- receding-horizon window chaining
// PathTimingTrajectory::Plan's window loop (path_timing_trajectory.cc:628-660 around
// ComputeTimingProfile :307-475) for B planners with joint-space spline paths, chained on the
// device: every iteration is k_plan_begin -> set-up -> K1 -> k_plan_project -> sweep ->
// k_plan_end -> k_plan_append, and only the number of planners still looping travels to the host.
// Planner state lives in the PlanParams arrays (a window history of `cap` samples per planner:
// the reference's *_at_path_samples_ vectors). Times are int64 nanoseconds; TimeFromSec truncates
// seconds * 1e9, TimeToSec divides by 1e9 (trajectory_planning/time.h:22-29).
struct PlanParams {
  int B, N, D, K, cap, max_iterations;
  double max_initial_velocity_error;
  const double *knots;             // [B][K]
  const double *delta;             // [B]
  const double *initial_velocity;  // [B][D]
  const long long *start_ns, *horizon_ns;   // [B] arguments of Plan
  // planner state (in/out)
  int *path_state;                 // [B] 1 new, 2 modified, 3 sampled (timeable_path.h:94-103)
  int *count;                      // [B] samples in the window history
  double *h_time, *h_s, *h_sd, *h_sdd;      // [B][cap]
  double *h_q, *h_qd, *h_qdd;               // [B][cap][D]
  int *planned_to_end;             // [B]
  double *path_horizon;            // [B]
  long long *final_decel_start_ns; // [B]
  // loop state
  int *active, *old_state, *offset, *loop, *append, *windows, *status;
  long long *loop_start_ns;
  int *num_active;                 // [1]
  // scalars of the window being solved (inputs of set-up / K1 / sweep)
  double *path_start, *sd_start, *time_start;   // [B]
  // outputs of the window just solved
  const double *w_time, *w_s, *w_sd, *w_sdd, *w_q, *w_qd, *w_qdd;   // [B][N](x D)
  const int *w_status, *w_lei;
};
enum { kPlanOk = 0, kPlanFailedPrecondition = 1, kPlanOutOfRange = 2, kPlanInvalidArgument = 3,
       kPlanInternal = 4, kPlanDeadlineExceeded = 5 };

// where the next window starts (path_timing_trajectory.cc:318-341)
static __global__ void k_plan_begin(PlanParams p, Workspace ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  if (!p.active[b]) return;
  const long long duration = p.start_ns[b] + p.horizon_ns[b] - p.loop_start_ns[b];
  if (duration <= 0) { p.status[b] = kPlanInvalidArgument; p.active[b] = 0; return; }   // :313-317
  const double start_sec = (double)p.loop_start_ns[b] / 1e9;
  const int old_state = p.path_state[b];
  p.old_state[b] = old_state;
  int offset = 0;
  if (old_state == 1) {
    p.path_start[b] = 0.0;
    p.sd_start[b] = 0.0;
    p.time_start[b] = start_sec;
  } else {
    const int num = p.count[b];
    if (num == 0) { p.status[b] = kPlanFailedPrecondition; p.active[b] = 0; return; }
    const double *ht = p.h_time + (size_t)b * p.cap;
    int lo = 0, hi = num;                       // lower_bound(time_at_path_samples_, start_sec)
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (ht[mid] < start_sec) lo = mid + 1; else hi = mid;
    }
    offset = min(max(lo - 1, 0), num - 1);
    p.path_start[b] = p.h_s[(size_t)b * p.cap + offset];
    p.sd_start[b] = p.h_sd[(size_t)b * p.cap + offset];
    p.time_start[b] = ht[offset];
  }
  p.offset[b] = offset;
  p.path_horizon[b] = p.path_start[b] + p.delta[b] * (p.N - 1);
  p.path_state[b] = 3;                          // SamplePath: kPathWasSampled
}

// skip marks for the planners that are not looping (after set-up wrote the error bits)
static __global__ void k_plan_mark_skipped(PlanParams p, Workspace ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  if (!p.active[b]) ws.err_bits[b] |= kErrSkip;
}

// least-squares start velocity along the start tangent (path_timing_trajectory.cc:360-393),
// after K1 has sampled the window: q'(0) is the first record's first components
static __global__ void k_plan_project(PlanParams p, Workspace ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  if (!p.active[b]) return;
  const int old_state = p.old_state[b];
  if (old_state != 1 && old_state != 2) return;
  const int D = p.D;
  const double *rec0 = ws.q12 + (size_t)b * p.N * (2 * D + 2);
  const double *iv = p.initial_velocity + (size_t)b * D;
  double nrm2 = 0.0;
  for (int d = 0; d < D; d++) nrm2 += rec0[2 * d] * rec0[2 * d];
  double v = p.sd_start[b];
  if (nrm2 > 100 * DBL_EPSILON) {
    double dot = 0.0;
    for (int d = 0; d < D; d++) dot += iv[d] * rec0[2 * d];
    const double q = dot / nrm2;
    v = (q > 0.0) ? q : 0.0;
  }
  double max_err = 0.0;
  for (int d = 0; d < D; d++) {
    const double e = fabs(rec0[2 * d] * v - iv[d]);
    if (e > max_err) max_err = e;
  }
  if (max_err > p.max_initial_velocity_error) {
    p.status[b] = kPlanInvalidArgument;
    p.active[b] = 0;
    ws.err_bits[b] |= kErrSkip;
    return;
  }
  p.sd_start[b] = v;
  ws.sd_start[b] = v;
}

// loop bookkeeping after a window (path_timing_trajectory.cc:639-660)
static __global__ void k_plan_end(PlanParams p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  p.append[b] = 0;
  if (!p.active[b]) return;
  const int st = p.w_status[b];
  if (st != 0) { p.status[b] = kPlanInternal; p.active[b] = 0; return; }   // :394-417
  p.append[b] = 1;
  p.windows[b] += 1;
  const int N = p.N;
  const double *t = p.w_time + (size_t)b * N;
  const int lei = p.w_lei[b];
  const int decel_start = max(lei, N / 2);
  p.final_decel_start_ns[b] = (long long)(t[decel_start] * 1e9);
  const double kend = p.knots[(size_t)b * p.K + p.K - 1];
  const int planned_to_end = p.path_horizon[b] >= kend - 1e-4;     // CloseToEnd, kSmall
  p.planned_to_end[b] = planned_to_end;
  const bool reached = (t[N - 1] - (double)p.start_ns[b] / 1e9) > (double)p.horizon_ns[b] / 1e9;
  if (p.loop[b] >= p.max_iterations) { p.status[b] = kPlanDeadlineExceeded; p.active[b] = 0; return; }
  p.loop_start_ns[b] = p.final_decel_start_ns[b];
  p.loop[b] += 1;
  if (planned_to_end || reached) p.active[b] = 0;
  else atomicAdd(p.num_active, 1);
}

// drop what the window replaces and append it (path_timing_trajectory.cc:418-472)
static __global__ void k_plan_append(PlanParams p) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.N || !p.append[b]) return;
  const int N = p.N, D = p.D;
  const size_t src = (size_t)b * N + i;
  const size_t dst = (size_t)b * p.cap + p.offset[b] + i;
  if (p.offset[b] + N > p.cap) return;          // the host checks the capacity beforehand
  p.h_time[dst] = p.w_time[src];
  p.h_s[dst] = p.w_s[src];
  p.h_sd[dst] = p.w_sd[src];
  p.h_sdd[dst] = p.w_sdd[src];
  for (int d = 0; d < D; d++) {
    p.h_q[dst * D + d] = p.w_q[src * D + d];
    p.h_qd[dst * D + d] = p.w_qd[src * D + d];
    p.h_qdd[dst * D + d] = p.w_qdd[src * D + d];
  }
  if (i == 0) p.count[b] = p.offset[b] + N;
}

// (q', q'') pairs of the records -> separate [B][N][D] arrays (GetFirst/SecondPathDerivativeAt)
static __global__ void k_unpack_records(int B, int N, int D, const double *rec, double *q1, double *q2) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)B * N * D) return;
  const size_t o = e / D;
  const int d = (int)(e - o * D);
  const double *r = rec + o * (2 * D + 2) + 2 * d;
  q1[e] = r[0];
  q2[e] = r[1];
}

// ------------------------------------------------------------------ s(t) query
// GetPathParameterAndDerivatives, time_optimal_path_timing.cc:1549-1627, with
// SampleIndexFromTime :1497-1524 (the bracket time[k] <= t < time[k+1] is unique
// for a non-decreasing time array, so a plain binary search finds the same k).
// sd2_g: the squared velocities sd2_ of the SAME solve that produced time/s/sd. ds_, s_start
// and s_end are recovered from the s row: s[0] = ds*0 + s_start and s[N-1] = s_end exactly
// (.cc:540-547), so (s[N-1] - s[0]) / (N-1) repeats the operation that formed ds_ (.cc:540).
static __global__ void k_query(int B, int N, int K, const double *time, const double *s,
                        const double *sd, const double *sd2_g, const int32_t *status,
                        const double *tq, double *os, double *osd, double *osdd,
                        int32_t *ok) {
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (size_t)B * K) return;
  const int b = (int)(o / K);
  if (status && status[b] != 0) {
    if (ok) ok[o] = 0;
    return;
  }
  const double *tm = time + (size_t)b * N, *sp = s + (size_t)b * N, *sdp = sd + (size_t)b * N,
               *sd2 = sd2_g + (size_t)b * N;
  const double s_start = sp[0], s_end = sp[N - 1];
  const double ds_ = (s_end - s_start) / (N - 1);
  const double inv_ds = 1.0 / ds_;
  const double t = tq[o];
  int good = 1;
  double rs, rsd, rsdd;
  if (t <= tm[0]) {
    rs = s_start; rsd = sdp[0]; rsdd = 0.5 * inv_ds * (sd2[1] - sd2[0]);
  } else if (t >= tm[N - 1]) {
    rs = s_end; rsd = sdp[N - 1]; rsdd = 0.0;
  } else {
    int lo = 0, hi = N - 1;  // invariant: tm[lo] <= t < tm[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tm[mid] <= t) lo = mid; else hi = mid;
    }
    const int k = lo;
    const double dt = t - tm[k];
    const double sda = sdp[k], sdb = sdp[k + 1];
    const double sd2a = sd2[k], sd2b = sd2[k + 1];
    if (sda > 0 || sdb > 0) {
      double dsv = sda * dt + dt * dt * 0.25 * inv_ds * (sd2b - sd2a);
      if (dsv > ds_) dsv = ds_;
      if (dt < 0 || dsv < 0) good = 0;
      const double cand = sp[k] + dsv;
      rs = (sp[k + 1] < cand) ? sp[k + 1] : cand;
      rsd = sqrt(sd2a + dsv * inv_ds * (sd2b - sd2a));
      rsdd = 0.5 * inv_ds * (sd2b - sd2a);
    } else {
      rs = sp[k] + (sp[k + 1] - sp[k]) * dt / (tm[k + 1] - tm[k]);
      rsd = 0.0; rsdd = 0.0;
    }
  }
  os[o] = rs; osd[o] = rsd; osdd[o] = rsdd;
  if (ok) ok[o] = good;
}

// ------------------------------------------------------- uniform-time resample
// ResampleEquidistantlyInTime (path_timing_trajectory.cc:755-783) with
// InterpolateAtTime (:709-753). The reference advances lower_index monotonically
// (TimeAtPathSamplesLowerIndex :686-695: first index >= previous with
// time[index+1] > t, else N-1); because the query times increase, that equals the
// first index overall with time[index+1] > t, found here by binary search.
// eigenmath::InterpolateLinear is restated as a + t*(b-a) (not in the reference
// tree; ulp-level parity of the resampled values is unpinned).
struct ResampleParams {
  int B, N, D, max_out;
  const double *time, *s, *sd, *sdd, *q, *qd, *qdd, *amax, *start_sec;
  double time_step;
  const int32_t *status;
  const int32_t *ns = nullptr;   // samples of each path (planner histories); null: N. N stays the stride.
  double *ot, *os, *osd, *osdd, *oq, *oqd, *oqdd;
  int32_t *count;
};

__device__ __forceinline__ double lerp_ref(double t, double a, double b) { return a + t * (b - a); }

static __global__ void k_resample(ResampleParams p) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = p.N, D = p.D;
  const int N = p.ns ? p.ns[b] : p.N;
  if ((p.status && p.status[b] != 0) || N < 2) {
    if (i == 0) p.count[b] = 0;
    return;
  }
  const double *tm = p.time + (size_t)b * stride;
  const double start = p.start_sec[b];
  const double duration = tm[N - 1] - start;
  const int M = (int)(ceil(duration / p.time_step) + 1);
  if (i == 0) p.count[b] = M;
  if (i >= M || i >= p.max_out) return;
  const double t = start + p.time_step * i;
  // first index in [0, N-2] with tm[index+1] > t, else N-1
  int lower;
  if (!(tm[N - 1] > t)) {
    lower = N - 1;
  } else {
    int lo = 1, hi = N - 1;  // first j in [1, N-1] with tm[j] > t (exists: tm[N-1] > t)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (tm[mid] > t) hi = mid; else lo = mid + 1;
    }
    lower = lo - 1;
  }
  const int upper = (N - 1 < lower + 1) ? N - 1 : lower + 1;
  const double at = (fabs(tm[upper] - tm[lower]) < DBL_EPSILON)
                        ? 0.5
                        : (t - tm[lower]) / (tm[upper] - tm[lower]);
  const size_t ob = (size_t)b * p.max_out + i;
  const size_t kl = (size_t)b * stride + lower, ku = (size_t)b * stride + upper;
  p.ot[ob] = t;
  p.os[ob] = lerp_ref(at, p.s[kl], p.s[ku]);
  p.osd[ob] = lerp_ref(at, p.sd[kl], p.sd[ku]);
  p.osdd[ob] = lerp_ref(at, p.sdd[kl], p.sdd[ku]);
  const bool last = (i == M - 1);
  for (int d = 0; d < D; d++) {
    const double am = p.amax[(size_t)b * D + d];
    double vq, vqd, vqdd;
    if (last) {
      vq = p.q[((size_t)b * stride + (N - 1)) * D + d];
      vqd = 0.0; vqdd = 0.0;
    } else {
      vq = lerp_ref(at, p.q[kl * D + d], p.q[ku * D + d]);
      vqd = lerp_ref(at, p.qd[kl * D + d], p.qd[ku * D + d]);
      vqdd = lerp_ref(at, p.qdd[kl * D + d], p.qdd[ku * D + d]);
      if (vqdd < -am) vqdd = -am;
      if (vqdd > am) vqdd = am;
    }
    p.oq[ob * D + d] = vq;
    p.oqd[ob * D + d] = vqd;
    p.oqdd[ob * D + d] = vqdd;
  }
}

// ResampleSkippingSamplesCloserThanTimeStep (path_timing_trajectory.cc:785-836), one
// 64-lane wave per path. The "keep" decision is a sequential recurrence on the last kept
// time (lane 0 walks the samples and records the kept indices in LDS); the first,
// interpolated output and the copies of the kept samples are done by all lanes.
// p.time_step carries the minimum time delta to keep (0.95 time step, :893-900).
// Dynamic LDS: int[N].
static __global__ void __launch_bounds__(64) k_resample_skip(ResampleParams p) {
  extern __shared__ int kept[];
  __shared__ int s_count, s_lower;
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int stride = p.N, D = p.D;
  const int N = p.ns ? p.ns[b] : p.N;
  if ((p.status && p.status[b] != 0) || N < 2) {
    if (lane == 0) p.count[b] = 0;
    return;
  }
  const double *tm = p.time + (size_t)b * stride;
  const double start = p.start_sec[b];
  if (lane == 0) {
    // TimeAtPathSamplesLowerIndex (:686-695): first i in [0, N-2] with tm[i+1] > start, else N-1
    int lower = N - 1;
    for (int i = 0; i < N - 1; i++)
      if (tm[i + 1] > start) { lower = i; break; }
    s_lower = lower;
    int n = 0;
    double last = start;
    for (int i = lower + 1; i < N; i++) {
      const double t = tm[i];
      if (fabs(t - last) < p.time_step) continue;
      last = t;
      kept[n++] = i;
    }
    s_count = n + 1;
    p.count[b] = n + 1;
  }
  __syncthreads();
  const int M = s_count, lower = s_lower;
  const int upper = (N - 1 < lower + 1) ? N - 1 : lower + 1;
  const double at = (fabs(tm[upper] - tm[lower]) < DBL_EPSILON) ? 0.5 : (start - tm[lower]) / (tm[upper] - tm[lower]);
  const size_t pb = (size_t)b * stride;
  for (int k = lane; k < M && k < p.max_out; k += 64) {
    const size_t ob = (size_t)b * p.max_out + k;
    const bool last_out = (k == M - 1);
    if (k == 0) {
      p.ot[ob] = start;
      p.os[ob] = lerp_ref(at, p.s[pb + lower], p.s[pb + upper]);
      p.osd[ob] = lerp_ref(at, p.sd[pb + lower], p.sd[pb + upper]);
      p.osdd[ob] = lerp_ref(at, p.sdd[pb + lower], p.sdd[pb + upper]);
    } else {
      const int i = kept[k - 1];
      p.ot[ob] = tm[i];
      p.os[ob] = p.s[pb + i];
      p.osd[ob] = p.sd[pb + i];
      p.osdd[ob] = p.sdd[pb + i];
    }
    for (int d = 0; d < D; d++) {
      double vq, vqd, vqdd;
      if (last_out) {
        vq = p.q[(pb + (N - 1)) * D + d];
        vqd = 0.0; vqdd = 0.0;
      } else if (k == 0) {
        const double am = p.amax[(size_t)b * D + d];
        vq = lerp_ref(at, p.q[(pb + lower) * D + d], p.q[(pb + upper) * D + d]);
        vqd = lerp_ref(at, p.qd[(pb + lower) * D + d], p.qd[(pb + upper) * D + d]);
        vqdd = lerp_ref(at, p.qdd[(pb + lower) * D + d], p.qdd[(pb + upper) * D + d]);
        if (vqdd < -am) vqdd = -am;
        if (vqdd > am) vqdd = am;
      } else {
        const int i = kept[k - 1];
        vq = p.q[(pb + i) * D + d];
        vqd = p.qd[(pb + i) * D + d];
        vqdd = p.qdd[(pb + i) * D + d];
      }
      p.oq[ob * D + d] = vq;
      p.oqd[ob * D + d] = vqd;
      p.oqdd[ob * D + d] = vqdd;
    }
  }
}

}  // namespace tpamd
