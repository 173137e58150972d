// tpamd_device.h -- device-side building blocks (gfx950, fp64, no FMA contraction).
//
// Everything here keeps the reference's floating-point operation order, because
// comparisons against kTiny decide control flow in the solver
// (trajectory_planning/time_optimal_path_timing.cc:778,:799,:805). Build with
// -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace tpamd {

// time_optimal_path_timing.h:275-279
constexpr double kTiny = 2.220446049250313e-16 * 1e5;
constexpr double kMaxSd2 = 1e6;

// Boundary classification, time_optimal_path_timing.h:226-231
enum : uint8_t { kBndNone = 0, kBndSource = 1, kBndSink = 2, kBndTrajectory = 4 };
// engine-internal cache bit in the type byte (see k_boundary_final); masked out of every
// comparison the reference makes and of the debug copy
enum : uint8_t { kBndEqualsZ00 = 8, kBndTypeMask = 7 };

// setup error bits (resolved to TPAMD_PATH_* in the reference's order of checks)
enum : uint32_t {
  kErrInfeasible = 1u,    // .cc:174-182
  kErrSRange = 2u,        // .cc:185
  kErrSdStartNeg = 4u,    // .cc:190
  kErrLowerGeUpper = 8u,  // .cc:557
  kErrTooFew = 16u,       // .cc:568
  kErrSkip = 32u          // engine-internal: the path takes no part in this solve (window chaining)
};

__device__ __forceinline__ bool is_tiny(double v) { return fabs(v) < kTiny; }
__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

__device__ __forceinline__ int status_from_bits(uint32_t bits) {
  if (bits & kErrInfeasible) return 2;
  if (bits & kErrSRange) return 3;
  if (bits & kErrSdStartNeg) return 4;
  if (bits & kErrLowerGeUpper) return 5;
  if (bits & kErrTooFew) return 6;
  return 0;
}

// ---------------------------------------------------------------------------
// Row accessors. A "row set" is the C constraint rows of ONE path sample:
//   lower(c) <= a(c)*sdd + b(c)*sd2 <= upper(c)
// (TimeOptimalPathProfile::Constraint, time_optimal_path_timing.h:65-102).
// ---------------------------------------------------------------------------

// Rows of this thread's sample staged in LDS, column-strided ([c][thread]).
struct LdsRows {
  const double *A, *B, *LO, *HI;  // already offset by the thread index
  int stride;                     // threads per block
  int lim_stride;                 // stride of LO/HI: `stride` (generic) or 0.. see below
  __device__ __forceinline__ double a(int c) const { return A[c * stride]; }
  __device__ __forceinline__ double b(int c) const { return B[c * stride]; }
  __device__ __forceinline__ double lo(int c) const { return LO[c * lim_stride]; }
  __device__ __forceinline__ double hi(int c) const { return HI[c * lim_stride]; }
};

// Joint-space rows of this thread's sample with only q' and q'' staged in LDS
// ([d][thread]); the velocity rows (A = 0, B = q'^2) are formed on access and the
// per-path limits are shared by the block.
struct LdsRowsJoint {
  const double *Q1, *Q2;          // already offset by the thread index
  const double *lim_lo, *lim_hi;  // [2D + E]
  int stride, D;
  int E = 0;                      // extra rows with A = 0 and an explicit B (Cartesian paths)
  const double *X = nullptr;      // their B values, [E][stride], offset by the thread index
  __device__ __forceinline__ double a(int c) const { return c < D ? Q1[c * stride] : 0.0; }
  __device__ __forceinline__ double b(int c) const {
    if (c < D) return Q2[c * stride];
    if (c >= 2 * D) return X[(c - 2 * D) * stride];
    const double v = Q1[(c - D) * stride];
    return v * v;
  }
  __device__ __forceinline__ double lo(int c) const { return lim_lo[c]; }
  __device__ __forceinline__ double hi(int c) const { return lim_hi[c]; }
};

// Rows of a joint-space sample computed on the fly from q' and q'' in global
// memory (timeable_path_joint_spline.cc:320-343): rows 0..D-1 are the
// acceleration rows (A = q', B = q''), rows D..2D-1 the velocity rows
// (A = 0, B = q'^2). Cartesian paths (timeable_path_cartesian_spline.cc:578-592) append
// E = 2 rows with A = 0 and B = |(J q')_{1..3}|^2, |(J q')_{4..6}|^2, stored after the
// pairs. lim_lo / lim_hi are the per-path limits [2D + E].
struct JointRowsAt {
  const double *q12;  // &record[sample][0]: pairs (q'[d], q''[d]), d = 0..D-1, then E extras
  const double *lim_lo, *lim_hi;
  int D;
  int E = 0;
  __device__ __forceinline__ double a(int c) const { return c < D ? q12[2 * c] : 0.0; }
  __device__ __forceinline__ double b(int c) const {
    if (c < D) return q12[2 * c + 1];
    if (c >= 2 * D) return q12[c];      // extras sit at doubles 2D .. 2D+E-1 of the record
    const double v = q12[2 * (c - D)];
    return v * v;
  }
  __device__ __forceinline__ double lo(int c) const { return lim_lo[c]; }
  __device__ __forceinline__ double hi(int c) const { return lim_hi[c]; }
};

// Rows of a sample stored explicitly in global memory ([C] each).
struct GlobalRowsAt {
  const double *A, *B, *LO, *HI;
  __device__ __forceinline__ double a(int c) const { return A[c]; }
  __device__ __forceinline__ double b(int c) const { return B[c]; }
  __device__ __forceinline__ double lo(int c) const { return LO[c]; }
  __device__ __forceinline__ double hi(int c) const { return HI[c]; }
};

// AreDerivativesValid, time_optimal_path_timing.cc:624-636 (one thread, all rows)
template <class R>
__device__ __forceinline__ bool rows_valid(const R &r, int C, double sdd, double sd2) {
  for (int i = 0; i < C; i++) {
    const double v = r.a(i) * sdd + r.b(i) * sd2;
    if (v + kTiny < r.lo(i) || v - kTiny > r.hi(i)) return false;
  }
  return true;
}

// FindSddMax and FindSddMin in one pass (time_optimal_path_timing.cc:638-695).
// The reference keeps "the largest (smallest) candidate that is valid"; the
// order candidates are visited in does not change that value.
template <class R>
__device__ void find_sdd_both(const R &r, int C, double sd2, double *sdd_max, double *sdd_min) {
  double smax = -DBL_MAX, smin = DBL_MAX;
  for (int i = 0; i < C; i++) {
    const double A = r.a(i);
    if (!is_tiny(A)) {
      const double bs = r.b(i) * sd2;
      for (int w = 0; w < 2; w++) {
        const double lim = w ? r.hi(i) : r.lo(i);
        const double sddi = (lim - bs) / A;
        if (((sddi > smax) || (sddi < smin)) && rows_valid(r, C, sddi, sd2)) {
          if (sddi > smax) smax = sddi;
          if (sddi < smin) smin = sddi;
        }
      }
    }
  }
  if (smax == -DBL_MAX) smax = 0;
  if (smin == DBL_MAX) smin = 0;
  *sdd_max = smax;
  *sdd_min = smin;
}

// FindSddMax/FindSddMin for rows with the joint-space structure
// (timeable_path_joint_spline.cc:320-343): rows D..2D-1 have A = 0, so they produce no
// candidates (.cc:650) and, for a finite candidate, their validity test
// v = 0*sdd + B*sd2 does not involve the candidate: it is evaluated once. A non-finite
// candidate is never selected by the reference either (an infinite one violates its
// own row, whose |A| >= kTiny; a NaN one fails "sddi > sdd"), so skipping them keeps
// the result identical to find_sdd_both on the same rows.
template <class R>
__device__ void find_sdd_both_joint(const R &r, int D, double sd2, double *sdd_max,
                                    double *sdd_min) {
  double smax = -DBL_MAX, smin = DBL_MAX;
  bool vel_ok = true;
  for (int j = D; j < 2 * D; j++) {
    const double v = r.b(j) * sd2;
    if (v + kTiny < r.lo(j) || v - kTiny > r.hi(j)) vel_ok = false;
  }
  if (vel_ok) {
    for (int i = 0; i < D; i++) {
      const double A = r.a(i);
      if (!is_tiny(A)) {
        const double bs = r.b(i) * sd2;
        for (int w = 0; w < 2; w++) {
          const double lim = w ? r.hi(i) : r.lo(i);
          const double sddi = (lim - bs) / A;
          if (!(fabs(sddi) <= DBL_MAX)) continue;
          if ((sddi > smax) || (sddi < smin)) {
            bool ok = true;
            for (int j = 0; j < D; j++) {
              const double v = r.a(j) * sddi + r.b(j) * sd2;
              if (v + kTiny < r.lo(j) || v - kTiny > r.hi(j)) { ok = false; break; }
            }
            if (ok) {
              if (sddi > smax) smax = sddi;
              if (sddi < smin) smin = sddi;
            }
          }
        }
      }
    }
  }
  if (smax == -DBL_MAX) smax = 0;
  if (smin == DBL_MAX) smin = 0;
  *sdd_max = smax;
  *sdd_min = smin;
}

// find_sdd_both_joint with the joint count known at compile time: q' (a) and q'' (b) of
// the sample live in registers, b*sd2 is formed once per row, every loop is unrolled and
// the row checks are branch-free. Same operations per candidate and per row check as
// find_sdd_both_joint, hence the same bits. hi[0..D) are the acceleration bounds (their
// lower bounds are -hi, timeable_path_joint_spline.cc:327-330), hi[D..2D) the velocity
// bounds (lower 0).
template <int D>
__device__ __forceinline__ void find_sdd_both_joint_fixed(const double (&a)[D], const double (&b)[D],
                                                          const double *lim_hi, double sd2,
                                                          double *sdd_max, double *sdd_min) {
  double smax = -DBL_MAX, smin = DBL_MAX;
  bool vel_ok = true;
#pragma unroll
  for (int j = 0; j < D; j++) {
    const double v = (a[j] * a[j]) * sd2;
    if (v + kTiny < 0.0 || v - kTiny > lim_hi[D + j]) vel_ok = false;
  }
  if (vel_ok) {
    double bs[D], hi[D];
#pragma unroll
    for (int j = 0; j < D; j++) { bs[j] = b[j] * sd2; hi[j] = lim_hi[j]; }
#pragma unroll
    for (int i = 0; i < D; i++) {
      const double A = a[i];
      if (!is_tiny(A)) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
          const double lim = w ? hi[i] : -hi[i];
          const double sddi = (lim - bs[i]) / A;
          if ((fabs(sddi) <= DBL_MAX) && ((sddi > smax) || (sddi < smin))) {
            bool bad = false;
#pragma unroll
            for (int j = 0; j < D; j++) {
              // (v + kTiny < -hi) || (v - kTiny > hi) in one comparison: for hi >= 0 only the
              // term on v's own side can be true, and fl(v + kTiny) = -fl(-v - kTiny), so both
              // cases read fl(|v| - kTiny) > hi. (hi < 0 makes lower >= upper: the path has
              // failed its setup check and this value is never used.)
              const double v = a[j] * sddi + bs[j];
              bad = bad | (fabs(v) - kTiny > hi[j]);
            }
            if (!bad) {
              if (sddi > smax) smax = sddi;
              if (sddi < smin) smin = sddi;
            }
          }
        }
      }
    }
  }
  if (smax == -DBL_MAX) smax = 0;
  if (smin == DBL_MAX) smin = 0;
  *sdd_max = smax;
  *sdd_min = smin;
}

// find_sdd_both_joint_fixed with a cheap screen in front of the exact work. A candidate
// sdd_i = (+-hi_i - b_i sd2) / a_i is admissible iff it lies in every row's interval
//   [(-hi_j - kTiny - b_j sd2) / a_j, (hi_j + kTiny - b_j sd2) / a_j]   (ends swapped for a_j < 0),
// i.e. in their intersection [Lmax, Umin]. The interval ends are formed with the hardware
// reciprocal estimate (the candidates themselves are among those ends), and only candidates
// inside the intersection widened by 1e-6 relative -- typically the two constraints that are
// active at the LP vertex -- go through the reference's operations: IEEE division, the row
// checks in the reference's arithmetic, max/min of the admissible ones. A candidate outside the
// widened intersection violates some row by far more than any rounding of the exact check, so
// the reference rejects it as well; NaN estimates are never screened out. The survivors are
// taken from the thread's LDS columns Q1/Q2 (dynamic row index per lane); the row checks use
// the register copies. Same result as find_sdd_both_joint_fixed, bit for bit.
template <int D>
__device__ __forceinline__ void find_sdd_both_joint_screened(const double (&a)[D], const double (&b)[D],
                                                             const double *Q1, const double *Q2, int stride,
                                                             const double *lim_hi, double sd2,
                                                             double *sdd_max, double *sdd_min) {
  double smax = -DBL_MAX, smin = DBL_MAX;
  // (no short-circuit conditions and no small if-bodies below: every divergent branch costs scalar
  // instructions of a unit the whole CU shares -- selects instead)
  bool vel_bad = false;
#pragma unroll
  for (int j = 0; j < D; j++) {
    const double v = (a[j] * a[j]) * sd2;
    vel_bad = vel_bad | (v + kTiny < 0.0) | (v - kTiny > lim_hi[D + j]);
  }
  const bool vel_ok = !vel_bad;
  if (vel_ok) {
    double bs[D], hi[D], c_lo[D], c_hi[D];
    double Lmax = -DBL_MAX, Umin = DBL_MAX;
#pragma unroll
    for (int j = 0; j < D; j++) {
      bs[j] = b[j] * sd2;
      hi[j] = lim_hi[j];
      const double r = __builtin_amdgcn_rcp(a[j]);
      c_lo[j] = (-hi[j] - bs[j]) * r;          // estimate of the candidate on the lower bound
      c_hi[j] = (hi[j] - bs[j]) * r;           // ... on the upper bound
      const bool counts = !is_tiny(a[j]);
      const double pad = kTiny * fabs(r);
      const double l = fmin(c_lo[j], c_hi[j]) - pad, u = fmax(c_lo[j], c_hi[j]) + pad;
      // NaN ends (overflowing estimates) leave the intersection unchanged: never screens
      Lmax = (counts & (l > Lmax)) ? l : Lmax;
      Umin = (counts & (u < Umin)) ? u : Umin;
    }
    unsigned survivors = 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
      const bool counts = !is_tiny(a[i]);
#pragma unroll
      for (int w = 0; w < 2; w++) {
        const double c = w ? c_hi[i] : c_lo[i];
        const double mg = 1e-6 * (fabs(Lmax) + fabs(Umin) + fabs(c)) + 1e-290;
        const bool out = (c < Lmax - mg) | (c > Umin + mg);
        survivors |= (counts & !out) ? (1u << (2 * i + w)) : 0u;
      }
    }
    while (__any(survivors != 0u)) {
      const bool live = survivors != 0u;
      const int slot = live ? (__ffs((int)survivors) - 1) : 0;
      survivors &= survivors - 1u;
      const int i = slot >> 1;
      const double A = Q1[i * stride];
      const double lim = (slot & 1) ? lim_hi[i] : -lim_hi[i];
      const double sddi = (lim - Q2[i * stride] * sd2) / A;
      if (live && (fabs(sddi) <= DBL_MAX) && ((sddi > smax) || (sddi < smin))) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < D; j++) {
          const double v = a[j] * sddi + bs[j];
          bad = bad | (fabs(v) - kTiny > hi[j]);
        }
        if (!bad) {
          if (sddi > smax) smax = sddi;
          if (sddi < smin) smin = sddi;
        }
      }
    }
  }
  if (smax == -DBL_MAX) smax = 0;
  if (smin == DBL_MAX) smin = 0;
  *sdd_max = smax;
  *sdd_min = smin;
}

// ---------------------------------------------------------------------------
// 2-variable LP: FindMaxSd2Simplex, time_optimal_path_timing.cc:1149-1363, with
// IsOptimal :1105-1147. The reference's constraint_set_ / active_set_ vectors
// become bit sets over the 2C (row, bound) slots in the reference's iteration
// order: slot 2c = (c, upper), slot 2c+1 = (c, lower). Erase keeps order, ties
// resolve by first-seen, exactly as the vectors do.
// ---------------------------------------------------------------------------
template <int WORDS>
struct SlotSet {
  uint64_t w[WORDS];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < WORDS; i++) w[i] = 0;
  }
  __device__ __forceinline__ void fill(int n) {
#pragma unroll
    for (int i = 0; i < WORDS; i++) {
      const int lo = i * 64;
      w[i] = (n >= lo + 64) ? ~0ull : (n > lo ? ((1ull << (n - lo)) - 1ull) : 0ull);
    }
  }
  // drop the odd slots (lower bounds) of rows first_row .. : see lp_find_max_sd2
  __device__ __forceinline__ void drop_lower_from(int first_row, int rows) {
#pragma unroll
    for (int i = 0; i < WORDS; i++) {
      uint64_t m = 0;
      for (int c = first_row; c < rows; c++) {
        const int s = 2 * c + 1;
        if ((s >> 6) == i) m |= 1ull << (s & 63);
      }
      w[i] &= ~m;
    }
  }
  __device__ __forceinline__ void set(int s) {
#pragma unroll
    for (int i = 0; i < WORDS; i++)
      if ((s >> 6) == i) w[i] |= 1ull << (s & 63);
  }
  __device__ __forceinline__ bool has(int s) const {
    bool h = false;
#pragma unroll
    for (int i = 0; i < WORDS; i++)
      if ((s >> 6) == i) h = (w[i] >> (s & 63)) & 1ull;
    return h;
  }
  __device__ __forceinline__ bool any() const {
    uint64_t o = 0;
#pragma unroll
    for (int i = 0; i < WORDS; i++) o |= w[i];
    return o != 0;
  }
  __device__ __forceinline__ int count() const {
    int n = 0;
#pragma unroll
    for (int i = 0; i < WORDS; i++) n += __popcll(w[i]);
    return n;
  }
  __device__ __forceinline__ void remove(const SlotSet &o) {
#pragma unroll
    for (int i = 0; i < WORDS; i++) w[i] &= ~o.w[i];
  }
  // smallest slot >= from, or -1
  __device__ __forceinline__ int next(int from) const {
#pragma unroll
    for (int i = 0; i < WORDS; i++) {
      const int base = i * 64;
      if (from < base + 64) {
        uint64_t m = w[i];
        if (from > base) m &= ~0ull << (from - base);
        if (m) return base + __ffsll((long long)m) - 1;
      }
    }
    return -1;
  }
};

// Cheap screen for the LP's vertex searches. The exact test on a slot is
//   tmp < hi_lim && tmp > lo_lim   with   tmp = num * (1 / den)   (two roundings).
// tmp_apx = num * rcp(den) uses the hardware reciprocal estimate (v_rcp_f64, relative
// error far below 1e-6); a slot whose estimate lies outside the window widened by 1e-6
// relative cannot pass the exact test, so its IEEE division is skipped. A NaN estimate
// never skips.
__device__ __forceinline__ bool lp_cannot_pass(double num, double den, double lo_lim,
                                               double hi_lim) {
  const double tmp_apx = num * __builtin_amdgcn_rcp(den);
  return (tmp_apx > hi_lim + fabs(hi_lim) * 1e-6) || (tmp_apx < lo_lim - fabs(lo_lim) * 1e-6);
}

template <class R>
__device__ __forceinline__ bool lp_pair_optimal(const R &r, int s1, int s2) {
  const int first = s1 >> 1, second = s2 >> 1;
  const bool first_upper = !(s1 & 1), second_upper = !(s2 & 1);
  const double a1 = r.a(first), a2 = r.a(second);
  const double denom = a2 * r.b(first) - a1 * r.b(second);
  const double t1 = denom * a1;
  const double t2 = denom * (-a2);
  // (.cc:1105-1147 as one expression: first_upper ? (second_upper ? t1 <= 0 && t2 <= 0 : t1 >= 0 && t2 <= 0)
  //  : (second_upper ? t1 <= 0 && t2 >= 0 : t1 >= 0 && t2 >= 0), false for |denom| < kTiny)
  // -- the sign asked of t1 follows the SECOND slot's bound, that of t2 the FIRST slot's
  const bool c1 = second_upper ? (t1 <= 0) : (t1 >= 0);
  const bool c2 = first_upper ? (t2 <= 0) : (t2 >= 0);
  return !(fabs(denom) < kTiny) & c1 & c2;
}

// SD > 0 (with SC = number of rows): the first SD rows' A and B are also available in the
// register arrays ra / rb, and the two row scans below are unrolled over the SC rows with
// static indices (no bit scans, no LDS reads for those rows) instead of walking the bit set;
// slots are still visited in ascending order and with the same operations.
template <int WORDS, class R, int SD = 0, int SC = 0>
__device__ void lp_find_max_sd2(const R &r, int C, double *sd2max, double *sddmax,
                                double *sd2zero, int b_only_from = -1,
                                const double *ra = nullptr, const double *rb = nullptr) {
  typedef SlotSet<WORDS> Set;
  Set cset, act;
  cset.fill(2 * C);
  // Rows b_only_from .. C-1 have A = 0, B >= 0 and lower <= 0 (the velocity rows of a joint
  // path, the Cartesian rows). Their lower-bound slot can never be entered: on the line
  // sdd = 0 (step 2) a row with B > 0 only offers its upper bound, and on a search line its
  // crossing (lower - 0*a)/B is <= 0, never beyond the current sd2 > 0. Dropping those slots
  // from the constraint set up front shortens every pass without changing any decision.
  if (b_only_from >= 0) cset.drop_lower_from(b_only_from, C);
  act.clear();

  auto vel_b = [&](int c) -> double {     // B of row c >= SD: q'^2 from the register copy where there is one
    if (SD > 0 && c < 2 * SD) { const double v = ra[c - SD]; return v * v; }
    return r.b(c);
  };
  // Step 2: largest feasible sd2 on the line sdd = 0.
  double sd2 = DBL_MAX, sdd = 0.0;
  // upper edge of the screen's window, hi_lim + |hi_lim| * 1e-6 with hi_lim = sd2 + kTiny: follows sd2
  double hi_w = (sd2 + kTiny) + fabs(sd2 + kTiny) * 1e-6;
  // (one branch per row: which bound, whether the row counts at all and the screen are folded into one
  // condition -- divergent branches cost scalar instructions, and the scalar unit is shared by the CU)
  auto step2_row = [&](int c, double Bc) {
    const bool pos = Bc > kTiny, neg = Bc < -kTiny;          // (|Bc| < kTiny, NaN: neither)
    const double lim = pos ? r.hi(c) : r.lo(c);
    // lp_cannot_pass(lim, Bc, 0, sd2 + kTiny) with the window's upper edge kept in hi_w
    const double ap = lim * __builtin_amdgcn_rcp(Bc);
    if ((pos | neg) & !((ap > hi_w) | (ap < 0.0))) {
      const double invB = 1.0 / Bc;
      const double tmp = lim * invB;
      if (tmp < (sd2 + kTiny) && tmp > 0) {
        if (tmp < sd2 - kTiny) act.clear();
        act.set(2 * c + (pos ? 0 : 1));
        sd2 = tmp;
        hi_w = (sd2 + kTiny) + fabs(sd2 + kTiny) * 1e-6;
      }
    }
  };
  if (SD > 0) {
#pragma unroll
    for (int c = 0; c < SC; c++) step2_row(c, (c < SD) ? rb[c < SD ? c : 0] : vel_b(c));
  } else {
    for (int c = 0; c < C; c++) step2_row(c, r.b(c));
  }
  if (sd2 > kMaxSd2 || !act.any()) {
    *sd2zero = kMaxSd2;
    *sd2max = kMaxSd2;
    *sddmax = 0.0;
    return;
  }
  *sd2zero = sd2;

  // Step 3 at the start point. slope = |A * (1/B)| as pushed at .cc:1199-1212.
  int search = act.next(0);
  if (act.count() >= 2) {
    double best_slope = fabs(r.a(search >> 1) * (1.0 / r.b(search >> 1)));
    for (int s1 = act.next(0); s1 >= 0; s1 = act.next(s1 + 1)) {
      const double slope = fabs(r.a(s1 >> 1) * (1.0 / r.b(s1 >> 1)));
      if (slope < best_slope) {
        best_slope = slope;
        search = s1;
      }
      for (int s2 = act.next(s1 + 1); s2 >= 0; s2 = act.next(s2 + 1)) {
        if (lp_pair_optimal(r, s1, s2)) {
          *sd2max = sd2;
          *sddmax = 0.0;
          return;
        }
      }
    }
  }
  cset.remove(act);

  for (int loop = 0; loop < C; loop++) {
    // The reference would index row -1 here if every slope was not < max()
    // (.cc:1322-1331 then :1260); treat as its "no optimum" fallback.
    if (search < 0) break;
    const int sc = search >> 1;
    const double As = r.a(sc);
    if (fabs(As) < kTiny) {
      *sd2max = sd2;
      *sddmax = sdd;
      return;
    }
    // search line sdd = a + b * sd2
    const double invA = 1.0 / As;
    const double b = -r.b(sc) * invA;
    const double a = (search & 1) ? r.lo(sc) * invA : r.hi(sc) * invA;

    act.clear();
    double next_sd2 = DBL_MAX, next_sdd = 0.0;
    // the screen's window of this pass (the margins of lp_cannot_pass, formed once; the upper edge
    // follows next_sd2)
    const double lo_w = sd2 - fabs(sd2) * 1e-6;
    double up_w = (next_sd2 + kTiny) + fabs(next_sd2 + kTiny) * 1e-6;
    auto visit = [&](int s, bool member, double Ac, double Brow, double lim) {
      const double Bc = Ac * b + Brow;
      const double num = lim - Ac * a;
      const double tmp_apx = num * __builtin_amdgcn_rcp(Bc);
      // membership, the |Bc| test and the screen in one condition (one divergent branch per slot)
      if (!(member & !(fabs(Bc) < kTiny) & !((tmp_apx > up_w) | (tmp_apx < lo_w)))) return;
      const double invB = 1.0 / Bc;
      const double tmp = num * invB;
      if (tmp < (next_sd2 + kTiny) && tmp > sd2) {
        if (tmp < next_sd2 - kTiny) act.clear();
        act.set(s);
        next_sd2 = tmp;
        next_sdd = a + b * next_sd2;
        up_w = (next_sd2 + kTiny) + fabs(next_sd2 + kTiny) * 1e-6;
      }
    };
    if (SD > 0) {
#pragma unroll
      for (int c = 0; c < SC; c++) {
        const double Ac = (c < SD) ? ra[c < SD ? c : 0] : r.a(c);
        const double Brow = (c < SD) ? rb[c < SD ? c : 0] : vel_b(c);
        visit(2 * c, cset.has(2 * c), Ac, Brow, r.hi(c));
        // (the lower-bound slots of the A = 0 rows were dropped up front and never come back)
        if (!(b_only_from >= 0 && c >= b_only_from)) visit(2 * c + 1, cset.has(2 * c + 1), Ac, Brow, r.lo(c));
      }
    } else {
      for (int s = cset.next(0); s >= 0; s = cset.next(s + 1)) {
        const int c = s >> 1;
        visit(s, true, r.a(c), r.b(c), (s & 1) ? r.lo(c) : r.hi(c));
      }
    }
    if (!act.any()) {
      *sd2max = *sd2zero;
      *sddmax = 0.0;
      return;
    }
    // Step 3: optimality of {active set, previous search} and next direction.
    const int old_search = search;
    search = -1;
    double best_slope = DBL_MAX;
    for (int s1 = act.next(0); s1 >= 0; s1 = act.next(s1 + 1)) {
      const int c1 = s1 >> 1;
      const double Ac = r.a(c1);
      const double slope = fabs(Ac * (1.0 / (Ac * b + r.b(c1))));
      if (slope < best_slope) {
        best_slope = slope;
        search = s1;
      }
      bool opt = false;
      for (int s2 = act.next(s1 + 1); s2 >= 0 && !opt; s2 = act.next(s2 + 1))
        opt = lp_pair_optimal(r, s1, s2);
      if (!opt) opt = lp_pair_optimal(r, s1, old_search);  // previous search is last
      if (opt) {
        *sd2max = next_sd2;
        *sddmax = next_sdd;
        if (next_sd2 > kMaxSd2) {
          *sd2max = kMaxSd2;
          *sddmax = 0.0;
        }
        return;
      }
    }
    cset.remove(act);
    sd2 = next_sd2;
    sdd = next_sdd;
  }
  *sd2max = *sd2zero;
  *sddmax = 0.0;
}

// ---------------------------------------------------------------------------
// Degree-2 B-spline: knot span (splines/bspline_base.cc:218-246), basis and
// derivatives up to order 2 (NURBS A2.3 as restated at :268-348, unrolled for
// p = 2, der = 2; every product, quotient and sum keeps the generic
// algorithm's order, including the additions of zero-valued "saved").
// ders[k][j], k = derivative order, j = basis index.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int knot_span_deg2(const double *knots, int num_knots, double u) {
  if (u == knots[num_knots - 1]) return num_knots - 2 - 2;
  int lo = 2, hi = num_knots - 2;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (knots[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

__device__ __forceinline__ void basis_ders_deg2(const double *knots, int span, double u,
                                                double ders[3][3]) {
  double ndu[3][3];
  const double left1 = u - knots[span];
  const double right1 = knots[span + 1] - u;
  const double left2 = u - knots[span - 1];
  const double right2 = knots[span + 2] - u;
  ndu[0][0] = 1.0;
  // j = 1
  {
    double saved = 0.0;
    ndu[1][0] = right1 + left1;
    const double tmp = ndu[0][0] / ndu[1][0];
    ndu[0][1] = saved + right1 * tmp;
    saved = left1 * tmp;
    ndu[1][1] = saved;
  }
  // j = 2
  {
    double saved = 0.0;
    ndu[2][0] = right1 + left2;
    double tmp = ndu[0][1] / ndu[2][0];
    ndu[0][2] = saved + right1 * tmp;
    saved = left2 * tmp;
    ndu[2][1] = right2 + left1;
    tmp = ndu[1][1] / ndu[2][1];
    ndu[1][2] = saved + right2 * tmp;
    saved = left1 * tmp;
    ndu[2][2] = saved;
  }
  ders[0][0] = ndu[0][2];
  ders[0][1] = ndu[1][2];
  ders[0][2] = ndu[2][2];

  // r = 0: k=1: rk=-1,pk=1: j1=1,j2=0 (empty); r<=pk: a[1][1] = -a[0][0]/ndu[2][0]
  //        k=2: rk=-2,pk=0: j1=2,j2=1 (empty); r<=pk: a[0][2] = -a[1][1]/ndu[1][0]
  {
    const double a00 = 1.0;
    double d = 0.0;
    const double a11 = -a00 / ndu[2][0];
    d += a11 * ndu[0][1];
    ders[1][0] = d;
    d = 0.0;
    const double a02 = -a11 / ndu[1][0];
    d += a02 * ndu[0][0];
    ders[2][0] = d;
  }
  // r = 1: k=1: rk=0,pk=1: r>=k: a[1][0] = a[0][0]/ndu[2][0], d = a10*ndu[0][1];
  //             j1=1, j2=0 (r-1<=pk -> k-1=0): empty; r<=pk: a[1][1] = -a[0][0]/ndu[2][1]
  //        k=2: rk=-1,pk=0: r<k; j1=1; r-1<=pk -> j2=1:
  //             a[0][1] = (a[1][1]-a[1][0])/ndu[1][0]; d += a01*ndu[0][0]; r<=pk? 1<=0 no.
  {
    const double a00 = 1.0;
    double d = 0.0;
    const double a10 = a00 / ndu[2][0];
    d = a10 * ndu[0][1];
    const double a11 = -a00 / ndu[2][1];
    d += a11 * ndu[1][1];
    ders[1][1] = d;
    d = 0.0;
    const double a01 = (a11 - a10) / ndu[1][0];
    d += a01 * ndu[0][0];
    ders[2][1] = d;
  }
  // r = 2: k=1: rk=1,pk=1: a[1][0] = a[0][0]/ndu[2][1], d = a10*ndu[1][1];
  //             j1=1; r-1<=pk (1<=1) -> j2=0: empty; r<=pk? no.
  //        k=2: rk=0,pk=0: a[0][0]' = a[1][0]/ndu[1][0], d = that*ndu[0][0];
  //             j1=1; r-1<=pk? 1<=0 no -> j2=p-r=0: empty; r<=pk? no.
  {
    const double a00 = 1.0;
    double d;
    const double a10 = a00 / ndu[2][1];
    d = a10 * ndu[1][1];
    ders[1][2] = d;
    const double a00b = a10 / ndu[1][0];
    d = a00b * ndu[0][0];
    ders[2][2] = d;
  }
  // factors: row 1 *= p (=2); row 2 *= p*(p-1) (=2)
#pragma unroll
  for (int j = 0; j < 3; j++) {
    ders[1][j] *= 2.0;
    ders[2][j] *= 2.0;
  }
}

// ---------------------------------------------------------------------------
// Wave-level helpers (64 lanes).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = (o > v) ? o : v;
  }
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = (o < v) ? o : v;
  }
  return v;
}
// value of lane `src_lane`, which must be wave-uniform (v_readlane_b32)
__device__ __forceinline__ double wave_bcast_const(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_bcast_f64(double v, int src_lane) {
  return __shfl(v, src_lane, 64);
}

}  // namespace tpamd
