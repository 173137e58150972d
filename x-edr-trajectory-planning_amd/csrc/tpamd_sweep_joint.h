// tpamd_sweep_joint.h -- the sweep kernel specialised for joint-space paths
// (C = 2D rows with the structure of timeable_path_joint_spline.cc:320-343).
//
// One 64-lane wave per path, as the generic k_sweep, but:
//  * D is a template parameter: every row loop is unrolled.
//  * Memory: each sample has one contiguous record of R = 2D+2 doubles
//    [q'_d, q''_d pairs | final sd2_max | type bits] (written by k_sample_lp_joint and
//    k_boundary_final). The sweep touches records strictly sequentially, so they are
//    staged through LDS in tiles of 32 samples: the tile after the current one (in
//    sweep direction) is fetched into registers by all 64 lanes with coalesced 16-byte
//    loads while the current tile is being consumed, and dropped into the 2-slot LDS
//    ring when the sweep reaches it. Per step the wave then reads its record from LDS
//    (broadcast reads), one step ahead of use. No global-memory latency is left on the
//    sequential chain except one tile fill at the start of an extremal that begins
//    outside the two resident tiles.
//  * sd2_ lives in LDS; sdd_ is written straight to the output array (it is never
//    read back inside the extremal loops); the type bytes are copied to LDS once for the
//    switching-point search.
//  * FindSddMax/FindSddMin exploit the row structure. Rows D..2D-1 have A = 0, so
//    (time_optimal_path_timing.cc:650) they generate no candidates, and for a
//    finite candidate sdd their validity test v = 0*sdd + q'^2*sd2 does not depend
//    on the candidate: it is evaluated once per step by D otherwise idle lanes.
//    Non-finite candidates can never be selected in the reference either: an
//    infinite one violates its own row (|A| >= kTiny there), a NaN one fails the
//    "sddi > sdd" comparison. The 2D candidates of rows 0..D-1 sit one per lane,
//    each lane validates its candidate against the D acceleration rows, and a
//    DPP butterfly over 16-lane rows keeps the extreme valid candidate.
//  * Solver scalars are re-uniformised with readfirstlane after cross-lane steps so
//    that the control flow is scalar.
// The arithmetic (operands, order, no contraction) is the generic kernel's and
// the reference's; results are bit-identical.
#pragma once

#include <type_traits>

#include "tpamd_kernels.h"

namespace tpamd {

__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// (bound_ctrl with a zero "old": every lane is written, so the destination needs no copy of the
// source first -- two moves per value instead of four. A lane whose source lies outside the
// wave (the ends of a wave shift) reads 0.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// max / min of two values that are never NaN: the bare instruction (fmax/fmin would first
// canonicalise both operands under IEEE mode, doubling the cost of every reduction level).
template <bool MAX>
__device__ __forceinline__ double ext2_raw(double a, double b) {
  double r;
  if (MAX) asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Extreme value over each 16-lane DPP row (butterfly: xor 1, xor 2, half-row
// mirror, row mirror). Inputs must not be NaN.
template <bool MAX>
__device__ __forceinline__ double row16_extreme(double v) {
  double o;
  o = dpp_f64<0xB1>(v);  v = ext2_raw<MAX>(o, v);   // quad_perm [1,0,3,2]
  o = dpp_f64<0x4E>(v);  v = ext2_raw<MAX>(o, v);   // quad_perm [2,3,0,1]
  o = dpp_f64<0x141>(v); v = ext2_raw<MAX>(o, v);   // row_half_mirror
  o = dpp_f64<0x140>(v); v = ext2_raw<MAX>(o, v);   // row_mirror
  return v;
}

// ---------------------------------------------------------------------------------------
// Ordered sum of 64 values, one per lane: lane L returns ((t + d_0) + d_1) ... + d_L, the
// additions in exactly that order (time_optimal_path_timing.cc:453-454 sums left to right).
// One instruction per value: v_fmac_f64 with a DPP row_newbcast operand computes
// acc = d[lane c of the row] * 1.0 + acc -- one rounding, the sum's -- in every enabled lane of
// the row, and EXEC loses one lane per value so that lane c keeps the sum through its own value
// (a lane never sees a later value, so an infinite increment leaves the earlier sums finite).
// Rows are chained through row_bcast:15. (tools/micro/chain_bench.hip: 14 cycles per value with
// two waves per SIMD, against 23 for an add plus a 64-bit lane shift per value.)
// ---------------------------------------------------------------------------------------
#define TPAMD_OS_STEP(HALF, C)                                                            \
  "s_mov_b32 exec_" HALF ", %[m" #C "]\n\t"                                               \
  "v_fmac_f64_dpp %[acc], %[d], %[one] row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t"
#define TPAMD_OS_ROW(HALF, OTHER)                                                                      \
  asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_" OTHER ", 0\n\t"                              \
               TPAMD_OS_STEP(HALF, 0) TPAMD_OS_STEP(HALF, 1) TPAMD_OS_STEP(HALF, 2) TPAMD_OS_STEP(HALF, 3)     \
               TPAMD_OS_STEP(HALF, 4) TPAMD_OS_STEP(HALF, 5) TPAMD_OS_STEP(HALF, 6) TPAMD_OS_STEP(HALF, 7)     \
               TPAMD_OS_STEP(HALF, 8) TPAMD_OS_STEP(HALF, 9) TPAMD_OS_STEP(HALF, 10) TPAMD_OS_STEP(HALF, 11)   \
               TPAMD_OS_STEP(HALF, 12) TPAMD_OS_STEP(HALF, 13) TPAMD_OS_STEP(HALF, 14) TPAMD_OS_STEP(HALF, 15) \
               "s_mov_b64 exec, %[sv]\n\t"                                                             \
               : [acc] "+v"(acc), [sv] "=&s"(save)                                                     \
               : [d] "v"(d), [one] "v"(one), TPAMD_OS_M(0), TPAMD_OS_M(1), TPAMD_OS_M(2), TPAMD_OS_M(3), \
                 TPAMD_OS_M(4), TPAMD_OS_M(5), TPAMD_OS_M(6), TPAMD_OS_M(7), TPAMD_OS_M(8), TPAMD_OS_M(9), \
                 TPAMD_OS_M(10), TPAMD_OS_M(11), TPAMD_OS_M(12), TPAMD_OS_M(13), TPAMD_OS_M(14), TPAMD_OS_M(15))
// lanes c..15 of 16-lane row ROW add value c of that row
template <int ROW>
__device__ __forceinline__ void ordered_sum_row(double &acc, double d, double one) {
  constexpr unsigned SH = (ROW & 1) ? 16u : 0u;
#define TPAMD_OS_M(C) [m##C] "n"((int)(((0xffffu << C) & 0xffffu) << SH))
  unsigned long long save;
  if (ROW < 2) TPAMD_OS_ROW("lo", "hi");
  else TPAMD_OS_ROW("hi", "lo");
#undef TPAMD_OS_M
}
// lane 15 of row ROW-1 to every lane of row ROW
template <int ROW>
__device__ __forceinline__ void ordered_sum_carry(double &acc) {
  int lo = __double2loint(acc), hi = __double2hiint(acc);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 1 << ROW, 0xf, false);   // row_bcast:15
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 1 << ROW, 0xf, false);
  acc = __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ordered_sum64(double t, double d) {
  double acc = t;
  const double one = 1.0;
  ordered_sum_row<0>(acc, d, one);
  ordered_sum_carry<1>(acc);
  ordered_sum_row<1>(acc, d, one);
  ordered_sum_carry<2>(acc);
  ordered_sum_row<2>(acc, d, one);
  ordered_sum_carry<3>(acc);
  ordered_sum_row<3>(acc, d, one);
  return acc;
}

// The same sum without touching EXEC, for FINITE values d >= 0: every lane of the row takes part
// in all sixteen steps, but multiplies the values after its own by 0.0 instead of 1.0 --
// acc = d_c * m + acc with m = 1.0 for lanes c..15, 0.0 below; d_c * 0.0 = +0.0 and acc + 0.0 = acc
// exactly (acc is never -0.0 after its first addition of a value >= +0.0). That leaves the bare
// chain of sixteen dependent v_fmac_f64 per row. The DPP row mask confines a pass to its row.
// Not for infinite or NaN values (inf * 0.0): the caller checks and falls back to ordered_sum64.
struct OrderedSumMasks {
  double m[15];   // m[c-1]: 1.0 in lanes whose position in the row is >= c, else 0.0 (c = 1..15)
  __device__ __forceinline__ void init(int lane) {
#pragma unroll
    for (int c = 1; c < 16; c++) m[c - 1] = ((lane & 15) >= c) ? 1.0 : 0.0;
  }
};
#define TPAMD_OSM_STEP(C, ROWMASK)                                                                    \
  "v_fmac_f64_dpp %[acc], %[d], %[m" #C "] row_newbcast:" #C " row_mask:" ROWMASK " bank_mask:0xf\n\t"
#define TPAMD_OSM_ROW(ROWMASK)                                                                          \
  asm volatile(TPAMD_OSM_STEP(0, ROWMASK) TPAMD_OSM_STEP(1, ROWMASK) TPAMD_OSM_STEP(2, ROWMASK)         \
               TPAMD_OSM_STEP(3, ROWMASK) TPAMD_OSM_STEP(4, ROWMASK) TPAMD_OSM_STEP(5, ROWMASK)         \
               TPAMD_OSM_STEP(6, ROWMASK) TPAMD_OSM_STEP(7, ROWMASK) TPAMD_OSM_STEP(8, ROWMASK)         \
               TPAMD_OSM_STEP(9, ROWMASK) TPAMD_OSM_STEP(10, ROWMASK) TPAMD_OSM_STEP(11, ROWMASK)       \
               TPAMD_OSM_STEP(12, ROWMASK) TPAMD_OSM_STEP(13, ROWMASK) TPAMD_OSM_STEP(14, ROWMASK)      \
               TPAMD_OSM_STEP(15, ROWMASK)                                                              \
               : [acc] "+v"(acc)                                                                        \
               : [d] "v"(d), [m0] "v"(one), [m1] "v"(k.m[0]), [m2] "v"(k.m[1]), [m3] "v"(k.m[2]),       \
                 [m4] "v"(k.m[3]), [m5] "v"(k.m[4]), [m6] "v"(k.m[5]), [m7] "v"(k.m[6]), [m8] "v"(k.m[7]), \
                 [m9] "v"(k.m[8]), [m10] "v"(k.m[9]), [m11] "v"(k.m[10]), [m12] "v"(k.m[11]),           \
                 [m13] "v"(k.m[12]), [m14] "v"(k.m[13]), [m15] "v"(k.m[14]))
__device__ __forceinline__ double ordered_sum64_finite(double t, double d, const OrderedSumMasks &k) {
  double acc = t;
  const double one = 1.0;
  // (the value d may have been written by the instruction just before: a DPP operand needs two
  // wait states after a VALU write, which the compiler cannot see inside the asm)
  asm volatile("s_nop 1" ::: "memory");
  TPAMD_OSM_ROW("0x1");
  ordered_sum_carry<1>(acc);
  TPAMD_OSM_ROW("0x2");
  ordered_sum_carry<2>(acc);
  TPAMD_OSM_ROW("0x4");
  ordered_sum_carry<3>(acc);
  TPAMD_OSM_ROW("0x8");
  return acc;
}

// Diagnostic build only (-DTPAMD_DIAG): per-path cycle counters written to ws.diag;
// the product build contains none of this.
#ifdef TPAMD_DIAG
// One asm statement per stamp (s_memtime returns through lgkmcnt), fenced against the
// scheduler on both sides (cdna_hip_programming.md section 7, "In-kernel stamps").
__device__ __forceinline__ long long tpamd_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return (long long)t;
}
#define TPAMD_T0(var) const long long var = tpamd_stamp()
#ifdef TPAMD_DIAG_REASONS   // (study build: slots 12, 13, 17, 19..23, 31 count why chain blocks end instead)
#define TPAMD_ACC(slot, var) do { if (!((slot) == 12 || (slot) == 13 || (slot) == 17 || ((slot) >= 19 && (slot) <= 23) || (slot) == 31)) diag[slot] += tpamd_stamp() - (var); } while (0)
#else
#define TPAMD_ACC(slot, var) diag[slot] += tpamd_stamp() - (var)
#endif
#define TPAMD_CNT(slot) diag[slot] += 1
#define TPAMD_ADD(slot, v) diag[slot] += (v)
// the coarse stamps (phases of the kernel: a dozen per path); -DTPAMD_DIAG_LIGHT keeps only these, so
// that the build runs like the product (the fine stamps serialise the hot loops)
#define TPAMD_T0C(var) const long long var = tpamd_stamp()
#define TPAMD_ACCC(slot, var) diag[slot] += tpamd_stamp() - (var)
#ifdef TPAMD_DIAG_LIGHT
#undef TPAMD_T0
#undef TPAMD_ACC
#define TPAMD_T0(var)
#define TPAMD_ACC(slot, var)
#endif
#else
#define TPAMD_T0(var)
#define TPAMD_ACC(slot, var)
#define TPAMD_T0C(var)
#define TPAMD_ACCC(slot, var)
#define TPAMD_CNT(slot)
#define TPAMD_ADD(slot, v)
#endif

#ifndef TPAMD_TILE_SAMPLES
#define TPAMD_TILE_SAMPLES 32
#endif
// After a speculative block that ended because the active constraint changed, speculate on the
// constraint that took over instead of finding it with a scalar FindSdd step. Measured (round 3):
// scalar steps per path 97 -> 89 and the mean path 2 % shorter in cycles, but 2.4 M VALU instructions
// more per launch (the hint bookkeeping of every block) -- the step, which is bound by the instruction
// count, got 0.5 % (pipelined) to 1.8 % (one kernel at a time) slower. Off; kept for the record.
#ifndef TPAMD_CHAIN_GUESS
#define TPAMD_CHAIN_GUESS 0
#endif
// A speculative block whose first failing step fails only because its (verified) new sd2 lies above
// the limit curve finishes that step itself (leave_on_curve) instead of handing it to a scalar step.
#ifndef TPAMD_CURVE_EXIT
#define TPAMD_CURVE_EXIT 1
#endif
constexpr int kTileSamples = TPAMD_TILE_SAMPLES;   // largest tile (the engine pads the records by one)
// Samples per tile of the D-joint sweep: wide records (D > 8: 30 doubles at D = 14) take 16-sample
// tiles, which halves the prefetch registers and the rings (the 14-joint kernel spilled 288 bytes
// per lane with 32-sample tiles).
template <int D>
struct TileCfg {
  static constexpr int kSamples = (D > 8 && TPAMD_TILE_SAMPLES > 16) ? 16 : TPAMD_TILE_SAMPLES;
};

// 16-byte pair as a native vector type: plain loads/stores that the compiler keeps in
// registers (copies of HIP's f64x2 struct are emitted as memcpy and can pin the
// destination in scratch).
typedef double f64x2 __attribute__((ext_vector_type(2)));

// K 16-byte registers as a recursive aggregate (no array, hence no indexing: every element
// is promoted to VGPRs). load/store walk the pack with compile-time recursion.
template <int K>
struct RegPack {
  f64x2 v;
  RegPack<K - 1> rest;
  template <int LIMIT>   // chunks beyond LIMIT are neither loaded nor stored
  __device__ __forceinline__ void load(const f64x2 *src, int c) {
    if (c < LIMIT) v = src[c];
    rest.template load<LIMIT>(src, c + 64);
  }
  template <int LIMIT>
  __device__ __forceinline__ void store(f64x2 *dst, int c) const {
    if (c < LIMIT) dst[c] = v;
    rest.template store<LIMIT>(dst, c + 64);
  }
};
template <>
struct RegPack<0> {
  template <int LIMIT> __device__ __forceinline__ void load(const f64x2 *, int) {}
  template <int LIMIT> __device__ __forceinline__ void store(f64x2 *, int) const {}
};

// Lane layout of one FindSddMax/FindSddMin step (D joints, 2D candidates):
//   lane = c * PARTS + p,  c = candidate (row r = c >> 1, bound = c & 1), p = part.
//   The PARTS lanes of a candidate all compute the same candidate value and each
//   validates it against its own share of the D acceleration rows (rows p, p+PARTS,
//   ...); a candidate is admissible iff all its parts agree. PARTS = 4 for D <= 7
//   (56 lanes for D = 7), 2 for D <= 16, so that 2D * PARTS <= 64.
//   Lanes [0, D) additionally check one velocity row each.
template <int D>
struct JointLayout {
  static constexpr int PARTS = (D <= 7) ? 4 : ((D <= 16) ? 2 : 1);
  static constexpr int RPL = (D + PARTS - 1) / PARTS;        // rows validated per lane
  static constexpr int kCandLanes = 2 * D * PARTS;
  static constexpr int kRows16 = (kCandLanes + 15) / 16;     // 16-lane DPP rows in use
  static_assert(kCandLanes <= 64, "candidate lanes must fit one wave");
  // Chain verification (follow_chain): kChain steps x GRP lanes; lane part p of a step's
  // group owns candidates p, p+GRP, ... and rows p, p+GRP, ...
#ifndef TPAMD_CHAIN_STEPS
#define TPAMD_CHAIN_STEPS 16
#endif
  static constexpr int kChain = TPAMD_CHAIN_STEPS;
  static constexpr int GRP = 64 / kChain;
  static constexpr int CPL = (2 * D + GRP - 1) / GRP;
  static constexpr int VPL = (D + GRP - 1) / GRP;
  static constexpr unsigned long long kPart0 =
      (GRP == 4) ? 0x1111111111111111ull : ((GRP == 2) ? 0x5555555555555555ull : ~0ull);
  static_assert(GRP == 1 || GRP == 2 || GRP == 4, "chain steps must be 16, 32 or 64");
};

template <bool MAX>
__device__ __forceinline__ double ext2(double a, double b) {
  return ext2_raw<MAX>(a, b);
}

// E: extra rows with A = 0 and an explicit B (Cartesian paths: 2, stored as one pair after
// the D (q', q'') pairs); they behave like velocity rows with lower = -upper.
template <int D, int E = 0>
struct JointSweep {
  typedef JointLayout<D> L;
  static_assert(E == 0 || E == 2, "extra rows come as one pair");
  static constexpr int R = 2 * D + E + 2;                   // doubles per record
  static constexpr int kMt = D + E / 2;                     // pair index of (sd2_max, type)
  static constexpr int kTile = TileCfg<D>::kSamples;        // samples per tile
  static constexpr int kChunks = kTile * R / 2;             // 16-byte chunks per tile
  static constexpr int kChunksPerLane = (kChunks + 63) / 64;
#ifdef TPAMD_DIAG
  // per wave: 0 extremal cycles inside the loop, 1 cycles waiting for the partner at the end of
  // a loop, 2 critical-point search, 3 tail, 4 chain-block cycles, 5 first pair + loop, 6 chain
  // blocks, 7 critical-point mismatches (must be 0), 8 boundary-following blocks, 9 first pair,
  // 10 scalar FindSdd steps, 11 loops, 12 qd/qdd written inside the loop (backward wave), 13
  // tail up to the start of the time samples, 14 chain steps accepted, 15 whole kernel after
  // set-up, 16 tail: sd/dt pass, 17 tail: time integral (backward wave) / last-extremal scan +
  // remaining qd/qdd (forward wave), 18 the literal critical-point walk of this build (to be
  // subtracted from 5 and 15), 19..22 boundary passes (cumulative: flags loaded, re-fits next to
  // isolated points, deferred fixes, final values + classification), 23 re-fits of the final
  // pass. During the boundary passes 10 / 11 count the re-fits of the final / the first pass
  // (the sweep then counts on top: scalar FindSdd steps / loops).
  // 24 scalar-step cycles, 25 boundary-follow cycles, 26 init_carry cycles, 27 tile fills, 28 tile-fill
  // cycles, 29 init_carry calls, 30 blocks started from a guessed constraint
  long long diag[32];
#endif
  int N, lane;
  double ds, two_ds;
  double *sd2;            // LDS [N]
  double *tiles;          // LDS [2][kTile][R]
  const uint8_t *typel;   // LDS [N] copy of the type bytes
  double *sdd_g;          // global: sdd output row of this path
  const double *m_g;      // global: final sd2_max of this path [N]
  const double *rec;      // global: records of this path [N][R]
  int tag0, tag1;         // tile index resident in ring slot 0 / 1 (-1: none)
  // Planner epilogue (see emit_range): output rows of this path [N][D] (null: not requested),
  // the clamp limits a_max[D] in LDS, and the LDS bitmap of samples whose qd/qdd are redone
  // after the tail changed their sdd.
  double *qd_g, *qdd_g;
  const double *aml;
  uint32_t *dirty;
  int end_idx;            // sample at which the last extremal of this wave ended
  // per-lane constants (roles are folded into data so that the hot loop has no role
  // branches): idle lanes carry lim = NaN (their candidate is NaN, hence rejected) and
  // vel_hi = +inf (their velocity check never fails).
  double lim;             // the bound defining this lane's candidate
  double chk_hi[L::RPL];  // upper bounds of the acceleration rows this lane validates
  double vel_hi;          // upper bound of the velocity row this lane checks
  double vel_lo;          // E > 0 only: its lower bound (0 for joint rows, -upper for extras)
  int vel_mode;           // E > 0 only: B = x*x (0), x (1) or y (2) of the pair at vel_off
  double ex_hi[E ? E : 1];   // E > 0 only: upper bounds of the extra rows (uniform)
  int own_off;            // offsets (in f64x2 units) of the pairs this lane reads
  int chk_off[L::RPL];
  int vel_off;
  double row_lo, row_hi;  // AreDerivativesValid: lane j < 2D owns row j
  // chain verification layout (lane = 4*step + part), see follow_chain
  double v_lim[L::CPL];   // bound of this lane's i-th candidate (NaN: none)
  int v_off[L::CPL];      // its row
  double vv_hi[L::VPL];   // upper bound of this lane's i-th velocity row (+inf: none)
  double va_hi[L::VPL];   // upper bound of the acceleration row of the same joint (+inf: none)
  int vv_off[L::VPL];

  struct Rows {
    f64x2 own;            // (q', q'') of the candidate's row
    f64x2 chk[L::RPL];    // pairs of the rows this lane validates
    f64x2 vel;            // pair of the velocity row this lane checks
  };

  // LDS traffic of ONE wave is processed in issue order, so lanes of the same wave see each
  // other's LDS writes without an s_barrier; the fence only pins the compiler's order. (A
  // workgroup barrier here would be wrong in the two-wave kernel, whose waves run
  // different extremals.)
  static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- tile ring -----------------------------------------------------------
  // The prefetch registers live in the calling extremal (a struct-member array is not
  // promoted to registers). Loads always cover a full tile: a partial last tile reads at
  // most 31 records past the path's end, still inside the engine workspace (never used).
  // The (sd2_max, type) pair at the end of a record is NOT read from global memory: the final
  // boundary value comes from the dense array ws.m (one coalesced load by lanes 0..31) and the
  // classification from the LDS type copy, and both are dropped into the tile's records as it
  // is stored (the boundary passes would otherwise have to patch 16 bytes into every 128-byte
  // record: one write transaction per sample).
  struct Prefetch {
    RegPack<kChunksPerLane> r;
    double m;  // lanes 0..kTile-1: final sd2_max of sample t*kTile + lane
    int tag;   // tile index held in r (-1: none)
  };
  __device__ __forceinline__ void issue_tile_loads(int t, Prefetch &pf) const {
    const f64x2 *src = reinterpret_cast<const f64x2 *>(rec + (size_t)t * kTile * R);
    pf.r.template load<kChunks>(src, lane);
    pf.m = m_g[min(t * kTile + (lane & (kTile - 1)), N - 1)];
    pf.tag = t;
  }
  __device__ __forceinline__ void store_tile(int slot, const Prefetch &pf) const {
    f64x2 *dst = reinterpret_cast<f64x2 *>(tiles + (size_t)slot * kTile * R);
    pf.r.template store<kChunks>(dst, lane);
    wave_lds_sync();                 // the chunk stores land before the (m, type) pairs
    if (lane < kTile) {
      const int i = min(pf.tag * kTile + lane, N - 1);
      f64x2 mt;
      mt.x = pf.m;
      mt.y = __longlong_as_double((long long)typel[i]);
      dst[(size_t)lane * (R / 2) + kMt] = mt;
    }
    wave_lds_sync();
  }
  // Make tile t resident; dir tells which neighbour tile to prefetch afterwards.
  __device__ __forceinline__ void fill_tile(int t, int dir, Prefetch &pf) {
    TPAMD_T0(tf_);
    TPAMD_CNT(27);
    if (pf.tag != t) issue_tile_loads(t, pf);
    wave_lds_sync();                 // earlier readers of this slot are done
    store_tile(t & 1, pf);           // waits for the loads, writes LDS
    if (t & 1) tag1 = t; else tag0 = t;
    const int tn = t + dir;
    pf.tag = -1;
    if (tn >= 0 && tn * kTile < N) issue_tile_loads(tn, pf);
    TPAMD_ACC(28, tf_);
  }
  __device__ __forceinline__ void ensure_tile(int idx, int dir, Prefetch &pf) {
    const int t = idx / kTile;
    const int tag = (t & 1) ? tag1 : tag0;
    if (tag != t) fill_tile(t, dir, pf);
  }
  __device__ __forceinline__ const f64x2 *record(int idx) const {
    // slot = (idx / 32) & 1, position = idx % 32  ==  idx & 63 in a 64-record ring
    return reinterpret_cast<const f64x2 *>(tiles) + (size_t)(idx & (2 * kTile - 1)) * (R / 2);
  }
  __device__ __forceinline__ void load_rows(int idx, Rows &r) const {
    const f64x2 *p = record(idx);
    r.own = p[own_off];
#pragma unroll
    for (int k = 0; k < L::RPL; k++) r.chk[k] = p[chk_off[k]];
    r.vel = p[vel_off];
  }
  __device__ __forceinline__ void load_mt(int idx, double &m, int &t) const {
    const f64x2 v = record(idx)[kMt];
    m = v.x;
    t = __double2loint(v.y);
  }

  __device__ __forceinline__ void put_sd2(int i, double v) {
    if (lane == 0) sd2[i] = v;
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void put_sdd(int i, double v) {
    if (lane == 0) sdd_g[i] = v;
  }

  // ---- planner epilogue, incrementally (path_timing_trajectory.cc:458-472) ---------------
  // qd = q' sd, qdd = clamp(q' sdd + q'' sd^2, +-a_max) with sd = sqrt(sd2). The backward wave
  // finishes its extremal of a switching point well before the forward wave does; instead of
  // idling at the loop's closing barrier it writes qd/qdd for the samples below the current
  // critical point, which no extremal of this loop touches any more (emit_range, called from
  // the kernel). A later backward extremal that reaches below that frontier makes the kernel
  // redo the overlap; whatever is left after the last loop, and the few samples whose sdd the
  // tail changes (mark_dirty), are written in the tail. The records are read a second time here
  // (mostly from L2: the forward wave streamed them shortly before), but spread over the whole
  // kernel instead of as one bandwidth-bound pass by every path at once.
  __device__ __forceinline__ bool emitting() const { return qd_g != nullptr || qdd_g != nullptr; }
  __device__ __forceinline__ void emit_value(int i, int d, f64x2 pr, double v, double a,
                                             double amx) const {
    const size_t o = (size_t)i * D + d;
    if (qd_g) qd_g[o] = pr.x * v;
    if (qdd_g) {
      const double v2 = v * v;
      double acc = pr.x * a + pr.y * v2;
      if (acc < -amx) acc = -amx;
      if (acc > amx) acc = amx;
      qdd_g[o] = acc;
    }
  }
  // Samples lo..hi (inclusive) from the current sd2 (LDS), sdd and records (global), 64 samples
  // per trip of a wave: lane l first takes sample s0 + l -- sd = sqrt(sd2) and sdd once per sample,
  // one coalesced load -- then the trip's 64*D (sample, joint) pairs are spread over D passes of
  // 64 lanes, consecutive lanes storing consecutive addresses, every lane fetching its sample's sd
  // and sdd from the lane that holds them (ds_bpermute: the LDS crossbar, no LDS memory). All
  // loads of a trip are issued before its arithmetic. Wave `wv` of `nwv` waves takes every nwv-th
  // trip (one wave: 0 of 1; the workgroup: tid >> 6 of 2).
  // stop != nullptr: give up as soon as *stop == stop_value (checked between trips) -- the
  // partner wave has finished and must not be kept waiting. Returns the first sample NOT
  // written (hi + 1 if the range was completed).
  __device__ __forceinline__ int emit_range(int lo, int hi, int wv, int nwv,
                                            const volatile int *stop = nullptr,
                                            int stop_value = 0) const {
    if (!emitting() || hi < lo) return hi + 1;
    const f64x2 *rec2 = reinterpret_cast<const f64x2 *>(rec);
    for (int s0 = lo + 64 * wv; s0 <= hi; s0 += 64 * nwv) {
      if (stop != nullptr && *stop == stop_value) return s0;                // (uniform)
      const int cnt = min(64, hi - s0 + 1);                                  // samples of this trip
      const int mine = s0 + min(lane, cnt - 1);
      const double s2l = sd2[mine];
      const double al = sdd_g[mine];
      const double vl = sqrt(s2l);
      constexpr int G = (D <= 8) ? D : 7;                                    // passes in flight
#pragma unroll
      for (int u0 = 0; u0 < D; u0 += G) {
        f64x2 pr[G];
        double amx[G];
        int se[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
          const int e = min(64 * (u0 + g) + lane, cnt * D - 1);
          se[g] = e / D;                                                     // sample within the trip
          const int dd = e - se[g] * D;
          pr[g] = rec2[(size_t)(s0 + se[g]) * (R / 2) + dd];
          amx[g] = aml[dd];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
          if (u0 + g >= D) continue;
          const int src = se[g] << 2;
          const double v = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(vl)),
                                            __builtin_amdgcn_ds_bpermute(src, __double2loint(vl)));
          const double a = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(al)),
                                            __builtin_amdgcn_ds_bpermute(src, __double2loint(al)));
          const int e = 64 * (u0 + g) + lane;
          if (e < cnt * D) emit_value(s0 + se[g], e - se[g] * D, pr[g], v, a, amx[g]);
        }
      }
    }
    return hi + 1;
  }
  __device__ __forceinline__ void mark_dirty(int i) const {
    if (lane == 0) atomicOr(&dirty[i >> 5], 1u << (i & 31));
  }

  // FindSddMax (MAX) / FindSddMin, time_optimal_path_timing.cc:638-695.
  // `win` receives a lane holding the selected candidate (-1: none was admissible); it
  // seeds the speculation of follow_chain and has no influence on the result.
  template <bool MAX>
  __device__ __forceinline__ double find_sdd(const Rows &r, double s2, int &win) const {
    constexpr double kSentinel = MAX ? -DBL_MAX : DBL_MAX;
    // velocity row of this lane: v = q'^2 * sd2 against [0, (vmax*safety)^2]; with extra
    // rows some lanes check v = B * sd2 against [-upper, upper] instead
    bool vel_bad;
    if (E == 0) {
      const double vv = (r.vel.x * r.vel.x) * s2;
      vel_bad = (vv + kTiny < 0.0) | (vv - kTiny > vel_hi);
    } else {
      const double sel = (vel_mode == 2) ? r.vel.y : r.vel.x;
      const double bb = (vel_mode == 0) ? sel * sel : sel;
      const double vv = bb * s2;
      vel_bad = (vv + kTiny < vel_lo) | (vv - kTiny > vel_hi);
    }
    // candidate of this lane's group
    const double sddi = (lim - r.own.y * s2) / r.own.x;
    bool bad = (fabs(r.own.x) < kTiny) | (sddi != sddi);
#pragma unroll
    for (int k = 0; k < L::RPL; k++) {
      // (v + kTiny < -hi) | (v - kTiny > hi) as one comparison, see find_sdd_both_joint_fixed
      const double v = r.chk[k].x * sddi + r.chk[k].y * s2;
      bad = bad | (fabs(v) - kTiny > chk_hi[k]);
    }
    double best = bad ? kSentinel : sddi;
    const double mine = best;
    // all parts of a candidate must agree: the group keeps the sentinel if any part set it
    if (L::PARTS >= 2) best = ext2<!MAX>(best, dpp_f64<0xB1>(best));   // xor 1
    if (L::PARTS >= 4) best = ext2<!MAX>(best, dpp_f64<0x4E>(best));   // xor 2
    // extreme over the candidates of each 16-lane row
    if (L::PARTS < 2) best = ext2<MAX>(best, dpp_f64<0xB1>(best));
    if (L::PARTS < 4) best = ext2<MAX>(best, dpp_f64<0x4E>(best));
    best = ext2<MAX>(best, dpp_f64<0x141>(best));                      // row_half_mirror
    best = ext2<MAX>(best, dpp_f64<0x140>(best));                      // row_mirror
    // across the 16-lane rows: every lane folds each row's value (a scalar operand) into its own
    double res = best;
#pragma unroll
    for (int k = 0; k < L::kRows16; k++) {
      const double rowv = readlane_f64(best, 16 * k);
      // (s_nop: a scalar register written by v_readlane needs two wait states before a vector
      // instruction reads it, and the compiler does not look into asm operands for that)
      if (MAX) asm("s_nop 1\n\tv_max_f64 %0, %1, %2" : "=v"(res) : "v"(res), "s"(rowv));
      else asm("s_nop 1\n\tv_min_f64 %0, %1, %2" : "=v"(res) : "v"(res), "s"(rowv));
    }
    const unsigned long long holders = __ballot(mine == res);
    win = __ffsll((long long)holders) - 1;
    if (res == kSentinel) { res = 0; win = -1; }
    if (__ballot(vel_bad) != 0ull) { res = 0; win = -1; }
    return res;
  }

  // Chain verification of one step by its 4 lanes: is `sddw` -- the candidate of row r at
  // sample j, sd2 = s2, already computed with the reference's operations -- exactly what
  // FindSddMax (MAX) / FindSddMin returns there? Yes if (1) it is admissible: all rows
  // hold, acceleration rows split over the 4 lanes, velocity rows likewise; and (2) no
  // other candidate beats it: every candidate that compares better must be inadmissible,
  // which is checked against ONE row only, row r (the one the speculated candidate sits
  // on: anything better than it normally violates that row). A better candidate that
  // passes row r makes the answer "unknown" (true = bad), never wrong: the step then falls
  // to the scalar code.
  // The other candidates are only ever compared, never used: their quotient is formed with a
  // reciprocal taken before the chain ran (rc, relative error far below 1e-6) and counts as
  // "not better" or "violates row r" only beyond a 2e-6 margin; anything closer is "unknown".
  // What the exact comparison would reject the screen rejects too.
  // pvv / pown / ex: the record pairs this lane needs, loaded before the chain ran.
  struct ChainOperands {
    f64x2 pvv[L::VPL];
    f64x2 pown[L::CPL];
    double rc[L::CPL];
    f64x2 ex;
  };
  __device__ __forceinline__ void load_chain_operands(int j, ChainOperands &o) const {
    const f64x2 *p = record(j);
#pragma unroll
    for (int i = 0; i < L::VPL; i++) o.pvv[i] = p[vv_off[i]];
#pragma unroll
    for (int i = 0; i < L::CPL; i++) o.pown[i] = p[v_off[i]];
    if (E > 0) o.ex = p[D];
#pragma unroll
    for (int i = 0; i < L::CPL; i++) o.rc[i] = __builtin_amdgcn_rcp(o.pown[i].x);
  }
  // `guess` (in: -1): the candidate (2 * row + bound) that this lane's failed check points at as the
  // NEXT active constraint -- the bound of an acceleration row the speculated candidate violates,
  // or another candidate that compares better and does not violate the speculated row. Only a hint
  // for the next block's speculation (see add_extremal); never part of a result.
  template <bool MAX>
  __device__ __forceinline__ bool chain_step_bad(const ChainOperands &o, int win_cand, double hi_r,
                                                 f64x2 arow, double s2, double sddw, int &guess) const {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < L::VPL; i++) {       // winner against this lane's share of the rows
      const f64x2 pr = o.pvv[i];
      // (v + kTiny < -hi) | (v - kTiny > hi) as one comparison, see find_sdd_both_joint_fixed
      const double v = pr.x * sddw + pr.y * s2;
      const bool viol = fabs(v) - kTiny > va_hi[i];
      bad = bad | viol;
      if (viol) guess = 2 * vv_off[i] + (v > 0.0 ? 1 : 0);
      // velocity row: q'^2 sd2 >= 0 here (a step whose sd2 is negative is never accepted: the
      // step before it is flagged), so only the upper bound can fail
      const double vv = (pr.x * pr.x) * s2;
      bad = bad | (vv - kTiny > vv_hi[i]);
    }
    if (E > 0) {                              // extra rows (every lane: two cheap checks)
      const double v0 = o.ex.x * s2, v1 = o.ex.y * s2;
      bad = bad | (fabs(v0) - kTiny > ex_hi[0]);
      bad = bad | (fabs(v1) - kTiny > ex_hi[E - 1]);
    }
    const int p0 = lane & (L::GRP - 1);
    // the screens below are comparisons only (never results): fused operations are fine
    const double better_than = MAX ? sddw - 2e-6 * fabs(sddw) : sddw + 2e-6 * fabs(sddw);
    const double row_limit = (hi_r + 2.0 * kTiny) * (1.0 + 2e-12);
    const double bs_r = arow.y * s2;
#pragma unroll
    for (int i = 0; i < L::CPL; i++) {       // this lane's other candidates
      const f64x2 own = o.pown[i];
      const double lim_i = v_lim[i];
      // no candidate here (NaN bound), a row with q' ~ 0 (the reference skips it), or the
      // speculated candidate itself (equal, hence not better)
      const bool skip = (lim_i != lim_i) | (fabs(own.x) < kTiny) | (p0 + L::GRP * i == win_cand);
      const double sa = (lim_i - own.y * s2) * o.rc[i];
      // sa -+ 2e-6 |sa| against sddw +- 2e-6 |sddw|
      const bool not_better = MAX ? (__builtin_fma(2e-6, fabs(sa), sa) < better_than)
                                  : (__builtin_fma(-2e-6, fabs(sa), sa) > better_than);
      // |a_r sa + b_r sd2| beyond row r's bound by more than any rounding of either evaluation
      const double ax = arow.x * sa;
      const double va = ax + bs_r;
      const bool violates = __builtin_fma(-2e-6, fabs(ax), fabs(va) * (1.0 - 1e-12)) > row_limit;
      const bool beats = !(skip | not_better | violates);
      bad = bad | beats;
      if (beats) guess = p0 + L::GRP * i;
    }
    return bad;
  }

  // AreDerivativesValid (.cc:624-636): lane j < 2D checks row j. idx is always the extremal's
  // current sample, whose record is in a resident tile.
  __device__ __forceinline__ bool derivs_valid(int idx, double sddv, double s2) const {
    bool bad = false;
    if (lane < 2 * D + E) {
      const bool extra = lane >= 2 * D;
      const int d = extra ? D : ((lane < D) ? lane : lane - D);      // pair index
      const f64x2 pr = record(idx)[d];
      const double A = (lane < D) ? pr.x : 0.0;
      double Bc = (lane < D) ? pr.y : pr.x * pr.x;
      if (E > 0 && extra) Bc = (lane == 2 * D) ? pr.x : pr.y;
      const double v = A * sddv + Bc * s2;
      bad = (v + kTiny < row_lo) || (v - kTiny > row_hi);
    }
    return __ballot(bad) == 0ull;
  }

  // ComputeSddAtIntersection, .cc:722-751: symmetric, forward, backward difference of
  // sd2 in that order; the first admissible one wins, else 0.
  __device__ __forceinline__ void sdd_at_intersection(int index) {
    const double s2 = sd2[index];
    const bool has_next = index < N - 1, has_prev = index > 0;
    double res = 0.0;
    bool done = false;
    if (has_next && has_prev) {
      const double c = 0.25 / ds * (sd2[index + 1] - sd2[index - 1]);
      if (derivs_valid(index, c, s2)) { res = c; done = true; }
    }
    if (!done && has_next) {
      const double c = 0.5 / ds * (sd2[index + 1] - s2);
      if (derivs_valid(index, c, s2)) { res = c; done = true; }
    }
    if (!done && has_prev) {
      const double c = 0.5 / ds * (s2 - sd2[index - 1]);
      if (derivs_valid(index, c, s2)) { res = c; done = true; }
    }
    put_sdd(index, res);
    end_idx = index;
  }

  // AddForwardExtremal (.cc:769-857) for FWD, AddBackwardExtremal (.cc:859-952)
  // otherwise. "n" = the neighbour the extremal moves to (idx+1 or idx-1).
  // Loop-carried scalars of an extremal (all wave-uniform).
  struct Carry {
    int idx;        // current sample
    double cur;     // sd2_[idx]
    double m_i;     // sd2_max[idx]
    double m_n;     // sd2_max[idx + dir]
    double nxt;     // sd2_[idx + dir] as left by earlier extremals (NaN: not visited)
    int t_i, t_n;   // type[idx], type[idx + dir]
    int win;        // lane of the candidate the last FindSdd step selected (-1: none)
  };
  static constexpr int kContinue = -2;

  // One step of AddForwardExtremal (.cc:774-855) / AddBackwardExtremal (.cc:864-950): uses
  // the rows in `use`, stages the next step's rows into `stage`. Returns kContinue or the
  // extremal's return value. pair_signal: see add_extremal.
  // pair_wait (two-wave kernel, first step of a forward extremal): the partner's backward
  // extremal of the same switching point is running its first step, the only one that touches
  // samples this extremal touches: it may write sd2[icrit-1] and sdd[icrit] and may read
  // sd2[icrit+1] (see add_extremal). This step therefore runs up to the point where it would
  // write (sd2[icrit+1], sdd[icrit]) or read sd2[icrit-1] (intersection, negative sd2) and takes
  // the partner's release barrier there, not before it starts: the two first steps overlap.
  template <bool FWD>
  __device__ __forceinline__ int extremal_step(Carry &c, const Rows &use, Rows &stage, Prefetch &pf,
                                               int idx_start, bool &pair_signal, bool &pair_wait) {
    constexpr int dir = FWD ? 1 : -1;
#define TPAMD_PAIR_SIGNAL()                                   \
  do {                                                        \
    if (pair_signal) {                                        \
      __threadfence_block();                                  \
      __syncthreads();                                        \
      pair_signal = false;                                    \
    }                                                         \
  } while (0)
#define TPAMD_PAIR_WAIT()                                     \
  do {                                                        \
    if (FWD && pair_wait) {                                   \
      __syncthreads();                                        \
      pair_wait = false;                                      \
    }                                                         \
  } while (0)
    const int idx = c.idx;
    const int nidx = idx + dir;
    const bool more = FWD ? (nidx < N - 2) : (nidx > 1);
    // stage the next step's data (1 <= nidx <= N-2, 0 <= nidx+dir <= N-1); a new tile can
    // only be entered at a tile edge
    if (((nidx + dir) & (kTile - 1)) == (FWD ? 0 : kTile - 1))
      ensure_tile(nidx + dir, dir, pf);
    load_rows(nidx, stage);
    double m_nn;
    int t_nn;
    load_mt(nidx + dir, m_nn, t_nn);
    const double nxt_nn = sd2[nidx + dir];   // this step only writes sd2[nidx]
    const double cur = c.cur, m_i = c.m_i, m_n = c.m_n, nxt = c.nxt;
    const bool on_boundary = is_tiny(cur - m_i);
    double sd2tmp, sddtmp;
    if (on_boundary && (c.t_i & kBndTrajectory) && (c.t_n & kBndTrajectory)) {
      sd2tmp = m_n;
      sddtmp = FWD ? 0.5 * (sd2tmp - cur) / ds : 0.5 * (cur - sd2tmp) / ds;
      c.win = -1;
    } else {
      TPAMD_CNT(10);
      int win;
      sddtmp = uniform_f64(find_sdd<FWD>(use, cur, win));
      c.win = uniform_i32(win);
      sd2tmp = FWD ? cur + two_ds * sddtmp : cur - two_ds * sddtmp;
    }
    if (!isnan(nxt) && (nxt < sd2tmp)) {
      TPAMD_PAIR_WAIT();
      sdd_at_intersection(idx);
      TPAMD_PAIR_SIGNAL();
      return FWD ? N - 1 : 0;
    }
    if (sd2tmp > m_n) {
      const double sdd_bound = FWD ? 0.5 * (m_n - cur) / ds : 0.5 * (cur - m_n) / ds;
      const bool deriv_invalid = !derivs_valid(idx, sdd_bound, FWD ? m_i : cur);
      const bool type_invalid = c.t_n & (FWD ? kBndSink : kBndSource);
      const bool stop = FWD ? (type_invalid || deriv_invalid)
                            : ((type_invalid || deriv_invalid) && !(idx_start != (N - 1)));
      if (stop) {
        end_idx = idx;
        TPAMD_PAIR_SIGNAL();
        TPAMD_PAIR_WAIT();
        return idx;
      }
      sd2tmp = m_n;
      sddtmp = sdd_bound;
      c.win = -1;
    }
    if (sd2tmp < 0) {
      TPAMD_PAIR_WAIT();
      c.win = -1;
      sd2tmp = 0.0;
      if (FWD) {
        if (idx <= 1) sddtmp = 0.0; else sddtmp = -sd2[idx - 1] / ds;
      } else {
        if (idx < N - 1) sddtmp = sd2[idx + 1] / ds; else sddtmp = 0.0;
      }
    }
    TPAMD_PAIR_WAIT();
    put_sd2(nidx, sd2tmp);
    put_sdd(idx, sddtmp);
    if (!more) end_idx = nidx;
    TPAMD_PAIR_SIGNAL();
    if (!more) return FWD ? N - 1 : 0;
    c.idx = nidx;
    c.cur = sd2tmp;
    c.m_i = m_n;
    c.t_i = c.t_n;
    c.m_n = uniform_f64(m_nn);
    c.t_n = uniform_i32(t_nn);
    c.nxt = uniform_f64(nxt_nn);
    return kContinue;
#undef TPAMD_PAIR_SIGNAL
#undef TPAMD_PAIR_WAIT
  }

  // Boundary following, 64 steps at a time. While an extremal rides the boundary curve
  // (sd2[idx] on the curve, type[idx] and type[idx+dir] both "trajectory",
  // .cc:778-787 / :868-877) every step just copies the curve: sd2[idx+dir] = sd2_max[idx+dir],
  // sdd[idx] = 0.5*(difference)/ds, and the next sample is on the curve again by
  // construction. Whether step k of such a run happens depends only on data known
  // beforehand (types, the curve, sd2 values left by earlier extremals), so lane k
  // evaluates step k: the run is the leading block of eligible lanes. Steps that the
  // scalar code would treat specially (intersection with an earlier extremal, negative
  // curve value, loop end) are not eligible and fall to the scalar step. Returns the
  // number of steps taken (0..64) and advances c.idx / c.cur.
  template <bool FWD>
  __device__ __forceinline__ int follow_boundary(Carry &c) {
    constexpr int dir = FWD ? 1 : -1;
    const int j = c.idx + dir * lane;            // this lane's step: j -> j + dir
    const bool in_loop = FWD ? (j < N - 2) : (j > 1);
    bool elig = false;
    double m_jn = 0.0, m_j = 0.0;
    // The curve and the types come from the records in the two resident LDS tiles (no global
    // latency); a step whose samples lie beyond them simply ends this call's run -- the caller
    // re-centres the tiles and comes back.
    const int jn = j + dir;
    const int tj = max(j, 0) / kTile, tjn = max(jn, 0) / kTile;
    const bool resident = (tj == tag0 || tj == tag1) && (tjn == tag0 || tjn == tag1);
    if (in_loop && j >= 0 && j < N && resident) {
      const f64x2 mt_j = record(j)[kMt], mt_n = record(jn)[kMt];
      m_j = mt_j.x;
      m_jn = mt_n.x;
      const double nxt = sd2[jn];
      elig = (__double2loint(mt_j.y) & kBndTrajectory) && (__double2loint(mt_n.y) & kBndTrajectory) &&
             !(!isnan(nxt) && (nxt < m_jn)) && !(m_jn < 0);
    }
    const unsigned long long mask = __ballot(elig);
    const int L = (~mask == 0ull) ? 64 : (__ffsll((long long)~mask) - 1);
    if (L == 0) return 0;
    const double cur_k = (lane == 0) ? c.cur : m_j;
    const double sddv = FWD ? 0.5 * (m_jn - cur_k) / ds : 0.5 * (cur_k - m_jn) / ds;
    if (lane < L) {
      sd2[j + dir] = m_jn;
      sdd_g[j] = sddv;
    }
    wave_lds_sync();
    c.idx += dir * L;
    c.cur = readlane_f64(m_jn, L - 1);
    return L;
  }

  // Active-constraint speculation, 16 steps at a time. Along an extremal the candidate that
  // FindSddMax/Min selects (one row, one bound) usually stays the same for many steps. If it
  // does, the extremal is the scalar recurrence
  //     sdd_k = (lim - q''_r[j_k] * sd2_k) / q'_r[j_k],   sd2_{k+1} = sd2_k +- 2 ds sdd_k
  // (the very operations of the selected candidate in find_sdd and of .cc:794 / :884). The
  // wave first runs that recurrence for 16 steps (every lane the same arithmetic), then
  // VERIFIES all 16 steps at once, four lanes per step: the exact FindSddMax/Min of sample j_k
  // at sd2_k must return the bits of the speculated sdd_k, and none of the special cases of
  // the scalar step may apply (boundary following .cc:778 / :868, intersection .cc:797 / :887,
  // limit curve exceeded .cc:806 / :896, negative sd2 .cc:839 / :934, loop end). sd2_k is exact
  // if steps 0..k-1 verified, so the leading block of verified steps is exactly what the
  // scalar steps would have produced; it is committed, anything after it is discarded and
  // redone by the scalar step. Returns the number of steps taken (0..16).
  // next_win: where a block ended short of K steps because the active constraint changed, the lane
  // (find_sdd layout) of the candidate that took over at the first unverified step, else -1.
  // curve: set when the first step that did not stand failed for ONE reason only: its FindSdd result
  // is verified (it is what the scalar step would compute) but the new sd2 exceeds the limit curve
  // (.cc:806 / :896). c.m_i / c.m_n / c.t_n then hold that step's boundary data and the caller goes
  // on with leave_on_curve instead of a scalar step that would find the same sdd again.
  template <bool FWD>
  __device__ __forceinline__ int follow_chain(Carry &c, Prefetch &pf, int &next_win, bool &curve) {
    constexpr int dir = FWD ? 1 : -1;
    constexpr int K = L::kChain;
    next_win = -1;
    curve = false;
    const int wl = c.win;
    const int r = (wl / L::PARTS) >> 1;                  // row of the speculated candidate
    const double alim = readlane_f64(lim, wl);           // its bound
    const int j0 = c.idx;
    int jl = j0 + dir * K;                               // furthest sample touched
    jl = FWD ? min(jl, N - 1) : max(jl, 0);
    ensure_tile(j0, dir, pf);
    ensure_tile(jl, dir, pf);
    constexpr int G = L::GRP;
    const int k = lane / G;
    int j = j0 + dir * k;                                // this group's sample
    const bool in_loop = FWD ? (j < N - 2) : (j > 1);
    j = min(max(j, 0), N - 1);
    const int jn = min(max(j + dir, 0), N - 1);
    const f64x2 arow = record(j)[r];
    const double hi_r = readlane_f64(row_hi, r);
    // everything the verification reads is requested now, so that it arrives while the chain runs
    ChainOperands ops;
    load_chain_operands(j, ops);
    const f64x2 mt_j = record(j)[kMt], mt_n = record(jn)[kMt];
    const double nxt = sd2[jn];
    // The recurrence. Its division is done with a reciprocal refined per lane beforehand
    // (the denominator-only part of the IEEE division sequence: v_rcp_f64 + two Newton
    // steps), so that one step is 7 dependent operations. Whether that quotient is the
    // correctly rounded one is NOT assumed: every quad recomputes its step with the real
    // division below and the chain value must match it bit for bit.
    double y = __builtin_amdgcn_rcp(arow.x);
    {
      double e = __builtin_fma(-arow.x, y, 1.0);
      y = __builtin_fma(y, e, y);
      e = __builtin_fma(-arow.x, y, 1.0);
      y = __builtin_fma(y, e, y);
    }
    // The recurrence runs as a pipeline over the lanes instead of sixteen times the same
    // arithmetic in every lane: in each round every step's lanes apply THEIR step (own q', q'',
    // reciprocal: no cross-lane operand traffic) to the value in front of them, and the results
    // move one step (four lanes) up: row_shr:4 inside a 16-lane row, row_bcast:15 into lanes 0-3
    // of the next row. Step 0's input is the extremal's current sd2 and never changes, so after
    // round k the input and output of step k are final and stay so; after 16 rounds every step
    // holds the sd2 it starts from (x) and the one the chain produced for it (y).
    static_assert(G == 4 && K == 16, "the lane pipeline below assumes four lanes per step and 16-lane DPP rows");
    double x = c.cur, y_out = 0.0;
#pragma unroll
    for (int s = 0; s < K; s++) {
      const double n = alim - arow.y * x;
      const double q0 = n * y;
      const double rem = __builtin_fma(-arow.x, q0, n);
      const double q = __builtin_fma(rem, y, q0);
      y_out = FWD ? x + two_ds * q : x - two_ds * q;
      if (s + 1 < K) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        const int ylo = __double2loint(y_out), yhi = __double2hiint(y_out);
        lo = __builtin_amdgcn_update_dpp(lo, ylo, 0x114, 0xf, 0xf, false);    // row_shr:4
        hi = __builtin_amdgcn_update_dpp(hi, yhi, 0x114, 0xf, 0xf, false);
        // into the next 16-lane row only when a row's last step has just become final (rounds 3,
        // 7, 11): lanes 0-3 of a row keep what they have in between
        if ((s & 3) == 3) {
          lo = __builtin_amdgcn_update_dpp(lo, ylo, 0x142, 0xe, 0x1, false);  // row_bcast:15 -> lanes 0-3 of rows 1-3
          hi = __builtin_amdgcn_update_dpp(hi, yhi, 0x142, 0xe, 0x1, false);
        }
        x = __hiloint2double(hi, lo);
      }
    }
    const double my_cur = x;
    // every quad redoes its own step from its sd2 with the reference's operations
    const double my_sdd = (alim - arow.y * my_cur) / arow.x;
    const double my_new = FWD ? my_cur + two_ds * my_sdd : my_cur - two_ds * my_sdd;
    const double chain_next = y_out;                     // what the chain fed to step k+1
    const bool bits_equal = __double_as_longlong(chain_next) == __double_as_longlong(my_new);
    int guess = -1;
    const bool bad = chain_step_bad<FWD>(ops, wl / L::PARTS, hi_r, arow, my_cur, my_sdd, guess);
    const int t_j = __double2loint(mt_j.y), t_n = __double2loint(mt_n.y);
    const double m_j = mt_j.x, m_n = mt_n.x;
    const bool riding = is_tiny(my_cur - m_j) & ((t_j & kBndTrajectory) != 0) & ((t_n & kBndTrajectory) != 0);
    const bool special = riding | (!isnan(nxt) & (nxt < my_new)) | (my_new < 0) | isnan(my_new);
    // a step stands if none of its lanes objects: two ballots (the limit curve apart, see `curve`),
    // folded over the lanes of a step
    const unsigned long long fo = __ballot(bad | !in_loop | !bits_equal | special);
    unsigned long long fm = fo | __ballot(my_new > m_n);
    if (G >= 2) fm |= fm >> 1;
    if (G >= 4) fm |= fm >> 2;
    const unsigned long long miss = fm & L::kPart0;     // bit 4k: step k failed
    const int Lc = miss ? ((__ffsll((long long)miss) - 1) / G) : K;
#ifdef TPAMD_DIAG_REASONS
    if (Lc < K) {
      const unsigned long long lanes_ = ((1ull << G) - 1ull) << (G * Lc);
      const bool r_loop = (__ballot(!in_loop) & lanes_) != 0, r_eq = (__ballot(!bits_equal) & lanes_) != 0;
      const bool r_ride = (__ballot(riding) & lanes_) != 0, r_isect = (__ballot(!isnan(nxt) & (nxt < my_new)) & lanes_) != 0;
      const bool r_curve = (__ballot(my_new > m_n) & lanes_) != 0, r_neg = (__ballot((my_new < 0) | isnan(my_new)) & lanes_) != 0;
      const bool r_bad = (__ballot(bad) & lanes_) != 0, r_hint = (__ballot(guess >= 0) & lanes_) != 0;
      if (r_loop) diag[19]++; else if (r_ride) diag[20]++; else if (r_isect) diag[21]++; else if (r_curve) diag[22]++;
      else if (r_neg) diag[23]++; else if (r_eq) diag[31]++; else if (r_bad && r_hint) diag[13]++; else if (r_bad) diag[12]++;
      if (Lc == 0) diag[17]++;
    }
#endif
    if (TPAMD_CHAIN_GUESS && Lc > 0 && Lc < K) {
      // Why did step Lc fail? If only because another constraint became active (no special case of
      // the scalar step, the chain value itself was right), its lanes know which one: the next
      // block can start from it without a scalar FindSdd step in between.
      const unsigned long long other = __ballot(!in_loop | !bits_equal | special | (my_new > m_n));
      const unsigned long long hints = __ballot(guess >= 0);
      const unsigned long long lanes = ((1ull << G) - 1ull) << (G * Lc);
      if ((other & lanes) == 0ull && (hints & lanes) != 0ull) {
        const int src = __ffsll((long long)(hints & lanes)) - 1;
        next_win = __builtin_amdgcn_readlane(guess, src) * L::PARTS;
      }
    }
    if (TPAMD_CURVE_EXIT && Lc < K) {
      // (my_new > m_n is then the reason: a step fails for one of the terms of `fm`)
      const unsigned long long lanes = ((1ull << G) - 1ull) << (G * Lc);
      if ((fo & lanes) == 0ull) {
        curve = true;
        c.m_i = readlane_f64(m_j, G * Lc);
        c.m_n = readlane_f64(m_n, G * Lc);
        c.t_n = __builtin_amdgcn_readlane(t_n, G * Lc);
      }
    }
    if (Lc == 0) return 0;
    if ((lane & (G - 1)) == 0 && k < Lc) {
      sd2[j + dir] = my_new;
      sdd_g[j] = my_sdd;
    }
    wave_lds_sync();
    c.idx = j0 + dir * Lc;
    c.cur = readlane_f64(my_new, G * (Lc - 1));
    return Lc;
  }

  // The rest of a step whose FindSdd result a chain block verified and whose new sd2 lies above the
  // limit curve (extremal_step from `if (sd2tmp > m_n)` on, .cc:806-855 / :896-950; neither the
  // boundary-following case nor the intersection test applied). c.idx / c.cur: the step's sample and
  // sd2; c.m_i, c.m_n, c.t_n as follow_chain left them.
  template <bool FWD>
  __device__ __forceinline__ int leave_on_curve(Carry &c, int idx_start) {
    constexpr int dir = FWD ? 1 : -1;
    const int idx = c.idx, nidx = idx + dir;
    const bool more = FWD ? (nidx < N - 2) : (nidx > 1);
    const double cur = c.cur, m_n = c.m_n;
    const double sdd_bound = FWD ? 0.5 * (m_n - cur) / ds : 0.5 * (cur - m_n) / ds;
    const bool deriv_invalid = !derivs_valid(idx, sdd_bound, FWD ? c.m_i : cur);
    const bool type_invalid = c.t_n & (FWD ? kBndSink : kBndSource);
    const bool stop = FWD ? (type_invalid || deriv_invalid)
                          : ((type_invalid || deriv_invalid) && !(idx_start != (N - 1)));
    if (stop) {
      end_idx = idx;
      return idx;
    }
    double sd2tmp = m_n, sddtmp = sdd_bound;
    c.win = -1;
    if (sd2tmp < 0) {
      sd2tmp = 0.0;
      if (FWD) {
        if (idx <= 1) sddtmp = 0.0; else sddtmp = -sd2[idx - 1] / ds;
      } else {
        if (idx < N - 1) sddtmp = sd2[idx + 1] / ds; else sddtmp = 0.0;
      }
    }
    put_sd2(nidx, sd2tmp);
    put_sdd(idx, sddtmp);
    if (!more) {
      end_idx = nidx;
      return FWD ? N - 1 : 0;
    }
    c.idx = nidx;
    c.cur = sd2tmp;
    return kContinue;
  }

  // (Re)start the carried state at sample idx: tiles, rows, neighbours.
  template <bool FWD>
  __device__ __forceinline__ void init_carry(int idx, Carry &c, Rows &rows, Prefetch &pf, bool load_cur) {
    constexpr int dir = FWD ? 1 : -1;
    TPAMD_T0(ti_);
    TPAMD_CNT(29);
    c.idx = idx;
    ensure_tile(idx, dir, pf);
    ensure_tile(idx + dir, dir, pf);
    load_rows(idx, rows);
    if (load_cur) c.cur = uniform_f64(sd2[idx]);
    double m0, m1;
    int t0, t1;
    load_mt(idx, m0, t0);
    load_mt(idx + dir, m1, t1);
    c.m_i = uniform_f64(m0); c.m_n = uniform_f64(m1);
    c.t_i = uniform_i32(t0); c.t_n = uniform_i32(t1);
    c.nxt = uniform_f64(sd2[idx + dir]);
    c.win = -1;
    TPAMD_ACC(26, ti_);
  }

  // AddForwardExtremal (.cc:769-857) for FWD, AddBackwardExtremal (.cc:859-952) otherwise.
  // pair_signal (two-wave kernel, backward extremal only): release the partner wave, which
  // runs the forward extremal of the same switching point, once this extremal's FIRST step
  // is complete. Only that step can touch what the partner reads (sd2[icrit-1], sdd[icrit]);
  // afterwards the two extremals work on disjoint index ranges.
  // The scalar loop is unrolled by two so that the two row sets swap roles without copies.
  template <bool FWD>
  // `pf`: tile prefetch registers, possibly already holding the start tile (see the kernel).
  // wait_pair (two-wave kernel, forward extremal only): the partner's release barrier is
  // taken after this extremal's own set-up (tile fill, row loads), which touches nothing the
  // partner's first step writes.
  __device__ __forceinline__ int add_extremal(int idx_start, Prefetch &pf, bool pair_signal = false,
                                              bool wait_pair = false) {
#define TPAMD_PAIR_SIGNAL()                                   \
  do {                                                        \
    if (pair_signal) {                                        \
      __threadfence_block();                                  \
      __syncthreads();                                        \
      pair_signal = false;                                    \
    }                                                         \
  } while (0)
    if (FWD ? !(idx_start < N - 2) : !(idx_start > 1)) {
      end_idx = idx_start;
      TPAMD_PAIR_SIGNAL();
      if (wait_pair) __syncthreads();
      return FWD ? N - 1 : 0;
    }
    Rows rows_a, rows_b;
    Carry c;
    init_carry<FWD>(idx_start, c, rows_a, pf, true);
    // An extremal that starts on the boundary curve writes a whole run at once: wait first. Else
    // the first step takes the barrier itself, as late as it can (extremal_step).
    if (wait_pair && is_tiny(c.cur - c.m_i) && (c.t_i & kBndTrajectory) && (c.t_n & kBndTrajectory)) {
      __syncthreads();
      wait_pair = false;
    }
    bool trust = true;
    int last_win = -1;
    for (;;) {
#define TPAMD_TRY_FOLLOW()                                                                   \
  if (is_tiny(c.cur - c.m_i) && (c.t_i & kBndTrajectory) && (c.t_n & kBndTrajectory)) {      \
    TPAMD_T0(tfb_);                                                                          \
    const int run = uniform_i32(follow_boundary<FWD>(c));                                    \
    TPAMD_ACC(25, tfb_);                                                                     \
    if (run > 0) {                                                                           \
      TPAMD_CNT(8);                                                                          \
      TPAMD_PAIR_SIGNAL();                                                                   \
      if (FWD ? !(c.idx < N - 2) : !(c.idx > 1)) {                                           \
        end_idx = c.idx;                                                                     \
        return FWD ? N - 1 : 0;                                                              \
      }                                                                                      \
      init_carry<FWD>(c.idx, c, rows_a, pf, false);                                          \
      continue;                                                                              \
    }                                                                                        \
  }
// After a FindSdd step selected a candidate: speculate on it (follow_chain) as long as whole
// blocks verify. `trust` drops after a block that verified nothing; it takes two scalar
// steps selecting the same candidate to try again.
#define TPAMD_TRY_CHAIN()                                                                    \
  if (c.win >= 0 && (trust || c.win == last_win)) {                                          \
    int total = 0, run, next_win;                                                            \
    bool again, curve;                                                                       \
    do {                                                                                     \
      { TPAMD_T0(tc_); run = uniform_i32(follow_chain<FWD>(c, pf, next_win, curve)); TPAMD_ACC(4, tc_); } \
      TPAMD_CNT(6);                                                                          \
      total += run;                                                                          \
      again = run == L::kChain;                                                              \
      /* the active constraint changed inside the block: go on with the one that took over */ \
      if (run > 0 && run < L::kChain && next_win >= 0) { c.win = uniform_i32(next_win); again = true; } \
    } while (again && (FWD ? (c.idx < N - 2) : (c.idx > 1)));                                \
    trust = total > 0;                                                                       \
    last_win = -1;                                                                           \
    TPAMD_ADD(14, total);                                                                    \
    if (curve) {                                                                             \
      /* the block verified this step's FindSdd and found the new sd2 above the limit curve */ \
      TPAMD_CNT(30);                                                                         \
      trust = true;                                                                          \
      const int rc = leave_on_curve<FWD>(c, idx_start);                                      \
      if (rc != kContinue) return rc;                                                        \
      total = 1;   /* (the extremal moved: re-initialise below, as after any block) */       \
    }                                                                                        \
    if (total > 0) {                                                                         \
      if (FWD ? !(c.idx < N - 2) : !(c.idx > 1)) {                                           \
        end_idx = c.idx;                                                                     \
        return FWD ? N - 1 : 0;                                                              \
      }                                                                                      \
      init_carry<FWD>(c.idx, c, rows_a, pf, false);                                          \
      continue;                                                                              \
    }                                                                                        \
  } else {                                                                                   \
    last_win = c.win;                                                                        \
  }
      TPAMD_TRY_FOLLOW();
      TPAMD_T0(ts1_);
      int r = extremal_step<FWD>(c, rows_a, rows_b, pf, idx_start, pair_signal, wait_pair);
      TPAMD_ACC(24, ts1_);
      if (r != kContinue) return r;
      TPAMD_TRY_CHAIN();
      if (is_tiny(c.cur - c.m_i) && (c.t_i & kBndTrajectory) && (c.t_n & kBndTrajectory)) {
        // a run starts here: restart the loop (it re-stages into rows_a)
        load_rows(c.idx, rows_a);
        continue;
      }
      TPAMD_T0(ts2_);
      r = extremal_step<FWD>(c, rows_b, rows_a, pf, idx_start, pair_signal, wait_pair);
      TPAMD_ACC(24, ts2_);
      if (r != kContinue) return r;
      TPAMD_TRY_CHAIN();
    }
#undef TPAMD_TRY_CHAIN
#undef TPAMD_TRY_FOLLOW
#undef TPAMD_PAIR_SIGNAL
  }

  // NextCriticalPoint, .cc:697-720. The reference walks idx = lo+1 .. hi: the first sample
  // classified source or trajectory becomes the candidate c0; after that every sample with
  // sd2_max[idx] == sd2_max_for_sdd0[0] (sic, index 0, .cc:710; cached as bit kBndEqualsZ00
  // of the type byte) replaces it; the walk returns at the first idx >= c0 whose sd2 is
  // already set. Inside the switching-point loop that idx is always hi itself: sd2 is set
  // on [0, icrit_lo] (forward extremals and the backward extremals that connected to them)
  // and on [icrit_hi, N-1] (the first backward extremal), and nothing in between has been
  // visited -- the forward extremal of a loop either stops below icrit_hi, leaving the next
  // (icrit_lo, icrit_hi) untouched, or reaches the upper region, which ends the loop
  // (.cc:337-341), and icrit_hi itself never moves after the first pair. So the answer is
  // the last flagged idx in (c0, hi] if there is one, else c0, and since hi is fixed the
  // last flagged idx <= hi (`zlast`) is found once per path (last_flagged_below).
  // Each pass of the scan loop covers 256 samples (four 64-sample ballots whose LDS reads are
  // issued together); successive searches scan disjoint ranges.
  __device__ __forceinline__ int next_critical_point(int idx_lo, int idx_hi, int zlast) const {
    int c0 = -1;
    for (int base = idx_lo + 1; base <= idx_hi && c0 < 0; base += 256) {
      uint8_t ty[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ty[u] = typel[min(base + 64 * u + lane, N - 1)];   // unconditional loads
      unsigned long long mask[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = base + 64 * u + lane;
        mask[u] = __ballot((idx <= idx_hi) & ((ty[u] & (kBndSource | kBndTrajectory)) != 0));
      }
#pragma unroll
      for (int u = 3; u >= 0; u--)
        if (mask[u]) c0 = base + 64 * u + __ffsll((long long)mask[u]) - 1;
    }
    if (c0 < 0) return -1;
    return (zlast > c0) ? zlast : c0;
  }

  // Largest idx in [1, idx_hi] whose type byte carries kBndEqualsZ00, or -1.
  __device__ __forceinline__ int last_flagged_below(int idx_hi) const {
    for (int top = idx_hi; top >= 1; top -= 256) {
      uint8_t ty[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ty[u] = typel[max(top - 64 * u - lane, 0)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = top - 64 * u - lane;
        const unsigned long long m = __ballot((idx >= 1) & ((ty[u] & kBndEqualsZ00) != 0));
        if (m) return top - 64 * u - (__ffsll((long long)m) - 1);
      }
    }
    return -1;
  }

#ifdef TPAMD_DIAG
  // The literal walk (first set sd2 found by scanning), to cross-check the shortcut above.
  __device__ __forceinline__ int next_critical_point_literal(int idx_lo, int idx_hi) const {
    int crit = -1;
    for (int idx = idx_lo + 1; idx <= idx_hi; idx++) {
      if (crit < 0) {
        if (typel[idx] & (kBndSource | kBndTrajectory)) crit = idx;
      } else if (typel[idx] & kBndEqualsZ00) {
        crit = idx;
      }
      if (crit > 0 && !isnan(sd2[idx])) return crit;
    }
    return -1;
  }
#endif
};

// ---------------------------------------------------------------------------------------
// CalculateBoundary passes 2-4 (time_optimal_path_timing.cc:1381-1484) for ONE path, run by
// the 128 threads of the path's sweep workgroup before the sweep starts. Same arithmetic as
// k_boundary_zfit / k_boundary_detect / k_boundary_final (tpamd_kernels.h, which stay in use
// for the generic kernels; the derivations of what the reference's in-place passes amount to
// per sample are in the comments there) -- but per path instead of per 256 samples of the
// whole batch: the pass-2/3 flags live in LDS, only the sparse re-fit values travel through
// global memory, and three launches with their round trips through HBM disappear.
// Every thread handles four samples per trip with all their loads issued up front (the
// values a sample does not need are loaded and dropped: valid addresses, no dependence on
// them); the sparse FindSddMax/Min re-fits are collected per wave and run four at a time with
// the rows of the next four already in flight.
//   at_l, ff_l: LDS bytes [N] (sd2_max_at_sdd0 flags of pass 1; deferred-fix flags of pass 2)
//   scratch:    LDS, kRefitLdsBytes per wave (re-fit lists)
//   typel:      LDS bytes [N], receives the final classification (incl. the kBndEqualsZ00 bit)
// keep: also store sdd_max/sdd_min/type to the workspace (tpamd_debug_copy_boundary).
// ---------------------------------------------------------------------------------------
constexpr int kRefitCap = 128;                                  // list entries per wave
constexpr int kRefitLdsBytes = kRefitCap * (4 + 8 + 8);          // idx | value | m_next

template <int D, int E>
struct PathBoundary {
  const JointSource &src;
  const Workspace &ws;
  int b, N, stride, lane;
  size_t pb;
  double ds, z00;
  const uint8_t *at_l;
  uint8_t *ff_l, *typel;
  bool keep;
  // the wave's re-fit list
  int *l_idx;
  double *l_val, *l_nxt;
  int count;

  __device__ __forceinline__ void append(bool need, int j, double val, double nxt) {
    const unsigned long long mask = __ballot(need);
    if (!mask) return;
    if (need) {
      const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
      l_idx[pos] = j;
      l_val[pos] = val;
      l_nxt[pos] = nxt;
    }
    count += __popcll(mask);
  }

  // FindSddMax / FindSddMin (.cc:638-695) of sample j at sd2 by one thread
  __device__ __forceinline__ void refit_one(int j, double sd2v, double *x, double *y) const {
    constexpr int R = 2 * D + E + 2;
    const f64x2 *rec2 = reinterpret_cast<const f64x2 *>(src.q12 + ((size_t)b * stride + j) * R);
    const double *lim_lo = src.lim + (size_t)b * 2 * (2 * D + E), *lim_hi = lim_lo + 2 * D + E;
    double a[D], bq[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      const f64x2 p = rec2[d];
      a[d] = p.x;
      bq[d] = p.y;
    }
    bool ok = true;
    if (E > 0) {
      const f64x2 ex = rec2[D];
      const double vt = ex.x * sd2v, vr = ex.y * sd2v;
      ok = !(vt + kTiny < lim_lo[2 * D] || vt - kTiny > lim_hi[2 * D]) &&
           !(vr + kTiny < lim_lo[2 * D + 1] || vr - kTiny > lim_hi[2 * D + 1]);
    }
    if (ok) find_sdd_both_joint_fixed<D>(a, bq, lim_hi, sd2v, x, y);
  }

  // pass 4 for one sample whose final boundary value and sdd range are known (.cc:1459-1484)
  __device__ __forceinline__ void finalize(int j, double m, double X, double Y, double m_next) const {
    uint8_t type = kBndNone;
    if (j >= 1 && j <= N - 2) {
      const double sd2p = (m_next - m) / ds;
      const double sd2p_min = 2 * Y;
      const double sd2p_max = 2 * X;
      if (sd2p < sd2p_min) type = kBndSink;
      else if (sd2p > sd2p_max) type = kBndSource;
      if ((sd2p <= sd2p_max) && (sd2p >= sd2p_min)) type = kBndTrajectory;
    }
    if (m == z00) type |= kBndEqualsZ00;   // cached comparison of NextCriticalPoint (.cc:710)
    ws.m[pb + j] = m;
    typel[j] = type;
    if (keep) {
      ws.X[pb + j] = X;
      ws.Y[pb + j] = Y;
      ws.type[pb + j] = type;
    }
  }

  // FindSddMax/Min (.cc:638-695) for every listed sample at its listed sd2. FINAL = false: the
  // results are sd2_max_for_sdd0's sdd range next to isolated points (Xz, Yz); FINAL = true: they
  // complete the sample (finalize).
  template <bool FINAL>
  __device__ __forceinline__ void flush(long long *diag) {
    TPAMD_T0(tf);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = count;
    // one listed sample per lane, all at once: the sample's (q', q'') pairs go to registers and
    // FindSddMax/Min run as in K1 (find_sdd_both_joint_fixed; the Cartesian extra rows have A = 0
    // and only gate the result, as in k_cartesian_lp)
    for (int e = lane; e < n; e += 64) {
      const int j = l_idx[e];
      const double val = l_val[e];
      double x = 0.0, y = 0.0;
      refit_one(j, val, &x, &y);
      if (FINAL) finalize(j, val, x, y, l_nxt[e]);
      else { ws.Xz[pb + j] = x; ws.Yz[pb + j] = y; }
    }
    count = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (FINAL) TPAMD_ACC(23, tf);
  }
};

template <int D, int E>
__device__ __forceinline__ void boundary_passes_for_path(const JointSource &src, const Workspace &ws, int b,
                                                         int N, int stride, int tid, uint8_t *at_l,
                                                         uint8_t *ff_l, char *scratch, uint8_t *typel,
                                                         bool keep, long long *diag) {
  TPAMD_T0C(tp0);
  const size_t pb = (size_t)b * stride;
  const int lane = tid & 63, w = tid >> 6;
  const double *m0 = ws.m0 + pb, *z0 = ws.z0 + pb, *X0 = ws.X0 + pb, *Y0 = ws.Y0 + pb;
  const double *Xz = ws.Xz + pb, *Yz = ws.Yz + pb, *fv = ws.fix_val + pb;
  PathBoundary<D, E> P{src, ws, b, N, stride, lane, pb, ws.ds[b], z0[0], at_l, ff_l, typel, keep,
                         nullptr, nullptr, nullptr, 0};
  {
    char *mine = scratch + (size_t)w * kRefitLdsBytes;
    P.l_val = reinterpret_cast<double *>(mine);
    P.l_nxt = P.l_val + kRefitCap;
    P.l_idx = reinterpret_cast<int *>(P.l_nxt + kRefitCap);
  }
  const double ds = P.ds;
  constexpr int U = 4;
  const int hi_i = N - 1;
  auto cl = [hi_i](int i) { return min(max(i, 0), hi_i); };     // clamp an index into the path

  for (int i = tid; i < N; i += 128) at_l[i] = ws.at0[pb + i];
  __syncthreads();
  // "isolated point" (iso_at) of every sample as a byte of its own, formed without branches: the
  // passes below then read a handful of bytes per sample, all loads issued together, instead of
  // walking through short-circuit conditions one LDS round trip at a time. iso_l[0] and
  // iso_l[N-1] are 0 by definition, so reads clamped into the path need no range checks.
  uint8_t *iso_l = ff_l + (ff_l - at_l);
  for (int i = tid; i < N; i += 128) {
    const int a0 = at_l[cl(i - 1)], a1 = at_l[i], a2 = at_l[cl(i + 1)];
    iso_l[i] = (uint8_t)((i >= 1) & (i <= N - 2) & (a0 == 0) & (a1 != 0) & (a2 == 0));
  }
  __syncthreads();
  TPAMD_ACCC(19, tp0);

#ifndef TPAMD_BOUNDARY_FAST
#define TPAMD_BOUNDARY_FAST 1
#endif
  // Passes 2 (second half), 3 and 4 from ONE round of loads per segment of 2108 samples, with no
  // load, branch or cross-lane operation inside the per-sample arithmetic. The path is cut into
  // chunks of 62 samples; a wave takes every other chunk and loads it with one sample of overlap
  // on either side (lane l of chunk c holds sample 62c + l - 1, lanes 1..62 own theirs), so a
  // neighbour's value is always the next lane's register (DPP wave shift). A sample with no
  // isolated point and no deferred fix within reach -- about nineteen in twenty -- needs nothing
  // but those registers in either pass. The others are listed per wave and handled one per lane
  // from memory, with the reference's case analysis spelled out (detect_one, final_one).
  constexpr int UF = 17;                 // chunks per wave
  // The per-wave lists hold every sample of the wave in the worst case (they are emptied only
  // after the register-resident part, which leaves no room for the re-fit code next to it); they
  // follow the three byte arrays in the LDS that later holds sd2. Very short paths, where that
  // does not fit, take the general code below.
  const int slow_cap = ((N + 127) / 128) * 64 + 64;
  const int flag_bytes = (int)(ff_l - at_l);
  // Paths of more than 2 x 17 chunks are handled segment by segment (the flags of a pass are
  // complete for the whole path before the next pass reads them), every pass loading its segment
  // afresh. A path of one segment keeps the 51 registers from the second pass to the third; doing
  // that inside the segment loops costs 24 VGPRs (220 instead of 196: no room for the sampling/LP
  // kernel beside two sweep waves), so the one-segment form is compiled separately.
  constexpr int kSegChunks = 2 * UF;
  const int nseg = (N + 62 * kSegChunks - 1) / (62 * kSegChunks);
  if (TPAMD_BOUNDARY_FAST && 3 * flag_bytes + 8 * slow_cap <= 8 * N) {
    // kOne: the path is one segment (N <= 2108) -- the segment loops fold away and the registers of
    // the second pass serve the third, as before the segments existed; both forms are compiled.
    auto fast_form = [&](auto one_segment) {
    constexpr bool kOne = decltype(one_segment)::value;
    const int nsg = kOne ? 1 : nseg;
    int *slow = reinterpret_cast<int *>(at_l + 3 * flag_bytes) + w * slow_cap;
    int nslow = 0;
    // the samples whose bit is set in `mask` (bit u: chunk cb + 2u + w), appended to this wave's list
    auto list_samples = [&](unsigned mask, int cb) {
#pragma unroll
      for (int u = 0; u < UF; u++) {
        const bool need = (mask >> u) & 1u;
        const unsigned long long bm = __ballot(need);
        if (bm) {
          if (need) slow[nslow + __popcll(bm & ((1ull << lane) - 1ull))] = 62 * (cb + 2 * u + w) + lane - 1;
          nslow += __popcll(bm);
        }
      }
    };
    // pass 2, first half (.cc:1386-1395): FindSddMax/Min at sd2_max_for_sdd0 next to isolated
    // points -- usually there are none, which one ballot per chunk establishes
    {
      for (int seg = 0; seg < nsg; seg++) {
        const int cb = seg * kSegChunks;
        unsigned zf = 0u;
#pragma unroll
        for (int u = 0; u < UF; u++) {
          const int j = 62 * (cb + 2 * u + w) + lane - 1;
          const int jb = max(j, 1);
          zf |= (unsigned)((lane >= 1) && (lane <= 62) && (j >= 0) && (j < N) &&
                           ((iso_l[jb - 1] | iso_l[jb + 1]) != 0)) << u;
        }
        list_samples(zf, cb);
      }
      JointSweep<D, E>::wave_lds_sync();
      for (int e0 = 0; e0 < nslow; e0 += 64) {
        const bool valid = e0 + lane < nslow;
        const int j = valid ? slow[e0 + lane] : 0;
        // (a stray byte may have listed a sample that is not next to an isolated point: harmless,
        // its Xz / Yz are then simply never read)
        if (P.count + 64 > kRefitCap) P.template flush<false>(diag);
        P.append(valid, j, valid ? z0[j] : 0.0, 0.0);
      }
      nslow = 0;
      P.template flush<false>(diag);
      __threadfence_block();
      __syncthreads();
      TPAMD_ACCC(20, tp0);
    }
    // pass 2, second half for one sample, everything from memory (k_boundary_detect)
    auto detect_one = [&](int k) {
      uint8_t flag = 0;
      if (k >= 1 && k <= N - 2) {
        const bool iso_k = iso_at(at_l, N, k), iso_km1 = iso_at(at_l, N, k - 1),
                   iso_km2 = iso_at(at_l, N, k - 2);
        const bool l_mod = iso_k || iso_km2;
        const double m_l = l_mod ? z0[k - 1] : m0[k - 1];
        const double fsmax_l = l_mod ? Xz[k - 1] : X0[k - 1];
        const double m_c = iso_km1 ? z0[k] : m0[k];
        const double X_c = iso_km1 ? Xz[k] : X0[k];
        const double Y_c = iso_km1 ? Xz[k] : Y0[k];
        const double m_r = iso_k ? z0[k + 1] : m0[k + 1];
        const double Y_r = iso_k ? Xz[k + 1] : Y0[k + 1];
        const double fsmin_r = iso_k ? Yz[k + 1] : Y0[k + 1];
        const double sd2p = (m_r - m_c) / ds;
        const double sd2p_min = 2 * Y_c;
        const double sd2p_max = 2 * X_c;
        const bool sink_or_source = (sd2p < sd2p_min) || (sd2p > sd2p_max);
        const bool skipped_sdd = (X_c > 0) && (Y_r < 0);
        const bool skipped_sd2 = (m_c > m_l - kTiny) && (m_c > m_r - kTiny);
        if ((skipped_sd2 || skipped_sdd) && sink_or_source) {
          const double fw = m_l + 2.0 * ds * fsmax_l;
          const double bw = m_r - 2.0 * ds * fsmin_r;
          double mn = z0[k];
          if (fw < mn) mn = fw;
          if (bw < mn) mn = bw;
          ws.fix_val[pb + k] = (0.0 < mn) ? mn : 0.0;
          flag = 1;
        }
      }
      ff_l[k] = flag;
    };
    auto drain_detect = [&]() {
      JointSweep<D, E>::wave_lds_sync();
      for (int e0 = 0; e0 < nslow; e0 += 64)
        if (e0 + lane < nslow) detect_one(slow[e0 + lane]);
      nslow = 0;
      JointSweep<D, E>::wave_lds_sync();
    };
    double m0c[UF], X0c[UF], Y0c[UF];
    // (indices stay affine in u wherever the chunk lies inside the path: one address register per
    // array and immediate offsets instead of 51 clamped addresses)
    auto load_segment = [&](int cb) {
#pragma unroll
      for (int u = 0; u < UF; u++) {
        const int c = cb + 2 * u + w;
        const int k = 62 * c + lane - 1;
        const int kc = (c >= 1 && 62 * c + 62 < N) ? k : cl(k);        // (uniform choice)
        m0c[u] = m0[kc]; X0c[u] = X0[kc]; Y0c[u] = Y0[kc];
      }
    };
    const bool owner = (lane >= 1) && (lane <= 62);
    for (int seg = 0; seg < nsg; seg++) {
    const int cb = seg * kSegChunks;
    load_segment(cb);
    unsigned listed = 0u;              // bit u: this lane's sample of chunk cb + 2u + w goes to the list
#pragma unroll
    for (int u = 0; u < UF; u++) {
      const int k = 62 * (cb + 2 * u + w) + lane - 1;
      const bool inside = owner && (k >= 1) && (k <= N - 2);
      // unclamped byte reads: a byte from outside [0, N) can only send a sample to the list, which
      // is always right (the reads stay inside this LDS region; lane 0 of the first chunk reads
      // below it and owns nothing)
      const int kb = max(k, 2);
      const bool near_iso = (iso_l[kb] | iso_l[kb - 1] | iso_l[kb - 2]) != 0;
      const double m_c = m0c[u], X_c = X0c[u], Y_c = Y0c[u];
      const double m0l = dpp_f64<0x138>(m_c);                                // wave_shr:1: sample k-1
      const double m0r = dpp_f64<0x130>(m_c), Y0r = dpp_f64<0x130>(Y_c);     // wave_shl:1: sample k+1
      const double sd2p = (m0r - m_c) / ds;
      const double sd2p_min = 2 * Y_c;
      const double sd2p_max = 2 * X_c;
      const bool sink_or_source = (sd2p < sd2p_min) || (sd2p > sd2p_max);
      const bool skipped_sdd = (X_c > 0) && (Y0r < 0);
      const bool skipped_sd2 = (m_c > m0l - kTiny) && (m_c > m0r - kTiny);
      // a deferred fix needs sd2_max_for_sdd0[k] from memory: listed as well
      const bool fix = (skipped_sd2 || skipped_sdd) && sink_or_source;
      const bool to_list = inside && (near_iso || fix);
      if (owner && k >= 0 && k < N && !to_list) ff_l[k] = 0;
      listed |= (unsigned)to_list << u;
    }
    list_samples(listed, cb);
    drain_detect();
    }
    __threadfence_block();
    __syncthreads();
    TPAMD_ACCC(21, tp0);

    // passes 3 and 4 for one sample, everything from memory (k_boundary_final)
    auto final_one = [&](int j, double &m, double &X, double &Y, double &m_next) -> bool {
      bool refit = false;
      m = 0.0; X = 0.0; Y = 0.0; m_next = 0.0;
      const bool f_next = (j + 1 <= N - 2) && ff_l[j + 1];
      const bool f_self = ff_l[j];
      const bool f_prev = (j >= 1) && ff_l[j - 1];
      const bool iso_p = iso_at(at_l, N, j - 1), iso_n = iso_at(at_l, N, j + 1);
      if (f_next || (!f_self && f_prev)) {
        m = z0[j];
        if (iso_p || iso_n) { X = Xz[j]; Y = Yz[j]; }     // pass 2 was here
        else refit = true;
      } else if (f_self) {
        m = fv[j];
        refit = true;
      } else if (iso_n) {
        m = z0[j]; X = Xz[j]; Y = Yz[j];
      } else if (iso_p) {
        m = z0[j]; X = Xz[j]; Y = Xz[j];      // sic, .cc:1394-1395
      } else {
        m = m0[j]; X = X0[j]; Y = Y0[j];
      }
      const int jn = j + 1;                    // final value of element j + 1
      if (jn <= N - 1) {
        if (jn + 1 <= N - 2 && ff_l[jn + 1]) m_next = z0[jn];
        else if (ff_l[jn]) m_next = fv[jn];
        else if (ff_l[jn - 1]) m_next = z0[jn];
        else if (iso_at(at_l, N, jn + 1) || iso_at(at_l, N, jn - 1)) m_next = z0[jn];
        else m_next = m0[jn];
      }
      return refit;
    };
    auto drain_final = [&]() {
      JointSweep<D, E>::wave_lds_sync();
      for (int e0 = 0; e0 < nslow; e0 += 64) {
        const bool valid = e0 + lane < nslow;
        double m = 0.0, X = 0.0, Y = 0.0, m_next = 0.0;
        bool refit = false;
        int j = 0;
        if (valid) {
          j = slow[e0 + lane];
          refit = final_one(j, m, X, Y, m_next);
          if (!refit) P.finalize(j, m, X, Y, m_next);
        }
        if (P.count + 64 > kRefitCap) P.template flush<true>(diag);
        P.append(refit, j, m, m_next);
      }
      nslow = 0;
      JointSweep<D, E>::wave_lds_sync();
    };
    for (int seg = 0; seg < nsg; seg++) {
    const int cb = seg * kSegChunks;
    if (!kOne) load_segment(cb);
    unsigned listed = 0u;
#pragma unroll
    for (int u = 0; u < UF; u++) {
      const int j = 62 * (cb + 2 * u + w) + lane - 1;
      const bool mine = owner && (j >= 0) && (j < N);
      const int jb = max(j, 1);
      // a deferred fix or an isolated point at j-1 .. j+2 (stray bytes only add list entries)
      const bool busy = (ff_l[jb - 1] | ff_l[jb] | ff_l[jb + 1] | ff_l[jb + 2] |
                         iso_l[jb - 1] | iso_l[jb] | iso_l[jb + 1] | iso_l[jb + 2]) != 0;
      const double m0r = dpp_f64<0x130>(m0c[u]);
      if (mine && !busy) P.finalize(j, m0c[u], X0c[u], Y0c[u], (j + 1 <= N - 1) ? m0r : 0.0);
      listed |= (unsigned)(mine && busy) << u;
    }
    list_samples(listed, cb);
    drain_final();
    }
    P.template flush<true>(diag);
    __threadfence_block();
    __syncthreads();
    TPAMD_ACCC(22, tp0);
    };
    if (nseg == 1) fast_form(std::true_type{});
    else fast_form(std::false_type{});
    return;
  }

  // pass 2, first half (.cc:1386-1395): FindSddMax/Min at sd2_max_for_sdd0 next to isolated points
  for (int base = 64 * w; base < N; base += 128) {
    const int j = base + lane;
    const bool need = (j < N) & ((iso_l[cl(j - 1)] | iso_l[cl(j + 1)]) != 0);
    if (P.count + 64 > kRefitCap) P.template flush<false>(diag);
    P.append(need, j, need ? z0[j] : 0.0, 0.0);
  }
  P.template flush<false>(diag);
  __threadfence_block();
  __syncthreads();
  TPAMD_ACCC(20, tp0);

  // pass 2, second half (.cc:1396-1431): skipped maxima -> deferred fixes
  for (int k0 = tid; k0 < N; k0 += 128 * U) {
    double m0l[U], m0c[U], m0r[U], z0l[U], z0c[U], z0r[U], X0l[U], X0c[U], Y0c[U], Y0r[U];
    double Xzl[U], Xzc[U], Xzr[U], Yzr[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + 128 * u;
      const int kl = cl(k - 1), kc = cl(k), kr = cl(k + 1);
      m0l[u] = m0[kl]; m0c[u] = m0[kc]; m0r[u] = m0[kr];
      X0l[u] = X0[kl]; X0c[u] = X0[kc];
      Y0c[u] = Y0[kc]; Y0r[u] = Y0[kr];
      // sd2_max_for_sdd0 and the values re-fitted there matter only next to isolated points
      // (about 3 % of the samples): loaded where needed instead of streaming the arrays
      const bool ik = iso_at(at_l, N, k), ik1 = iso_at(at_l, N, k - 1), ik2 = iso_at(at_l, N, k - 2);
      z0l[u] = (ik || ik2) ? z0[kl] : 0.0;
      z0c[u] = ik1 ? z0[kc] : 0.0;
      z0r[u] = ik ? z0[kr] : 0.0;
      Xzl[u] = (ik || ik2) ? Xz[kl] : 0.0;
      Xzc[u] = ik1 ? Xz[kc] : 0.0;
      Xzr[u] = ik ? Xz[kr] : 0.0;
      Yzr[u] = ik ? Yz[kr] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + 128 * u;
      if (k >= N) continue;
      uint8_t flag = 0;
      if (k >= 1 && k <= N - 2) {
        const bool iso_k = iso_at(at_l, N, k), iso_km1 = iso_at(at_l, N, k - 1),
                   iso_km2 = iso_at(at_l, N, k - 2);
        const bool l_mod = iso_k || iso_km2;
        const double m_l = l_mod ? z0l[u] : m0l[u];
        const double fsmax_l = l_mod ? Xzl[u] : X0l[u];      // FindSddMax(k-1, m_l)
        const double m_c = iso_km1 ? z0c[u] : m0c[u];
        const double X_c = iso_km1 ? Xzc[u] : X0c[u];
        const double Y_c = iso_km1 ? Xzc[u] : Y0c[u];
        const double m_r = iso_k ? z0r[u] : m0r[u];
        const double Y_r = iso_k ? Xzr[u] : Y0r[u];
        const double fsmin_r = iso_k ? Yzr[u] : Y0r[u];      // FindSddMin(k+1, m_r)
        const double sd2p = (m_r - m_c) / ds;
        const double sd2p_min = 2 * Y_c;
        const double sd2p_max = 2 * X_c;
        const bool sink_or_source = (sd2p < sd2p_min) || (sd2p > sd2p_max);
        const bool skipped_sdd = (X_c > 0) && (Y_r < 0);
        const bool skipped_sd2 = (m_c > m_l - kTiny) && (m_c > m_r - kTiny);
        if ((skipped_sd2 || skipped_sdd) && sink_or_source) {
          const double fw = m_l + 2.0 * ds * fsmax_l;  // OneForwardExtremalStep(k-1, m_l), .cc:753-759
          const double bw = m_r - 2.0 * ds * fsmin_r;  // OneBackwardExtremalStep(k+1, m_r), .cc:761-767
          double mn = z0[k];                           // rare: loaded here
          if (fw < mn) mn = fw;
          if (bw < mn) mn = bw;
          ws.fix_val[pb + k] = (0.0 < mn) ? mn : 0.0;
          flag = 1;
        }
      }
      ff_l[k] = flag;
    }
  }
  __threadfence_block();
  __syncthreads();
  TPAMD_ACCC(21, tp0);

  // passes 3 and 4 (.cc:1432-1484): final boundary value, its sdd range, classification.
  // final value of element j+1 (the last writer among the deferred fixes, see k_boundary_final)
  // is formed from unconditionally loaded candidates.
  for (int base = 256 * w; base < N; base += 512) {
    double m0c[U], z0c[U], X0c[U], Y0c[U], Xzc[U], Yzc[U], fvc[U], m0n[U], z0n[U], fvn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = base + 64 * u + lane;
      const int jc = cl(j), jn = cl(j + 1);
      m0c[u] = m0[jc]; z0c[u] = z0[jc]; X0c[u] = X0[jc]; Y0c[u] = Y0[jc];
      m0n[u] = m0[jn]; z0n[u] = z0[jn];
      // sparse: re-fitted values next to isolated points, deferred-fix values where flagged
      const bool near_iso = iso_at(at_l, N, jc - 1) || iso_at(at_l, N, jc + 1);
      Xzc[u] = near_iso ? Xz[jc] : 0.0;
      Yzc[u] = near_iso ? Yz[jc] : 0.0;
      fvc[u] = ff_l[jc] ? fv[jc] : 0.0;
      fvn[u] = ff_l[jn] ? fv[jn] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = base + 64 * u + lane;
      const bool active = j < N;
      double m = 0.0, X = 0.0, Y = 0.0, m_next = 0.0;
      bool refit = false;
      if (active) {
        const bool f_next = (j + 1 <= N - 2) && ff_l[j + 1];
        const bool f_self = ff_l[j];
        const bool f_prev = (j >= 1) && ff_l[j - 1];
        const bool iso_p = iso_at(at_l, N, j - 1), iso_n = iso_at(at_l, N, j + 1);
        if (f_next || (!f_self && f_prev)) {
          m = z0c[u];
          if (iso_p || iso_n) { X = Xzc[u]; Y = Yzc[u]; }     // pass 2 was here
          else refit = true;
        } else if (f_self) {
          m = fvc[u];
          refit = true;
        } else if (iso_n) {
          m = z0c[u]; X = Xzc[u]; Y = Yzc[u];
        } else if (iso_p) {
          m = z0c[u]; X = Xzc[u]; Y = Xzc[u];      // sic, .cc:1394-1395
        } else {
          m = m0c[u]; X = X0c[u]; Y = Y0c[u];
        }
        // final value of element j + 1
        const int jn = j + 1;
        if (jn <= N - 1) {
          if (jn + 1 <= N - 2 && ff_l[jn + 1]) m_next = z0n[u];
          else if (ff_l[jn]) m_next = fvn[u];
          else if (ff_l[jn - 1]) m_next = z0n[u];
          else if (iso_at(at_l, N, jn + 1) || iso_at(at_l, N, jn - 1)) m_next = z0n[u];
          else m_next = m0n[u];
        }
        if (!refit) P.finalize(j, m, X, Y, m_next);
      }
      if (P.count + 64 > kRefitCap) P.template flush<true>(diag);
      P.append(refit, j, m, m_next);
    }
  }
  P.template flush<true>(diag);
  __threadfence_block();
  __syncthreads();
  TPAMD_ACCC(22, tp0);
}

// Dynamic LDS of one path (bytes):
//   sd2[N2]*8 | tile rings 2 waves x 2 slots x 32 x R*8 | type copy N (padded to 16) |
//   exchange words 64 | dirty bitmap | zero-pair bitmap | a_max[16]*8 | reduction scratch 32
// N2 = N rounded up to even (16-byte alignment of the rings). After the switching-point loop
// the rings hold one chunk of dt / time values and the type copy holds the list of samples
// whose qd/qdd are redone (see the tail).
template <int D, int E = 0>
struct SweepLds {
  static constexpr int R = 2 * D + E + 2;
  static constexpr int kRingDoubles = 2 * 2 * TileCfg<D>::kSamples * R;
  __host__ __device__ static constexpr int n2(int N) { return (N + 1) & ~1; }
  __host__ __device__ static constexpr int words(int N) { return ((N + 31) / 32 + 1) & ~1; }
  __host__ __device__ static constexpr int type_bytes(int N) { return ((N + 15) / 16) * 16; }
  __host__ __device__ static constexpr size_t bytes(int N) {
    return (size_t)n2(N) * 8 + (size_t)kRingDoubles * 8 + type_bytes(N) + 64 +
           2 * (size_t)words(N) * 4 + 16 * 8 + 32;
  }
};
template <int D, int E = 0>
__host__ __device__ inline size_t sweep_joint_lds_bytes(int N) {
  return SweepLds<D, E>::bytes(N);
}

// One workgroup of two 64-lane waves per path: wave 0 runs the backward extremals, wave 1 the
// forward extremals; within one switching-point loop the two extremals run concurrently (they
// are data-independent after the backward extremal's first step, see add_extremal). Both
// waves keep identical copies of the loop scalars. qd/qdd are written by the extremals as
// the backward wave finishes with its share (emit_range); the tail is shared by the two waves.
// TPAMD_SWEEP_WAVES_PER_EU (build-time, A/B): ask the compiler for a register budget that lets
// that many waves share a SIMD (3 -> 168 VGPRs).
// Wide records (D > 8) get the budget of ONE wave per SIMD: the 14-joint kernel needs 300
// registers (256 + 44 in the accumulation half of the unified file) and spilled 184 bytes per
// lane to scratch memory -- loads on the sequential chain -- when held to 256. At 304 allocated
// registers one 7-joint sweep wave (196) or two sampling/LP waves still fit beside it on a SIMD.
#ifndef TPAMD_SWEEP_WAVES_PER_EU
#define TPAMD_SWEEP_WAVES_PER_EU 2
#endif
#ifndef TPAMD_SWEEP_WAVES_PER_EU_WIDE
#define TPAMD_SWEEP_WAVES_PER_EU_WIDE 1
#endif
#define TPAMD_SWEEP_OCCUPANCY                                                                          \
  __attribute__((amdgpu_waves_per_eu((D > 8) ? TPAMD_SWEEP_WAVES_PER_EU_WIDE : TPAMD_SWEEP_WAVES_PER_EU, \
                                     (D > 8) ? TPAMD_SWEEP_WAVES_PER_EU_WIDE : TPAMD_SWEEP_WAVES_PER_EU)))
#ifdef TPAMD_K1_STUDY
#define TPAMD_SWEEP_STUDY_REGS __attribute__((amdgpu_num_vgpr(200)))
#else
#define TPAMD_SWEEP_STUDY_REGS
#endif
template <int D, int E = 0>
__global__ void __launch_bounds__(128) TPAMD_SWEEP_OCCUPANCY TPAMD_SWEEP_STUDY_REGS
k_sweep_joint(int stride, int max_loops, JointSource src, Workspace ws, double *t_out, double *s_out,
              double *sd_out, double *sdd_out, int32_t *lei_out, double *dtmax_out,
              int32_t *status_out, double *qd_out, double *qdd_out) {
  extern __shared__ double lds[];
  typedef JointSweep<D, E> JS;
  typedef SweepLds<D, E> LL;
  const int b = uniform_i32(path_of_block(ws, blockIdx.x));
  const int lane = threadIdx.x & 63;
  const int N = path_samples(ws, b, stride);   // samples of this path; arrays use `stride`
  const int w = uniform_i32((int)(threadIdx.x >> 6));
  const int tid = threadIdx.x;
  const size_t pb = (size_t)b * stride;
  const uint32_t bits = ws.err_bits[b];
  if (bits & kErrSkip) return;          // not part of this solve: outputs stay as they are
  if (bits) {
    if (tid == 0) {
      status_out[b] = status_from_bits(bits);
      if (lei_out) lei_out[b] = 0;
      if (dtmax_out) dtmax_out[b] = -1.0;
    }
    return;
  }
#ifdef TPAMD_K1_STUDY
  // study build: start / end clock of every sweep workgroup, per workspace slot
  unsigned long long *study = reinterpret_cast<unsigned long long *>(ws.diag) + 40960 + ((ws.keep_boundary >> 1) & 1) * 4096;
  if (tid == 0) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    study[2 * b] = t | ((unsigned long long)(xcc & 15u) << 60);
  }
#endif
  JS S;
  S.N = N; S.lane = lane;
  S.ds = ws.ds[b];
  S.two_ds = 2.0 * S.ds;
  S.sd2 = lds;
  double *ring = lds + LL::n2(N);
  S.tiles = ring + (size_t)w * 2 * JS::kTile * JS::R;
  uint8_t *typel = reinterpret_cast<uint8_t *>(ring + LL::kRingDoubles);
  S.typel = typel;
  int *xchg = reinterpret_cast<int *>(typel + LL::type_bytes(N));
  const int nw = LL::words(N);
  uint32_t *dirty = reinterpret_cast<uint32_t *>(xchg + 16);
  uint32_t *zero = dirty + nw;
  double *aml = reinterpret_cast<double *>(zero + nw);
  double *red = aml + 16;
  S.dirty = dirty;
  S.aml = aml;
  S.sdd_g = sdd_out + pb;
  S.m_g = ws.m + pb;
  S.rec = src.q12 + pb * JS::R;
  S.qd_g = qd_out ? qd_out + pb * D : nullptr;
  S.qdd_g = qdd_out ? qdd_out + pb * D : nullptr;
  S.end_idx = 0;
  S.tag0 = -1; S.tag1 = -1;
  double *sd2 = S.sd2;
  const double sd_start = ws.sd_start[b];
#ifdef TPAMD_DIAG
  long long(&diag)[32] = S.diag;
  for (int k = 0; k < 32; k++) diag[k] = 0;
#define TPAMD_DIAG_PTR S.diag
#else
#define TPAMD_DIAG_PTR nullptr
#endif
  // CalculateBoundary passes 2-4 for this path; scratch in the LDS the sweep uses afterwards
  // (flags in the sd2 array, re-fit values in the tile rings)
  boundary_passes_for_path<D, E>(src, ws, b, N, stride, tid, reinterpret_cast<uint8_t *>(sd2),
                           reinterpret_cast<uint8_t *>(sd2) + LL::type_bytes(N),
                           reinterpret_cast<char *>(ring), typel, (ws.keep_boundary & 1) != 0,
                           TPAMD_DIAG_PTR);
  // (the boundary passes run before the per-lane constants below are loaded: they need most of
  // the register file for themselves)
  constexpr int C = 2 * D + E;
  const double *lim_lo = src.lim + (size_t)b * 2 * C, *lim_hi = lim_lo + C;
  {
    typedef JointLayout<D> L;
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    const bool is_cand = lane < L::kCandLanes;
    const int c = is_cand ? lane / L::PARTS : 0;     // candidate
    const int p = lane % L::PARTS;                   // part
    const int r = c >> 1;                            // its row
    S.own_off = r;
    S.lim = is_cand ? ((c & 1) ? lim_hi[r] : lim_lo[r]) : qnan();
#pragma unroll
    for (int k = 0; k < L::RPL; k++) {
      int j = p + k * L::PARTS;
      if (j >= D) j = (p < D) ? p : 0;               // short share: re-check a row again
      S.chk_off[k] = j;
      S.chk_hi[k] = lim_hi[j];
    }
    const bool is_vel = lane < D;
    S.vel_off = is_vel ? lane : 0;
    S.vel_hi = is_vel ? lim_hi[D + lane] : kInf;
    S.vel_lo = is_vel ? 0.0 : -kInf;
    S.vel_mode = 0;
    if (E > 0) {
      if (lane >= D && lane < D + E) {          // lanes D, D+1: the extra rows
        S.vel_off = D;
        S.vel_mode = 1 + (lane - D);
        S.vel_hi = lim_hi[2 * D + (lane - D)];
        S.vel_lo = lim_lo[2 * D + (lane - D)];
      }
#pragma unroll
      for (int k = 0; k < (E ? E : 1); k++) S.ex_hi[k] = lim_hi[2 * D + k];
    }
  }
  S.row_lo = (lane < C) ? lim_lo[lane] : 0.0;
  S.row_hi = (lane < C) ? lim_hi[lane] : 0.0;
  {
    typedef JointLayout<D> L;
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    const int p = lane & (L::GRP - 1);
#pragma unroll
    for (int i = 0; i < L::CPL; i++) {
      const int cand = p + L::GRP * i;
      const bool has = cand < 2 * D;
      const int row = has ? (cand >> 1) : 0;
      S.v_off[i] = row;
      S.v_lim[i] = has ? ((cand & 1) ? lim_hi[row] : lim_lo[row]) : qnan();
    }
#pragma unroll
    for (int i = 0; i < L::VPL; i++) {
      const int row = p + L::GRP * i;
      const bool has = row < D;
      S.vv_off[i] = has ? row : 0;
      S.vv_hi[i] = has ? lim_hi[D + row] : kInf;
      S.va_hi[i] = has ? lim_hi[row] : kInf;
    }
  }

  for (int i = tid; i < N; i += 128) {
    sd2[i] = qnan();
    S.sdd_g[i] = qnan();
  }
  for (int i = tid; i < 2 * nw; i += 128) dirty[i] = 0u;   // dirty and zero-pair bitmaps
  if (tid < D) aml[tid] = ws.amax ? ws.amax[(size_t)b * D + tid] : 0.0;
  __syncthreads();
  if (tid == 0) {
    xchg[2] = 0;
    xchg[6] = 0;
    sd2[0] = sd_start * sd_start;
    sd2[N - 1] = 0;
  }
  __syncthreads();

  int status = 0;
  int iforw_lo = 0, iback_hi = N - 1, iback_lo, iforw_hi, icrit, icrit_lo, icrit_hi;
  typename JS::Prefetch pf;
  pf.tag = -1;
  TPAMD_T0C(t_all);
  // First pair (time_optimal_path_timing.cc:325-326): the backward extremal from N-1, then the
  // forward one from 0, which may run into it. They are run SIDE BY SIDE first: if they end more
  // than 70 samples apart neither has seen anything of the other -- an extremal reads sd2 at
  // most 64 samples ahead of itself (boundary following) and the two regions only grow towards
  // each other -- so both are exactly what the sequential order produces. Otherwise (short
  // paths, paths without a limit-curve contact) sd2 and sdd are reset and the pair is redone in
  // order (`first_pair_in_order`, a copy of its own of the two extremals: a loop around one copy
  // costs 30 VGPRs, and above 208 the sampling/LP kernel no longer fits beside two sweep waves).
  // qd/qdd bookkeeping of wave 0: samples below emitted_hi and from upper_lo up have been
  // written (see emit_range)
  int emitted_hi = 0, upper_lo = N;
#ifndef TPAMD_FIRST_PAIR_CONCURRENT
#define TPAMD_FIRST_PAIR_CONCURRENT 1
#endif
#ifndef TPAMD_EMIT_IN_LOOP
#define TPAMD_EMIT_IN_LOOP 1
#endif
  bool first_pair_in_order = !TPAMD_FIRST_PAIR_CONCURRENT;
  if (TPAMD_FIRST_PAIR_CONCURRENT) {
    if (w == 0) {
      const int r = S.template add_extremal<false>(iback_hi, pf);
      if (lane == 0) { xchg[0] = r; xchg[7] = S.end_idx; }
    } else {
      const int r = S.template add_extremal<true>(iforw_lo, pf);
      if (lane == 0) { xchg[1] = r; xchg[8] = S.end_idx; }
    }
    __threadfence_block();
    __syncthreads();
    const int end_b = uniform_i32(xchg[7]), end_f = uniform_i32(xchg[8]);
    if (!(end_f + 70 < end_b)) {                     // (uniform over the workgroup)
      // the two met or came close: back to the state before the attempt
      first_pair_in_order = true;
      for (int i = tid; i < N; i += 128) {
        sd2[i] = qnan();
        S.sdd_g[i] = qnan();
      }
      __syncthreads();
      if (tid == 0) {
        sd2[0] = sd_start * sd_start;
        sd2[N - 1] = 0;
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  if (first_pair_in_order) {
    if (w == 0) {
      const int r = S.template add_extremal<false>(iback_hi, pf);
      if (lane == 0) xchg[0] = r;
    }
    __threadfence_block();
    __syncthreads();
    if (w == 1) {
      const int r = S.template add_extremal<true>(iforw_lo, pf);
      if (lane == 0) xchg[1] = r;
    } else {
      // meanwhile: the region the first backward extremal has just set, short of its lower end
      // (which the NaN mark below and the connecting forward extremal may still change; what
      // the latter rewrites is redone in the tail)
      if (TPAMD_EMIT_IN_LOOP) {
        upper_lo = min(S.end_idx + 3, N);
        S.emit_range(upper_lo, N - 1, 0, 1);
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  iback_lo = uniform_i32(xchg[0]);
  iforw_hi = uniform_i32(xchg[1]);
  TPAMD_ACCC(9, t_all);   // first pair (sequential)
  icrit_hi = iback_lo;
  if ((iforw_hi < icrit_hi) && ((icrit_hi < N - 2) && (icrit_hi >= 2))) {
    if (w == 0) S.put_sd2(icrit_hi, qnan());
    icrit_hi++;
    iback_lo++;
  }
  icrit_lo = iforw_hi;
  const int zlast = uniform_i32(S.last_flagged_below(icrit_hi));   // icrit_hi is final from here on
  const double *m_g = ws.m + pb;
  if (max_loops <= 0) max_loops = max(100, 10 * N);   // path_timing_trajectory.cc:398-400
  // The switching-point loop, .cc:329-397. What a loop needs before its two extremals can start --
  // the next critical point (a scan of the type bytes above the forward extremal's end) and the
  // limit-curve values next to it (two global loads) -- depends on nothing the backward wave
  // produces, so the forward wave finds them right after its own extremal, while the backward wave
  // finishes its qd/qdd trip, and posts them in LDS before the loop's closing barrier: a loop is two
  // workgroup barriers (A, C) with nothing but the marks between C and A.
  double *crit_m = reinterpret_cast<double *>(xchg + 10);     // [0] sd2_max[icrit], [1] sd2_max[icrit - 1]
  auto post_next_critical_point = [&](int lo) {               // forward wave only
    TPAMD_T0(t0);
    int ic = uniform_i32(S.next_critical_point(lo, icrit_hi, zlast));
    TPAMD_ACC(2, t0);
#if defined(TPAMD_DIAG) && !defined(TPAMD_DIAG_LIGHT)
    {
      TPAMD_T0(tl_);
      if (ic != S.next_critical_point_literal(lo, icrit_hi)) TPAMD_CNT(7);
      TPAMD_ACC(18, tl_);
    }
#endif
    if (ic < 0 || ic >= N) ic = (int)(0.5 * (lo + icrit_hi));
    const double mc = m_g[min(max(ic, 0), N - 1)], mp = m_g[min(max(ic - 1, 0), N - 1)];
    if (ic >= 1) {
      // the tile the next forward extremal starts in: its loads fly during the barrier
      const int ts = ic / JS::kTile;
      const int tag = (ts & 1) ? S.tag1 : S.tag0;
      if (tag != ts && pf.tag != ts) S.issue_tile_loads(ts, pf);
    }
    if (lane == 0) {
      xchg[9] = ic;
      crit_m[0] = mc;
      crit_m[1] = mp;
    }
  };
  __syncthreads();     // the NaN mark above is visible (the literal walk of the diagnostic build reads sd2)
  if (w == 1 && iforw_hi < icrit_hi) post_next_critical_point(icrit_lo);
  __threadfence_block();
  __syncthreads();
  for (int loop = 0; loop < max_loops; loop++) {
    if (iforw_hi >= icrit_hi) break;
    TPAMD_CNT(11);
    icrit = uniform_i32(xchg[9]);
    const double m_c = uniform_f64(crit_m[0]), m_p = uniform_f64(crit_m[1]);
    if (w == 0 && icrit >= 1) {
      const int ts = (icrit - 1) / JS::kTile;
      const int tag = (ts & 1) ? S.tag1 : S.tag0;
      if (tag != ts && pf.tag != ts) S.issue_tile_loads(ts, pf);
    }
    // (both waves are past their extremals: nobody reads sd2 while the marks are written)
    if (icrit > 0 && icrit < N - 1 && w == 0) S.put_sd2(icrit, m_c);
    if (icrit < 1) { status = 10; break; }
    if (m_p <= m_c) {
      iback_hi = icrit - 1;
      if (w == 0) S.put_sd2(icrit - 1, m_p);
    } else {
      iback_hi = icrit;
    }
    iforw_lo = icrit;
    __syncthreads();                 // A: the marks are visible to the forward wave
    {
      TPAMD_T0C(t0);
      if (w == 0) {
        const int r = S.template add_extremal<false>(iback_hi, pf, /*pair_signal=*/true);   // B inside
        if (lane == 0) xchg[0] = r;
        // The forward extremal of this loop works on samples >= icrit and usually takes
        // longer: write qd/qdd for everything below icrit that is new or was changed by this
        // backward extremal (it ended at end_idx) while waiting for it.
        // -- but only until that one is done (xchg[6] then holds this loop's number): the
        // rest waits for the next loop or the tail.
        TPAMD_T0(te);
        if (TPAMD_EMIT_IN_LOOP) {
          emitted_hi = uniform_i32(S.emit_range(max(min(S.end_idx - 1, emitted_hi), 0), icrit - 1,
                                                0, 1, xchg + 6, loop + 1));
        }
        TPAMD_ACC(12, te);
      } else {
        const int r = S.template add_extremal<true>(iforw_lo, pf, false, /*wait_pair=*/true);   // B inside
        if (lane == 0) {
          xchg[1] = r;
          *reinterpret_cast<volatile int *>(xchg + 6) = loop + 1;
        }
        if (r < icrit_hi) post_next_critical_point(r);      // for the next loop, if there is one
      }
      TPAMD_ACCC(0, t0);
    }
    {
      TPAMD_T0C(t0);
      __threadfence_block();
      __syncthreads();               // C
      TPAMD_ACCC(1, t0);             // time spent waiting for the partner's extremal
    }
    iback_lo = uniform_i32(xchg[0]);
    iforw_hi = uniform_i32(xchg[1]);
    if (iback_lo > icrit_lo) { status = 7; break; }
    icrit_lo = iforw_hi;
  }
  TPAMD_ACCC(5, t_all);
  if (lane == 0) {
    if (w == 0) { xchg[4] = emitted_hi; xchg[5] = upper_lo; }
    else xchg[3] = S.end_idx;          // where the last forward extremal ended
  }
  // sdd_, qd, qdd were written with plain global stores; make them visible to both waves
  __threadfence_block();
  __syncthreads();

  // ---------------------------------------------------------------------------------------
  // Tail (time_optimal_path_timing.cc:398-477), shared by the two waves. `status` is uniform
  // over the workgroup (both waves computed it from the same exchanged values).
  // ---------------------------------------------------------------------------------------
  TPAMD_T0C(t_tail);
  const double ds = S.ds;
  double *sdd = S.sdd_g;
  {
    // a sample never reached by an extremal: no solution (.cc:400-403)
    int has_nan = 0;
    if (status == 0)
      for (int idx = tid; idx < N; idx += 128)
        if (isnan(sd2[idx])) has_nan = 1;
    if (has_nan) xchg[2] = 1;           // zeroed at set-up; every writer stores the same value
    __syncthreads();
    if (status == 0 && xchg[2] != 0) status = 8;
  }
  if (status == 0) {
    // sdd at extremal intersections (.cc:404-411, ComputeSddAtIntersection :722-751 for one
    // sample alone); four independent loads per thread in flight
    for (int base = 0; base < N; base += 512) {
      double cur[4];
#pragma unroll
      for (int u = 0; u < 4; u++) cur[u] = sdd[min(base + 128 * u + tid, N - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int idx = base + 128 * u + tid;
        if (idx < N && isnan(cur[u])) {
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
          atomicOr(&dirty[idx >> 5], 1u << (idx & 31));
        }
      }
    }
    // start acceleration if admissible (.cc:413-416): rows over the lanes of wave 0
    if (w == 0) {
      const double sdd_start = ws.sdd_start[b];
      const auto r0 = src.at(b, stride, 0);
      const double s20 = sd2[0];
      bool bad = false;
      for (int i = lane; i < C; i += 64) {
        const double v = r0.a(i) * sdd_start + r0.b(i) * s20;
        if (v + kTiny < r0.lo(i) || v - kTiny > r0.hi(i)) bad = true;
      }
      if (!__any(bad) && lane == 0) {
        sdd[0] = sdd_start;
        atomicOr(&dirty[0], 1u);
      }
    }
    if (sd2[N - 1] != 0) status = 9;    // .cc:422-428
  }
  if (status != 0) {
    if (tid == 0) {
      status_out[b] = status;
      if (lei_out) lei_out[b] = 0;
      if (dtmax_out) dtmax_out[b] = -1.0;
    }
#ifdef TPAMD_DIAG
    if (lane == 0 && ws.diag)
      for (int k = 0; k < 32; k++) ws.diag[(size_t)b * 64 + 32 * w + k] = S.diag[k];
#endif
    return;
  }
  __threadfence_block();
  __syncthreads();                      // the filled-in sdd values are visible to both waves
  {
    // qd/qdd the loop left: from the backward wave's frontier up to the region written during
    // the first pair, or as far as the connecting forward extremal rewrote that region
    // (measured: handing all of it to wave 1 while wave 0 runs the time sum does not shorten the step)
    const int e_hi = uniform_i32(xchg[4]), u_lo = uniform_i32(xchg[5]);
    const int f_end = uniform_i32(xchg[3]);
    S.emit_range(e_hi, min(max(u_lo - 1, f_end + 1), N - 1), w, 2);
  }
  TPAMD_ACC(13, t_tail);
  const double t0v = ws.t_start[b];
  const double s0 = ws.s_start[b], s1 = ws.s_end[b];
  double *tl = ring;                    // one chunk of dt, then time, values
  constexpr int kCap = LL::kRingDoubles;
  double tprev = t0v;                   // time_[base - 1] (wave 0)
  double dtmax = 0.0;
  for (int base = 0; base < N; base += kCap) {
    const int n = min(kCap, N - base);
    const bool last_chunk = base + kCap >= N;
    TPAMD_T0(t_a);
    // (a) every sample independently: sd = sqrt(sd2) (.cc:420), the time increment of the
    //     pair (idx-1, idx) (.cc:450-452), s (.cc:540-547), the copies of sd2
    for (int k = tid; k < n; k += 128) {
      const int idx = base + k;
      const double s2 = sd2[idx];
      const double sdv = sqrt(s2);
      double dt = 0.0;
      if (idx >= 1) {
        const double s2p = sd2[idx - 1];
        if ((s2p > 0) || (s2 > 0)) {
          dt = 2.0 * ds / (sqrt(s2p) + sdv);
        } else {
          // stationary pair (.cc:463-465): both accelerations become 0 -- after the
          // last-extremal scan below has read them
          const uint32_t m0 = 1u << ((idx - 1) & 31), m1 = 1u << (idx & 31);
          atomicOr(&zero[(idx - 1) >> 5], m0); atomicOr(&dirty[(idx - 1) >> 5], m0);
          atomicOr(&zero[idx >> 5], m1);       atomicOr(&dirty[idx >> 5], m1);
        }
      }
      tl[k] = dt;
      sd_out[pb + idx] = sdv;
      s_out[pb + idx] = (idx == N - 1) ? s1 : ds * idx + s0;
      ws.sd2[pb + idx] = s2;
      if (ws.sd2_out) ws.sd2_out[pb + idx] = s2;
      if (dt > dtmax) dtmax = dt;
    }
    __syncthreads();
    TPAMD_ACC(16, t_a);
    TPAMD_T0(t_b);
    if (w == 0) {
      // (b) the time integral, strictly left to right (.cc:453-454), 64 samples at a time: lane L
      //     holds dt of sample L of the block and receives ((t + dt_0) + dt_1) ... + dt_L, the
      //     reference's order of additions, from ordered_sum64 -- one instruction per sample,
      //     nothing through LDS.
      double t = tprev;
      OrderedSumMasks osm;
      osm.init(lane);
      for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const double dtv = (k < n) ? tl[k] : 0.0;   // (+0.0 past the end: the sum passes through)
        // (increments are >= 0; an infinite one -- both velocities denormal -- takes the variant
        // that never multiplies)
        const bool finite = __ballot(!(dtv <= DBL_MAX)) == 0ull;
        const double y = finite ? ordered_sum64_finite(t, dtv, osm) : ordered_sum64(t, dtv);
        if (k < n) tl[k] = y;
        t = readlane_f64(y, 63);
      }
      JS::wave_lds_sync();
      tprev = t;
    } else {
      // (b') wave 1 meanwhile
      if (base == 0) {
        // last_extremal_index_ (.cc:430-445): scan down from N-2 for sdd > 0 or a sample on
        // the boundary curve
        int lei = 0;
        const int start = (1 > N - 2) ? 1 : N - 2;
        bool found = false;
        for (int top = start; top >= 1 && !found; top -= 256) {
          double a[4], mm[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {          // unconditional, independent loads
            const int ic = max(top - 64 * u - lane, 0);
            a[u] = sdd[ic];
            mm[u] = m_g[ic];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int idx = top - 64 * u - lane;
            const bool hit = (idx >= 1) & ((a[u] > 0.0) | (fabs(sd2[max(idx, 0)] - mm[u]) < kTiny));
            const unsigned long long mask = __ballot(hit);
            if (mask && !found) { lei = top - 64 * u - (__ffsll((long long)mask) - 1); found = true; }
          }
        }
        if (lane == 0 && lei_out) lei_out[b] = lei;
      }
      if (last_chunk) {
        // zero acceleration across stationary pairs (.cc:463-465)
        for (int wi = lane; wi < nw; wi += 64) {
          uint32_t zb = zero[wi];
          while (zb) {
            const int i = __ffs((int)zb) - 1;
            sdd[wi * 32 + i] = 0.0;
            zb &= zb - 1u;
          }
        }
        __threadfence_block();
        // qd/qdd of the samples the extremals could not finish (starts, ends, intersections,
        // filled-in or zeroed accelerations), from the final sd2 and sdd
        if (S.emitting()) {
          int *list = reinterpret_cast<int *>(typel);
          const int list_cap = LL::type_bytes(N) / 4;
          int count = 0;
          bool all = false;
          for (int wb = 0; wb < nw; wb += 64) {
            const int wi = wb + lane;
            uint32_t db = (wi < nw) ? dirty[wi] : 0u;
            const int c = __popc(db);
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
              const int o = __shfl_up(incl, off, 64);
              if (lane >= off) incl += o;
            }
            const int total = count + __shfl(incl, 63, 64);
            if (total > list_cap) { all = true; break; }     // (uniform) redo every sample
            int pos = count + incl - c;
            while (db) {
              const int i = __ffs((int)db) - 1;
              list[pos++] = wi * 32 + i;
              db &= db - 1u;
            }
            count = total;
          }
          JS::wave_lds_sync();
          constexpr int PER = 64 / D;          // samples per pass, one lane per (sample, joint)
          constexpr int U = 4;                 // passes in flight
          const int g = lane / D, d = lane - g * D;
          const f64x2 *rec2 = reinterpret_cast<const f64x2 *>(S.rec);
          const int total = all ? N : count;
          for (int k0 = 0; k0 < total; k0 += PER * U) {
            f64x2 pr[U];
            double s2v[U], av[U];
            int id[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
              const int k = k0 + u * PER + g;
              ok[u] = (g < PER) && (k < total);
              const int kk = ok[u] ? k : k0;
              id[u] = all ? kk : list[kk];
              pr[u] = rec2[(size_t)id[u] * (JS::R / 2) + d];
              s2v[u] = sd2[id[u]];
              av[u] = sdd[id[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
              if (ok[u]) S.emit_value(id[u], d, pr[u], sqrt(s2v[u]), av[u], aml[d]);
          }
        }
      }
    }
    TPAMD_ACC(17, t_b);
    __syncthreads();
    // (c) the chunk's time samples, coalesced
    for (int k = tid; k < n; k += 128) t_out[pb + base + k] = tl[k];
    if (!last_chunk) __syncthreads();   // before the next chunk reuses the buffer
  }
  dtmax = wave_max_f64(dtmax);
  if (lane == 0) red[w] = dtmax;
  __syncthreads();
  if (tid == 0) {
    status_out[b] = 0;
    if (dtmax_out) dtmax_out[b] = (red[0] > red[1]) ? red[0] : red[1];
  }
#ifdef TPAMD_K1_STUDY
  if (tid == 0) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    study[2 * b + 1] = t;
  }
#endif
  TPAMD_ACCC(3, t_tail);
  TPAMD_ACCC(15, t_all);
#ifdef TPAMD_DIAG
  if (lane == 0 && ws.diag)
    for (int k = 0; k < 32; k++) ws.diag[(size_t)b * 64 + 32 * w + k] = S.diag[k];
#endif
}

// ---------------------------------------------------------------------------------------
// The time samples of solved paths rebuilt from their velocities: time_[i] = time_[i-1] +
// 2 ds / (sd_[i-1] + sd_[i]), 0 across a stationary pair, summed left to right
// (time_optimal_path_timing.cc:447-455) -- the operations of the sweep kernel's tail, hence the
// same bits. One wave per path. Used on the root of a multi-GPU job: a shard then sends (sd, sdd)
// and two scalars per path instead of (t, sd, sdd), a third less through the root's xGMI links.
// Paths are addressed as shard r = p / paths_per_shard, b = p % paths_per_shard:
//   sd   at sd_base   + r * shard_stride + b * N      (doubles)
//   ds   at ds_base   + r * shard_stride + b,   t_start likewise
//   out  at t_out     + (r * paths_per_shard + b) * N
// ---------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(64)
k_rebuild_time(int N, int paths_per_shard, size_t shard_stride, const double *sd_base,
               const double *ds_base, const double *t_start_base, const int32_t *ns, double *t_out) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const int r = p / paths_per_shard, b = p - r * paths_per_shard;
  const double *sd = sd_base + (size_t)r * shard_stride + (size_t)b * N;
  const double ds = ds_base[(size_t)r * shard_stride + b];
  double t = t_start_base[(size_t)r * shard_stride + b];
  double *out = t_out + (size_t)p * N;
  const int Nb = ns ? min(ns[p], N) : N;
  OrderedSumMasks osm;
  osm.init(lane);
  for (int k0 = 0; k0 < Nb; k0 += 64) {
    const int i = k0 + lane;
    const bool in = i < Nb;
    const double sdv = in ? sd[i] : 0.0;
    const double sdp = (in && i >= 1) ? sd[i - 1] : 0.0;
    double dt = 0.0;
    if (in && i >= 1 && ((sdp > 0) || (sdv > 0))) dt = 2.0 * ds / (sdp + sdv);
    const bool finite = __ballot(!(dt <= DBL_MAX)) == 0ull;
    const double y = finite ? ordered_sum64_finite(t, dt, osm) : ordered_sum64(t, dt);
    if (in) out[i] = y;
    t = readlane_f64(y, 63);
  }
}

}  // namespace tpamd
