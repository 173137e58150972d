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

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Extreme value over each 16-lane DPP row (butterfly: xor 1, xor 2, half-row
// mirror, row mirror). Inputs must not be NaN.
template <bool MAX>
__device__ __forceinline__ double row16_extreme(double v) {
  double o;
  o = dpp_f64<0xB1>(v);  v = MAX ? __builtin_fmax(o, v) : __builtin_fmin(o, v);   // quad_perm [1,0,3,2]
  o = dpp_f64<0x4E>(v);  v = MAX ? __builtin_fmax(o, v) : __builtin_fmin(o, v);   // quad_perm [2,3,0,1]
  o = dpp_f64<0x141>(v); v = MAX ? __builtin_fmax(o, v) : __builtin_fmin(o, v);   // row_half_mirror
  o = dpp_f64<0x140>(v); v = MAX ? __builtin_fmax(o, v) : __builtin_fmin(o, v);   // row_mirror
  return v;
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
#define TPAMD_ACC(slot, var) diag[slot] += tpamd_stamp() - (var)
#define TPAMD_CNT(slot) diag[slot] += 1
#define TPAMD_ADD(slot, v) diag[slot] += (v)
#else
#define TPAMD_T0(var)
#define TPAMD_ACC(slot, var)
#define TPAMD_CNT(slot)
#define TPAMD_ADD(slot, v)
#endif

#ifndef TPAMD_TILE_SAMPLES
#define TPAMD_TILE_SAMPLES 32
#endif
constexpr int kTileSamples = TPAMD_TILE_SAMPLES;

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
  return MAX ? __builtin_fmax(a, b) : __builtin_fmin(a, b);
}

// E: extra rows with A = 0 and an explicit B (Cartesian paths: 2, stored as one pair after
// the D (q', q'') pairs); they behave like velocity rows with lower = -upper.
template <int D, int E = 0>
struct JointSweep {
  typedef JointLayout<D> L;
  static_assert(E == 0 || E == 2, "extra rows come as one pair");
  static constexpr int R = 2 * D + E + 2;                   // doubles per record
  static constexpr int kMt = D + E / 2;                     // pair index of (sd2_max, type)
  static constexpr int kChunks = kTileSamples * R / 2;      // 16-byte chunks per tile
  static constexpr int kChunksPerLane = (kChunks + 63) / 64;
#ifdef TPAMD_DIAG
  long long diag[16];
#endif
  int N, lane;
  double ds, two_ds;
  double *sd2;            // LDS [N]
  double *tiles;          // LDS [2][kTileSamples][R]
  const uint8_t *typel;   // LDS [N] copy of the type bytes
  double *sdd_g;          // global: sdd output row of this path
  const double *m_g;      // global: final sd2_max of this path [N]
  const double *rec;      // global: records of this path [N][R]
  int tag0, tag1;         // tile index resident in ring slot 0 / 1 (-1: none)
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
  struct Prefetch {
    RegPack<kChunksPerLane> r;
    int tag;   // tile index held in r (-1: none)
  };
  __device__ __forceinline__ void issue_tile_loads(int t, Prefetch &pf) const {
    const f64x2 *src = reinterpret_cast<const f64x2 *>(rec + (size_t)t * kTileSamples * R);
    pf.r.template load<kChunks>(src, lane);
    pf.tag = t;
  }
  __device__ __forceinline__ void store_tile(int slot, const Prefetch &pf) const {
    f64x2 *dst = reinterpret_cast<f64x2 *>(tiles + (size_t)slot * kTileSamples * R);
    pf.r.template store<kChunks>(dst, lane);
    wave_lds_sync();
  }
  // Make tile t resident; dir tells which neighbour tile to prefetch afterwards.
  __device__ __forceinline__ void fill_tile(int t, int dir, Prefetch &pf) {
    TPAMD_CNT(12);
    if (pf.tag != t) { TPAMD_CNT(13); issue_tile_loads(t, pf); }
    wave_lds_sync();                 // earlier readers of this slot are done
    store_tile(t & 1, pf);           // waits for the loads, writes LDS
    if (t & 1) tag1 = t; else tag0 = t;
    const int tn = t + dir;
    pf.tag = -1;
    if (tn >= 0 && tn * kTileSamples < N) issue_tile_loads(tn, pf);
  }
  __device__ __forceinline__ void ensure_tile(int idx, int dir, Prefetch &pf) {
    const int t = idx / kTileSamples;
    const int tag = (t & 1) ? tag1 : tag0;
    if (tag != t) fill_tile(t, dir, pf);
  }
  __device__ __forceinline__ const f64x2 *record(int idx) const {
    // slot = (idx / 32) & 1, position = idx % 32  ==  idx & 63 in a 64-record ring
    return reinterpret_cast<const f64x2 *>(tiles) + (size_t)(idx & (2 * kTileSamples - 1)) * (R / 2);
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
      const double v = r.chk[k].x * sddi + r.chk[k].y * s2;
      bad = bad | (v + kTiny < -chk_hi[k]) | (v - kTiny > chk_hi[k]);
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
    double res = readlane_f64(best, 0);
#pragma unroll
    for (int k = 1; k < L::kRows16; k++) res = ext2<MAX>(res, readlane_f64(best, 16 * k));
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
  // passes row r makes the answer "unknown" (false), never wrong: the step then falls to
  // the scalar code.
  template <bool MAX>
  __device__ __forceinline__ bool chain_step_exact(int j, double hi_r, f64x2 arow, double s2,
                                                   double sddw) const {
    const f64x2 *p = record(j);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < L::VPL; i++) {       // winner against this lane's share of the rows
      const f64x2 pr = p[vv_off[i]];
      const double v = pr.x * sddw + pr.y * s2;
      bad = bad | (v + kTiny < -va_hi[i]) | (v - kTiny > va_hi[i]);
      const double vv = (pr.x * pr.x) * s2;
      bad = bad | (vv + kTiny < 0.0) | (vv - kTiny > vv_hi[i]);
    }
    if (E > 0) {                              // extra rows (every lane: two cheap checks)
      const f64x2 ex = p[D];
      const double v0 = ex.x * s2, v1 = ex.y * s2;
      bad = bad | (v0 + kTiny < -ex_hi[0]) | (v0 - kTiny > ex_hi[0]);
      bad = bad | (v1 + kTiny < -ex_hi[E - 1]) | (v1 - kTiny > ex_hi[E - 1]);
    }
#pragma unroll
    for (int i = 0; i < L::CPL; i++) {       // this lane's other candidates
      const f64x2 own = p[v_off[i]];
      const double sddi = (v_lim[i] - own.y * s2) / own.x;
      const bool better = MAX ? (sddi > sddw) : (sddi < sddw);
      const double v = arow.x * sddi + arow.y * s2;
      const bool passes = !((v + kTiny < -hi_r) | (v - kTiny > hi_r));
      bad = bad | (better & passes & !(fabs(own.x) < kTiny));
    }
    const unsigned long long bm = __ballot(bad);
    return ((bm >> (lane & ~(L::GRP - 1))) & ((1ull << L::GRP) - 1ull)) == 0ull;
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
  template <bool FWD>
  __device__ __forceinline__ int extremal_step(Carry &c, const Rows &use, Rows &stage, Prefetch &pf,
                                               int idx_start, bool &pair_signal) {
    constexpr int dir = FWD ? 1 : -1;
#define TPAMD_PAIR_SIGNAL()                                   \
  do {                                                        \
    if (pair_signal) {                                        \
      __threadfence_block();                                  \
      __syncthreads();                                        \
      pair_signal = false;                                    \
    }                                                         \
  } while (0)
    const int idx = c.idx;
    const int nidx = idx + dir;
    const bool more = FWD ? (nidx < N - 2) : (nidx > 1);
    // stage the next step's data (1 <= nidx <= N-2, 0 <= nidx+dir <= N-1); a new tile can
    // only be entered at a tile edge
    if (((nidx + dir) & (kTileSamples - 1)) == (FWD ? 0 : kTileSamples - 1))
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
      TPAMD_CNT(FWD ? 8 : 9);
      sd2tmp = m_n;
      sddtmp = FWD ? 0.5 * (sd2tmp - cur) / ds : 0.5 * (cur - sd2tmp) / ds;
      c.win = -1;
    } else {
      TPAMD_CNT(FWD ? 10 : 11);
      int win;
      sddtmp = uniform_f64(find_sdd<FWD>(use, cur, win));
      c.win = uniform_i32(win);
      sd2tmp = FWD ? cur + two_ds * sddtmp : cur - two_ds * sddtmp;
    }
    if (!isnan(nxt) && (nxt < sd2tmp)) {
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
        TPAMD_PAIR_SIGNAL();
        return idx;
      }
      sd2tmp = m_n;
      sddtmp = sdd_bound;
      c.win = -1;
    }
    if (sd2tmp < 0) {
      c.win = -1;
      sd2tmp = 0.0;
      if (FWD) {
        if (idx <= 1) sddtmp = 0.0; else sddtmp = -sd2[idx - 1] / ds;
      } else {
        if (idx < N - 1) sddtmp = sd2[idx + 1] / ds; else sddtmp = 0.0;
      }
    }
    put_sd2(nidx, sd2tmp);
    put_sdd(idx, sddtmp);
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
    const int tj = max(j, 0) / kTileSamples, tjn = max(jn, 0) / kTileSamples;
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
    if (lane < L) {
      const double cur_k = (lane == 0) ? c.cur : m_j;
      const double sddv = FWD ? 0.5 * (m_jn - cur_k) / ds : 0.5 * (cur_k - m_jn) / ds;
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
  template <bool FWD>
  __device__ __forceinline__ int follow_chain(Carry &c, Prefetch &pf) {
    constexpr int dir = FWD ? 1 : -1;
    constexpr int K = L::kChain;
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
    double cur = c.cur;
    double my_cur = 0.0;
#pragma unroll
    for (int s = 0; s < K; s++) {
      const double a = readlane_f64(arow.x, G * s), b = readlane_f64(arow.y, G * s);
      const double ys = readlane_f64(y, G * s);
      if (k == s) my_cur = cur;
      const double n = alim - b * cur;
      const double q0 = n * ys;
      const double rem = __builtin_fma(-a, q0, n);
      const double q = __builtin_fma(rem, ys, q0);
      cur = FWD ? cur + two_ds * q : cur - two_ds * q;
    }
    // every quad redoes its own step from the captured sd2_k with the reference's operations
    const double my_sdd = (alim - arow.y * my_cur) / arow.x;
    const double my_new = FWD ? my_cur + two_ds * my_sdd : my_cur - two_ds * my_sdd;
    double chain_next = __shfl_down(my_cur, G, 64);      // what the recurrence fed to step k+1
    if (k == K - 1) chain_next = cur;
    const bool exact = (__double_as_longlong(chain_next) == __double_as_longlong(my_new)) &&
                       chain_step_exact<FWD>(j, hi_r, arow, my_cur, my_sdd);
    const f64x2 mt_j = record(j)[kMt], mt_n = record(jn)[kMt];
    const int t_j = __double2loint(mt_j.y), t_n = __double2loint(mt_n.y);
    const double m_j = mt_j.x, m_n = mt_n.x;
    const double nxt = sd2[jn];
    const bool riding = is_tiny(my_cur - m_j) && (t_j & kBndTrajectory) && (t_n & kBndTrajectory);
    const bool special = riding || (!isnan(nxt) && (nxt < my_new)) || (my_new > m_n) || (my_new < 0) ||
                         isnan(my_new);
    const bool ok = in_loop && exact && !special;
    const unsigned long long okm = __ballot(ok);
    const unsigned long long miss = ~okm & L::kPart0;   // part-0 lanes
    const int Lc = miss ? ((__ffsll((long long)miss) - 1) / G) : K;
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

  // (Re)start the carried state at sample idx: tiles, rows, neighbours.
  template <bool FWD>
  __device__ __forceinline__ void init_carry(int idx, Carry &c, Rows &rows, Prefetch &pf, bool load_cur) {
    constexpr int dir = FWD ? 1 : -1;
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
      TPAMD_PAIR_SIGNAL();
      if (wait_pair) __syncthreads();
      return FWD ? N - 1 : 0;
    }
    Rows rows_a, rows_b;
    Carry c;
    init_carry<FWD>(idx_start, c, rows_a, pf, true);
    if (wait_pair) __syncthreads();
    bool trust = true;
    int last_win = -1;
    for (;;) {
#define TPAMD_TRY_FOLLOW()                                                                   \
  if (is_tiny(c.cur - c.m_i) && (c.t_i & kBndTrajectory) && (c.t_n & kBndTrajectory)) {      \
    const int run = uniform_i32(follow_boundary<FWD>(c));                                    \
    if (run > 0) {                                                                           \
      TPAMD_CNT(FWD ? 8 : 9);                                                                \
      TPAMD_PAIR_SIGNAL();                                                                   \
      if (FWD ? !(c.idx < N - 2) : !(c.idx > 1)) return FWD ? N - 1 : 0;                     \
      init_carry<FWD>(c.idx, c, rows_a, pf, false);                                          \
      continue;                                                                              \
    }                                                                                        \
  }
// After a FindSdd step selected a candidate: speculate on it (follow_chain) as long as whole
// blocks verify. `trust` drops after a block that verified nothing; it takes two scalar
// steps selecting the same candidate to try again.
#define TPAMD_TRY_CHAIN()                                                                    \
  if (c.win >= 0 && (trust || c.win == last_win)) {                                          \
    int total = 0, run;                                                                      \
    do {                                                                                     \
      { TPAMD_T0(tc_); run = uniform_i32(follow_chain<FWD>(c, pf)); TPAMD_ACC(4, tc_); }     \
      TPAMD_CNT(6);                                                                          \
      total += run;                                                                          \
    } while (run == L::kChain && (FWD ? (c.idx < N - 2) : (c.idx > 1)));                     \
    trust = total > 0;                                                                       \
    last_win = -1;                                                                           \
    if (total > 0) {                                                                         \
      TPAMD_ADD(FWD ? 14 : 15, total);                                                       \
      if (FWD ? !(c.idx < N - 2) : !(c.idx > 1)) return FWD ? N - 1 : 0;                     \
      init_carry<FWD>(c.idx, c, rows_a, pf, false);                                          \
      continue;                                                                              \
    }                                                                                        \
  } else {                                                                                   \
    last_win = c.win;                                                                        \
  }
      TPAMD_TRY_FOLLOW();
      int r = extremal_step<FWD>(c, rows_a, rows_b, pf, idx_start, pair_signal);
      if (r != kContinue) return r;
      TPAMD_TRY_CHAIN();
      if (is_tiny(c.cur - c.m_i) && (c.t_i & kBndTrajectory) && (c.t_n & kBndTrajectory)) {
        // a run starts here: restart the loop (it re-stages into rows_a)
        load_rows(c.idx, rows_a);
        continue;
      }
      r = extremal_step<FWD>(c, rows_b, rows_a, pf, idx_start, pair_signal);
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

// Dynamic LDS (bytes): sd2[N]*8 | tile rings WAVES*2*32*R*8 | type copy N (padded to 16) |
// exchange words 16 B
template <int D, int E = 0>
__host__ __device__ inline size_t sweep_joint_lds_bytes(int N, int waves) {
  return (size_t)N * 8 + (size_t)waves * 2 * kTileSamples * (2 * D + E + 2) * 8 +
         (((size_t)N + 15) / 16) * 16 + 16;
}

// WAVES = 1: one wave runs everything. WAVES = 2: wave 0 runs the backward extremals and
// the tail, wave 1 the forward extremals; within one switching-point loop the two
// extremals run concurrently (they are data-independent after the backward extremal's
// first step, see add_extremal). Both waves keep identical copies of the loop scalars.
template <int D, int WAVES, int E = 0>
__global__ void __launch_bounds__(64 * WAVES)
k_sweep_joint(int stride, int max_loops, JointSource src, Workspace ws, double *t_out, double *s_out,
              double *sd_out, double *sdd_out, int32_t *lei_out, double *dtmax_out,
              int32_t *status_out, double *qd_out, double *qdd_out) {
  extern __shared__ double lds[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = (WAVES == 2) ? uniform_i32((int)(threadIdx.x >> 6)) : 0;
  const int tid = threadIdx.x;
  const int N = path_samples(ws, b, stride);   // samples of this path; arrays use `stride`
  const size_t pb = (size_t)b * stride;
  const uint32_t bits = ws.err_bits[b];
  if (bits) {
    if (tid == 0) {
      status_out[b] = status_from_bits(bits);
      if (lei_out) lei_out[b] = 0;
      if (dtmax_out) dtmax_out[b] = -1.0;
    }
    return;
  }
  typedef JointSweep<D, E> JS;
  JS S;
  S.N = N; S.lane = lane;
  S.ds = ws.ds[b];
  S.two_ds = 2.0 * S.ds;
  S.sd2 = lds;
  S.tiles = lds + N + (size_t)w * 2 * kTileSamples * JS::R;
  uint8_t *typel = reinterpret_cast<uint8_t *>(lds + N + (size_t)WAVES * 2 * kTileSamples * JS::R);
  S.typel = typel;
  int *xchg = reinterpret_cast<int *>(typel + ((N + 15) / 16) * 16);
  S.sdd_g = sdd_out + pb;
  S.m_g = ws.m + pb;
  S.rec = src.q12 + pb * JS::R;
  S.tag0 = -1; S.tag1 = -1;
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

  double *sd2 = S.sd2;
  const double sd_start = ws.sd_start[b];
  const uint8_t *type_g = ws.type + pb;
  for (int i = tid; i < N; i += 64 * WAVES) {
    sd2[i] = qnan();
    S.sdd_g[i] = qnan();
    typel[i] = type_g[i];
  }
  __syncthreads();
  if (tid == 0) { sd2[0] = sd_start * sd_start; sd2[N - 1] = 0; }
  __syncthreads();

  int status = 0;
  int iforw_lo = 0, iback_hi = N - 1, iback_lo, iforw_hi, icrit, icrit_lo, icrit_hi;
  typename JS::Prefetch pf;
  pf.tag = -1;
#ifdef TPAMD_DIAG
  long long(&diag)[16] = S.diag;
  for (int k = 0; k < 16; k++) diag[k] = 0;
#endif
  TPAMD_T0(t_all);
  // First pair: the forward extremal from 0 may run into the backward one from N-1, so the
  // two are sequential (time_optimal_path_timing.cc:325-326).
  if (WAVES == 1) {
    iback_lo = uniform_i32(S.template add_extremal<false>(iback_hi, pf));
    iforw_hi = uniform_i32(S.template add_extremal<true>(iforw_lo, pf));
  } else {
    if (w == 0) {
      const int r = S.template add_extremal<false>(iback_hi, pf);
      if (lane == 0) xchg[0] = r;
    }
    __threadfence_block();
    __syncthreads();
    if (w == 1) {
      const int r = S.template add_extremal<true>(iforw_lo, pf);
      if (lane == 0) xchg[1] = r;
    }
    __threadfence_block();
    __syncthreads();
    iback_lo = uniform_i32(xchg[0]);
    iforw_hi = uniform_i32(xchg[1]);
  }
  TPAMD_ACC(9, t_all);   // first pair (sequential)
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
  for (int loop = 0; loop < max_loops; loop++) {
    if (iforw_hi >= icrit_hi) break;
    TPAMD_CNT(11);
    if (WAVES == 2) __syncthreads();   // sd2 writes of the other wave / the NaN mark are visible
    {
      TPAMD_T0(t0);
      icrit = uniform_i32(S.next_critical_point(icrit_lo, icrit_hi, zlast));
      TPAMD_ACC(2, t0);
#ifdef TPAMD_DIAG
      if (icrit != S.next_critical_point_literal(icrit_lo, icrit_hi)) TPAMD_CNT(7);
#endif
    }
    if (icrit < 0 || icrit >= N) icrit = (int)(0.5 * (icrit_lo + icrit_hi));
    if (icrit >= 1) {
      // the tile this wave's extremal starts in: its loads fly together with the loads of
      // the boundary values below instead of after them
      const int ts = ((WAVES == 2 && w == 1) ? icrit : icrit - 1) / kTileSamples;
      const int tag = (ts & 1) ? S.tag1 : S.tag0;
      if (tag != ts && pf.tag != ts) S.issue_tile_loads(ts, pf);
    }
    if (WAVES == 2) __syncthreads();   // both waves finished reading sd2 before the marks below
    if (icrit > 0 && icrit < N - 1 && w == 0) S.put_sd2(icrit, m_g[icrit]);
    if (icrit < 1) { status = 10; break; }
    if (m_g[icrit - 1] <= m_g[icrit]) {
      iback_hi = icrit - 1;
      if (w == 0) S.put_sd2(icrit - 1, m_g[icrit - 1]);
    } else {
      iback_hi = icrit;
    }
    iforw_lo = icrit;
    if (WAVES == 1) {
      {
        TPAMD_T0(t0);
        iback_lo = uniform_i32(S.template add_extremal<false>(iback_hi, pf));
        TPAMD_ACC(1, t0);
      }
      {
        TPAMD_T0(t0);
        iforw_hi = uniform_i32(S.template add_extremal<true>(iforw_lo, pf));
        TPAMD_ACC(0, t0);
      }
    } else {
      __syncthreads();                 // A: the marks are visible to the forward wave
      if (w == 0) {
        TPAMD_T0(t0);
        const int r = S.template add_extremal<false>(iback_hi, pf, /*pair_signal=*/true);   // B inside
        if (lane == 0) xchg[0] = r;
        TPAMD_ACC(1, t0);
      } else {
        TPAMD_T0(t0);
        const int r = S.template add_extremal<true>(iforw_lo, pf, false, /*wait_pair=*/true);   // B inside
        if (lane == 0) xchg[1] = r;
        TPAMD_ACC(0, t0);
      }
      __threadfence_block();
      __syncthreads();                 // C
      iback_lo = uniform_i32(xchg[0]);
      iforw_hi = uniform_i32(xchg[1]);
    }
    if (iback_lo > icrit_lo) { status = 7; break; }
    icrit_lo = iforw_hi;
  }
  TPAMD_ACC(5, t_all);
  // sdd_ was written with plain global stores; make it visible to all lanes of the tail
  __threadfence_block();
  __syncthreads();
#ifdef TPAMD_DIAG
  if (lane == 0 && ws.diag && w == WAVES - 1)
    for (int k = 0; k < 16; k++)
      if (WAVES == 1 || k == 0 || k >= 8) ws.diag[(size_t)b * 16 + k] = S.diag[k];
#endif
  int fstatus = 0;
  if (w == 0) {
    TPAMD_T0(t0);
    fstatus = sweep_tail(src, ws, b, N, stride, lane, status, sd2, S.sdd_g, /*copy_sdd=*/false, t_out,
                         s_out, sd_out, sdd_out, lei_out, dtmax_out, status_out);
    TPAMD_ACC(3, t0);
  }
#ifdef TPAMD_DIAG
  if (w == 0 && lane == 0 && ws.diag)
    for (int k = 0; k < 16; k++)
      if (WAVES == 1 || (k != 0 && k < 8)) ws.diag[(size_t)b * 16 + k] = S.diag[k];
#endif
  // Planner epilogue (path_timing_trajectory.cc:458-472) by all lanes of the block:
  // qd = q' sd, qdd = clamp(q' sdd + q'' sd^2, +-a_max). sd and sdd are read back from the
  // output rows the tail has just written.
  if (qd_out == nullptr && qdd_out == nullptr) return;
  if (WAVES == 2) {
    if (w == 0 && lane == 0) xchg[0] = fstatus;
    __threadfence_block();
    __syncthreads();
    fstatus = uniform_i32(xchg[0]);
  } else {
    tail_sync();
  }
  if (fstatus != 0) return;
  {
    const double *am_g = ws.amax + (size_t)b * D;
    const f64x2 *rec2 = reinterpret_cast<const f64x2 *>(S.rec);
    const int total = N * D;
    constexpr int kBatch = 16;   // loads in flight per lane (the loop is latency-bound otherwise)
    constexpr int kStep = 64 * WAVES;
    for (int e0 = tid; e0 < total; e0 += kStep * kBatch) {
      f64x2 pr[kBatch];
      double v[kBatch], a[kBatch], am[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; u++) {
        const int e = min(e0 + u * kStep, total - 1);
        const int i = e / D;
        const int d = e - i * D;
        pr[u] = rec2[(size_t)i * (JS::R / 2) + d];
        v[u] = sd_out[pb + i];
        a[u] = sdd_out[pb + i];
        am[u] = am_g[d];
      }
#pragma unroll
      for (int u = 0; u < kBatch; u++) {
        const int e = e0 + u * kStep;
        if (e < total) {
          if (qd_out) qd_out[pb * D + e] = pr[u].x * v[u];
          if (qdd_out) {
            const double v2 = v[u] * v[u];
            double acc = pr[u].x * a[u] + pr[u].y * v2;
            if (acc < -am[u]) acc = -am[u];
            if (acc > am[u]) acc = am[u];
            qdd_out[pb * D + e] = acc;
          }
        }
      }
    }
  }
}

}  // namespace tpamd
