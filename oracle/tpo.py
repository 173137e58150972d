"""ctypes binding of the CPU ORACLE (oracle/libtp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KTINY = 2.220446049250313e-16 * 1e5
KMAXSD2 = 1e6

OK = 0
ERR_NAMES = {
    0: "ok", 2: "infeasible_bounds", 3: "s_range", 4: "sd_start_neg",
    5: "lower_ge_upper", 6: "too_few_samples", 7: "no_connection",
    8: "nan_sd2", 9: "nonzero_end", 10: "crit_index_zero", 11: "not_solved",
}

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "libtp_oracle.so")
    deps = [os.path.join(_HERE, f) for f in ("tp_oracle.c", "tp_oracle_plan.c", "tp_oracle_quat.c", "tp_oracle.h")]
    if force or not os.path.exists(so) or any(
            os.path.getmtime(f) > os.path.getmtime(so) for f in deps):
        subprocess.check_call(["make", "-C", _HERE, "libtp_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    L = C.CDLL(so)
    d, i, vp = C.c_double, C.c_int, C.c_void_p
    L.tpo_knot_span.restype = i
    L.tpo_knot_span.argtypes = [_dp, i, i, d]
    L.tpo_basis.argtypes = [_dp, i, i, d, _dp]
    L.tpo_basis_and_derivatives.argtypes = [_dp, i, i, i, d, _dp]
    L.tpo_eval_curve.restype = i
    L.tpo_eval_curve.argtypes = [_dp, i, i, _dp, i, d, _dp]
    L.tpo_eval_curve_and_derivatives.restype = i
    L.tpo_eval_curve_and_derivatives.argtypes = [_dp, i, i, _dp, i, d, i, _dp]
    L.tpo_make_uniform_knots.restype = i
    L.tpo_make_uniform_knots.argtypes = [i, i, d, d, _dp]
    L.tpo_polyline_to_bspline3_waypoints.restype = i
    L.tpo_polyline_to_bspline3_waypoints.argtypes = [_dp, i, i, d, _dp]
    L.tpo_joint_fit_spline.restype = i
    L.tpo_joint_fit_spline.argtypes = [_dp, i, i, d, _dp, _dp]
    L.tpo_joint_sample_path.restype = i
    L.tpo_joint_sample_path.argtypes = [_dp, i, _dp, i, i, d, d, i, _dp, _dp, _dp]
    L.tpo_joint_constraint_setup.argtypes = [_dp, _dp, i, i, _dp, _dp, d, _dp, _dp, _dp, _dp]
    L.tpo_cartesian_path_derivatives.argtypes = [_dp, i, i, d, _dp, _dp]
    L.tpo_cartesian_constraint_setup.argtypes = [_dp, _dp, _dp, i, i, _dp, _dp, d, d, d,
                                                 _dp, _dp, _dp, _dp]
    L.tpo_profile_create.restype = vp
    L.tpo_profile_create.argtypes = [i, i]
    L.tpo_profile_destroy.argtypes = [vp]
    L.tpo_profile_set_max_loops.argtypes = [vp, i]
    L.tpo_profile_setup.restype = i
    L.tpo_profile_setup.argtypes = [vp, _dp, _dp, _dp, _dp, d, d, d, d, d]
    L.tpo_profile_optimize.restype = i
    L.tpo_profile_optimize.argtypes = [vp]
    L.tpo_profile_calculate_boundary.restype = i
    L.tpo_profile_calculate_boundary.argtypes = [vp]
    for name in ("time", "s", "sd", "sdd", "sd2", "sd2_max", "sdd_max_for_sd2_max",
                 "sdd_min_for_sd2_max", "sd2_max_for_sdd0"):
        f = getattr(L, "tpo_profile_" + name)
        f.restype = C.POINTER(d)
        f.argtypes = [vp]
    L.tpo_profile_boundary_type.restype = C.POINTER(C.c_uint8)
    L.tpo_profile_boundary_type.argtypes = [vp]
    L.tpo_profile_last_extremal_index.restype = i
    L.tpo_profile_last_extremal_index.argtypes = [vp]
    L.tpo_profile_max_time_increment.restype = d
    L.tpo_profile_max_time_increment.argtypes = [vp]
    L.tpo_profile_num_loops_used.restype = i
    L.tpo_profile_num_loops_used.argtypes = [vp]
    L.tpo_profile_constraint_violations.restype = i
    L.tpo_profile_constraint_violations.argtypes = [vp]
    L.tpo_profile_query.restype = i
    L.tpo_profile_query.argtypes = [vp, d, C.POINTER(d), C.POINTER(d), C.POINTER(d)]
    L.tpo_profile_previous_index.restype = i
    L.tpo_profile_previous_index.argtypes = [vp, d]
    for name in ("simplex", "bruteforce"):
        f = getattr(L, "tpo_find_max_sd2_" + name)
        f.argtypes = [_dp, _dp, _dp, _dp, i, C.POINTER(d), C.POINTER(d), C.POINTER(d)]
    for name in ("max", "min"):
        f = getattr(L, "tpo_find_sdd_" + name)
        f.restype = d
        f.argtypes = [_dp, _dp, _dp, _dp, i, d]
    L.tpo_epilogue.argtypes = [_dp, _dp, i, i, _dp, _dp, _dp, _dp, _dp]
    L.tpo_resample_uniform.restype = i
    L.tpo_resample_uniform.argtypes = [_dp] * 7 + [i, i, d, d, _dp, i] + [_dp] * 7
    L.tpo_resample_skip.restype = i
    L.tpo_resample_skip.argtypes = [_dp] * 7 + [i, i, d, d, _dp, i] + [_dp] * 7
    L.tpo_resample_uniform_count.restype = i
    L.tpo_resample_uniform_count.argtypes = [d, d, d]
    L.tpo_time_joint_path.restype = i
    L.tpo_time_joint_path.argtypes = [_dp, i, _dp, i, i, _dp, _dp, d, d, d, i, d, d, d, vp,
                                      _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(i)]
    L.tpo_time_joint_batch.restype = i
    L.tpo_time_joint_batch.argtypes = [i, _dp, i, _dp, i, i, _dp, _dp, d, _dp, _dp, i, _dp, _dp,
                                       i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip]
    L.tpo_time_cartesian_batch.restype = i
    L.tpo_time_cartesian_batch.argtypes = [i, _dp, _dp, i, i, _dp, _dp, _dp, _dp, d, _dp, _dp, _dp,
                                           _dp, i, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip]
    _LIB = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ------------------------------------------------------------------ splines
def knot_span(knots, degree, u):
    k = _f64(knots)
    return lib().tpo_knot_span(k, len(k), degree, float(u))


def basis(knots, span, p, u):
    out = np.zeros(p + 1)
    lib().tpo_basis(_f64(knots), span, p, float(u), out)
    return out


def basis_and_derivatives(knots, span, p, der, u):
    out = np.zeros((der + 1, p + 1))
    lib().tpo_basis_and_derivatives(_f64(knots), span, p, der, float(u), out)
    return out


def eval_curve(knots, degree, points, u):
    k, pts = _f64(knots), _f64(points)
    out = np.zeros(pts.shape[1])
    rc = lib().tpo_eval_curve(k, len(k), degree, pts, pts.shape[1], float(u), out)
    return rc, out


def eval_curve_and_derivatives(knots, degree, points, u, nvalues):
    k, pts = _f64(knots), _f64(points)
    out = np.zeros((nvalues, pts.shape[1]))
    rc = lib().tpo_eval_curve_and_derivatives(k, len(k), degree, pts, pts.shape[1],
                                              float(u), nvalues, out)
    return rc, out


def make_uniform_knots(num_points, degree, low=0.0, high=1.0):
    out = np.zeros(num_points + degree + 1)
    rc = lib().tpo_make_uniform_knots(num_points, degree, low, high, out)
    return rc, out


def polyline_to_bspline3_waypoints(corners, radius):
    c = _f64(corners)
    W, D = c.shape
    out = np.zeros((max(3 * W - 2, 4), D))
    n = lib().tpo_polyline_to_bspline3_waypoints(c, W, D, float(radius), out)
    return out[:n]


def joint_fit_spline(waypoints, rounding=0.2):
    w = _f64(waypoints)
    W, D = w.shape
    P = 3 * W - 2 if W > 1 else 4
    cps = np.zeros((P, D))
    knots = np.zeros(P + 3)
    lib().tpo_joint_fit_spline(w, W, D, float(rounding), cps, knots)
    return cps, knots


def joint_sample_path(knots, cps, path_start, delta, N):
    k, c = _f64(knots), _f64(cps)
    P, D = c.shape
    q, q1, q2 = np.zeros((N, D)), np.zeros((N, D)), np.zeros((N, D))
    rc = lib().tpo_joint_sample_path(k, len(k), c, P, D, float(path_start), float(delta),
                                     N, q, q1, q2)
    assert rc == 0, rc
    return q, q1, q2


def joint_constraint_setup(q1, q2, vmax, amax, safety=0.8):
    q1, q2 = _f64(q1), _f64(q2)
    N, D = q1.shape
    A, B, lo, hi = (np.zeros((N, 2 * D)) for _ in range(4))
    lib().tpo_joint_constraint_setup(q1, q2, N, D, _f64(vmax), _f64(amax), float(safety),
                                     A, B, lo, hi)
    return A, B, lo, hi


def cartesian_path_derivatives(q, delta):
    q = _f64(q)
    N, D = q.shape
    q1, q2 = np.zeros((N, D)), np.zeros((N, D))
    lib().tpo_cartesian_path_derivatives(q, N, D, float(delta), q1, q2)
    return q1, q2


def cartesian_constraint_setup(q1, q2, jq1, vmax, amax, vtrans, vrot, safety=0.8):
    q1, q2, jq1 = _f64(q1), _f64(q2), _f64(jq1)
    N, D = q1.shape
    A, B, lo, hi = (np.zeros((N, 2 * D + 2)) for _ in range(4))
    lib().tpo_cartesian_constraint_setup(q1, q2, jq1, N, D, _f64(vmax), _f64(amax),
                                         float(vtrans), float(vrot), float(safety),
                                         A, B, lo, hi)
    return A, B, lo, hi


# ------------------------------------------------------------------- solver
class Profile:
    """Mirror of TimeOptimalPathProfile on the oracle."""

    def __init__(self, N, Cn):
        self.N, self.C = N, Cn
        self._p = lib().tpo_profile_create(N, Cn)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().tpo_profile_destroy(self._p)
            self._p = None

    def set_max_loops(self, n):
        lib().tpo_profile_set_max_loops(self._p, int(n))

    def setup(self, A, B, lo, hi, s_start, s_end, sd_start=0.0, sdd_start=0.0, t_start=0.0):
        return lib().tpo_profile_setup(self._p, _f64(A), _f64(B), _f64(lo), _f64(hi),
                                       s_start, s_end, sd_start, sdd_start, t_start)

    def optimize(self):
        return lib().tpo_profile_optimize(self._p)

    def calculate_boundary(self):
        return lib().tpo_profile_calculate_boundary(self._p)

    def _arr(self, name):
        ptr = getattr(lib(), "tpo_profile_" + name)(self._p)
        return np.ctypeslib.as_array(ptr, shape=(self.N,)).copy()

    time = property(lambda self: self._arr("time"))
    s = property(lambda self: self._arr("s"))
    sd = property(lambda self: self._arr("sd"))
    sdd = property(lambda self: self._arr("sdd"))
    sd2 = property(lambda self: self._arr("sd2"))
    sd2_max = property(lambda self: self._arr("sd2_max"))
    sdd_max_for_sd2_max = property(lambda self: self._arr("sdd_max_for_sd2_max"))
    sdd_min_for_sd2_max = property(lambda self: self._arr("sdd_min_for_sd2_max"))
    sd2_max_for_sdd0 = property(lambda self: self._arr("sd2_max_for_sdd0"))

    @property
    def boundary_type(self):
        ptr = lib().tpo_profile_boundary_type(self._p)
        return np.ctypeslib.as_array(ptr, shape=(self.N,)).copy()

    @property
    def last_extremal_index(self):
        return lib().tpo_profile_last_extremal_index(self._p)

    @property
    def max_time_increment(self):
        return lib().tpo_profile_max_time_increment(self._p)

    @property
    def loops_used(self):
        return lib().tpo_profile_num_loops_used(self._p)

    def constraint_violations(self):
        return lib().tpo_profile_constraint_violations(self._p)

    def query(self, t):
        s, sd, sdd = C.c_double(), C.c_double(), C.c_double()
        ok = lib().tpo_profile_query(self._p, float(t), C.byref(s), C.byref(sd), C.byref(sdd))
        return bool(ok), s.value, sd.value, sdd.value

    def previous_index(self, t):
        return lib().tpo_profile_previous_index(self._p, float(t))


def _lp(fn, A, B, lo, hi):
    A, B, lo, hi = _f64(A), _f64(B), _f64(lo), _f64(hi)
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    fn(A, B, lo, hi, len(A), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def find_max_sd2_simplex(A, B, lo, hi):
    return _lp(lib().tpo_find_max_sd2_simplex, A, B, lo, hi)


def find_max_sd2_bruteforce(A, B, lo, hi):
    return _lp(lib().tpo_find_max_sd2_bruteforce, A, B, lo, hi)


def find_sdd_max(A, B, lo, hi, sd2):
    A = _f64(A)
    return lib().tpo_find_sdd_max(A, _f64(B), _f64(lo), _f64(hi), len(A), float(sd2))


def find_sdd_min(A, B, lo, hi, sd2):
    A = _f64(A)
    return lib().tpo_find_sdd_min(A, _f64(B), _f64(lo), _f64(hi), len(A), float(sd2))


# ------------------------------------------------------- epilogue / resample
def epilogue(q1, q2, sd, sdd, amax):
    q1, q2 = _f64(q1), _f64(q2)
    N, D = q1.shape
    qd, qdd = np.zeros((N, D)), np.zeros((N, D))
    lib().tpo_epilogue(q1, q2, N, D, _f64(sd), _f64(sdd), _f64(amax), qd, qdd)
    return qd, qdd


def resample_uniform(t, s, sd, sdd, q, qd, qdd, start_sec, time_step, amax):
    q = _f64(q)
    N, D = q.shape
    M = lib().tpo_resample_uniform_count(float(t[-1]), float(start_sec), float(time_step))
    ot, os_, osd, osdd = (np.zeros(M) for _ in range(4))
    oq, oqd, oqdd = (np.zeros((M, D)) for _ in range(3))
    lib().tpo_resample_uniform(_f64(t), _f64(s), _f64(sd), _f64(sdd), q, _f64(qd), _f64(qdd),
                               N, D, float(start_sec), float(time_step), _f64(amax), M,
                               ot, os_, osd, osdd, oq, oqd, oqdd)
    return ot, os_, osd, osdd, oq, oqd, oqdd


def resample_skip(t, s, sd, sdd, q, qd, qdd, start_sec, time_step, amax):
    """kSkipSamplesCloserThanTimeStep resample; returns the kept samples only."""
    q = _f64(q)
    N, D = q.shape
    cap = N + 1
    ot, os_, osd, osdd = (np.zeros(cap) for _ in range(4))
    oq, oqd, oqdd = (np.zeros((cap, D)) for _ in range(3))
    M = lib().tpo_resample_skip(_f64(t), _f64(s), _f64(sd), _f64(sdd), q, _f64(qd), _f64(qdd),
                                N, D, float(start_sec), 0.95 * float(time_step), _f64(amax), cap,
                                ot, os_, osd, osdd, oq, oqd, oqdd)
    return tuple(a[:M] for a in (ot, os_, osd, osdd, oq, oqd, oqdd))


# ------------------------------------------------------------ whole hot path
def time_joint_batch(knots, cps, vmax, amax, path_start, delta, N, sd_start=None,
                     time_start=None, safety=0.8, nthreads=1):
    """knots [B][K], cps [B][P][D], vmax/amax [B][D], path_start/delta [B]."""
    knots, cps = _f64(knots), _f64(cps)
    B, K = knots.shape
    _, P, D = cps.shape
    path_start = _f64(np.broadcast_to(path_start, (B,)))
    delta = _f64(np.broadcast_to(delta, (B,)))
    sd_start = _f64(np.zeros(B) if sd_start is None else np.broadcast_to(sd_start, (B,)))
    time_start = _f64(np.zeros(B) if time_start is None else np.broadcast_to(time_start, (B,)))
    t, s, sd, sdd = (np.zeros((B, N)) for _ in range(4))
    q, qd, qdd = (np.zeros((B, N, D)) for _ in range(3))
    lei = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    lib().tpo_time_joint_batch(B, knots, K, cps, P, D, _f64(vmax), _f64(amax), float(safety),
                               path_start, delta, N, sd_start, time_start, int(nthreads),
                               t, s, sd, sdd, q, qd, qdd, lei, status)
    return dict(t=t, s=s, sd=sd, sdd=sdd, q=q, qd=qd, qdd=qdd,
                last_extremal_index=lei, status=status)


def time_cartesian_batch(q, J, vmax, amax, vtrans, vrot, path_start, delta, sd_start=None,
                         time_start=None, safety=0.8, nthreads=1):
    """IK positions q [B][N][D], Jacobians J [B][N][6][D] -> timing as time_joint_batch."""
    q, J = _f64(q), _f64(J)
    B, N, D = q.shape
    assert J.shape == (B, N, 6, D)
    bc = lambda v: _f64(np.broadcast_to(v, (B,)))
    sd_start = bc(0.0 if sd_start is None else sd_start)
    time_start = bc(0.0 if time_start is None else time_start)
    t, s, sd, sdd = (np.zeros((B, N)) for _ in range(4))
    qd, qdd = (np.zeros((B, N, D)) for _ in range(2))
    lei = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    lib().tpo_time_cartesian_batch(B, q, J, N, D, _f64(vmax), _f64(amax), bc(vtrans), bc(vrot),
                                   float(safety), bc(path_start), bc(delta), sd_start, time_start,
                                   int(nthreads), t, s, sd, sdd, qd, qdd, lei, status)
    return dict(t=t, s=s, sd=sd, sdd=sdd, q=q, qd=qd, qdd=qdd, last_extremal_index=lei,
                status=status)


# ------------------------------------------------------- receding-horizon planner
class Planner:
    """tp_oracle_plan.c: PathTimingTrajectory::Plan (path_timing_trajectory.cc:579-684) for one
    planner with a joint-space spline path. Times in integer nanoseconds."""
    STATUS = {0: "ok", 1: "failed_precondition", 2: "out_of_range", 3: "invalid_argument",
              4: "internal", 5: "deadline_exceeded"}

    def __init__(self, D, N, delta=0.005, safety=0.8, time_step_ns=4_000_000, skip=False,
                 max_planning_iterations=10000, max_initial_velocity_error=1e-3):
        L = lib()
        vp, i, d, i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
        L.tpo_planner_create.restype = vp
        L.tpo_planner_create.argtypes = [i, i, d, d, i64, i, i, d]
        L.tpo_planner_destroy.argtypes = [vp]
        L.tpo_planner_set_limits.argtypes = [vp, _dp, _dp]
        L.tpo_planner_set_initial_velocity.argtypes = [vp, _dp]
        L.tpo_planner_set_spline.argtypes = [vp, _dp, i, _dp, i, i]
        L.tpo_planner_plan.restype = i
        L.tpo_planner_plan.argtypes = [vp, i64, i64]
        for name in ("num_samples", "target_reached", "windows", "path_state"):
            getattr(L, "tpo_planner_" + name).restype = i
            getattr(L, "tpo_planner_" + name).argtypes = [vp]
        for name in ("end_time", "final_decel_start"):
            getattr(L, "tpo_planner_" + name).restype = i64
            getattr(L, "tpo_planner_" + name).argtypes = [vp]
        for name in ("time", "positions", "velocities", "accelerations", "path_parameter",
                     "path_velocity", "path_acceleration"):
            getattr(L, "tpo_planner_" + name).restype = C.POINTER(C.c_double)
            getattr(L, "tpo_planner_" + name).argtypes = [vp]
        self._L, self.D, self.N = L, D, N
        self._p = L.tpo_planner_create(D, N, float(delta), float(safety), int(time_step_ns),
                                       1 if skip else 0, int(max_planning_iterations),
                                       float(max_initial_velocity_error))

    def __del__(self):
        if getattr(self, "_p", None):
            self._L.tpo_planner_destroy(self._p)
            self._p = None

    def set_limits(self, vmax, amax):
        self._L.tpo_planner_set_limits(self._p, _f64(vmax), _f64(amax))

    def set_waypoints(self, waypoints, rounding=0.2, state=1):
        """SetWaypoints: fit the spline (timeable_path_joint_spline.cc:252-292), state kNewPath."""
        cps, knots = joint_fit_spline(waypoints, rounding)
        self.set_spline(knots, cps, state)
        return cps, knots

    def set_spline(self, knots, cps, state=1):
        knots, cps = _f64(knots), _f64(cps)
        self._L.tpo_planner_set_spline(self._p, knots, len(knots), cps, cps.shape[0], int(state))

    def plan(self, start_ns, horizon_ns):
        return self._L.tpo_planner_plan(self._p, int(start_ns), int(horizon_ns))

    def _vec(self, name, width=1):
        n = self.num_samples * width
        ptr = getattr(self._L, "tpo_planner_" + name)(self._p)
        a = np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0)
        return a.reshape(-1, width) if width > 1 else a

    num_samples = property(lambda self: self._L.tpo_planner_num_samples(self._p))
    time = property(lambda self: self._vec("time"))
    positions = property(lambda self: self._vec("positions", self.D))
    velocities = property(lambda self: self._vec("velocities", self.D))
    accelerations = property(lambda self: self._vec("accelerations", self.D))
    path_parameter = property(lambda self: self._vec("path_parameter"))
    end_time = property(lambda self: self._L.tpo_planner_end_time(self._p))
    final_decel_start = property(lambda self: self._L.tpo_planner_final_decel_start(self._p))
    target_reached = property(lambda self: bool(self._L.tpo_planner_target_reached(self._p)))
    windows = property(lambda self: self._L.tpo_planner_windows(self._p))


# ------------------------------------------------------------ quaternion splines
def _q4(fn, q, *extra):
    out = np.zeros(4)
    getattr(lib(), fn)(_f64(q), *extra, out)
    return out


def quat_log(q):
    lib().tpo_quat_log.argtypes = [_dp, _dp]
    return _q4("tpo_quat_log", q)


def quat_exp(q):
    lib().tpo_quat_exp.argtypes = [_dp, _dp]
    return _q4("tpo_quat_exp", q)


def quat_power(q, power):
    lib().tpo_quat_power.argtypes = [_dp, C.c_double, _dp]
    return _q4("tpo_quat_power", q, float(power))


def bsplineq_eval_curve(knots, degree, points, u):
    L = lib()
    L.tpo_bsplineq_eval_curve.restype = C.c_int
    L.tpo_bsplineq_eval_curve.argtypes = [_dp, C.c_int, C.c_int, _dp, C.c_double, _dp]
    k, p = _f64(knots), _f64(points)
    out = np.zeros(4)
    rc = L.tpo_bsplineq_eval_curve(k, len(k), int(degree), p, float(u), out)
    if rc != 0:
        raise ValueError("parameter outside the knot range")
    return out


def sample_pose_spline(knots, translation_points, rotation_points, path_start, delta, N):
    L = lib()
    L.tpo_sample_pose_spline.restype = C.c_int
    L.tpo_sample_pose_spline.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                         C.c_int, _dp]
    k, t, r = _f64(knots), _f64(translation_points), _f64(rotation_points)
    out = np.zeros((N, 7))
    rc = L.tpo_sample_pose_spline(k, len(k), t, r, t.shape[0], float(path_start), float(delta), int(N), out)
    if rc != 0:
        raise ValueError("pose spline evaluation failed")
    return out
