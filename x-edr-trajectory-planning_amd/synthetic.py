"""Seeded synthetic joint-space B-spline paths.

Host-side input generation, excluded from every timed region. The waypoint ->
control-point rule restates the vector variant of the reference's
PolyLineToBspline3Waypoints (splines/spline_utils.cc:25-102) and the knot
construction of TimeableJointSplinePath::FitSplineToWaypoints
(timeable_path_joint_spline.cc:252-292): degree 2, uniform knots built by
running accumulation (splines/bspline_base.cc:376-378), scaled by the control
polygon length. Norms are summed sequentially over the joints (Eigen's packet
reduction order is build dependent; the generator fixes one order).

RNG: splitmix64 seeded with 0x5EEDC0DE00000000 + path_index; doubles are
(x >> 11) * 2**-53. Per path the draw order is: W*D waypoint coordinates
(waypoint-major) in [-2, 2], D velocity limits in [1, 2], D acceleration limits
in [2, 4].
"""
import numpy as np

SEED_BASE = 0x5EEDC0DE00000000
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix_stream(path_index, count):
    """[len(path_index)][count] uniform doubles in [0,1)."""
    idx = np.asarray(path_index, dtype=np.uint64)
    state = (np.uint64(SEED_BASE) + idx) & _M64
    out = np.empty((idx.shape[0], count), dtype=np.float64)
    with np.errstate(over="ignore"):
        for k in range(count):
            state = (state + np.uint64(0x9E3779B97F4A7C15)) & _M64
            z = state.copy()
            z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
            z = z ^ (z >> np.uint64(31))
            out[:, k] = (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return out


def _norm_seq(v):
    """Euclidean norm over the last axis with a sequential sum."""
    acc = np.zeros(v.shape[:-1])
    for d in range(v.shape[-1]):
        acc = acc + v[..., d] * v[..., d]
    return np.sqrt(acc)


def _corner_offset(delta, radius):
    """spline_utils.cc:25-45, batched over leading axes."""
    k_min_norm = 1e-6
    spacing = 4.0  # kMinWaypointSpacingFactor, spline_utils.h:44-46
    norm = _norm_seq(delta)
    safe = np.where(norm > k_min_norm, norm, 1.0)
    unit = np.where((norm > k_min_norm)[..., None], delta / safe[..., None], 0.0)
    far = (norm > spacing * radius)[..., None]
    return np.where(far, unit * radius, unit * (1.0 / spacing) * norm[..., None])


def polyline_to_control_points(waypoints, radius):
    """[B][W][D] corners -> [B][3W-2][D] control points (W >= 2)."""
    wp = np.asarray(waypoints, dtype=np.float64)
    B, W, D = wp.shape
    assert W >= 2
    P = 3 * W - 2
    out = np.zeros((B, P, D))
    out[:, 0::3] = wp
    if W > 2:
        k = 3 * np.arange(1, W - 1)
        out[:, k + 1] = out[:, k] + _corner_offset(out[:, k + 3] - out[:, k], radius)
        out[:, k - 1] = out[:, k] + _corner_offset(out[:, k - 3] - out[:, k], radius)
    out[:, 1] = out[:, 0] + _corner_offset(out[:, 3] - out[:, 0], radius)
    out[:, P - 2] = out[:, P - 1] + _corner_offset(out[:, P - 4] - out[:, P - 1], radius)
    return out


def uniform_knots(num_points, degree=2):
    """bspline_base.cc:350-381 with low=0, high=1."""
    nk = num_points + degree + 1
    knots = np.zeros(nk)
    spacing = (1.0 / (nk - 2.0 * (degree + 1.0) + 1.0)) * (1.0 - 0.0)
    for i in range(degree + 1, nk - degree - 1):
        knots[i] = knots[i - 1] + spacing
    knots[nk - degree - 1:] = 1.0
    return knots


def fit_joint_splines(waypoints, rounding=0.2):
    """[B][W][D] waypoints -> (control points [B][P][D], knots [B][P+3])."""
    cps = polyline_to_control_points(waypoints, rounding)
    B, P, D = cps.shape
    length = np.zeros(B)
    for i in range(P - 1):
        length = length + _norm_seq(cps[:, i + 1] - cps[:, i])
    weighted = np.maximum(length * 1.0, 0.1)
    knots = uniform_knots(P, 2)[None, :] * weighted[:, None]
    return cps, knots


def make_joint_batch(num_paths, num_dofs=7, num_samples=2000, num_waypoints=10,
                     first_path_index=0, rounding=0.2, safety=0.8, path_indices=None):
    """The batch definition of SURVEY.md section 8(d).

    Returns a dict of host numpy arrays:
      control_points [B][P][D], knots [B][P+3], vmax [B][D], amax [B][D],
      path_start [B], delta [B], sd_start [B], time_start [B]
    plus scalars num_samples, safety.
    """
    B, D, W = int(num_paths), int(num_dofs), int(num_waypoints)
    if path_indices is not None:       # explicit path indices (mixed batches) instead of a range
        idx = np.asarray(path_indices, dtype=np.uint64)
        assert idx.shape == (B,)
    else:
        idx = np.arange(first_path_index, first_path_index + B, dtype=np.uint64)
    u = _splitmix_stream(idx, W * D + 2 * D)
    waypoints = (u[:, :W * D] * 4.0 - 2.0).reshape(B, W, D)
    vmax = 1.0 + u[:, W * D:W * D + D]
    amax = 2.0 + 2.0 * u[:, W * D + D:W * D + 2 * D]
    cps, knots = fit_joint_splines(waypoints, rounding)
    delta = knots[:, -1] / (num_samples - 1)
    return dict(
        control_points=np.ascontiguousarray(cps), knots=np.ascontiguousarray(knots),
        vmax=np.ascontiguousarray(vmax), amax=np.ascontiguousarray(amax),
        path_start=np.zeros(B), delta=np.ascontiguousarray(delta),
        sd_start=np.zeros(B), time_start=np.zeros(B),
        num_samples=int(num_samples), safety=float(safety), waypoints=waypoints,
    )


MIXED_DOFS = (6, 7, 14)
MIXED_SAMPLES = (500, 4000)
MIXED_BUCKET = 512


def mixed_batch_shape(num_paths, first_path_index=0):
    """BASELINE.json configs[4] (SURVEY.md 8d "Config 5"): per path a joint count drawn from
    {6, 7, 14} and a sample count uniform in [500, 4000], from the path's own splitmix64
    stream (salted so that it is independent of the waypoint draws).
    Returns (dofs [B] int32, samples [B] int32)."""
    idx = np.arange(first_path_index, first_path_index + int(num_paths), dtype=np.uint64)
    u = _splitmix_stream(idx + np.uint64(1 << 41), 2)
    dofs = np.asarray(MIXED_DOFS, dtype=np.int32)[np.minimum((u[:, 0] * 3).astype(np.int64), 2)]
    lo, hi = MIXED_SAMPLES
    samples = (lo + np.floor(u[:, 1] * (hi - lo + 1))).astype(np.int32)
    return dofs, np.minimum(samples, hi)


def mixed_batch_groups(dofs, samples, bucket=MIXED_BUCKET):
    """Bucket a mixed batch by (D, ceil(N / bucket)): one engine launch per bucket, with the
    bucket's upper edge as the common sample stride. Returns {(D, stride): positions}."""
    dofs, samples = np.asarray(dofs), np.asarray(samples)
    groups = {}
    for pos in range(len(dofs)):
        key = (int(dofs[pos]), int(-(-int(samples[pos]) // bucket) * bucket))
        groups.setdefault(key, []).append(pos)
    return {k: np.asarray(v, dtype=np.int64) for k, v in sorted(groups.items())}


def make_mixed_group(path_indices, num_dofs, samples, stride):
    """The joint batch of one bucket: paths `path_indices` (global indices), each sampled at
    its own count samples[i] <= stride over its whole length."""
    b = make_joint_batch(len(path_indices), num_dofs, stride, path_indices=path_indices)
    ns = np.asarray(samples, dtype=np.int32)
    b["delta"] = np.ascontiguousarray(b["knots"][:, -1] / (ns - 1))
    b["num_samples_per_path"] = ns
    return b


def eval_joint_splines(cps, knots, u):
    """Positions of degree-2 splines at parameters u [B][N] -> [B][N][D] (numpy; only used to
    make synthetic IK positions, so it need not match the engine bit for bit)."""
    cps, knots, u = np.asarray(cps), np.asarray(knots), np.asarray(u)
    B, P, D = cps.shape
    u = np.clip(u, knots[:, :1], knots[:, -1:])
    span = np.empty(u.shape, dtype=np.int64)
    for b in range(B):
        span[b] = np.clip(np.searchsorted(knots[b], u[b], side="right") - 1, 2, P - 1)
    bi = np.arange(B)[:, None]
    k0, k1, k2, k3 = (knots[bi, span + o] for o in (-1, 0, 1, 2))
    # Cox-de Boor, degree 2, on the span [k1, k2)
    w1 = (u - k1) / (k2 - k1)
    n0 = (1.0 - w1) * (k2 - u) / (k2 - k0)
    n2 = w1 * (u - k1) / (k3 - k1)
    n1 = 1.0 - n0 - n2
    return (n0[..., None] * cps[bi, span - 2] + n1[..., None] * cps[bi, span - 1] +
            n2[..., None] * cps[bi, span])


def make_cartesian_batch(num_paths, num_dofs=6, num_samples=2000, num_waypoints=10,
                         first_path_index=0, safety=0.8):
    """Synthetic stand-in for BASELINE.json configs[3]: what a TimeableCartesianSplinePath holds
    after its IK callback ran -- IK positions [B][N][D] (here: a joint-space spline sampled
    uniformly) and the Jacobian at every sample [B][N][6][D] (here: a smooth function of q
    with a unit block) -- plus joint and Cartesian velocity limits."""
    jb = make_joint_batch(num_paths, num_dofs, num_samples, num_waypoints, first_path_index,
                          safety=safety)
    B, D, N = int(num_paths), int(num_dofs), int(num_samples)
    u = jb["delta"][:, None] * np.arange(N)[None, :]
    q = eval_joint_splines(jb["control_points"], jb["knots"], u)
    r = np.arange(6)[None, None, :, None]
    d = np.arange(D)[None, None, None, :]
    J = 0.25 * np.sin(q[:, :, None, :] * (r + 1.0) + 0.37 * d) + (r == d % 6)
    idx = np.arange(first_path_index, first_path_index + B, dtype=np.uint64)
    lim = _splitmix_stream(idx + np.uint64(1 << 40), 2)
    return dict(
        ik_positions=np.ascontiguousarray(q), jacobians=np.ascontiguousarray(J),
        vmax=jb["vmax"], amax=jb["amax"], vtrans=0.6 + 0.9 * lim[:, 0], vrot=0.8 + 1.2 * lim[:, 1],
        path_start=np.zeros(B), delta=jb["delta"], sd_start=np.zeros(B), time_start=np.zeros(B),
        num_samples=N, safety=float(safety))


def upload_cartesian_batch(batch, device):
    import torch
    names = dict(ik_positions="ik_positions", jacobians="jacobians", vmax="max_velocity",
                 amax="max_acceleration", vtrans="max_translational_velocity",
                 vrot="max_rotational_velocity", path_start="path_start", delta="delta",
                 sd_start="sd_start", time_start="time_start")
    return {v: torch.from_numpy(np.ascontiguousarray(batch[k])).to(device)
            for k, v in names.items()}
