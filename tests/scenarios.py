"""Closed-form constraint scenarios of the reference's solver tests, restated as
formulas (trajectory_planning/time_optimal_path_timing_test.cc:49-74, :125-158,
:216-247, :299-322, :362-389). Rows are returned as (A, B, lower, upper), each
of shape [N][C]."""
import numpy as np


def _s(n, s0, s1):
    ds = (s1 - s0) / (n - 1)
    return np.arange(n) * ds + s0


def sine(n, s0, s1, R, vmax, amax):
    s = _s(n, s0, s1)
    A = np.stack([-R * np.sin(s), np.zeros(n)], 1)
    B = np.stack([-R * np.cos(s), (R * np.sin(s)) ** 2], 1)
    hi = np.tile([amax, vmax ** 2], (n, 1))
    lo = np.tile([-amax, 0.0], (n, 1))
    return A, B, lo, hi


def circle(n, s0, s1, R, vmax, amax):
    s = _s(n, s0, s1)
    z = np.zeros(n)
    A = np.stack([-R * np.sin(s), R * np.cos(s), z, z], 1)
    B = np.stack([-R * np.cos(s), -R * np.sin(s), (R * np.sin(s)) ** 2, (R * np.cos(s)) ** 2], 1)
    hi = np.tile([amax, amax, vmax ** 2, vmax ** 2], (n, 1))
    lo = np.tile([-amax, -amax, 0.0, 0.0], (n, 1))
    return A, B, lo, hi


def line(n, slope, vmax, amax):
    A = np.tile([1.0, slope, 0.0, 0.0], (n, 1))
    B = np.tile([0.0, 0.0, slope ** 2, 1.0], (n, 1))
    hi = np.tile([amax, amax, vmax ** 2, vmax ** 2], (n, 1))
    lo = np.tile([-amax, -amax, 0.0, 0.0], (n, 1))
    return A, B, lo, hi


def scalar_straight(n, vmax, amax):
    A = np.tile([1.0, 0.0], (n, 1))
    B = np.tile([0.0, 1.0], (n, 1))
    hi = np.tile([amax, vmax ** 2], (n, 1))
    lo = np.tile([-amax, 0.0], (n, 1))
    return A, B, lo, hi


def scalar_curved(n, s0, s1, m, vmax, amax):
    s = _s(n, s0, s1)
    d1 = 3.0 * m[0] * s * s + 2.0 * m[1] * s + m[2]
    A = np.stack([d1, np.zeros(n)], 1)
    B = np.stack([6.0 * m[0] * s + 2.0 * m[1], d1 ** 2.0], 1)
    hi = np.tile([amax, vmax], (n, 1))   # sic: upper(1) = vmax, not vmax^2 (:382)
    lo = np.tile([-amax, 0.0], (n, 1))
    return A, B, lo, hi


def all_cases():
    """(name, rows, s0, s1, sd_start, meta) for every case the reference's tests run."""
    cases = []
    for off in (0.0, np.pi / 2, np.pi / 4):
        for n in (30, 31, 100, 111):
            cases.append(("sine_off%.3f_n%d" % (off, n),
                          sine(n, off, np.pi + off, 2.0, 1.2, 1.0), off, np.pi + off, 0.0,
                          dict(kind="sine", R=2.0, vmax=1.2, amax=1.0)))
    for n, sd0 in ((50, 0.0), (51, 0.0), (50, 0.1), (51, 0.1)):
        cases.append(("circle_n%d_sd%.1f" % (n, sd0), circle(n, 0.0, np.pi, 2.0, 1.2, 1.0),
                      0.0, np.pi, sd0, dict(kind="circle", R=2.0, vmax=1.2, amax=1.0)))
    cases.append(("line", line(30, 2.0, 1.0, 1.0), 0.0, np.pi, 0.0,
                  dict(kind="line", slope=2.0, vmax=1.0, amax=1.0)))
    cases.append(("scalar_straight", scalar_straight(30, 0.5, 1.0), 0.0, 1.0, 0.0,
                  dict(kind="straight", vmax=0.5, amax=1.0)))
    cases.append(("scalar_curved", scalar_curved(100, -3.0, 1.0, (1.0, 1.0, 2.0), 1.0, 0.2),
                  -3.0, 1.0, 0.0, dict(kind="curved", m=(1.0, 1.0, 2.0), vmax=1.0, amax=0.2)))
    return cases


def max_violation(meta, s, sd, sdd):
    """Largest excess of the physical velocity/acceleration over the box limits,
    as the reference's Verify*ExampleSolution helpers compute it (:80-118,
    :164-210, :253-294, :327-356, :394-438)."""
    k = meta["kind"]
    if k == "sine":
        R = meta["R"]
        xd = -R * np.sin(s) * sd
        xdd = -R * np.sin(s) * sdd - R * np.cos(s) * sd * sd
        return max((abs(xd) - meta["vmax"]).max(), (abs(xdd) - meta["amax"]).max())
    if k == "circle":
        R = meta["R"]
        xd, yd = -R * np.sin(s) * sd, R * np.cos(s) * sd
        xdd = -R * np.sin(s) * sdd - R * np.cos(s) * sd * sd
        ydd = R * np.cos(s) * sdd - R * np.sin(s) * sd * sd
        return max((abs(xd) - meta["vmax"]).max(), (abs(xdd) - meta["amax"]).max(),
                   (abs(yd) - meta["vmax"]).max(), (abs(ydd) - meta["amax"]).max())
    if k == "line":
        sl = meta["slope"]
        return max((abs(sd) - meta["vmax"]).max(), (abs(sdd) - meta["amax"]).max(),
                   (abs(sl * sd) - meta["vmax"]).max(), (abs(sl * sdd) - meta["amax"]).max())
    if k == "straight":
        return max((abs(sd) - meta["vmax"]).max(), (abs(sdd) - meta["amax"]).max())
    m = meta["m"]
    xd = 3 * m[0] * s ** 2 * sd + 2 * m[1] * s * sd + m[2] * sd
    xdd = (6 * m[0] * s * sd ** 2 + 3 * m[0] * s ** 2 * sdd + 2 * m[1] * sd ** 2
           + 2 * m[1] * s * sdd + m[2] * sdd)
    return max((abs(xdd) - meta["amax"]).max(), (abs(xd) - meta["vmax"]).max())


def curved_mid_segment_error(meta, s, sd):
    m = meta["m"]
    xd = 3 * m[0] * s ** 2 * sd + 2 * m[1] * s * sd + m[2] * sd
    n = len(xd)
    seg = xd[int(0.3 * n):int(0.3 * n) + int(0.3 * n)]
    return (seg - meta["vmax"]).max()
