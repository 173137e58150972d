"""GPU parity tests: the HIP engine, called through its C-ABI, against the CPU
oracle on the same inputs. fp64 everywhere; the north-star tolerance is 1e-6
relative on t, s, sd, q. The kernels keep the reference's operation order and
are built with -ffp-contract=off, so most comparisons are additionally held to
bit-equality (flagged where they are)."""
import importlib
import json
import os

import numpy as np
import pytest

import scenarios
from conftest import PKG_NAME

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6   # BASELINE.json north_star


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X")
    eng = importlib.import_module(PKG_NAME + ".engine")
    syn = importlib.import_module(PKG_NAME + ".synthetic")
    from oracle import tpo
    return dict(torch=torch, eng=eng, syn=syn, tpo=tpo, E=eng.Engine(0), dev="cuda:0")


def assert_close(got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, what
    scale = np.maximum(np.abs(ref), 1e-9)
    err = np.max(np.abs(got - ref) / scale) if got.size else 0.0
    assert err <= REL_TOL, "%s: max rel err %.3e" % (what, err)


def solve_joint(env, batch, N, D, **kw):
    torch, eng, E = env["torch"], env["eng"], env["E"]
    B = batch["control_points"].shape[0]
    inp = eng.upload_joint_batch(batch, env["dev"])
    out = eng.alloc_joint_outputs(B, N, D, env["dev"])
    E.time_joint_paths(inp, out, N, **kw)
    torch.cuda.synchronize()
    return inp, out


def oracle_joint(env, batch, N, **kw):
    return env["tpo"].time_joint_batch(batch["knots"], batch["control_points"], batch["vmax"],
                                       batch["amax"], batch["path_start"], batch["delta"], N,
                                       sd_start=batch["sd_start"], time_start=batch["time_start"],
                                       nthreads=8, **kw)


# ---------------------------------------------------------------------- LP
def test_lp_literal_and_random(env, golden_dir):
    E, tpo = env["E"], env["tpo"]
    lp = json.load(open(os.path.join(golden_dir, "lp_regression.json")))
    for c in lp["cases"]:
        g = E.find_max_sd2(np.array([c["a"]]), np.array([c["b"]]), np.array([c["lower"]]),
                           np.array([c["upper"]]))
        o = tpo.find_max_sd2_simplex(c["a"], c["b"], c["lower"], c["upper"])
        r = tpo.find_max_sd2_bruteforce(c["a"], c["b"], c["lower"], c["upper"])
        assert (g[0][0], g[1][0], g[2][0]) == o            # bit-exact vs oracle
        assert abs(g[0][0] - r[0]) <= 1e-8 and abs(g[2][0] - r[2]) <= 1e-8   # reference's check
    rng = np.random.default_rng(0)
    for Cn in (2, 3, 7, 14, 30, 33, 50, 64):
        n = 3000
        A = rng.uniform(-100, 100, (n, Cn)); Bm = rng.uniform(-100, 100, (n, Cn))
        lo = rng.uniform(-10, 0, (n, Cn)); hi = rng.uniform(0, 10, (n, Cn))
        g = E.find_max_sd2(A, Bm, lo, hi)
        o = np.array([tpo.find_max_sd2_simplex(A[i], Bm[i], lo[i], hi[i]) for i in range(n)])
        for k in range(3):
            np.testing.assert_array_equal(g[k], o[:, k])   # bit-exact
        r = np.array([tpo.find_max_sd2_bruteforce(A[i], Bm[i], lo[i], hi[i]) for i in range(300)])
        for k in range(3):
            assert np.max(np.abs(g[k][:300] - r[:, k])) <= 1e-8
    # degenerate sets: all-zero rows and unbounded problems saturate at kMaxSd2
    z = np.zeros((2, 14))
    g = E.find_max_sd2(z, z, -np.ones((2, 14)), np.ones((2, 14)))
    assert g[0][0] == 1e6 and g[1][0] == 0.0 and g[2][0] == 1e6


# ------------------------------------------------ rows mode: reference scenarios
def _rows_solve(env, rows_list, s0, s1, sd0, max_loops=0, sdd0=None):
    E = env["E"]
    A = np.stack([r[0] for r in rows_list]); Bm = np.stack([r[1] for r in rows_list])
    lo = np.stack([r[2] for r in rows_list]); hi = np.stack([r[3] for r in rows_list])
    Bn, n, _ = A.shape
    inp = dict(a=np.ascontiguousarray(A), b=np.ascontiguousarray(Bm),
               lower=np.ascontiguousarray(lo), upper=np.ascontiguousarray(hi),
               s_start=np.full(Bn, s0, float), s_end=np.full(Bn, s1, float),
               sd_start=np.full(Bn, sd0, float),
               sdd_start=np.zeros(Bn) if sdd0 is None else np.full(Bn, sdd0, float),
               time_start=np.zeros(Bn))
    out = dict(time=np.zeros((Bn, n)), s=np.zeros((Bn, n)), sd=np.zeros((Bn, n)),
               sdd=np.zeros((Bn, n)), last_extremal_index=np.zeros(Bn, np.int32),
               max_time_increment=np.zeros(Bn), status=np.full(Bn, -1, np.int32))
    E.optimize_rows(inp, out, max_solver_loops=max_loops, host=True)
    return out


CASES = scenarios.all_cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_rows_mode_scenarios(env, case):
    tpo, E = env["tpo"], env["E"]
    name, rows, s0, s1, sd0, meta = case
    n, c = rows[0].shape
    p = tpo.Profile(n, c)
    assert p.setup(*rows, s0, s1, sd0, 0.0, 0.0) == 0 and p.optimize() == 0
    out = _rows_solve(env, [rows], s0, s1, sd0)
    assert out["status"][0] == 0
    bd = E.debug_boundary(1, n)
    # stage-wise, bit-exact: boundary curve, classification, solution
    np.testing.assert_array_equal(bd["sd2_max"][0], p.sd2_max)
    np.testing.assert_array_equal(bd["sdd_max"][0], p.sdd_max_for_sd2_max)
    np.testing.assert_array_equal(bd["sdd_min"][0], p.sdd_min_for_sd2_max)
    np.testing.assert_array_equal(bd["type"][0], p.boundary_type)
    for k, ref in (("time", p.time), ("s", p.s), ("sd", p.sd), ("sdd", p.sdd)):
        np.testing.assert_array_equal(out[k][0], ref)
        assert_close(out[k][0], ref, k)
    assert out["last_extremal_index"][0] == p.last_extremal_index
    assert out["max_time_increment"][0] == p.max_time_increment
    # and the reference's own property checks hold for the GPU result
    assert scenarios.max_violation(meta, out["s"][0], out["sd"][0], out["sdd"][0]) < tpo.KTINY


def test_rows_mode_setup_failures_and_mixed_batch(env):
    rows = scenarios.scalar_straight(30, 0.5, 1.0)
    A, Bm, lo, hi = rows
    bad_all = (A, Bm, lo, np.where(np.arange(30)[:, None] == 3, lo - 1.0, hi))
    bad_one = (A, Bm, lo, hi.copy())
    bad_one[3][3, 0] = lo[3, 0]
    out = _rows_solve(env, [rows, bad_all, bad_one, rows], 0.0, 1.0, 0.0)
    assert list(out["status"]) == [0, 2, 5, 0]          # a failed path does not abort the batch
    np.testing.assert_array_equal(out["time"][0], out["time"][3])
    assert _rows_solve(env, [rows], 1.0, 1.0, 0.0)["status"][0] == 3
    assert _rows_solve(env, [rows], 0.0, 1.0, -0.5)["status"][0] == 4
    assert _rows_solve(env, [bad_all], 1.0, 1.0, 0.0)["status"][0] == 2   # bounds checked first


def test_rows_mode_loop_limit_matches_oracle(env):
    # with too few solver loops the reference ends with NaNs in sd2 ("No solution found")
    tpo = env["tpo"]
    b = env["syn"].make_joint_batch(2, 7, 500)
    rows_list, refs = [], []
    for i in range(2):
        q, q1, q2 = tpo.joint_sample_path(b["knots"][i], b["control_points"][i], 0.0,
                                          b["delta"][i], 500)
        rows = tpo.joint_constraint_setup(q1, q2, b["vmax"][i], b["amax"][i])
        p = tpo.Profile(500, 14)
        p.set_max_loops(3)
        p.setup(*rows, 0.0, b["delta"][i] * 499)
        refs.append(p.optimize())
        rows_list.append(rows)
    # paths differ in s_end: solve one at a time
    for i in range(2):
        out = _rows_solve(env, [rows_list[i]], 0.0, b["delta"][i] * 499, 0.0, max_loops=3)
        assert out["status"][0] == refs[i] and refs[i] != 0


# -------------------------------------------------------- joint mode vs oracle
def test_generic_kernels_match_oracle_when_forced(env):
    """TPAMD_FORCE_GENERIC=1 (read at engine creation) routes every joint count through the
    generic kernels; they must give the same bits as the specialised ones."""
    eng, syn, torch = env["eng"], env["syn"], env["torch"]
    os.environ["TPAMD_FORCE_GENERIC"] = "1"
    try:
        E2 = eng.Engine(0)
    finally:
        del os.environ["TPAMD_FORCE_GENERIC"]
    for D, N, B in ((7, 700, 12), (5, 400, 8), (6, 300, 6)):
        b = syn.make_joint_batch(B, D, N)
        ref = oracle_joint(env, b, N)
        inp = eng.upload_joint_batch(b, env["dev"])
        out = eng.alloc_joint_outputs(B, N, D, env["dev"])
        E2.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out["status"].cpu().numpy(), ref["status"])
        ok = ref["status"] == 0
        for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
            np.testing.assert_array_equal(out[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok])
    cb = syn.make_cartesian_batch(6, 6, 300)
    cref = env["tpo"].time_cartesian_batch(cb["ik_positions"], cb["jacobians"], cb["vmax"], cb["amax"],
                                           cb["vtrans"], cb["vrot"], cb["path_start"], cb["delta"], nthreads=4)
    cout = eng.alloc_joint_outputs(6, 300, 6, env["dev"])
    E2.time_cartesian_paths(syn.upload_cartesian_batch(cb, env["dev"]), cout)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(cout["status"].cpu().numpy(), cref["status"])
    okc = cref["status"] == 0
    for k in ("time", "sd", "qdd"):
        np.testing.assert_array_equal(cout[k].cpu().numpy()[okc], cref["t" if k == "time" else k][okc])


@pytest.mark.parametrize("D,N,B", [(7, 500, 48), (7, 2000, 32), (6, 2000, 16), (14, 1000, 16),
                                   (3, 1000, 8), (4, 900, 8), (5, 1200, 8), (8, 700, 8),
                                   (1, 64, 4), (16, 300, 4), (7, 3, 2), (7, 4096, 2)])
def test_joint_mode_matches_oracle(env, D, N, B):
    b = env["syn"].make_joint_batch(B, D, N)
    ref = oracle_joint(env, b, N)
    _, out = solve_joint(env, b, N, D)
    st = out["status"].cpu().numpy()
    np.testing.assert_array_equal(st, ref["status"])
    ok = st == 0
    assert ok.sum() >= B // 2
    np.testing.assert_array_equal(out["last_extremal_index"].cpu().numpy()[ok],
                                  ref["last_extremal_index"][ok])
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
        g = out[k].cpu().numpy()[ok]
        r = ref["t" if k == "time" else k][ok]
        assert_close(g, r, k)
        np.testing.assert_array_equal(g, r)      # bit-exact


def test_joint_mode_start_velocity_time_offset_and_padding(env):
    # sd_start > 0, time_start != 0, path_start > 0 and a horizon that runs past the spline end
    syn, tpo = env["syn"], env["tpo"]
    D, N, B = 7, 800, 12
    b = syn.make_joint_batch(B, D, N)
    b["time_start"] = np.linspace(0.0, 50.0, B)
    b["sd_start"] = np.where(np.arange(B) % 3 == 0, 0.05, 0.0)
    b["path_start"] = np.where(np.arange(B) % 2 == 0, 0.0, 0.37)
    b["delta"] = b["delta"] * np.where(np.arange(B) % 4 == 1, 1.2, 1.0)   # overshoots the end
    ref = oracle_joint(env, b, N)
    _, out = solve_joint(env, b, N, D)
    st = out["status"].cpu().numpy()
    np.testing.assert_array_equal(st, ref["status"])
    ok = st == 0
    assert ok.any()
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
        np.testing.assert_array_equal(out[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok])


def _ragged_case(env, D, stride, counts, max_loops=0):
    """Per-path sample counts (BASELINE.json configs[4]); arrays keep the batch stride."""
    syn, tpo, torch, eng, E = env["syn"], env["tpo"], env["torch"], env["eng"], env["E"]
    B = len(counts)
    b = syn.make_joint_batch(B, D, stride)
    ns = np.asarray(counts, dtype=np.int32)
    b["delta"] = b["knots"][:, -1] / np.maximum(ns - 1, 1)
    inp = eng.upload_joint_batch(b, env["dev"])
    inp["num_samples_per_path"] = torch.from_numpy(ns).to(env["dev"])
    out = eng.alloc_joint_outputs(B, stride, D, env["dev"])
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
        out[k].fill_(-7.0)
    E.time_joint_paths(inp, out, stride, max_solver_loops=max_loops)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy()
    got = {k: out[k].cpu().numpy() for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd")}
    nok = 0
    for i, n in enumerate(counts):
        one = {k: b[k][i:i + 1] for k in ("knots", "control_points", "vmax", "amax", "path_start",
                                          "delta", "sd_start", "time_start")}
        if n < 2 or n > stride:          # n = 1 also has s_start == s_end, which is checked first
            assert st[i] == (6 if n > stride else 3)
            continue
        ref = oracle_joint(env, one, int(n))
        assert st[i] == ref["status"][0], (i, n)
        if st[i] != 0:
            continue
        nok += 1
        assert out["last_extremal_index"][i].item() == ref["last_extremal_index"][0]
        for k in got:
            np.testing.assert_array_equal(got[k][i, :n], ref["t" if k == "time" else k][0],
                                          err_msg="%s path %d n %d" % (k, i, n))
            assert (got[k][i, n:] == -7.0).all(), "wrote past n[b]"     # padding untouched
    return nok


@pytest.mark.parametrize("D", [7, 5])
def test_ragged_sample_counts_match_oracle_per_path(env, D):
    rng = np.random.default_rng(5)
    counts = [600, 3, 64, 65, 33, 1, 601] + list(rng.integers(50, 600, size=25))
    assert _ragged_case(env, D, 600, counts) >= 20


# ------------------------------------------ Cartesian paths (BASELINE.json configs[3])
@pytest.mark.parametrize("D,N,B", [(6, 800, 24), (7, 500, 8), (6, 2000, 8), (3, 64, 3)])
def test_cartesian_paths_match_oracle(env, D, N, B):
    syn, tpo, torch, eng, E = env["syn"], env["tpo"], env["torch"], env["eng"], env["E"]
    b = syn.make_cartesian_batch(B, D, N)
    b["time_start"] = np.linspace(0.0, 3.0, B)
    b["vtrans"][0] = 0.0          # degenerate Cartesian row (lower == upper): fails that path only
    ref = tpo.time_cartesian_batch(b["ik_positions"], b["jacobians"], b["vmax"], b["amax"],
                                   b["vtrans"], b["vrot"], b["path_start"], b["delta"],
                                   time_start=b["time_start"], nthreads=8)
    inp = syn.upload_cartesian_batch(b, env["dev"])
    out = eng.alloc_joint_outputs(B, N, D, env["dev"])
    E.time_cartesian_paths(inp, out)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy()
    np.testing.assert_array_equal(st, ref["status"])
    assert st[0] == 5
    ok = st == 0
    assert ok.sum() >= B - 2
    np.testing.assert_array_equal(out["last_extremal_index"].cpu().numpy()[ok],
                                  ref["last_extremal_index"][ok])
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
        np.testing.assert_array_equal(out[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok],
                                      err_msg=k)
    # Cartesian rows are respected: |J q'|^2 sd^2 <= v^2 (+ solver tolerance)
    q = b["ik_positions"]
    q1 = np.zeros_like(q)
    q1[:, :-1] = (q[:, 1:] - q[:, :-1]) / b["delta"][:, None, None]
    v6 = np.einsum("bnrd,bnd->bnr", b["jacobians"], q1)
    sd2 = out["sd"].cpu().numpy() ** 2
    assert (((v6[..., :3] ** 2).sum(-1) * sd2)[ok] <= (b["vtrans"][ok, None] ** 2) * (1 + 1e-6) + 1e-9).all()
    assert (((v6[..., 3:] ** 2).sum(-1) * sd2)[ok] <= (b["vrot"][ok, None] ** 2) * (1 + 1e-6) + 1e-9).all()
    # host-buffer entry point returns the same bits
    hin = dict(ik_positions=b["ik_positions"], jacobians=b["jacobians"], max_velocity=b["vmax"],
               max_acceleration=b["amax"], max_translational_velocity=b["vtrans"],
               max_rotational_velocity=b["vrot"], path_start=b["path_start"], delta=b["delta"],
               sd_start=b["sd_start"], time_start=b["time_start"])
    hout = dict(time=np.zeros((B, N)), s=np.zeros((B, N)), sd=np.zeros((B, N)),
                sdd=np.zeros((B, N)), qd=np.zeros((B, N, D)), qdd=np.zeros((B, N, D)),
                status=np.full(B, -1, np.int32))
    E.time_cartesian_paths(hin, hout, host=True)
    np.testing.assert_array_equal(hout["status"], st)
    for k in ("time", "sd", "qdd"):
        np.testing.assert_array_equal(hout[k][ok], out[k].cpu().numpy()[ok])


def test_engine_reproduces_committed_regression_vectors(env, golden_dir):
    """The oracle-generated vectors of tests/golden/solver_oracle_derived.npz, bit for bit."""
    stored = dict(np.load(os.path.join(golden_dir, "solver_oracle_derived.npz")))
    for name, rows, s0, s1, sd0, _meta in scenarios.all_cases():
        out = _rows_solve(env, [rows], s0, s1, sd0)
        assert out["status"][0] == stored["scn/%s/status" % name][0], name
        assert out["last_extremal_index"][0] == stored["scn/%s/lei" % name][0], name
        for k in ("time", "sd", "sdd"):
            np.testing.assert_array_equal(out[k][0], stored["scn/%s/%s" % (name, k)], err_msg=name + k)
    for D, N in ((7, 500), (7, 2000), (6, 2000), (14, 1000)):
        key = "joint/D%d_N%d" % (D, N)
        b = env["syn"].make_joint_batch(8, D, N)
        _, out = solve_joint(env, b, N, D)
        np.testing.assert_array_equal(out["status"].cpu().numpy(), stored[key + "/status"])
        np.testing.assert_array_equal(out["last_extremal_index"].cpu().numpy(), stored[key + "/lei"])
        for k in ("time", "sd", "sdd"):
            np.testing.assert_array_equal(out[k].cpu().numpy(), stored[key + "/" + k], err_msg=key + k)
        np.testing.assert_array_equal(out["qdd"].cpu().numpy()[:, :, -1], stored[key + "/qdd_last_joint"])


def test_api_limits_empty_batch_and_maximum_sizes(env):
    syn, tpo, torch, eng, E = env["syn"], env["tpo"], env["torch"], env["eng"], env["E"]
    # empty batch: a no-op that succeeds (reference: nothing to plan)
    b = syn.make_joint_batch(1, 7, 100)
    inp = eng.upload_joint_batch(b, env["dev"])
    empty_in = {k: v[:0].contiguous() for k, v in inp.items()}
    empty_out = eng.alloc_joint_outputs(0, 100, 7, env["dev"])
    E.time_joint_paths(empty_in, empty_out, 100)
    # sizes outside the supported range are refused with an error code, not clipped
    out = eng.alloc_joint_outputs(1, 100, 7, env["dev"])
    for bad_n in (2, 8193):
        with pytest.raises(eng.TpamdError, match="unsupported"):
            E.time_joint_paths(inp, out, bad_n)
    b17 = syn.make_joint_batch(1, 17, 100)
    with pytest.raises(eng.TpamdError, match="unsupported"):
        E.time_joint_paths(eng.upload_joint_batch(b17, env["dev"]),
                           eng.alloc_joint_outputs(1, 100, 17, env["dev"]), 100)
    # the largest supported problem: D = 16, N = 8192 (generic kernels), and N = 8192 at D = 7
    for D, N in ((16, 8192), (7, 8192)):
        bb = syn.make_joint_batch(2, D, N)
        ref = oracle_joint(env, bb, N)
        _, o = solve_joint(env, bb, N, D)
        np.testing.assert_array_equal(o["status"].cpu().numpy(), ref["status"])
        ok = ref["status"] == 0
        for k in ("time", "sd", "sdd", "qdd"):
            np.testing.assert_array_equal(o[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok])


def test_rows_mode_wide_constraint_sets(env):
    """C = 33 and C = 64 rows per sample (two-word LP bit sets, candidate loop > 64 lanes)."""
    tpo = env["tpo"]
    rng = np.random.default_rng(3)
    n = 120
    for C in (33, 64):
        s = np.linspace(0.0, 2.0, n)
        a = rng.uniform(0.2, 1.5, (n, C)) * np.sign(rng.uniform(-1, 1, (1, C)))
        a = a * (1.0 + 0.3 * np.sin(3.0 * s)[:, None])
        bcoef = rng.uniform(-0.5, 0.5, (n, C)) * np.cos(2.0 * s)[:, None]
        hi = rng.uniform(1.0, 3.0, (1, C)).repeat(n, 0)
        lo = -rng.uniform(1.0, 3.0, (1, C)).repeat(n, 0)
        a[:, C // 2:] = 0.0                               # velocity-like rows
        bcoef[:, C // 2:] = np.abs(bcoef[:, C // 2:]) + 0.05
        lo[:, C // 2:] = 0.0
        p = tpo.Profile(n, C)
        assert p.setup(a, bcoef, lo, hi, 0.0, 2.0, 0.0, 0.0, 0.0) == 0
        rc = p.optimize()
        out = _rows_solve(env, [(a, bcoef, lo, hi)], 0.0, 2.0, 0.0)
        assert out["status"][0] == rc
        if rc == 0:
            for k in ("time", "sd", "sdd"):
                np.testing.assert_array_equal(out[k][0], getattr(p, k), err_msg="C=%d %s" % (C, k))


def test_device_entry_point_captures_into_a_hip_graph(env):
    """the _device entries neither allocate (after reserve) nor synchronise,
    so a whole batch solve can be captured once and replayed."""
    syn, torch, eng, E = env["syn"], env["torch"], env["eng"], env["E"]
    D, N, B = 7, 600, 16
    b = syn.make_joint_batch(B, D, N)
    ref = oracle_joint(env, b, N)
    inp = eng.upload_joint_batch(b, env["dev"])
    out = eng.alloc_joint_outputs(B, N, D, env["dev"])
    E.reserve(B, N, 2 * D)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        E.time_joint_paths(inp, out, N, stream=side)          # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        E.time_joint_paths(inp, out, N, stream=side)
    for k in ("time", "sd", "qdd"):
        out[k].fill_(-1.0)
    out["status"].fill_(-5)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out["status"].cpu().numpy(), ref["status"])
    ok = ref["status"] == 0
    for k in ("time", "sd", "qdd"):
        np.testing.assert_array_equal(out[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok])


def test_diagnostic_build_confirms_the_critical_point_shortcut(env):
    """The sweep kernel replaces the sd2 scan of NextCriticalPoint by a loop invariant
    (tpamd_sweep_joint.h); the diagnostic build also runs the literal walk and counts
    disagreements."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = env["eng"].DIAG_SO
    if not os.path.exists(diag):
        env["eng"].build_diagnostic_library()
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_crosscheck.py")],
                         env=dict(os.environ, TPAMD_LIBRARY=diag), capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    report = json.loads(res.stdout.strip().splitlines()[-1])
    assert report["searches"] > 1000 and report["mismatches"] == 0, report


def test_joint_mode_bad_limits_fail_per_path(env):
    syn = env["syn"]
    b = syn.make_joint_batch(6, 7, 300)
    b["amax"][1, :] = 0.0          # every acceleration row lower == upper, but velocity rows fine
    b["vmax"][2, :] = 0.0
    b["amax"][2, :] = 0.0          # every row degenerate -> infeasible bounds
    b["sd_start"][3] = -1.0
    b["delta"][4] = 0.0            # s_start == s_end
    ref = oracle_joint(env, b, 300)
    _, out = solve_joint(env, b, 300, 7)
    st = out["status"].cpu().numpy()
    np.testing.assert_array_equal(st, ref["status"])
    assert list(st[1:5]) == [5, 2, 4, 3] and st[0] == 0 and st[5] == 0


def test_host_buffer_entry_point_equals_device_entry_point(env):
    syn, E = env["syn"], env["E"]
    D, N, B = 7, 500, 8
    b = syn.make_joint_batch(B, D, N)
    _, out = solve_joint(env, b, N, D)
    inp = dict(knots=b["knots"], control_points=b["control_points"], max_velocity=b["vmax"],
               max_acceleration=b["amax"], path_start=b["path_start"], delta=b["delta"],
               sd_start=b["sd_start"], time_start=b["time_start"])
    hout = dict(time=np.zeros((B, N)), s=np.zeros((B, N)), sd=np.zeros((B, N)),
                sdd=np.zeros((B, N)), q=np.zeros((B, N, D)), qd=np.zeros((B, N, D)),
                qdd=np.zeros((B, N, D)), last_extremal_index=np.zeros(B, np.int32),
                max_time_increment=np.zeros(B), status=np.full(B, -1, np.int32))
    E.time_joint_paths(inp, hout, N, host=True)
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd", "status", "last_extremal_index"):
        np.testing.assert_array_equal(hout[k], out[k].cpu().numpy())


# ------------------------------------------------------- query and resample
def test_query_and_resample_match_oracle(env):
    torch, syn, tpo, E = env["torch"], env["syn"], env["tpo"], env["E"]
    D, N, B, K = 7, 600, 6, 257
    b = syn.make_joint_batch(B, D, N)
    inp, out = solve_joint(env, b, N, D)
    t = out["time"].cpu().numpy()
    rng = np.random.default_rng(5)
    tq = np.sort(rng.uniform(-0.5, t[:, -1:] + 0.5, (B, K)), axis=1)
    tq[:, 10] = t[:, 37]           # exactly on a sample
    tq_d = torch.from_numpy(tq).to(env["dev"])
    qs, qsd, qsdd = (torch.empty(B, K, dtype=torch.float64, device=env["dev"]) for _ in range(3))
    ok = torch.zeros(B, K, dtype=torch.int32, device=env["dev"])
    E.query(out["time"], out["s"], out["sd"], out["status"], tq_d, qs, qsd, qsdd, ok)
    torch.cuda.synchronize()
    for i in range(B):
        q_, q1, q2 = tpo.joint_sample_path(b["knots"][i], b["control_points"][i], 0.0,
                                           b["delta"][i], N)
        rows = tpo.joint_constraint_setup(q1, q2, b["vmax"][i], b["amax"][i])
        p = tpo.Profile(N, 2 * D)
        p.set_max_loops(10 * N)
        assert p.setup(*rows, 0.0, b["delta"][i] * (N - 1)) == 0 and p.optimize() == 0
        ref = np.array([p.query(x)[1:] for x in tq[i]])
        np.testing.assert_array_equal(qs[i].cpu().numpy(), ref[:, 0])
        np.testing.assert_array_equal(qsd[i].cpu().numpy(), ref[:, 1])
        np.testing.assert_array_equal(qsdd[i].cpu().numpy(), ref[:, 2])
    assert int(ok.min()) == 1

    # resample (path_timing_trajectory.cc:755-783); 4 ms step as the reference's planner tests
    dt = 0.004
    start = np.zeros(B)
    counts_ref = [tpo.resample_uniform(t[i], *[out[k][i].cpu().numpy() for k in
                                               ("s", "sd", "sdd", "q", "qd", "qdd")],
                                       0.0, dt, b["amax"][i]) for i in range(B)]
    max_out = max(len(r[0]) for r in counts_ref) + 3
    f = dict(dtype=torch.float64, device=env["dev"])
    ro = dict(out_time=torch.zeros(B, max_out, **f), out_s=torch.zeros(B, max_out, **f),
              out_sd=torch.zeros(B, max_out, **f), out_sdd=torch.zeros(B, max_out, **f),
              out_q=torch.zeros(B, max_out, D, **f), out_qd=torch.zeros(B, max_out, D, **f),
              out_qdd=torch.zeros(B, max_out, D, **f),
              count=torch.zeros(B, dtype=torch.int32, device=env["dev"]))
    E.resample_uniform(out, inp["max_acceleration"], torch.from_numpy(start).to(env["dev"]), dt, ro)
    torch.cuda.synchronize()
    cnt = ro["count"].cpu().numpy()
    for i in range(B):
        r = counts_ref[i]
        M = len(r[0])
        assert cnt[i] == M
        for k, ref in zip(("out_time", "out_s", "out_sd", "out_sdd", "out_q", "out_qd", "out_qdd"), r):
            np.testing.assert_array_equal(ro[k][i, :M].cpu().numpy(), ref)
        np.testing.assert_array_equal(ro["out_q"][i, M - 1].cpu().numpy(), out["q"][i, -1].cpu().numpy())
        assert float(ro["out_qd"][i, M - 1].abs().max()) == 0.0

    # kSkipSamplesCloserThanTimeStep (path_timing_trajectory.cc:785-836), two time steps: one
    # denser than the path samples (nothing skipped) and one much coarser, from a start time
    # inside the trajectory
    for dt_skip, start_at in ((1e-6, 0.0), (0.05, 0.37), (0.004, 0.0)):
        start = np.full(B, start_at)
        refs = [tpo.resample_skip(t[i], *[out[k][i].cpu().numpy() for k in
                                          ("s", "sd", "sdd", "q", "qd", "qdd")],
                                  start_at, dt_skip, b["amax"][i]) for i in range(B)]
        cap = N + 1
        ro = dict(out_time=torch.zeros(B, cap, **f), out_s=torch.zeros(B, cap, **f),
                  out_sd=torch.zeros(B, cap, **f), out_sdd=torch.zeros(B, cap, **f),
                  out_q=torch.zeros(B, cap, D, **f), out_qd=torch.zeros(B, cap, D, **f),
                  out_qdd=torch.zeros(B, cap, D, **f),
                  count=torch.zeros(B, dtype=torch.int32, device=env["dev"]))
        E.resample_uniform(out, inp["max_acceleration"], torch.from_numpy(start).to(env["dev"]), dt_skip,
                           ro, skip=True)
        torch.cuda.synchronize()
        cnt = ro["count"].cpu().numpy()
        for i in range(B):
            M = len(refs[i][0])
            assert cnt[i] == M and M >= 2
            for k, ref in zip(("out_time", "out_s", "out_sd", "out_sdd", "out_q", "out_qd", "out_qdd"), refs[i]):
                np.testing.assert_array_equal(ro[k][i, :M].cpu().numpy(), ref, err_msg="%s dt=%g" % (k, dt_skip))
            gaps = np.diff(ro["out_time"][i, :M].cpu().numpy())
            assert (gaps >= 0.95 * dt_skip - 1e-15).all()


@pytest.mark.parametrize("D,N,B,first", [(7, 2000, 3072, 5000), (6, 1500, 1024, 9000),
                                         (14, 800, 512, 12000), (7, 333, 2048, 20000)])
def test_wide_parity_sweep_against_the_oracle(env, D, N, B, first):
    """Thousands of further paths (other seeds than the bench batch), with per-path limits,
    start velocities and time offsets: every output of every path bit for bit."""
    syn = env["syn"]
    b = syn.make_joint_batch(B, D, N, first_path_index=first)
    rng = np.random.default_rng(first)
    b["vmax"] = b["vmax"] * rng.uniform(0.3, 3.0, (B, 1))
    b["amax"] = b["amax"] * rng.uniform(0.2, 5.0, (B, 1))
    b["sd_start"] = np.where(rng.uniform(size=B) < 0.3, rng.uniform(0.0, 0.2, B), 0.0)
    b["time_start"] = rng.uniform(0.0, 100.0, B)
    ref = oracle_joint(env, b, N)
    _, out = solve_joint(env, b, N, D)
    st = out["status"].cpu().numpy()
    np.testing.assert_array_equal(st, ref["status"])
    ok = st == 0
    assert ok.mean() > 0.5
    np.testing.assert_array_equal(out["last_extremal_index"].cpu().numpy()[ok],
                                  ref["last_extremal_index"][ok])
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
        np.testing.assert_array_equal(out[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok],
                                      err_msg=k)


def test_pathological_inputs_terminate_and_match_the_oracle_status(env):
    """NaN / inf / huge / denormal inputs in some paths of a batch: the call returns, the
    other paths are unaffected, and every path reports the oracle's status."""
    syn = env["syn"]
    D, N, B = 7, 400, 16
    b = syn.make_joint_batch(B, D, N)
    b["control_points"][1, 5, 2] = np.nan
    b["control_points"][2, :, :] = 0.0               # a path of zero length
    b["vmax"][3, :] = np.inf
    b["amax"][4, 0] = np.nan
    b["control_points"][5] *= 1e150
    b["control_points"][6] *= 1e-300
    b["amax"][7, :] = 1e-310                         # denormal limits
    b["delta"][8] = np.nan
    b["vmax"][9, 3] = -1.0
    b["knots"][10, 10] = np.nan
    ref = oracle_joint(env, b, N)
    _, out = solve_joint(env, b, N, D)
    st = out["status"].cpu().numpy()
    np.testing.assert_array_equal(st, ref["status"])
    ok = st == 0
    assert ok[0] and ok[11:].all()
    for k in ("time", "sd", "qdd"):
        np.testing.assert_array_equal(out[k].cpu().numpy()[ok], ref["t" if k == "time" else k][ok])


def test_repeated_solves_are_bitwise_reproducible(env):
    """The two waves of a path share sd2 in LDS and exchange results through barriers: a
    race would show up as run-to-run differences. 300 solves of the bench batch and of a
    Cartesian batch must reproduce the first one bit for bit."""
    torch, syn, eng, E = env["torch"], env["syn"], env["eng"], env["E"]
    D, N, B = 7, 2000, 1024
    b = syn.make_joint_batch(B, D, N)
    inp = eng.upload_joint_batch(b, env["dev"])
    first = eng.alloc_joint_outputs(B, N, D, env["dev"])
    E.time_joint_paths(inp, first, N)
    out = eng.alloc_joint_outputs(B, N, D, env["dev"])
    keys = ("time", "s", "sd", "sdd", "qd", "qdd", "status", "last_extremal_index")
    for it in range(300):
        for k in ("time", "sd", "sdd", "qdd"):
            out[k].fill_(float(it))
        E.time_joint_paths(inp, out, N)
        if it % 10 == 9 or it < 3:
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(out[k], first[k]), (k, it)
    cb = syn.make_cartesian_batch(512, 6, 1500)
    cin = syn.upload_cartesian_batch(cb, env["dev"])
    cfirst = eng.alloc_joint_outputs(512, 1500, 6, env["dev"])
    E.time_cartesian_paths(cin, cfirst)
    cout = eng.alloc_joint_outputs(512, 1500, 6, env["dev"])
    for it in range(100):
        E.time_cartesian_paths(cin, cout)
        if it % 10 == 9:
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(cout[k], cfirst[k]), (k, it)


# ---------------------------------------------- BASELINE-size batch: properties
def test_config2_full_size_properties(env):
    """Config 2 of BASELINE.json (1024 paths, 7 dof, 2000 samples): size-independent
    properties on every path, bit-parity with the oracle on a strided subset."""
    torch, syn, tpo = env["torch"], env["syn"], env["tpo"]
    D, N, B = 7, 2000, 1024
    b = syn.make_joint_batch(B, D, N)
    inp, out = solve_joint(env, b, N, D)
    st = out["status"].cpu().numpy()
    assert (st == 0).all()
    t = out["time"].cpu().numpy(); s = out["s"].cpu().numpy()
    sd = out["sd"].cpu().numpy(); sdd = out["sdd"].cpu().numpy()
    qd = out["qd"].cpu().numpy(); qdd = out["qdd"].cpu().numpy(); q = out["q"].cpu().numpy()
    assert np.isfinite(t).all() and np.isfinite(sd).all() and np.isfinite(sdd).all()
    assert (np.diff(t, axis=1) >= 0).all() and (t[:, 0] == 0).all()
    assert (sd[:, -1] == 0).all() and (sd[:, 0] == 0).all() and (sd >= 0).all()
    assert (s[:, 0] == 0).all()
    np.testing.assert_array_equal(s[:, -1], b["delta"] * (N - 1))
    # joint limits with the 0.8 safety factor (timeable_path.h:80): |qd| <= 0.8 vmax, |qdd| <= amax
    assert (np.abs(qd) <= 0.8 * b["vmax"][:, None, :] * (1 + 1e-9) + 1e-12).all()
    assert (np.abs(qdd) <= b["amax"][:, None, :]).all()
    np.testing.assert_allclose(q[:, 0], b["control_points"][:, 0], atol=1e-12)
    np.testing.assert_allclose(q[:, -1], b["control_points"][:, -1], atol=1e-9)
    # a time-optimal profile rides a velocity limit on a sizeable share of the samples
    vact = np.abs(qd) >= 0.8 * b["vmax"][:, None, :] * (1 - 1e-6)
    sel = slice(0, B, 37)
    ref = tpo.time_joint_batch(b["knots"][sel], b["control_points"][sel], b["vmax"][sel],
                               b["amax"][sel], b["path_start"][sel], b["delta"][sel], N, nthreads=8)
    for k, g in (("t", t), ("s", s), ("sd", sd), ("sdd", sdd), ("q", q), ("qd", qd), ("qdd", qdd)):
        np.testing.assert_array_equal(g[sel], ref[k])
    assert vact.any(axis=2).mean() > 0.2
    # a checksum of checksums so that a silent change of any output shows up
    chk = float(np.sum(t[:, -1]) + np.sum(sd * 1e-3) + np.sum(q * 1e-6))
    chk_ref_subset = float(np.sum(ref["t"][:, -1]))
    assert abs(np.sum(t[sel, -1]) - chk_ref_subset) == 0.0 and np.isfinite(chk)
