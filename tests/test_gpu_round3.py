"""GPU tests added in round 3: concurrent joint groups (tpamd_time_joint_groups_*), the
longest-first order of ragged batches, and the ordering of unpipelined entry points against a
pipelined engine (the _host entry points, rows / Cartesian solves between pipelined ones)."""
import importlib

import numpy as np
import pytest

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu

KEYS = ("time", "s", "sd", "sdd", "q", "qd", "qdd")


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X")
    eng = importlib.import_module(PKG_NAME + ".engine")
    syn = importlib.import_module(PKG_NAME + ".synthetic")
    shd = importlib.import_module(PKG_NAME + ".sharding")
    from oracle import tpo
    return dict(torch=torch, eng=eng, syn=syn, shd=shd, tpo=tpo, E=eng.Engine(0), dev="cuda:0")


def _mixed_share(env, total=1536, world=8, rank=5):
    """Rank `rank`'s share of configs[4], bucketed by (D, ceil(N / 512))."""
    torch, eng, syn, shd = (env[k] for k in ("torch", "eng", "syn", "shd"))
    dofs, samples = syn.mixed_batch_shape(total)
    costs = samples.astype(np.float64) * (2.0 * dofs) ** 2
    lo, hi = shd.balanced_bounds(costs, world)[rank]
    groups = []
    for (D, stride), pos in syn.mixed_batch_groups(dofs[lo:hi], samples[lo:hi]).items():
        gidx = lo + pos
        ns = samples[gidx]
        b = syn.make_mixed_group(gidx, D, ns, stride)
        inp = eng.upload_joint_batch(b, env["dev"])
        inp["num_samples_per_path"] = torch.from_numpy(ns).to(env["dev"])
        out = eng.alloc_joint_outputs(len(pos), stride, D, env["dev"])
        groups.append(dict(D=D, stride=stride, ns=ns, host=b, inputs=inp, outputs=out, num_samples=stride))
    return groups


def test_joint_groups_run_side_by_side_and_every_path_matches_its_oracle_run(env):
    """One GPU's share of configs[4] as ONE call: all (D, ceil(N/512)) buckets on the engine's
    lanes at once, each bucket's sweep longest path first. Every path must equal the oracle run
    on that path alone, nothing may be written behind a path's own sample count, and the call must
    behave like one solve on the caller's stream (results complete when the stream has passed it,
    repeated calls reuse the lanes)."""
    torch, eng, tpo, E = (env[k] for k in ("torch", "eng", "tpo", "E"))
    groups = _mixed_share(env)
    assert {g["D"] for g in groups} == {6, 7, 14} and len(groups) >= 6
    for rep in range(3):                         # lanes and their workspaces are reused
        for g in groups:
            for k in KEYS:
                g["outputs"][k].fill_(-7.0)
            g["outputs"]["status"].fill_(-1)
        E.time_joint_groups(groups)
    torch.cuda.synchronize()
    checked = 0
    for g in groups:
        out, ns, b = g["outputs"], g["ns"], g["host"]
        st = out["status"].cpu().numpy()
        got = {k: out[k].cpu().numpy() for k in KEYS}
        lei = out["last_extremal_index"].cpu().numpy()
        for i, n in enumerate(ns):
            one = [b[k][i:i + 1] for k in ("knots", "control_points", "vmax", "amax", "path_start", "delta")]
            ref = tpo.time_joint_batch(*one, int(n), nthreads=1)
            assert st[i] == ref["status"][0] == 0, (g["D"], g["stride"], i, st[i])
            assert lei[i] == ref["last_extremal_index"][0]
            for k in got:
                np.testing.assert_array_equal(got[k][i, :n], ref["t" if k == "time" else k][0],
                                              err_msg="%s D=%d n=%d" % (k, g["D"], n))
                assert (got[k][i, n:] == -7.0).all(), "wrote past the path's sample count"
            checked += 1
    assert checked >= 150
    # the host-buffer variant (what BatchPathTiming calls) gives the same bits
    hgroups = []
    for g in groups[:4]:
        b, n_g, stride, D = g["host"], len(g["ns"]), g["stride"], g["D"]
        hin = dict(knots=b["knots"], control_points=b["control_points"], max_velocity=b["vmax"],
                   max_acceleration=b["amax"], path_start=b["path_start"], delta=b["delta"],
                   sd_start=b["sd_start"], time_start=b["time_start"],
                   num_samples_per_path=np.ascontiguousarray(g["ns"], dtype=np.int32))
        hout = dict(time=np.full((n_g, stride), -7.0), s=np.full((n_g, stride), -7.0),
                    sd=np.full((n_g, stride), -7.0), sdd=np.full((n_g, stride), -7.0),
                    q=np.full((n_g, stride, D), -7.0), qd=np.full((n_g, stride, D), -7.0),
                    qdd=np.full((n_g, stride, D), -7.0), status=np.full(n_g, -1, dtype=np.int32),
                    last_extremal_index=np.zeros(n_g, dtype=np.int32))
        hgroups.append(dict(inputs=hin, outputs=hout, num_samples=stride))
    E.time_joint_groups(hgroups, host=True)
    for g, h in zip(groups[:4], hgroups):
        np.testing.assert_array_equal(h["outputs"]["status"], g["outputs"]["status"].cpu().numpy())
        for k in KEYS:
            # (the device staging buffer is not pre-filled: compare the valid samples)
            dev = g["outputs"][k].cpu().numpy()
            for i, n in enumerate(g["ns"]):
                np.testing.assert_array_equal(h["outputs"][k][i, :n], dev[i, :n], err_msg=k)


def test_joint_groups_accept_uniform_and_empty_groups(env):
    """Groups without per-path sample counts, a group of one path, an empty group and a single
    group (which runs on the caller's stream) all give the single-call results."""
    torch, eng, syn, E = (env[k] for k in ("torch", "eng", "syn", "E"))
    shapes = [(40, 7, 700), (1, 6, 333), (24, 14, 450), (0, 7, 100), (17, 5, 260)]
    groups, refs = [], []
    for B, D, N in shapes:
        b = syn.make_joint_batch(max(B, 1), D, N, first_path_index=31 * D)
        inp = eng.upload_joint_batch(b, env["dev"])
        if B == 0:
            inp = {k: v[:0].contiguous() for k, v in inp.items()}
        ref = eng.alloc_joint_outputs(B, N, D, env["dev"])
        if B:
            E.time_joint_paths(inp, ref, N)
        out = eng.alloc_joint_outputs(B, N, D, env["dev"])
        groups.append(dict(inputs=inp, outputs=out, num_samples=N))
        refs.append(ref)
    E.time_joint_groups(groups)
    torch.cuda.synchronize()
    for g, ref in zip(groups, refs):
        for k in KEYS + ("status", "last_extremal_index", "max_time_increment"):
            assert torch.equal(g["outputs"][k], ref[k]), k
    one = dict(inputs=groups[0]["inputs"], outputs=eng.alloc_joint_outputs(40, 700, 7, env["dev"]),
               num_samples=700)
    E.time_joint_groups([one])
    torch.cuda.synchronize()
    assert torch.equal(one["outputs"]["time"], refs[0]["time"])
    E.time_joint_groups([])


def test_ragged_batch_results_do_not_depend_on_the_processing_order(env, monkeypatch):
    """The longest-first order of a ragged batch (k_order_paths) is scheduling only: an engine
    with TPAMD_ORDER_RAGGED=0 (paths in index order) gives the same bits, and a batch whose counts
    exceed the stride or fall below 2 still fails those paths, not the batch."""
    torch, eng, syn = env["torch"], env["eng"], env["syn"]
    B, D, N = 300, 7, 1500
    rng = np.random.default_rng(5)
    ns = rng.integers(3, N + 1, size=B).astype(np.int32)
    ns[7] = 1          # one sample: s_start == s_end (status 3 comes before "too few")
    ns[11] = N + 40    # more than the stride
    b = syn.make_joint_batch(B, D, N)
    b["delta"] = b["knots"][:, -1] / np.maximum(ns - 1, 1)
    inp = eng.upload_joint_batch(b, env["dev"])
    inp["num_samples_per_path"] = torch.from_numpy(ns).to(env["dev"])
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("TPAMD_ORDER_RAGGED", flag)
        Ex = eng.Engine(0)
        out = eng.alloc_joint_outputs(B, N, D, env["dev"])
        for k in KEYS:
            out[k].fill_(-7.0)
        Ex.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        outs.append(out)
        Ex.close()
    st = outs[0]["status"].cpu().numpy()
    assert st[7] == 3 and st[11] == 6 and (np.delete(st, [7, 11]) == 0).all()
    for k in KEYS + ("status", "last_extremal_index"):
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_host_entry_point_is_never_pipelined(env):
    """tpamd_time_joint_paths_host on an engine in pipelined mode 1 or 2 must return
    complete results (the copies back used to be ordered behind the PREVIOUS solve only)."""
    torch, eng, syn = env["torch"], env["eng"], env["syn"]
    B, D, N = 64, 7, 1200
    E1, E2 = eng.Engine(0), eng.Engine(0)
    batches = [syn.make_joint_batch(B, D, N, first_path_index=f) for f in (0, 4000, 9000)]

    def host_solve(E, b):
        hin = dict(knots=b["knots"], control_points=b["control_points"], max_velocity=b["vmax"],
                   max_acceleration=b["amax"], path_start=b["path_start"], delta=b["delta"],
                   sd_start=b["sd_start"], time_start=b["time_start"])
        hout = dict(time=np.zeros((B, N)), s=np.zeros((B, N)), sd=np.zeros((B, N)), sdd=np.zeros((B, N)),
                    q=np.zeros((B, N, D)), qd=np.zeros((B, N, D)), qdd=np.zeros((B, N, D)),
                    status=np.full(B, -1, dtype=np.int32), last_extremal_index=np.zeros(B, dtype=np.int32))
        E.time_joint_paths(hin, hout, N, host=True)
        return hout

    refs = [host_solve(E1, b) for b in batches]
    for mode in (1, 2):
        E2.set_pipelining(mode)
        # a pipelined device solve in flight, then host solves of other batches right behind it
        inp = eng.upload_joint_batch(batches[0], env["dev"])
        out = eng.alloc_joint_outputs(B, N, D, env["dev"])
        torch.cuda.synchronize()
        E2.time_joint_paths(inp, out, N)
        E2.time_joint_paths(inp, out, N)
        for b, ref in zip(batches, refs):
            got = host_solve(E2, b)
            for k in got:
                np.testing.assert_array_equal(got[k], ref[k], err_msg="mode %d %s" % (mode, k))
        E2.fence()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out["time"].cpu().numpy(), refs[0]["time"])
    E1.close(); E2.close()


def test_unpipelined_solves_between_pipelined_ones_share_the_workspace_safely(env):
    """a mode-2 joint solve, then a rows solve and a Cartesian solve (which use the
    engine's current workspace from the caller's stream), then two more joint solves -- no host
    synchronisation anywhere -- must all give the results of an unpipelined engine."""
    torch, eng, syn = env["torch"], env["eng"], env["syn"]
    E1, E2 = eng.Engine(0), eng.Engine(0)
    B, D, N = 192, 7, 2000
    jb = [syn.make_joint_batch(B, D, N, first_path_index=f) for f in (0, 7000)]
    jin = [eng.upload_joint_batch(b, env["dev"]) for b in jb]
    cb = syn.make_cartesian_batch(96, 6, 1500)
    cin = syn.upload_cartesian_batch(cb, env["dev"])
    # explicit rows: a small joint-like problem (A = q', B = q'' rows with box limits)
    rb = syn.make_joint_batch(64, 3, 900, first_path_index=123)
    q, q1, q2 = E1.sample_joint_paths(rb["knots"], rb["control_points"], rb["path_start"], rb["delta"], 900)
    a = np.concatenate([q1, np.zeros_like(q1)], axis=2)
    bq = np.concatenate([q2, q1 * q1], axis=2)
    hi = np.broadcast_to(np.concatenate([rb["amax"], rb["vmax"] ** 2], axis=1)[:, None, :], a.shape).copy()
    lo = np.broadcast_to(np.concatenate([-rb["amax"], np.zeros_like(rb["vmax"])], axis=1)[:, None, :], a.shape).copy()
    rin = {k: torch.from_numpy(np.ascontiguousarray(v)).to(env["dev"]) for k, v in dict(
        a=a, b=bq, lower=lo, upper=hi, s_start=np.zeros(64), s_end=rb["knots"][:, -1].copy(),
        sd_start=np.zeros(64), time_start=np.zeros(64)).items()}

    def run(E, piped):
        outs = dict(j0=eng.alloc_joint_outputs(B, N, D, env["dev"]), j1=eng.alloc_joint_outputs(B, N, D, env["dev"]),
                    j2=eng.alloc_joint_outputs(B, N, D, env["dev"]), c=eng.alloc_joint_outputs(96, 1500, 6, env["dev"]),
                    r=eng.alloc_joint_outputs(64, 900, 6, env["dev"], with_q=False, with_derivs=False))
        torch.cuda.synchronize()
        for rep in range(3):
            E.time_joint_paths(jin[0], outs["j0"], N)
            E.optimize_rows(rin, outs["r"])
            E.time_cartesian_paths(cin, outs["c"])
            E.time_joint_paths(jin[1], outs["j1"], N)
            E.time_joint_paths(jin[0], outs["j2"], N)
        if piped:
            E.fence()
        torch.cuda.synchronize()
        return outs

    ref = run(E1, False)
    E2.set_pipelining(2)
    got = run(E2, True)
    E2.set_pipelining(1)
    got1 = run(E2, True)
    for name in ref:
        for k in ("time", "sd", "sdd", "status", "last_extremal_index"):
            assert torch.equal(got[name][k], ref[name][k]), (name, k)
            assert torch.equal(got1[name][k], ref[name][k]), (name, k)
    assert (ref["r"]["status"] == 0).all() and (ref["c"]["status"] == 0).sum() > 80
    E1.close(); E2.close()


def test_config3_full_size_properties(env):
    """BASELINE.json configs[3] at its stated size -- 4096 Cartesian 6-joint paths of 2000 samples,
    one engine call: size-independent properties on EVERY path (monotone time, rest to rest, joint
    and Cartesian speed limits, the IK positions passed through), bit parity with the oracle on a
    strided subset, and a second solve that reproduces the first."""
    torch, eng, syn, tpo, E = (env[k] for k in ("torch", "eng", "syn", "tpo", "E"))
    B, D, N = 4096, 6, 2000
    b = syn.make_cartesian_batch(B, D, N)
    inp = syn.upload_cartesian_batch(b, env["dev"])
    out = eng.alloc_joint_outputs(B, N, D, env["dev"])
    E.time_cartesian_paths(inp, out)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy()
    assert (st == 0).all()
    t = out["time"].cpu().numpy(); s = out["s"].cpu().numpy()
    sd = out["sd"].cpu().numpy(); sdd = out["sdd"].cpu().numpy()
    qd = out["qd"].cpu().numpy(); qdd = out["qdd"].cpu().numpy()
    assert np.isfinite(t).all() and np.isfinite(sd).all() and np.isfinite(sdd).all()
    assert (np.diff(t, axis=1) >= 0).all() and (t[:, 0] == 0).all()
    assert (sd[:, 0] == 0).all() and (sd[:, -1] == 0).all() and (sd >= 0).all()
    assert (s[:, 0] == 0).all()
    np.testing.assert_array_equal(s[:, -1], b["delta"] * (N - 1))
    assert torch.equal(out["q"], inp["ik_positions"])
    assert (np.abs(qd) <= 0.8 * b["vmax"][:, None, :] * (1 + 1e-9) + 1e-12).all()
    assert (np.abs(qdd) <= b["amax"][:, None, :]).all()
    # the two Cartesian rows: |(J q')_{1..3}|^2 sd^2 <= v_trans^2, |(J q')_{4..6}|^2 sd^2 <= v_rot^2
    # (no safety factor on them, timeable_path_cartesian_spline.cc:578-592), in blocks of paths
    for lo in range(0, B, 512):
        q = b["ik_positions"][lo:lo + 512]
        q1 = np.zeros_like(q)
        q1[:, :-1] = (q[:, 1:] - q[:, :-1]) / b["delta"][lo:lo + 512, None, None]
        v6 = np.einsum("bnrd,bnd->bnr", b["jacobians"][lo:lo + 512], q1)
        sd2 = sd[lo:lo + 512] ** 2
        assert ((v6[..., :3] ** 2).sum(-1) * sd2 <= (b["vtrans"][lo:lo + 512, None] ** 2) * (1 + 1e-6) + 1e-9).all()
        assert ((v6[..., 3:] ** 2).sum(-1) * sd2 <= (b["vrot"][lo:lo + 512, None] ** 2) * (1 + 1e-6) + 1e-9).all()
    sel = slice(0, B, 131)
    ref = tpo.time_cartesian_batch(b["ik_positions"][sel], b["jacobians"][sel], b["vmax"][sel], b["amax"][sel],
                                   b["vtrans"][sel], b["vrot"][sel], b["path_start"][sel], b["delta"][sel],
                                   nthreads=16)
    assert (ref["status"] == 0).all()
    for k, g in (("t", t), ("s", s), ("sd", sd), ("sdd", sdd), ("qd", qd), ("qdd", qdd)):
        np.testing.assert_array_equal(g[sel], ref[k], err_msg=k)
    np.testing.assert_array_equal(out["last_extremal_index"].cpu().numpy()[sel], ref["last_extremal_index"])
    again = eng.alloc_joint_outputs(B, N, D, env["dev"])
    E.time_cartesian_paths(inp, again)
    torch.cuda.synchronize()
    for k in ("time", "sd", "sdd", "qd", "qdd", "status"):
        assert torch.equal(again[k], out[k]), k


@pytest.mark.parametrize("delay_us", ["0", "3", "60"])
def test_front_stage_delay_changes_no_result(env, monkeypatch, delay_us):
    """TPAMD_FRONT_DELAY_US (the idle kernel in front of a pipelined front stage, DESIGN.md section 4:
    it decides whether the overlap happens, never what is computed): pipelined solves of two alternating
    batches with no, a short and a long delay reproduce the unpipelined engine bit for bit."""
    torch, eng, syn = env["torch"], env["eng"], env["syn"]
    B, D, N = 256, 7, 1200
    batches = [syn.make_joint_batch(B, D, N, first_path_index=f) for f in (0, 3000)]
    inps = [eng.upload_joint_batch(b, env["dev"]) for b in batches]
    refs = []
    for inp in inps:
        ref = eng.alloc_joint_outputs(B, N, D, env["dev"])
        env["E"].time_joint_paths(inp, ref, N)
        refs.append(ref)
    monkeypatch.setenv("TPAMD_FRONT_DELAY_US", delay_us)
    E2 = eng.Engine(0)
    E2.set_pipelining(1)
    outs = [eng.alloc_joint_outputs(B, N, D, env["dev"]) for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(12):
        E2.time_joint_paths(inps[it % 2], outs[it % 2], N)
    torch.cuda.synchronize()
    for out, ref in zip(outs, refs):
        for k in KEYS + ("status", "last_extremal_index"):
            assert torch.equal(out[k], ref[k]), (delay_us, k)
    E2.close()
