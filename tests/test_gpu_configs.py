"""GPU tests of the BASELINE.json configurations at their stated shapes, one GPU's share at a
time, through the same sharding code bench.py uses (sharding.shard_bounds /
balanced_bounds + PipelinedGather at world size 1):

  configs[2]  65536 x 7-DOF x 2000 over 8 GPUs  -> rank share 8192 x 7 x 2000
  configs[4]  mixed 6/7/14-DOF, 500..4000 samples per path, balanced by sum N*C^2

plus the engine-state contracts of the C-ABI (stateful query, per-device setup).
"""
import importlib

import numpy as np
import pytest

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu


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


def test_config2_one_gpu_share_through_the_sharded_pipeline(env):
    """configs[2]: rank 5 of 8 solves its contiguous block of the 65536-path batch into the
    double-buffered gather payload, twice (both pipeline slots). Full-size properties on every
    path (on the device), bit-parity with the oracle on a strided subset."""
    torch, eng, syn, shd, tpo, E = (env[k] for k in ("torch", "eng", "syn", "shd", "tpo", "E"))
    total, world, rank, D, N = 65536, 8, 5, 7, 2000
    lo, hi = shd.shard_bounds(total, world, rank)
    B = hi - lo
    assert (B, lo) == (8192, 5 * 8192)
    batch = syn.make_joint_batch(B, D, N, first_path_index=lo)
    inp = eng.upload_joint_batch(batch, env["dev"])
    G = shd.PipelinedGather((3, B, N), torch.float64, env["dev"], depth=2)
    assert G.world == 1
    shared = eng.alloc_joint_outputs(B, N, D, env["dev"])
    outs = []
    for slot in range(2):
        o = dict(shared)
        o["time"], o["sd"], o["sdd"] = G.send[slot][0], G.send[slot][1], G.send[slot][2]
        outs.append(o)
    for k in range(2):
        G.buffer(k)
        E.time_joint_paths(inp, outs[k], N)
        G.launch(k)
    G.drain()
    torch.cuda.synchronize()
    r0, r1 = G.result(0), G.result(1)
    assert tuple(r0.shape) == (1, 3, B, N)
    assert torch.equal(r0, r1)                      # the two slots hold the same solve
    out = outs[0]
    assert int((out["status"] != 0).sum()) == 0     # every path of the shard solved
    t, s, sd, sdd = out["time"], out["s"], out["sd"], out["sdd"]
    assert bool(torch.isfinite(r0).all())
    assert bool((t[:, 1:] >= t[:, :-1]).all()) and bool((t[:, 0] == 0).all())
    assert bool((sd >= 0).all()) and bool((sd[:, 0] == 0).all()) and bool((sd[:, -1] == 0).all())
    vmax = inp["max_velocity"][:, None, :]
    amax = inp["max_acceleration"][:, None, :]
    assert bool((out["qd"].abs() <= 0.8 * vmax * (1 + 1e-9) + 1e-12).all())
    assert bool((out["qdd"].abs() <= amax).all())
    delta = torch.from_numpy(batch["delta"]).to(env["dev"])
    assert torch.equal(s[:, -1], delta * (N - 1)) and bool((s[:, 0] == 0).all())
    assert bool(((out["q"][:, 0] - inp["control_points"][:, 0]).abs() <= 1e-12).all())
    lei = out["last_extremal_index"]
    assert bool((lei >= 1).all()) and bool((lei <= N - 2).all())
    # strided subset against the oracle, every output bit for bit
    sel = np.arange(0, B, 331)
    ref = tpo.time_joint_batch(batch["knots"][sel], batch["control_points"][sel],
                               batch["vmax"][sel], batch["amax"][sel], batch["path_start"][sel],
                               batch["delta"][sel], N, nthreads=8)
    assert (ref["status"] == 0).all()
    sel_d = torch.from_numpy(sel).to(env["dev"])
    for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
        np.testing.assert_array_equal(out[k][sel_d].cpu().numpy(), ref["t" if k == "time" else k],
                                      err_msg=k)
    np.testing.assert_array_equal(lei[sel_d].cpu().numpy(), ref["last_extremal_index"])


def test_config4_mixed_dof_ragged_share_every_path_matches_its_oracle_run(env):
    """configs[4] at its stated shape: joint counts from {6, 7, 14}, 500..4000 samples per
    path, partitioned over 8 ranks by sum N*C^2 (sharding.balanced_bounds); rank 3's share is
    bucketed by (D, ceil(N/512)), solved bucket by bucket, and EVERY path is compared with the
    oracle run on that path alone with its own sample count."""
    torch, eng, syn, shd, tpo, E = (env[k] for k in ("torch", "eng", "syn", "shd", "tpo", "E"))
    total, world, rank = 1536, 8, 3
    dofs, samples = syn.mixed_batch_shape(total)
    assert set(np.unique(dofs)) == {6, 7, 14} and samples.min() >= 500 and samples.max() <= 4000
    costs = samples.astype(np.float64) * (2.0 * dofs) ** 2
    bounds = shd.balanced_bounds(costs, world)
    assert bounds[0][0] == 0 and bounds[-1][1] == total
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    share_costs = [costs[a:b].sum() for a, b in bounds]
    assert max(share_costs) <= 1.15 * (costs.sum() / world)          # balanced by cost, not count
    lo, hi = bounds[rank]
    groups = syn.mixed_batch_groups(dofs[lo:hi], samples[lo:hi])
    assert {k[0] for k in groups} == {6, 7, 14}
    checked = 0
    for (D, stride), pos in groups.items():
        gidx = lo + pos
        ns = samples[gidx]
        b = syn.make_mixed_group(gidx, D, ns, stride)
        inp = eng.upload_joint_batch(b, env["dev"])
        inp["num_samples_per_path"] = torch.from_numpy(ns).to(env["dev"])
        out = eng.alloc_joint_outputs(len(pos), stride, D, env["dev"])
        for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd"):
            out[k].fill_(-7.0)
        E.time_joint_paths(inp, out, stride)
        torch.cuda.synchronize()
        st = out["status"].cpu().numpy()
        got = {k: out[k].cpu().numpy() for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd")}
        lei = out["last_extremal_index"].cpu().numpy()
        for i, n in enumerate(ns):
            one = [b[k][i:i + 1] for k in ("knots", "control_points", "vmax", "amax",
                                            "path_start", "delta")]
            ref = tpo.time_joint_batch(*one, int(n), nthreads=1)
            assert st[i] == ref["status"][0] == 0, (D, stride, i, st[i])
            assert lei[i] == ref["last_extremal_index"][0]
            for k in got:
                np.testing.assert_array_equal(got[k][i, :n], ref["t" if k == "time" else k][0],
                                              err_msg="%s D=%d n=%d" % (k, D, n))
                assert (got[k][i, n:] == -7.0).all(), "wrote past the path's sample count"
            checked += 1
    assert checked == hi - lo and checked >= 150


def test_query_uses_the_solve_it_is_given_not_the_last_one(env):
    """include/tpamd.h tpamd_query_device: with an explicit sd2 the query is stateless; without
    it the engine refuses (TPAMD_E_STALE) once another solve of the same shape has run."""
    torch, eng, syn, tpo, E = (env[k] for k in ("torch", "eng", "syn", "tpo", "E"))
    D, N, B, K = 7, 400, 4, 65
    f = dict(dtype=torch.float64, device=env["dev"])

    def solve(first):
        b = syn.make_joint_batch(B, D, N, first_path_index=first)
        inp = eng.upload_joint_batch(b, env["dev"])
        out = eng.alloc_joint_outputs(B, N, D, env["dev"])
        out["sd2"] = torch.empty(B, N, **f)
        E.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        return b, out

    b1, o1 = solve(100)
    rng = np.random.default_rng(3)
    tq = np.sort(rng.uniform(-0.1, o1["time"][:, -1:].cpu().numpy() + 0.1, (B, K)), axis=1)
    tq_d = torch.from_numpy(tq).to(env["dev"])

    def query(out, sd2):
        qs, qsd, qsdd = (torch.empty(B, K, **f) for _ in range(3))
        ok = torch.zeros(B, K, dtype=torch.int32, device=env["dev"])
        E.query(out["time"], out["s"], out["sd"], out["status"], tq_d, qs, qsd, qsdd, ok, sd2=sd2)
        torch.cuda.synchronize()
        return qs.cpu().numpy(), qsd.cpu().numpy(), qsdd.cpu().numpy()

    first = query(o1, None)                     # engine state still belongs to solve 1
    b2, o2 = solve(900)                         # same shape, different paths, other buffers
    assert not torch.equal(o1["time"], o2["time"])
    with pytest.raises(eng.TpamdError, match="different solve"):
        query(o1, None)
    again = query(o1, o1["sd2"])                # stateless form: still solve 1's answer
    for a, c in zip(first, again):
        np.testing.assert_array_equal(a, c)
    for i in range(B):                          # and it is the oracle's answer
        q_, q1, q2 = tpo.joint_sample_path(b1["knots"][i], b1["control_points"][i], 0.0,
                                           b1["delta"][i], N)
        rows = tpo.joint_constraint_setup(q1, q2, b1["vmax"][i], b1["amax"][i])
        p = tpo.Profile(N, 2 * D)
        p.set_max_loops(10 * N)
        assert p.setup(*rows, 0.0, b1["delta"][i] * (N - 1)) == 0 and p.optimize() == 0
        ref = np.array([p.query(x)[1:] for x in tq[i]])
        for k in range(3):
            np.testing.assert_array_equal(again[k][i], ref[:, k])
    query(o2, None)                             # the last solve can still use the short form


def test_tiny_batches_do_not_read_past_the_workspace(env):
    """One path, 14 joints, 3..40 samples: the sweep's 32-record tile prefetch reaches past
    the path's last record; the workspace keeps a tile of slack behind the records
    (tpamd_capi.hip carve_workspace). A fresh engine so that nothing larger was reserved."""
    torch, eng, syn, tpo = (env[k] for k in ("torch", "eng", "syn", "tpo"))
    E = eng.Engine(0)
    for D in (14, 7):
        for N in (3, 4, 7, 31, 32, 33, 40):
            b = syn.make_joint_batch(1, D, N)
            inp = eng.upload_joint_batch(b, env["dev"])
            out = eng.alloc_joint_outputs(1, N, D, env["dev"])
            E.time_joint_paths(inp, out, N)
            torch.cuda.synchronize()
            ref = tpo.time_joint_batch(b["knots"], b["control_points"], b["vmax"], b["amax"],
                                       b["path_start"], b["delta"], N)
            assert out["status"].item() == ref["status"][0], (D, N)
            if ref["status"][0] == 0:
                for k in ("time", "sd", "sdd", "qd", "qdd"):
                    np.testing.assert_array_equal(out[k].cpu().numpy(),
                                                  ref["t" if k == "time" else k])
    E.close()


def test_engine_leaves_the_callers_device_current(env):
    torch, eng = env["torch"], env["eng"]
    before = torch.cuda.current_device()
    E2 = eng.Engine(0)
    E2.reserve(8, 100, 14)
    assert torch.cuda.current_device() == before
    E2.close()


@pytest.mark.parametrize("kind,D,N,B,vscale,ascale", [
    ("joint", 7, 700, 6, 1.0, 1.0), ("joint", 14, 300, 3, 1.0, 1.0), ("joint", 3, 130, 4, 1.0, 1.0),
    ("cartesian", 6, 500, 4, 1.0, 1.0),
    # tight velocity limits: isolated velocity-limited samples (the reference's pass 2,
    # time_optimal_path_timing.cc:1386-1395) and the deferred fixes next to them
    ("joint", 7, 700, 8, 0.3, 1.0), ("joint", 7, 2000, 4, 0.3, 1.0), ("joint", 6, 300, 8, 0.5, 5.0),
    ("joint", 7, 2100, 2, 0.3, 1.0), ("joint", 7, 2500, 2, 0.3, 1.0)])
def test_fused_boundary_passes_match_the_oracle_stage_by_stage(env, kind, D, N, B, vscale, ascale):
    """The specialised sweep kernels run CalculateBoundary's passes 2-4 for their path
    (tpamd_sweep_joint.h boundary_passes_for_path). With tpamd_debug_keep_boundary the final
    curve, its sdd range and the classification are stored: bit-equal to the oracle's
    CalculateBoundary on the same rows."""
    torch, eng, syn, tpo = (env[k] for k in ("torch", "eng", "syn", "tpo"))
    E = eng.Engine(0)
    E.debug_keep_boundary(True)
    out = eng.alloc_joint_outputs(B, N, D, env["dev"])
    if kind == "joint":
        b = syn.make_joint_batch(B, D, N, first_path_index=400)
        b["vmax"] = np.ascontiguousarray(b["vmax"] * vscale)
        b["amax"] = np.ascontiguousarray(b["amax"] * ascale)
        E.time_joint_paths(eng.upload_joint_batch(b, env["dev"]), out, N)
    else:
        b = syn.make_cartesian_batch(B, D, N, first_path_index=400)
        E.time_cartesian_paths(syn.upload_cartesian_batch(b, env["dev"]), out)
    torch.cuda.synchronize()
    bd = E.debug_boundary(B, N)
    for i in range(B):
        if kind == "joint":
            _, q1, q2 = tpo.joint_sample_path(b["knots"][i], b["control_points"][i], 0.0, b["delta"][i], N)
            rows = tpo.joint_constraint_setup(q1, q2, b["vmax"][i], b["amax"][i])
            C = 2 * D
        else:
            q1, q2 = tpo.cartesian_path_derivatives(b["ik_positions"][i], b["delta"][i])
            # J q' summed over the joints in index order, as the oracle and the engine do
            jq1 = np.stack([sum(b["jacobians"][i][:, r, d] * q1[:, d] for d in range(D))
                            for r in range(6)], axis=1)
            rows = tpo.cartesian_constraint_setup(q1, q2, np.ascontiguousarray(jq1), b["vmax"][i],
                                                  b["amax"][i], b["vtrans"][i], b["vrot"][i])
            C = 2 * D + 2
        p = tpo.Profile(N, C)
        p.set_max_loops(10 * N)
        assert p.setup(*rows, 0.0, b["delta"][i] * (N - 1)) == 0
        st = p.optimize()
        assert st == out["status"][i].item()
        np.testing.assert_array_equal(bd["sd2_max"][i], p.sd2_max)
        np.testing.assert_array_equal(bd["sdd_max"][i], p.sdd_max_for_sd2_max)
        np.testing.assert_array_equal(bd["sdd_min"][i], p.sdd_min_for_sd2_max)
        np.testing.assert_array_equal(bd["type"][i], p.boundary_type)
        if st == 0:
            np.testing.assert_array_equal(out["time"][i].cpu().numpy(), p.time)
            np.testing.assert_array_equal(out["sdd"][i].cpu().numpy(), p.sdd)
    E.close()


def test_pipelined_engine_gives_the_same_results_in_order(env):
    """tpamd_engine_set_pipelining: the sampling/LP kernel of a solve runs on the engine's own
    stream under the previous solve's sweep (two workspaces). Alternating two different batches
    (and two shapes) through a pipelined engine must reproduce the unpipelined results bit for
    bit, each complete when the caller's stream has passed its call."""
    torch, eng, syn = env["torch"], env["eng"], env["syn"]
    E1, E2 = eng.Engine(0), eng.Engine(0)
    E2.set_pipelining(True)
    cases = []
    for first, D, N, B in ((0, 7, 900, 96), (5000, 7, 900, 96), (77, 6, 333, 40)):
        b = syn.make_joint_batch(B, D, N, first_path_index=first)
        inp = eng.upload_joint_batch(b, env["dev"])
        ref = eng.alloc_joint_outputs(B, N, D, env["dev"])
        E1.time_joint_paths(inp, ref, N)
        cases.append((inp, ref, D, N, B))
    torch.cuda.synchronize()
    keys = ("time", "s", "sd", "sdd", "q", "qd", "qdd", "status", "last_extremal_index",
            "max_time_increment")
    outs = []
    order = [0, 1, 0, 2, 1, 2, 0, 0, 1]
    for k in order:                                   # enqueue everything, synchronise once
        inp, ref, D, N, B = cases[k]
        out = eng.alloc_joint_outputs(B, N, D, env["dev"])
        for name in ("time", "sd", "qdd", "q"):
            out[name].fill_(-3.0)
        torch.cuda.synchronize()                      # the fill is done before the call (contract)
        E2.time_joint_paths(inp, out, N)
        outs.append(out)
    torch.cuda.synchronize()
    for k, out in zip(order, outs):
        for name in keys:
            assert torch.equal(out[name], cases[k][1][name]), (k, name)
    # back to back without host synchronisation in between, outputs reused every other call
    inp, ref, D, N, B = cases[0]
    inp1, ref1 = cases[1][0], cases[1][1]
    a, c = eng.alloc_joint_outputs(B, N, D, env["dev"]), eng.alloc_joint_outputs(B, N, D, env["dev"])
    for it in range(40):
        E2.time_joint_paths(inp, a, N)
        E2.time_joint_paths(inp1, c, N)
    torch.cuda.synchronize()
    for name in keys:
        assert torch.equal(a[name], ref[name]) and torch.equal(c[name], ref1[name]), name
    # a call captured into a HIP graph cannot fork into the engine's stream: it runs unpipelined
    E2.reserve(B, N, 2 * D)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = eng.alloc_joint_outputs(B, N, D, env["dev"])
    with torch.cuda.stream(side):
        E2.time_joint_paths(inp, g, N, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        E2.time_joint_paths(inp1, g, N, stream=side)
    g["time"].fill_(-1.0)
    graph.replay()
    torch.cuda.synchronize()
    for name in keys:
        assert torch.equal(g[name], ref1[name]), name
    # mode 2: completion deferred by one call, fence() for the rest
    E2.set_pipelining(2)
    for it in range(6):
        E2.time_joint_paths(inp, a, N)
        E2.time_joint_paths(inp1, c, N)
    E2.fence()
    torch.cuda.synchronize()
    for name in keys:
        assert torch.equal(a[name], ref[name]) and torch.equal(c[name], ref1[name]), name
    E2.set_pipelining(0)
    E2.time_joint_paths(inp, a, N)
    torch.cuda.synchronize()
    assert torch.equal(a["time"], ref["time"])
    E1.close(); E2.close()


def test_pose_spline_sampling_matches_the_oracle(env):
    """tpamd_sample_pose_splines_*: BSplineQ::EvalCurve (splines/bsplineq.cc:223-244) and the
    translation spline for Cartesian batches. The basis and the translations are bit-equal to the
    oracle; the quaternions go through log/atan2/sin/cos/exp of the device math library and are
    held to 1e-12 (the reference's own IsApprox tolerance for this code, north star: 1e-6)."""
    eng, syn, tpo = env["eng"], env["syn"], env["tpo"]
    E = env["E"]
    rng = np.random.default_rng(7)
    B, W, N = 12, 6, 700
    P = 3 * W - 2
    knots = np.zeros((B, P + 3)); tr = np.zeros((B, P, 3)); rot = np.zeros((B, P, 4))
    for b in range(B):
        jb = syn.make_joint_batch(1, 3, N, num_waypoints=W, first_path_index=700 + b)
        knots[b] = jb["knots"][0]
        tr[b] = jb["control_points"][0]
        q = rng.normal(size=(P, 4))
        if b % 3 == 0:
            q[5] = q[4]                                  # identical neighbours: |v| = 0 branch of QuatLog
        if b % 4 == 1:
            q[7] = -q[6]                                 # antipodal representation of the same rotation
        rot[b] = q / np.linalg.norm(q, axis=1, keepdims=True)
    delta = knots[:, -1] / (N - 40)                      # the last samples run past the spline
    start = np.where(np.arange(B) % 2 == 0, 0.0, 0.3 * delta)
    poses = E.sample_pose_splines(knots, tr, rot, start, delta, N)
    past_end = 0
    for b in range(B):
        ref = tpo.sample_pose_spline(knots[b], tr[b], rot[b], start[b], delta[b], N)
        np.testing.assert_array_equal(poses[b, :, :3], ref[:, :3])
        assert np.max(np.abs(poses[b, :, 3:] - ref[:, 3:])) <= 1e-12
        assert np.max(np.abs(np.linalg.norm(poses[b, :, 3:], axis=1) - 1.0)) <= 1e-12
        past_end += int((poses[b, :, :3] == tr[b, -1]).all(axis=1).sum())
    assert past_end >= B * 30


def test_sweep_and_sampling_kernels_fit_one_simd_together(env):
    """The pipelined modes run the sampling/LP kernel of the next solve beside the resident sweep
    workgroups: one of its waves (and its 16 KB of LDS) must fit next to TWO sweep waves on a SIMD's
    512 registers. A sweep build above 208 VGPRs silently loses that (measured: 227 VGPRs, pipelined
    step 0.511 -> 0.534 ms)."""
    E = env["E"]
    k1, k2 = E.debug_kernel_vgprs(0), E.debug_kernel_vgprs(1)
    assert 0 < k1 <= 128 and 0 < k2 <= 256, (k1, k2)
    gran = 8                                    # VGPR allocation granule of gfx950 (wave64)
    up = lambda n: (n + gran - 1) // gran * gran
    assert 2 * up(k2) + up(k1) <= 512, "sweep %d + sampling/LP %d VGPRs do not fit one SIMD" % (k2, k1)


@pytest.mark.parametrize("shards,B,N,ragged", [(1, 64, 2000, False), (3, 40, 777, False), (2, 33, 500, True)])
def test_time_rebuilt_from_sd_equals_the_solved_time(env, shards, B, N, ragged):
    """tpamd_rebuild_time_device (the root of a multi-GPU job rebuilds t from the gathered sd):
    bit-identical to the solve's own time output, in the sharded payload layout bench.py uses
    (sd | sdd | ds | time_start per shard, shard_stride doubles apart)."""
    torch, eng, syn, E, dev = env["torch"], env["eng"], env["syn"], env["E"], env["dev"]
    D = 7
    flat = 2 * B * N + 2 * B
    payload = torch.zeros(shards, flat, dtype=torch.float64, device=dev)
    want = torch.empty(shards * B, N, dtype=torch.float64, device=dev)
    counts = torch.full((shards * B,), N, dtype=torch.int32, device=dev)
    for r in range(shards):
        b = syn.make_joint_batch(B, D, N, first_path_index=1000 * r)
        b["time_start"] = np.linspace(0.0, 3.0, B) * (r + 1)
        if ragged:
            b["num_samples_per_path"] = (N - (np.arange(B) * 7) % 200).astype(np.int32)
            b["delta"] = np.ascontiguousarray(b["knots"][:, -1] / (b["num_samples_per_path"] - 1))
        inp = eng.upload_joint_batch(b, dev)
        if ragged:
            inp["num_samples_per_path"] = torch.as_tensor(b["num_samples_per_path"], device=dev)
        out = eng.alloc_joint_outputs(B, N, D, dev)
        out["time"].fill_(-1.0)
        E.time_joint_paths(inp, out, N)
        torch.cuda.synchronize()
        assert int((out["status"] == 0).sum()) == B
        payload[r, :B * N] = out["sd"].reshape(-1)
        nb = b["num_samples_per_path"].astype(np.float64) - 1 if ragged else np.full(B, N - 1.0)
        ps = torch.as_tensor(b["path_start"], dtype=torch.float64, device=dev)
        dl = torch.as_tensor(b["delta"], dtype=torch.float64, device=dev)
        nbt = torch.as_tensor(nb, dtype=torch.float64, device=dev)
        payload[r, 2 * B * N:2 * B * N + B] = ((ps + dl * nbt) - ps) / nbt
        payload[r, 2 * B * N + B:] = torch.as_tensor(b["time_start"], dtype=torch.float64, device=dev)
        want[r * B:(r + 1) * B] = out["time"]
        if ragged:
            counts[r * B:(r + 1) * B] = torch.as_tensor(b["num_samples_per_path"], device=dev)
    got = torch.full((shards * B, N), -1.0, dtype=torch.float64, device=dev)
    E.rebuild_time(payload[0, :B * N], payload[0, 2 * B * N:2 * B * N + B], payload[0, 2 * B * N + B:],
                   got, shards, B, N, flat, num_samples_per_path=counts if ragged else None)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
