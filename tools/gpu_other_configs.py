"""Single-GPU timings of the BASELINE.json configurations other than the bench line's:
configs[2]'s per-GPU share (8192 7-DOF paths), configs[3] (4096 6-DOF Cartesian paths) and
a one-GPU share of configs[4] (mixed 6/7/14-DOF, 500..4000 samples per path). Each case is
checked against the oracle on a sample of its paths before it is timed. One JSON line per
case; run on the GPU box:  python tools/gpu_other_configs.py > gpurun_out/other_configs.jsonl
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import tpo  # noqa: E402

eng = importlib.import_module("x-edr-trajectory-planning_amd.engine")
syn = importlib.import_module("x-edr-trajectory-planning_amd.synthetic")
DEV = "cuda:0"
E = eng.Engine(0)


def timed(fn, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    E.profile_reset()
    E.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    E.profile_enable(False)
    return el, {k: round(v[0], 4) for k, v in E.profile_summary().items()}


def joint_case(name, B, D, N, check=32):
    b = syn.make_joint_batch(B, D, N)
    inp = eng.upload_joint_batch(b, DEV)
    out = eng.alloc_joint_outputs(B, N, D, DEV)
    el, kern = timed(lambda: E.time_joint_paths(inp, out, N))
    sub = {k: b[k][:check] for k in ("knots", "control_points", "vmax", "amax", "path_start", "delta")}
    ref = tpo.time_joint_batch(sub["knots"], sub["control_points"], sub["vmax"], sub["amax"],
                               sub["path_start"], sub["delta"], N, nthreads=16)
    exact = all(np.array_equal(out[k][:check].cpu().numpy(), ref["t" if k == "time" else k])
                for k in ("time", "s", "sd", "sdd", "qd", "qdd"))
    print(json.dumps({"case": name, "paths": B, "dofs": D, "samples": N,
                      "ms_per_batch": round(el * 1e3, 3), "paths_per_s": round(B / el, 1),
                      "solved": int((out["status"] == 0).sum()), "oracle_sample": check,
                      "bit_exact_on_sample": bool(exact), "kernels_ms": kern}), flush=True)


def cartesian_case(name, B, D, N, check=32):
    b = syn.make_cartesian_batch(B, D, N)
    inp = syn.upload_cartesian_batch(b, DEV)
    out = eng.alloc_joint_outputs(B, N, D, DEV)
    el, kern = timed(lambda: E.time_cartesian_paths(inp, out), steps=5)
    ref = tpo.time_cartesian_batch(b["ik_positions"][:check], b["jacobians"][:check], b["vmax"][:check],
                                   b["amax"][:check], b["vtrans"][:check], b["vrot"][:check],
                                   b["path_start"][:check], b["delta"][:check], nthreads=16)
    st = out["status"][:check].cpu().numpy()
    ok = st == 0
    exact = np.array_equal(st, ref["status"]) and all(
        np.array_equal(out[k][:check].cpu().numpy()[ok], ref["t" if k == "time" else k][ok])
        for k in ("time", "s", "sd", "sdd", "qd", "qdd"))
    t0 = time.perf_counter()
    tpo.time_cartesian_batch(b["ik_positions"][:256], b["jacobians"][:256], b["vmax"][:256],
                             b["amax"][:256], b["vtrans"][:256], b["vrot"][:256],
                             b["path_start"][:256], b["delta"][:256], nthreads=64)
    cpu = 256 / (time.perf_counter() - t0)
    print(json.dumps({"case": name, "paths": B, "dofs": D, "samples": N, "rows": 2 * D + 2,
                      "ms_per_batch": round(el * 1e3, 3), "paths_per_s": round(B / el, 1),
                      "solved": int((out["status"] == 0).sum()), "oracle_sample": check,
                      "bit_exact_on_sample": bool(exact), "kernels_ms": kern,
                      "cpu_oracle_64_threads_paths_per_s": round(cpu, 1)}), flush=True)


def mixed_case(name, per_group, check=8, bucket=0):
    """3 x per_group paths of 6/7/14 joints, 500..4000 samples each. Timed two ways: the DOF groups
    one after another at their common stride (how round 2 ran it), and as groups side by side
    (tpamd_time_joint_groups_device): one group per joint count (bucket = 0, the fastest: 1.52 ms),
    or bucketed by (D, ceil(N / bucket)) -- measured slower the finer the buckets (TPAMD_BUCKET=2560:
    1.93 ms, 512: 3.7 ms; every bucket costs four launches and the runtime has four hardware queues)."""
    rng = np.random.default_rng(11)
    groups, buckets = [], []
    for D in (6, 7, 14):
        ns = rng.integers(500, 4001, size=per_group).astype(np.int32)
        stride = int(ns.max())
        b = syn.make_joint_batch(per_group, D, stride)
        b["delta"] = b["knots"][:, -1] / (ns - 1)
        inp = eng.upload_joint_batch(b, DEV)
        inp["num_samples_per_path"] = torch.from_numpy(ns).to(DEV)
        out = eng.alloc_joint_outputs(per_group, stride, D, DEV)
        groups.append((D, ns, stride, b, inp, out))
        edges = -(-ns // bucket) * bucket if bucket else np.full_like(ns, stride)
        for edge in np.unique(edges):
            pos = np.nonzero(edges == edge)[0]
            bs = min(int(edge), stride)
            binp = {k: v[torch.from_numpy(pos).to(DEV)].contiguous() for k, v in inp.items()}
            bout = eng.alloc_joint_outputs(len(pos), bs, D, DEV)
            buckets.append(dict(D=D, pos=pos, inputs=binp, outputs=bout, num_samples=bs))

    def run_sequential():
        for D, ns, stride, b, inp, out in groups:
            E.time_joint_paths(inp, out, stride)

    def run_groups():
        E.time_joint_groups(buckets)

    def clock(fn, steps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    el_seq = clock(run_sequential)
    el = clock(run_groups)
    exact, solved, samples, same = True, 0, 0, True
    for D, ns, stride, b, inp, out in groups:
        samples += int(ns.sum())
        for i in range(check):
            one = {k: b[k][i:i + 1] for k in ("knots", "control_points", "vmax", "amax", "path_start", "delta")}
            ref = tpo.time_joint_batch(one["knots"], one["control_points"], one["vmax"], one["amax"],
                                       one["path_start"], one["delta"], int(ns[i]))
            if ref["status"][0] != int(out["status"][i]):
                exact = False
            elif ref["status"][0] == 0:
                for k in ("time", "sd", "qdd"):
                    exact = exact and np.array_equal(out[k][i, :ns[i]].cpu().numpy(),
                                                     ref["t" if k == "time" else k][0])
        # every path of the bucketed run against the sequential run of the same engine
        for bk in buckets:
            if bk["D"] != D:
                continue
            st = bk["outputs"]["status"].cpu().numpy()
            solved += int((st == 0).sum())
            for j, p_ in enumerate(bk["pos"]):
                n = int(ns[p_])
                for k in ("time", "sd", "sdd", "qd", "qdd"):
                    same = same and torch.equal(bk["outputs"][k][j, :n], out[k][p_, :n])
    B = 3 * per_group
    print(json.dumps({"case": name, "paths": B, "dofs": [6, 7, 14], "samples": "500..4000 per path",
                      "total_samples": samples, "buckets": len(buckets),
                      "ms_per_batch": round(el * 1e3, 3), "paths_per_s": round(B / el, 1),
                      "samples_per_s": round(samples / el, 1),
                      "dof_groups_in_turn_ms": round(el_seq * 1e3, 3),
                      "dof_groups_in_turn_paths_per_s": round(B / el_seq, 1),
                      "solved": solved, "oracle_sample": 3 * check,
                      "bit_exact_on_sample": bool(exact),
                      "buckets_equal_dof_groups_on_every_path": bool(same)}), flush=True)


if __name__ == "__main__":
    cases = sys.argv[1:] or ["1", "2", "3", "4"]     # which BASELINE.json configs to run
    print(json.dumps({"device": torch.cuda.get_device_name(0)}), flush=True)
    if "1" in cases:
        joint_case("configs[1] 1024 x 7-DOF x 2000", 1024, 7, 2000)
    if "2" in cases:
        joint_case("configs[2] per-GPU share: 8192 x 7-DOF x 2000", 8192, 7, 2000)
    if "3" in cases:
        cartesian_case("configs[3] 4096 x 6-DOF Cartesian x 2000", 4096, 6, 2000)
    if "4" in cases:
        mixed_case("configs[4] one-GPU share: 3 x 512 paths, 6/7/14-DOF, ragged", 512,
                   bucket=int(os.environ.get("TPAMD_BUCKET", "0")))
