#!/usr/bin/env python3
"""Benchmark of the batched time-optimal path-timing hot path on MI355X.

One "step" = one pass of the hot path (B-spline sampling -> constraint rows -> LP
boundary curve -> extremal sweeps -> time integration -> qd/qdd epilogue) over one
batch of synthetic 7-DOF, 2000-sample joint-space paths already resident in HBM
(BASELINE.json configs[1]: 1024 paths per GPU). With N > 1 every rank (one process
per GPU) times its own shard of N*1024 paths and the packed timing profile
(t, sd, sdd) is collected on rank 0 by ONE RCCL gather inside the timed region.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "x-edr-trajectory-planning_amd"

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def baseline_metric():
    """The metric string of BASELINE.json, verbatim."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "time-optimal path timings/s (7-DOF, 2000 s-samples), 1/2/4/8 MI355X + %HBM roofline"


def algorithmic_bytes_per_path(D, N, P):
    """SURVEY.md 8(d) primary figure: inputs 8*(P*D + P+3 + 2D + 4) plus the outputs the
    north star names, t, s, sd, q: 8*N*(3+D)."""
    return 8 * (P * D + P + 3 + 2 * D + 4) + 8 * N * (3 + D)


def cpu_baseline(batch, N, D):
    """The oracle (a C port of the reference algorithm; the reference itself cannot be
    built in this image) timed on this box's host cores on the same inputs."""
    from oracle import tpo
    import numpy as np
    # one GPU's share of the host is 16 cores on the measurement boxes: stay within it
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, 16))
    B = batch["control_points"].shape[0]

    def run(n, nthreads):
        return tpo.time_joint_batch(batch["knots"][:n], batch["control_points"][:n],
                                    batch["vmax"][:n], batch["amax"][:n],
                                    batch["path_start"][:n], batch["delta"][:n], N,
                                    nthreads=nthreads)

    # single thread = the reference's execution model; bounded sample
    n1 = min(B, 128)
    run(8, 1)                                                    # warm-up
    t0 = time.perf_counter()
    r = run(n1, 1)
    single = n1 / (time.perf_counter() - t0)
    assert (r["status"] == 0).all()
    run(B, nthreads)                                             # warm-up
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(B, nthreads)
        rates.append(B / (time.perf_counter() - t0))
    return {
        "value": round(statistics.median(rates), 1), "unit": "paths/s", "cores": nthreads,
        "kind": "port",
        "sample": "%d of the step's %d paths x3 on %d OpenMP threads (median); single thread "
                  "on %d paths: %.1f paths/s" % (B, B, nthreads, n1, single),
        "single_thread_paths_per_s": round(single, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths-per-gpu", type=int, default=1024)
    ap.add_argument("--dofs", type=int, default=7)
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal on a one-GPU box: TPAMD_BENCH_DEVICE=0 puts every rank on the same card
    dev_index = int(os.environ.get("TPAMD_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TPAMD_BENCH_BACKEND", "nccl")   # "gloo": one-GPU rehearsal only
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    eng = importlib.import_module(PKG + ".engine")
    syn = importlib.import_module(PKG + ".synthetic")
    shd = importlib.import_module(PKG + ".sharding")
    if not os.path.exists(os.path.join(ROOT, PKG, "csrc", "libtpamd.so")):
        if rank == 0:
            eng.build_library()
        if distributed:
            dist.barrier()

    B, D, N = args.paths_per_gpu, args.dofs, args.samples
    total_paths = B * world
    lo, hi = shd.shard_bounds(total_paths, world, rank)
    batch = syn.make_joint_batch(hi - lo, D, N, first_path_index=lo)
    P = batch["control_points"].shape[1]
    E = eng.Engine(dev_index)
    E.reserve(B, N, 2 * D)
    inp = eng.upload_joint_batch(batch, dev)
    # timing profile packed as [3][B][N] = (t, sd, sdd) so that the multi-GPU collection is ONE
    # gather per batch (s is not sent: it is the arithmetic sequence s_start + i*ds of
    # time_optimal_path_timing.cc:540-547, which the root rebuilds from per-path scalars);
    # two output buffers, so that the gather of batch k (rank 0's inbound xGMI links)
    # overlaps the solve of batch k+1
    G = shd.PipelinedGather((3, B, N), torch.float64, dev, depth=2)
    shared = eng.alloc_joint_outputs(B, N, D, dev)
    outs = []
    for slot in range(2):
        o = dict(shared)
        p = G.send[slot]
        o["time"], o["sd"], o["sdd"] = p[0], p[1], p[2]
        outs.append(o)
    counter = [0]

    def step():
        k = counter[0]
        counter[0] += 1
        G.buffer(k)                       # the gather that last read this buffer is done
        E.time_joint_paths(inp, outs[k % 2], N)
        G.launch(k)

    for _ in range(args.warmup):
        step()
    G.drain()
    torch.cuda.synchronize()
    out = outs[0]
    ok = int((out["status"] == 0).sum())

    timing = not args.no_kernel_timing
    E.profile_reset()
    E.profile_enable(2 if timing else False)   # timed region: events around the dominant kernel only
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    G.drain()                             # every batch's gather has landed on rank 0
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    E.profile_enable(False)
    dominant_ms = E.profile_mean_ms(eng.KERNEL_SWEEP)[0] if timing else 0.0
    if timing and rank == 0:
        # the other kernels' durations, outside the timed region (events around every kernel
        # cost a few per cent of a step)
        E.profile_reset()
        E.profile_enable(1)
        for _ in range(5):
            E.time_joint_paths(inp, outs[0], N)
        torch.cuda.synchronize()
        E.profile_enable(False)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    okt = torch.tensor([ok], dtype=torch.int64, device=dev)
    if distributed:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(okt, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    solved = int(okt.item())

    if rank == 0:
        kernels = E.profile_summary() if timing else {}
        ms_per_step = elapsed / args.steps * 1e3
        value = total_paths * args.steps / elapsed
        # dominant kernel = the one with the largest mean duration
        roofline = None
        if kernels:
            dom = max(kernels.items(), key=lambda kv: kv[1][0])
            # k_sweep is timed live over the K timed steps; should another kernel ever be the
            # longest (tiny --samples), its duration comes from the separate pass
            dom_ms = dominant_ms if dom[0] == "k_sweep" else dom[1][0]
            alg = algorithmic_bytes_per_path(D, N, P) * B      # bytes per launch (B paths)
            achieved = alg / (dom_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    key = "%s:B%d:D%d:N%d" % (dom[0], B, D, N)
                    traffic = tj.get(key)
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 3),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                        "kernel_ms": round(dom_ms, 4),
                        "algorithmic_bytes_per_launch": alg,
                        "all_kernels_ms": {k: round(v[0], 4) for k, v in kernels.items()}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(batch, N, D)
        line = {
            "metric": baseline_metric(),
            "value": round(value, 1), "unit": "paths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: %d random %d-DOF joint-space "
                                   "B-spline paths per GPU, %d s-samples each (10 waypoints, "
                                   "%d control points), inputs resident in HBM" % (B, D, N, P),
                       "paths_per_gpu": B, "total_paths": total_paths, "num_dofs": D,
                       "num_samples": N, "solved_paths": solved,
                       "gather": ("one RCCL gather of the packed timing profile "
                                  "(t,sd,sdd: 3*N*8 B/path; s = s_start + i*ds is rebuilt on the root) to rank 0 per step, overlapped "
                                  "with the next step's solve (double-buffered), all inside "
                                  "the timed region"
                                  if distributed else "none (single GPU)")},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
