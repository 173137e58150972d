#!/usr/bin/env python3
"""Benchmark of the batched time-optimal path-timing hot path on MI355X.

One "step" = one pass of the hot path (B-spline sampling -> constraint rows -> LP
boundary curve -> extremal sweeps -> time integration -> qd/qdd) over one batch of
synthetic 7-DOF, 2000-sample joint-space paths already resident in HBM.

  python bench.py                         BASELINE.json configs[1]: 1024 paths on one GPU
  ... --gpus N (under torch.distributed.run)   the same 1024 paths PER GPU ("weak" scaling), one
                                          contiguous shard of the N*1024-path batch per rank, ONE
                                          RCCL gather of the timing profile to rank 0 per step
  ... --workload configs2                 BASELINE.json configs[2]: 8192 paths per GPU
                                          (65536 over 8 GPUs)
  ... --gather full                       gather q(t) as well (t, sd, sdd, q), in chunks

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import hashlib
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
# fp64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-operations/s
# (78.6 TFLOP/s when every operation is an FMA), i.e. one wave64 VALU instruction per SIMD
# every 4 cycles.
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9
WORKLOADS = {"configs1": 1024, "configs2": 8192}


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


def gather_bytes_per_path(mode, D, N):
    """Bytes one path contributes to the gather payload. minimal: sd and the scalars (ds,
    time_start) -- the root rebuilds t from sd (and s = s_start + i*ds, an arithmetic sequence);
    compact: sdd as well; profile: (t, sd, sdd); full: q as well."""
    if mode == "none":
        return 0
    if mode == "minimal":
        return 8 * N + 16
    if mode == "compact":
        return 8 * N * 2 + 16
    return 8 * N * 3 + (8 * N * D if mode == "full" else 0)


def kernel_source_hash():
    """SHA-256 over csrc/*.hip and *.h; tools/profile_step.py stores the same value next to
    the counters it collects."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, PKG, "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            with open(os.path.join(csrc, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def recorded_counters(B, D, N):
    """profiles/counters.json (rocprofv3 PMC summary written by tools/profile_step.py) if it was
    measured on exactly these kernel sources and this batch shape, else None."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        with open(path) as f:
            c = json.load(f)
    except (OSError, ValueError):
        return None
    if c.get("source_sha256") != kernel_source_hash():
        return None
    if c.get("workload") != "B%d:D%d:N%d" % (B, D, N):
        return None
    return c


def cpu_baseline(batch, N, D):
    """The oracle (a C port of the reference algorithm; the reference itself cannot be
    built in this image) timed on this box's host cores on a bounded sample of the step's
    inputs."""
    from oracle import tpo
    # one GPU's share of the host is 16 cores on the measurement boxes: stay within it
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, 16))
    B = min(batch["control_points"].shape[0], 1024)

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
        "sample": "the first %d paths of the step's batch x3 on %d OpenMP threads (median); "
                  "single thread on %d paths: %.1f paths/s" % (B, nthreads, n1, single),
        "single_thread_paths_per_s": round(single, 1),
    }


def roofline_block(kernels, dominant_ms, B, D, N, P, ms_per_step):
    """kernels: {name: (mean ms, launches)} from the engine's HIP events. The HBM roofline of
    the dominant kernel as the contract asks, plus what actually limits each kernel."""
    dom = max(kernels.items(), key=lambda kv: kv[1][0])
    # k_sweep is timed live over the K timed steps; should another kernel ever be the longest
    # (tiny --samples), its duration comes from the separate pass
    dom_ms = dominant_ms if (dom[0] == "k_sweep" and dominant_ms > 0) else dom[1][0]
    alg = algorithmic_bytes_per_path(D, N, P) * B          # bytes per launch (B paths)
    achieved = alg / (dom_ms * 1e-3) / 1e9
    rec = recorded_counters(B, D, N)
    traffic = step_traffic = step_frac = valu = None
    per_kernel = {}
    for name, (ms, n) in kernels.items():
        if n == 0:
            continue
        per_kernel[name] = {"ms": round(ms, 4)}
    if rec:
        rk = rec["kernels"]
        traffic = rk.get(dom[0], {}).get("hbm_bytes")
        step_traffic = sum(k.get("hbm_bytes", 0) for k in rk.values())
        step_frac = round(step_traffic / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
        for name, k in rk.items():
            e = per_kernel.setdefault(name, {"ms": round(k.get("avg_us", 0.0) / 1e3, 4)})
            ms = e["ms"] or k.get("avg_us", 0.0) / 1e3
            if "hbm_bytes" in k and ms > 0:
                e["hbm_bytes"] = k["hbm_bytes"]
                e["hbm_frac"] = round(k["hbm_bytes"] / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
            if "valu_insts" in k and ms > 0:
                e["valu_frac"] = round(k["valu_insts"] * 64 / (ms * 1e-3) / VALU_PEAK_LANE_OPS, 4)
            if "valu_busy_pct" in k:
                e["valu_busy_pct"] = k["valu_busy_pct"]
            hb, vb = e.get("hbm_frac", 0.0), e.get("valu_busy_pct", 0.0) / 100.0
            e["limiter"] = ("fp64 VALU issue" if vb >= 0.6 else
                            "HBM bandwidth" if hb >= 0.5 else
                            "per-wave instruction issue (an in-order stream issues one instruction per "
                            "~4.4 cycles; neither the VALU nor the HBM is saturated at this occupancy)")
        d = rk.get(dom[0], {})
        if "valu_insts" in d:
            valu = {"kernel": dom[0], "insts_per_launch": d["valu_insts"],
                    "lane_ops_per_s": round(d["valu_insts"] * 64 / (dom_ms * 1e-3), 1),
                    "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                    "frac": round(d["valu_insts"] * 64 / (dom_ms * 1e-3) / VALU_PEAK_LANE_OPS, 4),
                    "busy_pct": d.get("valu_busy_pct"),
                    "note": "SQ_INSTS_VALU x 64 lanes / kernel time against the fp64 vector issue "
                            "peak (one wave64 instruction per SIMD per 4 cycles)"}
    return {
        "bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
        "kernel_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": alg,
        "limiter": per_kernel.get(dom[0], {}).get("limiter"),
        "step_traffic": step_traffic, "step_hbm_frac": step_frac, "valu": valu,
        "counters": ({"file": "profiles/counters.json", "tag": rec["tag"],
                      "source_sha256": rec["source_sha256"][:16]} if rec else
                     "profiles/counters.json was not measured on these kernel sources / this "
                     "batch shape: traffic and valu are null (tools/profile_step.py refreshes it)"),
        "kernels": per_kernel,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1",
                    help="configs1: 1024 paths per GPU (BASELINE.json configs[1]); configs2: 8192 "
                         "paths per GPU (configs[2], 65536 over 8 GPUs)")
    ap.add_argument("--paths-per-gpu", type=int, default=0, help="override the workload's batch")
    ap.add_argument("--dofs", type=int, default=7)
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--gather", choices=("minimal", "compact", "profile", "full"), default="minimal",
                    help="multi-GPU payload: minimal = sd + two scalars per path, the root rebuilds t "
                         "from sd with the solver's own operations inside the timed region and s from ds "
                         "(16 KB per path through the root's xGMI links: of north_star's t, s, sd, q "
                         "everything but q); compact = sdd as well (32 KB); profile = (t, sd, sdd), 48 KB; "
                         "full = q as well, 160 KB")
    ap.add_argument("--pipeline", type=int, choices=(0, 1, 2), default=1,
                    help="engine pipelining across steps (tpamd_engine_set_pipelining): 0 one kernel "
                         "at a time; 1 (default) the sampling/LP kernel of step k+1 runs under the sweep of "
                         "step k; 2 the sweep of step k+1 may also start while the slowest "
                         "paths of step k are still running")
    ap.add_argument("--no-pipeline", action="store_true", help="same as --pipeline 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-stride", type=int, default=4,
                    help="HIP events around the dominant kernel on every n-th timed step (an event "
                         "pair costs ~16 us of a 0.55 ms pipelined step)")
    ap.add_argument("--clock-warmup-ms", type=float, default=300.0,
                    help="untimed solves of the same batch before the W warm-up steps, until this much "
                         "wall time has passed: the GPU's clocks need a few hundred ms of load to settle "
                         "(0.56 -> 0.52 ms per step measured); 0 = off. Reported in config.")
    args = ap.parse_args()

    import numpy as np  # noqa: F401
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

    B = args.paths_per_gpu if args.paths_per_gpu > 0 else WORKLOADS[args.workload]
    D, N = args.dofs, args.samples
    total_paths = B * world
    lo, hi = shd.shard_bounds(total_paths, world, rank)
    assert hi - lo == B
    batch = syn.make_joint_batch(B, D, N, first_path_index=lo)
    P = batch["control_points"].shape[1]
    E = eng.Engine(dev_index)
    # Steps are independent batches whose inputs are resident before the timed region starts, so
    # the engine may run the front stage of step k+1 (sampling + LP) under the sweep of step k;
    # every step still runs every kernel, and the timed region ends when the last step is done.
    mode = 0 if args.no_pipeline else args.pipeline
    pipelined = mode != 0
    E.set_pipelining(mode)
    E.reserve(B, N, 2 * D)
    inp = eng.upload_joint_batch(batch, dev)
    # Gather payload, packed so that the multi-GPU collection is ONE gather per batch:
    #   profile: [3][B][N] = (t, sd, sdd)            48 KB/path at N = 2000
    #   full:    [3 + D][B][N]-sized block = (t, sd, sdd | q [B][N][D])   104 KB/path at D = 7
    # (s is never sent: it is the arithmetic sequence s_start + i*ds of
    # time_optimal_path_timing.cc:540-547, which the root rebuilds from per-path scalars.)
    # Two buffers, so that the gather of batch k (rank 0's inbound xGMI links) overlaps the solve
    # of batch k+1.
    compact = args.gather in ("compact", "minimal")     # t is rebuilt on the root
    with_sdd = args.gather == "compact"                 # sdd is part of the payload
    shared = eng.alloc_joint_outputs(B, N, D, dev, with_q=(args.gather != "full"))
    # mode 2 lets the sweeps of steps k and k+1 overlap: every output array then exists once per
    # slot (in the other modes the sweeps run in order on one stream and the arrays that are not
    # part of the payload are shared)
    per_slot = [shared, eng.alloc_joint_outputs(B, N, D, dev, with_q=(args.gather != "full")) if mode == 2
                else shared]
    outs = []
    if compact:
        # flat payload per rank: sd [B][N] | (compact: sdd [B][N] |) ds [B] | time_start [B]
        off_ds = (2 if with_sdd else 1) * B * N
        flat = off_ds + 2 * B
        root_time = None

        def rebuild(slot):
            # on the root, right after the gather of this slot has landed: t of every shard's
            # paths from its sd, ds and time_start (own shard included: one launch)
            r0 = G.recv[slot]
            E.rebuild_time(r0[0, :B * N], r0[0, off_ds:off_ds + B], r0[0, off_ds + B:],
                           root_time[slot], world, B, N, flat)

        G = shd.PipelinedGather((flat,), torch.float64, dev, depth=2,
                                on_complete=rebuild if distributed else None)
        if distributed and rank == 0:
            root_time = [torch.empty(world * B, N, dtype=torch.float64, device=dev) for _ in range(2)]
        # ds exactly as the set-up kernel forms it (tpamd_kernels.h k_setup_joint):
        # s1 = s0 + delta (N - 1); ds = (s1 - s0) / (N - 1) -- in numpy (IEEE fp64 division; a
        # torch division by a Python scalar multiplies by the reciprocal, which is a bit off for
        # 40 % of the paths and shows up in the last bits of t)
        ps_h = np.asarray(batch["path_start"], dtype=np.float64)
        dl_h = np.asarray(batch["delta"], dtype=np.float64)
        ds_h = ((ps_h + dl_h * np.float64(N - 1)) - ps_h) / np.float64(N - 1)
        ds_t = torch.from_numpy(np.ascontiguousarray(ds_h)).to(dev)
        t0_t = torch.as_tensor(np.asarray(batch["time_start"], dtype=np.float64), device=dev)
        for slot in range(2):
            o = dict(per_slot[slot])
            p = G.send[slot]
            o["sd"] = p[:B * N].view(B, N)
            if with_sdd:
                o["sdd"] = p[B * N:2 * B * N].view(B, N)      # (minimal: sdd stays in the solve's own buffer)
            p[off_ds:off_ds + B].copy_(ds_t)
            p[off_ds + B:].copy_(t0_t)
            outs.append(o)
    else:
        rows = 3 + (D if args.gather == "full" else 0)
        G = shd.PipelinedGather((rows, B, N), torch.float64, dev, depth=2)
        for slot in range(2):
            o = dict(per_slot[slot])
            p = G.send[slot]
            o["time"], o["sd"], o["sdd"] = p[0], p[1], p[2]
            if args.gather == "full":
                o["q"] = p[3:].view(B, N, D)          # the engine writes q straight into the payload
            outs.append(o)
    counter = [0]

    def step():
        k = counter[0]
        counter[0] += 1
        # the gather that last read this buffer is done (host-side too when the engine's own stream
        # is about to write q into the payload)
        G.buffer(k, host_sync=(pipelined and distributed and args.gather == "full"))
        E.time_joint_paths(inp, outs[k % 2], N)
        if mode == 2:
            # this stream is now ordered behind the PREVIOUS solve: its gather can go
            if k >= 1 and k - 1 >= first[0]:
                G.launch(k - 1)
        else:
            G.launch(k)

    first = [0]

    def finish():
        """Everything issued so far is complete on this stream and gathered."""
        if mode == 2 and counter[0] > first[0]:
            E.fence()
            G.launch(counter[0] - 1)
        G.drain()
        first[0] = counter[0]

    def timed_region(steps, with_events):
        """K steps bracketed by barrier + synchronize on both sides; returns the elapsed seconds."""
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        for k in range(steps):
            # (the LAST step of every group of `stride`: an event pair around the very first sweep after the
            # synchronize -- the pipeline is empty there -- costs 0.26 ms, twenty times what it costs in flight)
            E.profile_enable(2 if (with_events and k % stride == stride - 1) else False)
            step()
        finish()                          # every batch is solved and its gather has landed on rank 0
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        return time.perf_counter() - t_begin

    timing = not args.no_kernel_timing
    stride = max(1, min(args.timing_stride, max(args.steps, 1)))
    # Cold figure: the same K steps right after the driver's W warm-up steps, BEFORE the clock
    # warm-up below -- what a caller sees who starts solving on an idle GPU (about 10 % slower;
    # reported next to `value` as cold_value / cold_ms_per_step).
    cold_elapsed = None
    if args.clock_warmup_ms > 0 and args.steps > 0:
        for _ in range(args.warmup):
            step()
        finish()
        torch.cuda.synchronize()
        cold_elapsed = timed_region(args.steps, False)
    # The power management needs sustained load before the clocks settle: a 20-step timed region
    # right after a cold start runs 8 % slower than the same 20 steps after half a second of
    # load. Untimed, reported in config.clock_warmup_ms.
    if args.clock_warmup_ms > 0:
        t_end = time.perf_counter() + args.clock_warmup_ms * 1e-3
        while True:
            for _ in range(8):
                step()
            finish()
            torch.cuda.synchronize()
            go = time.perf_counter() < t_end
            if distributed:     # every rank must issue the same number of gathers: stop together
                flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go = bool(flag.item())
            if not go:
                break
    pairs_needed = (args.steps + stride - 1) // stride
    for i in range(args.warmup):
        # The first event pair of a process costs about 0.25 ms once (the runtime switches the
        # queue to timestamped dispatches) and every new pair two event creations: paid here, in
        # the warm-up steps, not in the timed region. profile_reset() below recycles the pairs.
        E.profile_enable(2 if (timing and i < pairs_needed) else False)
        step()
    E.profile_enable(False)
    finish()
    torch.cuda.synchronize()

    E.profile_reset()
    # timed region: events around the dominant kernel only, on every stride-th step
    elapsed = timed_region(args.steps, timing)
    E.profile_enable(False)
    # outside the timed region: the root's rebuilt time samples of its OWN shard against the time
    # output of its own solve (same batch every step), bit for bit
    rebuilt_ok = None
    if compact and distributed and rank == 0 and counter[0] > 0:
        last = (counter[0] - 1) % 2
        rebuilt_ok = bool(torch.equal(root_time[last][:B].view(torch.int64),
                                      outs[last]["time"].view(torch.int64)))
        if not rebuilt_ok:
            a, b_ = root_time[last][:B], outs[last]["time"]
            bad = (a.view(torch.int64) != b_.view(torch.int64))
            print("bench.py: rebuilt time differs in %d of %d samples (%d paths), max abs diff %.3e; "
                  "payload ds[0..2] %s, sd equal to local: %s"
                  % (int(bad.sum()), bad.numel(), int(bad.any(dim=1).sum()),
                     float((a - b_).abs().max()), G.recv[last][0, 2 * B * N:2 * B * N + 3].tolist(),
                     bool(torch.equal(G.recv[last][0, :B * N].view(B, N), outs[last]["sd"]))),
                  file=sys.stderr)
    # every path of the LAST timed step must have been solved (status is rewritten by each step)
    ok = int((outs[(counter[0] - 1) % 2]["status"] == 0).sum()) if counter[0] else 0
    dominant_ms = E.profile_mean_ms(eng.KERNEL_SWEEP)[0] if timing else 0.0
    if timing and rank == 0:
        # the other kernels' durations, outside the timed region (events around every kernel
        # cost a few per cent of a step)
        E.set_pipelining(False)           # one kernel at a time: undisturbed durations
        E.profile_reset()
        E.profile_enable(1)
        for _ in range(5):
            E.time_joint_paths(inp, outs[0], N)
        torch.cuda.synchronize()
        E.profile_enable(False)

    el = torch.tensor([elapsed, cold_elapsed if cold_elapsed is not None else 0.0], dtype=torch.float64, device=dev)
    okt = torch.tensor([ok], dtype=torch.int64, device=dev)
    if distributed:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(okt, op=dist.ReduceOp.SUM)
    elapsed = float(el[0].item())
    if cold_elapsed is not None:
        cold_elapsed = float(el[1].item())
    solved = int(okt.item())

    failed = args.steps > 0 and solved != total_paths
    if rank == 0:
        kernels = E.profile_summary() if timing else {}
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        value = total_paths * args.steps / elapsed
        roofline = roofline_block(kernels, dominant_ms, B, D, N, P, ms_per_step) if kernels else None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(batch, N, D)
        named = [k for k, v in WORKLOADS.items() if v == B]
        if named and D == 7 and N == 2000:
            idx = 1 if named[0] == "configs1" else 2
            what = "BASELINE.json configs[%d]" % idx
            if idx == 2 and world != 8:
                what += "'s per-GPU share (8192 of 65536 paths; the config itself needs 8 GPUs)"
            if idx == 1 and world > 1:
                what += " replicated per GPU (weak scaling; configs[2] is --workload configs2)"
        else:
            what = "custom shape (not a BASELINE.json config)"
        gb = gather_bytes_per_path(args.gather if distributed else "none", D, N)
        line = {
            "metric": baseline_metric(),
            "value": round(value, 1), "unit": "paths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            # the same K steps right after the W warm-up steps on an idle GPU, before the clock warm-up
            "cold_value": (round(total_paths * args.steps / cold_elapsed, 1) if cold_elapsed else None),
            "cold_ms_per_step": (round(cold_elapsed / max(args.steps, 1) * 1e3, 4) if cold_elapsed else None),
            "config": {"workload": "%s: %d random %d-DOF joint-space B-spline paths per GPU, %d "
                                   "s-samples each (10 waypoints, %d control points), inputs "
                                   "resident in HBM" % (what, B, D, N, P),
                       "paths_per_gpu": B, "total_paths": total_paths, "num_dofs": D,
                       "num_samples": N, "solved_paths": solved,
                       "clock_warmup_ms": args.clock_warmup_ms,
                       "dominant_kernel_timed_every": (stride if timing else None),
                       "pipelined": {0: "no: one kernel at a time",
                                     1: "mode 1: the sampling/LP kernel of step k+1 runs under the sweep "
                                        "of step k (two engine workspaces, one engine stream)",
                                     2: "mode 2: as mode 1, and the sweep of step k+1 may start while the "
                                        "slowest paths of step k are still running (sweeps on two engine "
                                        "streams; all K steps complete inside the timed region)"}[mode],
                       "gather": {"mode": args.gather if distributed else "none (single GPU)",
                                  "bytes_per_path": gb,
                                  # what reaches rank 0, against north_star's "t, s, s', q": q is part of
                                  # the payload only with --gather full (link-bound at about 4x on 8 GPUs)
                                  "named_outputs_at_root": (None if not distributed else
                                                            {"minimal": ["sd", "t (rebuilt on the root from sd)",
                                                                         "s (arithmetic sequence from ds)"],
                                                             "compact": ["sd", "sdd", "t (rebuilt on the root from sd)",
                                                                         "s (arithmetic sequence from ds)"],
                                                             "profile": ["t", "sd", "sdd", "s (arithmetic sequence from ds)"],
                                                             "full": ["t", "sd", "sdd", "q", "s (arithmetic sequence from ds)"]}
                                                            [args.gather]),
                                  "north_star_outputs": ["t", "s", "sd", "q"],
                                  "rebuilt_time_equals_local_solve": rebuilt_ok,
                                  "bytes_into_rank0_per_step": gb * B * (world - 1),
                                  "what": (("ONE RCCL gather per step of the packed payload "
                                            "(sd" + (", sdd" if with_sdd else "") + " and the scalars ds, time_start per path; the root "
                                            "rebuilds t from sd with the solver's operations -- "
                                            "tpamd_rebuild_time_device, inside the timed region -- and s = "
                                            "s_start + i*ds is an arithmetic sequence; " + ("" if with_sdd else "sdd, ") + "q, qd, qdd stay on the "
                                            "producing GPU) to rank 0, overlapped with the next step's solve "
                                            "(double-buffered), all inside the timed region") if compact else
                                           ("ONE RCCL gather per step of the packed payload "
                                            "(t, sd, sdd%s; s = s_start + i*ds is rebuilt on the "
                                            "root; qd, qdd%s stay on the producing GPU) to rank 0, "
                                            "overlapped with the next step's solve "
                                            "(double-buffered), all inside the timed region"
                                            % ((", q", "") if args.gather == "full" else ("", ", q"))))
                                  if distributed else None}},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
        if failed:
            print("bench.py: %d of %d paths solved in the last timed step" % (solved, total_paths),
                  file=sys.stderr)
        if rebuilt_ok is False:
            print("bench.py: the root's rebuilt time samples differ from its own solve", file=sys.stderr)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if failed or rebuilt_ok is False:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
