"""What can be checked of bench.py without a GPU: the metric string is BASELINE.json's, the
algorithmic-bytes figure is SURVEY.md 8(d)'s, and the script refuses to run without a GPU."""
import importlib.util
import json
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_metric_and_algorithmic_bytes():
    bench = _bench()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert bench.baseline_metric() == json.load(f)["metric"]
    # inputs 8*(P*D + P+3 + 2D + 4) + outputs t, s, sd, q: 8*N*(3+D); D = 7, N = 2000, P = 28
    assert bench.algorithmic_bytes_per_path(7, 2000, 28) == 8 * (28 * 7 + 31 + 14 + 4) + 8 * 2000 * 10
    assert bench.algorithmic_bytes_per_path(7, 2000, 28) == 161960
    assert bench.HBM_PEAK_GBS == 8000.0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "needs a GPU" in (res.stderr + res.stdout)


def test_gather_payload_bytes_and_counter_file_binding(tmp_path, monkeypatch):
    bench = _bench()
    assert bench.gather_bytes_per_path("none", 7, 2000) == 0
    assert bench.gather_bytes_per_path("minimal", 7, 2000) == 16016       # sd, ds, time_start (default)
    assert bench.gather_bytes_per_path("compact", 7, 2000) == 32016       # sd, sdd, ds, time_start
    assert bench.gather_bytes_per_path("profile", 7, 2000) == 48000       # t, sd, sdd
    assert bench.gather_bytes_per_path("full", 7, 2000) == 160000         # + q
    assert bench.WORKLOADS == {"configs1": 1024, "configs2": 8192}
    # roofline.traffic is only reported from counters measured on these very kernel sources
    h = bench.kernel_source_hash()
    assert len(h) == 64 and h == bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    good = {"tag": "t", "source_sha256": h, "workload": "B1024:D7:N2000",
            "kernels": {"k_sweep": {"hbm_bytes": 5, "avg_us": 400.0, "valu_insts": 10,
                                    "valu_busy_pct": 40.0}}}
    (prof / "counters.json").write_text(json.dumps(good))
    assert bench.recorded_counters(1024, 7, 2000)["kernels"]["k_sweep"]["hbm_bytes"] == 5
    assert bench.recorded_counters(2048, 7, 2000) is None                  # other batch shape
    (prof / "counters.json").write_text(json.dumps(dict(good, source_sha256="0" * 64)))
    assert bench.recorded_counters(1024, 7, 2000) is None                  # stale kernels
    r = bench.roofline_block({"k_sweep": (0.4, 20), "k_sample_lp": (0.2, 5)}, 0.4, 1024, 7, 2000,
                             28, 0.7)
    assert r["traffic"] is None and r["valu"] is None and r["bound"] == "hbm"
    assert abs(r["achieved"] - 161960 * 1024 / 0.4e-3 / 1e9) < 1e-2
    (prof / "counters.json").write_text(json.dumps(good))
    r = bench.roofline_block({"k_sweep": (0.4, 20)}, 0.4, 1024, 7, 2000, 28, 0.7)
    assert r["traffic"] == 5 and r["step_traffic"] == 5 and r["valu"]["busy_pct"] == 40.0
    assert "instruction issue" in r["limiter"]
