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
