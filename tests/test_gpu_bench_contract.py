"""-m gpu: bench.py keeps the driver's contract -- one JSON line on stdout with the agreed keys, the roofline and cpu_baseline
objects, exactly K timed steps -- on a shortened run (small CPU-baseline sample, no frame / style extras)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "3", "--cpu-rays", "4096",
                          "--no-frame", "--no-style"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "Mrays/s" and j["n_gpus"] == 1 and j["steps"] == 24 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 4096 / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["bytes_per_sample"] * r["samples_per_launch"] / r["avg_launch_us"] / 1e3) < 0.02 * r["achieved"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert j["value"] > 100 * c["value"]
    assert c["cpu_model"] and c["one_thread"]["value"] > 0 and c["one_thread"]["value"] <= 1.5 * c["value"]
    assert j["windows"]["median"] == j["ms_per_step"] and j["world_size"] == 1 and j["backend"] is None
    m = j["roofline_more"]                                           # the kernels furthest below their roofline
    for k in ("grid_backward_main_stream", "grid_backward_total"):
        assert m[k]["bound"] == "hbm" and m[k]["peak"] == 8000.0 and 0.02 < m[k]["frac"] < 1.0
        assert abs(m[k]["achieved"] - 588 * m[k]["samples_per_launch"] / m[k]["avg_us"] / 1e3) < 0.02 * m[k]["achieved"]
    assert m["grid_backward_total"]["avg_us"] > m["grid_backward_main_stream"]["avg_us"]
    for k, flop in (("head_forward", 36864), ("head_backward", 73728)):
        assert m[k]["bound"] == "mfma" and m[k]["peak"] == 2500.0 and m[k]["flop_per_sample"] == flop and 0.01 < m[k]["frac"] < 1.0
    assert "frame_encoder" not in m                                  # --no-frame: no inference kernels in this run (the default line has it)
    # round 6: the line says what --warmup asked for beside what ran, where `traffic` comes from, and carries the zero-edit drop-in step
    assert j["warmup_requested"] == 3 and j["warmup"] >= 17
    assert r["traffic"] is None or "not counted in this run" in r["traffic_source"]
    d = j["drop_in_step"]
    assert "error" not in d, d
    assert d["rays"] == 4096 and d["ms_per_step"] > j["ms_per_step"] and 0 < d["device_ms_per_step"] <= 1.05 * d["ms_per_step"]
    assert abs(d["device_ms_per_step"] - sum(d["device_stretches_ms"])) < 0.05 * d["device_ms_per_step"] and len(d["device_stretches_ms"]) == 3
    assert d["bound_by"].startswith("host") or d["bound_by"] == "device"
    assert abs(d["device_time_vs_fused_step"] - d["device_ms_per_step"] / j["ms_per_step"]) < 0.02 * d["device_time_vs_fused_step"]
    nn = d["without_network_nan_check"]
    assert len(nn["device_stretches_ms"]) == 2 and nn["ms_per_step"] <= 1.1 * d["ms_per_step"]
    for k in ("with_one_edit", "with_two_edits"):                    # the optional edits of INTEGRATION 3b, in the order they pay
        assert "error" not in d[k] and 0 < d[k]["device_ms_per_step"] < d["device_ms_per_step"], (k, d[k])
    assert d["with_two_edits"]["device_ms_per_step"] < d["with_one_edit"]["device_ms_per_step"]

