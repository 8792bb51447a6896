"""bench.py prints ONE JSON line with the keys the driver and the judge read (task contract)."""
import json
import os
import subprocess
import sys

import pytest

from tests.util import ROOT, visible_gpus

pytestmark = pytest.mark.gpu


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "3000", "--cpu-seconds", "2", "--e2e-reads", "2000"],
                       capture_output=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "reads/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    # the kernel is VALU-issue bound: that is the roofline the top-level fraction is against; the HBM view
    # north_star asks for sits beside it
    assert rf["bound"] == "valu" and rf["unit"] == "Tlane-op/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0 < rf["frac"] < 1 and abs(rf["achieved"] * 1e12 - rf["cells_per_s"] * rf["ops_per_cell"]) < 1e-3 * rf["achieved"] * 1e12
    hbm = rf["hbm"]
    assert hbm["unit"] == "GB/s" and hbm["peak"] == 8000.0 and abs(hbm["frac"] - hbm["achieved"] / hbm["peak"]) < 1e-5 and hbm["frac"] < rf["frac"]
    assert "traffic_over_algorithmic" in rf and set(d["scaling_definitions"]) == {"weak", "strong"}
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb.get("parity_on_sample") is True
    # the end-to-end leg (raw BLOW5 -> PAF through the command line, files generated in the run): extra keys, never `value`
    e2e = d["end_to_end"]
    assert "error" not in e2e, e2e
    for key in ("uncompressed_K4096", "uncompressed_K512", "compressed_K4096", "compressed_K512"):
        assert e2e[key] > 0, key
    assert e2e["reads"] == 2000 and e2e["unit"] == "reads/s"
    assert d["library_build"] and d["rccl_world_size"] == 1
    assert d["roofline"]["traffic"] is None  # a 3000-read batch was never profiled: no stale counters are quoted


def test_bench_over_rccl_on_every_gpu_the_box_has():
    """`bench.py --gpus G` over RCCL: on a box with G > 1 GPUs with min(G, 4) ranks (it starts them itself), on a one-GPU box as
    a one-rank rehearsal of the same path (SFA_DIST_FORCE=1: broadcast of the reference model, gather of the rows).  The
    line must say how many ranks really existed, and rank 0 must have received exactly the rows every rank computed."""
    g = min(visible_gpus(), 4)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if g <= 1:
        g = 1
        env["SFA_DIST_FORCE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(g), "--steps", "2", "--warmup", "1", "--reads", "20000",
                        "--no-cpu-baseline", "--no-e2e"], capture_output=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == g and d["rccl_world_size"] == g and d["config"]["sharding"] == f"reads x{g}"
    assert d["gather_verified"] is True
    assert d["scaling"] == "weak" and d["value"] > 0
    # strong scaling: one job of --total-reads reads split over the ranks by contiguous ranges (BASELINE.json configs[3] form)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(g), "--steps", "2", "--warmup", "1", "--total-reads", "30001",
                        "--no-cpu-baseline", "--no-e2e"], capture_output=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["total_reads"] == 30001 and sum(d["config"]["reads_per_rank"]) == 30001
    assert d["gather_verified"] is True and abs(d["value"] - 30001 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-3 * d["value"]
