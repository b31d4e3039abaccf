"""bench.py end to end: the one JSON line the driver reads (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline
+ cpu_baseline), for N = 1 and -- as a rehearsal of the multi-GPU path on one device, two ranks
over gloo -- for N = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "4"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["metric"] == "Mrays/s" and j["unit"] == "Mrays/s" and j["higher_is_better"] is True
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 20, 4) and j["vs_baseline"] is None
    assert j["scaling"] == "weak" and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["image"] == [1920, 1080]
    assert j["value"] > 1000 and abs(j["value"] - j["config"]["rays_per_step"] / j["ms_per_step"] / 1e3) / j["value"] < 0.01
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["kernel_ms"] > 0 and roof["launches_timed"] == 2
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "Mrays/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]


def test_two_rank_rehearsal_line():
    env = dict(os.environ, MI3PT_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert (j["n_gpus"], j["steps"]) == (2, 8) and j["config"]["image"] == [1920, 2160] and "cpu_baseline" not in j
    assert "tile-split x2" in j["config"]["parallelism"] and j["value"] > 100
