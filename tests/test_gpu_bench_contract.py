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
    """The driver's own arguments (--steps 20 --warmup 5; a step = 16 frames, four steps per launch): the headline is
    the ~870k-triangle scene the target is quoted on; the roofline fraction comes from counter passes
    of this very command line and is a fraction; the launch statistics are consistent with the wall clock."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["metric"] == "Mrays/s" and j["unit"] == "Mrays/s" and j["higher_is_better"] is True
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 20, 5) and j["vs_baseline"] is None
    assert j["scaling"] == "strong" and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["image"] == [1920, 1080]
    assert j["config"]["triangles"] > 800_000 and "dragon-class" in j["config"]["workload"] and j["config"]["frames_per_step"] == 16
    assert j["value"] > 1000 and abs(j["value"] - j["config"]["rays_per_step"] / j["ms_per_step"] / 1e3) / j["value"] < 0.01
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["kernel_ms"] > 0 and roof["launches_timed"] == 5 and roof["frames_per_launch"] == 64.0
    # the non-overlapped kernel time per frame cannot exceed the wall clock per frame
    assert roof["kernel_ms_exclusive"] / 4 <= j["ms_per_step"] * 1.02          # (one launch per four steps)
    assert roof["kernel_ms_exclusive"] <= roof["kernel_ms"] * 1.02
    # the average over every launch of the process (what `rocprofv3 --stats` averages): the warm-up's 80 frames are a
    # 64- and a 16-frame launch
    assert roof["launches_all"] == 7 and 0 < roof["kernel_ms_all_launches"] < roof["kernel_ms"]
    # measured by this run (rocprofv3 is on the box): HBM-side traffic, a real fraction, the issue figures
    assert roof["traffic"] is not None and roof["traffic"] > 0, roof.get("pmc_log")
    assert 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert 0 < roof["valu_issue_frac"] < 1 and 0 < roof["lane_utilisation"] <= 1
    assert roof["algorithmic_GBps"] > 0 and roof["box_tests_per_ray"] > 1
    assert j["also"]["demo"]["value"] > 1000
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "Mrays/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]


@pytest.mark.parametrize("scaling,image", [("strong", [1920, 1080]), ("weak", [1920, 2160])])
def test_two_rank_rehearsal_line(scaling, image):
    env = dict(os.environ, MI3PT_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--scaling", scaling, "--workload", "demo"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert (j["n_gpus"], j["steps"], j["scaling"]) == (2, 4, scaling) and j["config"]["image"] == image and "cpu_baseline" not in j
    assert "tile-split x2" in j["config"]["parallelism"] and j["value"] > 100
    assert j["config"]["frames_per_launch"] == 64.0         # 4 steps = 64 frames: one launch (a rank of a 2-way split batches up to 128)


def test_strong_and_weak_scaling_agree_on_one_gpu():
    vals = {}
    for scaling in ("strong", "weak"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--scaling", scaling,
                            "--workload", "demo", "--no-pmc", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        j = _line(r.stdout)
        assert j["config"]["image"] == [1920, 1080] and j["scaling"] == scaling
        vals[scaling] = (j["value"], j["config"]["rays_per_step"])
    assert vals["strong"][1] == vals["weak"][1]                              # the same job
    assert abs(vals["strong"][0] - vals["weak"][0]) / vals["weak"][0] < 0.1  # the same speed (run-to-run noise)
