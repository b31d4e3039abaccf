"""`python bench.py --gpus N` run plain (no torch.distributed.run around it) launches its own ranks: bench.self_launch.
Here, without a GPU: the launcher starts N children with the rendezvous variables set, relays their failure ("needs a HIP
device": there is no CPU fallback) as a non-zero exit code, and never prints the old "launch with: ..." refusal.
The GPU form of the same command is tests/test_gpu_bench_contract.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_multi_gpu_command_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", MI3PT_BENCH_ECHO_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "demo"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode not in (0, None)
    assert "launch with" not in r.stderr and "launch with" not in r.stdout
    # both ranks started, each with its own RANK of WORLD_SIZE 2 and the same rendezvous on 127.0.0.1, and each failed loudly
    ranks = sorted(l for l in r.stderr.splitlines() if l.startswith("bench.py rank "))
    assert len(ranks) == 2 and ranks[0].startswith("bench.py rank 0/2 ") and ranks[1].startswith("bench.py rank 1/2 ")
    assert ranks[0].split("rendezvous ")[1] == ranks[1].split("rendezvous ")[1] and "127.0.0.1:" in ranks[0]
    assert r.stderr.count("needs a HIP device") == 2
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launcher_env_of_a_wrapped_run_is_left_alone():
    """Under torch.distributed.run (WORLD_SIZE set) bench.py must not launch anything itself."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", MI3PT_BENCH_ECHO_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "demo"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stderr.count("needs a HIP device") == 1
    assert [l for l in r.stderr.splitlines() if l.startswith("bench.py rank ")] == ["bench.py rank 1/2 local 1 rendezvous 127.0.0.1:29999"]
