"""bench.py as its own launcher (`python bench.py --gpus N` without torch.distributed.run) and the configs[3] job
mode (`--lines`), BASELINE.json configs[3].  The GPU cases rehearse two ranks on the one GPU of the test box
(BENCH_FORCE_DEVICE=0); the CPU case checks the launcher's failure path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env=None, timeout=420):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out          # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


def test_launcher_fails_when_a_rank_fails():
    """No GPU here: both ranks stop with 'needs an MI355X'; the launcher must pass the failure on, not hang."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only case")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=300)
    assert p.returncode != 0
    assert "MI355X" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_gpus2_without_launcher_reports_two_ranks():
    p = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--lanes", "2", "--no-cpu-baseline"], {"BENCH_FORCE_DEVICE": "0"})
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "weak"
    assert j["config"]["lines_per_step"] == 512
    assert j["value"] > 0


@pytest.mark.gpu
def test_lines_job_two_ranks_equals_one_rank():
    """configs[3] at a small size: the job's gathered strings (digest in line order) must not depend on how many
    ranks shared it; 4096 + 100 lines = 16 full buckets and a ragged last one."""
    n = 4196
    one = _run(["--lines", str(n), "--lanes", "2"])
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = _json_line(one.stdout)
    two = _run(["--lines", str(n), "--gpus", "2", "--lanes", "2"], {"BENCH_FORCE_DEVICE": "0"})
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = _json_line(two.stdout)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2
    assert j1["config"]["lines"] == n and j2["config"]["lines"] == n
    assert sum(j2["lines_per_rank"]) == n and min(j2["lines_per_rank"]) > n // 4
    assert j1["decoded_chars"] == j2["decoded_chars"] > n          # noise lines decode to tens of characters each
    assert j1["strings_sha256"] == j2["strings_sha256"]
