"""The whole several-GPU flow of bench.py from the driver's plain command, rehearsed on the one-GPU box:
`python bench.py --gpus 2` starts its two ranks itself; with CGPS_BENCH_REHEARSAL_GLOO=1 they share the GPU and
talk over gloo (results exact, timing meaningless).  RCCL itself needs a box with several GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("sub_shards", [0, 2])
def test_plain_command_runs_two_ranks_and_prints_one_complete_line(sub_shards):
    env = dict(os.environ, CGPS_BENCH_REHEARSAL_GLOO="1", CGPS_BENCH_PREWARM_STEPS="10")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", str(1 << 22), "--steps", "5", "--warmup", "2"]
    if sub_shards:
        cmd += ["--sub-shards", str(sub_shards)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 5
    assert d["config"]["rows_total"] == 1 << 22 and d["config"]["rows_per_gpu"] == 1 << 21
    assert d["extras"]["speedup_vs_single_gpu"] > 0
    assert d["extras"]["records_per_rank"] == max(1, sub_shards)
    assert d["check"]["logdet_rel_err"] < 1e-10 and d["check"]["mahal_rel_err"] < 1e-9
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0
    assert "opB_decompose_plus_solve" in d["cpu_baseline"]
    assert d["roofline"]["kernel_min_us"] <= d["roofline"]["kernel_avg_us"]
