"""`python bench.py --gpus 2` exactly as the driver starts it (no launcher around it): the script spawns its two ranks
itself and rank 0's line says n_gpus 2.  The box has one GPU, so BIOSCAN_BENCH_REHEARSE=1 puts both ranks on cuda:0 and
the bench's own barrier / MAX / SUM on gloo -- control flow of the N > 1 path through bench.py, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BIOSCAN_BENCH_REHEARSE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--blocks", "8192", "--steps", "2",
                          "--warmup", "1"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bare_gpus_2_runs_config_5_shape():
    r = run_bench([])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["scaling"] == "weak" and r["config"]["mode"] == "indexed"
    assert r["config"]["file_blocks"] == 2 * 8192 and r["config"]["plan_partitions"] == 16
    assert r["config"]["workload"].startswith("config 5")
    # the ranks' runs of the one plan returned every record of the file exactly once (bench.py exits non-zero otherwise)
    assert r["value"] > 0 and r["config"]["file_records"] > 0
    # each rank holds only its part of the file
    assert r["config"]["rank0_resident_bytes"] < 0.75 * r["config"]["file_compressed_bytes"]


def test_bare_gpus_2_independent_shards():
    r = run_bench(["--mode", "shards"])
    assert r["n_gpus"] == 2 and r["config"]["mode"] == "shards"
    assert r["config"]["n_blocks_per_gpu"] == 8192
    assert r["config"]["workload"].startswith("independent shards")
