"""bench.py's own launcher (no GPU): `python bench.py --gpus N` started bare must start N ranks as child processes before
torch / HIP is touched in the parent, relay their status, and refuse a launcher whose WORLD_SIZE disagrees with --gpus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, os, sys, subprocess
sys.argv = ["bench.py"] + json.loads(os.environ["PROBE_ARGV"])
calls = []
def fake_run(cmd, **kw):
    calls.append({"cmd": cmd, "env_ipc": kw.get("env", {}).get("HSA_ENABLE_IPC_MODE_LEGACY"), "torch_loaded": "torch" in sys.modules})
    class R: returncode = 7
    return R()
subprocess.run = fake_run
sys.path.insert(0, os.environ["PROBE_ROOT"])
import bench
try:
    bench.main()
    code = 0
except SystemExit as e:
    code = e.code
print(json.dumps({"calls": calls, "code": code, "torch_loaded_after": "torch" in sys.modules}))
"""


def run_probe(argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"PROBE_ARGV": json.dumps(argv), "PROBE_ROOT": ROOT})
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, "-c", PROBE], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bare_gpus_n_spawns_n_ranks_before_torch():
    r = run_probe(["--gpus", "4", "--steps", "2", "--warmup", "1", "--blocks", "4096"])
    assert len(r["calls"]) == 1
    c = r["calls"][0]
    cmd = c["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1", "--blocks", "4096"]  # the ranks get the same flags
    assert c["env_ipc"] == "0"
    assert not c["torch_loaded"] and not r["torch_loaded_after"]   # the parent never imports torch, let alone HIP
    assert r["code"] == 7                                           # the children's status is the bench's status


def test_launcher_world_size_must_match_gpus():
    r = run_probe(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r["calls"] == []
    assert "WORLD_SIZE=3" in str(r["code"])
    assert not r["torch_loaded_after"]
