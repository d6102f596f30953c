"""The N > 1 bench compresses ONE shared file with all its ranks (bench.py: `synth_bam ... part K W` then `place K W`): the
result has to be the file one process writes -- same bytes, same BAI -- or the N = 8 point of the scaling curve would scan a
different input from the N = 1 point."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYNTH = os.path.join(ROOT, "tools", "_build", "synth_bam")


@pytest.fixture(scope="module")
def synth():
    if not os.path.exists(SYNTH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)
    return SYNTH


@pytest.mark.parametrize("world", [2, 3, 7])
def test_parts_written_by_ranks_are_the_one_process_file(synth, tmp_path, world):
    n = 1500
    one = str(tmp_path / "one.bam")
    meta_one = json.loads(subprocess.check_output([synth, one, str(n), "42", "2", "6"]).decode())
    shared = str(tmp_path / "shared.bam")
    common = [synth, shared, str(n), "42", "1", "6"]
    for k in range(world):                       # (the ranks do this concurrently; the order must not matter)
        subprocess.check_output(common + ["part", str(world - 1 - k), str(world)])
    metas = [subprocess.check_output(common + ["place", str(k), str(world)]).decode() for k in range(world)]
    meta = json.loads(metas[0])
    assert open(one, "rb").read() == open(shared, "rb").read()
    assert open(one + ".bai", "rb").read() == open(shared + ".bai", "rb").read()
    for key in ("n_blocks", "n_records", "inflated_bytes"):
        if key in meta_one:
            assert meta[key] == meta_one[key], key
