"""GPU parity at medium scale: synthetic BGZF-BAM (tools/synth_bam) scanned by the HIP path vs the
C oracle, all 12 core columns + 4 tag columns, bit-exact; plus size-independent properties."""
import json
import os
import subprocess

import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    exe = os.path.join(ROOT, "tools", "_build", "synth_bam")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    d = tmp_path_factory.mktemp("synth")
    path = str(d / "s2048.bam")
    meta = json.loads(subprocess.check_output([exe, path, "2048", "7", "8"]).decode())
    return path, meta


def _concat(batches, name):
    return pa.chunked_array([b.column(name) for b in batches]).combine_chunks()


def test_sequential_vs_c_oracle(pkg, synth):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    path, meta = synth
    tags, kinds = ["NM", "MD", "AS", "RG"], ["i", "s", "i", "s"]
    prov = pkg.BamTableProvider(path, None, True, tags, index_path="")
    got = list(prov.scan().execute(0, 8192))
    assert sum(b.num_rows for b in got) == meta["n_records"]
    assert all(b.num_rows == 8192 for b in got[:-1])
    st, want = c_oracle.scan(open(path, "rb").read(), True, 4, 0, tags, kinds)
    assert st["n_rows"] == meta["n_records"]
    for name, w in want.items():
        g = _concat(got, name)
        if pa.types.is_large_string(w.type):
            w = w.cast(pa.utf8())
        assert g.equals(w), name


def test_partition_invariants(pkg, synth):
    """count invariance over target_partitions and indexed == sequential as multisets of (name,chrom,start)
    (bam/tests/indexed_read_test.rs:208-237, 297-319)."""
    path, meta = synth
    prov = pkg.BamTableProvider(path)
    seq = pkg.BamTableProvider(path, index_path="")
    base = pa.Table.from_batches(list(seq.scan(projection=[0, 1, 2]).execute(0, 8192)))
    base_rows = sorted(zip(*[base.column(i).to_pylist() for i in range(3)]), key=lambda r: (r[0], str(r[1]), r[2] or -1))
    for target in (1, 3, 8, 16):
        plan = prov.scan(projection=[0, 1, 2], target_partitions=target)
        rows = []
        for p in range(plan.num_partitions()):
            for b in plan.execute(p, 8192):
                rows.extend(zip(*[b.column(i).to_pylist() for i in range(3)]))
        assert len(rows) == meta["n_records"], target
        assert sorted(rows, key=lambda r: (r[0], str(r[1]), r[2] or -1)) == base_rows, target


def test_multi_gpu_sharding_reproduces_single_gpu_order(pkg, synth):
    """contiguous runs of BAI partitions per GPU, no exchange.  Simulated on one GPU: the
    ranks' shards, executed independently (each decodes only the BGZF members its chunks touch) and
    concatenated in rank order, reproduce the single-GPU row order exactly."""
    path, meta = synth
    prov = pkg.BamTableProvider(path)
    plan = prov.scan(projection=[0, 2], target_partitions=16)
    n = plan.num_partitions()
    single = []
    for p in range(n):
        for b in plan.execute(p, 8192):
            single.extend(zip(b.column(0).to_pylist(), b.column(1).to_pylist()))
    assert len(single) == meta["n_records"]
    weights = [plan.partition_estimated_bytes(p) for p in range(n)]
    for world in (2, 4, 8):
        shards = pkg.shard_partitions_in_order(weights, world)
        assert [i for s in shards for i in s] == list(range(n))
        got = []
        for rank in range(world):
            rank_prov = pkg.BamTableProvider(path)          # every rank owns its provider / device buffers
            rank_plan = rank_prov.scan(projection=[0, 2], target_partitions=16)
            for p in shards[rank]:
                for b in rank_plan.execute(p, 8192):
                    got.extend(zip(b.column(0).to_pylist(), b.column(1).to_pylist()))
        assert got == single, world


def test_large_file_properties(pkg):
    """Size-independent properties on a file too large for a value-by-value comparison (650 000 members = BASELINE config 2 when the scratch space allows, else 65 536;
    BIOSCAN_TEST_LARGE_BLOCKS=650000 is BASELINE.json's config 2): every member's CRC32 and ISIZE hold (K2 / K1 gates,
    a failure raises), the record chain ends exactly at the end of the inflated stream, every record the generator wrote
    comes back, a second run gives the same totals, and the BAI plan returns the same number of rows in total."""
    from conftest import full_size_blocks
    blocks = full_size_blocks(650000, 65536)
    exe = os.path.join(ROOT, "tools", "_build", "synth_bam")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    from conftest import scratch_dir
    base = scratch_dir(blocks * 27000)
    path = os.path.join(base, f"bioscan_large_{os.getpid()}.bam")
    try:
        meta = json.loads(subprocess.check_output([exe, path, str(blocks), "11", str(min(16, os.cpu_count() or 1))]).decode())
        prov = pkg.BamTableProvider(path, index_path="")
        prov.make_resident()
        plan = prov.scan(target_partitions=1)
        first = plan.execute_device(0, 8192)
        from conftest import report_size
        report_size("test_large_file_properties[bam]", members=meta["n_blocks"], records=meta["n_records"],
                    compressed_GB=round(meta["compressed_bytes"] / 1e9, 2), inflated_GB=round(meta["inflated_bytes"] / 1e9, 2))
        assert first["n_rows"] == first["n_records"] == meta["n_records"]
        assert first["inflated_bytes"] >= meta["inflated_bytes"]          # + the header member(s)
        assert first["compressed_bytes"] == meta["compressed_bytes"]
        again = plan.execute_device(0, 8192)
        for k in ("n_rows", "n_records", "n_blocks", "inflated_bytes", "arrow_bytes"):
            assert again[k] == first[k], k
        count = prov.scan(projection=[], target_partitions=1).execute_device(0, 8192)
        assert count["n_rows"] == meta["n_records"]
        # the host stream (chunk pipeline, D2H overlapped, batches stitched across chunks): every row, batches of exactly
        # 8192 rows except the last
        drained = plan.execute_drain(0, 8192)
        assert drained["n_rows"] == meta["n_records"]
        assert drained["n_batches"] == (meta["n_records"] + 8191) // 8192
        del prov, plan
        indexed = pkg.BamTableProvider(path)
        iplan = indexed.scan(projection=[0, 2], target_partitions=8)
        assert sum(iplan.execute_device(p, 8192)["n_rows"] for p in range(iplan.num_partitions())) == meta["n_records"]
    finally:
        for p in (path, path + ".bai"):
            try:
                os.unlink(p)
            except OSError:
                pass
