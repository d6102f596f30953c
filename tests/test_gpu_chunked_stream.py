"""The chunk pipeline of a BAM stream (bounded memory, csrc/engine.cpp BamExecState): a partition is inflated, framed
and extracted `chunk_members` BGZF members at a time, the record that straddles a chunk end is carried over, and a
RecordBatch that straddles two chunks is stitched on the host.  None of that may change a single batch: every chunk
size, down to one member per chunk, must reproduce the oracle's batches exactly -- the same parity bar as the
whole-partition tests (batches of exactly batch_size rows, bam/src/physical_exec.rs:545-565)."""
import os
import random

import pytest

import bam_build as bb
from test_gpu_bam_parity import _cmp_batches
from test_gpu_bam_edge_cases import REFS, _records

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("chunk", [1, 2, 3, 7, 1000])
@pytest.mark.parametrize("fname,tags", [("multi_chrom_large.bam", None), ("nanopore_custom_tags.bam", ["NM", "AS", "pa", "de", "tp"]),
                                        ("10x_pbmc_tags.bam", ["CB", "CR", "NH", "RG"])])
def test_sequential_scan_any_chunk_size(pkg, oracle, fname, tags, chunk):
    path = os.path.join(G, fname)
    prov = pkg.BamTableProvider(path, None, True, tags, index_path="", chunk_members=chunk)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags, index_path=None)
    for bs in (8192, 100, 7):
        got = list(prov.scan().execute(0, bs))
        _, want = orc.execute_sequential(None, bs)
        _cmp_batches(got, want, (fname, chunk, bs))
    got = list(prov.scan(projection=[]).execute(0, 33))
    _, want = orc.execute_sequential([], 33)
    assert [b.num_rows for b in got] == [b.num_rows for b in want]


@pytest.mark.parametrize("chunk", [1, 2, 5])
@pytest.mark.parametrize("target", [1, 3, 8])
def test_indexed_partitions_any_chunk_size(pkg, oracle, chunk, target):
    path = os.path.join(G, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path, chunk_members=chunk)
    orc = oracle.BamOracle(path)
    for filters in ([], [("chrom", "in", ["chr2", "chrX"]), ("mapping_quality", ">=", 30)]):
        plan = prov.scan(filters=filters, target_partitions=target)
        parts, residual = orc.scan(filters=filters, target_partitions=target)
        assert plan.num_partitions() == len(parts)
        for p in range(plan.num_partitions()):
            got = list(plan.execute(p, 64))
            _, want = orc.execute_partition(parts[p].regions, None, residual, 64)
            _cmp_batches(got, want, ("indexed", chunk, target, p, filters))


@pytest.mark.parametrize("fname,tags", [("multi_chrom.bam", None), ("no_coor_only.bam", ["CB", "CR"]), ("bam_with_tags.bam", ["NM", "MD", "RG"])])
def test_unmapped_tails_and_no_coor_any_chunk_size(pkg, oracle, fname, tags):
    path = os.path.join(G, fname)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags)
    for chunk in (1, 2):
        prov = pkg.BamTableProvider(path, None, True, tags, chunk_members=chunk)
        for target in (1, 2, 4):
            plan = prov.scan(target_partitions=target)
            parts, residual = orc.scan(target_partitions=target)
            assert plan.num_partitions() == len(parts)
            for p in range(plan.num_partitions()):
                got = list(plan.execute(p, 10))
                _, want = orc.execute_partition(parts[p].regions, None, residual, 10)
                _cmp_batches(got, want, (fname, chunk, target, p))


@pytest.mark.parametrize("member,chunk", [(97, 1), (97, 5), (97, 64), (4096, 1), (4096, 3)])
def test_records_spanning_many_chunks(pkg, oracle, tmp_path, member, chunk):
    """97-byte members: a 5000-base record spans dozens of members, so with one member per chunk it is carried across
    dozens of chunks before it is complete; headers and block_size fields are cut by chunk ends."""
    rng = random.Random(member * 31 + chunk)
    recs = _records(rng, 60)
    path = str(tmp_path / "edge.bam")
    open(path, "wb").write(bb.bam(REFS, recs, member=member))
    tags = ["NM", "MD", "XA", "XB"]
    prov = pkg.BamTableProvider(path, None, True, tags, index_path="", chunk_members=chunk)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags, index_path=None)
    for bs in (8192, 7):
        got = list(prov.scan().execute(0, bs))
        _, want = orc.execute_sequential(None, bs)
        _cmp_batches(got, want, ("span", member, chunk, bs))


def test_truncated_last_record_is_an_error_in_every_chunking(pkg, tmp_path):
    rng = random.Random(5)
    payload_recs = _records(rng, 30)
    good = bb.bam(REFS, payload_recs, member=4096)
    # drop the last data member (keep the EOF marker): the record stream now ends inside a record
    eof = good[-28:]
    body = good[:-28]
    # find the start of the last member by re-framing
    o, starts = 0, []
    while o < len(body):
        starts.append(o)
        o += (body[o + 16] | (body[o + 17] << 8)) + 1
    cut = body[:starts[-1]] + eof
    path = str(tmp_path / "cut.bam")
    open(path, "wb").write(cut)
    for chunk in (1, 4, 0):
        prov = pkg.BamTableProvider(path, None, True, None, index_path="", chunk_members=chunk)
        with pytest.raises(pkg.BioscanError, match="record"):
            list(prov.scan().execute(0, 8192))
