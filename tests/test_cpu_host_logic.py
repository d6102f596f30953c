"""CPU: the C-ABI library loads and exports every declared symbol; the C++ host planning
(balance_partitions, estimate_sizes_from_bai) matches the oracle and the reference's own
partition_balancer tests (bio-format-core/src/partition_balancer.rs:321-1005); sharding logic."""
import os
import random
import re

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_declared_symbols(pkg):
    lib = pkg.load_library()
    hdr = open(os.path.join(ROOT, "include", "bioscan.h")).read()
    declared = set(re.findall(r"\b(bioscan_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_no_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.BioscanError, match="no HIP device"):
        pkg.BamTableProvider(os.path.join(G, "multi_chrom.bam"))


def _desc(parts):
    out = []
    for p in parts:
        rs = []
        for r in p.regions:
            rs.append(f"{r.chrom}:{r.start if r.start is not None else ''}-{r.end if r.end is not None else ''}{'*' if r.unmapped_tail else ''}")
        out.append(f"{p.total_estimated_bytes}|" + ";".join(rs))
    return "\n".join(out) + ("\n" if out else "")


def _py(oracle, ests, target):
    E = [oracle.RegionSizeEstimate(oracle.GenomicRegion(e["chrom"], e.get("start"), e.get("end")), e["bytes"],
                                   e.get("contig_len"), e.get("unmapped") or 0, list(e.get("bins") or []), e.get("leaf_span") or 0)
         for e in ests]
    return oracle.balance_partitions(E, target)


def _est(chrom, b, cl=None, unmapped=0, bins=None, leaf=0):
    return {"chrom": chrom, "bytes": b, "contig_len": cl, "unmapped": unmapped, "bins": bins, "leaf_span": leaf}


HUMAN = [("chr1", 249), ("chr2", 243), ("chr3", 198), ("chr4", 191), ("chr5", 181), ("chr6", 171), ("chr7", 159), ("chr8", 146),
         ("chr9", 141), ("chr10", 136), ("chr11", 135), ("chr12", 134), ("chr13", 115), ("chr14", 107), ("chr15", 102),
         ("chr16", 90), ("chr17", 84), ("chr18", 80), ("chr19", 59), ("chr20", 64), ("chr21", 47), ("chr22", 51), ("chrX", 155), ("chrY", 57)]


def test_balancer_reference_kats(pkg, oracle):
    """Properties asserted by the reference's own tests, checked on BOTH implementations."""
    def both(ests, target):
        py = _py(oracle, ests, target)
        assert pkg.debug_balance_partitions(ests, target) == _desc(py)
        return py
    assert both([], 4) == []
    r = both([_est("chr1", 100), _est("chr2", 50), _est("chrX", 30)], 1)
    assert len(r) == 1 and len(r[0].regions) == 3 and r[0].total_estimated_bytes == 180
    r = both([_est(f"chr{i}", 100) for i in range(1, 5)], 2)
    assert [p.total_estimated_bytes for p in r] == [200, 200]
    r = both([_est("chr1", 100), _est("chr2", 50), _est("chr3", 10)], 2)
    assert [p.total_estimated_bytes for p in r] == [100, 60]
    r = both([_est("chr1", 100, 249_000_000), _est("chr2", 50), _est("chr3", 10)], 2)
    assert [p.total_estimated_bytes for p in r] == [80, 80]
    r = both([_est("chr1", 200, 249_000_000), _est("chr2", 10)], 4)
    assert sum(len(p.regions) for p in r) > 2 and len(r) <= 4
    r = both([_est(f"chr{i}", 0) for i in range(1, 5)], 2)
    assert [len(p.regions) for p in r] == [2, 2]
    assert len(both([_est("chr1", 100), _est("chr2", 50)], 8)) == 2
    r = both([_est("chr1", 100, 249_000_000), _est("chr2", 50, 243_000_000)], 8)
    assert 2 < len(r) <= 8
    for target in (2, 4, 8):
        r = both([_est("chr1", 1000, 249_000_000)], target)
        assert len(r) == target
        regs = [x for p in r for x in p.regions]
        assert all(x.chrom == "chr1" and x.start is not None for x in regs)
        assert all(x.end is not None for x in regs[:-1]) and regs[-1].end is None
    ests = [_est(c, b, b * 1_000_000) for c, b in HUMAN]
    r = both(ests, 8)
    tot = [p.total_estimated_bytes for p in r]
    assert sum(tot) == sum(b for _, b in HUMAN) and max(tot) <= 2 * min(tot) and 0 < len(r) <= 8
    r = both([_est("chr1", 200, 249_000_000, 1000), _est("chr2", 10)], 4)
    tails = [x for p in r for x in p.regions if x.unmapped_tail]
    assert len(tails) == 1 and tails[0].chrom == "chr1" and tails[0].start is None and tails[0].end is None
    r = both([_est("chr1", 100, 249_000_000, 500), _est("chr2", 95, 243_000_000, 300), _est("chrM", 5, 16_569, 100)], 4)
    assert len([x for p in r for x in p.regions if x.chrom == "chrM" and x.unmapped_tail]) == 1
    r = both([_est("chr1", 249, 249_000_000), _est("chr2", 243, 243_000_000), _est("chr3", 198, 198_000_000), _est("chrX", 60, 155_000_000)], 4)
    assert len(r) == 4 and all(187 <= p.total_estimated_bytes <= 189 for p in r)
    r = both([_est("chr1", 0), _est("chr2", 100), _est("chrX", 0)], 4)
    assert sorted(x.chrom for p in r for x in p.regions) == ["chr1", "chr2", "chrX"]
    for target in (2, 3, 4, 8, 16):
        r = both([_est("chr1", 249, 249_000_000), _est("chr2", 243, 243_000_000), _est("chr3", 198, 198_000_000)], target)
        assert sum(p.total_estimated_bytes for p in r) == 690
    # bin-aware split concentrates on data (partition_balancer.rs:747-800)
    bins = [i * 16384 + 1 for i in range(100)]
    r = both([_est("chr1", 1000, 249_000_000, 0, bins, 16384)], 4)
    ends = [x.end for p in r for x in p.regions if x.end is not None]
    assert len(r) == 4 and all(e < 25_000_000 for e in ends)


def test_balancer_differential_fuzz(pkg, oracle):
    rng = random.Random(1234)
    for _ in range(300):
        n = rng.randint(1, 12)
        ests = []
        for i in range(n):
            cl = rng.choice([None, rng.randint(1, 300_000_000)])
            nb = rng.choice([0, 0, rng.randint(1, 200)])
            hi = max((cl or 1_000_000) // 16384, 1)
            bins = sorted({rng.randrange(0, hi) * 16384 + 1 for _ in range(nb)})
            ests.append(_est(f"c{i}", rng.choice([0, 1, rng.randint(0, 10_000), rng.randint(0, 1 << 40)]), cl,
                             rng.choice([0, 0, 7]), bins, 16384 if bins else 0))
            if rng.random() < 0.15:
                a = rng.randint(1, 1_000_000)
                ests[-1]["start"], ests[-1]["end"] = a, a + rng.randint(0, 5_000_000)
        target = rng.randint(1, 40)
        assert pkg.debug_balance_partitions(ests, target) == _desc(_py(oracle, ests, target)), (ests, target)


@pytest.mark.parametrize("fname", ["multi_chrom.bam", "multi_chrom_large.bam", "bam_with_tags.bam", "no_coor_only.bam",
                                   "nanopore_custom_tags.bam", "10x_pbmc_tags.bam"])
def test_full_scan_plan_matches_oracle(pkg, oracle, fname):
    """C++ parse_bai + estimate_sizes_from_bai + balance_partitions == oracle on the reference's BAI fixtures."""
    b = oracle.BamOracle(os.path.join(G, fname))
    for target in (1, 2, 3, 4, 8, 16, 64):
        parts, _ = b.scan(target_partitions=target)
        got = pkg.debug_plan_full_scan(os.path.join(G, fname + ".bai"), b.hdr.ref_names, b.hdr.ref_lengths, target)
        assert got == _desc(parts), (fname, target)


def test_shard_partitions_in_order(pkg):
    f = pkg.shard_partitions_in_order
    assert f([], 4) == [[], [], [], []]
    assert f([10, 10, 10, 10], 2) == [[0, 1], [2, 3]]
    assert f([100, 1, 1, 1], 2) == [[0], [1, 2, 3]]
    assert f([5, 5], 4) == [[0], [1], [], []]
    rng = random.Random(5)
    for _ in range(200):
        w = [rng.randint(0, 1000) for _ in range(rng.randint(1, 60))]
        world = rng.randint(1, 8)
        runs = f(w, world)
        assert len(runs) == world
        assert [i for r in runs for i in r] == list(range(len(w)))   # contiguous, ordered, complete
        assert sum(1 for r in runs if r) == min(world, len(w))


def test_f32_display_header_matches_oracle_and_numpy(oracle, tmp_path):
    """csrc/f32_display.h (Rust f32::to_string on device) compiled for the host: known answers, the oracle's exact
    rational search, and numpy's shortest printer (which differs only on exact ties, by one unit in the last place)."""
    import ctypes
    import subprocess
    import numpy as np
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = tmp_path / "t.cpp"
    src.write_text('#include "f32_display.h"\n'
                   'extern "C" void many(const uint32_t* b, uint64_t n, uint8_t* out, uint32_t* len) {\n'
                   '  for (uint64_t i = 0; i < n; i++) len[i] = f32disp::f32_display(b[i], out + i * 56); }\n')
    so = str(tmp_path / "t.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(root, "datafusion-bio-formats_amd", "csrc"), str(src), "-o", so])
    lib = ctypes.CDLL(so)

    def disp(bits):
        n = len(bits)
        out = np.zeros(n * 56, np.uint8)
        ln = np.zeros(n, np.uint32)
        lib.many(bits.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), out.ctypes.data_as(ctypes.c_void_p), ln.ctypes.data_as(ctypes.c_void_p))
        return [bytes(out[i * 56:i * 56 + ln[i]]).decode() for i in range(n)]

    kats = [(1.0, "1"), (0.1, "0.1"), (1e-7, "0.0000001"), (3.4028235e38, "340282350000000000000000000000000000000"),
            (1.1754944e-38, "0.000000000000000000000000000000000000011754944"), (1e-45, "0.000000000000000000000000000000000000000000001"),
            (16777216.0, "16777216"), (0.3, "0.3"), (1.5, "1.5"), (-0.0, "-0"), (0.0, "0"), (8999999488.0, "9000000000"),
            (9000000512.0, "9000001000"), (-2.5, "-2.5"), (1e20, "100000000000000000000"), (343126.125, "343126.13"),
            (float("nan"), "NaN"), (float("inf"), "inf"), (float("-inf"), "-inf")]
    got = disp(np.array([k for k, _ in kats], np.float32).view(np.uint32))
    for (k, w), g in zip(kats, got):
        assert g == w and oracle._f32_to_string(k) == w, (k, g, w)
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    extra = [(e << 23) | f for e in range(255) for f in (0, 1, 0x7FFFFF, 0x400000)]
    bits = np.concatenate([bits, np.array(extra, np.uint32)])
    ties = 0
    for g, f in zip(disp(bits), bits.view(np.float32)):
        assert g == oracle._f32_to_string(float(f)), (float(f), g)
        if np.isfinite(f):
            w = np.format_float_positional(f, unique=True, trim="-")
            if w != g:
                ties += 1
                assert len(w) == len(g) and int(g.replace(".", "").replace("-", "")) - int(w.replace(".", "").replace("-", "")) == 1, (g, w)
    assert ties < 100


# ---- partitions -> devices: partition_byte_ranges_in_order (bio-format-core/src/range_planning.rs:147-195) -----------
# The reference's own tests (range_planning.rs:328-374), transcribed as data, run against the Python restatement
# (sharding.py, used by bench.py) AND the C++ planner behind bioscan_scan_devices.
def _runs_from_c(pkg, weights, world):
    runs, run_of = pkg.debug_shard_partitions(weights, world)
    out = [[] for _ in range(runs)]
    for i, r in enumerate(run_of):
        out[r].append(i)
    return out


def test_in_order_partitions_are_contiguous_and_cover_every_range(pkg):
    # range_planning.rs:328-357: 17 ranges of lengths 10 + 5 i, targets 1..20
    weights = [10 + 5 * i for i in range(17)]
    for target in range(1, 21):
        for runs in (pkg.shard_partitions_in_order(weights, target)[:min(target, 17)], _runs_from_c(pkg, weights, target)):
            assert runs and len(runs) <= min(max(target, 1), len(weights))
            assert all(r for r in runs), (target, runs)                       # no empty partition
            assert [i for r in runs for i in r] == list(range(17)), target    # source order kept, every range once


def test_in_order_partitions_split_even_work_evenly(pkg):
    # range_planning.rs:359-370: 8 equal ranges over 4 partitions -> 2 each
    for runs in (pkg.shard_partitions_in_order([10] * 8, 4), _runs_from_c(pkg, [10] * 8, 4)):
        assert len(runs) == 4 and all(len(r) == 2 for r in runs)


def test_in_order_partitions_handle_an_empty_input(pkg):
    # range_planning.rs:372-374
    assert pkg.shard_partitions_in_order([], 4) == [[], [], [], []]
    assert pkg.debug_shard_partitions([], 4)[0] == 0


def test_cpp_sharding_equals_python_restatement(pkg):
    rng = random.Random(11)
    for _ in range(300):
        n = rng.randrange(1, 40)
        weights = [rng.choice([0, 1, 7, 1000, rng.randrange(1, 1 << 40)]) for _ in range(n)]
        world = rng.randrange(1, 12)
        want = [r for r in pkg.shard_partitions_in_order(weights, world) if r]
        assert _runs_from_c(pkg, weights, world) == want, (weights, world)


def test_shard_partitions_balanced_is_the_contiguous_optimum(pkg):
    """bench.py deals the partitions of a plan to its ranks with this rule: contiguous runs in plan order (so the ranks' outputs
    concatenated are the single-GPU order, as with the reference's in-order rule) whose heaviest run is as light as any contiguous
    split allows -- checked against every split of small random plans; on the bench's own shape (16 equal partitions + the empty
    no-coor one) two ranks get 8 real partitions each, eight ranks of 128 + 1 get 16 each."""
    import itertools
    import random
    f = pkg.shard_partitions_balanced
    assert [len(r) for r in f([52] * 16 + [0], 2)] == [8, 9]
    assert [len(r) for r in f([52] * 128 + [0], 8)] == [16] * 7 + [17]
    assert f([], 3) == [[], [], []]
    assert f([5, 5], 4) == [[0], [1], [], []]
    rng = random.Random(1)
    for _ in range(300):
        n, k = rng.randint(1, 9), rng.randint(1, 6)
        ws = [rng.choice([0, 1, 5, 7, 50, rng.randint(0, 100)]) for _ in range(n)]
        runs = f(ws, k)
        assert len(runs) == k and [i for r in runs for i in r] == list(range(n))
        filled = [r for r in runs if r]
        assert len(filled) == min(k, n)
        best = min(max(sum(ws[a:b]) for a, b in zip((0,) + cuts, cuts + (n,)))
                   for cuts in itertools.combinations(range(1, n), min(k, n) - 1))
        assert max(sum(ws[i] for i in r) for r in filled) == best, (ws, k, runs)
