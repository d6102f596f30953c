"""The rule behind the chunk cut of indexed scans (engine.cpp: build_work_uncached, vcf_engine.cpp), checked on the CPU against
the records themselves: for a region that ends at `end`, let V be the smallest chunk begin of the first non-empty LEAF bin
behind the 16 kb window that holds `end`.  Then every record at a virtual offset >= V starts behind `end` (the file is sorted
by start, and a record of that leaf bin starts inside its window), so no chunk that begins at or behind V can hold a row of the
region's answer -- the reference seeks to such chunks, reads them and filters every record out.  Random coordinate-sorted BAMs
and indexes from tools/fuzz_bam_indexed.py (crowded and empty bins, long reference spans, indexes that do not list every
read), random regions."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_records_behind_the_next_leaf_bin_start_behind_the_region():
    import bam_oracle as O
    import fuzz_bam_indexed as F
    rng = random.Random(2024)
    n_regions = n_cut = n_pruned_records = 0
    for _ in range(120):
        _, bai_bytes, refs, recs = F.make_bam(rng, want_records=True)
        bai = O.parse_bai(bai_bytes)
        for _ in range(12):
            ref = rng.randrange(len(refs))
            length = refs[ref][1]
            if rng.random() < 0.5 and any(r[0] == ref for r in recs):
                anchor = rng.choice([r for r in recs if r[0] == ref])[1]
                start1 = max(1, anchor + 1 - rng.choice([0, 10, 5000, 100000]))
            else:
                start1 = rng.randrange(1, length + 1)
            end1 = min(1 << 29, start1 + rng.choice([0, 100, 16383, 16384, 70000, 5_000_000]))
            chunks = O.bai_query_chunks(bai, ref, start1, end1)
            n_regions += 1
            w_end = (end1 - 1) >> 14
            later = sorted(b for b in bai.refs[ref].bins if 4681 + w_end < b < 37449)
            if not later or 4681 + w_end >= 37448:
                continue
            V = min(c[0] for c in bai.refs[ref].bins[later[0]])
            n_cut += 1
            # every record of this reference at or behind V starts behind the region's end ...
            for refid, pos0, span, flag, voff in recs:
                if refid == ref and voff >= V:
                    assert pos0 + 1 > end1, (refs, ref, start1, end1, pos0, hex(voff), hex(V))
                    n_pruned_records += 1
            # ... so the rows of the region all lie in chunks that begin in front of V (clipped at V)
            kept = [(a, min(b, V)) for a, b in chunks if a < V]
            for refid, pos0, span, flag, voff in recs:
                if refid != ref or span == 0:
                    continue
                inter = start1 <= pos0 + span and pos0 + 1 <= end1
                listed = any(a <= voff < b for a, b in chunks)
                if inter and listed:
                    assert any(a <= voff < b for a, b in kept), (refs, ref, start1, end1, pos0, hex(voff), hex(V))
    assert n_cut > 200 and n_pruned_records > 1000, (n_regions, n_cut, n_pruned_records)
