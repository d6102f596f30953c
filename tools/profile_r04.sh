#!/bin/bash
# Regenerates the rocprofv3 evidence of round 4 under gpurun_out/prof_r04 (copy what is to be kept into profiles/r03/):
#   kernel-trace stats of the DEFAULT bench command's device-resident leg (config 2, 650 000 members) and of the 65 536-member
#   run; HBM traffic of K1 (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ issue counters, stamped with the sha256 of
#   the K1 source they were collected on (bench.py only uses them when the stamp matches the built kernel).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end > $O/bench_full_under_rocprof.json 2> $O/stats_full.log || echo "stats_full failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_65536 -- python3 $R/bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_65536_under_rocprof.json 2> $O/stats_65536.log || echo "stats_65536 failed"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_$tag.json 2> $O/pmc_$tag.log || echo "pmc $tag failed"
done
sha256sum $R/datafusion-bio-formats_amd/csrc/inflate_v4.hip | cut -c1-16 > $O/k1_pmc_source.sha256
(cd $R && cat gpurun_commit.txt 2>/dev/null) > $O/commit.txt
python3 - <<PY
import glob, shutil, os
O = "$O"
def first(pat):
    fs = sorted(glob.glob(os.path.join(O, pat), recursive=True))
    return fs[0] if fs else None
for src, dst in (("stats_full/**/*kernel_stats.csv", "kernel_stats_default_bench_650000blocks.csv"),
                 ("stats_65536/**/*kernel_stats.csv", "kernel_stats_65536blocks.csv"),
                 ("pmc_FETCH_SIZE/**/*counter_collection.csv", "k1_pmc_FETCH_SIZE_65536blocks.csv"),
                 ("pmc_WRITE_SIZE/**/*counter_collection.csv", "k1_pmc_WRITE_SIZE_65536blocks.csv"),
                 ("pmc_SQ_WAVES/**/*counter_collection.csv", "k1_pmc_SQ_issue_65536blocks.csv")):
    f = first(src)
    print(dst, "<-", f)
    if f:
        # counter files list every dispatch of the process: keep the K1 rows (and the header line)
        if "pmc" in dst:
            with open(f) as fi, open(os.path.join(O, dst), "w") as fo:
                for i, line in enumerate(fi):
                    if i == 0 or "k_bgzf_inflate" in line or "k_bgzf_crc32" in line or "k_bgzf_headers" in line:
                        fo.write(line)
        else:
            shutil.copy(f, os.path.join(O, dst))
PY
ls -la $O | head -30
