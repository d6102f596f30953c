#!/bin/bash
# dev tool: the texture / L2 path of K1 -- TA_* / TCP_* / TCC_* counters, ONE small set per --pmc pass.
# (r02 asked for six TA_/TCP_ counters in one set; rocprofiler refused it -- "error code 38: Request exceeds the
# capabilities of the hardware to collect" -- and aborted the process at its first HIP call.  That was a refused counter
# configuration, not a hang: the sets below hold one or two counters of one block each.)
# The program sits directly after `--`; 65536 members, one timed step; the CSVs are stamped with the commit and the K1
# source hash by the caller (tools/stamp_profiles.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_tcp
rm -rf $O; mkdir -p $O
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
         "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/p$i.json 2> $O/p$i.log \
    || { echo "pmc pass $i ($c) failed:"; grep -m2 -i "error code\|exceeds\|invalid" $O/p$i.log; }
  i=$((i+1))
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("bioscan::", "")
        if int(r["Grid_Size"]) > 6400:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if any(t in k for t in ("inflate", "pass2", "pass1", "crc")): print(k, {a: round(b / 1e6, 3) for a, b in v.items()}, "(x1e6, summed over the launches of the run)")
PY
cat $O/summary.txt
