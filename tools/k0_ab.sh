#!/bin/bash
# dev tool (GPU box): K1 with and without the K0 header pre-pass -- kernel durations (kernel trace) and VALU counters, 65536 members
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/k0_ab
rm -rf $O; mkdir -p $O
for k in 1 0; do
  export BIOSCAN_K1_PREHEADERS=$k
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$k -- python3 $R/bench.py --blocks 65536 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/t$k.json 2> $O/t$k.log || echo "trace $k failed"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/p$k -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/p$k.json 2> $O/p$k.log || echo "pmc $k failed"
done
python3 - <<PY
import csv, glob, collections
for k in (1, 0):
    for f in glob.glob("$O/t%d/**/*kernel_stats.csv" % k, recursive=True):
        for r in csv.DictReader(open(f)):
            if "inflate" in r["Name"] or "headers" in r["Name"]: print("K0", "on " if k else "off", r["Name"].split("(")[0][-40:], "calls", r["Calls"], "avg ms", round(float(r["AverageNs"]) / 1e6, 3))
    for f in glob.glob("$O/p%d/**/*counter_collection.csv" % k, recursive=True):
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "inflate" in r["Kernel_Name"] and int(r["Grid_Size"]) > 6400: agg[r["Counter_Name"]] += float(r["Counter_Value"])
        print("K0", "on " if k else "off", {a: round(b / 2e9, 3) for a, b in agg.items()}, "(G per launch)")
PY
