#!/bin/bash
# SQ issue counters of K1 at 65536 members (own --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sq
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT --output-format csv -d $O -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1 || echo "pmc failed"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "inflate" in k: print(k, dict(v))
PY
