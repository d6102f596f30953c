#!/bin/bash
# SQ issue / activity counters of K1 v3 and v4 at 65536 members (own --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in 3 4; do
for GRP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  O=$R/gpurun_out/prof_sq_v${V}_$(echo $GRP | cut -c1-14 | tr ' ' '_')
  rm -rf $O; mkdir -p $O
  BIOSCAN_K1=$V rocprofv3 --kernel-trace --pmc $GRP --output-format csv -d $O -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/log.txt 2>&1 || echo "pmc failed"
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "inflate" in k or "headers" in k: print("K1 v$V", k, {a: "%.4g" % b for a, b in v.items()})
PY
done
done
