#!/bin/bash
# dev tool: instruction-cache and issue-wait counters of K1 (own --pmc passes, 65536 members)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_icache
rm -rf $O; mkdir -p $O
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/p$i.json 2> $O/p$i.log \
    || { echo "pmc pass $i ($c) failed:"; grep -m2 -i "error code\|exceeds\|invalid\|not found" $O/p$i.log; }
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Kernel_Name"] and int(r["Grid_Size"]) > 6400:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print({a: round(b / 2e6, 2) for a, b in agg.items()}, "(x1e6 per launch)")
PY
