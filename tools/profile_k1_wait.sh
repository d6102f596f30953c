#!/bin/bash
# dev tool: where K1's waves wait -- SQ wait / active counters, LDS conflicts, texture path (own --pmc passes, 65536 members)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_wait
rm -rf $O; mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"; do   # (the TA_* / TCP_* counters are in tools/profile_k1_tcp.sh, one or two per pass: six of them in ONE set were refused by rocprofiler -- error code 38, "Request exceeds the capabilities of the hardware to collect" -- which aborted the process; that was a refused counter set, not a hang)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/p$i.json 2> $O/p$i.log || echo "pmc pass $i failed"
  i=$((i+1))
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("bioscan::", "")
        if int(r["Grid_Size"]) > 6400: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if any(t in k for t in ("inflate", "pass2", "pass1", "crc")): print(k, {a: round(b / 1e6, 2) for a, b in v.items()})
PY
cat $O/summary.txt
