#!/bin/bash
# dev tool: which pipe of the CU K1 keeps busy -- LDS array cycles and bank conflicts, per-pipe active cycles; two SQ counters
# per --pmc pass (see profile_k1_tcp.sh for why the sets are small).
# The program sits directly after `--`; 65536 members, one timed step; the CSVs are stamped with the commit and the K1
# source hash by the caller (tools/stamp_profiles.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_lds
rm -rf $O; mkdir -p $O
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" \
         "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
         "SQ_INSTS_LDS SQ_BUSY_CYCLES" \
         "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"; do
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
