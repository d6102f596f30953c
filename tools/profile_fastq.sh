#!/bin/bash
# kernel-trace stats of the FASTQ bench (131072 members) -> gpurun_out/prof/fq
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/fq -- python3 $R/bench.py --format fastq --blocks 131072 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/fq.log 2>&1
python3 - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof/fq/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), ("%.3f"%(float(r["AverageNs"])/1e6)).rjust(9), r["Percentage"].rjust(6))
PY
