#!/bin/bash
# kernel-trace stats of the two VCF benches -> gpurun_out/prof/vcf_{sites,samples}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
for f in sites samples; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/vcf_$f -- python3 $R/bench.py --format vcf-$f --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/vcf_$f.log 2>&1
  python3 - <<PY
import csv,glob,os
fs=sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof/vcf_$f/*/*_kernel_stats.csv"), key=os.path.getmtime)
print("== vcf-$f")
for r in list(csv.DictReader(open(fs[-1])))[:16]:
    print(r["Name"][:58].ljust(58), r["Calls"].rjust(4), ("%.3f"%(float(r["AverageNs"])/1e6)).rjust(8), ("%.3f"%(float(r["TotalDurationNs"])/1e6/4)).rjust(8), r["Percentage"].rjust(6))
PY
done
