#!/bin/bash
# dev tool: per-kernel times of the FASTQ bench (rocprofv3 --kernel-trace --stats) for the ';'-separated build flags in $CFGS_STR
C=$GRAFT_REPO_ROOT/datafusion-bio-formats_amd/csrc
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/fastq_kernels.hip; make -C $C >/dev/null 2>&1' EXIT
O=$GRAFT_REPO_ROOT/gpurun_out/fastq_prof.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; : > $O
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra CFGS <<< "${CFGS_STR:-}"
[ ${#CFGS[@]} -eq 0 ] && CFGS=("")
i=0
for cfg in "${CFGS[@]}"; do
  touch $C/fastq_kernels.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg] BUILD FAILED" >> $O; continue; }
  rm -rf /tmp/fqp$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fqp$i -- python3 $GRAFT_REPO_ROOT/bench.py --format fastq --no-cpu-baseline > /tmp/fqp$i.log 2>&1
  echo "cfg [$cfg] $(tail -1 /tmp/fqp$i.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stage_ms"])' 2>/dev/null)" >> $O
  python3 - /tmp/fqp$i >> $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('bioscan::', '')
    if any(t in n for t in ('fastq_', 'nl_', 'scatter')): print('  %-28s calls %4s avg_ms %8.3f' % (n[:28], r['Calls'], float(r['AverageNs']) / 1e6))
PY
  i=$((i+1))
done
touch $C/fastq_kernels.hip; make -C $C >/dev/null 2>&1
cat $O
