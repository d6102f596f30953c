#!/bin/bash
# dev tool: per-kernel times of the extract at 65536 members (rocprofv3 --kernel-trace --stats) for the flags in $CFGS_STR
C=datafusion-bio-formats_amd/csrc
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/bam_rows.hip; make -C $C >/dev/null 2>&1' EXIT
O=$PWD/gpurun_out/extract_prof.txt
mkdir -p gpurun_out; : > $O
export TMPDIR=/tmp
IFS=';' read -ra CFGS <<< "${CFGS_STR:-}"
[ ${#CFGS[@]} -eq 0 ] && CFGS=("")
i=0
for cfg in "${CFGS[@]}"; do
  touch $C/bam_rows.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg] BUILD FAILED" >> $O; continue; }
  rm -rf /tmp/xp$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xp$i -- python3 bench.py --blocks 65536 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > /tmp/xp$i.log 2>&1
  echo "cfg [$cfg]" >> $O
  python3 - /tmp/xp$i >> $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('bioscan::', '')
    if any(t in n for t in ('bam_', 'seg_', 'crc', 'inflate')):
        print('  %-28s calls %4s avg_us %10.1f' % (n, r['Calls'], float(r['AverageNs']) / 1e3))
PY
  i=$((i+1))
done
touch $C/bam_rows.hip; make -C $C >/dev/null 2>&1
cat $O
