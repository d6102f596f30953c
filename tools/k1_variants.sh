#!/bin/bash
# dev tool (GPU box): time the scan with each library variant built by tools/build_variant.sh
# usage: k1_variants.sh "name1 name2 ..." [BLOCKS ...]
R=${GRAFT_REPO_ROOT:-.}
names=$1; shift
sizes=${@:-"65536 262144"}
cp $R/datafusion-bio-formats_amd/libbioscan.so /tmp/libbioscan_base.so
for n in base $names; do
  if [ $n = base ]; then cp /tmp/libbioscan_base.so $R/datafusion-bio-formats_amd/libbioscan.so; else cp $R/tools/_build/variants/$n/libbioscan.so $R/datafusion-bio-formats_amd/libbioscan.so; fi
  BIOSCAN_DEBUG=1 python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep "decode passes" | tail -1 | sed "s/^/$n: /"
  for B in $sizes; do
    python3 $R/bench.py --blocks $B --steps 6 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=r['stage_ms']
print('$n', 'blocks', r['config']['n_blocks_per_gpu'], 'step', r['ms_per_step'], 'inflate', s['inflate'], 'Mrec/s', r['value'])"
  done
done
cp /tmp/libbioscan_base.so $R/datafusion-bio-formats_amd/libbioscan.so
