#!/bin/bash
# dev tool (GPU box): BIOSCAN_DEBUG wave-cycle anatomy of K1 for library variants, 65536 members
R=${GRAFT_REPO_ROOT:-.}
cp $R/datafusion-bio-formats_amd/libbioscan.so /tmp/libbioscan_base.so
for n in base $1; do
  if [ $n = base ]; then cp /tmp/libbioscan_base.so $R/datafusion-bio-formats_amd/libbioscan.so; else cp $R/tools/_build/variants/$n/libbioscan.so $R/datafusion-bio-formats_amd/libbioscan.so; fi
  echo "== $n"
  BIOSCAN_DEBUG=1 python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep "bioscan\]" | grep -v "verify round\|residency" | tail -9
done
cp /tmp/libbioscan_base.so $R/datafusion-bio-formats_amd/libbioscan.so
