#!/bin/bash
# dev tool (GPU box): ratio and speed of W2 (k_bgzf_deflate) for ';'-separated build flags in $CFGS_STR (tools/experiments/deflate_ratio.py)
R=${GRAFT_REPO_ROOT:-.}
C=$R/datafusion-bio-formats_amd/csrc
O=$R/gpurun_out/w2_variants.txt
mkdir -p $R/gpurun_out; : > $O
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/bam_write.hip; make -C $C >/dev/null 2>&1' EXIT
IFS=';' read -ra CFGS <<< "${CFGS_STR:-}"
[ ${#CFGS[@]} -eq 0 ] && CFGS=("")
for cfg in "${CFGS[@]}"; do
  touch $C/bam_write.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg] BUILD FAILED" >> $O; continue; }
  timeout -k 10 200 python3 $R/tools/experiments/deflate_ratio.py 2>&1 | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print('cfg [$cfg]', d['input'][:12], 'ratio', d['w2_ratio'], 'GB/s', d['GB_per_s'], 'ms', d['kernel_ms'])
    except Exception: print('cfg [$cfg]', l.strip()[:200])
" >> $O
done
cat $O
