#!/bin/bash
# dev tool: K1 timing at 65536 members with the debug anatomy, for the build flags in $EXTRA (rebuilds, then restores)
C=datafusion-bio-formats_amd/csrc
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/inflate_v3.hip $C/bgzf_source.cpp; make -C $C >/dev/null 2>&1' EXIT
O=gpurun_out/k1_time.txt
mkdir -p gpurun_out; : > $O
IFS=';' read -ra CFGS <<< "${CFGS_STR:-}"
[ ${#CFGS[@]} -eq 0 ] && CFGS=("")
for cfg in "${CFGS[@]}"; do
  touch $C/inflate_v3.hip $C/bgzf_source.cpp
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg] BUILD FAILED" >> $O; continue; }
  BIOSCAN_DEBUG=1 timeout -k 10 300 python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/k1.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg] inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'], d['value'])" >> $O 2>&1
  grep -E "decode passes|of wave cycles|mini-rounds|fix pass" /tmp/k1.err | tail -${TAILN:-11} >> $O
done
touch $C/inflate_v3.hip $C/bgzf_source.cpp; make -C $C >/dev/null 2>&1
cat $O
