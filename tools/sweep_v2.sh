#!/bin/bash
# dev tool: rebuild K1 v2 with different round/window sizes on the GPU box and time the inflate stage
for cfg in "9 7168 9" "13 10240 7" "17 14336 6" "17 12288 6" "21 16384 5"; do
  set -- $cfg
  touch datafusion-bio-formats_amd/csrc/inflate_v2.hip
  make -C datafusion-bio-formats_amd/csrc EXTRA="-DV2_SUB_DW=$1 -DV2_WIN_BYTES=$2" >/dev/null 2>&1
  BIOSCAN_DEBUG=1 BIOSCAN_V2_WG_PER_CU=$3 python bench.py --blocks 65536 --steps 2 --warmup 1 --no-cpu-baseline 2> /tmp/v2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg $cfg', 'inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'])"
  grep "bioscan" /tmp/v2.err | grep -v chain | tail -6 | tr '\n' ';'; echo
done
