#!/bin/bash
# dev tool: rebuild K1 v2 with different compile-time knobs on the GPU box and time the inflate stage
C=/root/repo/datafusion-bio-formats_amd/csrc
for cfg in "${CFGS[@]}"; do
  touch $C/inflate_v2.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1
  BIOSCAN_DEBUG=1 python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/v2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg $cfg', 'inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'])"
  grep -E "decode passes|residency" /tmp/v2.err | tail -2
done
touch $C/inflate_v2.hip
make -C $C >/dev/null 2>&1
