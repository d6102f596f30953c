#!/bin/bash
# dev tool: rebuild K1 v2 with different compile-time knobs on the GPU box and time the inflate stage
for cfg in "-DV2_OV_BITS=32" "-DV2_OV_BITS=64" "-DV2_OV_BITS=96" "-DV2_OV_BITS=128" "-DV2_OV_BITS=160"; do
  touch datafusion-bio-formats_amd/csrc/inflate_v2.hip
  make -C datafusion-bio-formats_amd/csrc EXTRA="$cfg" >/dev/null 2>&1
  BIOSCAN_DEBUG=1 python bench.py --blocks 65536 --steps 2 --warmup 1 --no-cpu-baseline 2> /tmp/v2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg $cfg', 'inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'])"
  grep "inflate v2" /tmp/v2.err | tail -1
done
