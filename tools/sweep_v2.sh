#!/bin/bash
# dev tool: rebuild K1 v2 with different compile-time knobs on the GPU box and time the inflate stage
for cfg in "-DV2_SUB_DW=7 -DV2_WIN_BYTES=5632" "-DV2_SUB_DW=9 -DV2_WIN_BYTES=6656" "-DV2_SUB_DW=9 -DV2_WIN_BYTES=7168" "-DV2_SUB_DW=11 -DV2_WIN_BYTES=7680" "-DV2_SUB_DW=11 -DV2_WIN_BYTES=8704" "-DV2_SUB_DW=7 -DV2_WIN_BYTES=6144"; do
  touch datafusion-bio-formats_amd/csrc/inflate_v2.hip
  make -C datafusion-bio-formats_amd/csrc EXTRA="$cfg" >/dev/null 2>&1
  BIOSCAN_LAPS=1 python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/v2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg $cfg', 'inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'])"
done
touch datafusion-bio-formats_amd/csrc/inflate_v2.hip
make -C datafusion-bio-formats_amd/csrc >/dev/null 2>&1
