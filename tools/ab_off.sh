#!/bin/bash
# A/B of optional K1 fast paths: BIOSCAN_V2_OFF bit mask (1 = dword window copy off)
for rep in 1 2; do for off in 0 1; do
  BIOSCAN_V2_OFF=$off python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('off=$off', d['ms_per_step'], d['stage_ms']['inflate'])"
done; done
