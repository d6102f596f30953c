#!/bin/bash
# dev tool: A/B two K1 sources on the GPU box (A = $1, B = the tree), timing the inflate stage
run() { python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stage_ms']['inflate'])"; }
C=datafusion-bio-formats_amd/csrc
cp $C/inflate_v2.hip /tmp/B.hip
for rep in 1 2; do
  cp $1 $C/inflate_v2.hip; make -C $C >/dev/null 2>&1; run A
  cp /tmp/B.hip $C/inflate_v2.hip; make -C $C >/dev/null 2>&1; run B
done
