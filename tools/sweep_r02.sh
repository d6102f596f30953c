#!/bin/bash
# dev tool (round 2): sub-stream length x pre-roll sweep of K1 on the GPU box; results in gpurun_out/sweep_r02.txt
C=$GRAFT_REPO_ROOT/datafusion-bio-formats_amd/csrc
O=$GRAFT_REPO_ROOT/gpurun_out/sweep_r02.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $O
CFGS=("" "-DV2_SUB_DW=15 -DV2_OV_BITS=96" "-DV2_SUB_DW=15 -DV2_OV_BITS=192" "-DV2_SUB_DW=25 -DV2_OV_BITS=256" "-DV2_SUB_DW=41 -DV2_OV_BITS=256" "-DV2_SUB_DW=41 -DV2_OV_BITS=480" "-DV2_SUB_DW=63 -DV2_OV_BITS=480" "-DV2_SUB_DW=101 -DV2_OV_BITS=640")
for cfg in "${CFGS[@]}"; do
  touch $C/inflate_v2.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg $cfg BUILD FAILED" >> $O; continue; }
  BIOSCAN_DEBUG=1 timeout -k 10 240 python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/v2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg]', 'inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'])" >> $O 2>&1
  grep -E "decode passes|residency|of wave cycles|LZ77" /tmp/v2.err | tail -8 >> $O
done
touch $C/inflate_v2.hip
make -C $C >/dev/null 2>&1
cat $O
