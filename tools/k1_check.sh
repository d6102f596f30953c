#!/bin/bash
# dev tool: K1 correctness (inflate fuzz + BAM parity tests) then the 65536-member timing with the debug anatomy
set -o pipefail
O=gpurun_out/k1_check.txt
mkdir -p gpurun_out; : > $O
timeout -k 10 600 python -m pytest tests/test_gpu_inflate_fuzz.py tests/test_gpu_bam_parity.py tests/test_gpu_bam_edge_cases.py -m gpu -x -q >> $O 2>&1 || { tail -30 $O; exit 1; }
BIOSCAN_DEBUG=1 timeout -k 10 300 python bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/k1.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'], d['value'])" >> $O 2>&1
grep -E "inflate v2|decode passes|residency|of wave cycles|LZ77|mini-rounds" /tmp/k1.err | tail -9 >> $O
tail -14 $O
