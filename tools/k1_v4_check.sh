#!/bin/bash
# dev tool: K1 v4 (BIOSCAN_K1=4) correctness against the K1 tests and fuzzers, then v3 / v4 timing side by side
# QUICK=1: inflate fuzz test only, no fuzz campaign; VERS="4": time only these versions; EXTRA=...: rebuild with flags first
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
# a build with EXTRA flags replaces the product library in place: whatever ends this script, the default build comes back
if [ -n "$EXTRA" ]; then touch $R/datafusion-bio-formats_amd/csrc/inflate_v4.hip; trap 'touch $R/datafusion-bio-formats_amd/csrc/inflate_v4.hip; make -C $R/datafusion-bio-formats_amd/csrc >/dev/null 2>&1' EXIT; fi
make -C $R/datafusion-bio-formats_amd/csrc EXTRA="$EXTRA" >/dev/null 2>&1 || { echo BUILD FAILED; exit 1; }
O=$R/gpurun_out/k1_v4_check.txt
mkdir -p $R/gpurun_out; : > $O
export BIOSCAN_K1=4
if [ -n "$QUICK" ]; then
  timeout -k 10 600 python -m pytest $R/tests/test_gpu_inflate_fuzz.py -m gpu -x -q >> $O 2>&1 || { tail -40 $O; exit 1; }
else
  timeout -k 10 900 python -m pytest $R/tests/test_gpu_inflate_fuzz.py $R/tests/test_gpu_bam_parity.py $R/tests/test_gpu_bam_edge_cases.py -m gpu -x -q >> $O 2>&1 || { tail -40 $O; exit 1; }
  timeout -k 10 600 python $R/tools/fuzz_k1.py ${FUZZ_N:-12} 4000 >> $O 2>&1 || { tail -30 $O; exit 1; }
fi
grep -E "passed|failed|fuzz ok" $O
for V in ${VERS:-3 4}; do
  export BIOSCAN_K1=$V
  BIOSCAN_DEBUG=1 timeout -k 10 300 python $R/bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2> /tmp/k1.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K1 v$V 65536 blocks: inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'], d['value'])" >> $O 2>&1
  grep -E "decode passes|residency|of wave cycles|mini-rounds|LZ77" /tmp/k1.err | tail -20 >> $O
  timeout -k 10 300 python $R/bench.py --blocks 262144 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K1 v$V 262144 blocks: inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'], d['value'])" >> $O 2>&1
done
grep -A40 "K1 v" $O | tail -${TAILN:-44}
