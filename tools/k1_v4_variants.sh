#!/bin/bash
# dev tool: time K1 v4 builds with different compile-time parameters (semicolon-separated EXTRA strings in CFGS_STR)
R=${GRAFT_REPO_ROOT:-.}
C=$R/datafusion-bio-formats_amd/csrc
O=$R/gpurun_out/k1_v4_variants.txt
mkdir -p $R/gpurun_out; : > $O
export BIOSCAN_K1=4
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/inflate_v4.hip; make -C $C >/dev/null 2>&1' EXIT
IFS=';' read -ra CFGS <<< "${CFGS_STR:-}"
for cfg0 in "${CFGS[@]}"; do
  # "@NAME=VALUE flags": an environment setting for the runs of this configuration
  cfg="$cfg0"; unset BIOSCAN_K1_PREHEADERS
  if [[ "$cfg0" == @* ]]; then ev="${cfg0%% *}"; export "${ev:1}"; cfg="${cfg0#* }"; [ "$cfg" == "$cfg0" ] && cfg=""; fi
  touch $C/inflate_v4.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg0] BUILD FAILED" >> $O; continue; }
  timeout -k 10 200 python -m pytest $R/tests/test_gpu_inflate_fuzz.py -m gpu -x -q 2>&1 | tail -1 | sed "s|^|cfg [$cfg0] tests: |" >> $O
  BIOSCAN_DEBUG=1 timeout -k 10 300 python $R/bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2> /tmp/k1.err >/dev/null
  grep -E "residency|mini-rounds" /tmp/k1.err | tail -2 >> $O
  [ -n "$ANATOMY" ] && grep -E "of wave cycles" /tmp/k1.err | tail -14 >> $O
  timeout -k 10 300 python $R/bench.py --blocks 262144 --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg [$cfg0] 262144 blocks: inflate_ms', d['stage_ms']['inflate'], 'step', d['ms_per_step'])" >> $O 2>&1
done
cat $O
