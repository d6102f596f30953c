#!/bin/bash
# dev tool: look-ahead inflate knobs against the serial pipeline (bench.py, device-resident leg only)
# usage: la_sweep.sh [BLOCKS=262144]
B=${1:-262144}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/la_sweep
mkdir -p $O
run() {
  name=$1; shift
  env "$@" python3 $R/bench.py --blocks $B --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; return; }
  python3 - "$name" "$O/$name.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
s = r["stage_ms"]
print(f"{sys.argv[1]:34s} {r['value']:8.1f} Mrec/s  step {r['ms_per_step']:7.2f} ms | inflate {s['inflate']:7.2f} crc {s['crc32']:5.2f} chain {s['record_chain']:5.2f} extract {s['extract']:6.2f} | sum {s['inflate']+s['crc32']+s['record_chain']+s['extract']:7.2f}")
PY
}
run serial_onechunk       BIOSCAN_LOOKAHEAD=0
run serial_chunk65536     BIOSCAN_LOOKAHEAD=0 BIOSCAN_CHUNK_MEMBERS_DEVICE=65536
run la_wpw4_pw1           BIOSCAN_LOOKAHEAD=1
run la_wpw4_pw1_noprio    BIOSCAN_LOOKAHEAD=1 BIOSCAN_LA_PRIORITY=0
run la_wpw4_pw1_c131072   BIOSCAN_LOOKAHEAD=1 BIOSCAN_CHUNK_MEMBERS_DEVICE=131072
run la_wpw4_pw2           BIOSCAN_LOOKAHEAD=1 BIOSCAN_K1_PER_WAVE=2
run la_persistent16       BIOSCAN_LOOKAHEAD=1 BIOSCAN_K1_ONESHOT=0 BIOSCAN_K1_WAVES_PER_CU=16
