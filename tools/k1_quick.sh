#!/bin/bash
# dev tool: K1 counters (BIOSCAN_DEBUG=1) and time on 65536 members, then time on 262144 members
R=${GRAFT_REPO_ROOT:-.}
BIOSCAN_DEBUG=1 python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep -v "verify round" | grep "bioscan\]" | tail -9
for B in 65536 262144; do
python3 $R/bench.py --blocks $B --steps 6 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=r['stage_ms']
print('blocks', r['config']['n_blocks_per_gpu'], 'step', r['ms_per_step'], 'inflate', s['inflate'], 'Mrec/s', r['value'])"
done
