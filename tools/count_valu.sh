#!/bin/bash
# dev tool: static VALU / SALU / LDS instruction counts of the K1 decode loops (K1 is VALU-issue bound: SQ_INSTS_VALU x 4
# cycles = 96 % of the kernel's cycles), from the gfx950 ISA blocks of the loops that hold the V2LOOP markers
SRC=${1:-/root/repo/datafusion-bio-formats_amd/csrc/inflate_v3.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV3_ASM_MARKERS -DV4_ASM_MARKERS $EXTRA -S --cuda-device-only -o /tmp/v2m.s $SRC 2>/dev/null
python3 - <<'PY'
import re
lines=open('/tmp/v2m.s').read().split('\n')
blocks=[]  # (name, header, [instrs])
cur=None
for l in lines:
    m=re.match(r'^(\.LBB0_(\d+)):|^; %bb\.(\d+):',l)
    if m:
        name='BB0_'+(m.group(2) or m.group(3))
        h=re.search(r'in Loop: Header=(BB0_\d+) Depth=(\d+)',l)
        cur=[name,(h.group(1),int(h.group(2))) if h else None,[]]
        blocks.append(cur); continue
    if cur is None: continue
    h=re.search(r'=>\s+This (Inner )?Loop Header: Depth=(\d+)',l)
    if h and not cur[2]: cur[1]=(cur[0],int(h.group(2)))
    t=l.strip()
    if t and not t.startswith((';','.')): cur[2].append(t)
    if re.search('V[34]LOOP_BEGIN', l): cur[2].append(l.strip())
seen=set()
for b in blocks:
    for t in b[2]:
        m=re.search(r'V[34]LOOP_BEGIN (\d)',t)
        if m and b[1] and b[1] not in seen:
            seen.add(b[1])
            body=[x for bb in blocks if bb[1]==b[1] for x in bb[2]]
            v=sum(1 for x in body if x.startswith('v_')); s=sum(1 for x in body if x.startswith('s_') and not x.startswith(('s_waitcnt','s_nop')))
            d=sum(1 for x in body if x.startswith('ds_')); f=sum(1 for x in body if x.startswith(('flat_','global_','buffer_','scratch_')))
            print("MODE %s loop %s: VALU %d SALU %d LDS %d VMEM %d"%(m.group(1),b[1][0],v,s,d,f))
PY
grep -E "\.vgpr_count|\.private_segment_fixed_size" /tmp/v2m.s | head -3
