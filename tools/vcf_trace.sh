#!/bin/bash
# dev tool (GPU box): kernel trace of the VCF benches -- launches, busy and idle time inside one timed step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/vcf_trace
rm -rf $O; mkdir -p $O
for f in vcf-sites vcf-samples; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$f -- python3 $R/bench.py --format $f --steps 3 --warmup 1 > $O/$f.json 2> $O/$f.log || echo "$f failed"
  python3 - <<PY
import csv, glob, json
f = sorted(glob.glob("$O/$f/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last inflate launch backwards ... take the last third of the K1 launches' first
k1 = [i for i, r in enumerate(rows) if "inflate_v3" in r["Kernel_Name"] and int(r.get("Grid_Size_X") or r.get("Grid_Size")) > 64 * 64]
per = max(1, len(k1) // 4)
i0 = k1[-per]
seg = rows[i0:]
t0 = int(seg[0]["Start_Timestamp"]); cur = t0; idle = 0; busy = {}; cnt = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("bioscan::", "").replace("void ", "")[:38]
    busy[n] = busy.get(n, 0) + e - s; cnt[n] = cnt.get(n, 0) + 1
    if s > cur: idle += s - cur
    cur = max(cur, e)
print("$f: last step span %.3f ms, %d launches, device idle %.3f ms" % ((cur - t0) / 1e6, len(seg), idle / 1e6))
for n, v in sorted(busy.items(), key=lambda x: -x[1])[:12]: print("   %-40s %4d %8.3f ms" % (n, cnt[n], v / 1e6))
r = json.loads([l for l in open("$O/$f.json") if l.startswith("{")][-1]); print("   bench:", r["value"], r["ms_per_step"], r["stage_ms"])
PY
done
