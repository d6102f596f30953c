#!/bin/bash
# dev tool (GPU box): kernel trace of the indexed scan of config 2 (8 partitions, one after the other) -- K1's launches
# one by one, the gaps between the kernels of the step, against the one launch of the sequential scan
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/indexed_trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --mode indexed --partition-threads 1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench.json 2> $O/bench.log || echo "trace failed"
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
k1 = [r for r in rows if "inflate_v3" in r["Kernel_Name"]]
# the last step = the last 8 big launches (+ small header launches)
big = [r for r in k1 if int(r["Grid_Size"]) >= 64 * 4000][-8:]
t0 = int(big[0]["Start_Timestamp"])
last = None
tot = 0
for r in big:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("K1 start %8.3f ms  dur %7.3f ms  grid %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Grid_Size"]))
    tot += e - s
print("sum of the 8 K1 launches %.3f ms" % (tot / 1e6))
# everything between the first K1 of the step and the end of the step: busy time by kernel, idle time
seg = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
busy = {}
cur_end = t0; idle = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("bioscan::", "").replace("void ", "")[:40]
    busy[n] = busy.get(n, 0) + e - s
    if s > cur_end: idle += s - cur_end
    cur_end = max(cur_end, e)
print("step span %.3f ms, device idle inside it %.3f ms" % ((cur_end - t0) / 1e6, idle / 1e6))
for n, v in sorted(busy.items(), key=lambda x: -x[1])[:14]: print("  %-42s %8.3f ms" % (n, v / 1e6))
PY
tail -1 $O/bench.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['stage_ms'])"
