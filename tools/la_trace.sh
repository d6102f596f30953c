#!/bin/bash
# dev tool: kernel timeline of one look-ahead scan (rocprofv3 --kernel-trace): who runs when, what overlaps
B=${1:-262144}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/la_trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --blocks $B --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench.json 2> $O/bench.err || { echo failed; tail -5 $O/bench.err; }
python3 - <<PY
import csv, glob
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bioscan" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed step = the last half of the K1 launches
k1 = [r for r in rows if "inflate" in r["Kernel_Name"] and int(r["Grid_Size"]) > 6400]
half = k1[len(k1) // 2]
t0 = int(half["Start_Timestamp"])
out = open("$O/timeline.txt", "w")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0: continue
    name = r["Kernel_Name"].split("(")[0].replace("bioscan::", "")
    if (e - s) < 20000 and "inflate" not in name: continue
    out.write(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} ms  q{r.get('Queue_Id','?'):>3s} grid {r['Grid_Size']:>9s} {name}\n")
out.close()
print(open("$O/timeline.txt").read()[:6000])
PY
