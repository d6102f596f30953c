#!/usr/bin/env python3
"""dev tool: copy-kernel / compute-kernel interleaving of the host stream from a rocprofv3 kernel trace (newest under DIR)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/e2e_trace"
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cp = [r for r in rows if "copyBuffer" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
half = cp[len(cp) // 2:]
t0 = int(half[0]["Start_Timestamp"]); t1 = max(int(x["End_Timestamp"]) for x in half)
busy = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in half)
print(f, "\nsecond run copies: span %.1f ms busy %.1f ms n %d" % ((t1 - t0) / 1e6, busy / 1e6, len(half)))
mid = int(half[len(half) // 2]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < mid - 2_000_000 or s > mid + float(sys.argv[2] if len(sys.argv) > 2 else 60) * 1e6:
        continue
    if (e - s) < 150000:
        continue
    print(f"{(s - mid) / 1e6:8.2f} {(e - mid) / 1e6:8.2f} {(e - s) / 1e6:7.2f} q{r['Queue_Id']:>2s} {r['Kernel_Name'].split('(')[0][-40:]}")
