#!/usr/bin/env python3
"""dev tool: print the kernel timeline of the last timed step from a rocprofv3 kernel trace (newest file under DIR)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/la_trace"
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "bioscan" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k1 = [r for r in rows if "inflate" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 6400]
t0 = int(k1[len(k1) // 2]["Start_Timestamp"])
print(f)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("bioscan::", "").replace("void ", "")
    if (e - s) / 1e6 < minus and "inflate" not in name:
        continue
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} ms  q{r.get('Queue_Id', '?'):>3s} grid {r['Grid_Size_X']:>9s} {name}")
