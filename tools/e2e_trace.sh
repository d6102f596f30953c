#!/bin/bash
# dev tool: kernel + memory-copy timeline of the host stream (bioscan_execute / bioscan_next) on 262144 members
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2e_trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $R/bench.py --blocks ${1:-262144} --steps 1 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo failed; tail -5 $O/bench.err; }
python3 - <<PY
import csv, glob, json
r = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("end_to_end", r["end_to_end"])
f = glob.glob("$O/t/**/*memory_copy_trace.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
d2h = [x for x in rows if "DEVICE_TO_HOST" in x.get("Direction", "") and int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) > 0]
big = [x for x in d2h if int(x.get("Bytes", x.get("Size", 0)) or 0) > (1 << 20)]
print(len(d2h), "d2h copies,", len(big), "> 1 MiB")
if big:
    t0 = int(big[0]["Start_Timestamp"]); t1 = max(int(x["End_Timestamp"]) for x in big)
    tot = sum(int(x.get("Bytes", x.get("Size", 0))) for x in big)
    busy = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in big)
    print(f"span {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, {tot / 1e9:.2f} GB, {tot / (t1 - t0):.2f} GB/s over the span, {tot / busy:.2f} GB/s while copying")
    # the last run only (second half)
    half = big[len(big) // 2:]
    t0 = int(half[0]["Start_Timestamp"]); t1 = max(int(x["End_Timestamp"]) for x in half)
    tot = sum(int(x.get("Bytes", x.get("Size", 0))) for x in half); busy = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in half)
    print(f"second run: span {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, {tot / 1e9:.2f} GB, {tot / (t1 - t0):.2f} GB/s over the span, {tot / busy:.2f} GB/s while copying")
    gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6, int(a["End_Timestamp"])) for a, b in zip(half, half[1:]))
    print("largest gaps between consecutive big copies (ms):", [round(g, 3) for g, _ in gaps[-12:]])
    sizes = sorted(int(x.get("Bytes", x.get("Size", 0))) for x in half)
    print("copy sizes MB: min %.1f median %.1f max %.1f, count %d" % (sizes[0] / 1e6, sizes[len(sizes) // 2] / 1e6, sizes[-1] / 1e6, len(sizes)))
PY
