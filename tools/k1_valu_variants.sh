#!/bin/bash
# dev tool (GPU box): SQ instruction counters of K1 for library variants (tools/build_variant.sh), 65536 members
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/k1_valu
rm -rf $O; mkdir -p $O
cp $R/datafusion-bio-formats_amd/libbioscan.so /tmp/libbioscan_base.so
for n in base $1; do
  if [ $n = base ]; then cp /tmp/libbioscan_base.so $R/datafusion-bio-formats_amd/libbioscan.so; else cp $R/tools/_build/variants/$n/libbioscan.so $R/datafusion-bio-formats_amd/libbioscan.so; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT --output-format csv -d $O/$n -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/$n.json 2> $O/$n.log || echo "$n failed"
  python3 - "$n" "$O" <<'PY'
import csv, glob, sys, collections, json
n, O = sys.argv[1], sys.argv[2]
f = glob.glob(f"{O}/{n}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float); cnt = 0
for r in csv.DictReader(open(f)):
    if "inflate" in r["Kernel_Name"] and int(r["Grid_Size"]) > 6400:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt += r["Counter_Name"] == "SQ_WAVES"
b = json.loads(open(f"{O}/{n}.json").read().strip().splitlines()[-1])
print(n, "launches", cnt, {k: round(v / cnt / 1e9, 3) for k, v in agg.items()}, "G per launch; inflate ms (under pmc)", b["stage_ms"]["inflate"])
PY
done
cp /tmp/libbioscan_base.so $R/datafusion-bio-formats_amd/libbioscan.so
