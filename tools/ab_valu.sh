#!/bin/bash
# dev tool: SQ_INSTS_VALU / SQ_BUSY_CYCLES of K1 per phase -- the ablation builds (see ab_cfg.sh) under a --pmc pass
# (16384 members; outputs of the ablated builds are wrong on purpose, only the counters are meaningful)
R=$GRAFT_REPO_ROOT
make -C $R/tools >/dev/null 2>&1
$R/tools/_build/synth_bam /tmp/ab.bam 16384 42 >/dev/null 2>&1
C=$R/datafusion-bio-formats_amd/csrc
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/inflate_v3.hip; make -C $C >/dev/null 2>&1' EXIT
cat > /tmp/ab_run.py <<PY
import sys
sys.path.insert(0, '$R/tests')
from conftest import load_pkg
pkg = load_pkg()
data = open('/tmp/ab.bam','rb').read()
for _ in range(2): ms = pkg.bgzf_inflate(data, check_crc=False)[1]
print('k1_ms', round(ms, 3))
PY
O=$R/gpurun_out/ab_valu.txt; : > $O
cd /tmp && export TMPDIR=/tmp
for ab in ${ABS:-0 1 2 3}; do
  touch $C/inflate_v3.hip; make -C $C EXTRA="$( [ "$ab" = 1 ] && echo -DV3_ABLATE_RESOLVE; [ "$ab" = 2 ] && echo -DV3_ABLATE_WRITE; [ "$ab" = 3 ] && echo "-DV3_ABLATE_RESOLVE -DV3_ABLATE_WRITE" ) $EXTRA_ALL" >/dev/null 2>&1
  rm -rf /tmp/abv$ab
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/abv$ab -- python3 /tmp/ab_run.py > /tmp/abv$ab.log 2>&1
  python3 - $ab >> $O <<'PY'
import csv, glob, collections, sys
ab = sys.argv[1]
for f in glob.glob("/tmp/abv%s/**/*counter_collection.csv" % ab, recursive=True):
    agg = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n += 1
    print("ablate=%s" % ab, {k: round(v / 1e6, 1) for k, v in agg.items()}, "rows", n)
PY
done
touch $C/inflate_v3.hip; make -C $C >/dev/null 2>&1
cat $O
