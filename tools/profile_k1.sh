#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof (copy what is to be kept into profiles/):
#   kernel-trace stats of the BAM bench at 65536 members, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and
#   the SQ instruction-issue counters that show what bounds K1.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
ARGS="$R/bench.py --blocks 65536 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ARGS > $O/stats.log 2>&1 || echo "stats failed"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
find $O -name "*.csv" | head -20
