#!/bin/bash
# dev: SQ counter passes over the BAM bench (65536 members)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/bench.py --blocks 65536 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/$tag.log 2>&1 || echo "FAILED $tag"
done
ls -R $R/gpurun_out/pmc | head -50
