#!/bin/bash
# dev tool (GPU box): the fuzz campaigns on the round's binary -- K1 v4 with the header pre-pass (default) and without it, and
# K1 v3 (BIOSCAN_K1=3: the wide-table fallback as the only decoder); every python process prints its own totals, a failure
# stops the chain.  Output: gpurun_out/fuzz_r04.txt
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fuzz_r04.txt
: > $O
set -e
for cfg in "BIOSCAN_K1=4 BIOSCAN_K1_PREHEADERS=1" "BIOSCAN_K1=4 BIOSCAN_K1_PREHEADERS=0" "BIOSCAN_K1=3 BIOSCAN_K1_PREHEADERS=0"; do
  echo "== $cfg" | tee -a $O
  env $cfg timeout -k 10 300 python3 $R/tools/fuzz_k1.py ${N_K1:-40} ${SEED_K1:-5000} 2>&1 | tail -1 | tee -a $O
  env $cfg timeout -k 10 300 python3 $R/tools/fuzz_k1_corrupt.py ${N_CORRUPT:-1500} ${SEED_CORRUPT:-31} 2>&1 | tail -1 | tee -a $O
  env $cfg timeout -k 10 300 python3 $R/tools/fuzz_bam_parity.py ${T_BAM:-60} ${SEED_BAM:-41} 2>&1 | tail -1 | tee -a $O
done
timeout -k 10 200 python3 $R/tools/fuzz_w2.py 40 9 2>&1 | tail -1 | tee -a $O
timeout -k 10 300 python3 $R/tools/fuzz_bam_indexed.py ${T_IDX:-90} ${SEED_IDX:-17} 2>&1 | tail -1 | tee -a $O
timeout -k 10 300 python3 $R/tools/fuzz_vcf_parity.py ${T_VCF:-90} ${SEED_VCF:-19} 2>&1 | tail -1 | tee -a $O
timeout -k 10 300 python3 $R/tools/fuzz_fastq_parity.py ${T_FASTQ:-60} ${SEED_FASTQ:-23} 2>&1 | tail -1 | tee -a $O
