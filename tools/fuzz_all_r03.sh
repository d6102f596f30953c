#!/bin/bash
# dev tool (GPU box): the fuzz campaigns on the final binary, K0 off (default) and on; every python process prints its own
# totals, a failure stops the chain
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fuzz_r03
mkdir -p $O
set -e
for k in 0 1; do
  export BIOSCAN_K1_PREHEADERS=$k
  echo "== BIOSCAN_K1_PREHEADERS=$k"
  timeout -k 10 300 python3 $R/tools/fuzz_k1.py 30 $((3000 + k * 100)) 2>&1 | tail -2
  timeout -k 10 300 python3 $R/tools/fuzz_k1_corrupt.py 600 $((11 + k)) 2>&1 | tail -2
  timeout -k 10 300 python3 $R/tools/fuzz_bam_parity.py 90 $((21 + k)) 2>&1 | tail -2
done
unset BIOSCAN_K1_PREHEADERS
timeout -k 10 200 python3 $R/tools/fuzz_w2.py 60 5 2>&1 | tail -2
timeout -k 10 200 python3 $R/tools/fuzz_bam_indexed.py 90 7 2>&1 | tail -1
timeout -k 10 200 python3 $R/tools/fuzz_vcf_parity.py 90 7 2>&1 | tail -1
