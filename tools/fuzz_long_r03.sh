#!/bin/bash
# dev tool (GPU box): longer campaigns of every differential fuzzer with fresh seeds; prints one line per tool
R=${GRAFT_REPO_ROOT:-.}
S=${1:-500}
T=${2:-170}
timeout -k 10 $((T+60)) python3 $R/tools/fuzz_vcf_parity.py $T $((S+1)) 2>&1 | tail -1
timeout -k 10 $((T+60)) python3 $R/tools/fuzz_bam_indexed.py $T $((S+2)) 2>&1 | tail -1
timeout -k 10 $((T+60)) python3 $R/tools/fuzz_bam_parity.py $T $((S+3)) 2>&1 | tail -1
timeout -k 10 $((T+60)) python3 $R/tools/fuzz_k1_corrupt.py 4000 $((S+4)) 2>&1 | tail -1
BIOSCAN_K1_PREHEADERS=1 timeout -k 10 $((T+60)) python3 $R/tools/fuzz_k1_corrupt.py 4000 $((S+5)) 2>&1 | tail -1
timeout -k 10 $((T+60)) python3 $R/tools/fuzz_k1.py 60 $((S+6)) 2>&1 | tail -1
