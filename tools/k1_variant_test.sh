#!/bin/bash
# dev tool (GPU box): parity tests of the inflate path with a library variant swapped in
R=${GRAFT_REPO_ROOT:-.}
cp $R/datafusion-bio-formats_amd/libbioscan.so /tmp/libbioscan_keep.so
cp $R/tools/_build/variants/$1/libbioscan.so $R/datafusion-bio-formats_amd/libbioscan.so
python -m pytest $R/tests/test_gpu_inflate_fuzz.py $R/tests/test_gpu_bam_parity.py $R/tests/test_gpu_synth_parity.py -x -q 2>&1 | tail -3
cp /tmp/libbioscan_keep.so $R/datafusion-bio-formats_amd/libbioscan.so
