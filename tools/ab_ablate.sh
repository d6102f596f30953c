#!/bin/bash
# dev tool: what each K1 phase really costs in time (the BIOSCAN_DEBUG anatomy is in wave-elapsed cycles, which over-weights
# latency-bound phases): a -DV2_ABLATE=n build, bit 1 skips resolve + window flush, bit 2 skips the write pass (outputs are then
# wrong; only the kernel time is meaningful)
make -C tools >/dev/null 2>&1
tools/_build/synth_bam /tmp/ab.bam 16384 42 >/dev/null 2>&1 || tools/_build/synth_bam --help
C=datafusion-bio-formats_amd/csrc
for ab in ${ABS:-0 1 2 3}; do
  touch $C/inflate_v3.hip; make -C $C EXTRA="$( [ "$ab" = 1 ] && echo -DV3_ABLATE_RESOLVE; [ "$ab" = 2 ] && echo -DV3_ABLATE_WRITE; [ "$ab" = 3 ] && echo "-DV3_ABLATE_RESOLVE -DV3_ABLATE_WRITE" )" >/dev/null 2>&1   # ablation is a compile-time build: never the shipped library
  python - <<PY
import sys, importlib.util, os
sys.path.insert(0, 'tests')
from conftest import load_pkg
pkg = load_pkg()
data = open('/tmp/ab.bam','rb').read()
ms = min(pkg.bgzf_inflate(data, check_crc=False)[1] for _ in range(3))
print('ablate=$ab k1_ms', round(ms, 3))
PY
done
touch $C/inflate_v3.hip; make -C $C >/dev/null 2>&1
