#!/bin/bash
# dev tool: K1 time on 16384 members for the ';'-separated build flags in $CFGS_STR (ablation builds: outputs may be wrong on
# purpose, CRC is not checked; only the kernel time is meaningful)
make -C tools >/dev/null 2>&1
tools/_build/synth_bam /tmp/ab.bam 16384 42 >/dev/null 2>&1
C=datafusion-bio-formats_amd/csrc
# the product library is rebuilt in place per configuration: whatever ends this script, the default build comes back
trap 'touch $C/inflate_v3.hip; make -C $C >/dev/null 2>&1' EXIT
IFS=';' read -ra CFGS <<< "${CFGS_STR:-}"
[ ${#CFGS[@]} -eq 0 ] && CFGS=("")
for cfg in "${CFGS[@]}"; do
  touch $C/inflate_v3.hip; make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg] BUILD FAILED"; continue; }
  python - "$cfg" <<'PY'
import sys
sys.path.insert(0, 'tests')
from conftest import load_pkg
pkg = load_pkg()
data = open('/tmp/ab.bam','rb').read()
ms = min(pkg.bgzf_inflate(data, check_crc=False)[1] for _ in range(4))
print('cfg [%s] k1_ms %.3f' % (sys.argv[1], ms))
PY
done
touch $C/inflate_v3.hip; make -C $C >/dev/null 2>&1
