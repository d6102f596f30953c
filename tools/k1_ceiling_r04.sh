#!/bin/bash
# dev tool (GPU box): what the decode loops of K1 v4 sustain on gfx950 -- the kernel timed on 32768 config-2 members with its
# write and resolve phases compiled out (outputs are wrong on purpose; CRC is not checked; only the kernel time is meaningful),
# at the occupancies the LDS working set allows, next to the full kernel on the same box.  Symbols per member from the CPU
# (tools/experiments/deflate_stats.py) turn the times into symbol rates.  Output: gpurun_out/k1_ceiling_r04.txt
R=${GRAFT_REPO_ROOT:-.}
C=$R/datafusion-bio-formats_amd/csrc
O=$R/gpurun_out/k1_ceiling_r04.txt
mkdir -p $R/gpurun_out; : > $O
export BIOSCAN_K1=4
trap 'touch $C/inflate_v4.hip; make -C $C >/dev/null 2>&1' EXIT
make -C $R/tools >/dev/null 2>&1
$R/tools/_build/synth_bam /tmp/ceil.bam 32768 42 >/dev/null 2>&1 || { echo "generator failed" | tee -a $O; exit 1; }
python3 $R/tools/experiments/deflate_stats.py /tmp/ceil.bam 100 48 2>&1 | grep -E "^members|^symbols|^output|^lookups" | sed 's/^/cpu: /' >> $O
CFGS=(""
      "-DV4_ABLATE_RESOLVE"
      "-DV4_ABLATE_WRITE -DV4_ABLATE_RESOLVE"
      "-DV4_ABLATE_WRITE -DV4_ABLATE_RESOLVE -DV4_SUB_DW=8 -DV4_LCAP_N=64 -DV4_WIN_BYTES=1280"
      "-DV4_ABLATE_WRITE -DV4_ABLATE_RESOLVE -DV4_SUB_DW=8 -DV4_LCAP_N=64 -DV4_WIN_BYTES=1280 -DV4_WAVES_PER_EU=5")
for cfg in "${CFGS[@]}"; do
  touch $C/inflate_v4.hip
  make -C $C EXTRA="$cfg" >/dev/null 2>&1 || { echo "cfg [$cfg] BUILD FAILED" >> $O; continue; }
  BIOSCAN_DEBUG=1 timeout -k 10 300 python3 - "$cfg" >> $O 2>/tmp/ceil.err <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
from conftest import load_pkg
pkg = load_pkg()
data = open('/tmp/ceil.bam', 'rb').read()
ms = sorted(pkg.bgzf_inflate(data, check_crc=False)[1] for _ in range(5))
print('cfg [%s] k1_ms min %.3f median %.3f' % (sys.argv[1], ms[0], ms[2]))
PY
  grep -E "residency|decode passes" /tmp/ceil.err | tail -2 >> $O
done
cat $O
