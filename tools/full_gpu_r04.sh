#!/bin/bash
# the round's GPU suite + the default bench (what the driver runs), logs under gpurun_out/
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/gpu_suite.txt 2>&1; rc=$?
tail -5 $R/gpurun_out/gpu_suite.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err && python - <<PY
import json
d=json.loads(open("$R/gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "stage_ms", d["stage_ms"], "roofline frac", d["roofline"]["frac"], "e2e", (d.get("end_to_end") or {}).get("Mrec_s"))
PY
