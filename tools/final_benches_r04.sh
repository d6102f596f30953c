#!/bin/bash
# dev tool: the bench lines kept under profiles/r04 (run after tools/profile_r04.sh so that bench.py finds the PMC stamp)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r04
mkdir -p $O
cp $R/gpurun_out/prof_r04/k1_pmc_*_65536blocks.csv $R/gpurun_out/prof_r04/k1_pmc_source.sha256 $R/profiles/r04/ 2>/dev/null
python bench.py > $O/bench_full_r04.json 2> $O/bench_full_r04.err || echo "bam failed"
BIOSCAN_HOST_POOL_GB=4 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_full_hostpool4.json 2>/dev/null || echo "pool4 failed"
python bench.py --mode indexed --no-cpu-baseline --no-end-to-end > $O/bench_indexed.json 2>/dev/null || echo "indexed failed"
python bench.py --mode indexed --partition-threads 8 --no-cpu-baseline --no-end-to-end > $O/bench_indexed_threads8.json 2>/dev/null || echo "indexed8 failed"
python bench.py --projection chrom,start --no-cpu-baseline --no-end-to-end > $O/bench_proj_chrom_start.json 2>/dev/null || echo "proj failed"
python bench.py --projection count --no-cpu-baseline --no-end-to-end > $O/bench_proj_count.json 2>/dev/null || echo "count failed"
python bench.py --format fastq > $O/bench_fastq.json 2>/dev/null || echo "fastq failed"
python bench.py --format vcf-sites > $O/bench_vcf_sites.json 2>/dev/null || echo "vcf-sites failed"
python bench.py --format vcf-samples > $O/bench_vcf_samples.json 2>/dev/null || echo "vcf-samples failed"
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], d.get("stage_ms"), (d.get("end_to_end") or {}).get("Mrec_s"), d["roofline"]["frac"], d["roofline"].get("traffic"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
