#!/bin/bash
# One bench line WITH cpu_baseline (the reference's own align_db on this box's host cores, subsample sizes of BASELINE.md section 3)
# per BASELINE.json workload shape + the 32-row classes (test/test.sh:70 runs RNA at -q 500).  Run on the GPU box, after
# `tools/profile_all.sh <tag>` with its summaries copied into profiles/, so that the lines quote counter traffic of THIS build:
#   bash tools/bench_configs.sh <round tag>   ->  gpurun_out/<tag>_bench_<workload>.json
R=${1:?round tag}
run() {  # workload cpu_reads [bench args...]
  local w=$1 c=$2; shift 2
  timeout -k 10 600 python bench.py --workload $w --cpu-reads $c --no-e2e --steps 5 --warmup 1 "$@" 2> gpurun_out/${R}_bench_$w.err | tail -1 > gpurun_out/${R}_bench_$w.json
  python -c "
import json
d = json.load(open('gpurun_out/${R}_bench_$w.json'))
print('$w', d['value'], 'reads/s  %.3e cells/s' % d['dp_cells_per_s'], 'valu frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'],
      'parity', d['cpu_baseline']['parity_on_sample'], 'traffic', d['roofline']['traffic'], flush=True)"
}
run ncov_r9_dna_q250 2048
run sequin_r9_rna_q250 2048
run r10_dna_1mb_q250 64 --reads 125000
run rna004_fullref_dtwstd_q250 1024
run ncov_r9_dna_q500 1024
run ncov_r9_dna_q1000 512
