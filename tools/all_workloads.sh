#!/bin/bash
# one bench line per BASELINE.json workload shape and per query-length class (run on the GPU box); the table in DESIGN.md section 4
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --steps 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['reads_per_gpu'], 'reads/s', d['value'], 'cells/s %.3e' % d['dp_cells_per_s'], 'fill', d['roofline']['kernel_ms_per_step'], 'trace', d['roofline']['trace_kernel_ms_per_step'], flush=True)"; }
run --workload ncov_r9_dna_q250
run --workload sequin_r9_rna_q250
run --workload rna004_fullref_dtwstd_q250
run --workload r10_dna_1mb_q250 --reads 125000
run --workload ncov_r9_dna_q500
run --workload ncov_r9_dna_q1000
run --workload ncov_r9_dna_q2000
run --workload ncov_r9_dna_q3000
run --workload ncov_r9_dna_q4000
run --workload ncov_r9_dna_q8000
