#!/bin/bash
# row strips (queries beyond 2048 events): pipelined pass 1 against one wave per (read, job); run on the GPU box
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e --steps 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['reads_per_gpu'], 'reads/s', d['value'], 'cells/s %.3e' % d['dp_cells_per_s'], 'fill', d['roofline']['kernel_ms_per_step'], 'trace', d['roofline']['trace_kernel_ms_per_step'], flush=True)"; }
for o in 1 0; do
  echo "== strip_pipeline $o"
  run --workload ncov_r9_dna_q3000 --opt strip_pipeline=$o
  run --workload ncov_r9_dna_q4000 --opt strip_pipeline=$o
  run --workload ncov_r9_dna_q8000 --opt strip_pipeline=$o
done
