#!/bin/bash
# batch-size sweep of the alignment stage (run on the GPU box): reads per batch x lane widening
for n in 256 512 1024 2048 4096 8192 16384 32768 65536; do
  for w in 1 2 4 0; do
    timeout -k 10 100 python bench.py --steps 5 --no-cpu-baseline --no-e2e --reads $n --opt lane_widening=$w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reads', $n, 'w', $w, 'reads/s', d['value'], 'ms/batch', d['ms_per_step'], 'fill', d['roofline']['kernel_ms_per_step'], 'trace', d['roofline']['trace_kernel_ms_per_step'])"
  done
done
