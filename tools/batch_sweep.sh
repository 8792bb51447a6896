#!/bin/bash
# fill / step time against batch size with the planner's defaults (run on the GPU box); extra bench.py args pass through
# usage: tools/batch_sweep.sh [--opt key=value ...]
for n in ${NS:-2048 4096 8192 12288 16384 24576 32768 65536 100000 200000}; do
  timeout -k 10 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --reads $n "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reads', $n, 'reads/s', d['value'], 'ms/batch', d['ms_per_step'], 'fill', d['roofline']['kernel_ms_per_step'], 'trace', d['roofline']['trace_kernel_ms_per_step'], 'linear_fill_0.74us', round($n*0.00074,2))"
done
