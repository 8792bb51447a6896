#!/bin/bash
# pass 2 inside the fill launch against pass 2 as its own launch (run on the GPU box), interleaved
for round in 1 2; do
  for o in 1 0; do
    echo "== fused_trace $o (round $round)"
    NS="${NS:-8192 16384 32768 100000}" bash tools/batch_sweep.sh --opt fused_trace=$o
  done
done
