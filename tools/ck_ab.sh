#!/bin/bash
# LDS checkpoints against HBM checkpoints: headline workload and a batch-size sweep, interleaved (run on the GPU box)
for round in 1 2; do
  for o in 1 0; do
    echo "== lds_ckpt $o (round $round)"
    NS="${NS:-8192 16384 32768 100000}" bash tools/batch_sweep.sh --opt lds_ckpt=$o
  done
done
