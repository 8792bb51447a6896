#!/bin/bash
# fill time against batch size for several settings of the fill's issue-priority unit (0 = off); run on the GPU box
for u in ${US:-0 512 1024 2048 4096 8192}; do
  echo "== prio_unit $u"
  NS="${NS:-8192 12288 16384 32768 100000}" bash tools/batch_sweep.sh --opt prio_unit=$u "$@"
done
